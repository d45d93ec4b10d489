// empty stand-in: the reference header utils/common.h includes <tf/transform_listener.h>, the factor layer uses nothing of it (oracle/ref_shim, test infrastructure)
#pragma once
