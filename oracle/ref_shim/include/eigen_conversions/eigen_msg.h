// empty stand-in: the reference header utils/common.h includes <eigen_conversions/eigen_msg.h>, the factor layer uses nothing of it (oracle/ref_shim, test infrastructure)
#pragma once
