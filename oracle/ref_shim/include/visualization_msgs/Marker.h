// empty stand-in: the reference header utils/common.h includes <visualization_msgs/Marker.h>, the factor layer uses nothing of it (oracle/ref_shim, test infrastructure)
#pragma once
