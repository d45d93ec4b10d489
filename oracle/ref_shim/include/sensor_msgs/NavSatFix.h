// empty stand-in: the reference header utils/common.h includes <sensor_msgs/NavSatFix.h>, the factor layer uses nothing of it (oracle/ref_shim, test infrastructure)
#pragma once
