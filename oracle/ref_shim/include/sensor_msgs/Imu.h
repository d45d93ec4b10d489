// empty stand-in: the reference header utils/common.h includes <sensor_msgs/Imu.h>, the factor layer uses nothing of it (oracle/ref_shim, test infrastructure)
#pragma once
