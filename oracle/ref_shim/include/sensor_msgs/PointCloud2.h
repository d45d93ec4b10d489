// empty stand-in: the reference header utils/common.h includes <sensor_msgs/PointCloud2.h>, the factor layer uses nothing of it (oracle/ref_shim, test infrastructure)
#pragma once
