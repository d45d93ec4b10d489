// empty stand-in: the reference header utils/common.h includes <pcl_conversions/pcl_conversions.h>, the factor layer uses nothing of it (oracle/ref_shim, test infrastructure)
#pragma once
