// empty stand-in: the reference header utils/common.h includes <pcl/common/common.h>, the factor layer uses nothing of it (oracle/ref_shim, test infrastructure)
#pragma once
