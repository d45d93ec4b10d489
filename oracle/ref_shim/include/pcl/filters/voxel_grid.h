// empty stand-in: the reference header utils/common.h includes <pcl/filters/voxel_grid.h>, the factor layer uses nothing of it (oracle/ref_shim, test infrastructure)
#pragma once
