// Stand-in for <pcl/pcl_macros.h> (oracle/ref_shim, test infrastructure)
#pragma once
#define PCL_ADD_POINT4D float x; float y; float z; float pad_
