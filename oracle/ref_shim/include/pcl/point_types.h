// Stand-in for <pcl/point_types.h> (oracle/ref_shim, test infrastructure): utils/common.h only names PointXYZI and registers two structs.
#pragma once
#include <cstdint>
#include <pcl/pcl_macros.h>
namespace pcl {
struct PointXYZ { float x, y, z, pad_; };
struct PointXYZI { float x, y, z, pad_; float intensity; float pad2_[3]; };      // 32 B as PCL lays it out
}
#define POINT_CLOUD_REGISTER_POINT_STRUCT(name, fseq)
