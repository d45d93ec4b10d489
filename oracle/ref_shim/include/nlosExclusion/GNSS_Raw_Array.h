// nlosExclusion/msg/GNSS_Raw_Array.msg:1-3 as a plain struct (header dropped; GNSS_Raws_mf is not used by the factor layer) -- oracle/ref_shim, test infrastructure
#pragma once
#include <vector>
#include <nlosExclusion/GNSS_Raw.h>
namespace nlosExclusion {
struct GNSS_Raw_Array { std::vector<GNSS_Raw> GNSS_Raws; };
}
