// ceres/rotation.h stand-in (oracle/ref_shim, test infrastructure): LidarKeyframeFactor.h:6 includes it and uses nothing of it.
#pragma once
#include "ceres/jet.h"
