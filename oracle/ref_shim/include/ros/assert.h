// Stand-in for <ros/assert.h> (oracle/ref_shim, test infrastructure)
#pragma once
#include <cassert>
#include <cstdlib>
#define ROS_ASSERT(c) assert(c)
#define ROS_ASSERT_MSG(c, ...) assert(c)
#define ROS_BREAK() std::abort()
