// Stand-in for <ros/ros.h> (oracle/ref_shim, test infrastructure): the factor layer of the reference touches ROS only for
// parameters with defaults (Preintegration.h:47-51: nh.param<double>("/IMU/acc_n", acc_n, 0.00059) ...) and for log macros.
// NodeHandle::param returns the DEFAULT unless a value was registered through glio_ref_shim::params() -- which is how the test
// driver plays the role of the yaml file (config_urban_hk.yaml:7-11).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
namespace glio_ref_shim {
inline std::map<std::string, double>& params() { static std::map<std::string, double> p; return p; }
}
namespace ros {
class NodeHandle {
  public:
    NodeHandle() {}
    explicit NodeHandle(const std::string&) {}
    template <class T> bool param(const std::string& name, T& value, const T& def) const {
        auto it = glio_ref_shim::params().find(name);
        if (it != glio_ref_shim::params().end()) { value = T(it->second); return true; }
        value = def; return false;
    }
    template <class T> bool getParam(const std::string& name, T& value) const {
        auto it = glio_ref_shim::params().find(name);
        if (it == glio_ref_shim::params().end()) return false;
        value = T(it->second); return true;
    }
};
namespace this_node { inline std::string getName() { return "glio_ref_shim"; } }
namespace param {
inline bool search(const std::string& name, std::string& key) { key = name; return glio_ref_shim::params().count(name) != 0; }
inline bool has(const std::string& key) { return glio_ref_shim::params().count(key) != 0; }
template <class T> bool get(const std::string& key, T& v) { auto it = glio_ref_shim::params().find(key); if (it == glio_ref_shim::params().end()) return false; v = T(it->second); return true; }
inline bool get(const std::string&, std::string&) { return false; }
}
struct Time { double t; Time() : t(0) {} explicit Time(double s) : t(s) {} double toSec() const { return t; } static Time now() { return Time(); } };
}  // namespace ros
#include <ros/assert.h>
#define ROS_INFO(...) do { } while (0)
#define ROS_DEBUG(...) do { } while (0)
#define ROS_WARN(...) do { std::fprintf(stderr, "[ref ROS_WARN] " __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)
#define ROS_ERROR(...) do { std::fprintf(stderr, "[ref ROS_ERROR] " __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)
#define ROS_INFO_STREAM(x) do { } while (0)
#define ROS_WARN_STREAM(x) do { } while (0)
#define ROS_ERROR_STREAM(x) do { } while (0)
#define ROS_DEBUG_STREAM(x) do { } while (0)
