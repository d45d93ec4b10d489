// Stand-in for <glog/logging.h> (oracle/ref_shim, test infrastructure): LOG(x) << ... streams are swallowed, LOG(FATAL) aborts.
#pragma once
#include <cstdlib>
#include <iostream>
#include <sstream>
namespace glio_ref_shim {
struct LogSink { bool fatal; explicit LogSink(bool f) : fatal(f) {} ~LogSink() { if (fatal) { std::cerr << "[ref LOG(FATAL)] " << ss.str() << std::endl; std::abort(); } }
    std::ostringstream ss; template <class T> LogSink& operator<<(const T& v) { ss << v; return *this; }
    LogSink& operator<<(std::ostream& (*f)(std::ostream&)) { ss << f; return *this; } };
}
#define GLIO_REF_LOG_INFO false
#define GLIO_REF_LOG_WARNING false
#define GLIO_REF_LOG_ERROR false
#define GLIO_REF_LOG_FATAL true
#define LOG(sev) glio_ref_shim::LogSink(GLIO_REF_LOG_##sev)
#define LOG_IF(sev, c) if (c) glio_ref_shim::LogSink(GLIO_REF_LOG_##sev)
#define CHECK(c) if (!(c)) glio_ref_shim::LogSink(true) << "CHECK failed: " #c " "
#define CHECK_EQ(a, b) CHECK((a) == (b))
#define CHECK_NE(a, b) CHECK((a) != (b))
#define CHECK_LT(a, b) CHECK((a) < (b))
#define CHECK_LE(a, b) CHECK((a) <= (b))
#define CHECK_GT(a, b) CHECK((a) > (b))
#define CHECK_GE(a, b) CHECK((a) >= (b))
#define CHECK_NOTNULL(p) (p)
#define DCHECK(c) CHECK(c)
