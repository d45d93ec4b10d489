// ceres/jet.h stand-in (oracle/ref_shim, TEST INFRASTRUCTURE).  Ceres Solver is a third-party dependency that is not under
// /root/reference (the tarball is a stripped blob; the bundled docs pin 1.14.0).  This is the published dual-number
// algebra of ceres::Jet<T, N> (include/ceres/jet.h of 1.14: f = a + v.eps, eps^2 = 0; the quotient and square-root forms below
// follow that header's formulas) restated for the operations the reference's functors use.
#pragma once
#include <cmath>
#include <limits>
#include <ostream>
namespace ceres {
template <typename T, int N> struct Jet {
    enum { DIMENSION = N };
    typedef T Scalar;
    T a; T v[N];
    Jet() : a() { for (int i = 0; i < N; ++i) v[i] = T(); }
    Jet(const T& value) : a(value) { for (int i = 0; i < N; ++i) v[i] = T(); }        // (explicit in Ceres; implicit here so that `0` literals of the comma initialiser convert)
    Jet(const T& value, int k) : a(value) { for (int i = 0; i < N; ++i) v[i] = T(); v[k] = T(1.0); }
    template <class I, typename std::enable_if<std::is_integral<I>::value, int>::type = 0> Jet(I value) : a(T(value)) { for (int i = 0; i < N; ++i) v[i] = T(); }
    Jet& operator+=(const Jet& y) { *this = *this + y; return *this; }
    Jet& operator-=(const Jet& y) { *this = *this - y; return *this; }
    Jet& operator*=(const Jet& y) { *this = *this * y; return *this; }
    Jet& operator/=(const Jet& y) { *this = *this / y; return *this; }
    explicit operator double() const { return (double)a; }
};
#define GLIO_JET_LOOP for (int i = 0; i < N; ++i)
template <typename T, int N> inline Jet<T, N> operator+(const Jet<T, N>& f) { return f; }
template <typename T, int N> inline Jet<T, N> operator-(const Jet<T, N>& f) { Jet<T, N> h; h.a = -f.a; GLIO_JET_LOOP h.v[i] = -f.v[i]; return h; }
template <typename T, int N> inline Jet<T, N> operator+(const Jet<T, N>& f, const Jet<T, N>& g) { Jet<T, N> h; h.a = f.a + g.a; GLIO_JET_LOOP h.v[i] = f.v[i] + g.v[i]; return h; }
template <typename T, int N> inline Jet<T, N> operator+(const Jet<T, N>& f, T s) { Jet<T, N> h(f); h.a = f.a + s; return h; }
template <typename T, int N> inline Jet<T, N> operator+(T s, const Jet<T, N>& f) { Jet<T, N> h(f); h.a = f.a + s; return h; }
template <typename T, int N> inline Jet<T, N> operator-(const Jet<T, N>& f, const Jet<T, N>& g) { Jet<T, N> h; h.a = f.a - g.a; GLIO_JET_LOOP h.v[i] = f.v[i] - g.v[i]; return h; }
template <typename T, int N> inline Jet<T, N> operator-(const Jet<T, N>& f, T s) { Jet<T, N> h(f); h.a = f.a - s; return h; }
template <typename T, int N> inline Jet<T, N> operator-(T s, const Jet<T, N>& f) { Jet<T, N> h; h.a = s - f.a; GLIO_JET_LOOP h.v[i] = -f.v[i]; return h; }
template <typename T, int N> inline Jet<T, N> operator*(const Jet<T, N>& f, const Jet<T, N>& g) { Jet<T, N> h; h.a = f.a * g.a; GLIO_JET_LOOP h.v[i] = f.a * g.v[i] + f.v[i] * g.a; return h; }
template <typename T, int N> inline Jet<T, N> operator*(const Jet<T, N>& f, T s) { Jet<T, N> h; h.a = f.a * s; GLIO_JET_LOOP h.v[i] = f.v[i] * s; return h; }
template <typename T, int N> inline Jet<T, N> operator*(T s, const Jet<T, N>& f) { Jet<T, N> h; h.a = f.a * s; GLIO_JET_LOOP h.v[i] = f.v[i] * s; return h; }
template <typename T, int N> inline Jet<T, N> operator/(const Jet<T, N>& f, const Jet<T, N>& g) {
    // (b + v eps) / (c + u eps) = b/c + (v - (b/c) u) / c eps      (the form jet.h of 1.14 evaluates)
    Jet<T, N> h; const T g_a_inverse = T(1.0) / g.a; const T f_a_by_g_a = f.a * g_a_inverse;
    h.a = f_a_by_g_a; GLIO_JET_LOOP h.v[i] = (f.v[i] - f_a_by_g_a * g.v[i]) * g_a_inverse; return h;
}
template <typename T, int N> inline Jet<T, N> operator/(T s, const Jet<T, N>& g) { Jet<T, N> h; h.a = s / g.a; const T minus_s_g_a_inverse2 = -s / (g.a * g.a); GLIO_JET_LOOP h.v[i] = g.v[i] * minus_s_g_a_inverse2; return h; }
template <typename T, int N> inline Jet<T, N> operator/(const Jet<T, N>& f, T s) { Jet<T, N> h; const T s_inverse = T(1.0) / s; h.a = f.a * s_inverse; GLIO_JET_LOOP h.v[i] = f.v[i] * s_inverse; return h; }
#define GLIO_JET_CMP(op) \
    template <typename T, int N> inline bool operator op(const Jet<T, N>& f, const Jet<T, N>& g) { return f.a op g.a; } \
    template <typename T, int N> inline bool operator op(const T& s, const Jet<T, N>& g) { return s op g.a; } \
    template <typename T, int N> inline bool operator op(const Jet<T, N>& f, const T& s) { return f.a op s; }
GLIO_JET_CMP(<) GLIO_JET_CMP(<=) GLIO_JET_CMP(>) GLIO_JET_CMP(>=) GLIO_JET_CMP(==) GLIO_JET_CMP(!=)
#undef GLIO_JET_CMP
template <typename T, int N> inline Jet<T, N> abs(const Jet<T, N>& f) { return f.a < T(0.0) ? -f : f; }
template <typename T, int N> inline Jet<T, N> sqrt(const Jet<T, N>& f) { Jet<T, N> h; const T tmp = std::sqrt(f.a); const T two_a_inverse = T(1.0) / (T(2.0) * tmp); h.a = tmp; GLIO_JET_LOOP h.v[i] = f.v[i] * two_a_inverse; return h; }
template <typename T, int N> inline Jet<T, N> cos(const Jet<T, N>& f) { Jet<T, N> h; h.a = std::cos(f.a); const T s = -std::sin(f.a); GLIO_JET_LOOP h.v[i] = s * f.v[i]; return h; }
template <typename T, int N> inline Jet<T, N> sin(const Jet<T, N>& f) { Jet<T, N> h; h.a = std::sin(f.a); const T c = std::cos(f.a); GLIO_JET_LOOP h.v[i] = c * f.v[i]; return h; }
template <typename T, int N> inline Jet<T, N> acos(const Jet<T, N>& f) { Jet<T, N> h; h.a = std::acos(f.a); const T tmp = -T(1.0) / std::sqrt(T(1.0) - f.a * f.a); GLIO_JET_LOOP h.v[i] = tmp * f.v[i]; return h; }
template <typename T, int N> inline Jet<T, N> asin(const Jet<T, N>& f) { Jet<T, N> h; h.a = std::asin(f.a); const T tmp = T(1.0) / std::sqrt(T(1.0) - f.a * f.a); GLIO_JET_LOOP h.v[i] = tmp * f.v[i]; return h; }
template <typename T, int N> inline Jet<T, N> atan2(const Jet<T, N>& g, const Jet<T, N>& f) { Jet<T, N> h; const T tmp = T(1.0) / (f.a * f.a + g.a * g.a); h.a = std::atan2(g.a, f.a); GLIO_JET_LOOP h.v[i] = tmp * (-g.a * f.v[i] + f.a * g.v[i]); return h; }
template <typename T, int N> inline Jet<T, N> exp(const Jet<T, N>& f) { Jet<T, N> h; const T tmp = std::exp(f.a); h.a = tmp; GLIO_JET_LOOP h.v[i] = tmp * f.v[i]; return h; }
template <typename T, int N> inline Jet<T, N> log(const Jet<T, N>& f) { Jet<T, N> h; const T a_inverse = T(1.0) / f.a; h.a = std::log(f.a); GLIO_JET_LOOP h.v[i] = f.v[i] * a_inverse; return h; }
template <typename T, int N> inline Jet<T, N> pow(const Jet<T, N>& f, double g) { Jet<T, N> h; const T tmp = g * std::pow(f.a, g - T(1.0)); h.a = std::pow(f.a, g); GLIO_JET_LOOP h.v[i] = tmp * f.v[i]; return h; }
template <typename T, int N> inline Jet<T, N> floor(const Jet<T, N>& f) { return Jet<T, N>(std::floor(f.a)); }
template <typename T, int N> inline bool isfinite(const Jet<T, N>& f) { if (!std::isfinite(f.a)) return false; GLIO_JET_LOOP if (!std::isfinite(f.v[i])) return false; return true; }
template <typename T, int N> inline std::ostream& operator<<(std::ostream& s, const Jet<T, N>& z) { s << "[" << z.a << " ; "; GLIO_JET_LOOP s << z.v[i] << (i + 1 < N ? ", " : ""); return s << "]"; }
#undef GLIO_JET_LOOP
inline double abs(double x) { return std::fabs(x); }
using std::sqrt; using std::sin; using std::cos; using std::acos; using std::asin; using std::atan2; using std::exp; using std::log; using std::pow; using std::floor; using std::isfinite;
}  // namespace ceres
