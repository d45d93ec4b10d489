/*
 * orc_math.h -- tiny fixed-size linear algebra + quaternion helpers for the CPU ORACLE
 * (test infrastructure).  The quaternion routines restate the Eigen 3.3 formulas the reference
 * relies on (Eigen::Quaternion::operator*, inverse(), _transformVector, toRotationMatrix) and the
 * helpers of GLIO/include/utils/math_tools.h (Qleft :36-42, Qright :45-51, deltaQ :126-138,
 * LeftQuatMatrix :141-150).  Quaternions are (w,x,y,z).
 */
#ifndef ORC_MATH_H_
#define ORC_MATH_H_

#include <math.h>
#include <string.h>

static inline void v3_cross(const double a[3], const double b[3], double o[3]) {
    double x = a[1] * b[2] - a[2] * b[1];
    double y = a[2] * b[0] - a[0] * b[2];
    double z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline double v3_dot(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline double v3_norm(const double a[3]) { return sqrt(v3_dot(a, a)); }

/* skewSymmetric, math_tools.h:25-33; row-major 3x3 */
static inline void skew3(const double v[3], double M[9]) {
    M[0] = 0; M[1] = -v[2]; M[2] = v[1];
    M[3] = v[2]; M[4] = 0; M[5] = -v[0];
    M[6] = -v[1]; M[7] = v[0]; M[8] = 0;
}

/* C[m x n] = A[m x k] * B[k x n], row-major, no aliasing */
static inline void mat_mul(const double* A, const double* B, double* C, int m, int k, int n) {
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0;
            for (int l = 0; l < k; ++l) s += A[i * k + l] * B[l * n + j];
            C[i * n + j] = s;
        }
}
static inline void mat3_vec(const double M[9], const double v[3], double o[3]) {
    double x = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
    double y = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
    double z = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline void mat3_T(const double M[9], double T[9]) {
    double t[9] = {M[0], M[3], M[6], M[1], M[4], M[7], M[2], M[5], M[8]};
    memcpy(T, t, sizeof t);
}

/* Eigen quaternion product a*b */
static inline void q_mul(const double a[4], const double b[4], double o[4]) {
    double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    double y = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
    double z = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
    o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}
/* Eigen inverse(): conjugate / squaredNorm (valid for non-unit quaternions, used on the
 * non-normalised corrected_delta_q of ImuFactor.h:86-89) */
static inline void q_inv(const double q[4], double o[4]) {
    double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    o[0] = q[0] / n2; o[1] = -q[1] / n2; o[2] = -q[2] / n2; o[3] = -q[3] / n2;
}
static inline void q_normalize(double q[4]) {
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
/* Eigen QuaternionBase::_transformVector: v + w*2(u x v) + u x 2(u x v)  (assumes |q|=1, which is
 * exactly what Ceres Jets differentiate through in LidarPlaneNormFactor) */
static inline void q_rot(const double q[4], const double v[3], double o[3]) {
    double uv[3], uuv[3];
    v3_cross(q + 1, v, uv);
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    v3_cross(q + 1, uv, uuv);
    o[0] = v[0] + q[0] * uv[0] + uuv[0];
    o[1] = v[1] + q[0] * uv[1] + uuv[1];
    o[2] = v[2] + q[0] * uv[2] + uuv[2];
}
/* Eigen toRotationMatrix, row-major */
static inline void q_to_R(const double q[4], double R[9]) {
    double w = q[0], x = q[1], y = q[2], z = q[3];
    double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    double twx = tx * w, twy = ty * w, twz = tz * w;
    double txx = tx * x, txy = ty * x, txz = tz * x;
    double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
/* Qleft(q), math_tools.h:36-42: [w -u^T; u wI+[u]x], 4x4 row-major */
static inline void q_left(const double q[4], double M[16]) {
    double w = q[0], x = q[1], y = q[2], z = q[3];
    double m[16] = {w, -x, -y, -z,
                    x, w, -z, y,
                    y, z, w, -x,
                    z, -y, x, w};
    memcpy(M, m, sizeof m);
}
/* Qright(p), math_tools.h:45-51: [w -u^T; u wI-[u]x] */
static inline void q_right(const double p[4], double M[16]) {
    double w = p[0], x = p[1], y = p[2], z = p[3];
    double m[16] = {w, -x, -y, -z,
                    x, w, z, -y,
                    y, -z, w, x,
                    z, y, -x, w};
    memcpy(M, m, sizeof m);
}
/* deltaQ(theta) = (1, theta/2), NOT normalised, math_tools.h:126-138 */
static inline void delta_q(const double th[3], double q[4]) {
    q[0] = 1.0; q[1] = th[0] / 2.0; q[2] = th[1] / 2.0; q[3] = th[2] / 2.0;
}

/* In-place Cholesky of a dense SPD n x n row-major matrix (lower triangle L, upper untouched).
 * Returns 0 on success, -1 if a pivot is not positive / not finite. */
static inline int chol_lower(double* A, int n) {
    for (int j = 0; j < n; ++j) {
        double d = A[j * n + j];
        for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0.0) || !isfinite(d)) return -1;
        d = sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    return 0;
}
/* solve L L^T x = b with L from chol_lower; x may alias b */
static inline void chol_solve(const double* L, int n, const double* b, double* x) {
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[i * n + k] * x[k];
        x[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = x[i];
        for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * x[k];
        x[i] = s / L[i * n + i];
    }
}

/* General inverse by Gauss-Jordan with partial pivoting (Eigen uses PartialPivLU for 15x15
 * Matrix::inverse()).  A is n x n row-major, overwritten; Ainv output.  Returns 0 / -1. */
static inline int mat_inverse(double* A, double* Ainv, int n) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) Ainv[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int c = 0; c < n; ++c) {
        int piv = c;
        double best = fabs(A[c * n + c]);
        for (int r = c + 1; r < n; ++r)
            if (fabs(A[r * n + c]) > best) { best = fabs(A[r * n + c]); piv = r; }
        if (best == 0.0) return -1;
        if (piv != c)
            for (int j = 0; j < n; ++j) {
                double t = A[c * n + j]; A[c * n + j] = A[piv * n + j]; A[piv * n + j] = t;
                t = Ainv[c * n + j]; Ainv[c * n + j] = Ainv[piv * n + j]; Ainv[piv * n + j] = t;
            }
        double d = A[c * n + c];
        for (int j = 0; j < n; ++j) { A[c * n + j] /= d; Ainv[c * n + j] /= d; }
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            double f = A[r * n + c];
            if (f == 0.0) continue;
            for (int j = 0; j < n; ++j) { A[r * n + j] -= f * A[c * n + j]; Ainv[r * n + j] -= f * Ainv[c * n + j]; }
        }
    }
    return 0;
}

#endif
