/*
 * orc_batch2.c -- CPU ORACLE (test infrastructure): the COMPLETE batch problem of
 * Estimator::optimizeBatchWithLandMark with sms_fusion_level == 1 (GLIO/src/Estimator.cpp:2739-3410):
 *   parameter blocks per keyframe  gl_tmpTrans[3], gl_tmpQuat[4] (QuaternionParameterization), gl_tmpSpeedBias[9]   (:2809-2819)
 *   ImuFactor between every pair of consecutive keyframes                                                         (:2990-3001)
 *   BinaryLidarPlaneNormFactor blocks, no loss                                                                     (:3004-3076, :2768)
 *   delta_q_factor_auto attitude constraints                                                                       (:2831-2891)
 *   dd_psr_factor_20 per GNSS epoch                                                                                (:3197-3271)
 * and its solve: ceres::Solve with DOGLEG / SUBSPACE_DOGLEG / use_nonmonotonic_steps (:3275-3284), restated from the
 * Ceres 1.14 sources' published algorithm (trust_region_minimizer.cc, dogleg_strategy.cc, trust_region_step_evaluator.cc,
 * polynomial.cc; the bundled docs GraphGNSSLibV1.1/docs/source/nnls_solving.rst:83-260 describe the same loop):
 *   - DoglegStrategy::ComputeSubspaceModel: orthonormal basis of span{gradient, Gauss-Newton step} by a column-pivoted
 *     Householder QR, B = (J D^-1 U)^T (J D^-1 U), g = U^T gradient
 *   - ComputeSubspaceDoglegStep / FindMinimumOnTrustRegionBoundary: the quartic in the Lagrange multiplier
 *     (MakePolynomialForBoundaryConstrainedProblem), the real parts of ALL its roots as candidates, x(y) = -(B + y I)^-1 g by a
 *     partially pivoted 2x2 LU, the candidate with the least model value on the boundary
 *   - TrustRegionMinimizer: the user's parameters receive x only when its cost is below every cost seen so far
 *     (FinalizeIterationAndCheckIfMinimizerCanContinue), and final_cost is the minimum over the iterations
 *     (SetSummaryFinalCost) -- with non-monotonic steps the returned point is the BEST one, not the last one.
 * Unknown order of the dense system: keyframe-major, [dt3 dtheta3 | dv3 dba3 dbg3] (15) with the IMU chain, [dt3 dtheta3] (6)
 * without it (then the speed-bias blocks have no residual and Ceres drops them from the reduced program).
 * PARITY UNPINNED -- see glio_oracle.h.
 */
#include <complex.h>
#include <stdlib.h>
#include "glio_oracle.h"
#include "orc_math.h"

static void plusJ2(const double q[4], double P[12]) {
    P[0] = -q[1]; P[1] = -q[2]; P[2] = -q[3];
    P[3] = q[0];  P[4] = q[3];  P[5] = -q[2];
    P[6] = -q[3]; P[7] = q[0];  P[8] = q[1];
    P[9] = q[2];  P[10] = -q[1]; P[11] = q[0];
}

int orc_batch2_dim(const orc_batch_problem* p) { return (p->n_imu > 0 ? 15 : 6) * p->K; }

/* ---- symmetric band storage of the normal matrix: the LOWER band, row i holds columns i - hbw .. i at a[i * (hbw + 1) + (j - i + hbw)].
 * Half-bandwidth: pose blocks couple keyframes up to `band` apart (B * band + 5 scalar columns), the IMU edge couples all 15 states of
 * neighbours (29).  K = 2000 with the IMU chain: 30 000 x 96 doubles = 23 MB where the dense matrix would be 7.2 GB.  Every sum below runs
 * over the columns inside the band in ascending order, i.e. the dense loops minus their exact-zero terms: the banded solve returns the SAME
 * bits as the dense one did (tests/golden/batch*_small.npz did not move). */
typedef struct { int n, hbw; double* a; } bandm;
int orc_batch2_half_bandwidth(const orc_batch_problem* p) {
    const int B = p->n_imu > 0 ? 15 : 6, n = B * p->K;
    int h = B * p->band + 5;
    if (B == 15 && h < 29) h = 29;
    if (h > n - 1) h = n - 1;
    return h;
}
static inline double* bat(const bandm* m, int i, int j) { return m->a + (size_t)i * (m->hbw + 1) + (j - i + m->hbw); }      /* i >= j >= i - hbw */
static inline double bget(const bandm* m, int i, int j) {
    if (i < j) { const int t = i; i = j; j = t; }
    return (i - j > m->hbw) ? 0.0 : *bat(m, i, j);
}
static int band_alloc(bandm* m, int n, int hbw) { m->n = n; m->hbw = hbw; m->a = (double*)calloc((size_t)n * (hbw + 1), sizeof(double)); return m->a != NULL; }

/* banded H (lower band), g (n), cost of all factors at (poses, speed_bias) */
static int batch2_linearize_band(const orc_batch_problem* p, const double* poses, const double* sb, bandm* H, double* g, double* cost_out) {
    const int K = p->K, band = p->band, B = p->n_imu > 0 ? 15 : 6, n = B * K;
    const size_t hb = (size_t)K * (band + 1) * 36;
    double* Hb = (double*)malloc(sizeof(double) * hb);
    double* gb = (double*)malloc(sizeof(double) * 6 * (size_t)K);
    double cost = 0;
    int ok = orc_batch_linearize_full(p, poses, Hb, gb, &cost);
    if (ok) {
        memset(H->a, 0, sizeof(double) * (size_t)n * (H->hbw + 1));
        memset(g, 0, sizeof(double) * n);
        for (int k = 0; k < K; ++k) {
            for (int r = 0; r < 6; ++r) g[B * k + r] = gb[6 * k + r];
            for (int d = 0; d <= band && k + d < K; ++d) {
                const double* blk = Hb + ((size_t)k * (band + 1) + d) * 36;
                for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) {
                    const int row = B * k + r, col = B * (k + d) + c;         /* upper entry (row, col); the lower band holds (col, row) */
                    if (col >= row) *bat(H, col, row) = blk[r * 6 + c];
                }
            }
        }
        /* ImuFactor (k, k+1): blocks Pi Qi SBi Pj Qj SBj; local columns [Pi3 thi3 SBi9 | Pj3 thj3 SBj9] */
        glio_opts o;
        memset(&o, 0, sizeof o);
        o.gravity = p->gravity;
        for (int e = 0; e < p->n_imu && ok; ++e) {
            const double* P[6] = {poses + 7 * (size_t)e, poses + 7 * (size_t)e + 3, sb + 9 * (size_t)e,
                                  poses + 7 * (size_t)(e + 1), poses + 7 * (size_t)(e + 1) + 3, sb + 9 * (size_t)(e + 1)};
            double r[15], J0[45], J1[60], J2[135], J3[45], J4[60], J5[135];
            double* J[6] = {J0, J1, J2, J3, J4, J5};
            if (!orc_eval_imu(&o, &p->imu[e], P, r, J)) { ok = 0; break; }
            double Pi[12], Pj[12], Jl[15 * 30];
            plusJ2(P[1], Pi); plusJ2(P[4], Pj);
            for (int i = 0; i < 15; ++i) {
                double* row = Jl + i * 30;
                for (int c = 0; c < 3; ++c) {
                    row[c] = J0[i * 3 + c];
                    row[3 + c] = J1[i * 4] * Pi[c] + J1[i * 4 + 1] * Pi[3 + c] + J1[i * 4 + 2] * Pi[6 + c] + J1[i * 4 + 3] * Pi[9 + c];
                    row[15 + c] = J3[i * 3 + c];
                    row[18 + c] = J4[i * 4] * Pj[c] + J4[i * 4 + 1] * Pj[3 + c] + J4[i * 4 + 2] * Pj[6 + c] + J4[i * 4 + 3] * Pj[9 + c];
                }
                for (int c = 0; c < 9; ++c) { row[6 + c] = J2[i * 9 + c]; row[21 + c] = J5[i * 9 + c]; }
            }
            for (int i = 0; i < 15; ++i) cost += 0.5 * r[i] * r[i];
            const int base = 15 * e;                      /* the 30 columns are contiguous: keyframes e and e+1 */
            for (int u = 0; u < 30; ++u) {
                double gu = 0;
                for (int i = 0; i < 15; ++i) gu += Jl[i * 30 + u] * r[i];
                g[base + u] += gu;
                for (int v = 0; v <= u; ++v) {
                    double s = 0;
                    for (int i = 0; i < 15; ++i) s += Jl[i * 30 + u] * Jl[i * 30 + v];
                    *bat(H, base + u, base + v) += s;
                }
            }
        }
    }
    *cost_out = cost;
    free(Hb); free(gb);
    return ok;
}
/* the same with the lower band handed out: Hband [n][hbw + 1], hbw = orc_batch2_half_bandwidth (the K = 2000 checks of the GPU suite) */
int orc_batch2_linearize_banded(const orc_batch_problem* p, const double* poses, const double* sb, double* Hband, double* g, double* cost_out) {
    bandm H;
    H.n = orc_batch2_dim(p); H.hbw = orc_batch2_half_bandwidth(p); H.a = Hband;
    return batch2_linearize_band(p, poses, sb, &H, g, cost_out);
}
/* dense H (n x n), g (n), cost: the band expanded (small problems; tests) */
int orc_batch2_linearize(const orc_batch_problem* p, const double* poses, const double* sb, double* H, double* g, double* cost_out) {
    const int n = orc_batch2_dim(p);
    bandm Hb;
    if (!band_alloc(&Hb, n, orc_batch2_half_bandwidth(p))) return 0;
    const int ok = batch2_linearize_band(p, poses, sb, &Hb, g, cost_out);
    if (ok) for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) H[(size_t)i * n + j] = bget(&Hb, i, j);
    free(Hb.a);
    return ok;
}

/* scaled entry S H S of the band, evaluated exactly as the dense code did: (scale_i * H_ij) * scale_j */
static inline double hs_at(const bandm* H, const double* scale, int i, int j) { return scale[i] * bget(H, i, j) * scale[j]; }
/* y = (S H S) x */
static void band_scaled_matvec(const bandm* H, const double* scale, const double* x, double* y) {
    const int n = H->n, w = H->hbw;
    for (int i = 0; i < n; ++i) {
        const int j0 = i - w < 0 ? 0 : i - w, j1 = i + w > n - 1 ? n - 1 : i + w;
        double s = 0;
        for (int j = j0; j <= j1; ++j) s += hs_at(H, scale, i, j) * x[j];
        y[i] = s;
    }
}
/* banded Cholesky of the lower band in place (the dense chol_lower restricted to the band); 0 / -1 */
static int band_chol(bandm* A) {
    const int n = A->n, w = A->hbw;
    for (int j = 0; j < n; ++j) {
        double d = *bat(A, j, j);
        for (int k = j - w < 0 ? 0 : j - w; k < j; ++k) d -= *bat(A, j, k) * *bat(A, j, k);
        if (!(d > 0.0) || !isfinite(d)) return -1;
        d = sqrt(d);
        *bat(A, j, j) = d;
        const int i1 = j + w > n - 1 ? n - 1 : j + w;
        for (int i = j + 1; i <= i1; ++i) {
            double s = *bat(A, i, j);
            for (int k = i - w < 0 ? 0 : i - w; k < j; ++k) s -= *bat(A, i, k) * *bat(A, j, k);
            *bat(A, i, j) = s / d;
        }
    }
    return 0;
}
static void band_chol_solve(const bandm* L, const double* b, double* x) {
    const int n = L->n, w = L->hbw;
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = i - w < 0 ? 0 : i - w; k < i; ++k) s -= *bat(L, i, k) * x[k];
        x[i] = s / *bat(L, i, i);
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = x[i];
        const int k1 = i + w > n - 1 ? n - 1 : i + w;
        for (int k = i + 1; k <= k1; ++k) s -= *bat(L, k, i) * x[k];
        x[i] = s / *bat(L, i, i);
    }
}

static void batch2_plus(const orc_batch_problem* p, const double* poses, const double* sb, const double* delta, double* poses_o, double* sb_o) {
    const int K = p->K, B = p->n_imu > 0 ? 15 : 6;
    for (int k = 0; k < K; ++k) {
        for (int c = 0; c < 3; ++c) poses_o[7 * k + c] = poses[7 * k + c] + delta[B * k + c];
        orc_quat_plus(poses + 7 * k + 3, delta + B * k + 3, poses_o + 7 * k + 3);
        if (B == 15) for (int c = 0; c < 9; ++c) sb_o[9 * k + c] = sb[9 * k + c] + delta[B * k + 6 + c];
    }
}

/* ---- polynomial.cc: real parts of all roots of c[0] x^deg + ... + c[deg] (leading coefficient first).  Ceres builds the
 * balanced companion matrix and takes its eigenvalues; here the same roots come from the Aberth-Ehrlich simultaneous iteration
 * (degree <= 2 in closed form, as FindLinear/QuadraticPolynomialRoots do). */
static int poly_roots_real(const double* c_in, int deg_in, double* re, int* n_out) {
    const double* c = c_in;
    int deg = deg_in;
    while (deg > 0 && c[0] == 0.0) { ++c; --deg; }              /* RemoveLeadingZeros */
    *n_out = deg;
    if (deg == 0) return 1;
    if (deg == 1) { re[0] = -c[1] / c[0]; return 1; }
    if (deg == 2) {
        const double a = c[0], b = c[1], cc = c[2], D = b * b - 4 * a * cc, sD = sqrt(fabs(D));
        if (D >= 0) {
            if (b >= 0) { re[0] = (-b - sD) / (2.0 * a); re[1] = (2.0 * cc) / (-b - sD); }
            else { re[0] = (2.0 * cc) / (-b + sD); re[1] = (-b + sD) / (2.0 * a); }
        } else { re[0] = -b / (2.0 * a); re[1] = -b / (2.0 * a); }
        return 1;
    }
    if (deg > 8) return 0;
    double m[9];
    for (int i = 0; i <= deg; ++i) { m[i] = c[i] / c[0]; if (!isfinite(m[i])) return 0; }
    double bound = 0;                                           /* Cauchy: |z| <= 1 + max |m_i| */
    for (int i = 1; i <= deg; ++i) if (fabs(m[i]) > bound) bound = fabs(m[i]);
    bound = 1.0 + bound;
    /* a tighter start radius: max_i (deg |m_i|)^(1/i) */
    double rad = 0;
    for (int i = 1; i <= deg; ++i) { const double t = pow(deg * fabs(m[i]), 1.0 / i); if (t > rad) rad = t; }
    if (!(rad > 0)) { for (int i = 0; i < deg; ++i) re[i] = 0.0; return 1; }
    if (rad > bound) rad = bound;
    double complex z[8];
    for (int i = 0; i < deg; ++i) z[i] = rad * cexp(I * (2.0 * M_PI * i / deg + 0.4));
    for (int it = 0; it < 100; ++it) {
        double move = 0, size = 0;
        for (int i = 0; i < deg; ++i) {
            double complex pv = 1.0, dv = 0.0;                 /* Horner: p and p' */
            for (int k = 1; k <= deg; ++k) { dv = dv * z[i] + pv; pv = pv * z[i] + m[k]; }
            if (cabs(pv) == 0.0) continue;
            double complex s = 0;
            for (int j = 0; j < deg; ++j) if (j != i) s += 1.0 / (z[i] - z[j]);
            const double complex nw = pv / dv;
            const double complex w = nw / (1.0 - nw * s);
            if (!isfinite(creal(w)) || !isfinite(cimag(w))) continue;
            z[i] -= w;
            if (cabs(w) > move) move = cabs(w);
            if (cabs(z[i]) > size) size = cabs(z[i]);
        }
        if (move <= 4e-15 * size) break;
    }
    for (int i = 0; i < deg; ++i) { re[i] = creal(z[i]); if (!isfinite(re[i])) return 0; }
    return 1;
}
/* test hook */
int orc_poly_roots_real(const double* coeffs, int degree, double* roots_real, int* n_roots) { return poly_roots_real(coeffs, degree, roots_real, n_roots); }

/* makeHouseholder (Eigen/src/Householder/Householder.h): x -> (beta, tau, essential) with H x = beta e1, H = I - tau v v^T, v = [1; essential] */
static void householder(double* x, int len, double* tau, double* beta) {
    double tail = 0;
    for (int i = 1; i < len; ++i) tail += x[i] * x[i];
    const double c0 = x[0];
    if (len <= 1 || tail <= 2.2250738585072014e-308) { *tau = 0; *beta = c0; for (int i = 1; i < len; ++i) x[i] = 0; return; }
    double b = sqrt(c0 * c0 + tail);
    if (c0 >= 0) b = -b;
    for (int i = 1; i < len; ++i) x[i] /= (c0 - b);
    *tau = (b - c0) / b;
    *beta = b;
}

typedef struct {
    int one_dim;
    double B[4], g[2];
    double* basis;      /* [n][2] */
} subspace_model;

/* DoglegStrategy::ComputeSubspaceModel.  S H S = the scaled normal matrix (band + scale vector), gradient = D^-1 g_s, gn = scaled Gauss-Newton step, D = diagonal */
static int compute_subspace_model(int n, const bandm* H, const double* scale, const double* diag, const double* gradient, const double* gn, subspace_model* M, double* work /* 4 n */) {
    double* c0 = work; double* c1 = work + n; double* t0 = work + 2 * n; double* t1 = work + 3 * n;
    /* ColPivHouseholderQR of [gradient gn]: the column with the larger norm first */
    double n0 = 0, n1 = 0;
    for (int i = 0; i < n; ++i) { n0 += gradient[i] * gradient[i]; n1 += gn[i] * gn[i]; }
    const int swap = n1 > n0;
    for (int i = 0; i < n; ++i) { c0[i] = swap ? gn[i] : gradient[i]; c1[i] = swap ? gradient[i] : gn[i]; }
    double tau0, beta0, tau1 = 0, beta1 = 0;
    householder(c0, n, &tau0, &beta0);
    {   /* apply H0 to the second column: c1 -= tau0 v (v^T c1), v = [1; essential(c0)] */
        double s = c1[0];
        for (int i = 1; i < n; ++i) s += c0[i] * c1[i];
        c1[0] -= tau0 * s;
        for (int i = 1; i < n; ++i) c1[i] -= tau0 * s * c0[i];
    }
    const double r01 = c1[0];
    (void)r01;
    householder(c1 + 1, n - 1, &tau1, &beta1);
    const double maxpivot = fabs(beta0) > fabs(beta1) ? fabs(beta0) : fabs(beta1);
    const double thr = maxpivot * (2.220446049250313e-16 * 2.0);
    const int rank = (fabs(beta0) > thr) + (fabs(beta1) > thr);
    if (rank == 0) return 0;
    if (rank == 1) { M->one_dim = 1; return 1; }
    M->one_dim = 0;
    /* basis = Q * I(n, 2) = H0 H1 [e1 e2] */
    for (int col = 0; col < 2; ++col) {
        double* q = col == 0 ? t0 : t1;
        for (int i = 0; i < n; ++i) q[i] = 0.0;
        q[col] = 1.0;
        {   /* H1 acts on rows 1.. with v = [1; essential(c1 + 1)] */
            double s = q[1];
            for (int i = 2; i < n; ++i) s += c1[i] * q[i];
            q[1] -= tau1 * s;
            for (int i = 2; i < n; ++i) q[i] -= tau1 * s * c1[i];
        }
        {
            double s = q[0];
            for (int i = 1; i < n; ++i) s += c0[i] * q[i];
            q[0] -= tau0 * s;
            for (int i = 1; i < n; ++i) q[i] -= tau0 * s * c0[i];
        }
        for (int i = 0; i < n; ++i) M->basis[2 * i + col] = q[i];
    }
    M->g[0] = M->g[1] = 0;
    for (int i = 0; i < n; ++i) { M->g[0] += M->basis[2 * i] * gradient[i]; M->g[1] += M->basis[2 * i + 1] * gradient[i]; }
    /* B = (J D^-1 U)^T (J D^-1 U) = (D^-1 U)^T Hs (D^-1 U) */
    for (int i = 0; i < n; ++i) { c0[i] = M->basis[2 * i] / diag[i]; c1[i] = M->basis[2 * i + 1] / diag[i]; }
    band_scaled_matvec(H, scale, c0, t0);
    band_scaled_matvec(H, scale, c1, t1);
    double b00 = 0, b01 = 0, b11 = 0;
    for (int i = 0; i < n; ++i) { b00 += c0[i] * t0[i]; b01 += c0[i] * t1[i]; b11 += c1[i] * t1[i]; }
    M->B[0] = b00; M->B[1] = b01; M->B[2] = b01; M->B[3] = b11;
    return 1;
}

/* ComputeSubspaceStepFromRoot: -(B + y I)^-1 g by partialPivLu */
static void subspace_step_from_root(const subspace_model* M, double y, double x[2]) {
    double a = M->B[0] + y, b = M->B[1], c = M->B[2], d = M->B[3] + y, r0 = M->g[0], r1 = M->g[1];
    if (fabs(c) > fabs(a)) { double t = a; a = c; c = t; t = b; b = d; d = t; t = r0; r0 = r1; r1 = t; }
    const double l = c / a, u11 = d - l * b, y1 = r1 - l * r0;
    const double x1 = y1 / u11, x0 = (r0 - b * x1) / a;
    x[0] = -x0; x[1] = -x1;
}
static double subspace_eval(const subspace_model* M, const double x[2]) {
    return 0.5 * (x[0] * (M->B[0] * x[0] + M->B[1] * x[1]) + x[1] * (M->B[2] * x[0] + M->B[3] * x[1])) + M->g[0] * x[0] + M->g[1] * x[1];
}
static int find_minimum_on_boundary(const subspace_model* M, double radius, double minimum[2]) {
    minimum[0] = minimum[1] = 0;
    const double detB = M->B[0] * M->B[3] - M->B[1] * M->B[2], trB = M->B[0] + M->B[3], r2 = radius * radius;
    const double adj[4] = {M->B[3], -M->B[1], -M->B[2], M->B[0]};
    const double ag[2] = {adj[0] * M->g[0] + adj[1] * M->g[1], adj[2] * M->g[0] + adj[3] * M->g[1]};
    double poly[5];
    poly[0] = r2;
    poly[1] = 2.0 * r2 * trB;
    poly[2] = r2 * (trB * trB + 2.0 * detB) - (M->g[0] * M->g[0] + M->g[1] * M->g[1]);
    poly[3] = -2.0 * ((M->g[0] * ag[0] + M->g[1] * ag[1]) - r2 * detB * trB);
    poly[4] = r2 * detB * detB - (ag[0] * ag[0] + ag[1] * ag[1]);
    double roots[8];
    int nr = 0;
    if (!poly_roots_real(poly, 4, roots, &nr)) return 0;
    double best = 1.7976931348623157e308;
    int found = 0;
    for (int i = 0; i < nr; ++i) {
        double x[2];
        subspace_step_from_root(M, roots[i], x);
        const double nx = sqrt(x[0] * x[0] + x[1] * x[1]);
        if (nx > 0) {
            const double xs[2] = {radius / nx * x[0], radius / nx * x[1]};
            const double f = subspace_eval(M, xs);
            found = 1;
            if (f < best) { best = f; minimum[0] = x[0]; minimum[1] = x[1]; }
        }
    }
    return found;
}
/* test hook: the boundary-constrained 2-D problem alone */
int orc_subspace_boundary_minimum(const double B[4], const double g[2], double radius, double out[2]) {
    subspace_model M;
    memset(&M, 0, sizeof M);
    memcpy(M.B, B, sizeof M.B); memcpy(M.g, g, sizeof M.g);
    return find_minimum_on_boundary(&M, radius, out);
}

int orc_batch2_solve(const orc_batch_problem* p, const glio_batch_tr_opts* o, double* x_pose, double* x_sb, glio_summary* sum, double* history /* may be NULL: [max_iterations][4] per iteration: candidate cost, radius the step was computed with, |x - candidate|, step quality */) {
    const int K = p->K, B = p->n_imu > 0 ? 15 : 6, n = B * K, np = 7 * K, ns = B == 15 ? 9 * K : 0;
    const int hbw = orc_batch2_half_bandwidth(p);
    bandm Hm, Hcm, Lm;                                  /* normal matrix at the current point, at the candidate, Cholesky work: lower bands */
    band_alloc(&Hm, n, hbw); band_alloc(&Hcm, n, hbw); band_alloc(&Lm, n, hbw);
    bandm* H = &Hm; bandm* Hc = &Hcm; bandm* L = &Lm;
    double* vec = (double*)calloc((size_t)16 * n, sizeof(double));
    double* g = vec, *gc = vec + n, *gs = vec + 2 * n, *scale = vec + 3 * n, *diag = vec + 4 * n, *grad = vec + 5 * n, *gn = vec + 6 * n,
          *step = vec + 7 * n, *delta = vec + 8 * n, *tmp = vec + 9 * n, *work = vec + 10 * n;      /* work: 4 n */
    double* basis = vec + 14 * n;                                                                        /* 2 n */
    double* xp = (double*)malloc(sizeof(double) * np), *xs = (double*)malloc(sizeof(double) * (ns + 1));
    double* cp = (double*)malloc(sizeof(double) * np), *cs = (double*)malloc(sizeof(double) * (ns + 1));
    double* np_ = (double*)malloc(sizeof(double) * np), *ns_ = (double*)malloc(sizeof(double) * (ns + 1));
    memcpy(xp, x_pose, sizeof(double) * np);
    if (ns) memcpy(xs, x_sb, sizeof(double) * ns);
    memset(sum, 0, sizeof *sum);
    subspace_model SM;
    memset(&SM, 0, sizeof SM);
    SM.basis = basis;
    double cost;
    int ok = batch2_linearize_band(p, xp, xs, H, g, &cost);
    if (!ok) { sum->termination = GLIO_TERM_FAILURE; goto done; }
    sum->initial_cost = cost;
    for (int i = 0; i < n; ++i) scale[i] = o->jacobi_scaling ? 1.0 / (1.0 + sqrt(*bat(H, i, i))) : 1.0;
    double radius = o->initial_trust_region_radius, mu = 1e-8, alpha = 0, dogleg_step_norm = 0;
    int reuse = 0, iteration = 0, invalid = 0;
    double minimum_cost = cost, current_cost = cost, reference_cost = cost, candidate_cost = cost;
    double acc_ref = 0, acc_cand = 0;
    double user_min_cost = cost;                   /* TrustRegionMinimizer::minimum_cost_: x_pose / x_sb hold the point of this cost */
    int n_nonmono = 0;
    const int max_nonmono = o->use_nonmonotonic_steps ? o->max_consecutive_nonmonotonic_steps : 0;
    sum->termination = GLIO_TERM_NO_CONVERGENCE;
    for (;;) {
        for (int i = 0; i < n; ++i) tmp[i] = -g[i];
        batch2_plus(p, xp, xs, tmp, np_, ns_);
        double gm = 0;
        for (int i = 0; i < np; ++i) if (fabs(xp[i] - np_[i]) > gm) gm = fabs(xp[i] - np_[i]);
        for (int i = 0; i < ns; ++i) if (fabs(xs[i] - ns_[i]) > gm) gm = fabs(xs[i] - ns_[i]);
        sum->gradient_max_norm = gm;
        if (iteration >= o->max_iterations) { sum->termination = GLIO_TERM_NO_CONVERGENCE; break; }
        if (gm <= o->gradient_tolerance) { sum->termination = GLIO_TERM_GRADIENT_TOL; break; }
        if (radius <= o->min_trust_region_radius) { sum->termination = GLIO_TERM_MIN_RADIUS; break; }
        ++iteration;
        for (int i = 0; i < n; ++i) gs[i] = scale[i] * g[i];            /* (S H S is evaluated entry by entry from the band: hs_at) */
        int step_valid = 1;
        if (!reuse) {
            for (int i = 0; i < n; ++i) {
                double d = hs_at(H, scale, i, i);
                d = d < 1e-6 ? 1e-6 : (d > 1e32 ? 1e32 : d);
                diag[i] = sqrt(d);
                grad[i] = gs[i] / diag[i];
            }
            for (int i = 0; i < n; ++i) tmp[i] = grad[i] / diag[i];
            double Jg2 = 0, gg = 0;
            band_scaled_matvec(H, scale, tmp, work);
            for (int i = 0; i < n; ++i) { Jg2 += tmp[i] * work[i]; gg += grad[i] * grad[i]; }
            alpha = gg / Jg2;
            int solved = 0;
            while (mu < 1.0) {
                for (int i = 0; i < n; ++i) for (int j = i - hbw < 0 ? 0 : i - hbw; j <= i; ++j) *bat(L, i, j) = hs_at(H, scale, i, j);
                for (int i = 0; i < n; ++i) *bat(L, i, i) += mu * diag[i] * diag[i];
                if (band_chol(L) == 0) {
                    band_chol_solve(L, gs, tmp);
                    int fin = 1;
                    for (int i = 0; i < n; ++i) if (!isfinite(tmp[i])) fin = 0;
                    if (fin) { solved = 1; break; }
                }
                mu *= 10.0;
            }
            if (!solved) step_valid = 0;
            else {
                for (int i = 0; i < n; ++i) gn[i] = -diag[i] * tmp[i];
                if (o->dogleg_type == GLIO_DOGLEG_SUBSPACE && !compute_subspace_model(n, H, scale, diag, grad, gn, &SM, work)) step_valid = 0;
            }
        }
        if (step_valid) {
            double gg = 0, nn2 = 0, gd = 0;
            for (int i = 0; i < n; ++i) { gg += grad[i] * grad[i]; nn2 += gn[i] * gn[i]; gd += grad[i] * gn[i]; }
            const double gnorm = sqrt(gg), gnn = sqrt(nn2);
            int traditional = o->dogleg_type != GLIO_DOGLEG_SUBSPACE;
            if (!traditional) {           /* ComputeSubspaceDoglegStep */
                if (gnn <= radius) { for (int i = 0; i < n; ++i) step[i] = gn[i]; dogleg_step_norm = gnn; }
                else if (SM.one_dim) { for (int i = 0; i < n; ++i) step[i] = -(radius / gnorm) * grad[i]; dogleg_step_norm = radius; }
                else {
                    double m2[2];
                    if (!find_minimum_on_boundary(&SM, radius, m2)) traditional = 1;       /* "Taking traditional dogleg step instead" */
                    else { for (int i = 0; i < n; ++i) step[i] = basis[2 * i] * m2[0] + basis[2 * i + 1] * m2[1]; dogleg_step_norm = radius; }
                }
            }
            if (traditional) {            /* ComputeTraditionalDoglegStep */
                if (gnn <= radius) { for (int i = 0; i < n; ++i) step[i] = gn[i]; dogleg_step_norm = gnn; }
                else if (gnorm * alpha >= radius) { for (int i = 0; i < n; ++i) step[i] = -(radius / gnorm) * grad[i]; dogleg_step_norm = radius; }
                else {
                    const double b_dot_a = -alpha * gd, a_sq = alpha * alpha * gg;
                    const double b_minus_a_sq = nn2 - 2 * b_dot_a + a_sq, c = b_dot_a - a_sq;
                    const double d = sqrt(c * c + b_minus_a_sq * (radius * radius - a_sq));
                    const double beta = (c <= 0) ? (d - c) / b_minus_a_sq : (radius * radius - a_sq) / (d + c);
                    double s2 = 0;
                    for (int i = 0; i < n; ++i) { step[i] = (-alpha * (1.0 - beta)) * grad[i] + beta * gn[i]; s2 += step[i] * step[i]; }
                    dogleg_step_norm = sqrt(s2);
                }
            }
            for (int i = 0; i < n; ++i) step[i] /= diag[i];
        }
        double mcc = 0;
        if (step_valid) {
            double lin = 0, quad = 0;
            band_scaled_matvec(H, scale, step, work);
            for (int i = 0; i < n; ++i) { quad += step[i] * work[i]; lin += gs[i] * step[i]; }
            mcc = -(lin + 0.5 * quad);
            if (!(mcc > 0.0)) step_valid = 0;
        }
        if (!step_valid) {
            if (++invalid >= 5) { sum->termination = GLIO_TERM_FAILURE; break; }
            mu *= 10.0; reuse = 0;
            if (history) { double* h = history + 4 * (iteration - 1); h[0] = current_cost; h[1] = radius / 1.0; h[2] = 0; h[3] = 0; }
            continue;
        }
        invalid = 0;
        for (int i = 0; i < n; ++i) delta[i] = step[i] * scale[i];
        batch2_plus(p, xp, xs, delta, cp, cs);
        double ccost;
        if (!batch2_linearize_band(p, cp, cs, Hc, gc, &ccost)) { radius *= 0.5; reuse = 1; continue; }
        {
            double d2 = 0, x2 = 0;
            for (int i = 0; i < np; ++i) { d2 += (xp[i] - cp[i]) * (xp[i] - cp[i]); x2 += xp[i] * xp[i]; }
            for (int i = 0; i < ns; ++i) { d2 += (xs[i] - cs[i]) * (xs[i] - cs[i]); x2 += xs[i] * xs[i]; }
            if (history) { double* h = history + 4 * (iteration - 1); h[0] = ccost; h[1] = radius; h[2] = sqrt(d2); h[3] = 0; }
            if (sqrt(d2) <= o->parameter_tolerance * (sqrt(x2) + o->parameter_tolerance)) { sum->termination = GLIO_TERM_PARAMETER_TOL; break; }
        }
        if (fabs(current_cost - ccost) <= o->function_tolerance * current_cost) { sum->termination = GLIO_TERM_FUNCTION_TOL; break; }
        const double rel = (current_cost - ccost) / mcc;
        const double hist = (reference_cost - ccost) / (acc_ref + mcc);
        const double quality = max_nonmono > 0 ? (rel > hist ? rel : hist) : rel;
        if (history) history[4 * (iteration - 1) + 3] = quality;
        if (quality > o->min_relative_decrease) {
            memcpy(xp, cp, sizeof(double) * np);
            if (ns) memcpy(xs, cs, sizeof(double) * ns);
            { bandm* t = H; H = Hc; Hc = t; }
            memcpy(g, gc, sizeof(double) * n);
            ++sum->successful_steps;
            if (quality < 0.25) radius *= 0.5;
            if (quality > 0.75) radius = radius > 3.0 * dogleg_step_norm ? radius : 3.0 * dogleg_step_norm;
            mu = mu * 2.0 / 10.0 > 1e-8 ? mu * 2.0 / 10.0 : 1e-8;
            reuse = 0;
            current_cost = ccost;
            acc_cand += mcc; acc_ref += mcc;
            if (current_cost < minimum_cost) { minimum_cost = current_cost; n_nonmono = 0; candidate_cost = current_cost; acc_cand = 0; }
            else { ++n_nonmono; if (current_cost > candidate_cost) { candidate_cost = current_cost; acc_cand = 0; } }
            if (n_nonmono == max_nonmono) { reference_cost = candidate_cost; acc_ref = acc_cand; }
            /* FinalizeIterationAndCheckIfMinimizerCanContinue: the user's parameters follow the best point only */
            if (current_cost < user_min_cost) {
                user_min_cost = current_cost;
                memcpy(x_pose, xp, sizeof(double) * np);
                if (ns) memcpy(x_sb, xs, sizeof(double) * ns);
            }
        } else { radius *= 0.5; reuse = 1; }
    }
    sum->iterations = iteration;
    sum->final_cost = user_min_cost;
    sum->final_radius = radius;
done:
    free(Hm.a); free(Hcm.a); free(Lm.a); free(vec); free(xp); free(xs); free(cp); free(cs); free(np_); free(ns_);
    return sum->termination != GLIO_TERM_FAILURE;
}
