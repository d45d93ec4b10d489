// batch_tr_kernels.hip -- the batch problem of Estimator::optimizeBatchWithLandMark beyond its scan-to-multiscan constraints, and its
// trust-region solve, behind the C-ABI (reference GLIO/src/Estimator.cpp:2739-3410):
//   * the "small" factors: delta_q_factor_auto attitude constraints (:2831-2891, LidarKeyframeFactor.h:283-303) and dd_psr_factor_20 per
//     GNSS epoch between the bracketing keyframes (:3197-3271, :1899-1911; dd_psr_factor.hpp:25-171) -- one wavefront per factor, a
//     121-double record each, summed into the band by a gather in fixed order (no atomics);
//   * round 3: the IMU chain (:2990-3001, parameter blocks :2809-2819) -- an ImuFactor between consecutive keyframes, 15 unknowns per
//     keyframe (pose 6 + speed/bias 9); the factor is the sliding window's device code (factor_kernels.hip), one workgroup per edge;
//   * glio_batch_solve_tr2: ceres::Solve as configured at :3275-3284 -- DOGLEG, SUBSPACE_DOGLEG, non-monotonic steps, max_num_iter --
//     DEVICE RESIDENT: the trust-region state machine (BtStatus) lives in device memory, every kernel of an iteration reads it and
//     returns at once when it has nothing to do, the host only feeds kernel groups and watches a progress word in mapped memory
//     (one stream synchronisation per solve).  Jacobi scaling, Cauchy point, Gauss-Newton step by block cyclic reduction with Ceres'
//     mu D^2 regularisation, the two-dimensional subspace model (orthonormal basis of {gradient, Gauss-Newton}, B = U^T Hs U by two
//     matrix-vector products, the quartic of the boundary problem solved by the Aberth iteration on one lane), Ceres' step
//     evaluator, and the minimum-cost iterate as the result (Ceres copies x to the user's parameters only when its cost is below
//     every earlier one).
//   * SHARDED over ranks (glio_batch_set_shard): a rank owns a contiguous range of super-blocks of keyframes: its constraints
//     (K8), its small factors, its IMU edges, the rows of every matrix-vector product and its part of the elimination tree.
//     Per iteration the caller's all-reduce hook is called on FIVE small buffers (sizes at 2000 keyframes, 8 ranks, 15 states):
//       A  the assembly buffer: the band rows within `band` keyframes of a range boundary + the diagonal, the gradient and the
//          cost as partial sums (0.65 MB)       -- instead of the whole 4.1 MB band
//       B  the separator system of the block cyclic reduction + the Cauchy curvature (0.85 MB)
//       C  the Gauss-Newton step, every rank its own keyframes (0.24 MB)
//       D  three curvature sums of the subspace model (64 B)         E  the model cost change's sums (64 B)
//     and every rank takes identical decisions from identical numbers.  With one rank no hook is called at all.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

#include "batch_device.h"

#define SREC 124           // Haa 36 | Hbb 36 | Hab 36 | ga 6 | gb 6 | cost 1 | pad
#define TRV_THREADS 256

// Trust-region state machine of the batch solve, in device memory for the whole solve
struct BtStatus {
    int done, termination, iteration, successful;
    int invalid, reuse, cur, first;
    int skip_solve;            // done || the stored Gauss-Newton / Cauchy data are reused: the solve kernels of this group return
    int step_pending;          // a candidate has been produced and linearised: the next state machine judges it
    int solve_failed;          // the factorisation of this group broke down (or gave non-finite numbers)
    int step_valid;            // the step of this group is usable (solve succeeded, model cost change > 0)
    int one_dim, n_nonmono, solve_id, group;
    int copy_min, retry, skip_step, pad1_;   // skip_step: done || this group has no usable step: the step kernels and the candidate's linearisation return
    double radius, mu, alpha, dogleg_step_norm;
    double cost, cand_cost, minimum_cost, reference_cost, candidate_cost, acc_ref, acc_cand, user_min_cost, initial_cost;
    double gg, nn2, gd;        // |g~|^2, |gn|^2, g~ . gn
    double puu;                // (g~/D)^T Hs (g~/D): the Cauchy curvature
    double w2;                 // |w2|^2 of the second (unnormalised) basis vector
    double p11, p12, p22;      // curvature of the subspace basis (u1, w2)
    double c_g, c_n, c_1, c_2; // step (D-space) = c_g g~ + c_n gn + c_1 u1 + c_2 w2
    double lin, quad, sd2;     // gs . s, s^T Hs s, |s_D|^2
    double mcc;
    double dx2, xn2, cand_grad_max, grad_max;
    double pad_[4];
};

// ------------------------------------------------------------------------------------------------ small factors
__device__ __forceinline__ void bt_plus_jac(const double q[4], double P[12]) {
    P[0] = -q[1]; P[1] = -q[2]; P[2] = -q[3];
    P[3] = q[0];  P[4] = q[3];  P[5] = -q[2];
    P[6] = -q[3]; P[7] = q[0];  P[8] = q[1];
    P[9] = q[2];  P[10] = -q[1]; P[11] = q[0];
}
__device__ __forceinline__ void bt_qleft(const double q[4], double M[16]) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    M[0] = w; M[1] = -x; M[2] = -y; M[3] = -z;
    M[4] = x; M[5] = w; M[6] = -z; M[7] = y;
    M[8] = y; M[9] = z; M[10] = w; M[11] = -x;
    M[12] = z; M[13] = -y; M[14] = x; M[15] = w;
}
__device__ __forceinline__ void bt_qright(const double p[4], double M[16]) {
    const double w = p[0], x = p[1], y = p[2], z = p[3];
    M[0] = w; M[1] = -x; M[2] = -y; M[3] = -z;
    M[4] = x; M[5] = w; M[6] = z; M[7] = -y;
    M[8] = y; M[9] = -z; M[10] = w; M[11] = x;
    M[12] = z; M[13] = y; M[14] = -x; M[15] = w;
}

// one wavefront per factor: nr residuals with local Jacobians Ja, Jb (nr x 6, LDS) -> the record
__global__ __launch_bounds__(64) void k_small_eval(const BtSel sel, const int n_fac, const int* __restrict__ fa, const int* __restrict__ fb, const int* __restrict__ ftype,
                                                   const int* __restrict__ fidx, const double* __restrict__ poses0, const double* __restrict__ poses1,
                                                   const double* __restrict__ dq_const, const double* __restrict__ rp_const /* [n_rp][7]: delta_q (w,x,y,z), delta_p */,
                                                   const glio_dd_psr* __restrict__ dd, const double* __restrict__ Rel /* [9] R_ecef_local, [3] anchor */,
                                                   double* __restrict__ frec) {
    __shared__ double Ja[19 * 6], Jb[19 * 6], rr[19], raw[19], Jri[57], Jrj[57];
    const int f = blockIdx.x, lane = threadIdx.x;
    if (f >= n_fac || bt_skip(sel)) return;
    const double* poses = bt_pick(sel) ? poses1 : poses0;
    const int a = fa[f], b = fb[f];
    const double* pa = poses + 7 * (size_t)a;
    const double* pb = poses + 7 * (size_t)b;
    for (int k = lane; k < 19 * 6; k += 64) { Ja[k] = 0.0; Jb[k] = 0.0; }
    if (lane < 19) { rr[lane] = 0.0; raw[lane] = 0.0; }
    for (int k = lane; k < 57; k += 64) { Jri[k] = 0.0; Jrj[k] = 0.0; }
    GLIO_WAVE_LDS_SYNC();
    int nr;
    if (ftype[f] == 0) {
        // delta_q_factor_auto: r = 10000 (dq^-1 qi^-1 qj).vec; global 3x4 Jacobians, then Ceres' QuaternionParameterization
        nr = 3;
        if (lane < 3) {            // lane k takes residual row k (the common quaternion products are cheap enough to be repeated by the three lanes)
            const double* dq = dq_const + 4 * (size_t)fidx[f];
            const double* qi = pa + 3;
            const double* qj = pb + 3;
            double A[4], u[4], Au[4], p[4];
            d_qinv(dq, A); d_qinv(qi, u);
            d_qmul(A, u, Au); d_qmul(Au, qj, p);
            double LA[16], Rv[16], M[16], LAu[16], Pa[12], Pb[12];
            bt_qleft(A, LA); bt_qright(qj, Rv); bt_qleft(Au, LAu);
            for (int x = 0; x < 4; ++x) for (int y = 0; y < 4; ++y) { double s = 0; for (int k = 0; k < 4; ++k) s += LA[x * 4 + k] * Rv[k * 4 + y]; M[x * 4 + y] = s; }
            const double n2 = qi[0] * qi[0] + qi[1] * qi[1] + qi[2] * qi[2] + qi[3] * qi[3];
            const double Cq[4] = {qi[0], -qi[1], -qi[2], -qi[3]};
            bt_plus_jac(qi, Pa); bt_plus_jac(qj, Pb);
            {
                const int k = lane;
                // row 1 + k of M and of LAu by selects (a run-time index into a register array would put the arrays into scratch memory)
                double Mk[4], Lk[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) { Mk[m] = k == 0 ? M[4 + m] : (k == 1 ? M[8 + m] : M[12 + m]); Lk[m] = k == 0 ? LAu[4 + m] : (k == 1 ? LAu[8 + m] : LAu[12 + m]); }
                rr[k] = 10000.0 * (k == 0 ? p[1] : (k == 1 ? p[2] : p[3]));
                double Jgi[4], Jgj[4];
                for (int c = 0; c < 4; ++c) {
                    double s = 0;
                    for (int m = 0; m < 4; ++m) s += Mk[m] * (((m == c ? (m == 0 ? 1.0 : -1.0) : 0.0) - 2.0 * Cq[m] * qi[c] / n2) / n2);
                    Jgi[c] = 10000.0 * s;
                    Jgj[c] = 10000.0 * Lk[c];
                }
                for (int c = 0; c < 3; ++c) {
                    Ja[k * 6 + 3 + c] = Jgi[0] * Pa[c] + Jgi[1] * Pa[3 + c] + Jgi[2] * Pa[6 + c] + Jgi[3] * Pa[9 + c];
                    Jb[k * 6 + 3 + c] = Jgj[0] * Pb[c] + Jgj[1] * Pb[3 + c] + Jgj[2] * Pb[6 + c] + Jgj[3] * Pb[9 + c];
                }
            }
        }
    } else if (ftype[f] == 2) {
        // LidarPoseFactorBatchRelativeAutoDiff (LidarPoseFactor.h:55-97; the scan-to-multiscan factor of sms_fusion_level 0, Estimator.cpp:2897-2955):
        //   r[0:3] = 10 * 2 (dq^-1 q1^-1 q2).vec,  r[3:6] = 20 (q1^-1 (p2 - p1) - dp),  Eigen's inverse() = conjugate / |q|^2 and Eigen's q * v on the
        //   NON-normalised inverse -- what the reference's Jets differentiate; global Jacobians, then Ceres' QuaternionParameterization.  Lane k = row k.
        nr = 6;
        if (lane < 6) {
            const double* cst = rp_const + 7 * (size_t)fidx[f];
            const double* q1 = pa + 3;
            const double* q2 = pb + 3;
            double A[4], u[4], Au[4], p[4], v[3];
            d_qinv(cst, A); d_qinv(q1, u);
            d_qmul(A, u, Au); d_qmul(Au, q2, p);
            for (int k = 0; k < 3; ++k) v[k] = pb[k] - pa[k];
            double Pa[12], Pb[12];
            bt_plus_jac(q1, Pa); bt_plus_jac(q2, Pb);
            const double n2 = q1[0] * q1[0] + q1[1] * q1[1] + q1[2] * q1[2] + q1[3] * q1[3];
            const double Cq[4] = {q1[0], -q1[1], -q1[2], -q1[3]};
            const int k = lane < 3 ? lane : lane - 3;
            double Jg1[4], Jg2[4] = {0, 0, 0, 0}, Jp[3] = {0, 0, 0};        // this row's global Jacobians wrt q1, q2 and wrt p2 (= - wrt p1)
            double G[4];                                                       // the row of d r / d u (u = q1^-1), before d u / d q1
            double res;
            if (lane < 3) {
                double LA[16], Rv[16], LAu[16];
                bt_qleft(A, LA); bt_qright(q2, Rv); bt_qleft(Au, LAu);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    double sM = 0;
#pragma unroll
                    for (int x = 0; x < 4; ++x) sM += (k == 0 ? LA[4 + x] : (k == 1 ? LA[8 + x] : LA[12 + x])) * Rv[x * 4 + m];
                    G[m] = sM;
                    Jg2[m] = 20.0 * (k == 0 ? LAu[4 + m] : (k == 1 ? LAu[8 + m] : LAu[12 + m]));
                }
                res = 20.0 * (k == 0 ? p[1] : (k == 1 ? p[2] : p[3]));
            } else {
                // M(u) v = v + 2 w (q x v) + 2 q x (q x v);  d/dw = 2 (q x v),  d/dq = -2 w [v]x + 2 ((q.v) I + q v^T - 2 v q^T);  d/dv = M(u)
                const double w = u[0], qx = u[1], qy = u[2], qz = u[3];
                const double cx = qy * v[2] - qz * v[1], cy = qz * v[0] - qx * v[2], cz = qx * v[1] - qy * v[0];          // q x v
                const double ccx = qy * cz - qz * cy, ccy = qz * cx - qx * cz, ccz = qx * cy - qy * cx;                  // q x (q x v)
                const double rv = k == 0 ? v[0] + 2.0 * w * cx + 2.0 * ccx : (k == 1 ? v[1] + 2.0 * w * cy + 2.0 * ccy : v[2] + 2.0 * w * cz + 2.0 * ccz);
                res = 20.0 * (rv - cst[4 + k]);
                const double qv = qx * v[0] + qy * v[1] + qz * v[2];
                const double qk = k == 0 ? qx : (k == 1 ? qy : qz), vk = k == 0 ? v[0] : (k == 1 ? v[1] : v[2]);
                const double qq[3] = {qx, qy, qz};
                // row k of [v]x: (0, -v2, v1), (v2, 0, -v0), (-v1, v0, 0)
                const double sv[3] = {k == 0 ? 0.0 : (k == 1 ? v[2] : -v[1]), k == 0 ? -v[2] : (k == 1 ? 0.0 : v[0]), k == 0 ? v[1] : (k == 1 ? -v[0] : 0.0)};
                // row k of [q]x and of [q]x [q]x = q q^T - (q.q) I
                const double sq[3] = {k == 0 ? 0.0 : (k == 1 ? qz : -qy), k == 0 ? -qz : (k == 1 ? 0.0 : qx), k == 0 ? qy : (k == 1 ? -qx : 0.0)};
                const double q2n = qx * qx + qy * qy + qz * qz;
                G[0] = 2.0 * (k == 0 ? cx : (k == 1 ? cy : cz));
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    G[1 + c] = -2.0 * w * sv[c] + 2.0 * ((k == c ? qv : 0.0) + qk * v[c] - 2.0 * vk * qq[c]);
                    Jp[c] = 20.0 * ((k == c ? 1.0 : 0.0) + 2.0 * w * sq[c] + 2.0 * (qk * qq[c] - (k == c ? q2n : 0.0)));
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                double sacc = 0;
#pragma unroll
                for (int m = 0; m < 4; ++m) sacc += G[m] * (((m == c ? (m == 0 ? 1.0 : -1.0) : 0.0) - 2.0 * Cq[m] * q1[c] / n2) / n2);
                Jg1[c] = 20.0 * sacc;
            }
            rr[lane] = res;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                Ja[lane * 6 + c] = -Jp[c];
                Jb[lane * 6 + c] = Jp[c];
                Ja[lane * 6 + 3 + c] = Jg1[0] * Pa[c] + Jg1[1] * Pa[3 + c] + Jg1[2] * Pa[6 + c] + Jg1[3] * Pa[9 + c];
                Jb[lane * 6 + 3 + c] = Jg2[0] * Pb[c] + Jg2[1] * Pb[3 + c] + Jg2[2] * Pb[6 + c] + Jg2[3] * Pb[9 + c];
            }
        }
    } else {
#pragma clang fp contract(off)
        // dd_psr_factor_20::Evaluate (dd_psr_factor.hpp:25-171): one lane per satellite, then W r / W J by one lane per row
        // (no FMA contraction: double differences of ~2.6e7 m ranges round like the reference's scalar build)
        nr = 19;
        const glio_dd_psr& F = dd[fidx[f]];
        const int ns = F.n_sat, m = F.master, nw = ns - 1, i = lane;
        const double* R = Rel;
        if (i < ns && i != m) {
            double lp[3], Pe[3];
            for (int k = 0; k < 3; ++k) lp[k] = F.ratio * pa[k] + (1.0 - F.ratio) * pb[k];
            for (int k = 0; k < 3; ++k) Pe[k] = R[3 * k] * lp[0] + R[3 * k + 1] * lp[1] + R[3 * k + 2] * lp[2] + Rel[9 + k];
            const int ri = i < m ? i : i - 1;
            double d_ui[3], d_um[3], d_ri[3], d_rm[3];
            for (int k = 0; k < 3; ++k) {
                d_ui[k] = F.user_sat_pos[i][k] - Pe[k]; d_um[k] = F.user_sat_pos[m][k] - Pe[k];
                d_ri[k] = F.ref_sat_pos[i][k] - F.station[k]; d_rm[k] = F.ref_sat_pos[m][k] - F.station[k];
            }
            const double r_ui = sqrt(d_dot3_nc(d_ui, d_ui)), r_um = sqrt(d_dot3_nc(d_um, d_um)), r_ri = sqrt(d_dot3_nc(d_ri, d_ri)), r_rm = sqrt(d_dot3_nc(d_rm, d_rm));
            const double est = (r_ui - r_ri) - (r_um - r_rm);
            const double obs = (F.user_psr[i] - F.ref_psr[i]) - (F.user_psr[m] - F.ref_psr[m]);
            const double wgt = fabs(est - obs) > F.threshold ? 0.05 : 1.0;
            raw[ri] = wgt * (est - obs);
            for (int c = 0; c < 3; ++c) {
                const double ei = (d_ui[0] * R[c] + d_ui[1] * R[3 + c] + d_ui[2] * R[6 + c]) / r_ui;
                const double em = (d_um[0] * R[c] + d_um[1] * R[3 + c] + d_um[2] * R[6 + c]) / r_um;
                Jri[ri * 3 + c] = (-ei * wgt * F.ratio) - (-em * wgt * F.ratio);
                Jrj[ri * 3 + c] = (-ei * wgt * (1.0 - F.ratio)) - (-em * wgt * (1.0 - F.ratio));
            }
        }
        GLIO_WAVE_LDS_SYNC();
        if (i < nw) {
            double sr = 0, si[3] = {0, 0, 0}, sj[3] = {0, 0, 0};
            for (int q = 0; q < nw; ++q) {
                const double wv = F.weight[i * nw + q];
                sr += wv * raw[q];
                for (int k = 0; k < 3; ++k) { si[k] += wv * Jri[q * 3 + k]; sj[k] += wv * Jrj[q * 3 + k]; }
            }
            rr[i] = sr;
            for (int k = 0; k < 3; ++k) { Ja[i * 6 + k] = si[k]; Jb[i * 6 + k] = sj[k]; }
        }
    }
    GLIO_WAVE_LDS_SYNC();
    double* rec = frec + (size_t)f * SREC;
    for (int e = lane; e < 121; e += 64) {
        double s = 0;
        if (e < 108) {
            const int blk = e / 36, u = (e % 36) / 6, v = e % 6;
            const double* X = blk == 1 ? Jb : Ja;
            const double* Y = blk == 0 ? Ja : Jb;
            for (int q = 0; q < nr; ++q) s += X[q * 6 + u] * Y[q * 6 + v];
        } else if (e < 120) {
            const double* X = e < 114 ? Ja : Jb;
            const int u = (e - 108) % 6;
            for (int q = 0; q < nr; ++q) s += X[q * 6 + u] * rr[q];
        } else {
            for (int q = 0; q < nr; ++q) s += rr[q] * rr[q];
            s *= 0.5;
        }
        rec[e] = s;
        if (e == 120) frec[(size_t)n_fac * SREC + f] = s;          // costs once more, densely, behind the records: what k_small_cost sums
    }
}

// adds the factor records into the (reduced) band buffer: one thread per entry, factors in index order
__global__ void k_small_add(const BtSel sel, const int K, const int band, const int2* __restrict__ sidx, const double* __restrict__ frec, const int n_fac,
                            double* __restrict__ Hg0, double* __restrict__ Hg1) {
    if (bt_skip(sel)) return;
    double* Hg = bt_pick(sel) ? Hg1 : Hg0;
    const long long nH = (long long)K * (band + 1) * 36, nG = (long long)K * 6;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int wdt = 2 * band + 1;
    if (e < nH) {
        const int k = (int)(e / ((band + 1) * 36)), rem = (int)(e % ((band + 1) * 36)), d = rem / 36, r = (rem % 36) / 6, c = rem % 6;
        double s = 0;
        if (d == 0) {
            for (int o = -band; o <= band; ++o) {
                if (o == 0 || k + o < 0 || k + o >= K) continue;
                const int2 pa = sidx[(size_t)k * wdt + o + band];                 // (a = k, b = k + o): Haa
                for (int q = 0; q < pa.y; ++q) s += frec[(size_t)(pa.x + q) * SREC + r * 6 + c];
                const int2 pb = sidx[(size_t)(k + o) * wdt + (-o) + band];        // (a = k + o, b = k): Hbb
                for (int q = 0; q < pb.y; ++q) s += frec[(size_t)(pb.x + q) * SREC + 36 + r * 6 + c];
            }
        } else if (k + d < K) {
            const int2 pa = sidx[(size_t)k * wdt + d + band];                     // (a = k, b = k + d): H(k, k+d) = Ja^T Jb
            for (int q = 0; q < pa.y; ++q) s += frec[(size_t)(pa.x + q) * SREC + 72 + r * 6 + c];
            const int2 pb = sidx[(size_t)(k + d) * wdt + (-d) + band];            // (a = k + d, b = k): H(k, k+d) = Jb^T Ja = (Ja^T Jb)^T
            for (int q = 0; q < pb.y; ++q) s += frec[(size_t)(pb.x + q) * SREC + 72 + c * 6 + r];
        }
        Hg[e] += s;
    } else if (e < nH + nG) {
        const int k = (int)((e - nH) / 6), r = (int)((e - nH) % 6);
        double s = 0;
        for (int o = -band; o <= band; ++o) {
            if (o == 0 || k + o < 0 || k + o >= K) continue;
            const int2 pa = sidx[(size_t)k * wdt + o + band];
            for (int q = 0; q < pa.y; ++q) s += frec[(size_t)(pa.x + q) * SREC + 108 + r];
            const int2 pb = sidx[(size_t)(k + o) * wdt + (-o) + band];
            for (int q = 0; q < pb.y; ++q) s += frec[(size_t)(pb.x + q) * SREC + 114 + r];
        }
        Hg[e] += s;
    }
}
// the cost of the small factors: fixed assignment of factors to threads and a fixed reduction tree (deterministic)
__global__ __launch_bounds__(256) void k_small_cost(const BtSel sel, const double* __restrict__ frec, const int n_fac, double* __restrict__ cost0, double* __restrict__ cost1) {
    __shared__ double red[4];
    if (bt_skip(sel)) return;
    double* cost = bt_pick(sel) ? cost1 : cost0;
    double s = 0;
    for (int q = threadIdx.x; q < n_fac; q += 256) s += frec[(size_t)n_fac * SREC + q];           // the dense copy of the records' cost entries (coalesced)
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *cost += (red[0] + red[1]) + (red[2] + red[3]);
}


// ------------------------------------------------------------------------------------------------ trust-region vectors
// vector slots (n = B K doubles each, every slot starts on its own 128 B line)
enum { V_SC = 0, V_DG, V_GR, V_UU, V_GS, V_DA, V_GN, V_U1, V_W2, V_X1, V_X2, V_ST, V_SD, V_DL, V_T1, V_T2, V_COUNT };

struct BtOpts {        // glio_batch_tr_opts, by value
    int max_iterations, max_nonmono, jacobi, dogleg_type;
    double min_radius, min_rel, ftol, gtol, ptol;
};
struct BtBufs {        // what most kernels need
    BtStatus* st;
    int K, band, B, lo, hi;          // owned keyframes [lo, hi)
    double* x[2]; double* s[2];      // poses [K][7], speed-bias [K][9] of the current / candidate point
    double* xmin; double* smin;
    double* hg[2];                   // pose band (local rows), g6, cost
    PairBlock* rec[2];               // IMU edge records (null without the chain)
    double* A[2];                    // replicated after the assembly all-reduce: diag [n] | g [n] | cost
    double* vec; long long vstride;
};
#define BVEC(a, k) ((a).vec + (size_t)(k) * (a).vstride)

__device__ __forceinline__ HView bt_view(const BtBufs& a, const int which) {
    HView v; v.Hg = a.hg[which]; v.imu = a.rec[which]; v.K = a.K; v.band = a.band; v.B = a.B;
    return v;
}

// ---- assembly buffer: stage = [boundary rows NB x 2 band x (band+1) 36 | diag n | g n | cost | pad]
struct BtAsm { int NB; int eo0, eo1; long long bnd_doubles; const int* bnd_kf; /* [NB] first keyframe of the next rank */ double* stage; int rank; };

__global__ void k_bt_pack(const BtBufs a, const BtAsm m, const BtSel sel) {
    if (bt_skip(sel)) return;
    const int which = bt_pick(sel), K = a.K, band = a.band, B = a.B, n = B * K, bw = band + 1;
    const double* Hg = a.hg[which];
    const PairBlock* rec = a.rec[which];
    const long long nH = (long long)K * bw * 36;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long row_d = (long long)bw * 36, per_b = 2LL * band * row_d;
    if (t < m.bnd_doubles) {
        const int j = (int)(t / per_b);
        const long long r = t - (long long)j * per_b;
        const int k = m.bnd_kf[j] - band + (int)(r / row_d);
        m.stage[t] = (k >= 0 && k < K) ? Hg[(size_t)k * row_d + (r % row_d)] : 0.0;
        return;
    }
    const long long i = t - m.bnd_doubles;
    if (i >= 2LL * n) return;
    const int isg = i >= n, idx = (int)(isg ? i - n : i), k = idx / B, r = idx % B;
    double v = 0.0;
    if (r < 6) v = isg ? Hg[nH + (size_t)k * 6 + r] : Hg[(size_t)k * row_d + r * 7];
    if (rec) {          // my own IMU edges only: these are partial sums
        if (k >= m.eo0 && k < m.eo1) v += isg ? rec[k].g[r] : rec[k].H[r * 30 + r];
        if (k - 1 >= m.eo0 && k - 1 < m.eo1) v += isg ? rec[k - 1].g[15 + r] : rec[k - 1].H[(15 + r) * 30 + 15 + r];
    }
    m.stage[m.bnd_doubles + i] = v;
}
// the cost of this rank: plane constraints + small factors (in Hg) + its IMU edges
__global__ __launch_bounds__(256) void k_bt_pack_cost(const BtBufs a, const BtAsm m, const BtSel sel) {
    __shared__ double red[4];
    if (bt_skip(sel)) return;
    const int which = bt_pick(sel), K = a.K, n = a.B * K;
    const PairBlock* rec = a.rec[which];
    double s = 0;
    if (rec) for (int e = m.eo0 + threadIdx.x; e < m.eo1; e += 256) s += rec[e].cost;
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) m.stage[m.bnd_doubles + 2LL * n] = a.hg[which][(size_t)K * (a.band + 1) * 36 + (size_t)K * 6] + ((red[0] + red[1]) + (red[2] + red[3]));
}
// after the all-reduce: diag | g | cost become the replicated A[which]; the boundary rows next to this rank replace its local ones
__global__ void k_bt_unpack(const BtBufs a, const BtAsm m, const BtSel sel) {
    if (bt_skip(sel)) return;
    const int which = bt_pick(sel), K = a.K, band = a.band, n = a.B * K, bw = band + 1;
    double* Hg = a.hg[which];
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long row_d = (long long)bw * 36, per_b = 2LL * band * row_d;
    if (t < m.bnd_doubles) {
        const int j = (int)(t / per_b);
        if (j != m.rank && j != m.rank - 1) return;
        const long long r = t - (long long)j * per_b;
        const int k = m.bnd_kf[j] - band + (int)(r / row_d);
        if (k >= 0 && k < K) Hg[(size_t)k * row_d + (r % row_d)] = m.stage[t];
        return;
    }
    const long long i = t - m.bnd_doubles;
    if (i <= 2LL * n) a.A[which][i] = m.stage[m.bnd_doubles + i];
}

// |x - x_c|^2, |x|^2 (all parameter blocks) and the gradient max norm |x_c - Plus(x_c, -g_c)|_inf of the candidate.  One lane per keyframe,
// BT_NORM_THREADS per workgroup; the workgroups' parts are added by the last one to arrive, in workgroup order (the same bits on every rank).
#define BT_NORM_THREADS 64
struct NormParts { double* parts; unsigned int* ticket; };
__global__ __launch_bounds__(BT_NORM_THREADS) void k_bt_norms(const BtBufs a, const BtSel sel, const NormParts np) {
    __shared__ int s_last;
    if (bt_skip(sel)) return;
    const int cur = a.st->cur & 1, cand = cur ^ 1, K = a.K, B = a.B, n = B * K;
    const double* x = a.x[cur]; const double* xc = a.x[cand];
    const double* s = a.s[cur]; const double* sc = a.s[cand];
    const double* g = a.A[cand] + n;
    double d2 = 0, x2 = 0, gm = 0;
    const int k = blockIdx.x * BT_NORM_THREADS + threadIdx.x;
    if (k < K) {
        double xv[7], xcv[7], gv[15], sv[9], scv[9];
#pragma unroll
        for (int c = 0; c < 7; ++c) { xv[c] = x[7 * k + c]; xcv[c] = xc[7 * k + c]; }
#pragma unroll
        for (int c = 0; c < 15; ++c) gv[c] = c < B ? g[B * k + c] : 0.0;
        if (B == 15) {
#pragma unroll
            for (int c = 0; c < 9; ++c) { sv[c] = s[9 * k + c]; scv[c] = sc[9 * k + c]; }
        }
        double nd[3], qn[4];
#pragma unroll
        for (int c = 0; c < 7; ++c) { const double v = xv[c], d = v - xcv[c]; d2 += d * d; x2 += v * v; }
#pragma unroll
        for (int c = 0; c < 3; ++c) { gm = fmax(gm, fabs(gv[c])); nd[c] = -gv[3 + c]; }
        d_quat_plus(xcv + 3, nd, qn);
#pragma unroll
        for (int c = 0; c < 4; ++c) gm = fmax(gm, fabs(xcv[3 + c] - qn[c]));
        if (B == 15) {
#pragma unroll
            for (int c = 0; c < 9; ++c) { const double v = sv[c], d = v - scv[c]; d2 += d * d; x2 += v * v; gm = fmax(gm, fabs(gv[6 + c])); }
        }
    }
    // the lanes of a wavefront are added lane by lane in a fixed tree; workgroups in index order
    d2 = wave_sum(d2); x2 = wave_sum(x2);
    gm = wave_max(gm);
    if (threadIdx.x == 0) { np.parts[3 * blockIdx.x] = d2; np.parts[3 * blockIdx.x + 1] = x2; np.parts[3 * blockIdx.x + 2] = gm; }
    __threadfence();
    if (threadIdx.x == 0) s_last = atomicAdd(np.ticket, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (threadIdx.x == 0) {
        double p = 0, q = 0, m = 0;
        for (int w = 0; w < (int)gridDim.x; ++w) {
            p += __builtin_nontemporal_load(&np.parts[3 * w]); q += __builtin_nontemporal_load(&np.parts[3 * w + 1]);
            m = fmax(m, __builtin_nontemporal_load(&np.parts[3 * w + 2]));
        }
        a.st->dx2 = p; a.st->xn2 = q; a.st->cand_grad_max = m; a.st->cand_cost = a.A[cand][2 * n];
        a.st->step_pending = 1;
        *np.ticket = 0;
    }
}

// ------------------------------------------------------------------------------------------------ the state machine
struct BtHost { int* progress; BtStatus* result; };     // mapped host memory
__global__ __launch_bounds__(64) void k_bt_state_machine(const BtBufs a, const BtOpts o, const BtHost h) {
    __shared__ int sh_copy, sh_cur;
    if (threadIdx.x == 0) {
        BtStatus s = *a.st;
        sh_copy = 0; sh_cur = s.cur & 1;
        if (!s.done) {
            bool retry = false;
            if (s.first) {
                s.first = 0; s.cur ^= 1;
                s.cost = s.cand_cost; s.initial_cost = s.cost;
                s.minimum_cost = s.reference_cost = s.candidate_cost = s.user_min_cost = s.cost;
                s.acc_ref = s.acc_cand = 0; s.n_nonmono = 0;
                s.grad_max = s.cand_grad_max;
                sh_copy = 1;
            } else if (s.solve_failed) {
                s.mu *= 10.0;                                         // ComputeGaussNewtonStep: while (mu < max_mu) { ...; mu *= 10 }
                if (s.mu < 1.0) retry = true;
                else { if (++s.invalid >= 5) { s.done = 1; s.termination = GLIO_TERM_FAILURE; } else { s.mu *= 10.0; s.reuse = 0; } }
            } else if (s.step_pending) {
                const double ccost = s.cand_cost;
                if (sqrt(s.dx2) <= o.ptol * (sqrt(s.xn2) + o.ptol)) { s.done = 1; s.termination = GLIO_TERM_PARAMETER_TOL; }
                else if (fabs(s.cost - ccost) <= o.ftol * s.cost) { s.done = 1; s.termination = GLIO_TERM_FUNCTION_TOL; }
                else {
                    const double rel = (s.cost - ccost) / s.mcc, hist = (s.reference_cost - ccost) / (s.acc_ref + s.mcc);
                    const double quality = o.max_nonmono > 0 ? fmax(rel, hist) : rel;
                    if (quality > o.min_rel) {
                        s.cur ^= 1; s.successful += 1;
                        if (quality < 0.25) s.radius *= 0.5;
                        if (quality > 0.75) s.radius = fmax(s.radius, 3.0 * s.dogleg_step_norm);
                        s.mu = fmax(1e-8, 2.0 * s.mu / 10.0);
                        s.reuse = 0;
                        s.cost = ccost; s.grad_max = s.cand_grad_max;
                        s.acc_cand += s.mcc; s.acc_ref += s.mcc;
                        if (s.cost < s.minimum_cost) { s.minimum_cost = s.cost; s.n_nonmono = 0; s.candidate_cost = s.cost; s.acc_cand = 0; }
                        else { ++s.n_nonmono; if (s.cost > s.candidate_cost) { s.candidate_cost = s.cost; s.acc_cand = 0; } }
                        if (s.n_nonmono == o.max_nonmono) { s.reference_cost = s.candidate_cost; s.acc_ref = s.acc_cand; }
                        if (s.cost < s.user_min_cost) { s.user_min_cost = s.cost; sh_copy = 1; }     // the user's parameters follow the best point only
                    } else { s.radius *= 0.5; s.reuse = 1; }
                }
                s.invalid = 0;
            } else {                                                  // the model cost change was not positive: an invalid step
                if (++s.invalid >= 5) { s.done = 1; s.termination = GLIO_TERM_FAILURE; } else { s.mu *= 10.0; s.reuse = 0; }
            }
            // FinalizeIterationAndCheckIfMinimizerCanContinue, then the next iteration
            while (!s.done && !retry) {
                if (s.iteration >= o.max_iterations) { s.done = 1; s.termination = GLIO_TERM_NO_CONVERGENCE; break; }
                if (s.grad_max <= o.gtol) { s.done = 1; s.termination = GLIO_TERM_GRADIENT_TOL; break; }
                if (s.radius <= o.min_radius) { s.done = 1; s.termination = GLIO_TERM_MIN_RADIUS; break; }
                ++s.iteration;
                if (s.reuse || s.mu < 1.0) break;
                // the linear solve cannot even be attempted (mu >= max_mu): this iteration's step is invalid at once
                if (++s.invalid >= 5) { s.done = 1; s.termination = GLIO_TERM_FAILURE; break; }
                s.mu *= 10.0; s.reuse = 0;
            }
            s.step_pending = 0; s.solve_failed = 0; s.retry = retry ? 1 : 0;
            s.skip_solve = (s.done || s.reuse) ? 1 : 0;
            s.step_valid = s.done ? 0 : 1; s.skip_step = s.done ? 1 : 0;
            s.copy_min = sh_copy;
            s.group += 1;
            sh_cur = s.cur & 1;
            *a.st = s;
            if (s.done) { *h.result = s; __threadfence_system(); h.progress[1] = s.solve_id; }
            else { __threadfence_system(); h.progress[0] = (s.solve_id << 16) | (s.group & 0xffff); }
            __threadfence_system();
        }
    }
    // (the copy of the new best point into xmin / smin, flagged by copy_min, is done by k_bt_prepare, which follows with one thread per unknown)
}
#define BT_SKIP_SOLVE(a) ((a).st->skip_solve != 0)
#define BT_SKIP_STEP(a) ((a).st->step_valid == 0)

// scale (first group), D, g_s, g~, u = g~ / D, mu D^2
__global__ void k_bt_prepare(const BtBufs a, const int jacobi) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, n = a.B * a.K;
    if (i >= n) return;
    const int cur = a.st->cur & 1;
    if (a.st->copy_min) {            // the state machine accepted a point that is the best so far: the user's parameters follow it (two entries per thread)
        const int nx = 7 * a.K, ns = a.B == 15 ? 9 * a.K : 0;
        for (int e = i; e < nx + ns; e += n) { if (e < nx) a.xmin[e] = a.x[cur][e]; else a.smin[e - nx] = a.s[cur][e - nx]; }
    }
    if (BT_SKIP_SOLVE(a)) return;
    const double h = a.A[cur][i], g = a.A[cur][n + i];
    double* sc = BVEC(a, V_SC);
    if (a.st->group == 1 && !a.st->retry) sc[i] = jacobi ? 1.0 / (1.0 + sqrt(h)) : 1.0;
    const double s = sc[i];
    double d = s * s * h;
    d = d < 1e-6 ? 1e-6 : (d > 1e32 ? 1e32 : d);
    const double D = sqrt(d);
    BVEC(a, V_DG)[i] = D; BVEC(a, V_GS)[i] = s * g; BVEC(a, V_GR)[i] = s * g / D; BVEC(a, V_UU)[i] = (s * g / D) / D;
    BVEC(a, V_DA)[i] = a.st->mu * D * D;
}
// y = Hs x for the OWNED rows (Hs = S H S on the fly) of up to two vectors.  One wavefront per owned keyframe: the scaled inputs of
// the 2 band + 1 neighbouring keyframes go to LDS; then two passes whose loads are independent of each other (a lane's six or sixteen
// matrix entries are fetched together):
//   pose band   lane (r < 6, part of 8): row r of the 6 x 6 blocks (k, kb), kb = kb0 + part, + 8 -- from the stored block for kb >= k,
//               from its transpose for kb < k;
//   IMU chain   lane (r < 15, part of 4): row r of edge k's record against [x_k | x_k+1] and row 15 + r of edge k - 1's against
//               [x_k-1 | x_k], every fourth of the 30 columns.
// The parts meet by shuffles; the pose rows' sums travel through LDS to the lanes that write.
__global__ __launch_bounds__(64) void k_bt_matvec(const BtBufs a, const int solve_phase, const int vx0, const int vy0, const int vx1, const int vy1) {
    __shared__ double xs[2][(2 * 16 + 1) * 15];
    __shared__ double ys[2][8];
    if (solve_phase ? BT_SKIP_SOLVE(a) : BT_SKIP_STEP(a)) return;
    const int B = a.B, K = a.K, band = a.band, bw = band + 1, k = a.lo + blockIdx.x, lane = threadIdx.x;
    if (k >= a.hi) return;
    const HView v = bt_view(a, a.st->cur & 1);
    const double* sc = BVEC(a, V_SC);
    const double* x0 = BVEC(a, vx0);
    const double* x1 = vx1 >= 0 ? BVEC(a, vx1) : x0;
    const int kb0 = k - band < 0 ? 0 : k - band, kb1 = k + band >= K ? K - 1 : k + band, nkb = kb1 - kb0 + 1;
    for (int e = lane; e < nkb * B; e += 64) { const size_t i = (size_t)kb0 * B + e; const double sv = sc[i]; xs[0][e] = sv * x0[i]; xs[1][e] = sv * x1[i]; }
    __syncthreads();
    {   // ---- pose band
        const int r = lane & 7, part = lane >> 3;
        double s0 = 0, s1 = 0;
        if (r < 6) {
            for (int kbi = part; kbi < nkb; kbi += 8) {
                const int kb = kb0 + kbi, d = kb - k;
                const double* p = d >= 0 ? v.Hg + ((size_t)k * bw + d) * 36 + r * 6 : v.Hg + ((size_t)kb * bw - d) * 36 + r;
                const int stride = d >= 0 ? 1 : 6;
                double hv[6];
#pragma unroll
                for (int c = 0; c < 6; ++c) hv[c] = p[c * stride];
#pragma unroll
                for (int c = 0; c < 6; ++c) { s0 += hv[c] * xs[0][kbi * B + c]; s1 += hv[c] * xs[1][kbi * B + c]; }
            }
        }
        s0 = lane_xor_sum<8>(s0); s0 = lane_xor_sum<16>(s0); s0 = lane_xor_sum<32>(s0);
        s1 = lane_xor_sum<8>(s1); s1 = lane_xor_sum<16>(s1); s1 = lane_xor_sum<32>(s1);
        if (lane < 8) { ys[0][lane] = s0; ys[1][lane] = s1; }
    }
    double t0 = 0, t1 = 0;
    if (B == 15 && v.imu) {   // ---- IMU chain
        const int r = lane & 15, part = lane >> 4, kc = k - kb0;
        if (r < 15) {
            double ha[8], hb[8];
            const bool ea = k < K - 1, eb = k > 0;
            const double* pa = v.imu[ea ? k : 0].H + r * 30;
            const double* pb = v.imu[eb ? k - 1 : 0].H + (15 + r) * 30;
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int q = part + 4 * u; ha[u] = (ea && q < 30) ? pa[q] : 0.0; hb[u] = (eb && q < 30) ? pb[q] : 0.0; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = part + 4 * u;
                if (q >= 30) continue;
                const int ia = (q < 15 ? kc * B + q : (kc + 1) * B + q - 15), ib = (q < 15 ? (kc - 1) * B + q : kc * B + q - 15);
                if (ea) { t0 += ha[u] * xs[0][ia]; t1 += ha[u] * xs[1][ia]; }
                if (eb) { t0 += hb[u] * xs[0][ib]; t1 += hb[u] * xs[1][ib]; }
            }
        }
        t0 = lane_xor_sum<16>(t0); t0 = lane_xor_sum<32>(t0);
        t1 = lane_xor_sum<16>(t1); t1 = lane_xor_sum<32>(t1);
    }
    __syncthreads();
    if (lane < B) {
        const size_t i = (size_t)k * B + lane;
        const double p0 = lane < 6 ? ys[0][lane] : 0.0, p1 = lane < 6 ? ys[1][lane] : 0.0;
        BVEC(a, vy0)[i] = sc[i] * (p0 + t0);
        if (vx1 >= 0) BVEC(a, vy1)[i] = sc[i] * (p1 + t1);
    }
}
// up to six dot products in one pass: BT_DOT_BLOCKS workgroups write their parts, the last one to finish adds them in block order
// (fixed order: every rank gets the same bits from the same data)
#define BT_DOT_BLOCKS 32
struct DotJobs { int n; int va[6], vb[6], owned[6]; double* out[6]; double* parts; unsigned int* ticket; };
__global__ __launch_bounds__(256) void k_bt_dots(const BtBufs a, const int solve_phase, const DotJobs j) {
    __shared__ double red[6][4];
    __shared__ int s_last;
    if (solve_phase ? BT_SKIP_SOLVE(a) : BT_SKIP_STEP(a)) return;
    const int n = a.B * a.K, o0 = a.lo * a.B, o1 = a.hi * a.B;
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += BT_DOT_BLOCKS * 256) {
        const bool own = i >= o0 && i < o1;
#pragma unroll
        for (int q = 0; q < 6; ++q)
            if (q < j.n && (own || !j.owned[q])) s[q] += BVEC(a, j.va[q])[i] * BVEC(a, j.vb[q])[i];
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) { s[q] = wave_sum(s[q]); if ((threadIdx.x & 63) == 0) red[q][threadIdx.x >> 6] = s[q]; }
    __syncthreads();
    if (threadIdx.x < 6) j.parts[blockIdx.x * 6 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(j.ticket, 1u) == BT_DOT_BLOCKS - 1;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (threadIdx.x < j.n) {
        double t = 0;
        for (int b2 = 0; b2 < BT_DOT_BLOCKS; ++b2) t += __builtin_nontemporal_load(&j.parts[b2 * 6 + threadIdx.x]);
        *j.out[threadIdx.x] = t;
    }
    if (threadIdx.x == 0) *j.ticket = 0;
}
// zero the step buffer outside the owned range and publish the factorisation's failure flag behind it (the buffer is all-reduced)
__global__ void k_bt_dz_prepare(const BtBufs a, double* dz, const int* fail) {
    if (BT_SKIP_SOLVE(a)) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x, n = a.B * a.K;
    if (i < n) { if (i < a.lo * a.B || i >= a.hi * a.B) dz[i] = 0.0; }
    else if (i == n) dz[n] = *fail ? 1.0 : 0.0;
}
// gn = D v; a failed factorisation (on any rank) invalidates the group
__global__ void k_bt_gn(const BtBufs a, const double* __restrict__ dz) {
    if (BT_SKIP_SOLVE(a)) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x, n = a.B * a.K;
    const bool bad = dz[n] != 0.0;
    if (i == 0 && bad) { a.st->solve_failed = 1; a.st->step_valid = 0; a.st->skip_step = 1; }
    if (i < n) BVEC(a, V_GN)[i] = bad ? 0.0 : BVEC(a, V_DG)[i] * dz[i];
}
// DoglegStrategy::ComputeSubspaceModel, first half: u1 = the longer of {g~, gn} normalised, w2 = the other one minus its part along u1
__global__ void k_bt_basis(const BtBufs a, const double* __restrict__ puu_sum) {
    if (BT_SKIP_SOLVE(a)) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x, n = a.B * a.K;
    const double gg = a.st->gg, nn2 = a.st->nn2, gd = a.st->gd;
    if (i == 0) { a.st->puu = *puu_sum; a.st->alpha = gg / *puu_sum; }
    if (i >= n) return;
    const bool swap = nn2 > gg;
    const double c0 = swap ? BVEC(a, V_GN)[i] : BVEC(a, V_GR)[i], c1 = swap ? BVEC(a, V_GR)[i] : BVEC(a, V_GN)[i];
    const double n0 = swap ? nn2 : gg;
    const double u1 = c0 / sqrt(n0), w2 = c1 - (gd / n0) * c0;
    const double D = BVEC(a, V_DG)[i];
    BVEC(a, V_U1)[i] = u1; BVEC(a, V_W2)[i] = w2; BVEC(a, V_X1)[i] = u1 / D; BVEC(a, V_X2)[i] = w2 / D;
}

// ---- the boundary-constrained two-dimensional problem (DoglegStrategy::FindMinimumOnTrustRegionBoundary)
struct cplx { double re, im; };
__device__ __forceinline__ cplx c_mul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cplx c_div(cplx a, cplx b) { const double d = b.re * b.re + b.im * b.im; return {(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d}; }
__device__ __forceinline__ double c_abs(cplx a) { return hypot(a.re, a.im); }
// real parts of the four roots of c[0] y^4 + ... + c[4] (c[0] > 0) by the Aberth-Ehrlich iteration
__device__ bool bt_quartic_roots_real(const double* c, double* re) {
    double m[5];
    for (int i = 0; i <= 4; ++i) { m[i] = c[i] / c[0]; if (!isfinite(m[i])) return false; }
    double bound = 0, rad = 0;
    for (int i = 1; i <= 4; ++i) { bound = fmax(bound, fabs(m[i])); rad = fmax(rad, pow(4.0 * fabs(m[i]), 1.0 / i)); }
    bound += 1.0;
    if (!(rad > 0)) { for (int i = 0; i < 4; ++i) re[i] = 0.0; return true; }
    if (rad > bound) rad = bound;
    cplx z[4];
    for (int i = 0; i < 4; ++i) { double sn, cs; sincos(2.0 * M_PI * i / 4 + 0.4, &sn, &cs); z[i] = {rad * cs, rad * sn}; }
    for (int it = 0; it < 100; ++it) {
        double move = 0, size = 0;
        for (int i = 0; i < 4; ++i) {
            cplx pv = {1.0, 0.0}, dv = {0.0, 0.0};
            for (int k = 1; k <= 4; ++k) { dv = c_mul(dv, z[i]); dv.re += pv.re; dv.im += pv.im; pv = c_mul(pv, z[i]); pv.re += m[k]; }
            if (c_abs(pv) == 0.0) continue;
            cplx s = {0.0, 0.0};
            for (int j = 0; j < 4; ++j) if (j != i) { const cplx d = {z[i].re - z[j].re, z[i].im - z[j].im}; const cplx q = c_div({1.0, 0.0}, d); s.re += q.re; s.im += q.im; }
            const cplx nw = c_div(pv, dv);
            const cplx ns = c_mul(nw, s);
            const cplx w = c_div(nw, {1.0 - ns.re, -ns.im});
            if (!isfinite(w.re) || !isfinite(w.im)) continue;
            z[i].re -= w.re; z[i].im -= w.im;
            move = fmax(move, c_abs(w)); size = fmax(size, c_abs(z[i]));
        }
        if (move <= 4e-15 * size) break;
    }
    for (int i = 0; i < 4; ++i) { re[i] = z[i].re; if (!isfinite(re[i])) return false; }
    return true;
}
__device__ bool bt_boundary_minimum(const double b00, const double b01, const double b11, const double g0, const double g1, const double radius, double* m0, double* m1) {
    const double detB = b00 * b11 - b01 * b01, trB = b00 + b11, r2 = radius * radius;
    const double ag0 = b11 * g0 - b01 * g1, ag1 = -b01 * g0 + b00 * g1;
    double poly[5], roots[4];
    poly[0] = r2;
    poly[1] = 2.0 * r2 * trB;
    poly[2] = r2 * (trB * trB + 2.0 * detB) - (g0 * g0 + g1 * g1);
    poly[3] = -2.0 * ((g0 * ag0 + g1 * ag1) - r2 * detB * trB);
    poly[4] = r2 * detB * detB - (ag0 * ag0 + ag1 * ag1);
    if (!bt_quartic_roots_real(poly, roots)) return false;
    double best = 1.7976931348623157e308;
    bool found = false;
    for (int i = 0; i < 4; ++i) {
        // -(B + y I)^-1 g by a partially pivoted 2x2 LU
        double a = b00 + roots[i], b = b01, c = b01, d = b11 + roots[i], r0 = g0, r1 = g1;
        if (fabs(c) > fabs(a)) { double t = a; a = c; c = t; t = b; b = d; d = t; t = r0; r0 = r1; r1 = t; }
        const double l = c / a, u11 = d - l * b, y1 = r1 - l * r0;
        const double x1 = -(y1 / u11), x0 = -((r0 - b * (y1 / u11)) / a);
        const double nx = sqrt(x0 * x0 + x1 * x1);
        if (nx > 0) {
            const double s0 = radius / nx * x0, s1 = radius / nx * x1;
            const double f = 0.5 * (s0 * (b00 * s0 + b01 * s1) + s1 * (b01 * s0 + b11 * s1)) + g0 * s0 + g1 * s1;
            found = true;
            if (f < best) { best = f; *m0 = x0; *m1 = x1; }
        }
    }
    return found;
}
// curvature of the basis from the all-reduced sums (stage D), then the step coefficients for the present radius
__global__ void k_bt_dogleg(const BtBufs a, const double* __restrict__ stage_d, const int dogleg_type) {
    if (threadIdx.x != 0 || blockIdx.x != 0 || BT_SKIP_STEP(a)) return;
    BtStatus& s = *a.st;
    if (!s.skip_solve) {          // fresh Gauss-Newton / Cauchy data: (re)build the subspace model
        s.p11 = stage_d[0]; s.p12 = stage_d[1]; s.p22 = stage_d[2];
        const double n0 = fmax(s.gg, s.nn2);
        s.one_dim = sqrt(s.w2) <= sqrt(n0) * (2.220446049250313e-16 * 2.0) ? 1 : 0;
    }
    const double gg = s.gg, nn2 = s.nn2, gd = s.gd, radius = s.radius, alpha = s.alpha;
    const double gnorm = sqrt(gg), gnn = sqrt(nn2);
    double cg = 0, cn = 0, c1 = 0, c2 = 0, norm = -1.0;
    bool traditional = dogleg_type != GLIO_DOGLEG_SUBSPACE;
    if (!traditional) {
        if (gnn <= radius) { cn = 1.0; norm = gnn; }
        else if (s.one_dim) { cg = -(radius / gnorm); norm = radius; }
        else {
            const bool swap = nn2 > gg;
            const double wn = sqrt(s.w2);
            // g2 = U^T g~ ; w2 . g~ = (c1 - (gd / n0) c0) . g~
            const double n0 = swap ? nn2 : gg;
            const double g0 = (swap ? gd : gg) / sqrt(n0);
            const double g1 = ((swap ? gg : gd) - (gd / n0) * (swap ? gd : gg)) / wn;
            double m0 = 0, m1 = 0;
            if (bt_boundary_minimum(s.p11, s.p12 / wn, s.p22 / s.w2, g0, g1, radius, &m0, &m1)) { c1 = m0; c2 = m1 / wn; norm = radius; }
            else traditional = true;
        }
    }
    if (traditional) {
        if (gnn <= radius) { cg = 0; cn = 1.0; norm = gnn; }
        else if (gnorm * alpha >= radius) { cg = -(radius / gnorm); cn = 0; norm = radius; }
        else {
            const double b_dot_a = -alpha * gd, a_sq = alpha * alpha * gg;
            const double b_minus_a_sq = nn2 - 2 * b_dot_a + a_sq, c = b_dot_a - a_sq;
            const double d = sqrt(c * c + b_minus_a_sq * (radius * radius - a_sq));
            const double beta = (c <= 0) ? (d - c) / b_minus_a_sq : (radius * radius - a_sq) / (d + c);
            cg = -alpha * (1.0 - beta); cn = beta; norm = -1.0;
        }
    }
    s.c_g = cg; s.c_n = cn; s.c_1 = c1; s.c_2 = c2; s.dogleg_step_norm = norm;
}
// step in D-space, in the scaled variables, in the parameters; the candidate x (+) delta; one thread per keyframe
#define BT_STEP_KF 16
__global__ __launch_bounds__(BT_STEP_KF * 15) void k_bt_step(const BtBufs a) {
    __shared__ double sdl[BT_STEP_KF * 15];
    if (BT_SKIP_STEP(a)) return;
    const int B = a.B, tid = threadIdx.x;
    const bool active = tid < BT_STEP_KF * B;
    const int kl = tid / B, r = tid - B * kl, k = active ? blockIdx.x * BT_STEP_KF + kl : a.K;
    const BtStatus& s = *a.st;
    const int cur = s.cur & 1, cand = cur ^ 1;
    double dl = 0.0;
    if (k < a.K) {
        const size_t i = (size_t)k * B + r;
        const double sD = s.c_g * BVEC(a, V_GR)[i] + s.c_n * BVEC(a, V_GN)[i] + s.c_1 * BVEC(a, V_U1)[i] + s.c_2 * BVEC(a, V_W2)[i];
        const double st = sD / BVEC(a, V_DG)[i];
        BVEC(a, V_SD)[i] = sD; BVEC(a, V_ST)[i] = st;
        dl = BVEC(a, V_SC)[i] * st;
        BVEC(a, V_DL)[i] = dl;
    }
    sdl[tid] = dl;
    __syncthreads();
    if (k >= a.K) return;
    const double* x = a.x[cur] + 7 * (size_t)k;
    double* xo = a.x[cand] + 7 * (size_t)k;
    if (r < 3) xo[r] = x[r] + dl;
    else if (r == 3) d_quat_plus(x + 3, sdl + tid, xo + 3);
    else if (r >= 6) a.s[cand][9 * (size_t)k + r - 6] = a.s[cur][9 * (size_t)k + r - 6] + dl;
}
// model cost change = -(gs . s + s^T Hs s / 2) with the all-reduced quadratic term (stage E)
__global__ void k_bt_mcc(const BtBufs a, const double* __restrict__ stage_e) {
    if (threadIdx.x != 0 || blockIdx.x != 0 || BT_SKIP_STEP(a)) return;
    BtStatus& s = *a.st;
    s.quad = stage_e[0];
    s.mcc = -(s.lin + 0.5 * s.quad);
    if (s.dogleg_step_norm < 0) s.dogleg_step_norm = sqrt(s.sd2);
    if (!(s.mcc > 0.0)) { s.step_valid = 0; s.skip_step = 1; }
}

// ------------------------------------------------------------------------------------------------ host
#define BT_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { glio_set_error("%s failed: %s", #expr, hipGetErrorString(e_)); return GLIO_E_HIP; } } while (0)

struct BatchSmall {
    int n_dq, n_dd, n_fac;
    int* d_fa; int* d_fb; int* d_ftype; int* d_fidx;      // sorted by ordered pair (a, b); only the factors this rank owns (a in [lo, hi))
    double* d_dq_const; glio_dd_psr* d_dd;
    // LidarPoseFactorBatchRelativeAutoDiff factors (glio_batch_set_relative_pose_factors): kept on the host until glio_batch_set_small_factors builds the table
    int n_rp; int* h_rp_i; int* h_rp_j; double* h_rp_c; double* d_rp_const; size_t cap_rp;
    int2* d_small_index;       // [K][2 band + 1] (first, count) of the factors with (a = k, b = k + o)
    double* d_frec;            // [n_fac][SREC]
    size_t cap_fac, cap_dd;
    double R_ecef_local[9], anc[3];
    double* d_rel;             // R_ecef_local (9) and the anchor (3)
    // shard
    int rank, world, lo, hi;
    int* d_bnd_kf;
    // IMU chain
    int n_imu; ImuEdgeDev* d_imu; PairBlock* d_rec[2]; double gravity;
    // trust region
    int B;                     // unknowns per keyframe of the configured solve (6 / 15)
    void* bcr; int bcr_B, bcr_rank, bcr_world;
    double* d_vec; long long vstride;
    double* d_hg[2]; double* d_A[2]; double* d_x[2]; double* d_s[2]; double* d_xmin; double* d_smin;
    double* d_stage; long long stage_doubles, bnd_doubles;     // assembly all-reduce buffer
    double* d_dz;              // [n + 2] Gauss-Newton step of the scaled system (all-reduced), then the failure flag
    double* d_scal;            // [16] stage D (0..7) and stage E (8..15)
    double* d_dot_parts; unsigned int* d_dot_ticket;      // k_bt_dots: per-workgroup parts and the arrival counter
    BtStatus* d_st;
    int* h_prog; int* d_prog; BtStatus* h_res; BtStatus* d_res;     // mapped host memory
    double* h_x;               // pinned [K][16]
    int solve_id;
    int enqueue_lead;          // trust-region groups kept in flight by glio_batch_solve_tr2 (0 = default 2; 1 = wait for every group's decision)
    long long hook_calls, hook_doubles, groups;
    // the small factors and the IMU edges are evaluated on a second stream while K8 streams this rank's constraints (they are latency-bound
    // launches of a few thousand wavefronts; K8 is bandwidth-bound): forked and joined with events inside enqueue_tr_linearize
    hipStream_t side; hipEvent_t ev_fork, ev_join;
};
void glio_host_ecef_local(const double anc[3], double yaw, double R[9]);     // capi.hip

static void shard_range(int K, int band, int rank, int world, int* lo, int* hi) {
    const int sbk = band <= 6 ? 6 : 12, S = (K + sbk - 1) / sbk;
    const int Slo = (int)((long long)S * rank / world), Shi = (int)((long long)S * (rank + 1) / world);
    *lo = std::min(K, Slo * sbk); *hi = std::min(K, Shi * sbk);
}

static int small_ensure(glio_batch* b) {
    if (b->small) return GLIO_OK;
    BatchSmall* s = new BatchSmall();
    memset(s, 0, sizeof *s);
    const int K = b->K, band = b->band;
    s->rank = 0; s->world = 1; s->lo = 0; s->hi = K; s->B = 6;
    BT_CHECK(hipMalloc((void**)&s->d_small_index, (size_t)K * (2 * band + 1) * sizeof(int2)));
    BT_CHECK(hipMemset(s->d_small_index, 0, (size_t)K * (2 * band + 1) * sizeof(int2)));
    BT_CHECK(hipMalloc((void**)&s->d_rel, 12 * 8));
    BT_CHECK(hipMemset(s->d_rel, 0, 12 * 8));
    BT_CHECK(hipMalloc((void**)&s->d_bnd_kf, 64 * 4));
    BT_CHECK(hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking));
    BT_CHECK(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
    BT_CHECK(hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming));
    b->small = s;
    return GLIO_OK;
}
static void tr_free(BatchSmall* s) {
    void* p[] = {s->d_vec, s->d_hg[0], s->d_hg[1], s->d_A[0], s->d_A[1], s->d_x[0], s->d_x[1], s->d_s[0], s->d_s[1], s->d_xmin, s->d_smin, s->d_stage, s->d_dz,
                 s->d_scal, s->d_st, s->d_dot_parts, s->d_dot_ticket};
    for (void* q : p) if (q) hipFree(q);
    if (s->h_prog) hipHostFree(s->h_prog);
    if (s->h_res) hipHostFree(s->h_res);
    if (s->h_x) hipHostFree(s->h_x);
    s->d_vec = nullptr; s->d_hg[0] = s->d_hg[1] = s->d_A[0] = s->d_A[1] = s->d_x[0] = s->d_x[1] = s->d_s[0] = s->d_s[1] = s->d_xmin = s->d_smin = s->d_stage = s->d_dz = s->d_scal = nullptr; s->d_dot_parts = nullptr; s->d_dot_ticket = nullptr;
    s->d_st = nullptr; s->h_prog = nullptr; s->h_res = nullptr; s->h_x = nullptr;
    if (s->bcr) { glio_bcr_destroy(s->bcr); s->bcr = nullptr; }
}
void glio_batch_small_destroy(glio_batch* b) {
    BatchSmall* s = b->small;
    if (!s) return;
    tr_free(s);
    void* p[] = {s->d_fa, s->d_fb, s->d_ftype, s->d_fidx, s->d_dq_const, s->d_dd, s->d_small_index, s->d_frec, s->d_rel, s->d_bnd_kf, s->d_imu, s->d_rec[0], s->d_rec[1], s->d_rp_const};
    for (void* q : p) if (q) hipFree(q);
    free(s->h_rp_i); free(s->h_rp_j); free(s->h_rp_c);
    if (s->side) hipStreamDestroy(s->side);
    if (s->ev_fork) hipEventDestroy(s->ev_fork);
    if (s->ev_join) hipEventDestroy(s->ev_join);
    delete s;
    b->small = nullptr;
}

// workspaces of the trust-region solve for the present (B, rank, world); (re)built when one of them changed
static int tr_build(glio_batch* b, const int B) {
    BatchSmall* s = b->small;
    const int K = b->K, band = b->band;
    if (s->world - 1 > 64) { glio_set_error("more than 65 ranks"); return GLIO_E_ARG; }
    s->B = B;
    s->bcr = glio_bcr_create2(K, band, B, s->rank, s->world);
    if (!s->bcr) return GLIO_E_ARG;
    int lo, hi;
    glio_bcr_owned_range(s->bcr, &lo, &hi);
    if (lo != s->lo || hi != s->hi) { glio_set_error("shard range mismatch"); return GLIO_E_STATE; }
    const long long n = (long long)B * K;
    s->vstride = (n + 15) / 16 * 16;
    const size_t hg = (size_t)glio_batch_hg_size(K, band);
    const int NB = s->world - 1;
    s->bnd_doubles = (long long)NB * 2 * band * (band + 1) * 36;
    s->stage_doubles = s->bnd_doubles + 2 * n + 2;
    BT_CHECK(hipMalloc((void**)&s->d_vec, (size_t)V_COUNT * s->vstride * 8));
    BT_CHECK(hipMemset(s->d_vec, 0, (size_t)V_COUNT * s->vstride * 8));
    for (int k = 0; k < 2; ++k) {
        BT_CHECK(hipMalloc((void**)&s->d_hg[k], hg * 8)); BT_CHECK(hipMemset(s->d_hg[k], 0, hg * 8));
        BT_CHECK(hipMalloc((void**)&s->d_A[k], (size_t)(2 * n + 2) * 8));
        BT_CHECK(hipMalloc((void**)&s->d_x[k], (size_t)K * 7 * 8)); BT_CHECK(hipMalloc((void**)&s->d_s[k], (size_t)K * 9 * 8));
        BT_CHECK(hipMemset(s->d_s[k], 0, (size_t)K * 9 * 8));
    }
    BT_CHECK(hipMalloc((void**)&s->d_xmin, (size_t)K * 7 * 8)); BT_CHECK(hipMalloc((void**)&s->d_smin, (size_t)K * 9 * 8));
    BT_CHECK(hipMemset(s->d_smin, 0, (size_t)K * 9 * 8));
    BT_CHECK(hipMalloc((void**)&s->d_stage, (size_t)s->stage_doubles * 8));
    BT_CHECK(hipMalloc((void**)&s->d_dz, (size_t)(n + 2) * 8));
    BT_CHECK(hipMalloc((void**)&s->d_scal, 16 * 8)); BT_CHECK(hipMemset(s->d_scal, 0, 16 * 8));
    BT_CHECK(hipMalloc((void**)&s->d_dot_parts, (BT_DOT_BLOCKS * 6 + 3 * (size_t)((K + BT_NORM_THREADS - 1) / BT_NORM_THREADS)) * 8)); BT_CHECK(hipMalloc((void**)&s->d_dot_ticket, 16)); BT_CHECK(hipMemset(s->d_dot_ticket, 0, 16));
    BT_CHECK(hipMalloc((void**)&s->d_st, sizeof(BtStatus)));
    BT_CHECK(hipHostMalloc((void**)&s->h_prog, 64, hipHostMallocMapped | hipHostMallocCoherent));
    BT_CHECK(hipHostGetDevicePointer((void**)&s->d_prog, (void*)s->h_prog, 0));
    BT_CHECK(hipHostMalloc((void**)&s->h_res, sizeof(BtStatus), hipHostMallocMapped | hipHostMallocCoherent));
    BT_CHECK(hipHostGetDevicePointer((void**)&s->d_res, (void*)s->h_res, 0));
    BT_CHECK(hipHostMalloc((void**)&s->h_x, (size_t)K * 16 * 8 + sizeof(BtStatus) + 64));
    memset(s->h_prog, 0, 64);
    std::vector<int> bnd(std::max(NB, 1), 0);
    for (int j = 0; j < NB; ++j) { int l, h2; shard_range(K, band, j, s->world, &l, &h2); bnd[j] = h2; }
    BT_CHECK(hipMemcpy(s->d_bnd_kf, bnd.data(), (size_t)std::max(NB, 1) * 4, hipMemcpyHostToDevice));
    return GLIO_OK;
}
static int tr_ensure(glio_batch* b) {
    BatchSmall* s = b->small;
    const int B = s->n_imu > 0 ? 15 : 6;
    if (s->bcr && s->bcr_B == B && s->bcr_rank == s->rank && s->bcr_world == s->world) return GLIO_OK;
    tr_free(s);
    const int rc = tr_build(b, B);
    if (rc != GLIO_OK) { tr_free(s); return rc; }     // a half-built set (an allocation failed midway) must not pass the check above next time
    s->bcr_B = B; s->bcr_rank = s->rank; s->bcr_world = s->world;   // the cache keys only once every buffer exists
    return GLIO_OK;
}

static BtBufs make_bufs(glio_batch* b) {
    BatchSmall* s = b->small;
    BtBufs a;
    a.st = s->d_st; a.K = b->K; a.band = b->band; a.B = s->B; a.lo = s->lo; a.hi = s->hi;
    for (int k = 0; k < 2; ++k) { a.x[k] = s->d_x[k]; a.s[k] = s->d_s[k]; a.hg[k] = s->d_hg[k]; a.rec[k] = s->n_imu > 0 ? s->d_rec[k] : nullptr; a.A[k] = s->d_A[k]; }
    a.xmin = s->d_xmin; a.smin = s->d_smin; a.vec = s->d_vec; a.vstride = s->vstride;
    return a;
}
static BtAsm make_asm(glio_batch* b) {
    BatchSmall* s = b->small;
    BtAsm m;
    m.NB = s->world - 1; m.bnd_doubles = s->bnd_doubles; m.bnd_kf = s->d_bnd_kf; m.stage = s->d_stage; m.rank = s->rank;
    m.eo0 = s->n_imu > 0 ? s->lo : 0; m.eo1 = s->n_imu > 0 ? std::min(s->hi, b->K - 1) : 0;
    return m;
}

// small factors of this rank evaluated at the selected poses and added into the selected band buffer
static void enqueue_small_eval(glio_batch* b, const BtSel& sel, const double* p0, const double* p1, hipStream_t st) {
    BatchSmall* s = b->small;
    if (!s || s->n_fac == 0) return;
    hipLaunchKernelGGL(k_small_eval, dim3(s->n_fac), dim3(64), 0, st, sel, s->n_fac, s->d_fa, s->d_fb, s->d_ftype, s->d_fidx, p0, p1, s->d_dq_const, s->d_rp_const, s->d_dd,
                       s->d_rel, s->d_frec);
}
static void enqueue_small_accumulate(glio_batch* b, const BtSel& sel, double* Hg0, double* Hg1) {
    BatchSmall* s = b->small;
    if (!s || s->n_fac == 0) return;
    const int K = b->K, band = b->band;
    const long long tot = glio_batch_hg_size(K, band);
    hipLaunchKernelGGL(k_small_add, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, b->stream, sel, K, band, s->d_small_index, s->d_frec, s->n_fac, Hg0, Hg1);
    hipLaunchKernelGGL(k_small_cost, dim3(1), dim3(256), 0, b->stream, sel, s->d_frec, s->n_fac, Hg0 + tot - 1, Hg1 + tot - 1);
}
static void enqueue_small(glio_batch* b, const double* poses_dev, double* Hg_dev) {
    BtSel sel; sel.cur = nullptr; sel.skip = nullptr; sel.want = 0;
    enqueue_small_eval(b, sel, poses_dev, poses_dev, b->stream);
    enqueue_small_accumulate(b, sel, Hg_dev, Hg_dev);
}

// GLIO_BATCH_MOMENTS=0: every linearisation streams the constraints (k_batch_pairs), as before round 3's moment form -- kept for A/B runs and parity
static int g_batch_moments = getenv("GLIO_BATCH_MOMENTS") ? atoi(getenv("GLIO_BATCH_MOMENTS")) : 1;
typedef void (*bt_hook_fn)(double*, int64_t, void*, void*);
struct BtRun { glio_batch* b; bt_hook_fn hook; void* user; };
static void call_hook(const BtRun& r, double* dev, long long count) {
    BatchSmall* s = r.b->small;
    if (s->world <= 1 || !r.hook) return;
    r.hook(dev, (int64_t)count, (void*)r.b->stream, r.user);
    s->hook_calls += 1; s->hook_doubles += count;
}

// linearise the point selected by `want` (1: the candidate) -- K8 on this rank's constraints, its small factors, its IMU edges (plus
// the one entering from the left neighbour, for complete diagonal blocks) -- then the assembly all-reduce and the candidate's norms
static void enqueue_tr_linearize(const BtRun& r, int want, bool initial) {
    glio_batch* b = r.b;
    BatchSmall* s = b->small;
    const BtBufs a = make_bufs(b);
    const BtAsm m = make_asm(b);
    hipStream_t st = b->stream;
    BtSel sel; sel.cur = &s->d_st->cur; sel.skip = initial ? nullptr : &s->d_st->skip_step; sel.want = want;
    const int K = b->K, band = b->band;
    const int k0 = std::max(0, s->lo - band), k1 = std::min(K, s->hi + band);
    const bool fork = s->n_fac > 0 || s->n_imu > 0;
    if (fork) {                                    // small factors + IMU edges on the second stream, under K8
        hipEventRecord(s->ev_fork, st);
        hipStreamWaitEvent(s->side, s->ev_fork, 0);
        enqueue_small_eval(b, sel, s->d_x[0], s->d_x[1], s->side);
        if (s->n_imu > 0) {
            const int e0 = std::max(0, s->lo - 1), e1 = std::min(s->hi, K - 1);
            glio_launch_batch_imu(s->side, sel, s->gravity, s->d_imu, e0, e1, s->d_x[0], s->d_x[1], s->d_s[0], s->d_s[1], s->d_rec[0], s->d_rec[1]);
        }
        hipEventRecord(s->ev_join, s->side);
    }
    // K8: the first linearisation of a solve takes the pairs' moments at its poses (one pass over the constraints), the later ones evaluate them
    glio_batch_enqueue_linearize_sel(b, sel, s->d_x[0], s->d_x[1], s->d_hg[0], s->d_hg[1], k0, k1, g_batch_moments ? (initial ? 1 : 2) : 0);
    if (fork) hipStreamWaitEvent(st, s->ev_join, 0);
    enqueue_small_accumulate(b, sel, s->d_hg[0], s->d_hg[1]);
    const long long tot = s->bnd_doubles + 2LL * a.B * K;
    hipLaunchKernelGGL(k_bt_pack, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, a, m, sel);
    hipLaunchKernelGGL(k_bt_pack_cost, dim3(1), dim3(256), 0, st, a, m, sel);
    call_hook(r, s->d_stage, s->stage_doubles);
    hipLaunchKernelGGL(k_bt_unpack, dim3((unsigned)((tot + 1 + 255) / 256)), dim3(256), 0, st, a, m, sel);
    NormParts np; np.parts = s->d_dot_parts + BT_DOT_BLOCKS * 6; np.ticket = s->d_dot_ticket + 1;
    hipLaunchKernelGGL(k_bt_norms, dim3((K + BT_NORM_THREADS - 1) / BT_NORM_THREADS), dim3(BT_NORM_THREADS), 0, st, a, sel, np);
}

// one trust-region group: state machine, (Cauchy + Gauss-Newton + subspace model), step, model cost change, the candidate's linearisation
static void enqueue_tr_group(const BtRun& r, const BtOpts& o) {
    glio_batch* b = r.b;
    BatchSmall* s = b->small;
    const BtBufs a = make_bufs(b);
    hipStream_t st = b->stream;
    const int K = b->K, B = s->B, n = B * K;
    const int nbv = (n + TRV_THREADS - 1) / TRV_THREADS;
    const int nbo = std::max(1, s->hi - s->lo);
    BtHost h; h.progress = s->d_prog; h.result = s->d_res;
    glio_lds_poison_stream(st);
    hipLaunchKernelGGL(k_bt_state_machine, dim3(1), dim3(64), 0, st, a, o, h);
    // ---- Cauchy point and Gauss-Newton step (skipped when the stored ones are reused)
    hipLaunchKernelGGL(k_bt_prepare, dim3(nbv), dim3(TRV_THREADS), 0, st, a, o.jacobi);
    hipLaunchKernelGGL(k_bt_matvec, dim3(nbo), dim3(64), 0, st, a, 1, (int)V_UU, (int)V_T1, -1, -1);
    long long sep_count = 0;
    double* sep = glio_bcr_sepbuf(s->bcr, &sep_count);
    double* extra = sep + sep_count - 16;
    {
        DotJobs j; memset(&j, 0, sizeof j); j.parts = s->d_dot_parts; j.ticket = s->d_dot_ticket;
        j.n = 2;
        j.va[0] = V_GR; j.vb[0] = V_GR; j.owned[0] = 0; j.out[0] = &s->d_st->gg;
        j.va[1] = V_UU; j.vb[1] = V_T1; j.owned[1] = 1; j.out[1] = extra;          // the Cauchy curvature travels with the separator system
        hipLaunchKernelGGL(k_bt_dots, dim3(BT_DOT_BLOCKS), dim3(256), 0, st, a, 1, j);
    }
    glio_lds_poison_stream(st);
    BcrOp op; memset(&op, 0, sizeof op);
    for (int k = 0; k < 2; ++k) { op.Hg[k] = s->d_hg[k]; op.imu[k] = s->n_imu > 0 ? s->d_rec[k] : nullptr; op.gfull[k] = s->d_A[k] + n; }
    op.cur = &s->d_st->cur; op.sc = BVEC(a, V_SC); op.dadd = BVEC(a, V_DA); op.lambda = 0.0; op.skip = &s->d_st->skip_solve;
    glio_bcr_enqueue_local(s->bcr, op, st);
    call_hook(r, sep, sep_count);
    glio_bcr_enqueue_finish(s->bcr, op, s->d_dz, st);
    hipLaunchKernelGGL(k_bt_dz_prepare, dim3((n + 1 + 255) / 256), dim3(256), 0, st, a, s->d_dz, glio_bcr_fail_flag(s->bcr));
    call_hook(r, s->d_dz, n + 2);
    hipLaunchKernelGGL(k_bt_gn, dim3(nbv), dim3(TRV_THREADS), 0, st, a, s->d_dz);
    {
        DotJobs j; memset(&j, 0, sizeof j); j.parts = s->d_dot_parts; j.ticket = s->d_dot_ticket;
        j.n = 2;
        j.va[0] = V_GN; j.vb[0] = V_GN; j.out[0] = &s->d_st->nn2;
        j.va[1] = V_GR; j.vb[1] = V_GN; j.out[1] = &s->d_st->gd;
        hipLaunchKernelGGL(k_bt_dots, dim3(BT_DOT_BLOCKS), dim3(256), 0, st, a, 1, j);
    }
    hipLaunchKernelGGL(k_bt_basis, dim3(nbv), dim3(TRV_THREADS), 0, st, a, extra);
    hipLaunchKernelGGL(k_bt_matvec, dim3(nbo), dim3(64), 0, st, a, 1, (int)V_X1, (int)V_T1, (int)V_X2, (int)V_T2);
    {
        DotJobs j; memset(&j, 0, sizeof j); j.parts = s->d_dot_parts; j.ticket = s->d_dot_ticket;
        j.n = 4;
        j.va[0] = V_X1; j.vb[0] = V_T1; j.owned[0] = 1; j.out[0] = s->d_scal + 0;
        j.va[1] = V_X1; j.vb[1] = V_T2; j.owned[1] = 1; j.out[1] = s->d_scal + 1;
        j.va[2] = V_X2; j.vb[2] = V_T2; j.owned[2] = 1; j.out[2] = s->d_scal + 2;
        j.va[3] = V_W2; j.vb[3] = V_W2; j.owned[3] = 0; j.out[3] = &s->d_st->w2;
        hipLaunchKernelGGL(k_bt_dots, dim3(BT_DOT_BLOCKS), dim3(256), 0, st, a, 1, j);
    }
    call_hook(r, s->d_scal, 8);
    // ---- the step for the present radius, its model cost change, the candidate
    hipLaunchKernelGGL(k_bt_dogleg, dim3(1), dim3(64), 0, st, a, s->d_scal, o.dogleg_type);
    hipLaunchKernelGGL(k_bt_step, dim3((K + BT_STEP_KF - 1) / BT_STEP_KF), dim3(BT_STEP_KF * 15), 0, st, a);
    hipLaunchKernelGGL(k_bt_matvec, dim3(nbo), dim3(64), 0, st, a, 0, (int)V_ST, (int)V_T1, -1, -1);
    {
        DotJobs j; memset(&j, 0, sizeof j); j.parts = s->d_dot_parts; j.ticket = s->d_dot_ticket;
        j.n = 3;
        j.va[0] = V_GS; j.vb[0] = V_ST; j.out[0] = &s->d_st->lin;
        j.va[1] = V_ST; j.vb[1] = V_T1; j.owned[1] = 1; j.out[1] = s->d_scal + 8;
        j.va[2] = V_SD; j.vb[2] = V_SD; j.out[2] = &s->d_st->sd2;
        hipLaunchKernelGGL(k_bt_dots, dim3(BT_DOT_BLOCKS), dim3(256), 0, st, a, 0, j);
    }
    call_hook(r, s->d_scal + 8, 8);
    hipLaunchKernelGGL(k_bt_mcc, dim3(1), dim3(64), 0, st, a, s->d_scal + 8);
    enqueue_tr_linearize(r, 1, false);          // the candidate's linearisation (returns at once when the group has no usable step)
    s->groups += 1;
}

extern "C" {

int glio_batch_shard_range(int K, int band, int rank, int world, int32_t* lo, int32_t* hi) {
    if (K < 1 || band < 1 || world < 1 || rank < 0 || rank >= world || !lo || !hi) return GLIO_E_ARG;
    int l, h;
    shard_range(K, band, rank, world, &l, &h);
    *lo = l; *hi = h;
    return GLIO_OK;
}
// This object is rank `rank` of `world`: it owns the keyframes of glio_batch_shard_range -- the constraints given to
// glio_batch_set_constraints* must have their source keyframe there; small factors and IMU edges are given whole, the library keeps its own.
int glio_batch_set_shard(glio_batch* b, int rank, int world) {
    if (!b || world < 1 || rank < 0 || rank >= world) return GLIO_E_ARG;
    BT_CHECK(hipSetDevice(b->device));
    { const int rc = small_ensure(b); if (rc) return rc; }
    BatchSmall* s = b->small;
    if (s->n_fac > 0 && (rank != s->rank || world != s->world)) { glio_set_error("glio_batch_set_shard must precede glio_batch_set_small_factors"); return GLIO_E_STATE; }
    const int sbk = b->band <= 6 ? 6 : 12, S = (b->K + sbk - 1) / sbk;
    if (S < world) { glio_set_error("%d keyframes are too few for %d ranks", b->K, world); return GLIO_E_ARG; }
    s->rank = rank; s->world = world;
    shard_range(b->K, b->band, rank, world, &s->lo, &s->hi);
    return GLIO_OK;
}

// The ImuFactor chain (Estimator.cpp:2990-3001): edges[k] is the pre-integration between keyframes k and k + 1 (n_edges = K - 1), or
// n_edges = 0 for the pose-only problem.  With the chain every keyframe has 15 unknowns and glio_batch_solve_tr2 takes the speed-bias blocks.
int glio_batch_set_imu(glio_batch* b, int n_edges, const glio_preint* edges, double gravity) {
    if (!b || (n_edges != 0 && n_edges != b->K - 1) || (n_edges > 0 && !edges)) return GLIO_E_ARG;
    BT_CHECK(hipSetDevice(b->device));
    { const int rc = small_ensure(b); if (rc) return rc; }
    BatchSmall* s = b->small;
    if (n_edges > 0 && !s->d_imu) {
        BT_CHECK(hipMalloc((void**)&s->d_imu, (size_t)(b->K - 1) * sizeof(ImuEdgeDev)));
        for (int k = 0; k < 2; ++k) BT_CHECK(hipMalloc((void**)&s->d_rec[k], (size_t)(b->K - 1) * sizeof(PairBlock)));
    }
    if (n_edges > 0) {
        std::vector<ImuEdgeDev> h((size_t)n_edges);
        for (int k = 0; k < n_edges; ++k)
            if (!glio_digest_imu_edge(&edges[k], k, &h[k])) { glio_set_error("IMU edge %d: covariance not invertible", k); return GLIO_E_ARG; }
        BT_CHECK(hipMemcpy(s->d_imu, h.data(), (size_t)n_edges * sizeof(ImuEdgeDev), hipMemcpyHostToDevice));
    }
    s->n_imu = n_edges; s->gravity = gravity;
    return GLIO_OK;
}

int glio_batch_set_small_factors(glio_batch* b, const glio_gnss_frame* frame, int n_dq, const int32_t* dq_i, const int32_t* dq_j, const double* dq_const,
                                 int n_dd, const glio_dd_psr* dd) {
    if (!b || n_dq < 0 || n_dd < 0 || (n_dq > 0 && (!dq_i || !dq_j || !dq_const)) || (n_dd > 0 && (!dd || !frame))) return GLIO_E_ARG;
    BT_CHECK(hipSetDevice(b->device));
    { const int rc = small_ensure(b); if (rc) return rc; }
    BatchSmall* s = b->small;
    const int K = b->K, band = b->band, wdt = 2 * band + 1;
    struct Fac { int a, b, type, idx; };
    std::vector<Fac> fac;
    fac.reserve((size_t)n_dq + n_dd + s->n_rp);
    for (int f = 0; f < n_dq; ++f) fac.push_back({dq_i[f], dq_j[f], 0, f});
    for (int f = 0; f < n_dd; ++f) fac.push_back({dd[f].slot_i, dd[f].slot_j, 1, f});
    for (int f = 0; f < s->n_rp; ++f) fac.push_back({s->h_rp_i[f], s->h_rp_j[f], 2, f});
    for (const Fac& f : fac) {
        if (f.a < 0 || f.a >= K || f.b < 0 || f.b >= K || f.a == f.b || std::abs(f.a - f.b) > band) { glio_set_error("small factor on keyframes (%d, %d) outside band %d", f.a, f.b, band); return GLIO_E_ARG; }
        if (f.type == 1 && (dd[f.idx].n_sat < 2 || dd[f.idx].n_sat > GLIO_DD_MAX_SAT || dd[f.idx].master < 0 || dd[f.idx].master >= dd[f.idx].n_sat)) { glio_set_error("bad DD factor"); return GLIO_E_ARG; }
    }
    // sharded by the first keyframe of the factor: this rank keeps its own
    std::vector<Fac> mine;
    for (const Fac& f : fac) if (f.a >= s->lo && f.a < s->hi) mine.push_back(f);
    fac.swap(mine);
    const int nf = (int)fac.size();
    std::stable_sort(fac.begin(), fac.end(), [](const Fac& x, const Fac& y) { return x.a != y.a ? x.a < y.a : x.b < y.b; });
    std::vector<int2> index((size_t)K * wdt, make_int2(0, 0));
    std::vector<int> fa(std::max(nf, 1)), fb(std::max(nf, 1)), ft(std::max(nf, 1)), fi(std::max(nf, 1));
    for (int q = 0; q < nf; ++q) {
        fa[q] = fac[q].a; fb[q] = fac[q].b; ft[q] = fac[q].type; fi[q] = fac[q].idx;
        int2& e = index[(size_t)fac[q].a * wdt + (fac[q].b - fac[q].a) + band];
        if (e.y == 0) e.x = q;
        e.y += 1;
    }
    const size_t need_fac = std::max((size_t)nf, (size_t)n_dq);
    if (need_fac > s->cap_fac) {
        void** p[] = {(void**)&s->d_fa, (void**)&s->d_fb, (void**)&s->d_ftype, (void**)&s->d_fidx, (void**)&s->d_dq_const, (void**)&s->d_frec};
        for (void** q : p) { if (*q) hipFree(*q); *q = nullptr; }
        s->cap_fac = 0; s->n_fac = 0;
        const size_t cap = need_fac + need_fac / 2 + 16;
        BT_CHECK(hipMalloc((void**)&s->d_fa, cap * 4)); BT_CHECK(hipMalloc((void**)&s->d_fb, cap * 4));
        BT_CHECK(hipMalloc((void**)&s->d_ftype, cap * 4)); BT_CHECK(hipMalloc((void**)&s->d_fidx, cap * 4));
        BT_CHECK(hipMalloc((void**)&s->d_dq_const, cap * 32)); BT_CHECK(hipMalloc((void**)&s->d_frec, cap * (SREC + 1) * 8));
        s->cap_fac = cap;             // only after every allocation succeeded
    }
    if ((size_t)s->n_rp > s->cap_rp) {
        if (s->d_rp_const) { hipFree(s->d_rp_const); s->d_rp_const = nullptr; }
        s->cap_rp = 0;
        const size_t cap = (size_t)s->n_rp + s->n_rp / 2 + 16;
        BT_CHECK(hipMalloc((void**)&s->d_rp_const, cap * 56));
        s->cap_rp = cap;
    }
    if (s->n_rp) BT_CHECK(hipMemcpy(s->d_rp_const, s->h_rp_c, (size_t)s->n_rp * 56, hipMemcpyHostToDevice));
    if ((size_t)n_dd > s->cap_dd) {
        if (s->d_dd) { hipFree(s->d_dd); s->d_dd = nullptr; }
        s->cap_dd = 0; s->n_dd = 0;
        const size_t cap = (size_t)n_dd + n_dd / 2 + 16;
        BT_CHECK(hipMalloc((void**)&s->d_dd, cap * sizeof(glio_dd_psr)));
        s->cap_dd = cap;
    }
    if (nf) {
        BT_CHECK(hipMemcpy(s->d_fa, fa.data(), (size_t)nf * 4, hipMemcpyHostToDevice)); BT_CHECK(hipMemcpy(s->d_fb, fb.data(), (size_t)nf * 4, hipMemcpyHostToDevice));
        BT_CHECK(hipMemcpy(s->d_ftype, ft.data(), (size_t)nf * 4, hipMemcpyHostToDevice)); BT_CHECK(hipMemcpy(s->d_fidx, fi.data(), (size_t)nf * 4, hipMemcpyHostToDevice));
    }
    if (n_dq) BT_CHECK(hipMemcpy(s->d_dq_const, dq_const, (size_t)n_dq * 32, hipMemcpyHostToDevice));
    if (n_dd) BT_CHECK(hipMemcpy(s->d_dd, dd, (size_t)n_dd * sizeof(glio_dd_psr), hipMemcpyHostToDevice));
    BT_CHECK(hipMemcpy(s->d_small_index, index.data(), index.size() * sizeof(int2), hipMemcpyHostToDevice));
    if (frame) {
        glio_host_ecef_local(frame->anc_ecef, frame->yaw_enu_local, s->R_ecef_local);
        for (int k = 0; k < 3; ++k) s->anc[k] = frame->anc_ecef[k];
        double rel[12];
        memcpy(rel, s->R_ecef_local, 72); memcpy(rel + 9, s->anc, 24);
        BT_CHECK(hipMemcpy(s->d_rel, rel, 96, hipMemcpyHostToDevice));
    }
    s->n_dq = n_dq; s->n_dd = n_dd; s->n_fac = nf;
    return GLIO_OK;
}

// LidarPoseFactorBatchRelativeAutoDiff factors between keyframes rp_i[f] and rp_j[f] (blocks P1 Q1 = keyframe rp_i, P2 Q2 = keyframe rp_j;
// rp_const [n][7] = delta_q (w,x,y,z), delta_p as the estimator computes them from the odometry poses, Estimator.cpp:2901-2923): the scan-to-multiscan
// factor of sms_fusion_level == 0, the released default (config_urban_hk.yaml:63).  Stored; the NEXT glio_batch_set_small_factors builds the factor
// table with them (call this first; n_rp = 0 removes them).
int glio_batch_set_relative_pose_factors(glio_batch* b, int n_rp, const int32_t* rp_i, const int32_t* rp_j, const double* rp_const) {
    if (!b || n_rp < 0 || (n_rp > 0 && (!rp_i || !rp_j || !rp_const))) return GLIO_E_ARG;
    BT_CHECK(hipSetDevice(b->device));
    { const int rc = small_ensure(b); if (rc) return rc; }
    BatchSmall* s = b->small;
    free(s->h_rp_i); free(s->h_rp_j); free(s->h_rp_c);
    s->h_rp_i = s->h_rp_j = nullptr; s->h_rp_c = nullptr; s->n_rp = 0;
    if (n_rp > 0) {
        s->h_rp_i = (int*)malloc((size_t)n_rp * 4); s->h_rp_j = (int*)malloc((size_t)n_rp * 4); s->h_rp_c = (double*)malloc((size_t)n_rp * 56);
        if (!s->h_rp_i || !s->h_rp_j || !s->h_rp_c) return GLIO_E_ARG;
        memcpy(s->h_rp_i, rp_i, (size_t)n_rp * 4); memcpy(s->h_rp_j, rp_j, (size_t)n_rp * 4); memcpy(s->h_rp_c, rp_const, (size_t)n_rp * 56);
        s->n_rp = n_rp;
    }
    return GLIO_OK;
}

__global__ void k_set_dd_threshold(glio_dd_psr* dd, const int n, const double thr) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < n) dd[f].threshold = thr;
}
// DDpsr_threshold of the next outer round (Estimator.cpp:2764-2767) for every DD factor already on the device
int glio_batch_set_dd_threshold(glio_batch* b, double threshold) {
    if (!b) return GLIO_E_ARG;
    BT_CHECK(hipSetDevice(b->device));
    BatchSmall* s = b->small;
    if (!s || s->n_dd == 0) return GLIO_OK;
    hipLaunchKernelGGL(k_set_dd_threshold, dim3((s->n_dd + 255) / 256), dim3(256), 0, b->stream, s->d_dd, s->n_dd, threshold);
    BT_CHECK(hipGetLastError());
    return GLIO_OK;
}

// adds the small factors evaluated at `poses` ([K][7], host) into a band buffer (the damped Gauss-Newton path; one rank)
int glio_batch_add_small_dev(glio_batch* b, const double* poses, double* Hg_dev) {
    if (!b || !poses || !Hg_dev) return GLIO_E_ARG;
    BT_CHECK(hipSetDevice(b->device));
    if (!b->small || b->small->n_fac == 0) return GLIO_OK;
    memcpy(b->h_poses, poses, (size_t)b->K * 7 * 8);
    BT_CHECK(hipMemcpyAsync(b->d_poses, b->h_poses, (size_t)b->K * 7 * 8, hipMemcpyHostToDevice, b->stream));
    enqueue_small(b, b->d_poses, Hg_dev);
    BT_CHECK(hipGetLastError());
    BT_CHECK(hipStreamSynchronize(b->stream));
    return GLIO_OK;
}

// One linearisation of the whole problem at (poses, speed_bias) through the solver's own path (shard, hook, assembly): the
// replicated diagonal [n], gradient [n] and the cost, n = B K (parity checks; `allreduce` as in glio_batch_solve_tr2).
int glio_batch_linearize_full(glio_batch* b, const double* poses, const double* speed_bias, glio_allreduce_fn allreduce, void* user, double* diag, double* grad,
                              double* cost) {
    if (!b || !poses) return GLIO_E_ARG;
    BT_CHECK(hipSetDevice(b->device));
    { const int rc = small_ensure(b); if (rc) return rc; }
    { const int rc = tr_ensure(b); if (rc) return rc; }
    if (g_batch_moments) { const int rm = glio_batch_moments_ensure(b); if (rm) return rm; }
    BatchSmall* s = b->small;
    if (s->n_imu > 0 && !speed_bias) return GLIO_E_ARG;
    const int K = b->K, n = s->B * K;
    BtStatus st; memset(&st, 0, sizeof st);
    st.cur = 1; st.first = 1;
    BT_CHECK(hipMemcpyAsync(s->d_st, &st, sizeof st, hipMemcpyHostToDevice, b->stream));
    BT_CHECK(hipMemcpyAsync(s->d_x[0], poses, (size_t)K * 7 * 8, hipMemcpyHostToDevice, b->stream));
    if (s->n_imu > 0) BT_CHECK(hipMemcpyAsync(s->d_s[0], speed_bias, (size_t)K * 9 * 8, hipMemcpyHostToDevice, b->stream));
    BT_CHECK(hipMemcpyAsync(s->d_x[1], s->d_x[0], (size_t)K * 7 * 8, hipMemcpyDeviceToDevice, b->stream));
    BT_CHECK(hipMemcpyAsync(s->d_s[1], s->d_s[0], (size_t)K * 9 * 8, hipMemcpyDeviceToDevice, b->stream));
    BtRun r; r.b = b; r.hook = allreduce; r.user = user;
    enqueue_tr_linearize(r, 1, true);
    BT_CHECK(hipGetLastError());
    std::vector<double> h((size_t)2 * n + 1);
    BT_CHECK(hipMemcpyAsync(h.data(), s->d_A[0], ((size_t)2 * n + 1) * 8, hipMemcpyDeviceToHost, b->stream));
    BT_CHECK(hipStreamSynchronize(b->stream));
    if (diag) memcpy(diag, h.data(), (size_t)n * 8);
    if (grad) memcpy(grad, h.data() + n, (size_t)n * 8);
    if (cost) *cost = h[(size_t)2 * n];
    return GLIO_OK;
}

// The trust-region solve.  `allreduce` (NULL for one rank) is called with a device buffer, its length in doubles, the HIP stream the
// buffer is produced and consumed on, and `user`; it must leave the SUM over the ranks in place, ORDERED ON THAT STREAM (ncclAllReduce on
// it; torch.distributed.all_reduce with that stream current): the host does not wait for it.  Five calls per trust-region group (file
// header), the same on every rank.  poses [K][7] in/out, speed_bias [K][9] in/out (with the IMU chain; else ignored, may be NULL).
int glio_batch_solve_tr2(glio_batch* b, double* poses, double* speed_bias, const glio_batch_tr_opts* o, glio_allreduce_fn allreduce, void* user, glio_summary* sum) {
    GLIO_TRACE("K8 + trust region glio_batch_solve_tr2");
    if (!b || !poses || !o || !sum) return GLIO_E_ARG;
    BT_CHECK(hipSetDevice(b->device));
    { const int rc = small_ensure(b); if (rc) return rc; }
    { const int rc = tr_ensure(b); if (rc) return rc; }
    if (g_batch_moments) { const int rm = glio_batch_moments_ensure(b); if (rm) return rm; }
    BatchSmall* s = b->small;
    if (s->n_imu > 0 && !speed_bias) { glio_set_error("the IMU chain is set: speed_bias [K][9] is needed"); return GLIO_E_ARG; }
    if (s->world > 1 && !allreduce) { glio_set_error("rank %d of %d needs the all-reduce hook", s->rank, s->world); return GLIO_E_ARG; }
    // ownership: a sharded stage assembles the rows of keyframes lo - band .. hi + band only, so a constraint whose SOURCE keyframe another rank
    // owns would be summed into rows this rank never hands to the all-reduce (silently wrong sums) -- refuse it instead
    if (s->world > 1 && b->src_max >= b->src_min && (b->src_min < s->lo || b->src_max >= s->hi)) {
        glio_set_error("rank %d of %d owns keyframes [%d, %d): constraint source keyframes %d..%d lie outside (shard the pairs with glio_batch_shard_range at THIS band)",
                       s->rank, s->world, s->lo, s->hi, b->src_min, b->src_max);
        return GLIO_E_ARG;
    }
    const int K = b->K;
    hipStream_t st = b->stream;
    memset(sum, 0, sizeof *sum);
    BtOpts bo;
    bo.max_iterations = o->max_iterations; bo.max_nonmono = o->use_nonmonotonic_steps ? o->max_consecutive_nonmonotonic_steps : 0;
    bo.jacobi = o->jacobi_scaling; bo.dogleg_type = o->dogleg_type;
    bo.min_radius = o->min_trust_region_radius; bo.min_rel = o->min_relative_decrease; bo.ftol = o->function_tolerance; bo.gtol = o->gradient_tolerance;
    bo.ptol = o->parameter_tolerance;
    BtStatus init; memset(&init, 0, sizeof init);
    init.cur = 1; init.first = 1; init.step_valid = 1;
    init.radius = o->initial_trust_region_radius; init.mu = 1e-8;
    s->solve_id = s->solve_id % 30000 + 1;
    init.solve_id = s->solve_id;
    const int id = s->solve_id;
    memcpy(s->h_x, poses, (size_t)K * 7 * 8);
    if (s->n_imu > 0) memcpy(s->h_x + (size_t)K * 7, speed_bias, (size_t)K * 9 * 8);
    BT_CHECK(hipMemcpyAsync(s->d_st, &init, sizeof init, hipMemcpyHostToDevice, st));
    BT_CHECK(hipMemcpyAsync(s->d_x[0], s->h_x, (size_t)K * 7 * 8, hipMemcpyHostToDevice, st));
    if (s->n_imu > 0) BT_CHECK(hipMemcpyAsync(s->d_s[0], s->h_x + (size_t)K * 7, (size_t)K * 9 * 8, hipMemcpyHostToDevice, st));
    BT_CHECK(hipMemcpyAsync(s->d_x[1], s->d_x[0], (size_t)K * 7 * 8, hipMemcpyDeviceToDevice, st));
    BT_CHECK(hipMemcpyAsync(s->d_s[1], s->d_s[0], (size_t)K * 9 * 8, hipMemcpyDeviceToDevice, st));
    BtRun r; r.b = b; r.hook = allreduce; r.user = user;
    enqueue_tr_linearize(r, 1, true);            // the starting point is "the candidate" of a first group that accepts it unconditionally
    // The loop lives on the device; the host only keeps the queue fed.  Up to `lead` groups are in flight: group g is enqueued as soon as the state machine of
    // group g - lead (its first kernel) has said "the solve goes on"; every kernel of a group enqueued behind the deciding one exits at once on the done flag
    // (and its collectives run on stale buffers: harmless, and every rank calls them alike).  The number of groups a rank enqueues is a function of the device's
    // decisions alone -- f + lead - 1 when group f decides to stop -- never of host timing: a group is enqueued whenever the progress word allows it, and the
    // finished flag only ends the loop when it does not (progress is written before finished, and read again after finished was seen), so the collective
    // sequences of the ranks match.  lead = 1 is the round-4 loop (wait for group g before enqueuing g + 1).
    const int lead = s->enqueue_lead > 0 ? s->enqueue_lead : 2;
    const int max_groups = o->max_iterations + 16 + 8 * 5 + lead;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = GLIO_OK;
    auto progress_allows = [&](int g) {          // group g - lead has reported "goes on" (groups 1 .. lead need nobody's word)
        if (g <= lead) return true;
        const int w = __atomic_load_n(&s->h_prog[0], __ATOMIC_ACQUIRE);      // GPU-written mapped words: every poll is a real load
        return (w >> 16) == id && (w & 0xffff) >= ((g - lead) & 0xffff);
    };
    for (int g = 1; g <= max_groups; ++g) {
        long long spins = 0;
        bool go = false;
        for (;;) {
            if (progress_allows(g)) { go = true; break; }
            if (__atomic_load_n(&s->h_prog[1], __ATOMIC_ACQUIRE) == id) { go = progress_allows(g); break; }
            if (((++spins) & 0x3f) == 0) std::this_thread::yield();        // a group lasts a millisecond or more: polling does not need the whole core
            if ((spins & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
                glio_set_error("batch solve: no progress for 120 s (group %d)", g);
                return GLIO_E_HIP;
            }
        }
        if (!go) break;
        enqueue_tr_group(r, bo);
        if (hipGetLastError() != hipSuccess) { glio_set_error("batch solve: launch failure in group %d", g); return GLIO_E_HIP; }
    }
    {   // the groups in flight end with the one that decided to stop
        long long spins = 0;
        while (__atomic_load_n(&s->h_prog[1], __ATOMIC_ACQUIRE) != id) {
            if (((++spins) & 0x3f) == 0) std::this_thread::yield();
            if ((spins & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) { glio_set_error("batch solve: no progress for 120 s"); return GLIO_E_HIP; }
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    BT_CHECK(hipMemcpyAsync(s->h_x, s->d_xmin, (size_t)K * 7 * 8, hipMemcpyDeviceToHost, st));
    if (s->n_imu > 0) BT_CHECK(hipMemcpyAsync(s->h_x + (size_t)K * 7, s->d_smin, (size_t)K * 9 * 8, hipMemcpyDeviceToHost, st));
    // the final status by a stream-ordered copy of the device's own record (the mapped copy h_res only tells the loop above when to stop:
    // host-mapped words are not a safe carrier for a payload, see SolverStatus::checksum in glio_device.h)
    BtStatus* h_fin = reinterpret_cast<BtStatus*>(s->h_x + (size_t)K * 16);
    BT_CHECK(hipMemcpyAsync(h_fin, s->d_st, sizeof(BtStatus), hipMemcpyDeviceToHost, st));
    BT_CHECK(hipStreamSynchronize(st));
    const BtStatus res = *h_fin;
    if (!res.done || res.solve_id != id) { glio_set_error("batch solve did not finish"); return GLIO_E_STATE; }
    memcpy(poses, s->h_x, (size_t)K * 7 * 8);
    if (s->n_imu > 0) memcpy(speed_bias, s->h_x + (size_t)K * 7, (size_t)K * 9 * 8);
    sum->iterations = res.iteration; sum->successful_steps = res.successful; sum->termination = res.termination;
    sum->initial_cost = res.initial_cost; sum->final_cost = res.user_min_cost; sum->final_radius = res.radius; sum->gradient_max_norm = res.grad_max;
    sum->n_lidar_residuals = (int32_t)std::min<int64_t>(b->n_con, 2147483647);
    if (res.termination == GLIO_TERM_FAILURE) { glio_set_error("batch trust-region solver failure (mu %g, iteration %d)", res.mu, res.iteration); rc = GLIO_E_NUMERIC; }
    return rc;
}
int glio_batch_solve_tr(glio_batch* b, double* poses, const glio_batch_tr_opts* o, glio_allreduce_fn allreduce, void* user, glio_summary* sum) {
    if (b && b->small && b->small->n_imu > 0) { glio_set_error("the IMU chain is set: use glio_batch_solve_tr2"); return GLIO_E_STATE; }
    return glio_batch_solve_tr2(b, poses, nullptr, o, allreduce, user, sum);
}
// trust-region groups the host keeps in flight (1 = the round-4 loop: wait for group g's decision before enqueuing g + 1; default 2)
int glio_batch_debug_set_enqueue_lead(glio_batch* b, int lead) {
    if (!b || lead < 0 || lead > 8) return GLIO_E_ARG;
    { const int rc = small_ensure(b); if (rc) return rc; }
    b->small->enqueue_lead = lead;
    return GLIO_OK;
}
// counters of the last solves (bench / tests): hook calls, doubles handed to the hook, trust-region groups enqueued, BCR levels; reset on read
int glio_batch_debug_counters(glio_batch* b, int64_t* out4) {
    if (!b || !b->small || !out4) return GLIO_E_ARG;
    BatchSmall* s = b->small;
    out4[0] = s->hook_calls; out4[1] = s->hook_doubles; out4[2] = s->groups; out4[3] = s->bcr ? glio_bcr_levels(s->bcr) : 0;
    s->hook_calls = s->hook_doubles = s->groups = 0;
    return GLIO_OK;
}

}  // extern "C"
