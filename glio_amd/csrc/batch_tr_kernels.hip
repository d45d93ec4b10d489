// batch_tr_kernels.hip -- the rest of the batch problem on keyframe poses and its trust-region solve, behind the C-ABI:
//   * the "small" factors that every rank evaluates for itself and adds AFTER the all-reduce of the LiDAR band buffer
//     (SURVEY 8e): delta_q_factor_auto attitude constraints (reference GLIO/src/Estimator.cpp:2831-2891,
//     GLIO/include/factors/LidarKeyframeFactor.h:283-303) and dd_psr_factor_20 per GNSS epoch between the bracketing
//     keyframes, identity weight, station position (:3197-3271, :1899-1911; dd_psr_factor.hpp:25-171) -- one wavefront per
//     factor, a 121-double record each, summed into the band by a gather in fixed order (no atomics);
//   * glio_batch_solve_tr: Ceres' trust-region loop as configured at Estimator.cpp:3275-3281 (DOGLEG, non-monotonic steps,
//     max_num_iter) -- Jacobi scaling, Cauchy point, Gauss-Newton step from the block-cyclic-reduction solver with Ceres'
//     mu D^2 regularisation, dogleg combination, model cost change, candidate poses: all on the device; the host keeps only
//     the scalar state machine and calls the caller's all-reduce hook between "linearise my shard" and everything else.
//     Deviation, stated: TRADITIONAL dogleg in place of SUBSPACE_DOGLEG (the same two-dimensional subspace {gradient,
//     Gauss-Newton}; Ceres minimises the model over the whole subspace, the dogleg path is a curve in it); the CPU
//     checker used by the tests restates exactly this.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "batch_device.h"

#define SREC 124           // Haa 36 | Hbb 36 | Hab 36 | ga 6 | gb 6 | cost 1 | pad
#define TRV_THREADS 256

struct BatchSmall {
    int n_dq, n_dd, n_fac;
    int* d_fa; int* d_fb; int* d_ftype; int* d_fidx;      // sorted by ordered pair (a, b)
    double* d_dq_const; glio_dd_psr* d_dd;
    int2* d_small_index;       // [K][2 band + 1] (first, count) of the factors with (a = k, b = k + o)
    double* d_frec;            // [n_fac][SREC]
    size_t cap_fac, cap_dd;
    double R_ecef_local[9], anc[3];
    // trust region
    double* d_vec;             // 12 vectors of 6 K
    double* d_Hs;              // scaled copy of the band buffer
    double* d_hg[2];           // linearisation of the current point / the candidate
    double* d_x[2];            // poses of the current point / the candidate
    double* d_red; double* h_red;   // reduction scratch (device, pinned host)
    double* d_rel;             // R_ecef_local (9) and the anchor (3)
    int have_scale;
};
void glio_host_ecef_local(const double anc[3], double yaw, double R[9]);     // capi.hip

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
__global__ __launch_bounds__(64) void k_small_eval(const int n_fac, const int* __restrict__ fa, const int* __restrict__ fb, const int* __restrict__ ftype,
                                                   const int* __restrict__ fidx, const double* __restrict__ poses, const double* __restrict__ dq_const,
                                                   const glio_dd_psr* __restrict__ dd, const double* __restrict__ Rel /* [9] R_ecef_local, [3] anchor */,
                                                   double* __restrict__ frec) {
    __shared__ double Ja[19 * 6], Jb[19 * 6], rr[19], raw[19], Jri[57], Jrj[57];
    const int f = blockIdx.x, lane = threadIdx.x;
    if (f >= n_fac) return;
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
        if (lane == 0) {
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
            for (int k = 0; k < 3; ++k) {
                rr[k] = 10000.0 * p[1 + k];
                double Jgi[4], Jgj[4];
                for (int c = 0; c < 4; ++c) {
                    double s = 0;
                    for (int m = 0; m < 4; ++m) s += M[(1 + k) * 4 + m] * (((m == c ? (m == 0 ? 1.0 : -1.0) : 0.0) - 2.0 * Cq[m] * qi[c] / n2) / n2);
                    Jgi[c] = 10000.0 * s;
                    Jgj[c] = 10000.0 * LAu[(1 + k) * 4 + c];
                }
                for (int c = 0; c < 3; ++c) {
                    Ja[k * 6 + 3 + c] = Jgi[0] * Pa[c] + Jgi[1] * Pa[3 + c] + Jgi[2] * Pa[6 + c] + Jgi[3] * Pa[9 + c];
                    Jb[k * 6 + 3 + c] = Jgj[0] * Pb[c] + Jgj[1] * Pb[3 + c] + Jgj[2] * Pb[6 + c] + Jgj[3] * Pb[9 + c];
                }
            }
        }
    } else {
        // dd_psr_factor_20::Evaluate (dd_psr_factor.hpp:25-171): one lane per satellite, then W r / W J by one lane per row
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
            const double r_ui = sqrt(d_dot3(d_ui, d_ui)), r_um = sqrt(d_dot3(d_um, d_um)), r_ri = sqrt(d_dot3(d_ri, d_ri)), r_rm = sqrt(d_dot3(d_rm, d_rm));
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
__global__ void k_small_add(const int K, const int band, const int2* __restrict__ sidx, const double* __restrict__ frec, const int n_fac, double* __restrict__ Hg) {
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
__global__ __launch_bounds__(256) void k_small_cost(const double* __restrict__ frec, const int n_fac, double* __restrict__ cost) {
    __shared__ double red[4];
    double s = 0;
    for (int q = threadIdx.x; q < n_fac; q += 256) s += frec[(size_t)n_fac * SREC + q];           // the dense copy of the records' cost entries (coalesced)
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *cost += (red[0] + red[1]) + (red[2] + red[3]);
}

// ------------------------------------------------------------------------------------------------ trust-region vector kernels
// vector slots (6 K doubles each)
enum { V_SC = 0, V_DG, V_GR, V_UU, V_TT, V_GS, V_GN, V_YY, V_ST, V_DL, V_DA, V_HS, V_COUNT };
#define BV(b, k) ((b)->small->d_vec + (size_t)(k) * 6 * (b)->K)

__device__ __forceinline__ double bt_diagH(const double* Hg, const int band, const int i) { return Hg[((size_t)(i / 6) * (band + 1)) * 36 + (i % 6) * 7]; }

// scale (first call), D, g_s, g~, u, mu D^2
__global__ void k_bt_prepare(const double* __restrict__ Hg, const int K, const int band, const int set_scale, const int jacobi, const double mu,
                             double* __restrict__ sc, double* __restrict__ dg, double* __restrict__ gr, double* __restrict__ uu, double* __restrict__ gs,
                             double* __restrict__ da) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, n = 6 * K;
    if (i >= n) return;
    const double h = bt_diagH(Hg, band, i);
    if (set_scale) sc[i] = jacobi ? 1.0 / (1.0 + sqrt(h)) : 1.0;
    const double s = sc[i];
    double d = s * s * h;
    d = d < 1e-6 ? 1e-6 : (d > 1e32 ? 1e32 : d);
    const double D = sqrt(d);
    const double g = Hg[(size_t)K * (band + 1) * 36 + i];
    dg[i] = D; gs[i] = s * g; gr[i] = s * g / D; uu[i] = (s * g / D) / D;
    da[i] = mu * D * D;
}
// Hs = S H S (band layout), rhs and cost copied
__global__ void k_bt_scale_band(const double* __restrict__ Hg, const int K, const int band, const double* __restrict__ sc, double* __restrict__ Hs) {
    const long long nH = (long long)K * (band + 1) * 36, tot = nH + 6LL * K + 1;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= tot) return;
    if (e < nH) {
        const int k = (int)(e / ((band + 1) * 36)), rem = (int)(e % ((band + 1) * 36)), d = rem / 36, r = (rem % 36) / 6, c = rem % 6;
        Hs[e] = (k + d < K) ? sc[6 * k + r] * Hg[e] * sc[6 * (k + d) + c] : 0.0;
    } else if (e < nH + 6LL * K) { const int i = (int)(e - nH); Hs[e] = sc[i] * Hg[e]; }
    else Hs[e] = Hg[e];
}
// y = Hband x (symmetric band, upper blocks stored): one thread per row
__global__ void k_bt_matvec(const double* __restrict__ Hb, const int K, const int band, const double* __restrict__ x, double* __restrict__ y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 6 * K) return;
    const int k = i / 6, r = i % 6, bw = band + 1;
    double s = 0;
    for (int d = 0; d <= band && k + d < K; ++d) {
        const double* blk = Hb + ((size_t)k * bw + d) * 36 + r * 6;
        const double* xv = x + 6 * (size_t)(k + d);
#pragma unroll
        for (int c = 0; c < 6; ++c) s += blk[c] * xv[c];
    }
    for (int d = 1; d <= band && k - d >= 0; ++d) {
        const double* blk = Hb + ((size_t)(k - d) * bw + d) * 36;          // H(k-d, k): transpose
        const double* xv = x + 6 * (size_t)(k - d);
#pragma unroll
        for (int c = 0; c < 6; ++c) s += blk[c * 6 + r] * xv[c];
    }
    y[i] = s;
}
// up to four dot products a_k . b_k in one pass: per-workgroup parts, then k_bt_sum adds them in order (deterministic)
struct DotArgs { const double* a[4]; const double* b[4]; int nd; };
__global__ __launch_bounds__(TRV_THREADS) void k_bt_dots(const DotArgs da, const int n, double* __restrict__ parts) {
    __shared__ double red[4][TRV_THREADS / 64];
    double s[4] = {0, 0, 0, 0};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < da.nd) s[k] += da.a[k][i] * da.b[k][i];
#pragma unroll
    for (int k = 0; k < 4; ++k) { s[k] = wave_sum(s[k]); if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = s[k]; }
    __syncthreads();
    if (threadIdx.x < 4) { double t = 0; for (int w = 0; w < TRV_THREADS / 64; ++w) t += red[threadIdx.x][w]; parts[(size_t)blockIdx.x * 4 + threadIdx.x] = t; }
}
__global__ void k_bt_sum(const double* __restrict__ parts, const int nb, double* __restrict__ out) {
    if (threadIdx.x < 4) { double t = 0; for (int b = 0; b < nb; ++b) t += parts[(size_t)b * 4 + threadIdx.x]; out[threadIdx.x] = t; }
}
// gn = D v with v = -(Hs + mu D^2)^-1 gs as the banded solver returns it; step = ca g~ + cb gn (D-space); step_s = step / D; delta = S step_s
__global__ void k_bt_gn(const double* __restrict__ dg, const double* __restrict__ v, double* __restrict__ gn, const int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) gn[i] = dg[i] * v[i];
}
__global__ void k_bt_step(const double ca, const double cb, const double* __restrict__ gr, const double* __restrict__ gn, const double* __restrict__ dg,
                          const double* __restrict__ sc, double* __restrict__ st, double* __restrict__ stD, double* __restrict__ dl, const int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double sv = ca * gr[i] + cb * gn[i];
    stD[i] = sv;                      // D-space step (its norm is the dogleg step norm)
    st[i] = sv / dg[i];
    dl[i] = sc[i] * (sv / dg[i]);
}
// candidate = x (+) delta;  parts: |x - cand|^2, |x|^2  and the gradient max norm needs | x - Plus(x, -g) |_inf (k_bt_gradmax)
__global__ void k_bt_plus(const double* __restrict__ x, const double* __restrict__ dl, const int K, double* __restrict__ xo) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    for (int c = 0; c < 3; ++c) xo[7 * k + c] = x[7 * k + c] + dl[6 * k + c];
    d_quat_plus(x + 7 * k + 3, dl + 6 * k + 3, xo + 7 * k + 3);
}
__global__ __launch_bounds__(TRV_THREADS) void k_bt_state_norms(const double* __restrict__ x, const double* __restrict__ xc, const double* __restrict__ g, const int K,
                                                                double* __restrict__ out /* [0] |x - xc|^2 [1] |x|^2 [2] max |x - Plus(x, -g)| */) {
    __shared__ double red[3][TRV_THREADS / 64];
    double d2 = 0, x2 = 0, gm = 0;
    for (int k = threadIdx.x; k < K; k += TRV_THREADS) {
        double nd[3], qn[4];
        for (int c = 0; c < 7; ++c) { const double a = x[7 * k + c], d = a - xc[7 * k + c]; d2 += d * d; x2 += a * a; }
        for (int c = 0; c < 3; ++c) { gm = fmax(gm, fabs(g[6 * k + c])); nd[c] = -g[6 * k + 3 + c]; }
        d_quat_plus(x + 7 * k + 3, nd, qn);
        for (int c = 0; c < 4; ++c) gm = fmax(gm, fabs(x[7 * k + 3 + c] - qn[c]));
    }
    d2 = wave_sum(d2); x2 = wave_sum(x2);
    for (int off = 32; off > 0; off >>= 1) gm = fmax(gm, __shfl_xor(gm, off, 64));
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = d2; red[1][threadIdx.x >> 6] = x2; red[2][threadIdx.x >> 6] = gm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, b = 0, c = 0;
        for (int w = 0; w < TRV_THREADS / 64; ++w) { a += red[0][w]; b += red[1][w]; c = fmax(c, red[2][w]); }
        out[0] = a; out[1] = b; out[2] = c;
    }
}

// ------------------------------------------------------------------------------------------------ host
#define BT_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { glio_set_error("%s failed: %s", #expr, hipGetErrorString(e_)); return GLIO_E_HIP; } } while (0)

static int small_ensure(glio_batch* b) {
    if (b->small) return GLIO_OK;
    BatchSmall* s = new BatchSmall();
    memset(s, 0, sizeof *s);
    const int K = b->K, band = b->band;
    const size_t hg = (size_t)glio_batch_hg_size(K, band);
    BT_CHECK(hipMalloc((void**)&s->d_small_index, (size_t)K * (2 * band + 1) * sizeof(int2)));
    BT_CHECK(hipMemset(s->d_small_index, 0, (size_t)K * (2 * band + 1) * sizeof(int2)));
    BT_CHECK(hipMalloc((void**)&s->d_vec, (size_t)V_COUNT * 6 * K * 8));
    BT_CHECK(hipMalloc((void**)&s->d_Hs, hg * 8));
    for (int k = 0; k < 2; ++k) { BT_CHECK(hipMalloc((void**)&s->d_hg[k], hg * 8)); BT_CHECK(hipMalloc((void**)&s->d_x[k], (size_t)K * 7 * 8)); }
    BT_CHECK(hipMalloc((void**)&s->d_red, (size_t)(4 * 256 + 16) * 8));
    BT_CHECK(hipMalloc((void**)&s->d_rel, 12 * 8));
    BT_CHECK(hipMemset(s->d_rel, 0, 12 * 8));
    BT_CHECK(hipHostMalloc((void**)&s->h_red, 16 * 8));
    b->small = s;
    return GLIO_OK;
}
void glio_batch_small_destroy(glio_batch* b) {
    BatchSmall* s = b->small;
    if (!s) return;
    void* p[] = {s->d_fa, s->d_fb, s->d_ftype, s->d_fidx, s->d_dq_const, s->d_dd, s->d_small_index, s->d_frec, s->d_vec, s->d_Hs, s->d_hg[0], s->d_hg[1],
                 s->d_x[0], s->d_x[1], s->d_red, s->d_rel};
    for (void* q : p) if (q) hipFree(q);
    if (s->h_red) hipHostFree(s->h_red);
    delete s;
    b->small = nullptr;
}

// evaluate the small factors at the poses in `poses_dev` and add them into Hg_dev (call AFTER the all-reduce)
static void enqueue_small(glio_batch* b, const double* poses_dev, double* Hg_dev) {
    BatchSmall* s = b->small;
    if (!s || s->n_fac == 0) return;
    const int K = b->K, band = b->band;
    hipLaunchKernelGGL(k_small_eval, dim3(s->n_fac), dim3(64), 0, b->stream, s->n_fac, s->d_fa, s->d_fb, s->d_ftype, s->d_fidx, poses_dev, s->d_dq_const, s->d_dd,
                       s->d_rel, s->d_frec);
    const long long tot = glio_batch_hg_size(K, band);
    hipLaunchKernelGGL(k_small_add, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, b->stream, K, band, s->d_small_index, s->d_frec, s->n_fac, Hg_dev);
    hipLaunchKernelGGL(k_small_cost, dim3(1), dim3(256), 0, b->stream, s->d_frec, s->n_fac, Hg_dev + tot - 1);
}

extern "C" {

int glio_batch_set_small_factors(glio_batch* b, const glio_gnss_frame* frame, int n_dq, const int32_t* dq_i, const int32_t* dq_j, const double* dq_const,
                                 int n_dd, const glio_dd_psr* dd) {
    if (!b || n_dq < 0 || n_dd < 0 || (n_dq > 0 && (!dq_i || !dq_j || !dq_const)) || (n_dd > 0 && (!dd || !frame))) return GLIO_E_ARG;
    BT_CHECK(hipSetDevice(b->device));
    { const int rc = small_ensure(b); if (rc) return rc; }
    BatchSmall* s = b->small;
    const int K = b->K, band = b->band, wdt = 2 * band + 1, nf = n_dq + n_dd;
    struct Fac { int a, b, type, idx; };
    std::vector<Fac> fac;
    fac.reserve(nf);
    for (int f = 0; f < n_dq; ++f) fac.push_back({dq_i[f], dq_j[f], 0, f});
    for (int f = 0; f < n_dd; ++f) fac.push_back({dd[f].slot_i, dd[f].slot_j, 1, f});
    for (const Fac& f : fac) {
        if (f.a < 0 || f.a >= K || f.b < 0 || f.b >= K || f.a == f.b || std::abs(f.a - f.b) > band) { glio_set_error("small factor on keyframes (%d, %d) outside band %d", f.a, f.b, band); return GLIO_E_ARG; }
        if (f.type == 1 && (dd[f.idx].n_sat < 2 || dd[f.idx].n_sat > GLIO_DD_MAX_SAT || dd[f.idx].master < 0 || dd[f.idx].master >= dd[f.idx].n_sat)) { glio_set_error("bad DD factor"); return GLIO_E_ARG; }
    }
    std::stable_sort(fac.begin(), fac.end(), [](const Fac& x, const Fac& y) { return x.a != y.a ? x.a < y.a : x.b < y.b; });
    std::vector<int2> index((size_t)K * wdt, make_int2(0, 0));
    std::vector<int> fa(std::max(nf, 1)), fb(std::max(nf, 1)), ft(std::max(nf, 1)), fi(std::max(nf, 1));
    for (int q = 0; q < nf; ++q) {
        fa[q] = fac[q].a; fb[q] = fac[q].b; ft[q] = fac[q].type; fi[q] = fac[q].idx;
        int2& e = index[(size_t)fac[q].a * wdt + (fac[q].b - fac[q].a) + band];
        if (e.y == 0) e.x = q;
        e.y += 1;
    }
    if ((size_t)nf > s->cap_fac) {
        void* p[] = {s->d_fa, s->d_fb, s->d_ftype, s->d_fidx, s->d_dq_const, s->d_frec};
        for (void* q : p) if (q) hipFree(q);
        s->cap_fac = (size_t)nf + nf / 2 + 16;
        BT_CHECK(hipMalloc((void**)&s->d_fa, s->cap_fac * 4)); BT_CHECK(hipMalloc((void**)&s->d_fb, s->cap_fac * 4));
        BT_CHECK(hipMalloc((void**)&s->d_ftype, s->cap_fac * 4)); BT_CHECK(hipMalloc((void**)&s->d_fidx, s->cap_fac * 4));
        BT_CHECK(hipMalloc((void**)&s->d_dq_const, s->cap_fac * 32)); BT_CHECK(hipMalloc((void**)&s->d_frec, s->cap_fac * (SREC + 1) * 8));
    }
    if ((size_t)n_dd > s->cap_dd) {
        if (s->d_dd) hipFree(s->d_dd);
        s->cap_dd = (size_t)n_dd + n_dd / 2 + 16;
        BT_CHECK(hipMalloc((void**)&s->d_dd, s->cap_dd * sizeof(glio_dd_psr)));
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

// adds the small factors evaluated at `poses` ([K][7], host) into the (reduced) buffer: the call that follows the all-reduce
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

// The trust-region solve.  `allreduce` (may be NULL for one rank) is called with the device buffer of this rank's LiDAR
// linearisation, its length in doubles, the HIP stream it was produced on and `user`; it must leave the SUM over the ranks in
// place (e.g. ncclAllReduce on that stream, or torch.distributed.all_reduce).  Every rank then adds the replicated small
// factors and takes identical decisions.
int glio_batch_solve_tr(glio_batch* b, double* poses, const glio_batch_tr_opts* o, void (*allreduce)(double*, int64_t, void*, void*), void* user,
                        glio_summary* sum) {
    if (!b || !poses || !o || !sum) return GLIO_E_ARG;
    if (!b->bcr) { glio_set_error("the trust-region batch solve needs the block-cyclic-reduction solver (band <= 12)"); return GLIO_E_ARG; }
    BT_CHECK(hipSetDevice(b->device));
    { const int rc = small_ensure(b); if (rc) return rc; }
    BatchSmall* s = b->small;
    const int K = b->K, band = b->band, n = 6 * K;
    const long long hg = glio_batch_hg_size(K, band);
    const int nbv = (n + TRV_THREADS - 1) / TRV_THREADS, nbd = std::min(nbv, 256);
    hipStream_t st = b->stream;
    memset(sum, 0, sizeof *sum);
    s->have_scale = 0;
    auto linearize = [&](int buf) -> int {        // poses in d_x[buf] -> d_hg[buf] (shard, reduce, small factors); returns through h_red[15] the cost
        BT_CHECK(hipMemcpyAsync(b->d_poses, s->d_x[buf], (size_t)K * 7 * 8, hipMemcpyDeviceToDevice, st));
        glio_batch_enqueue_linearize(b, s->d_hg[buf]);
        if (allreduce) { BT_CHECK(hipStreamSynchronize(st)); allreduce(s->d_hg[buf], (int64_t)hg, (void*)st, user); }
        enqueue_small(b, s->d_x[buf], s->d_hg[buf]);
        BT_CHECK(hipMemcpyAsync(s->h_red + 15, s->d_hg[buf] + hg - 1, 8, hipMemcpyDeviceToHost, st));
        BT_CHECK(hipStreamSynchronize(st));
        return GLIO_OK;
    };
    auto dots = [&](int nd, const double* a0, const double* b0, const double* a1, const double* b1, const double* a2, const double* b2, const double* a3,
                    const double* b3, double out[4]) -> int {
        DotArgs da; da.nd = nd;
        const double* A[4] = {a0, a1, a2, a3}; const double* B[4] = {b0, b1, b2, b3};
        for (int k = 0; k < 4; ++k) { da.a[k] = A[k] ? A[k] : a0; da.b[k] = B[k] ? B[k] : b0; }
        hipLaunchKernelGGL(k_bt_dots, dim3(nbd), dim3(TRV_THREADS), 0, st, da, n, s->d_red);
        hipLaunchKernelGGL(k_bt_sum, dim3(1), dim3(64), 0, st, s->d_red, nbd, s->d_red + 4 * 256);
        BT_CHECK(hipMemcpyAsync(s->h_red, s->d_red + 4 * 256, 4 * 8, hipMemcpyDeviceToHost, st));
        BT_CHECK(hipStreamSynchronize(st));
        for (int k = 0; k < 4; ++k) out[k] = s->h_red[k];
        return GLIO_OK;
    };
    int cur = 0;
    memcpy(b->h_poses, poses, (size_t)K * 7 * 8);
    BT_CHECK(hipMemcpyAsync(s->d_x[cur], b->h_poses, (size_t)K * 7 * 8, hipMemcpyHostToDevice, st));
    { const int rc = linearize(cur); if (rc) return rc; }
    double cost = s->h_red[15];
    sum->initial_cost = cost;
    double radius = o->initial_trust_region_radius, mu = 1e-8, alpha = 0, dogleg_step_norm = 0;
    int reuse = 0, iteration = 0, invalid = 0;
    double minimum_cost = cost, current_cost = cost, reference_cost = cost, candidate_cost = cost, acc_ref = 0, acc_cand = 0;
    int n_nonmono = 0;
    const int max_nonmono = o->use_nonmonotonic_steps ? o->max_consecutive_nonmonotonic_steps : 0;
    double gg = 0, nn2 = 0, gd = 0;         // |g~|^2, |gn|^2, g~.gn of the stored Gauss-Newton / Cauchy data
    sum->termination = GLIO_TERM_NO_CONVERGENCE;
    int rc = GLIO_OK;
    for (;;) {
        // gradient max norm and the loop-top checks
        const double* gcur = s->d_hg[cur] + (size_t)K * (band + 1) * 36;
        hipLaunchKernelGGL(k_bt_state_norms, dim3(1), dim3(TRV_THREADS), 0, st, s->d_x[cur], s->d_x[cur], gcur, K, s->d_red + 4 * 256 + 4);
        BT_CHECK(hipMemcpyAsync(s->h_red + 4, s->d_red + 4 * 256 + 4, 3 * 8, hipMemcpyDeviceToHost, st));
        BT_CHECK(hipStreamSynchronize(st));
        sum->gradient_max_norm = s->h_red[6];
        if (iteration >= o->max_iterations) { sum->termination = GLIO_TERM_NO_CONVERGENCE; break; }
        if (sum->gradient_max_norm <= o->gradient_tolerance) { sum->termination = GLIO_TERM_GRADIENT_TOL; break; }
        if (radius <= o->min_trust_region_radius) { sum->termination = GLIO_TERM_MIN_RADIUS; break; }
        ++iteration;
        bool step_valid = true;
        if (!reuse) {
            hipLaunchKernelGGL(k_bt_prepare, dim3(nbv), dim3(TRV_THREADS), 0, st, s->d_hg[cur], K, band, s->have_scale ? 0 : 1, o->jacobi_scaling, mu,
                               BV(b, V_SC), BV(b, V_DG), BV(b, V_GR), BV(b, V_UU), BV(b, V_GS), BV(b, V_DA));
            s->have_scale = 1;
            hipLaunchKernelGGL(k_bt_scale_band, dim3((unsigned)((hg + 255) / 256)), dim3(256), 0, st, s->d_hg[cur], K, band, BV(b, V_SC), s->d_Hs);
            // Cauchy: alpha = |g~|^2 / (w^T Hs w), w = g~ / D
            hipLaunchKernelGGL(k_bt_matvec, dim3(nbv), dim3(TRV_THREADS), 0, st, s->d_Hs, K, band, BV(b, V_UU), BV(b, V_TT));
            double d4[4];
            rc = dots(2, BV(b, V_GR), BV(b, V_GR), BV(b, V_UU), BV(b, V_TT), nullptr, nullptr, nullptr, nullptr, d4);
            if (rc) return rc;
            gg = d4[0];
            alpha = gg / d4[1];
            bool solved = false;
            while (mu < 1.0) {
                int* fail_dev = nullptr;
                glio_bcr_solve_shift(b->bcr, s->d_Hs, 0.0, BV(b, V_DA), BV(b, V_YY), &fail_dev, st);      // delta = -(Hs + mu D^2)^-1 gs
                int fail = 0;
                BT_CHECK(hipMemcpyAsync(s->h_red + 8, fail_dev, 4, hipMemcpyDeviceToHost, st));
                BT_CHECK(hipStreamSynchronize(st));
                memcpy(&fail, s->h_red + 8, 4);
                if (!fail) { solved = true; break; }
                mu *= 10.0;
                hipLaunchKernelGGL(k_bt_prepare, dim3(nbv), dim3(TRV_THREADS), 0, st, s->d_hg[cur], K, band, 0, o->jacobi_scaling, mu,
                                   BV(b, V_SC), BV(b, V_DG), BV(b, V_GR), BV(b, V_UU), BV(b, V_GS), BV(b, V_DA));
            }
            if (!solved) step_valid = false;
            else {
                hipLaunchKernelGGL(k_bt_gn, dim3(nbv), dim3(TRV_THREADS), 0, st, BV(b, V_DG), BV(b, V_YY), BV(b, V_GN), n);
                double d3[4];
                rc = dots(2, BV(b, V_GN), BV(b, V_GN), BV(b, V_GR), BV(b, V_GN), nullptr, nullptr, nullptr, nullptr, d3);
                if (rc) return rc;
                nn2 = d3[0]; gd = d3[1];
            }
        }
        double ca = 0, cb = 1, snorm = -1;
        if (step_valid) {
            const double gnorm = std::sqrt(gg), gnn = std::sqrt(nn2);
            if (gnn <= radius) { ca = 0; cb = 1; snorm = gnn; }
            else if (gnorm * alpha >= radius) { ca = -(radius / gnorm); cb = 0; snorm = radius; }
            else {
                const double b_dot_a = -alpha * gd, a_sq = alpha * alpha * gg;
                const double b_minus_a_sq = nn2 - 2 * b_dot_a + a_sq, c = b_dot_a - a_sq;
                const double d = std::sqrt(c * c + b_minus_a_sq * (radius * radius - a_sq));
                const double beta = (c <= 0) ? (d - c) / b_minus_a_sq : (radius * radius - a_sq) / (d + c);
                ca = -alpha * (1.0 - beta); cb = beta;
            }
            hipLaunchKernelGGL(k_bt_step, dim3(nbv), dim3(TRV_THREADS), 0, st, ca, cb, BV(b, V_GR), BV(b, V_GN), BV(b, V_DG), BV(b, V_SC),
                               BV(b, V_ST), BV(b, V_HS), BV(b, V_DL), n);
            // model cost change = -(gs . s + s^T Hs s / 2)
            hipLaunchKernelGGL(k_bt_matvec, dim3(nbv), dim3(TRV_THREADS), 0, st, s->d_Hs, K, band, BV(b, V_ST), BV(b, V_TT));
            double d4[4];
            rc = dots(3, BV(b, V_GS), BV(b, V_ST), BV(b, V_ST), BV(b, V_TT), BV(b, V_HS), BV(b, V_HS), nullptr, nullptr, d4);
            if (rc) return rc;
            const double mcc = -(d4[0] + 0.5 * d4[1]);
            if (snorm < 0) snorm = std::sqrt(d4[2]);
            dogleg_step_norm = snorm;
            if (!(mcc > 0.0)) step_valid = false;
            else {
                invalid = 0;
                hipLaunchKernelGGL(k_bt_plus, dim3((K + 255) / 256), dim3(256), 0, st, s->d_x[cur], BV(b, V_DL), K, s->d_x[1 - cur]);
                rc = linearize(1 - cur);
                if (rc) return rc;
                const double ccost = s->h_red[15];
                hipLaunchKernelGGL(k_bt_state_norms, dim3(1), dim3(TRV_THREADS), 0, st, s->d_x[cur], s->d_x[1 - cur], gcur, K, s->d_red + 4 * 256 + 4);
                BT_CHECK(hipMemcpyAsync(s->h_red + 4, s->d_red + 4 * 256 + 4, 3 * 8, hipMemcpyDeviceToHost, st));
                BT_CHECK(hipStreamSynchronize(st));
                const double step_norm = std::sqrt(s->h_red[4]), x_norm = std::sqrt(s->h_red[5]);
                if (step_norm <= o->parameter_tolerance * (x_norm + o->parameter_tolerance)) { sum->termination = GLIO_TERM_PARAMETER_TOL; break; }
                if (std::fabs(current_cost - ccost) <= o->function_tolerance * current_cost) { sum->termination = GLIO_TERM_FUNCTION_TOL; break; }
                const double rel = (current_cost - ccost) / mcc;
                const double hist = (reference_cost - ccost) / (acc_ref + mcc);
                const double quality = max_nonmono > 0 ? std::max(rel, hist) : rel;
                if (quality > o->min_relative_decrease) {
                    cur = 1 - cur;
                    sum->successful_steps += 1;
                    if (quality < 0.25) radius *= 0.5;
                    if (quality > 0.75) radius = std::max(radius, 3.0 * dogleg_step_norm);
                    mu = std::max(1e-8, 2.0 * mu / 10.0);
                    reuse = 0;
                    current_cost = ccost;
                    acc_cand += mcc; acc_ref += mcc;
                    if (current_cost < minimum_cost) { minimum_cost = current_cost; n_nonmono = 0; candidate_cost = current_cost; acc_cand = 0; }
                    else { ++n_nonmono; if (current_cost > candidate_cost) { candidate_cost = current_cost; acc_cand = 0; } }
                    if (n_nonmono == max_nonmono) { reference_cost = candidate_cost; acc_ref = acc_cand; }
                } else { radius *= 0.5; reuse = 1; }
                continue;
            }
        }
        // invalid step
        if (++invalid >= 5) { sum->termination = GLIO_TERM_FAILURE; break; }
        mu *= 10.0; reuse = 0;
    }
    sum->iterations = iteration;
    sum->final_cost = current_cost;
    sum->final_radius = radius;
    sum->n_lidar_residuals = (int32_t)std::min<int64_t>(b->n_con, 2147483647);
    BT_CHECK(hipMemcpyAsync(b->h_poses, s->d_x[cur], (size_t)K * 7 * 8, hipMemcpyDeviceToHost, st));
    BT_CHECK(hipStreamSynchronize(st));
    memcpy(poses, b->h_poses, (size_t)K * 7 * 8);
    if (sum->termination == GLIO_TERM_FAILURE) { glio_set_error("batch trust-region solver failure (mu %g, iteration %d)", mu, iteration); return GLIO_E_NUMERIC; }
    return GLIO_OK;
}

}  // extern "C"
