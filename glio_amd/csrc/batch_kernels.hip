// batch_kernels.hip -- K8: the scan-to-multiscan constraints of the batch stage
// (Estimator::optimizeBatchWithLandMark, reference GLIO/src/Estimator.cpp:3004-3076) on gfx950, plus the
// block-banded normal-equation assembly and a damped banded solve.
//
// Factor (BinaryLidarPlaneNormFactor::operator(), GLIO/include/factors/LidarKeyframeFactor.h:132-150, no
// extrinsic -- quirk Q10 -- and no loss function, Estimator.cpp:2768):
//   p_w = R1 p + t1 ; n_w = R2 n_l ; c_w = R2 c_l + t2 ; r = s n_w . (p_w - c_w)
// Local Jacobians under Ceres' left (+) on both quaternions (fused form of autodiff + QuaternionParameterization):
//   d r/d t1 = s n_w =: u      d r/d th1 = 2 s (R1 p) x n_w =: v
//   d r/d t2 = -u              d r/d th2 = 2 s n_w x (p_w - t2) =: w
// so one residual is described by the 9-vector j = [u v w]; all 12x12 blocks follow from the 9x9 Gram matrix.
//
// k_batch_pairs : ONE WAVEFRONT PER KEYFRAME PAIR (constraints are stored pair-major), 72 B streamed per
//                 residual (float4 point, 6 f64 plane normal + centroid, f64 score), 55 fp64 accumulators per
//                 lane (45 Gram + 9 J^T r + cost), reduced with a 63-shuffle value-splitting butterfly ->
//                 one 55-double record per pair.  HBM-bound streaming reduction, no atomics.
// k_batch_assemble : gathers the pair records into the block-banded H (upper band), g and the cost.
// The per-rank Hg buffers are summed by the caller with ONE RCCL all-reduce (torch.distributed on the same
// device buffer); k_batch_factor / k_batch_backsolve then run the replicated block-banded Cholesky.
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <vector>

#include "glio_device.h"

#include "batch_device.h"

__device__ __forceinline__ int gram_idx(int i, int j) {       // packed upper triangle of a symmetric 9x9
    const int a = i < j ? i : j, b = i < j ? j : i;
    return a * 9 - (a * (a - 1)) / 2 + (b - a);
}

// value-splitting butterfly over 64 values: lane L ends with the wave-wide sum of value
// k = b5 + 2 b4 + 4 b3 + 8 b2 + 16 b1 + 32 b0 (b_i = bit i of L); 63 shuffles
__device__ __forceinline__ double butterfly64(const double (&in)[64], int lane, int* k_out) {
    // (exchanges through V_PERMLANE32/16_SWAP and DPP instead of ds_bpermute: glio_device.h, "Cross-lane exchange"; same sums)
    double v32[32], v16[16], v8[8], v4[4], v2[2];
#pragma unroll
    for (int i = 0; i < 32; ++i) { double x, y; lane_swap32(in[2 * i], in[2 * i + 1], x, y); v32[i] = x + y; }
#pragma unroll
    for (int i = 0; i < 16; ++i) { double x, y; lane_swap16(v32[2 * i], v32[2 * i + 1], x, y); v16[i] = x + y; }
    { const bool hi = (lane & 8) != 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) { const double keep = hi ? v16[2 * i + 1] : v16[2 * i], send = hi ? v16[2 * i] : v16[2 * i + 1]; v8[i] = keep + lane_xor_row_d<8>(send); } }
    { const bool hi = (lane & 4) != 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) { const double keep = hi ? v8[2 * i + 1] : v8[2 * i], send = hi ? v8[2 * i] : v8[2 * i + 1]; v4[i] = keep + lane_xor_row_d<4>(send); } }
    { const bool hi = (lane & 2) != 0;
#pragma unroll
      for (int i = 0; i < 2; ++i) { const double keep = hi ? v4[2 * i + 1] : v4[2 * i], send = hi ? v4[2 * i] : v4[2 * i + 1]; v2[i] = keep + lane_xor_row_d<2>(send); } }
    const bool hi = (lane & 1) != 0;
    const double keep = hi ? v2[1] : v2[0], send = hi ? v2[0] : v2[1];
    *k_out = ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2) | (((lane >> 2) & 1) << 3) | (((lane >> 1) & 1) << 4) | ((lane & 1) << 5);
    return keep + lane_xor_row_d<1>(send);
}

__device__ __forceinline__ double uniform_d(const double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__global__ __launch_bounds__(256) void k_batch_pairs(const float4* __restrict__ cp, const double* __restrict__ nc,
                                                     const double* __restrict__ score, const int* __restrict__ pair_i,
                                                     const int* __restrict__ pair_j, const long long* __restrict__ pair_off,
                                                     const int n_pairs, const double* __restrict__ poses0, double* __restrict__ rec,
                                                     const BtSel sel, const double* __restrict__ poses1, double* __restrict__ cost_dense) {
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= n_pairs || bt_skip(sel)) return;
    const double* poses = bt_pick(sel) ? poses1 : poses0;
    const int a = pair_i[p], b = pair_j[p];
    double R1[9], R2[9], t1[3], t2[3];
    d_q2R(poses + 7 * (size_t)a + 3, R1);
    d_q2R(poses + 7 * (size_t)b + 3, R2);
#pragma unroll
    for (int k = 0; k < 3; ++k) { t1[k] = poses[7 * (size_t)a + k]; t2[k] = poses[7 * (size_t)b + k]; }
    // the two poses are the same for the whole wavefront: kept in scalar registers (24 doubles = 48 VGPRs less, which is what lets a
    // third wavefront per SIMD in: the kernel alternates between waiting for 72-byte records and ~160 fp64 multiply-adds on each)
#pragma unroll
    for (int k = 0; k < 9; ++k) { R1[k] = uniform_d(R1[k]); R2[k] = uniform_d(R2[k]); }
#pragma unroll
    for (int k = 0; k < 3; ++k) { t1[k] = uniform_d(t1[k]); t2[k] = uniform_d(t2[k]); }
    double acc[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) acc[k] = 0.0;
    const long long beg = pair_off[2 * p], end = pair_off[2 * p + 1];      // [begin, end) of the pair's records: pairs need not be adjacent in memory
    // software pipeline: the next constraint's 72 bytes are requested before the present one's ~160 multiply-adds are issued
    long long i = beg + lane;
    float4 pt_n = make_float4(0.f, 0.f, 0.f, 0.f);
    double2 n01_n = make_double2(0.0, 0.0), n2c0_n = n01_n, c12_n = n01_n;
    double s_n = 0.0;
    if (i < end) {
        pt_n = cp[i];
        const double2* ncp = reinterpret_cast<const double2*>(nc + 6 * i);
        n01_n = ncp[0]; n2c0_n = ncp[1]; c12_n = ncp[2];
        s_n = score[i];
    }
    for (; i < end; i += 64) {
        const float4 pt = pt_n;
        const double2 n01 = n01_n, n2c0 = n2c0_n, c12 = c12_n;
        const double s = s_n;
        {
            const long long in = i + 64 < end ? i + 64 : i;          // (the last iteration re-reads its own record: no branch in the loop body)
            pt_n = cp[in];
            const double2* ncp = reinterpret_cast<const double2*>(nc + 6 * in);
            n01_n = ncp[0]; n2c0_n = ncp[1]; c12_n = ncp[2];
            s_n = score[in];
        }
        const double px = pt.x, py = pt.y, pz = pt.z;
        const double rp[3] = {R1[0] * px + R1[1] * py + R1[2] * pz, R1[3] * px + R1[4] * py + R1[5] * pz, R1[6] * px + R1[7] * py + R1[8] * pz};
        const double nl[3] = {n01.x, n01.y, n2c0.x}, cl[3] = {n2c0.y, c12.x, c12.y};
        const double nw[3] = {R2[0] * nl[0] + R2[1] * nl[1] + R2[2] * nl[2], R2[3] * nl[0] + R2[4] * nl[1] + R2[5] * nl[2], R2[6] * nl[0] + R2[7] * nl[1] + R2[8] * nl[2]};
        const double rc[3] = {R2[0] * cl[0] + R2[1] * cl[1] + R2[2] * cl[2], R2[3] * cl[0] + R2[4] * cl[1] + R2[5] * cl[2], R2[6] * cl[0] + R2[7] * cl[1] + R2[8] * cl[2]};
        const double pw[3] = {rp[0] + t1[0], rp[1] + t1[1], rp[2] + t1[2]};
        const double av[3] = {pw[0] - rc[0] - t2[0], pw[1] - rc[1] - t2[1], pw[2] - rc[2] - t2[2]};
        const double r = s * (nw[0] * av[0] + nw[1] * av[1] + nw[2] * av[2]);
        const double q[3] = {pw[0] - t2[0], pw[1] - t2[1], pw[2] - t2[2]};
        const double s2 = 2.0 * s;
        double j[9];
        j[0] = s * nw[0]; j[1] = s * nw[1]; j[2] = s * nw[2];
        j[3] = s2 * (rp[1] * nw[2] - rp[2] * nw[1]); j[4] = s2 * (rp[2] * nw[0] - rp[0] * nw[2]); j[5] = s2 * (rp[0] * nw[1] - rp[1] * nw[0]);
        j[6] = s2 * (nw[1] * q[2] - nw[2] * q[1]); j[7] = s2 * (nw[2] * q[0] - nw[0] * q[2]); j[8] = s2 * (nw[0] * q[1] - nw[1] * q[0]);
        int k = 0;
#pragma unroll
        for (int u = 0; u < 9; ++u) {
#pragma unroll
            for (int v = u; v < 9; ++v) acc[k++] += j[u] * j[v];
            acc[BP_GRAM + u] += j[u] * r;
        }
        acc[54] += 0.5 * r * r;
    }
    int k;
    const double tot = butterfly64(acc, lane, &k);
    if (k < 55) rec[(size_t)p * BP_REC + k] = tot;
    if (k == 54) cost_dense[p] = tot;          // the costs once more, densely: what k_batch_cost sums
}

// ------------------------------------------------------------------------------------------------ K8 by MOMENTS (round 3)
// The residual of a binary plane constraint is LINEAR in theta = (M, tau) = (R_b^T R_a, R_b^T (t_a - t_b)):
//     r_i = s_i n_i^T (M p_i + tau) - s_i n_i^T c_i = phi_i^T theta - d_i,    phi_i = s_i [n_i (x) p_i (9) | n_i (3)],  d_i = s_i n_i . c_i
// (BinaryLidarPlaneNormFactor carries no loss function, LidarKeyframeFactor.h:124-164), so everything a pair contributes to the normal
// equations at ANY pose follows from twelve-dimensional moments of its constraints, which do not depend on the poses:
//     Phi = sum phi phi^T,   m = sum phi r0,   c0 = sum r0^2      with r0 = r(theta0) at the poses the moments were taken at;
//     at theta = theta0 + delta:   sum r^2 = c0 + delta^T (m + gamma),  gamma = m + Phi delta = sum phi r,
//     J_i = phi_i^T D  (D = d theta / d (t_a, rot_a, rot_b), 12 x 9, from the poses alone; d/d t_b = - d/d t_a)  =>
//     J^T J = D^T Phi D,   J^T r = D^T gamma        -- the very sums k_batch_pairs forms constraint by constraint.
// k_batch_moments streams the 72 bytes per constraint ONCE per constraint set (at the solve's first linearisation, centred there: the
// quadratic in delta has no cancellation while the poses stay near theta0); every later linearisation of the solve is k_batch_moment_eval,
// ~3000 multiply-adds per pair on 104 doubles instead of a pass over 4.7 GB (C4).  Same record layout as k_batch_pairs: the assembly does
// not know the difference.  Exact (not an approximation); the sums are associated differently, which shows at the 1e-13 level.
#define BM_REC 104          /* [0,78) Phi packed upper | [78,90) m | [90] c0 | [91,103) theta0 | pad */
__host__ __device__ __forceinline__ int bm_idx(const int a, const int b) { return a * 12 - (a * (a - 1)) / 2 + (b - a); }      // a <= b
__device__ __forceinline__ void bm_theta(const double R1[9], const double R2[9], const double t1[3], const double t2[3], double th[12]) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int k = 0; k < 3; ++k) th[3 * j + k] = R2[j] * R1[k] + R2[3 + j] * R1[3 + k] + R2[6 + j] * R1[6 + k];      // (R2^T R1)[j][k]
        th[9 + j] = R2[j] * (t1[0] - t2[0]) + R2[3 + j] * (t1[1] - t2[1]) + R2[6 + j] * (t1[2] - t2[2]);
    }
}
__global__ __launch_bounds__(256) void k_batch_moments(const float4* __restrict__ cp, const double* __restrict__ nc, const double* __restrict__ score,
                                                       const int* __restrict__ pair_i, const int* __restrict__ pair_j, const long long* __restrict__ pair_off,
                                                       const int n_pairs, const double* __restrict__ poses0, const BtSel sel, const double* __restrict__ poses1,
                                                       double* __restrict__ mom, const int* __restrict__ slots, const int n_slots) {
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= (slots ? n_slots : n_pairs) || bt_skip(sel)) return;
    const int p = slots ? slots[idx] : idx;          // (a list of pairs whose constraints were replaced, or all of them)
    const double* poses = bt_pick(sel) ? poses1 : poses0;
    const int a = pair_i[p], b = pair_j[p];
    double th0[12];
    {
        double R1[9], R2[9], t1[3], t2[3];
        d_q2R(poses + 7 * (size_t)a + 3, R1);
        d_q2R(poses + 7 * (size_t)b + 3, R2);
#pragma unroll
        for (int k = 0; k < 3; ++k) { t1[k] = poses[7 * (size_t)a + k]; t2[k] = poses[7 * (size_t)b + k]; }
        bm_theta(R1, R2, t1, t2, th0);
#pragma unroll
        for (int k = 0; k < 12; ++k) th0[k] = uniform_d(th0[k]);
    }
    double acc[128];
#pragma unroll
    for (int k = 0; k < 128; ++k) acc[k] = 0.0;
    const long long beg = pair_off[2 * p], end = pair_off[2 * p + 1];      // [begin, end) of the pair's records: pairs need not be adjacent in memory
    long long i = beg + lane;
    float4 pt_n = make_float4(0.f, 0.f, 0.f, 0.f);
    double2 n01_n = make_double2(0.0, 0.0), n2c0_n = n01_n, c12_n = n01_n;
    double s_n = 0.0;
    if (i < end) {
        pt_n = cp[i];
        const double2* ncp = reinterpret_cast<const double2*>(nc + 6 * i);
        n01_n = ncp[0]; n2c0_n = ncp[1]; c12_n = ncp[2];
        s_n = score[i];
    }
    for (; i < end; i += 64) {
        const float4 pt = pt_n;
        const double2 n01 = n01_n, n2c0 = n2c0_n, c12 = c12_n;
        const double s = s_n;
        {
            const long long in = i + 64 < end ? i + 64 : i;
            pt_n = cp[in];
            const double2* ncp = reinterpret_cast<const double2*>(nc + 6 * in);
            n01_n = ncp[0]; n2c0_n = ncp[1]; c12_n = ncp[2];
            s_n = score[in];
        }
        const double pv[3] = {(double)pt.x, (double)pt.y, (double)pt.z};
        const double sn[3] = {s * n01.x, s * n01.y, s * n2c0.x};
        double ph[12];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
#pragma unroll
            for (int k = 0; k < 3; ++k) ph[3 * j + k] = sn[j] * pv[k];
            ph[9 + j] = sn[j];
        }
        double r0 = -(sn[0] * n2c0.y + sn[1] * c12.x + sn[2] * c12.y);
#pragma unroll
        for (int k = 0; k < 12; ++k) r0 += ph[k] * th0[k];
        int q = 0;
#pragma unroll
        for (int u = 0; u < 12; ++u) {
#pragma unroll
            for (int v = u; v < 12; ++v) acc[q++] += ph[u] * ph[v];
        }
#pragma unroll
        for (int u = 0; u < 12; ++u) acc[78 + u] += ph[u] * r0;
        acc[90] += r0 * r0;
    }
    double lo[64], hi[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) { lo[k] = acc[k]; hi[k] = acc[64 + k]; }
    int k1, k2;
    const double t_lo = butterfly64(lo, lane, &k1);
    const double t_hi = butterfly64(hi, lane, &k2);
    double* out = mom + (size_t)p * BM_REC;
    out[k1] = t_lo;
    if (k2 < 27) out[64 + k2] = t_hi;
    double tv = 0.0;
#pragma unroll
    for (int k = 0; k < 12; ++k) tv = lane == k ? th0[k] : tv;
    if (lane < 12) out[91 + lane] = tv;
}
// the pair's record [9 x 9 Gram packed | J^T r (9) | cost] at the selected poses from its moments; one wavefront per pair
__global__ __launch_bounds__(256) void k_batch_moment_eval(const double* __restrict__ mom, const int* __restrict__ pair_i, const int* __restrict__ pair_j, const int n_pairs,
                                                           const double* __restrict__ poses0, double* __restrict__ rec, const BtSel sel, const double* __restrict__ poses1,
                                                           double* __restrict__ cost_dense) {
    __shared__ double sPhi[4][12 * 12 + 4], sD[4][12 * 9], sP[4][12 * 9], sV[4][48];      // sV: delta | m | gamma | (c0 ..)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int p = blockIdx.x * 4 + wv;
    if (p >= n_pairs || bt_skip(sel)) return;
    const double* poses = bt_pick(sel) ? poses1 : poses0;
    const int a = pair_i[p], b = pair_j[p];
    double R1[9], R2[9], t1[3], t2[3], th[12];
    d_q2R(poses + 7 * (size_t)a + 3, R1);
    d_q2R(poses + 7 * (size_t)b + 3, R2);
#pragma unroll
    for (int k = 0; k < 3; ++k) { t1[k] = poses[7 * (size_t)a + k]; t2[k] = poses[7 * (size_t)b + k]; }
    bm_theta(R1, R2, t1, t2, th);
    const double* M = mom + (size_t)p * BM_REC;
    double* Phi = sPhi[wv]; double* D = sD[wv]; double* P = sP[wv]; double* V = sV[wv];
    // Phi (full), m, delta
    for (int e = lane; e < 144; e += 64) { const int r = e / 12, c = e - 12 * r; Phi[e] = M[r <= c ? bm_idx(r, c) : bm_idx(c, r)]; }
    if (lane < 12) {
        double tv = 0.0;
#pragma unroll
        for (int k = 0; k < 12; ++k) tv = lane == k ? th[k] : tv;
        V[lane] = tv - M[91 + lane];
        V[12 + lane] = M[78 + lane];
    }
    // D = d theta / d (t_a | rot_a | rot_b), the local parameterisation of k_batch_pairs' Jacobian: J = [s n_w, 2 s (R1 p) x n_w, 2 s n_w x q]
    for (int e = lane; e < 108; e += 64) {
        const int ar = e / 9, u = e - 9 * ar;
        double v = 0.0;
        if (u < 3) { if (ar >= 9) v = R2[3 * u + (ar - 9)]; }
        else {
            const int k = u < 6 ? u - 3 : u - 6, l = (k + 1) % 3, m2 = (k + 2) % 3;
            if (ar < 9) {
                const int bj = ar / 3, aa = ar - 3 * bj;            // theta entry M[bj][aa] <-> phi = s n_bj p_aa
                if (u < 6) v = 2.0 * (R1[3 * l + aa] * R2[3 * m2 + bj] - R1[3 * m2 + aa] * R2[3 * l + bj]);
                else v = 2.0 * (R2[3 * l + bj] * R1[3 * m2 + aa] - R2[3 * m2 + bj] * R1[3 * l + aa]);
            } else if (u >= 6) {
                const int bj = ar - 9;
                v = 2.0 * (R2[3 * l + bj] * (t1[m2] - t2[m2]) - R2[3 * m2 + bj] * (t1[l] - t2[l]));
            }
        }
        D[e] = v;
    }
    GLIO_WAVE_LDS_SYNC();
    if (lane < 12) {
        double g = V[12 + lane];
#pragma unroll
        for (int k = 0; k < 12; ++k) g += Phi[lane * 12 + k] * V[k];
        V[24 + lane] = g;
    }
    for (int e = lane; e < 108; e += 64) {
        const int ar = e / 9, u = e - 9 * ar;
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 12; ++k) v += Phi[ar * 12 + k] * D[k * 9 + u];
        P[e] = v;
    }
    GLIO_WAVE_LDS_SYNC();
    double outv = 0.0;
    if (lane < 45) {
        int u = 0, rem = lane;
        while (rem >= 9 - u) { rem -= 9 - u; ++u; }
        const int v2 = u + rem;
#pragma unroll
        for (int k = 0; k < 12; ++k) outv += D[k * 9 + u] * P[k * 9 + v2];
    } else if (lane < 54) {
        const int u = lane - 45;
#pragma unroll
        for (int k = 0; k < 12; ++k) outv += D[k * 9 + u] * V[24 + k];
    } else if (lane == 54) {
        double q = M[90];
#pragma unroll
        for (int k = 0; k < 12; ++k) q += V[k] * (V[12 + k] + V[24 + k]);
        outv = 0.5 * q;
    }
    if (lane < 55) rec[(size_t)p * BP_REC + lane] = outv;
    if (lane == 54) cost_dense[p] = outv;
}

#define BB_MAX_BAND 16
// block-banded assembly: thread per entry of Hg = [H band K*(band+1)*36 | g K*6 | cost]
__global__ __launch_bounds__(256) void k_batch_assemble(const double* __restrict__ rec, const int* __restrict__ pair_index, const int K, const int band,
                                 const int n_pairs, double* __restrict__ Hg0, const BtSel sel, double* __restrict__ Hg1, const int k0, const int k1) {
    if (bt_skip(sel)) return;
    double* Hg = bt_pick(sel) ? Hg1 : Hg0;
    const long long nH = (long long)K * (band + 1) * 36, nG = (long long)K * 6;
    // rows [k0, k1) only: thread t -> entry t of the rows' H part, then of their g part
    const long long t_ = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nHr = (long long)(k1 - k0) * (band + 1) * 36, nGr = (long long)(k1 - k0) * 6;
    if (t_ >= nHr + nGr) return;
    const long long e = t_ < nHr ? (long long)k0 * (band + 1) * 36 + t_ : nH + (long long)k0 * 6 + (t_ - nHr);
    const int A[6] = {0, 1, 2, 3, 4, 5}, B[6] = {0, 1, 2, 6, 7, 8};
    const double sB[6] = {-1, -1, -1, 1, 1, 1};
    const int wdt = 2 * band + 1;
    // sum over the neighbours o = -band .. band (o != 0) of rec[pair (k, k+o)][ia] + sgn rec[pair (k+o, k)][ib], in that order.  The pair indices
    // and then the record entries are fetched as two batches of independent loads (a loop of "index -> entry -> add" rounds paid two memory
    // latencies per neighbour: 26 us for this kernel at C4)
    auto sum_pairs = [&](const int k, const int ia, const int ib, const double sgn) {
        int pa[2 * BB_MAX_BAND], pb[2 * BB_MAX_BAND];
#pragma unroll
        for (int q = 0; q < 2 * BB_MAX_BAND; ++q) {
            const int o = q < BB_MAX_BAND ? q - BB_MAX_BAND : q - BB_MAX_BAND + 1;
            const bool ok = o >= -band && o <= band && k + o >= 0 && k + o < K;
            pa[q] = ok ? pair_index[(size_t)k * wdt + o + band] : -1;            // a = k, b = k+o
            pb[q] = ok ? pair_index[(size_t)(k + o) * wdt + (-o) + band] : -1;   // a = k+o, b = k
        }
        double va[2 * BB_MAX_BAND], vb[2 * BB_MAX_BAND];
#pragma unroll
        for (int q = 0; q < 2 * BB_MAX_BAND; ++q) {
            va[q] = pa[q] >= 0 ? rec[(size_t)pa[q] * BP_REC + ia] : 0.0;
            vb[q] = pb[q] >= 0 ? rec[(size_t)pb[q] * BP_REC + ib] : 0.0;
        }
        double s = 0;
#pragma unroll
        for (int q = 0; q < 2 * BB_MAX_BAND; ++q) {
            if (pa[q] >= 0) s += va[q];
            if (pb[q] >= 0) s += sgn * vb[q];
        }
        return s;
    };
    if (e < nH) {
        const int k = (int)(e / ((band + 1) * 36)), rem = (int)(e % ((band + 1) * 36)), d = rem / 36, r = (rem % 36) / 6, c = rem % 6;
        double s = 0;
        if (d == 0) {
            s = sum_pairs(k, gram_idx(A[r], A[c]), gram_idx(B[r], B[c]), sB[r] * sB[c]);      // Ja^T Ja of the pairs (k, .), Jb^T Jb of the pairs (., k)
        } else if (k + d < K) {
            const int pa = pair_index[(size_t)k * wdt + d + band];                // a = k, b = k+d : H(k,k+d) = Ja^T Jb
            if (pa >= 0) s += sB[c] * rec[(size_t)pa * BP_REC + gram_idx(A[r], B[c])];
            const int pb = pair_index[(size_t)(k + d) * wdt + (-d) + band];       // a = k+d, b = k : H(k,k+d) = Jb^T Ja
            if (pb >= 0) s += sB[r] * rec[(size_t)pb * BP_REC + gram_idx(B[r], A[c])];
        }
        Hg[e] = s;
    } else if (e < nH + nG) {
        const int k = (int)((e - nH) / 6), r = (int)((e - nH) % 6);
        Hg[e] = sum_pairs(k, BP_GRAM + A[r], BP_GRAM + B[r], sB[r]);
    }
}
__global__ __launch_bounds__(1024) void k_batch_cost(const double* __restrict__ cost_dense, int n_pairs, double* out0, const BtSel sel, double* out1) {
    __shared__ double red[16];
    if (bt_skip(sel)) return;
    double* out = bt_pick(sel) ? out1 : out0;
    double s = 0;
    for (int p = threadIdx.x; p < n_pairs; p += 1024) s += cost_dense[p];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0; for (int w = 0; w < 16; ++w) t += red[w]; *out = t; }
}

// ------------------------------------------------------------------------------------------------
// damped block-banded Cholesky (one workgroup), rhs carried; M[k][d] = block (k+d, k), row-major 6x6
// ------------------------------------------------------------------------------------------------
// Block-banded Cholesky of H + lambda diag(H) (K block rows of 6, half band `band` blocks) by ONE wavefront, no workgroup
// barriers.  Block column k = the (band+1) blocks A(k+d, k), d = 0..band, plus the right-hand side as one more row:
// 6 (band+1) + 1 rows of 6 -- one row per lane (two row slots per lane when band > 9).  A ring of band+1 block columns
// lives in LDS.  Step k: the panel is factored in six register steps (pivot and multipliers by v_readlane, as in the
// window solver), written back (LDS for the update, global for the back substitution), the rank-6 update is applied to the
// band+... following columns of the ring (entries strided over the lanes, operands read from LDS), and block column
// k+band+1, whose global loads were issued at the top of the step, takes the freed ring slot.
// M out: [K][band+1][36], block (k, d) = L(k+d, k) row-major;  y out: L^-1 g.
#define BB_ROWS(band) (6 * ((band) + 1) + 1)
__host__ __device__ __forceinline__ size_t batch_factor_lds_doubles(int band) {
    size_t items = 0;
    for (int j = 1; j <= band; ++j) items += (size_t)(6 * (band - j + 1) + 1) * 6;
    return (size_t)(band + 2) * BB_ROWS(band) * 6 + 8 + items + 4;              // ring (band + 2 slots) + the packed (int2) item table
}

__device__ __forceinline__ void bb_load_column(const double* __restrict__ Hg, const long long nH, const int K, const int band, const double lambda,
                                               const int k, const int lane, double (&r0)[6], double (&r1)[6]) {
    // row rho of block column k: rho = 6 d + r -> A(k+d, k)[r][c] = H(k, k+d)[c][r]; the last row is the right-hand side g_k
    const int bw = band + 1, R = 6 * bw;
#pragma unroll
    for (int slot = 0; slot < 2; ++slot) {
        const int rho = lane + 64 * slot;
        double (&dst)[6] = slot == 0 ? r0 : r1;
#pragma unroll
        for (int c = 0; c < 6; ++c) dst[c] = 0.0;
        if (k >= K || rho > R) continue;
        if (rho == R) {
#pragma unroll
            for (int c = 0; c < 6; ++c) dst[c] = Hg[nH + (size_t)k * 6 + c];
        } else {
            const int d = rho / 6, r = rho - 6 * d;
            if (k + d < K) {
                const double* blk = Hg + ((size_t)k * bw + d) * 36;
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    double v = blk[c * 6 + r];
                    if (d == 0 && r == c) v += lambda * v + 1e-12;
                    dst[c] = (d == 0 && c > r) ? 0.0 : v;
                }
            }
        }
    }
}

#define BBF_THREADS 256
// One workgroup, four wavefronts with fixed roles per step k (ring of band+2 block-column slots in LDS, so the column that
// enters the band never shares a slot with one still in use):
//   wavefront 0   factors the panel [A(k..k+band, k); rhs_k] held one row per lane (six register steps, v_readlane pivots)
//   -- barrier --
//   all           rank-6 update of the following columns / right-hand sides (tabulated item list, four items per lane at a
//                 time: dot products first, read-modify-writes after)
//   wavefront 1   also streams the factored panel to global memory (M, y) for the back substitution
//   wavefront 3   also brings block column k+band+1 from global memory into the free slot (its loads were issued before
//                 the first barrier, i.e. they overlap the factorisation)
//   -- barrier --
template <bool TWO>
__global__ __launch_bounds__(BBF_THREADS) void k_batch_factor(const double* __restrict__ Hg, const int K, const int band, const double lambda,
                                                              double* M, double* y, int* fail) {
    extern __shared__ double bb_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, bw = band + 1, ns = bw + 1, R = 6 * bw, rows = R + 1;
    const long long nH = (long long)K * bw * 36;
    double* ring = bb_lds;                                   // [ns][rows][6]
    if (tid == 0) *fail = 0;
    for (int k = wv; k < bw; k += BBF_THREADS / 64) {        // preload block columns 0 .. bw-1
        double r0[6], r1[6];
        bb_load_column(Hg, nH, K, band, lambda, k, lane, r0, r1);
        double* col = ring + (size_t)(k % ns) * rows * 6;
        if (lane < rows) { for (int c = 0; c < 6; ++c) col[lane * 6 + c] = r0[c]; }
        if (TWO && lane + 64 < rows) { for (int c = 0; c < 6; ++c) col[(lane + 64) * 6 + c] = r1[c]; }
    }
    // item table of the trailing update for a full band, packed: offset of panel row 6 i + r (or the rhs row) | offset of panel
    // row 6 j + c | target row * 6 + c (10 bits each); j | max(i, j) << 8
    int2* desc = reinterpret_cast<int2*>(ring + (size_t)ns * rows * 6 + 2);
    int n_items = 0;
    for (int j = 1; j <= band; ++j) n_items += (6 * (band - j + 1) + 1) * 6;
    {
        int base = 0;
        for (int j = 1; j <= band; ++j) {
            const int nrow = 6 * (band - j + 1) + 1;
            for (int e = tid; e < nrow * 6; e += BBF_THREADS) {
                const int rr = e / 6, c = e - 6 * rr;
                const bool is_rhs = rr == nrow - 1;
                int w = is_rhs ? j : j + rr / 6;
                if (!is_rhs && rr < 6 && c > rr) w = 255;                 // upper part of a diagonal block: never valid
                int2 ds;                                                  // offsets < 1024: 10 bits each
                ds.x = ((is_rhs ? R : 6 * j + rr) * 6) | (((6 * j + c) * 6) << 10) | ((((is_rhs ? R : rr) * 6 + c)) << 20);
                ds.y = j | (w << 8);
                desc[base + e] = ds;
            }
            base += nrow * 6;
        }
    }
    __syncthreads();
    for (int k = 0; k < K; ++k) {
        double* col = ring + (size_t)(k % ns) * rows * 6;
        const int nd = min(band, K - 1 - k);
        double n0[6], n1[6];
        if (wv == 3) bb_load_column(Hg, nH, K, band, lambda, k + bw, lane, n0, n1);        // consumed after the barrier
        if (wv == 0) {
            double a0[6], a1[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) { a0[c] = lane < rows ? col[lane * 6 + c] : 0.0; a1[c] = (TWO && lane + 64 < rows) ? col[(lane + 64) * 6 + c] : 0.0; }
            bool bad = false;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                double djj = readlane_d(a0[j], j);
                if (!(djj > 0.0) || !isfinite(djj)) { bad = true; djj = 1.0; }
                const double rd = rsqrt(djj);
                const double l0 = (lane == j) ? djj * rd : a0[j] * rd;
                a0[j] = l0;
                double l1 = 0.0;
                if (TWO) { l1 = a1[j] * rd; a1[j] = l1; }
#pragma unroll
                for (int c = j + 1; c < 6; ++c) { const double m = readlane_d(l0, c); a0[c] -= l0 * m; if (TWO) a1[c] -= l1 * m; }
            }
            if (bad && lane == 0 && *fail == 0) *fail = 1 + k;
            if (lane < rows) {
#pragma unroll
                for (int c = 0; c < 6; ++c) col[lane * 6 + c] = (lane < 6 && c > lane) ? 0.0 : a0[c];
            }
            if (TWO && lane + 64 < rows) {
#pragma unroll
                for (int c = 0; c < 6; ++c) col[(lane + 64) * 6 + c] = a1[c];
            }
        }
        __syncthreads();
        if (wv == 1) {                           // the factored panel -> global: L blocks of block column k, and y_k
            const int nL = 36 * (nd + 1);
            for (int e = lane; e < nL; e += 64) M[(size_t)k * bw * 36 + e] = col[e];
            if (lane < 6) y[(size_t)k * 6 + lane] = col[R * 6 + lane];
        }
        for (int e0 = tid; e0 < n_items; e0 += BBF_THREADS * 4) {
            double sacc[4];
            int toff[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + BBF_THREADS * u;
                sacc[u] = 0.0; toff[u] = -1;
                if (e < n_items) {
                    const int2 ds = desc[e];
                    if ((ds.y >> 8) <= nd) {                 // max(i, j): the block row exists for this (shorter) last band
                        const double* xi = col + (ds.x & 1023);
                        const double* xj = col + ((ds.x >> 10) & 1023);
                        double v = 0.0;
#pragma unroll
                        for (int q = 0; q < 6; ++q) v += xi[q] * xj[q];
                        sacc[u] = v;
                        toff[u] = ((k + (ds.y & 255)) % ns) * rows * 6 + ((ds.x >> 20) & 1023);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (toff[u] >= 0) ring[toff[u]] -= sacc[u];
        }
        if (wv == 3) {                           // block column k+bw enters the band through the free slot
            double* dst = ring + (size_t)((k + bw) % ns) * rows * 6;
            if (lane < rows) { for (int c = 0; c < 6; ++c) dst[lane * 6 + c] = n0[c]; }
            if (TWO && lane + 64 < rows) { for (int c = 0; c < 6; ++c) dst[(lane + 64) * 6 + c] = n1[c]; }
        }
        __syncthreads();
    }
}

// back substitution L^T x = y by one wavefront; delta = -x.  The L column of the next step is prefetched while the current
// one is consumed; the last `band` solutions stay in LDS.
__global__ __launch_bounds__(64) void k_batch_backsolve(const double* __restrict__ M, const double* __restrict__ y, const int K, const int band,
                                                        double* delta) {
    __shared__ double xr[(BB_MAX_BAND + 1) * 6];
    __shared__ double colL[(BB_MAX_BAND + 1) * 36];
    const int lane = threadIdx.x, bw = band + 1, n36 = bw * 36;
    if (band <= 8) {
        // register path: lane c < 6 owns column c of every block of block column k -- L(k+d,k)[r][c], d = 0..band -- and
        // fetches the values of step k-1 while step k computes: no staging, nothing but 6 lanes' own loads
        const int c = lane < 6 ? lane : 0;
        double cur[9 * 6], nxt[9 * 6];
        double ycur = 0.0, ynxt = 0.0;
#pragma unroll
        for (int e = 0; e < 54; ++e) { const int d = e / 6, r = e % 6; nxt[e] = d <= band ? M[(size_t)(K - 1) * n36 + d * 36 + r * 6 + c] : 0.0; }
        ynxt = y[(size_t)(K - 1) * 6 + c];
        for (int k = K - 1; k >= 0; --k) {
            const int nd = min(band, K - 1 - k);
#pragma unroll
            for (int e = 0; e < 54; ++e) cur[e] = nxt[e];
            ycur = ynxt;
            if (k > 0) {
#pragma unroll
                for (int e = 0; e < 54; ++e) { const int d = e / 6, r = e % 6; nxt[e] = d <= band ? M[(size_t)(k - 1) * n36 + d * 36 + r * 6 + c] : 0.0; }
                ynxt = y[(size_t)(k - 1) * 6 + c];
            }
            double v = ycur;
#pragma unroll
            for (int d = 1; d <= 8; ++d) {
                if (d <= nd) {
                    const double* xd = xr + ((k + d) % bw) * 6;
                    double s0 = 0, s1 = 0;
#pragma unroll
                    for (int r = 0; r < 6; r += 2) { s0 += cur[d * 6 + r] * xd[r]; s1 += cur[d * 6 + r + 1] * xd[r + 1]; }
                    v -= s0 + s1;
                }
            }
            // L_kk^T x = v: lane c holds x_c; cur[q] = L_kk[q][c]
            double dg = 1.0;
#pragma unroll
            for (int q = 0; q < 6; ++q) if (lane == q) dg = cur[q];
            const double rdiag = 1.0 / dg;
#pragma unroll
            for (int q = 5; q >= 0; --q) {
                const double zq = readlane_d(v, q) * readlane_d(rdiag, q);
                if (lane == q) v = zq;
                else if (lane < q) v -= cur[q] * zq;
            }
            if (lane < 6) { xr[(k % bw) * 6 + lane] = v; delta[(size_t)k * 6 + lane] = -v; }
            GLIO_WAVE_LDS_SYNC();
        }
        return;
    }
    double pf[((BB_MAX_BAND + 1) * 36 + 63) / 64];
    const int npf = (n36 + 63) / 64;
    for (int q = 0; q < npf; ++q) { const int e = lane + 64 * q; pf[q] = (e < n36) ? M[(size_t)(K - 1) * n36 + e] : 0.0; }
    double ypf = lane < 6 ? y[(size_t)(K - 1) * 6 + lane] : 0.0;
    for (int k = K - 1; k >= 0; --k) {
        const int nd = min(band, K - 1 - k);
        for (int q = 0; q < npf; ++q) { const int e = lane + 64 * q; if (e < n36) colL[e] = pf[q]; }
        if (k > 0) { for (int q = 0; q < npf; ++q) { const int e = lane + 64 * q; pf[q] = (e < n36) ? M[(size_t)(k - 1) * n36 + e] : 0.0; } }
        const double yk = ypf;
        if (k > 0) ypf = lane < 6 ? y[(size_t)(k - 1) * 6 + lane] : 0.0;
        GLIO_WAVE_LDS_SYNC();
        double v = yk;
        if (lane < 6) {
            for (int d = 1; d <= nd; ++d) {
                const double* xd = xr + ((k + d) % bw) * 6;
                const double* blk = colL + d * 36;
#pragma unroll
                for (int r = 0; r < 6; ++r) v -= blk[r * 6 + lane] * xd[r];
            }
        }
        const double lc[6] = {colL[0 * 6 + (lane < 6 ? lane : 0)], colL[1 * 6 + (lane < 6 ? lane : 0)], colL[2 * 6 + (lane < 6 ? lane : 0)],
                              colL[3 * 6 + (lane < 6 ? lane : 0)], colL[4 * 6 + (lane < 6 ? lane : 0)], colL[5 * 6 + (lane < 6 ? lane : 0)]};
        const double rdg = lane < 6 ? 1.0 / colL[lane * 7] : 1.0;
#pragma unroll
        for (int q = 5; q >= 0; --q) {               // L_kk^T x = v: lane c holds x_c; column c of L_kk below the diagonal = lc[q], q > c
            const double zq = readlane_d(v, q) * readlane_d(rdg, q);
            if (lane == q) v = zq;
            else if (lane < q) v -= lc[q] * zq;
        }
        if (lane < 6) { xr[(k % bw) * 6 + lane] = v; delta[(size_t)k * 6 + lane] = -v; }
        GLIO_WAVE_LDS_SYNC();
    }
}

// poses (+) delta, and the model decrease -(g.d + d^T H d / 2) from the band
__global__ __launch_bounds__(256) void k_batch_apply(const double* __restrict__ Hg, const double* __restrict__ delta, const double* __restrict__ poses,
                                                     const int K, const int band, double* newposes, double* model_dec) {
    __shared__ double red[4];
    const int bw = band + 1;
    const double* g = Hg + (size_t)K * bw * 36;
    double acc = 0;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < K; k += gridDim.x * blockDim.x) {
        const double* dk = delta + (size_t)k * 6;
        double Hd[6] = {0, 0, 0, 0, 0, 0};
        for (int d = 0; d <= band && k + d < K; ++d) {
            const double* blk = Hg + ((size_t)k * bw + d) * 36;
            const double* dd = delta + (size_t)(k + d) * 6;
            for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) Hd[r] += blk[r * 6 + c] * dd[c];
        }
        for (int d = 1; d <= band && k - d >= 0; ++d) {
            const double* blk = Hg + ((size_t)(k - d) * bw + d) * 36;     // H(k-d, k); need its transpose
            const double* dd = delta + (size_t)(k - d) * 6;
            for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) Hd[r] += blk[c * 6 + r] * dd[c];
        }
        for (int r = 0; r < 6; ++r) acc += dk[r] * (g[(size_t)k * 6 + r] + 0.5 * Hd[r]);
        const double* p = poses + (size_t)k * 7;
        double* o = newposes + (size_t)k * 7;
        for (int c = 0; c < 3; ++c) o[c] = p[c] + dk[c];
        d_quat_plus(p + 3, dk + 3, o + 3);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) model_dec[blockIdx.x] = -(red[0] + red[1] + red[2] + red[3]);       // per-workgroup part; k_batch_apply_sum adds them in order
}
__global__ void k_batch_apply_sum(const double* __restrict__ parts, const int n, double* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0;
    for (int k = 0; k < n; ++k) s += parts[k];
    *out = s;
}

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
#define BALLOC(ptr, bytes) GLIO_HIP_CHECK(hipMalloc((void**)&(ptr), (size_t)(bytes) > 0 ? (size_t)(bytes) : 16))

extern "C" {

int64_t glio_batch_hg_size(int K, int band) { return (int64_t)K * (band + 1) * 36 + (int64_t)K * 6 + 1; }

int glio_batch_create(int device, int K, int band, int64_t max_constraints, glio_batch** out) {
    if (!out || K < 2 || band < 1 || band > BB_MAX_BAND || max_constraints < 1) { glio_set_error("bad batch shape"); return GLIO_E_ARG; }
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_batch_factor<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(batch_factor_lds_doubles(BB_MAX_BAND) * 8));
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_batch_factor<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(batch_factor_lds_doubles(9) * 8));
    if (glio_device_count() < 1) { glio_set_error("no HIP device visible: the batch stage has no CPU fallback"); return GLIO_E_HIP; }
    GLIO_HIP_CHECK(hipSetDevice(device));
    glio_batch* b = new glio_batch();
    memset(b, 0, sizeof *b);
    b->device = device; b->K = K; b->band = band; b->max_con = max_constraints;
    b->max_pairs = K * 2 * band;
    GLIO_HIP_CHECK(hipStreamCreateWithFlags(&b->own_stream, hipStreamNonBlocking));
    b->stream = b->own_stream;
    BALLOC(b->d_pair_i, b->max_pairs * 4); BALLOC(b->d_pair_j, b->max_pairs * 4); BALLOC(b->d_pair_off, (size_t)(b->max_pairs + 1) * 16);
    BALLOC(b->d_pair_rec, (size_t)b->max_pairs * (BP_REC + 1) * 8);
    BALLOC(b->d_pair_index, (size_t)K * (2 * band + 1) * 4);
    BALLOC(b->d_poses, (size_t)K * 7 * 8); BALLOC(b->d_newposes, (size_t)K * 7 * 8);
    BALLOC(b->d_M, (size_t)K * (band + 1) * 36 * 8); BALLOC(b->d_y, (size_t)K * 6 * 8); BALLOC(b->d_delta, (size_t)K * 6 * 8);
    BALLOC(b->d_scalar, 4 * 8);
    BALLOC(b->d_parts, (size_t)((K + 255) / 256 + 1) * 8);
    GLIO_HIP_CHECK(hipHostMalloc((void**)&b->h_poses, (size_t)K * 7 * 8));
    GLIO_HIP_CHECK(hipHostMalloc((void**)&b->h_scalar, 4 * 8));
    GLIO_HIP_CHECK(hipEventCreate(&b->ev0)); GLIO_HIP_CHECK(hipEventCreate(&b->ev1));
    b->bcr = glio_bcr_create(K, band);
    b->solver_mode = b->bcr ? 1 : 0;
    *out = b;
    return GLIO_OK;
}

// test / measurement hook: 0 = sequential banded Cholesky (one workgroup), 1 = block cyclic reduction
int glio_batch_debug_set_solver(glio_batch* b, int mode) {
    if (!b || mode < 0 || mode > 1 || (mode == 1 && !b->bcr)) return GLIO_E_ARG;
    b->solver_mode = mode;
    return GLIO_OK;
}

void glio_batch_destroy(glio_batch* b) {
    if (!b) return;
    hipSetDevice(b->device);
    hipStreamSynchronize(b->stream);
    glio_bcr_destroy(b->bcr);
    glio_batch_small_destroy(b);
    if (b->d_moments) hipFree(b->d_moments);
    if (b->d_mom_slots) hipFree(b->d_mom_slots);
    free(b->h_prev_pi); free(b->h_prev_pj); free(b->h_prev_off);
    void* ptrs[] = {b->d_cp, b->d_nc, b->d_score, b->d_pair_i, b->d_pair_j, b->d_pair_off, b->d_pair_rec, b->d_pair_index, b->d_poses,
                    b->d_newposes, b->d_M, b->d_y, b->d_delta, b->d_scalar, b->d_parts};
    for (void* p : ptrs) if (p) hipFree(p);
    hipHostFree(b->h_poses); hipHostFree(b->h_scalar);
    hipEventDestroy(b->ev0); hipEventDestroy(b->ev1);
    hipStreamDestroy(b->own_stream);
    delete b;
}

int glio_batch_set_stream(glio_batch* b, void* s) {
    if (!b) return GLIO_E_ARG;
    b->stream = s ? (hipStream_t)s : b->own_stream;
    return GLIO_OK;
}

// pair segmentation from the (sorted) host index arrays
static int build_pairs(glio_batch* b, int64_t n, const int32_t* ci, const int32_t* cj) {
    const int K = b->K, band = b->band, wdt = 2 * band + 1;
    std::vector<int> pi, pj, index((size_t)K * wdt, -1);
    std::vector<long long> off;
    for (int64_t i = 0; i < n; ++i) {
        const int a = ci[i], c = cj[i];
        if (a < 0 || a >= K || c < 0 || c >= K || a == c || std::abs(a - c) > band) { glio_set_error("constraint %lld: keyframes (%d,%d) outside band %d", (long long)i, a, c, band); return GLIO_E_ARG; }
        if (pi.empty() || pi.back() != a || pj.back() != c) {
            if (!pi.empty() && (a < pi.back() || (a == pi.back() && c < pj.back()))) { glio_set_error("constraints must be sorted by (ci, cj)"); return GLIO_E_ARG; }
            if (index[(size_t)a * wdt + (c - a) + band] >= 0) { glio_set_error("constraints must be sorted by (ci, cj)"); return GLIO_E_ARG; }
            index[(size_t)a * wdt + (c - a) + band] = (int)pi.size();
            pi.push_back(a); pj.push_back(c); off.push_back(i);
        }
    }
    off.push_back(n);
    if ((int)pi.size() > b->max_pairs) return GLIO_E_ARG;
    b->src_min = pi.empty() ? 0 : pi.front(); b->src_max = pi.empty() ? -1 : pi.back();      // (sorted by ci)
    b->n_pairs = (int)pi.size();
    if (b->n_pairs) {
        GLIO_HIP_CHECK(hipMemcpy(b->d_pair_i, pi.data(), pi.size() * 4, hipMemcpyHostToDevice));
        GLIO_HIP_CHECK(hipMemcpy(b->d_pair_j, pj.data(), pj.size() * 4, hipMemcpyHostToDevice));
    }
    std::vector<long long> be(2 * pi.size() + 2, 0);
    for (size_t q = 0; q < pi.size(); ++q) { be[2 * q] = off[q]; be[2 * q + 1] = off[q + 1]; }
    GLIO_HIP_CHECK(hipMemcpy(b->d_pair_off, be.data(), be.size() * 8, hipMemcpyHostToDevice));
    GLIO_HIP_CHECK(hipMemcpy(b->d_pair_index, index.data(), index.size() * 4, hipMemcpyHostToDevice));
    b->n_con = n;
    return GLIO_OK;
}

int glio_batch_set_constraints_dev(glio_batch* b, int64_t n, const int32_t* ci, const int32_t* cj, const float* cp_dev,
                                   const double* nc_dev, const double* score_dev) {
    if (!b || n < 0 || (n > 0 && (!ci || !cj || !cp_dev || !nc_dev || !score_dev))) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipSetDevice(b->device));
    const int rc = build_pairs(b, n, ci, cj);
    if (rc) return rc;
    b->cp = reinterpret_cast<const float4*>(cp_dev); b->nc = nc_dev; b->score = score_dev;
    b->moments_valid = 0; b->h_prev_n = -1;
    return GLIO_OK;
}

// the same, from a pair list (what glio_bassoc_run produces): pair p = keyframes (pair_ci[p], pair_cj[p]) with
// pair_count[p] consecutive records; pairs sorted by (ci, cj); empty pairs are skipped
int glio_batch_set_constraints_pairs_dev(glio_batch* b, int n_pairs, const int32_t* pair_ci, const int32_t* pair_cj, const int64_t* pair_count,
                                         const float* cp_dev, const double* nc_dev, const double* score_dev) {
    return glio_batch_update_constraints_pairs_dev(b, n_pairs, pair_ci, pair_cj, pair_count, cp_dev, nc_dev, score_dev, nullptr);
}
// the same for a constraint set that differs from the previous one only in the pairs marked in `pair_changed` (one byte per input pair, non-zero =
// its records were replaced; NULL = everything may have changed).  What the outer rounds of optimizeBatch do: the first / last search_range keyframes
// are re-searched, the interior constraints are the stored ones (Estimator.cpp:3018-3030).  The next solve then takes the moments of the marked pairs
// only.  Falls back to "everything changed" when the list of non-empty pairs is not the previous one.
int glio_batch_update_constraints_pairs_dev(glio_batch* b, int n_pairs, const int32_t* pair_ci, const int32_t* pair_cj, const int64_t* pair_count,
                                            const float* cp_dev, const double* nc_dev, const double* score_dev, const uint8_t* pair_changed) {
    return glio_batch_update_constraints_pairs_at_dev(b, n_pairs, pair_ci, pair_cj, pair_count, nullptr, cp_dev, nc_dev, score_dev, pair_changed);
}
// the same with the record range of every pair given explicitly: pair p's records are [pair_offset[p], pair_offset[p] + pair_count[p]) of the three
// arrays (NULL: the pairs follow each other).  For a caller that keeps the stored interior constraints in place and rewrites only the regions of the
// re-searched end keyframes (whose counts change from round to round) instead of re-packing tens of gigabytes every round.
int glio_batch_update_constraints_pairs_at_dev(glio_batch* b, int n_pairs, const int32_t* pair_ci, const int32_t* pair_cj, const int64_t* pair_count,
                                               const int64_t* pair_offset, const float* cp_dev, const double* nc_dev, const double* score_dev, const uint8_t* pair_changed) {
    if (!b || n_pairs < 0 || (n_pairs > 0 && (!pair_ci || !pair_cj || !pair_count || !cp_dev || !nc_dev || !score_dev))) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipSetDevice(b->device));
    const int K = b->K, band = b->band, wdt = 2 * band + 1;
    std::vector<int> pi, pj, index((size_t)K * wdt, -1), changed_slots;
    std::vector<long long> off;
    long long run = 0;
    for (int p = 0; p < n_pairs; ++p) {
        const int a = pair_ci[p], c = pair_cj[p];
        if (a < 0 || a >= K || c < 0 || c >= K || a == c || std::abs(a - c) > band || pair_count[p] < 0) { glio_set_error("pair %d: keyframes (%d,%d) outside band %d", p, a, c, band); return GLIO_E_ARG; }
        if (p > 0 && (a < pair_ci[p - 1] || (a == pair_ci[p - 1] && c <= pair_cj[p - 1]))) { glio_set_error("pairs must be sorted by (ci, cj)"); return GLIO_E_ARG; }
        if (pair_offset && pair_offset[p] < 0) { glio_set_error("pair %d: negative record offset", p); return GLIO_E_ARG; }
        if (pair_count[p] > 0) {
            index[(size_t)a * wdt + (c - a) + band] = (int)pi.size();
            if (!pair_changed || pair_changed[p]) changed_slots.push_back((int)pi.size());
            pi.push_back(a); pj.push_back(c);
            const long long beg = pair_offset ? (long long)pair_offset[p] : run;
            off.push_back(beg); off.push_back(beg + (long long)pair_count[p]);
        }
        run += pair_count[p];
    }
    if ((int)pi.size() > b->max_pairs) return GLIO_E_ARG;
    if (pair_offset && pi.size() > 1) {            // explicit ranges must not overlap (they are borrowed memory: their upper end is the caller's to keep inside its arrays)
        std::vector<std::pair<long long, long long>> rg(pi.size());
        for (size_t q = 0; q < pi.size(); ++q) rg[q] = {off[2 * q], off[2 * q + 1]};
        std::sort(rg.begin(), rg.end());
        for (size_t q = 1; q < rg.size(); ++q) if (rg[q].first < rg[q - 1].second) { glio_set_error("record ranges of two pairs overlap ([%lld, %lld) and [%lld, %lld))", rg[q - 1].first, rg[q - 1].second, rg[q].first, rg[q].second); return GLIO_E_ARG; }
    }
    b->src_min = pi.empty() ? 0 : pi.front(); b->src_max = pi.empty() ? -1 : pi.back();      // (sorted by ci)
    {   // do the moment records of the unmarked pairs still stand?  only when the list of non-empty pairs is the one they were taken for
        bool same = pair_changed && b->moments_valid && b->h_prev_n == (int)pi.size() && b->d_moments && b->moments_pairs >= (int)pi.size();
        for (size_t q = 0; same && q < pi.size(); ++q) same = b->h_prev_pi[q] == pi[q] && b->h_prev_pj[q] == pj[q];
        if (same && b->h_prev_off) {                // an unmarked pair whose record range moved or changed length is a replaced pair all the same (advisor, round 4)
            std::vector<char> marked(pi.size(), 0);
            for (int sl : changed_slots) marked[(size_t)sl] = 1;
            for (size_t q = 0; q < pi.size(); ++q)
                if (!marked[q] && (b->h_prev_off[2 * q] != off[2 * q] || b->h_prev_off[2 * q + 1] != off[2 * q + 1])) { marked[q] = 1; changed_slots.push_back((int)q); }
            std::sort(changed_slots.begin(), changed_slots.end());
        }
        if (same) {
            if ((int)changed_slots.size() > b->mom_slots_cap) {
                if (b->d_mom_slots) hipFree(b->d_mom_slots);
                b->d_mom_slots = nullptr; b->mom_slots_cap = 0;
                const int cap = (int)changed_slots.size() + 256;
                GLIO_HIP_CHECK(hipMalloc((void**)&b->d_mom_slots, (size_t)cap * 4));
                b->mom_slots_cap = cap;
            }
            // (slots marked by an earlier update and not yet consumed by a solve stay marked: merge)
            if (b->n_mom_changed > 0) { b->moments_valid = 0; }
            else {
                if (!changed_slots.empty()) GLIO_HIP_CHECK(hipMemcpy(b->d_mom_slots, changed_slots.data(), changed_slots.size() * 4, hipMemcpyHostToDevice));
                b->n_mom_changed = (int)changed_slots.size();
            }
        } else b->moments_valid = 0;
        free(b->h_prev_pi); free(b->h_prev_pj); free(b->h_prev_off);
        b->h_prev_pi = (int*)malloc((pi.size() + 1) * 4); b->h_prev_pj = (int*)malloc((pi.size() + 1) * 4); b->h_prev_off = (long long*)malloc((2 * pi.size() + 2) * 8);
        if (!pi.empty()) { memcpy(b->h_prev_pi, pi.data(), pi.size() * 4); memcpy(b->h_prev_pj, pj.data(), pj.size() * 4); memcpy(b->h_prev_off, off.data(), 2 * pi.size() * 8); }
        b->h_prev_n = (int)pi.size();
    }
    b->n_pairs = (int)pi.size();
    if (b->n_pairs) {
        GLIO_HIP_CHECK(hipMemcpy(b->d_pair_i, pi.data(), pi.size() * 4, hipMemcpyHostToDevice));
        GLIO_HIP_CHECK(hipMemcpy(b->d_pair_j, pj.data(), pj.size() * 4, hipMemcpyHostToDevice));
    }
    off.push_back(0); off.push_back(0);
    GLIO_HIP_CHECK(hipMemcpy(b->d_pair_off, off.data(), off.size() * 8, hipMemcpyHostToDevice));
    GLIO_HIP_CHECK(hipMemcpy(b->d_pair_index, index.data(), index.size() * 4, hipMemcpyHostToDevice));
    b->n_con = run;
    b->cp = reinterpret_cast<const float4*>(cp_dev); b->nc = nc_dev; b->score = score_dev;
    return GLIO_OK;
}

int glio_batch_set_constraints(glio_batch* b, int64_t n, const int32_t* ci, const int32_t* cj, const float* cp,
                               const double* nc, const double* score) {
    if (!b || n < 0 || n > b->max_con) { glio_set_error("too many constraints"); return GLIO_E_ARG; }
    GLIO_HIP_CHECK(hipSetDevice(b->device));
    if (!b->d_cp) { BALLOC(b->d_cp, (size_t)b->max_con * 16); BALLOC(b->d_nc, (size_t)b->max_con * 48); BALLOC(b->d_score, (size_t)b->max_con * 8); }
    if (n > 0) {
        GLIO_HIP_CHECK(hipMemcpy(b->d_cp, cp, (size_t)n * 16, hipMemcpyHostToDevice));
        GLIO_HIP_CHECK(hipMemcpy(b->d_nc, nc, (size_t)n * 48, hipMemcpyHostToDevice));
        GLIO_HIP_CHECK(hipMemcpy(b->d_score, score, (size_t)n * 8, hipMemcpyHostToDevice));
    }
    return glio_batch_set_constraints_dev(b, n, ci, cj, reinterpret_cast<const float*>(b->d_cp), b->d_nc, b->d_score);
}

static void enqueue_batch_linearize(glio_batch* b, double* Hg_dev);
extern "C++" void glio_batch_enqueue_linearize(glio_batch* b, double* Hg_dev) { enqueue_batch_linearize(b, Hg_dev); }
static void enqueue_batch_linearize_sel(glio_batch* b, const BtSel& sel, const double* poses0, const double* poses1, double* Hg0, double* Hg1, int k0, int k1, int mode = 0) {
    const int K = b->K, band = b->band;
    const long long nH = (long long)K * (band + 1) * 36, total = nH + (long long)K * 6;
    double* cost_dense = b->d_pair_rec + (size_t)b->max_pairs * BP_REC;
    if (mode != 0 && (!b->d_moments || b->moments_pairs < b->n_pairs)) mode = 0;          // (no moment buffer: stream)
    if (b->n_pairs > 0 && mode == 0)
        hipLaunchKernelGGL(k_batch_pairs, dim3((b->n_pairs + 3) / 4), dim3(256), 0, b->stream, b->cp, b->nc, b->score, b->d_pair_i, b->d_pair_j,
                           b->d_pair_off, b->n_pairs, poses0, b->d_pair_rec, sel, poses1, cost_dense);
    if (b->n_pairs > 0 && mode == 1) {
        // the records are a cache keyed by the constraint set: all pairs after glio_batch_set_constraints*, only the pairs the caller marked as
        // replaced after glio_batch_update_constraints_pairs_dev (the end keyframes of a round, Estimator.cpp:3018-3030), none when nothing changed
        if (!b->moments_valid)
            hipLaunchKernelGGL(k_batch_moments, dim3((b->n_pairs + 3) / 4), dim3(256), 0, b->stream, b->cp, b->nc, b->score, b->d_pair_i, b->d_pair_j,
                               b->d_pair_off, b->n_pairs, poses0, sel, poses1, b->d_moments, (const int*)nullptr, 0);
        else if (b->n_mom_changed > 0)
            hipLaunchKernelGGL(k_batch_moments, dim3((b->n_mom_changed + 3) / 4), dim3(256), 0, b->stream, b->cp, b->nc, b->score, b->d_pair_i, b->d_pair_j,
                               b->d_pair_off, b->n_pairs, poses0, sel, poses1, b->d_moments, (const int*)b->d_mom_slots, b->n_mom_changed);
        b->moments_valid = 1; b->n_mom_changed = 0;
    }
    if (b->n_pairs > 0 && mode != 0)
        hipLaunchKernelGGL(k_batch_moment_eval, dim3((b->n_pairs + 3) / 4), dim3(256), 0, b->stream, b->d_moments, b->d_pair_i, b->d_pair_j, b->n_pairs,
                           poses0, b->d_pair_rec, sel, poses1, cost_dense);
    const long long rows = (long long)(k1 - k0) * ((band + 1) * 36 + 6);
    if (rows > 0)
        hipLaunchKernelGGL(k_batch_assemble, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, b->stream, b->d_pair_rec, b->d_pair_index, K, band,
                           b->n_pairs, Hg0, sel, Hg1, k0, k1);
    hipLaunchKernelGGL(k_batch_cost, dim3(1), dim3(1024), 0, b->stream, cost_dense, b->n_pairs, Hg0 + total, sel, Hg1 ? Hg1 + total : nullptr);
}
extern "C++" void glio_batch_enqueue_linearize_sel(glio_batch* b, const BtSel& sel, const double* poses0, const double* poses1, double* Hg0, double* Hg1, int k0, int k1, int mode) {
    enqueue_batch_linearize_sel(b, sel, poses0, poses1, Hg0, Hg1, k0, k1, mode);
}
extern "C++" int glio_batch_moments_ensure(glio_batch* b) {
    if (b->d_moments && b->moments_pairs >= b->n_pairs) return GLIO_OK;
    if (b->d_moments) { hipFree(b->d_moments); b->d_moments = nullptr; b->moments_pairs = 0; }
    b->moments_valid = 0;
    const int cap = b->n_pairs + b->n_pairs / 8 + 64;
    GLIO_HIP_CHECK(hipMalloc((void**)&b->d_moments, (size_t)cap * BM_REC * 8));
    b->moments_pairs = cap;
    return GLIO_OK;
}
static void enqueue_batch_linearize(glio_batch* b, double* Hg_dev) {
    BtSel sel; sel.cur = nullptr; sel.skip = nullptr; sel.want = 0;
    enqueue_batch_linearize_sel(b, sel, b->d_poses, b->d_poses, Hg_dev, Hg_dev, 0, b->K);
}

int glio_batch_linearize_dev(glio_batch* b, const double* poses, double* Hg_dev) {
    GLIO_TRACE("K8 glio_batch_linearize_dev");
    if (!b || !poses || !Hg_dev) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipSetDevice(b->device));
    memcpy(b->h_poses, poses, (size_t)b->K * 7 * 8);
    GLIO_HIP_CHECK(hipMemcpyAsync(b->d_poses, b->h_poses, (size_t)b->K * 7 * 8, hipMemcpyHostToDevice, b->stream));
    enqueue_batch_linearize(b, Hg_dev);
    GLIO_HIP_CHECK(hipGetLastError());
    GLIO_HIP_CHECK(hipStreamSynchronize(b->stream));
    return GLIO_OK;
}

int glio_batch_time_linearize(glio_batch* b, const double* poses, double* Hg_dev, int reps, float* ms_out) {
    if (!b || !poses || !Hg_dev || reps < 1 || !ms_out) return GLIO_E_ARG;
    int rc = glio_batch_linearize_dev(b, poses, Hg_dev);      // warm-up + pose upload
    if (rc) return rc;
    GLIO_HIP_CHECK(hipEventRecord(b->ev0, b->stream));
    for (int r = 0; r < reps; ++r) enqueue_batch_linearize(b, Hg_dev);
    GLIO_HIP_CHECK(hipEventRecord(b->ev1, b->stream));
    GLIO_HIP_CHECK(hipStreamSynchronize(b->stream));
    float ms = 0;
    GLIO_HIP_CHECK(hipEventElapsedTime(&ms, b->ev0, b->ev1));
    *ms_out = ms / reps;
    return GLIO_OK;
}

// test hook: one linearisation of the pose problem's plane constraints through the moment form (mode 1: take the moments at `poses`, then
// evaluate; mode 2: evaluate the moments stored by an earlier mode-1 call at `poses`) or by streaming (mode 0)
extern "C" int glio_debug_batch_linearize_mode(glio_batch* b, const double* poses, double* Hg_dev, int mode) {
    if (!b || !poses || !Hg_dev || mode < 0 || mode > 2) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipSetDevice(b->device));
    { const int rc = glio_batch_moments_ensure(b); if (rc) return rc; }
    memcpy(b->h_poses, poses, (size_t)b->K * 7 * 8);
    GLIO_HIP_CHECK(hipMemcpyAsync(b->d_poses, b->h_poses, (size_t)b->K * 7 * 8, hipMemcpyHostToDevice, b->stream));
    BtSel sel; sel.cur = nullptr; sel.skip = nullptr; sel.want = 0;
    enqueue_batch_linearize_sel(b, sel, b->d_poses, b->d_poses, Hg_dev, Hg_dev, 0, b->K, mode);
    GLIO_HIP_CHECK(hipGetLastError());
    GLIO_HIP_CHECK(hipStreamSynchronize(b->stream));
    return GLIO_OK;
}
extern "C" int glio_debug_batch_moment_state(glio_batch* b, int32_t out2[2]) {      // test hook: {records valid, pairs marked as replaced}
    if (!b || !out2) return GLIO_E_ARG;
    out2[0] = b->moments_valid; out2[1] = b->moments_valid ? b->n_mom_changed : -1;
    return GLIO_OK;
}
// measurement hook: the same with the linearisation taken through the pairs' moments -- mode 1: moments pass + evaluation (what the first
// linearisation of a solve costs), mode 2: evaluation of the stored moments (every later one)
extern "C" int glio_debug_batch_time_linearize_mode(glio_batch* b, const double* poses, double* Hg_dev, int mode, int reps, float* ms_out) {
    if (!b || !poses || !Hg_dev || reps < 1 || !ms_out || mode < 0 || mode > 2) return GLIO_E_ARG;
    int rc = glio_batch_linearize_dev(b, poses, Hg_dev);
    if (rc) return rc;
    rc = glio_batch_moments_ensure(b);
    if (rc) return rc;
    BtSel sel; sel.cur = nullptr; sel.skip = nullptr; sel.want = 0;
    enqueue_batch_linearize_sel(b, sel, b->d_poses, b->d_poses, Hg_dev, Hg_dev, 0, b->K, 1);
    GLIO_HIP_CHECK(hipEventRecord(b->ev0, b->stream));
    for (int r = 0; r < reps; ++r) {
        if (mode == 1) b->moments_valid = 0;          // (the records are a cache: every repetition of the measurement takes them again)
        enqueue_batch_linearize_sel(b, sel, b->d_poses, b->d_poses, Hg_dev, Hg_dev, 0, b->K, mode);
    }
    GLIO_HIP_CHECK(hipEventRecord(b->ev1, b->stream));
    GLIO_HIP_CHECK(hipStreamSynchronize(b->stream));
    float ms = 0;
    GLIO_HIP_CHECK(hipEventElapsedTime(&ms, b->ev0, b->ev1));
    *ms_out = ms / reps;
    return GLIO_OK;
}

// ---- helpers for a host that never includes HIP headers (glio_amd/host/glio_batch_backend.hpp): the reduced buffer lives on
// the device, the host hands its pointer and the batch stream to the collective (ncclAllReduce) between linearise and step
int glio_batch_hg_alloc_dev(glio_batch* b, double** out) {
    if (!b || !out) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipSetDevice(b->device));
    GLIO_HIP_CHECK(hipMalloc((void**)out, (size_t)glio_batch_hg_size(b->K, b->band) * 8));
    return GLIO_OK;
}
int glio_batch_hg_free_dev(glio_batch* b, double* p) {
    if (!b) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipSetDevice(b->device));
    if (p) GLIO_HIP_CHECK(hipFree(p));
    return GLIO_OK;
}
int glio_batch_get_stream(glio_batch* b, void** out) {
    if (!b || !out) return GLIO_E_ARG;
    *out = (void*)b->stream;
    return GLIO_OK;
}
int glio_batch_read_dev(glio_batch* b, const double* dev, int64_t first, int64_t n, double* out) {
    if (!b || !dev || !out || first < 0 || n < 0) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipSetDevice(b->device));
    GLIO_HIP_CHECK(hipMemcpyAsync(out, dev + first, (size_t)n * 8, hipMemcpyDeviceToHost, b->stream));
    GLIO_HIP_CHECK(hipStreamSynchronize(b->stream));
    return GLIO_OK;
}
int glio_batch_synchronize(glio_batch* b) {
    if (!b) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipSetDevice(b->device));
    GLIO_HIP_CHECK(hipStreamSynchronize(b->stream));
    return GLIO_OK;
}

// timing hook: average ms of `reps` banded solves (H + lambda diag H) x = g with the selected solver (HIP events on the batch stream)
int glio_batch_time_solve(glio_batch* b, const double* Hg_dev, double lambda, int reps, float* ms_out) {
    if (!b || !Hg_dev || reps < 1 || !ms_out) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipSetDevice(b->device));
    const int K = b->K, band = b->band;
    int* d_fail = reinterpret_cast<int*>(b->d_scalar + 2);
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) GLIO_HIP_CHECK(hipEventRecord(b->ev0, b->stream));
        for (int r = 0; r < (pass == 0 ? 1 : reps); ++r) {
            int* d_fail_bcr = nullptr;
            if (b->solver_mode == 1) glio_bcr_solve(b->bcr, Hg_dev, lambda, b->d_delta, &d_fail_bcr, b->stream);
            else {
                if (BB_ROWS(band) > 64)
                    hipLaunchKernelGGL(k_batch_factor<true>, dim3(1), dim3(BBF_THREADS), batch_factor_lds_doubles(band) * 8, b->stream, Hg_dev, K, band, lambda, b->d_M, b->d_y, d_fail);
                else
                    hipLaunchKernelGGL(k_batch_factor<false>, dim3(1), dim3(BBF_THREADS), batch_factor_lds_doubles(band) * 8, b->stream, Hg_dev, K, band, lambda, b->d_M, b->d_y, d_fail);
                hipLaunchKernelGGL(k_batch_backsolve, dim3(1), dim3(64), 0, b->stream, b->d_M, b->d_y, K, band, b->d_delta);
            }
        }
        if (pass == 1) GLIO_HIP_CHECK(hipEventRecord(b->ev1, b->stream));
        GLIO_HIP_CHECK(hipStreamSynchronize(b->stream));
    }
    float ms = 0;
    GLIO_HIP_CHECK(hipEventElapsedTime(&ms, b->ev0, b->ev1));
    *ms_out = ms / reps;
    return GLIO_OK;
}

int glio_batch_step_dev(glio_batch* b, const double* Hg_dev, double lambda, const double* poses_in, double* poses_out, double* model_decrease) {
    GLIO_TRACE("glio_batch_step_dev (banded solve)");
    if (!b || !Hg_dev || !poses_in || !poses_out) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipSetDevice(b->device));
    const int K = b->K, band = b->band;
    memcpy(b->h_poses, poses_in, (size_t)K * 7 * 8);
    GLIO_HIP_CHECK(hipMemcpyAsync(b->d_poses, b->h_poses, (size_t)K * 7 * 8, hipMemcpyHostToDevice, b->stream));
    GLIO_HIP_CHECK(hipMemsetAsync(b->d_scalar, 0, 4 * 8, b->stream));
    int* d_fail = reinterpret_cast<int*>(b->d_scalar + 2);
    int* d_fail_bcr = nullptr;
    if (b->solver_mode == 1) glio_bcr_solve(b->bcr, Hg_dev, lambda, b->d_delta, &d_fail_bcr, b->stream);
    else if (BB_ROWS(band) > 64)
        hipLaunchKernelGGL(k_batch_factor<true>, dim3(1), dim3(BBF_THREADS), batch_factor_lds_doubles(band) * 8, b->stream, Hg_dev, K, band, lambda, b->d_M, b->d_y, d_fail);
    else
        hipLaunchKernelGGL(k_batch_factor<false>, dim3(1), dim3(BBF_THREADS), batch_factor_lds_doubles(band) * 8, b->stream, Hg_dev, K, band, lambda, b->d_M, b->d_y, d_fail);
    if (b->solver_mode != 1) hipLaunchKernelGGL(k_batch_backsolve, dim3(1), dim3(64), 0, b->stream, b->d_M, b->d_y, K, band, b->d_delta);
    if (d_fail_bcr) GLIO_HIP_CHECK(hipMemcpyAsync(d_fail, d_fail_bcr, 4, hipMemcpyDeviceToDevice, b->stream));
    hipLaunchKernelGGL(k_batch_apply, dim3((K + 255) / 256), dim3(256), 0, b->stream, Hg_dev, b->d_delta, b->d_poses, K, band, b->d_newposes, b->d_parts);
    hipLaunchKernelGGL(k_batch_apply_sum, dim3(1), dim3(64), 0, b->stream, b->d_parts, (K + 255) / 256, b->d_scalar);
    GLIO_HIP_CHECK(hipGetLastError());
    GLIO_HIP_CHECK(hipMemcpyAsync(b->h_poses, b->d_newposes, (size_t)K * 7 * 8, hipMemcpyDeviceToHost, b->stream));
    GLIO_HIP_CHECK(hipMemcpyAsync(b->h_scalar, b->d_scalar, 4 * 8, hipMemcpyDeviceToHost, b->stream));
    GLIO_HIP_CHECK(hipStreamSynchronize(b->stream));
    int fail = 0;
    memcpy(&fail, b->h_scalar + 2, 4);
    if (fail) { glio_set_error("banded Cholesky breakdown (%s, code %d)", b->solver_mode == 1 ? "block cyclic reduction" : "sequential", fail); return GLIO_E_NUMERIC; }
    memcpy(poses_out, b->h_poses, (size_t)K * 7 * 8);
    if (model_decrease) *model_decrease = b->h_scalar[0];
    return GLIO_OK;
}

}  // extern "C"
