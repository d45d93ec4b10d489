// k3_device.h -- device code of K3 (LiDAR point-to-plane linearisation) shared by lidar_kernels.hip (K3 as its own launch,
// the LDS-DMA variant, the stream-read ceiling) and factor_kernels.hip (k_linearize_all: K3 beside the small factors).
// See lidar_kernels.hip for the math and the roofline.
#pragma once
#include "glio_device.h"

struct LidarConst {
    double RlbT[9];   // R(q_lb)^T
    double tlb[3];
    double huber;
};

// MARG = true: the marginalization's convention for the quaternion block (reference
// GLIO/src/MarginalizationFactor.cpp:9-17, quirk Q8): the x,y,z columns of the GLOBAL 1x4 Jacobian that
// autodiff produces through Eigen's q*v formula, i.e. s n^T (-2w[v]x - 2[u x v]x - 2[u]x[v]x) with
// v = p_b, q = (w,u)  =  -2 s [ w (n x v) + n x (u x v) + (n x u) x v ].
template <bool MARG>
__device__ __forceinline__ void lidar_accumulate(const float4 p, const float4 pl, const double s,
                                                 const double M[9], const double t[3], const double tlb[3],
                                                 const double a, double acc[GLIO_LIDAR_ACC],
                                                 const double RlbT[9], const double q[4]) {
    const double cx = (double)p.x - tlb[0], cy = (double)p.y - tlb[1], cz = (double)p.z - tlb[2];
    const double rx = M[0] * cx + M[1] * cy + M[2] * cz;
    const double ry = M[3] * cx + M[4] * cy + M[5] * cz;
    const double rz = M[6] * cx + M[7] * cy + M[8] * cz;
    const double nx = (double)pl.x, ny = (double)pl.y, nz = (double)pl.z;
    const double e = nx * (rx + t[0]) + ny * (ry + t[1]) + nz * (rz + t[2]) + (double)pl.w;
    const double r = s * e;
    double J[6];
    J[0] = s * nx; J[1] = s * ny; J[2] = s * nz;
    const double s2 = 2.0 * s;
    if (!MARG) {
        J[3] = s2 * (ry * nz - rz * ny);
        J[4] = s2 * (rz * nx - rx * nz);
        J[5] = s2 * (rx * ny - ry * nx);
    } else {
        const double vx = RlbT[0] * cx + RlbT[1] * cy + RlbT[2] * cz, vy = RlbT[3] * cx + RlbT[4] * cy + RlbT[5] * cz, vz = RlbT[6] * cx + RlbT[7] * cy + RlbT[8] * cz;
        const double w = q[0], ux = q[1], uy = q[2], uz = q[3];
        const double nvx = ny * vz - nz * vy, nvy = nz * vx - nx * vz, nvz = nx * vy - ny * vx;          // n x v
        const double uvx = uy * vz - uz * vy, uvy = uz * vx - ux * vz, uvz = ux * vy - uy * vx;          // u x v
        const double nux = ny * uz - nz * uy, nuy = nz * ux - nx * uz, nuz = nx * uy - ny * ux;          // n x u
        J[3] = -s2 * (w * nvx + (ny * uvz - nz * uvy) + (nuy * vz - nuz * vy));
        J[4] = -s2 * (w * nvy + (nz * uvx - nx * uvz) + (nuz * vx - nux * vz));
        J[5] = -s2 * (w * nvz + (nx * uvy - ny * uvx) + (nux * vy - nuy * vx));
    }
    const double ar = fabs(r);
    const bool inl = ar <= a;
    // rho' = a / |r| outside the Huber radius.  The IEEE division is ~14 of the ~115 vector instructions of a residual (v_div_scale x2, v_rcp, five fma,
    // v_div_fmas, v_div_fixup) and the kernel's fp64 issue is not hidden behind its loads; |r| > a > 0 here, so no special case can occur: the hardware
    // reciprocal and two Newton steps (<= 1 ulp from the quotient; the factor tests hold K3 to the oracle at 1e-10)
    double rc = __builtin_amdgcn_rcp(inl ? 1.0 : ar);
    rc = fma(fma(-(inl ? 1.0 : ar), rc, 1.0), rc, rc);
    rc = fma(fma(-(inl ? 1.0 : ar), rc, 1.0), rc, rc);
    const double w = inl ? 1.0 : a * rc;                  // rho'
    const double rho = inl ? r * r : 2.0 * a * ar - a * a;
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const double wi = w * J[i];
#pragma unroll
        for (int j = i; j < 6; ++j) acc[k++] += wi * J[j];
        acc[21 + i] += wi * r;
    }
    acc[27] += 0.5 * rho;
}

__device__ __forceinline__ void k3_reduce_store(const double acc[GLIO_LIDAR_ACC], double* __restrict__ partials, const int kf, const int bx, const int nb) {
    // Wave reduction as a value-splitting butterfly: at every halving step a lane keeps one half of its
    // values and ships the other half to its partner, so 32 (padded) accumulators need 16+8+4+2+1+1 = 32
    // 64-bit shuffles instead of 28 x 6 = 168, in six dependent rounds.  After the xor-2 round lane L owns
    // accumulator k = b5 + 2 b4 + 4 b3 + 8 b2 + 16 b1 (b_i = bit i of L); the xor-1 round completes the sum.
    __shared__ double red[GLIO_K3_THREADS / GLIO_WAVE][32];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // (the exchanges go through V_PERMLANE32/16_SWAP and DPP, not through ds_bpermute: glio_device.h, "Cross-lane exchange".  The two swap
    // stages need no keep / send selects at all: swapping a's upper half with b's lower half IS the reduce-scatter step.)
    double v16[16], v8[8], v4[4], v2[2], v1;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const double a = acc[2 * i], b = (2 * i + 1 < GLIO_LIDAR_ACC) ? acc[2 * i + 1] : 0.0;
        double x, y;
        lane_swap32(a, b, x, y);
        v16[i] = x + y;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        double x, y;
        lane_swap16(v16[2 * i], v16[2 * i + 1], x, y);
        v8[i] = x + y;
    }
    {
        const bool hi = (lane & 8) != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double keep = hi ? v8[2 * i + 1] : v8[2 * i], send = hi ? v8[2 * i] : v8[2 * i + 1];
            v4[i] = keep + lane_xor_row_d<8>(send);
        }
    }
    {
        const bool hi = (lane & 4) != 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const double keep = hi ? v4[2 * i + 1] : v4[2 * i], send = hi ? v4[2 * i] : v4[2 * i + 1];
            v2[i] = keep + lane_xor_row_d<4>(send);
        }
    }
    {
        const bool hi = (lane & 2) != 0;
        const double keep = hi ? v2[1] : v2[0], send = hi ? v2[0] : v2[1];
        v1 = keep + lane_xor_row_d<2>(send);
    }
    v1 += lane_xor_row_d<1>(v1);
    if ((lane & 1) == 0) {
        const int k = ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2) | (((lane >> 2) & 1) << 3) | (((lane >> 1) & 1) << 4);
        red[wv][k] = v1;
    }
    __syncthreads();
    if (threadIdx.x < GLIO_LIDAR_ACC) {
        double v = red[0][threadIdx.x];
#pragma unroll
        for (int w2 = 1; w2 < GLIO_K3_THREADS / GLIO_WAVE; ++w2) v += red[w2][threadIdx.x];
        partials[((size_t)kf * nb + bx) * GLIO_LIDAR_ACC + threadIdx.x] = v;
    }
}

typedef float k3_f4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ float4 k3_load4(const float4* p) {
    if (NT) { const k3_f4 v = __builtin_nontemporal_load(reinterpret_cast<const k3_f4*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
    return *p;
}
template <bool NT> __device__ __forceinline__ double k3_load1(const double* p) { return NT ? __builtin_nontemporal_load(p) : *p; }

// The work of one K3 workgroup: block bx of the nb that share keyframe kf.  Called by k_lidar_linearize (one launch for K3
// alone) and by k_linearize_all (factor_kernels.hip), where K3 workgroups run beside the small-factor workgroups.
template <int UNROLL, bool MARG, bool NT = false, bool PIPE = false>
__device__ __forceinline__ void k3_body(
    const float4* __restrict__ pts, const float4* __restrict__ planes, const double* __restrict__ scores,
    const int* __restrict__ count, const int cap, const double* __restrict__ x, const int W,
    const LidarConst& lc, double* __restrict__ partials, const int kf, const int bx, const int nb) {
    const int n = count[kf];

    // per-keyframe constants: M = R(q) R_lb^T, t
    double q[4], R[9], M[9], t[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k] = x[3 * kf + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = x[3 * W + 4 * kf + k];
    d_q2R(q, R);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            M[i * 3 + j] = R[i * 3 + 0] * lc.RlbT[0 * 3 + j] + R[i * 3 + 1] * lc.RlbT[1 * 3 + j] + R[i * 3 + 2] * lc.RlbT[2 * 3 + j];

    double acc[GLIO_LIDAR_ACC];
#pragma unroll
    for (int k = 0; k < GLIO_LIDAR_ACC; ++k) acc[k] = 0.0;

    const size_t base = (size_t)kf * cap;
    const float4* __restrict__ P = pts + base;
    const float4* __restrict__ Q = planes + base;
    const double* __restrict__ S = scores + base;
    const int stride = nb * GLIO_K3_THREADS;
    int i = bx * GLIO_K3_THREADS + threadIdx.x;
    if (PIPE) {
        // software pipeline: the loads of batch k+1 are issued before the arithmetic of batch k, so that one resident
        // wavefront overlaps its own ~2.3 us of fp64 work with the memory stream (few, long-running workgroups)
        float4 p[UNROLL], pl[UNROLL];
        double s[UNROLL];
        bool have = i + (UNROLL - 1) * stride < n;
        if (have) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { p[u] = k3_load4<NT>(P + i + u * stride); pl[u] = k3_load4<NT>(Q + i + u * stride); s[u] = k3_load1<NT>(S + i + u * stride); }
        }
        while (have) {
            const int inext = i + UNROLL * stride;
            const bool more = inext + (UNROLL - 1) * stride < n;
            float4 p2[UNROLL], pl2[UNROLL];
            double s2[UNROLL];
            if (more) {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) { p2[u] = k3_load4<NT>(P + inext + u * stride); pl2[u] = k3_load4<NT>(Q + inext + u * stride); s2[u] = k3_load1<NT>(S + inext + u * stride); }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) lidar_accumulate<MARG>(p[u], pl[u], s[u], M, t, lc.tlb, lc.huber, acc, lc.RlbT, q);
            if (more) {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) { p[u] = p2[u]; pl[u] = pl2[u]; s[u] = s2[u]; }
            }
            i = inext; have = more;
        }
    }
    // main loop: UNROLL independent 16+16+8 B loads in flight per lane before any math
    for (; !PIPE && i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        float4 p[UNROLL], pl[UNROLL];
        double s[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            p[u] = k3_load4<NT>(P + i + u * stride);
            pl[u] = k3_load4<NT>(Q + i + u * stride);
            s[u] = k3_load1<NT>(S + i + u * stride);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) lidar_accumulate<MARG>(p[u], pl[u], s[u], M, t, lc.tlb, lc.huber, acc, lc.RlbT, q);
    }
    for (; i < n; i += stride) lidar_accumulate<MARG>(P[i], Q[i], S[i], M, t, lc.tlb, lc.huber, acc, lc.RlbT, q);

    k3_reduce_store(acc, partials, kf, bx, nb);
}


// ------------------------------------------------------------------------------------------------
// K3, fp32-Jacobian form with the J^T J contraction on the matrix core (BASELINE config C5; opts.lidar_precision =
// GLIO_LIDAR_F32_MFMA).  32 B per residual: float4 point with the SCORE as a float in .w (k_pack_points_f32) + float4
// plane.  Per residual the error e = n^.(M c + t) + d^ is still formed in double (world coordinates are ~100 m: a float
// there costs ~1e-5 m, the size of the answers the gate is stated in), the 1x6 Jacobian and the loss weights in float.
// A wavefront takes 64 consecutive residuals, writes the weighted row  a = sqrt(rho') [J0..J5, r, 0]  of each into an
// LDS tile [8 columns][64 residuals] and reads it back transposed into the operand layout of v_mfma_f32_16x16x4_f32:
//   A[i][k] = lane (i + 16 k), B[k][j] = lane (j + 16 k)  ->  for A^T A both operands are the SAME register,
//   lane l = (column c = l & 7, residual group grp = (l >> 3) & 1, k = l >> 4): columns 0-7 of the 16 carry one set of
//   four residuals, columns 8-15 another, so one instruction contracts 8 residuals and the two 8x8 diagonal blocks of
//   the 16x16 result are two partial sums of [J^T J | J^T r] (the off-diagonal blocks are cross terms, discarded).
// Eight instructions per chunk on two interleaved accumulators (the dependent latency is 40 cycles, the issue interval
// 32); after every chunk the float accumulators (16 terms each) are added into double ones, so the float rounding does not
// grow with the number of residuals.  The cost rho/2 is summed in double on the vector ALU.
// The partial that leaves the workgroup has the same 28-double layout as the fp64 kernel's: the consumers do not change.
// ------------------------------------------------------------------------------------------------
typedef float k3_v4f32 __attribute__((ext_vector_type(4)));
#define K3F_TILE_FLOATS 512     /* 8 columns x 64 residuals per wavefront */

template <bool NT>
__device__ __forceinline__ void k3_body_f32(
    const float4* __restrict__ pts_s, const float4* __restrict__ planes, const int* __restrict__ count, const int cap,
    const double* __restrict__ x, const int W, const LidarConst& lc, double* __restrict__ partials, const int kf, const int bx, const int nb,
    float* tile /* [waves][K3F_TILE_FLOATS] */, double* red /* [waves][72] */) {
    const int n = count[kf];
    double q[4], R[9], M[9], t[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k] = x[3 * kf + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = x[3 * W + 4 * kf + k];
    d_q2R(q, R);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            M[i * 3 + j] = R[i * 3 + 0] * lc.RlbT[0 * 3 + j] + R[i * 3 + 1] * lc.RlbT[1 * 3 + j] + R[i * 3 + 2] * lc.RlbT[2 * 3 + j];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* T = tile + wv * K3F_TILE_FLOATS;
    const int nwaves = nb * (GLIO_K3_THREADS / GLIO_WAVE), wid = bx * (GLIO_K3_THREADS / GLIO_WAVE) + wv;
    const int nchunks = (n + 63) >> 6;
    const size_t base = (size_t)kf * cap;
    const float4* __restrict__ P = pts_s + base;
    const float4* __restrict__ Q = planes + base;
    // operand role of this lane: column c7 of the residuals 8 seg .. 8 seg + 7 of the chunk (seg = 2 k + grp)
    const int c7 = lane & 7, seg = ((lane >> 4) << 1) | ((lane >> 3) & 1);
    const float* rd = T + c7 * 64 + seg * 8;
    T[7 * 64 + lane] = 0.0f;                               // padding column, never rewritten
    double acc64[4] = {0.0, 0.0, 0.0, 0.0};
    double cost = 0.0;
    const double a = lc.huber;
    const float af = (float)a;
    // One chunk: residual in double, Jacobian row in float, LDS transpose, eight MFMAs, float partials into the double accumulators.
    auto chunk = [&](const float4 p, const float4 pl, const bool live) {
        // ---- residual in double
        const double cx = (double)p.x - lc.tlb[0], cy = (double)p.y - lc.tlb[1], cz = (double)p.z - lc.tlb[2];
        const double rx = M[0] * cx + M[1] * cy + M[2] * cz;
        const double ry = M[3] * cx + M[4] * cy + M[5] * cz;
        const double rz = M[6] * cx + M[7] * cy + M[8] * cz;
        const double e = (double)pl.x * (rx + t[0]) + (double)pl.y * (ry + t[1]) + (double)pl.z * (rz + t[2]) + (double)pl.w;
        const double r = (double)p.w * e;
        const double ar = fabs(r);
        const bool inl = ar <= a;
        const double rho = inl ? r * r : 2.0 * a * ar - a * a;
        cost += live ? 0.5 * rho : 0.0;
        // ---- Jacobian and loss weight in float
        const float sf = p.w, s2 = 2.0f * sf;
        const float rxf = (float)rx, ryf = (float)ry, rzf = (float)rz, rf = (float)r, arf = (float)ar;
        float sw = inl ? 1.0f : __builtin_amdgcn_sqrtf(af * __builtin_amdgcn_rcpf(arf));        // sqrt(rho')
        sw = live ? sw : 0.0f;
        const float ws = sw * sf, ws2 = sw * s2;
        T[0 * 64 + lane] = ws * pl.x;
        T[1 * 64 + lane] = ws * pl.y;
        T[2 * 64 + lane] = ws * pl.z;
        T[3 * 64 + lane] = ws2 * (ryf * pl.z - rzf * pl.y);
        T[4 * 64 + lane] = ws2 * (rzf * pl.x - rxf * pl.z);
        T[5 * 64 + lane] = ws2 * (rxf * pl.y - ryf * pl.x);
        T[6 * 64 + lane] = sw * rf;
        GLIO_WAVE_LDS_SYNC();
        const k3_v4f32 a0 = *reinterpret_cast<const k3_v4f32*>(rd), a1 = *reinterpret_cast<const k3_v4f32*>(rd + 4);
        GLIO_WAVE_LDS_SYNC();                              // the tile is in registers: the next chunk may overwrite it
        k3_v4f32 c0 = {0.0f, 0.0f, 0.0f, 0.0f}, c1 = {0.0f, 0.0f, 0.0f, 0.0f};
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, a0.x, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, a0.y, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, a0.z, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, a0.w, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, a1.x, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, a1.y, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, a1.z, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, a1.w, c1, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc64[k] += (double)c0[k] + (double)c1[k];
    };
    // Two chunks per trip, and the loads of the NEXT two chunks (four 16-byte loads per lane) issued before this trip's arithmetic: with one chunk
    // ahead a wavefront had 2 KB of the stream in flight, a CU ~57 KB -- a third of what the fp64 kernel keeps in flight, and the 32 B / residual
    // form lost the bandwidth race it exists to win (r03: 101 us vs 90 us at the C5 shape).
    int ch = wid;
    float4 pA = make_float4(0, 0, 0, 0), plA = pA, pB = pA, plB = pA;
    bool haveA = ch < nchunks, haveB = ch + nwaves < nchunks;
    if (haveA) { const int i0 = min(ch * 64 + lane, n - 1); pA = k3_load4<NT>(P + i0); plA = k3_load4<NT>(Q + i0); }
    if (haveB) { const int i0 = min((ch + nwaves) * 64 + lane, n - 1); pB = k3_load4<NT>(P + i0); plB = k3_load4<NT>(Q + i0); }
    while (haveA) {
        const int chA2 = ch + 2 * nwaves, chB2 = ch + 3 * nwaves;
        const bool moreA = chA2 < nchunks, moreB = chB2 < nchunks;
        float4 pA2 = pA, plA2 = plA, pB2 = pB, plB2 = plB;
        if (moreA) { const int i1 = min(chA2 * 64 + lane, n - 1); pA2 = k3_load4<NT>(P + i1); plA2 = k3_load4<NT>(Q + i1); }
        if (moreB) { const int i1 = min(chB2 * 64 + lane, n - 1); pB2 = k3_load4<NT>(P + i1); plB2 = k3_load4<NT>(Q + i1); }
        chunk(pA, plA, ch * 64 + lane < n);
        if (haveB) chunk(pB, plB, (ch + nwaves) * 64 + lane < n);
        ch = chA2; pA = pA2; plA = plA2; pB = pB2; plB = plB2; haveA = moreA; haveB = moreB;
    }
    // ---- C layout: lane l, register k -> row 4 (l >> 4) + k, column l & 15.  Block 0 = rows, columns 0..7 (lanes with
    // l >> 4 < 2, l & 15 < 8), block 1 = rows, columns 8..15 = the same entries 40 lanes further on.
    double* myred = red + wv * 72;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double other = __shfl(acc64[k], (lane + 40) & 63, 64);
        if ((lane >> 4) < 2 && (lane & 15) < 8) myred[(4 * (lane >> 4) + k) * 8 + (lane & 7)] = acc64[k] + other;
    }
    cost = wave_sum(cost);
    if (lane == 0) myred[64] = cost;
    __syncthreads();
    if (threadIdx.x < GLIO_LIDAR_ACC) {
        // packed upper triangle (i <= j < 6) -> (i, j); 21..26 -> row 6 (J^T r); 27 -> cost
        const int ti[21] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5};
        const int tj[21] = {0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 4, 5, 5};
        const int k = threadIdx.x;
        const int idx = k < 21 ? ti[k] * 8 + tj[k] : (k < 27 ? 6 * 8 + (k - 21) : 64);
        double v = red[idx];
#pragma unroll
        for (int w2 = 1; w2 < GLIO_K3_THREADS / GLIO_WAVE; ++w2) v += red[w2 * 72 + idx];
        partials[((size_t)kf * nb + bx) * GLIO_LIDAR_ACC + k] = v;
    }
}

// host side: R(q_lb)^T, t_lb, Huber width of the context
LidarConst glio_lidar_const(const glio_ctx* c);
