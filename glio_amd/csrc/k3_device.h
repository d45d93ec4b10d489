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
    const double w = inl ? 1.0 : a / ar;                  // rho'
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
    double v16[16], v8[8], v4[4], v2[2], v1;
    {
        const bool hi = (lane & 32) != 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const double a = acc[2 * i], b = (2 * i + 1 < GLIO_LIDAR_ACC) ? acc[2 * i + 1] : 0.0;
            const double keep = hi ? b : a, send = hi ? a : b;
            v16[i] = keep + __shfl_xor(send, 32, 64);
        }
    }
    {
        const bool hi = (lane & 16) != 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double keep = hi ? v16[2 * i + 1] : v16[2 * i], send = hi ? v16[2 * i] : v16[2 * i + 1];
            v8[i] = keep + __shfl_xor(send, 16, 64);
        }
    }
    {
        const bool hi = (lane & 8) != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double keep = hi ? v8[2 * i + 1] : v8[2 * i], send = hi ? v8[2 * i] : v8[2 * i + 1];
            v4[i] = keep + __shfl_xor(send, 8, 64);
        }
    }
    {
        const bool hi = (lane & 4) != 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const double keep = hi ? v4[2 * i + 1] : v4[2 * i], send = hi ? v4[2 * i] : v4[2 * i + 1];
            v2[i] = keep + __shfl_xor(send, 4, 64);
        }
    }
    {
        const bool hi = (lane & 2) != 0;
        const double keep = hi ? v2[1] : v2[0], send = hi ? v2[0] : v2[1];
        v1 = keep + __shfl_xor(send, 2, 64);
    }
    v1 += __shfl_xor(v1, 1, 64);
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


// host side: R(q_lb)^T, t_lb, Huber width of the context
LidarConst glio_lidar_const(const glio_ctx* c);
