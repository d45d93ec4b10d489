// lidar_kernels.hip -- K3: LiDAR point-to-plane residual + 1x6 local Jacobian + Huber + per-keyframe
// J^T J / J^T r / cost reduction, hand-written for gfx950.
//
// Replaces, per linearisation, the Ceres evaluation of every
//   AutoDiffCostFunction<LidarPlaneNormFactor,1,3,4> + HuberLoss(1.0) + QuaternionParameterization
// residual block (reference: GLIO/include/factors/LidarKeyframeFactor.h:87-103, created at
// GLIO/src/Estimator.cpp:2226-2242) and the normal-equation build that follows inside ceres::Solve.
//
// Math (fused form of the chain global-Jacobian -> loss corrector -> local parameterisation):
//   p_b = R_lb^T (cp - t_lb) ; rp = R(q) p_b ; p_w = rp + t
//   r   = s (n^.p_w + d^)                               s = score, (n^,d^) = weighted plane
//   J   = s [ n^ , 2 (rp x n^) ]                        (1x6: dt, dtheta under Ceres' left (+))
//   rho'(r^2) = 1 (|r|<=a) or a/|r|  -> H += rho' J^T J, g += rho' J^T r, cost += rho/2
//
// Roofline: HBM-bound streaming reduction, 40 B per residual (float4 point + float4 plane + f64 score),
// ~120 fp64 flop per residual (3 flop/B << fp64 ridge).  One wavefront touches exactly one keyframe
// (arrays are keyframe-major), so the reduction is a pure 64-lane shuffle butterfly -> LDS across the 4
// waves -> one 28-double partial per workgroup, summed in fixed order downstream (deterministic, no
// atomics).
#include "k3_device.h"

template <int UNROLL, bool MARG, bool NT = false, bool PIPE = false>
__global__ __launch_bounds__(GLIO_K3_THREADS) void k_lidar_linearize(
    const float4* __restrict__ pts, const float4* __restrict__ planes, const double* __restrict__ scores,
    const int* __restrict__ count, const int cap, const double* __restrict__ x0, const double* __restrict__ x1,
    const SolverStatus* __restrict__ st, const int use_status, const int fixed_which, const int W,
    const LidarConst lc, double* __restrict__ partials, const size_t pstride) {
    int which = fixed_which;
    if (use_status) {
        if (st->done || !st->cand_pending) return;
        which = 1 - st->cur;
    }
    // the partials are double buffered like every other factor block: buffer `which` belongs to the point being linearised
    k3_body<UNROLL, MARG, NT, PIPE>(pts, planes, scores, count, cap, which ? x1 : x0, W, lc, partials + (size_t)which * pstride, blockIdx.y, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------------------------
// K3, LDS-DMA form.  The three streams go global -> LDS with `global_load_lds_dwordx4` (no VGPR staging), each
// wavefront owning a private ring of K3D_STAGES chunks of 64 residuals (1 KiB points + 1 KiB planes + 512 B scores),
// so a CU keeps (waves x (stages - 1) x 2.5 KiB) of reads in flight regardless of register pressure while the fp64
// arithmetic of the chunk that has landed runs.  Chunks are dealt round-robin to the waves of a keyframe.  The
// counter discipline is manual: 3 VMEM operations per chunk, `s_waitcnt vmcnt(3 * chunks still in flight behind
// this one)` before the ds_reads of a slot, and the slot is re-armed only after those reads have returned.
// Needs cap % 64 == 0 (16 B aligned score rows, no clamping of the last chunk); other shapes take the register path.
// ------------------------------------------------------------------------------------------------
#define K3D_CHUNK_BYTES 2560
typedef __attribute__((address_space(1))) const void* k3_gptr;
typedef __attribute__((address_space(3))) void* k3_lptr;
template <int N> __device__ __forceinline__ void k3_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int STAGES, bool DRAIN = false>
__global__ __launch_bounds__(GLIO_K3_THREADS) void k_lidar_linearize_dma(
    const float4* __restrict__ pts, const float4* __restrict__ planes, const double* __restrict__ scores,
    const int* __restrict__ count, const int cap, const double* __restrict__ x0, const double* __restrict__ x1,
    const SolverStatus* __restrict__ st, const int use_status, const int fixed_which, const int W,
    const LidarConst lc, double* __restrict__ partials_base, const size_t pstride) {
    __shared__ __attribute__((aligned(16))) char ring[GLIO_K3_THREADS / GLIO_WAVE][STAGES][K3D_CHUNK_BYTES];
    int which = fixed_which;
    if (use_status) {
        if (st->done || !st->cand_pending) return;
        which = 1 - st->cur;
    }
    const double* __restrict__ x = which ? x1 : x0;
    double* __restrict__ partials = partials_base + (size_t)which * pstride;
    const int kf = blockIdx.y, nb = gridDim.x, n = count[kf];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);    // scalar: the waits below branch on it
    const int nwaves = nb * (GLIO_K3_THREADS / GLIO_WAVE), wid = blockIdx.x * (GLIO_K3_THREADS / GLIO_WAVE) + wv;
    const int nchunks = (n + 63) >> 6;
    const int mine = wid < nchunks ? (nchunks - wid + nwaves - 1) / nwaves : 0;     // chunks wid, wid + nwaves, ...
    const size_t base = (size_t)kf * cap;
    const float4* __restrict__ P = pts + base;
    const float4* __restrict__ Q = planes + base;
    const double* __restrict__ S = scores + base;
    char* my = &ring[wv][0][0];
    auto issue = [&](const int k) {           // chunk number k of this wave -> slot k % STAGES
        const int c0 = (wid + k * nwaves) << 6;
        char* dst = my + (k % STAGES) * K3D_CHUNK_BYTES;
        __builtin_amdgcn_global_load_lds((k3_gptr)(P + c0 + lane), (k3_lptr)dst, 16, 0, 2);
        __builtin_amdgcn_global_load_lds((k3_gptr)(Q + c0 + lane), (k3_lptr)(dst + 1024), 16, 0, 2);
        if (lane < 32) __builtin_amdgcn_global_load_lds((k3_gptr)(S + c0 + 2 * lane), (k3_lptr)(dst + 2048), 16, 0, 2);
    };
    // per-keyframe constants: the pose loads go out first, the ring's prologue right behind them
    double q[4], R[9], M[9], t[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k] = x[3 * kf + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = x[3 * W + 4 * kf + k];
#pragma unroll
    for (int k = 0; k < STAGES; ++k) if (k < mine) issue(k);
    d_q2R(q, R);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            M[i * 3 + j] = R[i * 3 + 0] * lc.RlbT[0 * 3 + j] + R[i * 3 + 1] * lc.RlbT[1 * 3 + j] + R[i * 3 + 2] * lc.RlbT[2 * 3 + j];
    double acc[GLIO_LIDAR_ACC];
#pragma unroll
    for (int k = 0; k < GLIO_LIDAR_ACC; ++k) acc[k] = 0.0;

    for (int k = 0; k < mine; ++k) {
        const int behind = min(STAGES - 1, mine - 1 - k);        // chunks issued after this one and still in flight
        if (DRAIN) k3_wait_vm<0>();
        else if (behind >= 3 && STAGES > 3) k3_wait_vm<9>();
        else if (behind == 2 && STAGES > 2) k3_wait_vm<6>();
        else if (behind == 1) k3_wait_vm<3>();
        else k3_wait_vm<0>();
        // ds_reads as inline asm: for a C++ LDS load that may alias an LDS-DMA destination the compiler inserts
        // s_waitcnt vmcnt(0) (it cannot count the ring), which would serialise the whole pipeline
        const unsigned slot = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(my + (k % STAGES) * K3D_CHUNK_BYTES);
        k3_f4 pv, plv;
        double sc;
        asm volatile("ds_read_b128 %0, %1" : "=v"(pv) : "v"(slot + 16u * lane) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(plv) : "v"(slot + 16u * lane) : "memory");
        asm volatile("ds_read_b64 %0, %1 offset:2048" : "=v"(sc) : "v"(slot + 8u * lane) : "memory");
        // the wait names the three results as read-write operands: the compiler sees them live until here, so it can
        // neither reuse a dead component (p.w) while the read is in flight nor hoist a conversion above the wait
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pv), "+v"(plv), "+v"(sc) : : "memory");   // slot is in registers: it may be re-armed
        const float4 p = make_float4(pv.x, pv.y, pv.z, pv.w), pl = make_float4(plv.x, plv.y, plv.z, plv.w);
        if (k + STAGES < mine) issue(k + STAGES);
        if (((wid + k * nwaves) << 6) + lane < n) lidar_accumulate<false>(p, pl, sc, M, t, lc.tlb, lc.huber, acc, lc.RlbT, q);
    }
    k3_reduce_store(acc, partials, kf, blockIdx.x, nb);
}

// ------------------------------------------------------------------------------------------------
// K3, fp32-Jacobian / MFMA form (k3_body_f32 in k3_device.h) as its own launch, and the packing of its 32 B/residual
// input: point.w <- (float) score.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GLIO_K3_THREADS) void k_lidar_linearize_f32(
    const float4* __restrict__ pts_s, const float4* __restrict__ planes, const int* __restrict__ count, const int cap,
    const double* __restrict__ x0, const double* __restrict__ x1, const SolverStatus* __restrict__ st, const int use_status,
    const int fixed_which, const int W, const LidarConst lc, double* __restrict__ partials, const size_t pstride) {
    __shared__ __attribute__((aligned(16))) float tile[(GLIO_K3_THREADS / GLIO_WAVE) * K3F_TILE_FLOATS];
    __shared__ double red[(GLIO_K3_THREADS / GLIO_WAVE) * 72];
    int which = fixed_which;
    if (use_status) {
        if (st->done || !st->cand_pending) return;
        which = 1 - st->cur;
    }
    k3_body_f32<true>(pts_s, planes, count, cap, which ? x1 : x0, W, lc, partials + (size_t)which * pstride, blockIdx.y, blockIdx.x, gridDim.x, tile, red);
}
__global__ __launch_bounds__(256) void k_pack_points_f32(const float4* __restrict__ pts, const double* __restrict__ scores,
                                                         const int* __restrict__ count, const int cap, float4* __restrict__ out) {
    const int kf = blockIdx.y, n = count[kf];
    const size_t base = (size_t)kf * cap;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float4 p = pts[base + i];
        p.w = (float)scores[base + i];
        out[base + i] = p;
    }
}
// (re)build the packed points when the correspondences changed since the last linearisation
void glio_lidar_pack_f32(glio_ctx* c) {
    if (c->opts.lidar_precision != GLIO_LIDAR_F32_MFMA || !c->f32_dirty) return;
    int blocks = (c->cap + 255) / 256;
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(k_pack_points_f32, dim3(blocks, c->W), dim3(256), 0, c->stream, c->d_pts, c->d_scores, c->d_count, c->cap, c->d_pts_s);
    c->f32_dirty = 0;
}
// read-only ceiling of the 32 B/residual stream (same grid as the f32 kernel)
__global__ __launch_bounds__(GLIO_K3_THREADS) void k_stream_read32(const float4* __restrict__ pts, const float4* __restrict__ planes,
                                                                   const int* __restrict__ count, const int cap, double* __restrict__ out) {
    const int kf = blockIdx.y, n = count[kf];
    const size_t base = (size_t)kf * cap;
    const int stride = gridDim.x * GLIO_K3_THREADS;
    float acc = 0.0f;
    int i = blockIdx.x * GLIO_K3_THREADS + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        float4 p[4], q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { p[u] = k3_load4<true>(pts + base + i + u * stride); q[u] = k3_load4<true>(planes + base + i + u * stride); }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += (p[u].x + p[u].y + p[u].z + p[u].w) + (q[u].x + q[u].y + q[u].z + q[u].w);
    }
    for (; i < n; i += stride) { const float4 p = pts[base + i], q = planes[base + i]; acc += (p.x + p.y + p.z + p.w) + (q.x + q.y + q.z + q.w); }
    double a = wave_sum((double)acc);
    if ((threadIdx.x & 63) == 0) out[((size_t)kf * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)] = a;
}

// Practical ceiling for K3: the same three streams (16 + 16 + 8 B per residual), same grid, same non-temporal loads,
// but only a trivial sum -- what the memory system delivers for a launch of this size (bench.py reports it next to K3).
__global__ __launch_bounds__(GLIO_K3_THREADS) void k_stream_read(const float4* __restrict__ pts, const float4* __restrict__ planes,
                                                                 const double* __restrict__ scores, const int* __restrict__ count,
                                                                 const int cap, double* __restrict__ out) {
    const int kf = blockIdx.y, n = count[kf];
    const size_t base = (size_t)kf * cap;
    const int stride = gridDim.x * GLIO_K3_THREADS;
    double acc = 0.0;
    int i = blockIdx.x * GLIO_K3_THREADS + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        float4 p[4], q[4]; double s[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { p[u] = k3_load4<true>(pts + base + i + u * stride); q[u] = k3_load4<true>(planes + base + i + u * stride); s[u] = k3_load1<true>(scores + base + i + u * stride); }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += (double)(p[u].x + p[u].y + p[u].z + p[u].w) + (double)(q[u].x + q[u].y + q[u].z + q[u].w) + s[u];
    }
    for (; i < n; i += stride) { const float4 p = pts[base + i], q = planes[base + i]; acc += (double)(p.x + p.y + p.z + p.w) + (double)(q.x + q.y + q.z + q.w) + scores[base + i]; }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) out[((size_t)kf * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)] = acc;
}
void glio_launch_stream_read(glio_ctx* c) {
    if (c->opts.lidar_precision == GLIO_LIDAR_F32_MFMA) {
        hipLaunchKernelGGL(k_stream_read32, dim3(c->k3_bpk, c->W), dim3(GLIO_K3_THREADS), 0, c->stream, c->d_pts_s, c->d_planes, c->d_count,
                           c->cap, c->d_lidar_partials);
        return;
    }
    hipLaunchKernelGGL(k_stream_read, dim3(c->k3_bpk, c->W), dim3(GLIO_K3_THREADS), 0, c->stream, c->d_pts, c->d_planes, c->d_scores, c->d_count,
                       c->cap, c->d_lidar_partials);
}

LidarConst glio_lidar_const(const glio_ctx* c) {
    LidarConst lc;
    // R(q_lb)^T via Eigen's inverse(): conj / |q|^2
    const double* ql = c->opts.q_lb;
    const double n2 = ql[0] * ql[0] + ql[1] * ql[1] + ql[2] * ql[2] + ql[3] * ql[3];
    const double w = ql[0] / n2, x = -ql[1] / n2, y = -ql[2] / n2, z = -ql[3] / n2;
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    lc.RlbT[0] = 1 - (tyy + tzz); lc.RlbT[1] = txy - twz; lc.RlbT[2] = txz + twy;
    lc.RlbT[3] = txy + twz; lc.RlbT[4] = 1 - (txx + tzz); lc.RlbT[5] = tyz - twx;
    lc.RlbT[6] = txz - twy; lc.RlbT[7] = tyz + twx; lc.RlbT[8] = 1 - (txx + tyy);
    for (int k = 0; k < 3; ++k) lc.tlb[k] = c->opts.t_lb[k];
    lc.huber = c->opts.huber_delta;
    return lc;
}

void glio_launch_lidar_linearize(glio_ctx* c, int use_status_cand, int which, int marg) {
    const LidarConst lc = glio_lidar_const(c);
    c->last_k3_nb = c->k3_bpk;
    dim3 grid(c->k3_bpk, c->W);
#define K3_LAUNCH_(U, MG, ...) hipLaunchKernelGGL((k_lidar_linearize<U, MG, ##__VA_ARGS__>), grid, dim3(GLIO_K3_THREADS), 0, c->stream, \
                       c->d_pts, c->d_planes, c->d_scores, c->d_count, c->cap, c->d_x[0], c->d_x[1], \
                       c->d_status, use_status_cand, which, c->W, lc, c->d_lidar_partials, glio_partials_stride(c))
    if (marg) { K3_LAUNCH_(4, true); return; }          // the marginalization keeps the fp64 form (its Jacobian convention differs, Q8)
    if (c->opts.lidar_precision == GLIO_LIDAR_F32_MFMA) {
        glio_lidar_pack_f32(c);
        hipLaunchKernelGGL(k_lidar_linearize_f32, grid, dim3(GLIO_K3_THREADS), 0, c->stream, c->d_pts_s, c->d_planes, c->d_count, c->cap,
                           c->d_x[0], c->d_x[1], c->d_status, use_status_cand, which, c->W, lc, c->d_lidar_partials, glio_partials_stride(c));
        return;
    }
#define K3_DMA_(...) hipLaunchKernelGGL((k_lidar_linearize_dma<__VA_ARGS__>), grid, dim3(GLIO_K3_THREADS), 0, c->stream, \
                       c->d_pts, c->d_planes, c->d_scores, c->d_count, c->cap, c->d_x[0], c->d_x[1], \
                       c->d_status, use_status_cand, which, c->W, lc, c->d_lidar_partials, glio_partials_stride(c))
    if ((c->cap & 63) == 0) {
        if (c->k3_unroll == 32) { K3_DMA_(2); return; }
        if (c->k3_unroll == 33) { K3_DMA_(3); return; }
        if (c->k3_unroll == 34) { K3_DMA_(4); return; }
        if (c->k3_unroll == 35) { K3_DMA_(3, true); return; }
    }
#undef K3_DMA_
    switch (c->k3_unroll >= 32 && c->k3_unroll <= 35 ? 22 : c->k3_unroll) {
        case 1: K3_LAUNCH_(1, false); break;
        case 2: K3_LAUNCH_(2, false); break;
        case 8: K3_LAUNCH_(8, false); break;
        case 12: K3_LAUNCH_(2, false, true); break;
        case 14: K3_LAUNCH_(4, false, true); break;
        case 18: K3_LAUNCH_(8, false, true); break;
        case 21: K3_LAUNCH_(1, false, true, true); break;
        case 22: K3_LAUNCH_(2, false, true, true); break;
        case 24: K3_LAUNCH_(4, false, true, true); break;
        default: K3_LAUNCH_(4, false); break;
    }
#undef K3_LAUNCH_
}
