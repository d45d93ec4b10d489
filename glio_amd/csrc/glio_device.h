// glio_device.h -- shared declarations of the gfx950 implementation: context layout, device-side
// small math, launch wrappers between translation units.  HIP only; never included by host callers
// (they see include/glio_hip.h).
#pragma once
#include <chrono>

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/glio_hip.h"

// Debug allocator (GLIO_DEBUG_POISON_ALLOC=1): every device allocation of the library is filled with 0xFF bytes (NaN as a double, -1 as an
// int) before it is handed out.  Fresh device memory of a process that runs alone is zero, so a kernel that READS a buffer it was never
// given -- and silently relies on the zero -- passes every single-process test; with several processes on one GPU the pages may come back
// with another process's data (found in round 3: `contention_repro.py`).  With the poison such a read fails loudly and reproducibly.
#include <cstdlib>
static inline hipError_t glio_dbg_malloc(void** p, size_t bytes) {
    static const int poison = getenv("GLIO_DEBUG_POISON_ALLOC") ? atoi(getenv("GLIO_DEBUG_POISON_ALLOC")) : 0;
    const hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess && poison && bytes > 0) { (void)hipMemset(*p, 0xFF, bytes); (void)hipDeviceSynchronize(); }
    return e;
}
template <typename T> static inline hipError_t glio_dbg_malloc(T** p, size_t bytes) { return glio_dbg_malloc(reinterpret_cast<void**>(p), bytes); }
#define hipMalloc(p, bytes) glio_dbg_malloc((p), (bytes))
// hipMemset (the synchronous-looking form) is enqueued on the NULL stream: asynchronous for the host and NOT ordered with this library's
// non-blocking streams.  Every use here initialises a buffer at object creation that kernels on such a stream read soon after; with several
// processes on the GPU the fill could still be pending when they ran (round 3, scripts/contention_more.py: 1 % of the first associations of a fresh
// context lost a third of their queries -- the presort's hash table was cleared AFTER it had been filled).  The wrapper waits for the fill.
static inline hipError_t glio_memset_done(void* p, int value, size_t bytes) {
    const hipError_t e = hipMemset(p, value, bytes);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(nullptr);
}
#define hipMemset(p, value, bytes) glio_memset_done((p), (value), (bytes))

#define GLIO_WAVE 64

// ------------------------------------------------------------------------------------------------
// device-resident layouts
// ------------------------------------------------------------------------------------------------
// Per-keyframe LiDAR accumulator: 21 (upper triangle of the 6x6 J^T J) + 6 (J^T r) + 1 (cost)
#define GLIO_LIDAR_ACC 28
// K3 launch geometry: GLIO_K3_BLOCKS_PER_KF workgroups of GLIO_K3_THREADS per keyframe slot
#define GLIO_K3_THREADS 256
#define GLIO_K3_BLOCKS_PER_KF 32       /* default; ctx->k3_bpk is the live value */
#define GLIO_K3_MAX_BLOCKS_PER_KF 256

// IMU edge, device form (pre-digested on upload: sqrt_info is LLT(cov^-1).L^T, ImuFactor.h:44-45)
struct ImuEdgeDev {
    double delta_p[3], delta_q[4], delta_v[3], lin_ba[3], lin_bg[3];
    double sum_dt;
    double dp_dba[9], dp_dbg[9], dq_dbg[9], dv_dba[9], dv_dbg[9];
    double sqrt_info[225];
    int slot_i;
    int pad_;
};
// dense local block of one IMU edge / GNSS group over [slot_a(15) | slot_b(15)]
#define GLIO_PAIR_DIM 30
struct PairBlock {
    double H[GLIO_PAIR_DIM * GLIO_PAIR_DIM];
    double g[GLIO_PAIR_DIM];
    double cost;
    int slot_a, slot_b;
};
// GNSS group = all DD / Doppler factors that share one (slot_i, slot_j) pair
struct GnssGroup {
    int slot_i, slot_j;
    int dd_begin, dd_end;      // range in the sorted glio_dd_psr array
    int dop_begin, dop_end;    // range in the sorted glio_doppler array (sorted by epoch inside)
    int run_begin, run_end;    // range in the DopRun array
};
// Doppler rows of one epoch inside a group (contiguous in the sorted glio_doppler array)
struct DopRun { int begin, end, epoch, group; };
struct GnssDevExtra { DopRun* d_runs; int n_runs; int* d_prior_colblk; };
// per-epoch Doppler coupling with the 12 pose variables (t_i v_i t_j v_j) of its group
struct DdtBlock {
    double c[12];
    double h, g;
    int group;
    int used;
};

// Trust-region state machine, lives in device memory for the whole solve (no host round trips)
struct SolverStatus {
    int done;
    int termination;
    int iteration;
    int successful;
    int cur;              // index (0/1) of the buffers holding the current point x, H, g, cost
    int phase;            // 0 = first evaluation pending, 1 = running
    int reuse;
    int invalid;
    int n_ddt;
    int cand_pending;     // 1 = the buffers 1-cur hold a point whose linearisation is wanted / available
    double radius, mu;
    double cost, model_cost_change;
    double alpha, dogleg_step_norm;
    double initial_cost, grad_max_norm;
    double mu_used;       // mu of the factorisation behind the stored Gauss-Newton step
    int group;            // number of trust-region kernel groups started (k_tr_prepare launches) in this solve
    int lin_fail;         // Levenberg-Marquardt: the linear solve of this iteration failed -> the step is invalid
    double decrease_factor;   // LevenbergMarquardtStrategy::decrease_factor_
    int solve_id;             // tag of this solve in the host-mapped progress words (kernels of an earlier solve may still be draining)
    int pad_;
    unsigned long long checksum;   // of the published result (status words + state), glio_result_mix: the host accepts the mapped copy only when it adds up
};
// Publication of the result in host-mapped memory.  The kernel that ends a solve writes [status | state] and then the solve's tag; the
// host polls the tag.  Round 3 found (N processes sharing one GPU, scripts/contention_loop.py) that the tag can become visible BEFORE
// parts of the payload -- whole keyframes of the state still zero, `done` still 0 -- although every writer executes a system-scope
// fence before the tag is written: writes to host memory are not delivered in order under load.  (It only ever showed on the FIRST solve of
// a context: later solves of the same window find the previous, identical result in the buffer.)  So the payload carries a checksum: every
// 8-byte word enters with a position-dependent odd multiplier, the host recomputes it over what it read and keeps re-reading until it
// adds up (a torn or stale payload passes with probability 2^-64); after 2 ms without a match it falls back to a stream-ordered copy.
__host__ __device__ inline unsigned long long glio_result_mix(const unsigned long long word, const unsigned long long k) {
    return (word + k) * (0x9E3779B97F4A7C15ull + 2ull * k);
}

// Structured ("arrow") linear solver of the trust-region step (solver_kernels.hip): buffers + structure tables
struct ArrowDev {
    int mode;                 // 1 = use when the structure permits (default), 0 = dense factorisation only
    int gnss_ok, prior_ok;    // structure checks made when the factors are set
    int gnss_chain, prior_chain;   // stronger: EVERY GNSS pair couples neighbouring keyframes / the prior is block diagonal by keyframe
    int max_epoch;            // largest clock-drift epoch referenced by a Doppler factor (-1: none)
    int last_path;            // factorisation the last trust-region step was enqueued with: 0 dense, 1 arrow, 2 keyframe chain
    int last_fronts;          // elimination fronts of the last k_chain_step launch (2 or 4), 0 = none yet
    int2* d_ep_slots;         // [n_ddt_max] keyframe slots (lo, hi) coupled by clock-drift epoch e, (-1,-1) if unused
    int* d_ep_off;            // [W+1] CSR over slots: epochs touching slot i
    int* d_ep_list;           // [2 n_ddt_max]
    double* d_Y;              // (n_ddt + 9W) x (6W+2): L^-1 [M_ep | b_e]
    double* d_blk;            // [W][544] staged keyframe blocks of k_chain_solve<true> (windows whose blocks do not fit the LDS)
    double* d_Lblk;           // [W][190] factored speed-bias chain: L_ii (9x9), L_{i+1,i} (9x9), row stride 10, 9 reciprocal pivots
    double* d_Sp;             // (6W+1) x 6W pose Schur complement + carried right-hand side
    double* d_z;              // [n] solution in elimination order
    int* d_flag;              // 0 running, 1 breakdown, 2 solved
    double* d_chain_sum;      // [W][GLIO_CS_STRIDE] k_chain_step's helper workgroups: the candidate's block entries summed over their six sources
    int* d_chain_done;        // [W] their completion words: 2 * sequence number + (produced ? 1 : 0)
    int chain_seq;            // sequence number of the next k_chain_step launch (kernel argument)
    double* d_fat_ep;         // [n_ddt_max][34] the fat helpers' epoch products (eliminated column V, 1 / sqrt(m), y, t); their blocks go to d_blk
    long long* d_dbg;         // [64] wall-clock stamps of the last launch (100 MHz), development aid
};

// Chain-layout contributions ("slices"): every factor role also writes what it contributes to the block-tridiagonal part of H
// straight into the layout k_chain_step factors, one slice of GLIO_CS_STRIDE doubles per (keyframe i, source):
//   entry w < 120: D_i[r][c], c <= r, w = r (r + 1) / 2 + c;  w >= 120: B_i[lr][c] = H[15 (i+1) + lr][15 i + c], w = 120 + 15 lr + c
//   sources: 0 IMU edge (i, i+1) [aa -> D_i, ba -> B_i], 1 IMU edge (i-1, i) [bb -> D_i], 2 / 3 first / second GNSS group
//   touching i (ascending group index), 4 prior.  Slices of absent sources stay zero (re-zeroed when the structure changes).
#define GLIO_CS_STRIDE 352
#define GLIO_CS_SOURCES 5
struct ChainKf { short e0, e1, k0, k1, o0, o1, kp, pad_; };   // per keyframe: IMU edges, GNSS groups (+ whether i is their slot_a), group of (i, i+1)

// A host cloud whose points are records of `stride` bytes: x, y, z as three floats at offset 0, the intensity as a float at `ioff` (pcl::PointXYZI: 32 / 16).
// The raw records are uploaded in ONE copy into a grow-only staging buffer and unpacked to float4 on the device (glio_upload_points, capi.hip).
struct GlioRawStage { void* d; size_t cap; };
extern "C" {       // (defined inside capi.hip's extern "C" block; internal: not part of include/glio_hip.h)
int glio_point_layout_ok(int stride, int ioff);
int glio_upload_points(hipStream_t stream, GlioRawStage* st, const void* host, int n, int stride, int ioff, float4* d_out);
}
struct glio_ctx {
    GlioRawStage raw_stage;       // staging of strided point input (glio_set_scan_strided, glio_set_map_strided, glio_localmap_push_strided)
    std::chrono::steady_clock::time_point solve_t0;   // start of the solve in flight (max_solver_time_s is watched in the enqueue and in the wait loop)
    glio_opts opts;
    int device;
    hipStream_t own_stream, stream;
    int W, cap, n_max, n_ddt_max;
    // ---- LiDAR correspondences [W][cap]
    float4* d_pts;
    float4* d_planes;
    double* d_scores;
    float4* d_pts_s;              // [W][cap] points with the score as a float in .w (opts.lidar_precision = GLIO_LIDAR_F32_MFMA only)
    int f32_dirty;                // the correspondences changed since d_pts_s was packed
    int* d_count;                 // [W]
    int h_count[GLIO_MAX_WINDOW];
    // ---- scans + map (association)
    float4* d_scan;               // [W][cap] resident scans, a RING: window slot s lives in row (scan_base + s) % W (glio_slide_window advances scan_base)
    int scan_base;
    int h_scan_count[GLIO_MAX_WINDOW];
    float4* d_map_sorted;         // [max_map] sorted by cell, .w = original index bits
    int map_n;
    struct AssocWork* assoc;      // hash table etc. (assoc_kernels.hip)
    struct LocalMap* localmap;    // device-resident keyframe ring + voxel grid (localmap_kernels.hip), created on demand
    // ---- small factors
    ImuEdgeDev* d_imu; int n_imu;
    PairBlock* d_imu_blocks;      // [2][W]
    glio_dd_psr* d_dd; int n_dd;
    glio_doppler* d_dop; int n_dop;
    size_t dd_cap, dop_cap;       // grow-only capacities (elements) of d_dd / d_dop
    char* h_stage; size_t h_stage_cap, h_stage_used; char* d_stage; size_t h_stage_top; int n_stage_seg; struct { void* dst; const void* src; size_t bytes; } stage_seg[32];   // pinned upload arena of the factor tables: every table of a set_* call
                                                       // goes through it with asynchronous copies and ONE synchronisation
    GnssGroup* d_groups; int n_groups;
    PairBlock* d_gnss_blocks;     // [2][W*W] (n_groups used)
    DdtBlock* d_ddt_blocks;       // [2][n_ddt_max]
    glio_gnss_frame frame;
    double R_ecef_local[9];       // R_ecef_enu(anchor) * Rz(yaw)
    // prior
    int prior_n, prior_nb;
    double* d_prior_J0;           // [np][np]
    double* d_prior_A0;           // J0^T J0
    double* d_prior_r0;
    double* d_prior_x0;           // [nb][9]
    int* d_prior_slot; int* d_prior_kind; int* d_prior_idx;
    int* d_prior_index;           // [15*W] state index -> prior column or -1
    double* d_prior_H;            // [2][np*np]
    double* d_prior_g;            // [2][np]
    double* d_prior_cost;         // [2]
    double* d_prior_work;         // dx, r, M blocks
    // ---- state + normal equations, double buffered (cur / candidate)
    double* d_x[2];               // layout: trans[3W] quat[4W] sb[9W] ddt[n_ddt_max]
    double* d_xout;
    double* d_lidar_partials;     // [2][W][GLIO_K3_MAX_BLOCKS_PER_KF][28]: double buffered (current point / candidate) like every factor block
    double* d_hdiag[2];           // [n_max] diag(H) of the current point / candidate (keyframe-chain path: no dense H is built)
    double* d_lidar_blocks;       // [2][W][28]
    double* d_H[2];
    double* d_g[2];
    double* d_cost[2];
    // ---- solver workspace
    double* d_L;                  // (n+1) x n factor workspace
    double* d_vec;                // scale, diag, grad, gn, step, delta, tmp ... 10 x vstride
    int vstride;                  // n_max rounded up to 16 doubles: the work vectors do not share 128 B lines (a line read for the tail of
                                  // one vector would otherwise also cache the head of the next, which the same kernel may rewrite)
    SolverStatus* d_status;
    SolverStatus* h_status;       // pinned
    void* d_h_status;             // its device address: k_stage_in pulls status + initial state from it (no blit, no barrier packet behind it)
    volatile int* h_progress;     // pinned + mapped: [0] (solve id << 16) | groups started, [1] id of the solve that is done -- written by the GPU, polled by the host; [2] written by the HOST, read by the GPU: id of the solve whose time budget is spent
    unsigned char* h_result; unsigned char* d_result;   // pinned + mapped: [SolverStatus | pad to 512 B | final state]: the last kernel of a solve
                                                        // publishes its result here, glio_solve reads it without a copy or a stream sync
    int solve_id;
    int* d_progress;              // device alias of h_progress
    int enqueue_lead;             // kernel groups the host keeps queued ahead of the GPU
    double* h_xbuf;               // pinned staging for state upload/download
    hipEvent_t ev0, ev1;
    int have_factors;
    int last_n_ddt;
    int k3_bpk, k3_unroll;        // K3 launch geometry (tunable, glio_debug_set_k3)
    int last_k3_nb;               // partials per keyframe written by the most recent K3 (its consumers sum that many)
    int merged_linearize;         // K3 and the small factors in one launch (k_linearize_all)
    ArrowDev arrow;
    void* extra;                  // CtxExtra (capi.hip): evaluator scratch, GNSS run tables -- owned by the context
    // ---- host mirrors of the factor graph's structure + the gather tables k_chain_step reads (built from them when dirty)
    int h_imu_slot[GLIO_MAX_WINDOW];          // slot_i of IMU edge k
    GnssGroup* h_groups;                      // [W*W] (n_groups used)
    int* h_prior_index;                       // [15 W] state index -> prior column or -1
    short* d_chain_tabs; short* h_chain_tabs; // [8 W + 15 W] ChainKf per keyframe, then the prior index (h_: pinned)
    int chain_tabs_dirty; int chain_tabs_half, chain_tabs_on_device; hipEvent_t ev_tabs[2];      // (h_chain_tabs holds two copies used in turn, each with the event of its last upload)
    int h_band_clean;             // n for which both dense H buffers are zero outside the band a band-only k_assemble writes, else 0 (reset by every full
                                  // assembly and structure change)
    int want_pair_H;              // 0: the linearisation in flight feeds k_chain_step only (chain slices, g, cost): the 30 x 30 pair blocks are not written
    double* d_chain_src;                      // [2][W][GLIO_CS_SOURCES][GLIO_CS_STRIDE] chain-layout contributions, double buffered
    hipEvent_t ev_ext_read; int ext_read_pending;   // another object's stream is still READING the resident scans (glio_bassoc_set_frame_from_scan copies one on the
                                                    // batch association's stream): the next write to a scan row, and glio_destroy, come behind this event
    int prior_device_made;                    // the installed prior is glio_marginalize_keep's own product: block diagonal with EXACT zeros (a caller's prior is only held to a tolerance)
    int n_cu;                                 // compute units of THIS context's device (the helper workgroups of k_chain_step need 2 (1 + W) of them)
};

static inline int glio_x_size(int W, int n_ddt) { return 16 * W + n_ddt; }
static inline size_t glio_partials_stride(const glio_ctx* c) { return (size_t)c->W * GLIO_K3_MAX_BLOCKS_PER_KF * GLIO_LIDAR_ACC; }

// ------------------------------------------------------------------------------------------------
// device math (quaternions are w,x,y,z)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void d_cross(const double a[3], const double b[3], double o[3]) {
    const double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
__device__ __forceinline__ double d_dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
// the same with every product rounded before it is added (no FMA contraction): the GNSS roles difference ranges of ~2.6e7 m, where the
// reference's build (scalar x86, no FMA) and a contracted sum differ by ulps of the RANGE -- orders above the residual's own rounding
__device__ __forceinline__ double d_dot3_nc(const double a[3], const double b[3]) {
#pragma clang fp contract(off)
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
__device__ __forceinline__ void d_qmul(const double a[4], const double b[4], double o[4]) {
    const double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    const double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    const double y = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
    const double z = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
    o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}
__device__ __forceinline__ void d_qinv(const double q[4], double o[4]) {
    const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    o[0] = q[0] / n2; o[1] = -q[1] / n2; o[2] = -q[2] / n2; o[3] = -q[3] / n2;
}
__device__ __forceinline__ void d_qnormalize(double q[4]) {
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
// v + w 2(u x v) + u x 2(u x v)
__device__ __forceinline__ void d_qrot(const double q[4], const double v[3], double o[3]) {
    double uv[3], uuv[3];
    d_cross(q + 1, v, uv);
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    d_cross(q + 1, uv, uuv);
    o[0] = v[0] + q[0] * uv[0] + uuv[0];
    o[1] = v[1] + q[0] * uv[1] + uuv[1];
    o[2] = v[2] + q[0] * uv[2] + uuv[2];
}
__device__ __forceinline__ void d_q2R(const double q[4], double R[9]) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
// Ceres QuaternionParameterization::Plus
__device__ __forceinline__ void d_quat_plus(const double q[4], const double d[3], double o[4]) {
    const double nrm = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (nrm > 0.0) {
        double sn, cs;
        sincos(nrm, &sn, &cs);          // one argument reduction for both
        const double s = sn / nrm;
        const double dq[4] = {cs, s * d[0], s * d[1], s * d[2]};
        d_qmul(dq, q, o);
    } else {
        o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; o[3] = q[3];
    }
}
// d([1,delta] (x) q)/d delta, 4x3 row-major
__device__ __forceinline__ void d_plus_jac(const double q[4], double P[12]) {
    P[0] = -q[1]; P[1] = -q[2]; P[2] = -q[3];
    P[3] = q[0];  P[4] = q[3];  P[5] = -q[2];
    P[6] = -q[3]; P[7] = q[0];  P[8] = q[1];
    P[9] = q[2];  P[10] = -q[1]; P[11] = q[0];
}

// broadcast lane `l` (compile-time constant) of a double through v_readlane_b32 (SALU path, no LDS)
// Ordering point for LDS traffic between the lanes of ONE wavefront: earlier ds_writes are complete and visible, later
// ds_reads are not hoisted above it.  Restricted to the LDS address space on purpose: a generic wavefront-scope fence also
// drains every outstanding GLOBAL load/store (s_waitcnt vmcnt(0)), which would expose the latency of loads issued early
// for prefetching.
static inline int glio_scan_row(const glio_ctx* c, int slot) { return (c->scan_base + slot) % c->W; }
#define GLIO_WAVE_LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local"); __builtin_amdgcn_wave_barrier(); \
                                  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local"); } while (0)

// Workgroup barrier that orders LDS traffic ONLY: s_waitcnt lgkmcnt(0) + s_barrier.  __syncthreads() also drains every outstanding global
// store of the wavefront (s_waitcnt vmcnt(0): on gfx9 stores count in vmcnt), i.e. it waits ~0.5 us for the write acknowledgements from L2 (more
// for host-mapped memory) each time a phase has written results out.  For barriers between phases that hand data over through LDS and whose
// global stores are only read by LATER launches.
#define GLIO_BLOCK_LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); \
                                   __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); } while (0)

__device__ __forceinline__ double readlane_d(double v, int l) {
#ifdef GLIO_NO_READLANE
    return __shfl(v, l, 64);
#endif
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// ------------------------------------------------------------------------------------------------
// Cross-lane exchange for the xor butterflies WITHOUT the LDS crossbar.  __shfl_xor compiles to ds_bpermute_b32 (two per double, ~100+
// cycles of latency each, dependent from stage to stage): a six-stage butterfly costs ~0.4 us, and it ends every workgroup of the K3
// linearisation and every block reduction of the single-workgroup solver kernels.  gfx950 has V_PERMLANE32_SWAP / V_PERMLANE16_SWAP for the
// two stages that cross 16-lane rows, and DPP (row_ror:8, row_shl/shr:4 under bank masks, quad_perm) covers xor 8, 4, 2, 1 inside a row.
// The values that meet are the same as with __shfl_xor, so every sum below is bit-identical to the shuffle form it replaces.
// ------------------------------------------------------------------------------------------------
typedef unsigned glio_v2u __attribute__((ext_vector_type(2)));
// One swap per 32-bit half: afterwards x holds, in lanes with bit 5 (resp. 4) CLEAR, a of the lane itself and, in lanes with it SET, b of
// the partner lane; y holds a of the partner resp. b of the lane itself.  x + y is a whole reduce-scatter step of a value-splitting
// butterfly (lanes with the bit clear end with a_own + a_partner, the others with b_partner + b_own) without a single select.
__device__ __forceinline__ void lane_swap32(const double a, const double b, double& x, double& y) {
    const glio_v2u lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const glio_v2u hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    x = __hiloint2double((int)hi.x, (int)lo.x); y = __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ void lane_swap16(const double a, const double b, double& x, double& y) {
    const glio_v2u lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const glio_v2u hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    x = __hiloint2double((int)hi.x, (int)lo.x); y = __hiloint2double((int)hi.y, (int)lo.y);
}
// value of lane (i ^ OFF) for OFF = 8, 4, 2, 1 (inside a 16-lane row): DPP moves
template <int OFF> __device__ __forceinline__ int lane_xor_row_i(const int v) {
    static_assert(OFF == 8 || OFF == 4 || OFF == 2 || OFF == 1, "row-local xor distances");
    if (OFF == 8) return __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0xF, false);                 // row_ror:8
    if (OFF == 4) {
        const int p = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xF, 0x5, false);                    // row_shl:4 into lanes 0-3, 8-11 (from i + 4)
        return __builtin_amdgcn_update_dpp(p, v, 0x114, 0xF, 0xA, false);                           // row_shr:4 into lanes 4-7, 12-15 (from i - 4)
    }
    if (OFF == 2) return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false);                  // quad_perm:[2,3,0,1]
    return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false);                                // quad_perm:[1,0,3,2]
}
template <int OFF> __device__ __forceinline__ double lane_xor_row_d(const double v) {
    return __hiloint2double(lane_xor_row_i<OFF>(__double2hiint(v)), lane_xor_row_i<OFF>(__double2loint(v)));
}
// v of this lane + v of lane (i ^ OFF): one butterfly stage, any distance
template <int OFF> __device__ __forceinline__ double lane_xor_sum(const double v) {
    if (OFF == 32) { double x, y; lane_swap32(v, v, x, y); return x + y; }
    else if (OFF == 16) { double x, y; lane_swap16(v, v, x, y); return x + y; }
    else return v + lane_xor_row_d<(OFF == 32 || OFF == 16) ? 1 : OFF>(v);
}
// the six-stage xor butterfly (distances 32, 16, 8, 4, 2, 1 in this order) with an associative, commutative op: every lane ends with the
// same value, the one the __shfl_xor loop `for (off = 32; off > 0; off >>= 1) v = op(v, __shfl_xor(v, off))` produces
template <class Op> __device__ __forceinline__ double wave_butterfly_d(double v, const Op op) {
    double x, y;
    lane_swap32(v, v, x, y); v = op(x, y);
    lane_swap16(v, v, x, y); v = op(x, y);
    v = op(v, lane_xor_row_d<8>(v));
    v = op(v, lane_xor_row_d<4>(v));
    v = op(v, lane_xor_row_d<2>(v));
    v = op(v, lane_xor_row_d<1>(v));
    return v;
}
__device__ __forceinline__ double wave_sum(const double v) { return wave_butterfly_d(v, [](const double a, const double b) { return a + b; }); }
__device__ __forceinline__ double wave_max(const double v) { return wave_butterfly_d(v, [](const double a, const double b) { return fmax(a, b); }); }
// the shuffle forms, kept for the bit-for-bit check of the above (glio_debug_wave_reduce_check)
__device__ __forceinline__ double wave_sum_shfl(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ------------------------------------------------------------------------------------------------
// launch wrappers (one per translation unit)
// ------------------------------------------------------------------------------------------------
// lidar_kernels.hip
void glio_launch_lidar_linearize(glio_ctx* c, int use_status_cand, int which, int marg = 0);
void glio_lidar_pack_f32(glio_ctx* c);                      // no-op unless the f32 form is selected and the correspondences changed
// factor_kernels.hip
void glio_launch_small_factors(glio_ctx* c, int use_status_cand, int which, int n_ddt, int marg = 0);
void glio_launch_linearize_all(glio_ctx* c, int use_status_cand, int which, int n_ddt);   // K3 + small factors, one launch
void glio_launch_lidar_reduce(glio_ctx* c, int which);      // K3 partials -> d_lidar_blocks (the marginalization's assembly reads the blocks)
void glio_launch_assemble(glio_ctx* c, int use_status_cand, int which, int n_ddt, int band = 0);      // band: the block-tridiagonal part + epoch columns only
int glio_chain_kind(const glio_ctx* c, int n_ddt);
void glio_launch_stream_read(glio_ctx* c);
int glio_assoc_build_map_dev(glio_ctx* c, const float4* d_pts, int n);
int glio_assoc_select(glio_ctx* c, int slot, const int32_t* indices, int n);
int glio_assoc_select_window(glio_ctx* c, const int32_t* offsets, const int32_t* indices, const uint8_t* changed);
int glio_assoc_run_window(glio_ctx* c, const double* quats, const double* trans, int32_t* out_counts);
int glio_assoc_run_window_async(glio_ctx* c, const double* quats, const double* trans);
int glio_assoc_finish_pending(glio_ctx* c);
void glio_localmap_destroy(glio_ctx* c);
// solver_kernels.hip
void glio_launch_tr_step(glio_ctx* c, int n_ddt);
void glio_chain_tabs_upload(glio_ctx* c);
int glio_solver_path(const glio_ctx* c, int n_ddt);           // 2 keyframe chain, 1 arrow, 0 dense
int glio_solver_needs_dense_H(const glio_ctx* c, int n_ddt);  // 0: the step (k_chain_step) gathers from the factor blocks itself
// marginalization of slot 0 from lidar_blocks/imu_blocks/prior H of buffer 0 (evaluated with marg = 1)
int glio_launch_marginalize(glio_ctx* c, int imu_edge0, double** J0_dev, double** r0_dev, int** ok_dev);
size_t glio_tr_step_lds_bytes(int n);
// assoc_kernels.hip
int glio_assoc_create(glio_ctx* c);
// the resident scan of a slot changed (uploaded / moved by the slide): keep the presorted copy the tiled search reads in step
void glio_assoc_scan_uploaded(glio_ctx* c, int slot, int n);
void glio_assoc_presort_row(glio_ctx* c, hipStream_t stream, size_t row_offset, int n);
void glio_assoc_destroy(glio_ctx* c);
int glio_assoc_build_map(glio_ctx* c, const void* map_points, int n, int stride, int ioff);
int glio_assoc_run(glio_ctx* c, int slot, const double q[4], const double t[3], int* out_count);
void glio_assoc_time_hooks(glio_ctx* c, int which, int reps, float* ms);

void glio_set_error(const char* fmt, ...);
// roctx range around a C-ABI entry point (SURVEY section 5 "tracing"): visible in rocprofv3 --marker-trace when GLIO_ROCTX=1 (libroctx64 is
// opened with dlopen: no link dependency, no cost when the switch is off)
struct GlioTraceRange { explicit GlioTraceRange(const char* name); ~GlioTraceRange(); bool on; };
#define GLIO_TRACE(name) GlioTraceRange glio_trace_range_(name)
#define GLIO_HIP_CHECK(expr)                                                              \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            glio_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return GLIO_E_HIP;                                                            \
        }                                                                                 \
    } while (0)
