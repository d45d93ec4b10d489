// solver_kernels.hip -- K7: one trust-region iteration of ceres::Solve as configured by the reference
// (GLIO/src/Estimator.cpp:2424-2433: SPARSE_NORMAL_CHOLESKY, DOGLEG, 15 iterations, monotonic steps),
// restated on the dense device-resident normal equations and run entirely on the GPU.
//
// The steps of one iteration, all reading / writing the SolverStatus record in device memory:
//   prepare (tr_prepare_body; its own launch k_tr_prepare, or the prologue of k_chain_solve)   step evaluation of the
//                 candidate produced by the previous iteration: parameter / function tolerance, relative decrease ->
//                 accept (swap the double-buffered H,g,x) or reject, dogleg radius update (Ceres 1.14
//                 TrustRegionMinimizer); loop-top checks (max iterations, gradient tolerance, min radius); Jacobi
//                 scaling, D = sqrt(clamp(diag)), Cauchy direction u = S g~/D
//   k_tr_scale   (n/4 workgroups, one wavefront per row)  t = H u  and  L = S H S + mu D^2 (lower triangle,
//                 right-hand side S g carried as row n): the only O(n^2) streaming pass, spread over the chip
//                 (not launched on the keyframe-chain path: k_chain_solve forms what it needs while staging)
//   the structured factorisation when the graph permits: k_chain_solve (one launch, incl. prepare and scale) or
//                 k_arrow_forward / k_arrow_schur / k_arrow_solve
//   k_tr_finish  (1 workgroup) = tr_factor_body + tr_dogleg_body:
//                 Cauchy step length; the structured solver's result, or the blocked LEFT-looking Cholesky of L: per
//                 16-column panel every wavefront owns 16x16 output tiles and runs one long v_mfma_f64_16x16x4_f64 chain
//                 A_tile -= L[rows, 0:k0] L[k0:k0+16, 0:k0]^T (B operand = the 16 pivot rows staged in LDS,
//                 A operand streamed from L2 as 16-byte loads), then the diagonal block is factored in
//                 registers by wavefront 0 (v_readlane broadcasts) and the rows below are solved one lane
//                 per row; back substitution; mu retry x10 on breakdown (DoglegStrategy);
//                 then the traditional dogleg interpolation, model cost change (O(n): H S step follows
//                 from the two products already known), candidate x (+) delta.  An invalid step (model cost
//                 change <= 0) consumes an iteration without producing a candidate, as in Ceres.
// The host enqueues such groups interleaved with the linearisation kernels one at a time (capi.hip: enqueue_solve) and
// takes the result from mapped host memory; kernels exit immediately once status.done is set, and the
// linearisation kernels also when no candidate is pending.
//
// The Cholesky is the one GEMM-shaped piece of the whole path and runs on the matrix cores; it is
// latency/LDS-bound dense fp64 on a single CU (n <= ~1000 unknowns), not roofline material.
//
// NOTE: no __restrict__ on anything in this file: every buffer here is handed between lanes of the
// workgroup across s_barrier.  And a kernel here must not re-read global data it has itself rewritten unless the two
// cannot share a 128 B line with anything it loaded earlier (see glio_ctx::vstride and DESIGN.md, "coherence trap").
#include <cstring>
#include <type_traits>
#include <vector>

#include "glio_device.h"

#define TR_THREADS 512
#define TR_WAVES (TR_THREADS / 64)
#define TR_NB 16
#define TR_PS (TR_NB + 2)      // padded LDS row stride of the 16x16 diagonal block

struct TrArgs {
    int W, n, n_ddt, max_iterations;
    const int* stop_word;        // mapped host word: equals the solve's id once the host's clock passed max_solver_time_s (nullptr: no limit)
    double min_relative_decrease, function_tolerance, gradient_tolerance, parameter_tolerance;
    double min_radius, initial_radius, max_radius;
    int jacobi_scaling;
    int lm;                   // 1 = Levenberg-Marquardt strategy (mu = 1 / radius, no dogleg interpolation)
    int perm_mode;            // elimination order written by k_tr_scale: 0 = [d | s | p] (arrow / dense), 1 = [d | keyframes] (chain)
    int fused_chain;          // 1 = k_chain_solve did k_tr_prepare's and k_tr_scale's work itself: a.L is NOT filled (the dense
                              // fallback rebuilds it), t = H u comes from the chain kernel
    double* x0; double* x1; double* xout;
    const double* H0; const double* H1; const double* g0; const double* g1; const double* c0; const double* c1;
    double* L; double* vec; int vstride;
    SolverStatus* status;
    int* arrow_flag; const double* arrow_z;      // structured solver result (null = dense only)
    const double* hd0; const double* hd1;        // diag(H) of buffers 0 / 1 when no dense H exists (k_chain_step), else null
    int* progress;                               // host-mapped {(solve id << 16) | groups started, id of the finished solve}
    SolverStatus* status_host; double* xout_host; // host-mapped copy of the result (glio_solve reads it without a device-to-host copy)
};

// workspace vectors (global, persist across the launches of one solve)
#define V_SCALE(a) ((a).vec + 0 * (a).vstride)
#define V_DIAG(a) ((a).vec + 1 * (a).vstride)
#define V_GRAD(a) ((a).vec + 2 * (a).vstride)   /* g~ = S g / D                     */
#define V_GN(a) ((a).vec + 3 * (a).vstride)     /* Gauss-Newton step in D-space      */
#define V_Y(a) ((a).vec + 4 * (a).vstride)      /* y = (S H S + mu D^2)^-1 S g       */
#define V_W(a) ((a).vec + 5 * (a).vstride)      /* scale * step = delta              */
#define V_U(a) ((a).vec + 6 * (a).vstride)      /* u = S g~ / D (Cauchy direction)   */
#define V_T(a) ((a).vec + 7 * (a).vstride)      /* t = H u                           */

// Elimination ordering of the linear system: [clock-drift epochs (diagonal block) | speed-bias blocks, 9 per keyframe
// (block tridiagonal: only an IMU / Doppler edge couples neighbours) | poses, 6 per keyframe (dense through the prior)].
// Natural index (15 per keyframe: t q v ba bg, then the epochs) -> position in the factored matrix.
// mode 1 (keyframe chain, prior block diagonal): [clock-drift epochs | keyframe blocks of 15 in natural order].
__device__ __forceinline__ int tr_perm(const int i, const int W, const int nd, const int mode = 0) {
    if (i >= 15 * W) return i - 15 * W;
    if (mode == 1) return nd + i;
    const int sl = i / 15, l = i - 15 * sl;
    return l < 6 ? nd + 9 * W + 6 * sl + l : nd + 9 * sl + (l - 6);
}

// SolverStatus read through L2 (agent-scope loads bypass this CU's vector L1): for code that re-reads the record after the
// same kernel rewrote it (k_chain_step runs the state machine, the factorisation and the dogleg step in one launch)
__device__ __forceinline__ SolverStatus status_load_l2(const SolverStatus* p) {
    static_assert(sizeof(SolverStatus) % 8 == 0, "SolverStatus is read as 64-bit words");
    SolverStatus s;
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(p);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(&s);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(SolverStatus) / 8); ++k) dst[k] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return s;
}

__device__ __forceinline__ double block_sum(double v, double* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    double s = 0;
#pragma unroll
    for (int k = 0; k < TR_WAVES; ++k) s += red[k];
    return s;
}
// N sums through ONE pair of barriers: the same butterfly, the same inter-wave order per value as block_sum, so each result is
// bit-identical to a separate block_sum call.  red needs N * TR_WAVES doubles.
// LB: LDS-only barriers (GLIO_BLOCK_LDS_SYNC) -- for callers whose global stores nobody reads back in this launch.
template <int N, bool LB = false>
__device__ __forceinline__ void block_sum_n(double (&v)[N], double* red) {
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = wave_sum(v[k]);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (LB) GLIO_BLOCK_LDS_SYNC(); else __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) red[k * TR_WAVES + wv] = v[k];
    }
    if (LB) GLIO_BLOCK_LDS_SYNC(); else __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double s = 0;
#pragma unroll
        for (int w = 0; w < TR_WAVES; ++w) s += red[k * TR_WAVES + w];
        v[k] = s;
    }
}
template <bool LB = false>
__device__ __forceinline__ double block_max(double v, double* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (LB) GLIO_BLOCK_LDS_SYNC(); else __syncthreads();
    if (lane == 0) red[wv] = v;
    if (LB) GLIO_BLOCK_LDS_SYNC(); else __syncthreads();
    double s = 0;
#pragma unroll
    for (int k = 0; k < TR_WAVES; ++k) s = fmax(s, red[k]);
    return s;
}

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double v2f64 __attribute__((ext_vector_type(2)));

extern __shared__ __attribute__((aligned(16))) unsigned char tr_lds[];

// LDS carve of the factor kernel
__device__ __host__ __forceinline__ int bp_stride(int n) { return ((n + 15) & ~15) + 2; }   // even -> 16-B aligned rows

// ------------------------------------------------------------------------------------------------
// Blocked left-looking Cholesky of the leading n x n block of the (n+1) x n row-major matrix A (lower
// triangle), row n carried along (= forward substitution of the right-hand side).
// ------------------------------------------------------------------------------------------------
// SEMI = true (marginalization): `dtol[k]` is the threshold below which pivot k counts as zero; such a column of L
// (and its entry of the carried row) is set to zero -- the Cholesky form of the reference's eigenvalue truncation.
template <bool SEMI = false>
__device__ __forceinline__ bool chol_left_looking(double* A, const int n, double* Bp, double* part, double* sD, int* flag, const int skip = 0,
                                                  const double* dtol = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ld = n, SB = bp_stride(n);
    const int li = lane & 15, lk = lane >> 4;
    if (tid == 0) *flag = 0;
    for (int k0 = 0; k0 < n; k0 += TR_NB) {
        const int nb = min(TR_NB, n - k0);
        const int m = n + 1 - k0;                 // rows of this panel (k0 .. n)
        const int ntile = (m + 15) >> 4;
        if (k0 > 0 && !(skip & 1)) {
            // (0) the 16 pivot rows L[k0:k0+16, 0:k0] -> LDS (B operand of every tile of this panel)
            for (int j = wv; j < TR_NB; j += TR_WAVES) {
                const bool live = j < nb;
                const double* src = A + (size_t)(k0 + (live ? j : 0)) * ld;
                for (int c = lane; c < k0; c += 64) Bp[j * SB + c] = live ? src[c] : 0.0;
            }
            __syncthreads();
            // (1) tile update on the matrix cores: acc(16x16) = A_tile - sum_c L[rows,c] L[k0+j,c].
            //     Work unit = (tile, k-split): late panels have few tiles and long chains, so the chain is cut
            //     into KS pieces (<= 16 units in flight) and the partial accumulators are summed through LDS.
            const int KS = ntile >= 9 ? 1 : min(8, 16 / ntile);
            const int nk8 = k0 >> 3, ck8 = (nk8 + KS - 1) / KS;
            for (int unit = wv; unit < ntile * KS; unit += TR_WAVES) {
                const int I = unit / KS, ks = unit - I * KS;
                const int cb = ks * ck8 * 8, ce = min(k0, cb + ck8 * 8);
                const int R0 = k0 + 16 * I;
                const int arow = R0 + li;                       // A-operand row of this lane
                const bool arow_ok = arow <= n;
                const double* ap = A + (size_t)(arow_ok ? arow : n) * ld + 2 * lk;
                const double* bp = Bp + li * SB + 2 * lk;
                v4f64 acc;
                double* cbase = A + (size_t)(R0 + lk) * ld + k0 + li;   // C: lane l, reg r -> row (l>>4)+4r, col l&15
                bool okr[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    okr[r] = (R0 + lk + 4 * r) <= n && li < nb;
                    acc[r] = (okr[r] && ks == 0) ? cbase[(size_t)(4 * r) * ld] : 0.0;
                }
                int c0 = cb;
                for (; c0 + 32 <= ce; c0 += 32) {               // 4 x (16-byte A load + 16-byte LDS read) in flight
                    v2f64 av[4], bv[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        av[q] = *reinterpret_cast<const v2f64*>(ap + c0 + 8 * q);
                        bv[q] = *reinterpret_cast<const v2f64*>(bp + c0 + 8 * q);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double ax = arow_ok ? -av[q][0] : 0.0, ay = arow_ok ? -av[q][1] : 0.0;
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ax, bv[q][0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ay, bv[q][1], acc, 0, 0, 0);
                    }
                }
                for (; c0 < ce; c0 += 8) {
                    const v2f64 a1 = *reinterpret_cast<const v2f64*>(ap + c0);
                    const v2f64 b1 = *reinterpret_cast<const v2f64*>(bp + c0);
                    const double ax = arow_ok ? -a1[0] : 0.0, ay = arow_ok ? -a1[1] : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ax, b1[0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ay, b1[1], acc, 0, 0, 0);
                }
                if (KS == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (okr[r]) cbase[(size_t)(4 * r) * ld] = acc[r];
                } else {
                    *reinterpret_cast<v4f64*>(part + (size_t)unit * 256 + lane * 4) = acc;
                }
            }
            __syncthreads();
            if (KS > 1) {
                for (int I = wv; I < ntile; I += TR_WAVES) {
                    v4f64 acc = *reinterpret_cast<const v4f64*>(part + (size_t)(I * KS) * 256 + lane * 4);
                    for (int ks = 1; ks < KS; ++ks) acc += *reinterpret_cast<const v4f64*>(part + (size_t)(I * KS + ks) * 256 + lane * 4);
                    const int R0 = k0 + 16 * I;
                    double* cbase = A + (size_t)(R0 + lk) * ld + k0 + li;
#pragma unroll
                    for (int r = 0; r < 4; ++r) if ((R0 + lk + 4 * r) <= n && li < nb) cbase[(size_t)(4 * r) * ld] = acc[r];
                }
                __syncthreads();
            }
        }
        // (2+3) fused panel factorisation: every wavefront carries the diagonal block in lanes 0-15 (factored
        //       redundantly, identical arithmetic) and 48 of the rows below (incl. the carried row n) in lanes 16-63.
        //       The 16 right-looking register steps -- pivot and multipliers broadcast with v_readlane -- factor the
        //       block AND solve X L_kk^T = A_panel for those rows: no separate triangular-solve phase, no LDS.
        const int r0 = k0 + nb;
        const int mb = n + 1 - r0;
        // The factored diagonal block goes back to memory only AFTER the barrier below: with more than TR_WAVES * 48 = 384 rows under the panel
        // the wavefronts come round a second time and load the diagonal block again -- it must still be the unfactored one.  (Until the end of
        // round 3 wavefront 0 stored it at the end of its first round: every system with n >= 400 was factored wrongly, rel. error 3e-2; found
        // with scripts/dense_probe_414.py, pinned by tests/test_hip_parity.py::test_blocked_cholesky_sizes.)
        double adg[TR_NB];
#pragma unroll
        for (int j = 0; j < TR_NB; ++j) adg[j] = 0.0;
        if (!(skip & 2)) {
            for (int base = wv * 48; base < mb; base += TR_WAVES * 48) {       // mb >= 1: wavefront 0 always runs
                const bool isdiag = lane < TR_NB;
                const int bi = base + lane - TR_NB;
                const bool live = isdiag ? lane < nb : bi < mb;
                double* prow = A + (size_t)(live ? (isdiag ? k0 + lane : r0 + bi) : 0) * ld + k0;
                double a[TR_NB];
#pragma unroll
                for (int j = 0; j < TR_NB; ++j) a[j] = (live && j < nb && (!isdiag || j <= lane)) ? prow[j] : ((isdiag && lane == j) ? 1.0 : 0.0);
                bool bad = false;
                double tolv = 0.0;
                if (SEMI) tolv = (lane < nb) ? dtol[k0 + lane] : 0.0;
#pragma unroll
                for (int j = 0; j < TR_NB; ++j) {
                    double djj = readlane_d(a[j], j);
                    bool null_pivot = false;
                    if (SEMI) null_pivot = j < nb && djj <= readlane_d(tolv, j);
                    if (!null_pivot && (!(djj > 0.0) || !isfinite(djj))) { bad = true; djj = 1.0; }
                    const double rd = null_pivot ? 0.0 : rsqrt(djj);
                    const double lij = (lane == j) ? djj * rd : a[j] * rd;
                    a[j] = lij;
#pragma unroll
                    for (int c = j + 1; c < TR_NB; ++c) a[c] -= lij * readlane_d(lij, c);   // unmasked: entries above the diagonal
                }                                                                           // are never read or stored
                if (live && !isdiag) {
#pragma unroll
                    for (int j = 0; j < TR_NB; ++j) if (j < nb) prow[j] = a[j];
                }
                if (base == 0 && wv == 0) {
#pragma unroll
                    for (int j = 0; j < TR_NB; ++j) adg[j] = a[j];
                }
                if (bad && wv == 0 && lane == 0) *flag = 1 + k0;
            }
        }
        __syncthreads();
        if (*flag) return false;
        if (!(skip & 2) && wv == 0 && lane < nb) {
            double* prow = A + (size_t)(k0 + lane) * ld + k0;
#pragma unroll
            for (int j = 0; j < TR_NB; ++j) if (j < nb && j <= lane) prow[j] = adg[j];
        }
        // Inside the loop nothing reads this diagonal block again (later panels touch rows >= k0 + 16 only), but after the LAST panel
        // the caller does (the marginalization kernel copies L out with all eight wavefronts straight after the call): the store
        // above must be ordered before the return by a barrier, not by timing.
        if (k0 + TR_NB >= n) __syncthreads();
    }
    return true;
}

// Solve L^T z = y (L lower n x n in A, y in LDS, overwritten by z), blocked from the bottom up.
__device__ __forceinline__ void back_substitute(const double* A, const int n, double* y, double* sD) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ld = n;
    const int nblk = (n + TR_NB - 1) / TR_NB;
    for (int b = nblk - 1; b >= 0; --b) {
        const int k0 = b * TR_NB, nb = min(TR_NB, n - k0);
        if (tid < TR_NB * TR_NB) {
            const int i = tid / TR_NB, j = tid % TR_NB;
            sD[i * TR_PS + j] = (i < nb && j <= i) ? A[(size_t)(k0 + i) * ld + k0 + j] : (i == j ? 1.0 : 0.0);
        }
        __syncthreads();
        if (wv == 0) {        // lane i holds y_i; columns eliminated from the bottom with readlane broadcasts
            double yi = (lane < nb) ? y[k0 + lane] : 0.0;
            const double rdiag = (lane < TR_NB) ? 1.0 / sD[lane * TR_PS + lane] : 1.0;
#pragma unroll
            for (int k = TR_NB - 1; k >= 0; --k) {
                const double zk = readlane_d(yi, k) * readlane_d(rdiag, k);
                if (lane == k) yi = zk;
                else if (lane < k) yi -= sD[k * TR_PS + lane] * zk;
            }
            if (lane < nb) y[k0 + lane] = yi;
        }
        __syncthreads();
        for (int i = tid; i < k0; i += TR_THREADS) {
            double s = y[i];
#pragma unroll 4
            for (int k = 0; k < nb; ++k) s -= A[(size_t)(k0 + k) * ld + i] * y[k0 + k];
            y[i] = s;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-resident variant for small systems (the 6W x 6W pose block of the arrow solver): the lower triangle is kept
// PACKED in LDS (row i at i(i+1)/2, the carried right-hand side as row n), right-looking: per 16-column panel the
// diagonal block is factored in the registers of wavefront 0, the rows below are solved one lane per row, and the
// trailing triangle takes its rank-16 update on the matrix cores (one 16x16 tile per wavefront, operands read from
// LDS).  No global-memory round trip between the phases.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int pk_off(const int i) { return (i * (i + 1)) >> 1; }
__host__ __device__ __forceinline__ size_t pk_doubles(const int n) { const size_t d = (size_t)(n + 1) * (n + 2) / 2; return d + (d & 1); }

__device__ __forceinline__ bool chol_packed_lds(double* P, const int n, double* sD, int* flag) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    if (tid == 0) *flag = 0;
    __syncthreads();
    for (int k0 = 0; k0 < n; k0 += TR_NB) {
        const int nb = min(TR_NB, n - k0);
        // (a+b) fused: every participating wavefront carries the diagonal block in lanes 0-15 (factored redundantly)
        //       and 48 of the rows below in lanes 16-63; the 16 right-looking register steps then factor the block
        //       AND solve those rows with the same v_readlane broadcasts -- no separate triangular-solve phase.
        const int r0 = k0 + nb;
        const int mb = n + 1 - r0;
        double adg[TR_NB];                        // wavefront 0: the factored diagonal block, stored after the barrier (the other wavefronts
#pragma unroll                                    // read the unfactored one at the start of their step; nothing orders that read before a store here)
        for (int j = 0; j < TR_NB; ++j) adg[j] = 0.0;
        if (wv == 0 || wv * 48 < mb) {
            const bool isdiag = lane < TR_NB;
            const int bi = wv * 48 + lane - TR_NB;
            const bool live = isdiag ? lane < nb : bi < mb;
            double* prow = P + pk_off(live ? (isdiag ? k0 + lane : r0 + bi) : 0) + k0;
            double a[TR_NB];
#pragma unroll
            for (int j = 0; j < TR_NB; ++j) a[j] = (live && j < nb && (!isdiag || j <= lane)) ? prow[j] : ((isdiag && lane == j) ? 1.0 : 0.0);
            bool bad = false;
#pragma unroll
            for (int j = 0; j < TR_NB; ++j) {
                double djj = readlane_d(a[j], j);
                if (!(djj > 0.0) || !isfinite(djj)) { bad = true; djj = 1.0; }
                const double rd = rsqrt(djj);
                const double lij = (lane == j) ? djj * rd : a[j] * rd;
                a[j] = lij;
#pragma unroll
                for (int c = j + 1; c < TR_NB; ++c) a[c] -= lij * readlane_d(lij, c);
            }
            if (live && !isdiag) {
#pragma unroll
                for (int j = 0; j < TR_NB; ++j) if (j < nb) prow[j] = a[j];
            }
            if (wv == 0) {
#pragma unroll
                for (int j = 0; j < TR_NB; ++j) adg[j] = a[j];
            }
            if (bad && wv == 0 && lane == 0) *flag = 1 + k0;
        }
        __syncthreads();
        if (*flag) return false;
        if (wv == 0 && lane < nb) {
            double* prow = P + pk_off(k0 + lane) + k0;
#pragma unroll
            for (int j = 0; j < TR_NB; ++j) if (j < nb && j <= lane) prow[j] = adg[j];
        }
        if (nb == TR_NB && r0 < n) {
            const int T = (n + 1 - r0 + 15) >> 4;
            const int ntiles = (T * (T + 1)) >> 1;
            for (int t = wv; t < ntiles; t += TR_WAVES) {
                int I = 0;
                while (((I + 1) * (I + 2)) >> 1 <= t) ++I;
                const int J = t - ((I * (I + 1)) >> 1);
                const int ra = r0 + 16 * I + li, rb = r0 + 16 * J + li;
                const bool oka = ra <= n, okb = rb < n;
                const double* pa = P + pk_off(oka ? ra : n) + k0 + lk;
                const double* pb = P + pk_off(okb ? rb : 0) + k0 + lk;
                double ax[4], bx[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { ax[q] = pa[4 * q]; bx[q] = pb[4 * q]; }
                v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(oka ? ax[q] : 0.0, okb ? bx[q] : 0.0, acc, 0, 0, 0);
                const int colc = r0 + 16 * J + li;            // C: lane l, reg r -> row (l>>4)+4r, col l&15
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rowc = r0 + 16 * I + lk + 4 * r;
                    if (rowc <= n && colc < n && colc <= rowc) P[pk_off(rowc) + colc] -= acc[r];
                }
            }
        }
        __syncthreads();
    }
    return true;
}

// L^T z = y for the packed LDS factor; y (LDS) is overwritten by z
__device__ __forceinline__ void backsub_packed_lds(const double* P, const int n, double* y, double* sD) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nblk = (n + TR_NB - 1) / TR_NB;
    for (int b = nblk - 1; b >= 0; --b) {
        const int k0 = b * TR_NB, nb = min(TR_NB, n - k0);
        if (tid < TR_NB * TR_NB) {
            const int i = tid / TR_NB, j = tid % TR_NB;
            sD[i * TR_PS + j] = (i < nb && j <= i) ? P[pk_off(k0 + i) + k0 + j] : (i == j ? 1.0 : 0.0);
        }
        __syncthreads();
        if (wv == 0) {
            double yi = (lane < nb) ? y[k0 + lane] : 0.0;
            const double rdiag = (lane < TR_NB) ? 1.0 / sD[lane * TR_PS + lane] : 1.0;
#pragma unroll
            for (int k = TR_NB - 1; k >= 0; --k) {
                const double zk = readlane_d(yi, k) * readlane_d(rdiag, k);
                if (lane == k) yi = zk;
                else if (lane < k) yi -= sD[k * TR_PS + lane] * zk;
            }
            if (lane < nb) y[k0 + lane] = yi;
        }
        __syncthreads();
        for (int i = tid; i < k0; i += TR_THREADS) {
            double sacc = y[i];
#pragma unroll 4
            for (int k = 0; k < nb; ++k) sacc -= P[pk_off(k0 + k) + i] * y[k0 + k];
            y[i] = sacc;
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void finalize(const TrArgs& a, const SolverStatus& s) {
    // Publication in host-mapped memory WITHOUT fences: [state | status with the checksum of both] and then the solve's tag, all as plain /
    // relaxed stores.  The host accepts the payload only when its checksum adds up and re-reads until it does (capi.hip, glio_solve), so the
    // order in which the pieces become visible does not matter for correctness -- three system-scope fences here (each a PCIe round trip that
    // every wavefront of the workgroup sat out) only delayed the moment the host could see the tag.
    const double* xc = s.cur ? a.x1 : a.x0;
    const int nx = 16 * a.W + a.n_ddt;
    for (int k = threadIdx.x; k < nx; k += blockDim.x) { const double v = xc[k]; a.xout[k] = v; a.xout_host[k] = v; }
    if (threadIdx.x < 64) {
        // the checksum of the payload (glio_device.h), by the first wavefront alone (no LDS: some callers have none to spare)
        unsigned long long part = 0;
        for (int k = threadIdx.x; k < nx; k += 64) part += glio_result_mix((unsigned long long)__double_as_longlong(xc[k]), 64ull + (unsigned long long)k);
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned lo = __shfl_xor((unsigned)(part & 0xffffffffull), off, 64), hi = __shfl_xor((unsigned)(part >> 32), off, 64);
            part += ((unsigned long long)hi << 32) | lo;
        }
        if (threadIdx.x == 0) {
            SolverStatus t = s;
            t.checksum = 0;
            unsigned long long sum = part;
            const unsigned long long* words = reinterpret_cast<const unsigned long long*>(&t);
            for (int w = 0; w < (int)(sizeof(SolverStatus) / 8); ++w) sum += glio_result_mix(words[w], (unsigned long long)w);
            t.checksum = sum;
            *a.status = t; *a.status_host = t;
            __hip_atomic_store(a.progress + 1, s.solve_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K7a  k_tr_prepare
// ------------------------------------------------------------------------------------------------
// returns true when a linear solve has to follow (status written, vectors ready), false when this group has nothing to do
// (the solve is finished -- possibly just now -- or was finished before)
struct TrDecision {            // what a kernel that continues after tr_prepare_body needs of the new status: it must not re-read
    int cur, reuse; double mu; // *a.status -- the line it loaded at entry may still sit in this CU's L1, which is not refreshed by the
    SolverStatus* full = nullptr;   // store that rewrote it.  full (optional, LDS): the whole record as written back.
};
// LDS mirrors that k_chain_step hands to the state machine (FAST): with them it has no global round trip of its own.  All of it concerns the
// CANDIDATE (buffer 1 - st_in->cur), whose diag(H), g and cost the caller has just gathered; when the step is rejected the state machine
// goes back to the global vectors of the current point.
struct PrepMirror {
    const SolverStatus* st_in;                                 // the record as loaded at kernel entry
    const double* x0; const double* x1;                        // both state buffers
    const double* hd; const double* g; const double* cost;     // diag(H), g, cost of the candidate
    double* scale; double* diag; double* grad;                 // scale: the global vector on entry (phase > 0); all three are left filled for the caller
    long long* dbg;                                            // stamped builds: device-clock marks of the state machine's sections (slots 80..)
    // k_chain_step's helper workgroups read the status record this function rewrites at its end: their completion words (4 * hseq + 2 * fat + produced) are
    // requested at entry and checked before that store; *hprod is cleared when one of them had nothing to produce
    const int* hdone; int hseq; int helpers; int* hprod; int hpolls;
    int* hfat;                                                 // cleared when a helper did not deliver the fat products (bit 1 of its word)
};
#ifdef GLIO_DEV_STAMPS
#define PM_STAMP(k) do { if (FAST && m->dbg && threadIdx.x == 0) m->dbg[k] = wall_clock64(); } while (0)
#else
#define PM_STAMP(k) do { } while (0)
#endif
template <bool FAST = false>
__device__ __forceinline__ bool tr_prepare_body(const TrArgs& a, TrDecision* out = nullptr, const PrepMirror* m = nullptr) {
    // FAST: every hand-over between threads goes through LDS and nothing this function stores to global memory is read back in the
    // same launch, so its barriers do not wait for the stores' acknowledgements
#define PB_SYNC() do { if (FAST) GLIO_BLOCK_LDS_SYNC(); else __syncthreads(); } while (0)
    __shared__ double red[32];
    __shared__ SolverStatus s;
    const int tid = threadIdx.x;
    const int n = a.n, W = a.W, nx = 16 * W + a.n_ddt;
    if (tid == 0) {
        if (FAST) s = *m->st_in; else s = *a.status;
        s.group += 1;
        a.status->group = s.group;
    }
    int hword = 0;
    if (FAST && m->helpers && tid < m->helpers) hword = __hip_atomic_load(&m->hdone[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    PB_SYNC();
    if (s.done) return false;
    PM_STAMP(80);
    double* scale = V_SCALE(a); double* diag = V_DIAG(a); double* grad = V_GRAD(a); double* u = V_U(a);
    const int cand0 = 1 - s.cur;                     // (FAST: the buffer the mirrors describe)

    if (s.cand_pending) {
        const int cand = 1 - s.cur;
        if (s.phase == 0) {
            const double* Hc = cand ? a.H1 : a.H0;
            const double* hdc = cand ? a.hd1 : a.hd0;
            for (int i = tid; i < n; i += TR_THREADS) {
                const double h = FAST ? m->hd[i] : (a.hd0 ? hdc[i] : Hc[(size_t)i * n + i]);
                const double sc = a.jacobi_scaling ? 1.0 / (1.0 + sqrt(h)) : 1.0;
                scale[i] = sc;
                if (FAST) m->scale[i] = sc;
            }
            PB_SYNC();
            if (tid == 0) {
                s.cur = cand;
                s.cost = FAST ? *m->cost : *(cand ? a.c1 : a.c0);
                s.initial_cost = s.cost;
                s.phase = 1;
            }
        } else {
            double d2 = 0, x2 = 0;
            if (FAST) {
                const double* xc = s.cur ? m->x1 : m->x0;
                const double* xn = cand ? m->x1 : m->x0;
                for (int k = tid; k < nx; k += TR_THREADS) { const double d = xc[k] - xn[k]; d2 += d * d; x2 += xc[k] * xc[k]; }
            } else {
                const double* xc = s.cur ? a.x1 : a.x0;
                const double* xn = cand ? a.x1 : a.x0;
                for (int k = tid; k < nx; k += TR_THREADS) { const double d = xc[k] - xn[k]; d2 += d * d; x2 += xc[k] * xc[k]; }
            }
            { double v2[2] = {d2, x2}; block_sum_n<2, FAST>(v2, red); d2 = v2[0]; x2 = v2[1]; }
            PM_STAMP(81);
            if (tid == 0) {
                const double ccost = FAST ? *m->cost : *(cand ? a.c1 : a.c0);
                const double step_norm = sqrt(d2), x_norm = sqrt(x2);
                if (step_norm <= a.parameter_tolerance * (x_norm + a.parameter_tolerance)) {
                    s.done = 1; s.termination = GLIO_TERM_PARAMETER_TOL;
                } else if (fabs(s.cost - ccost) <= a.function_tolerance * s.cost) {
                    s.done = 1; s.termination = GLIO_TERM_FUNCTION_TOL;
                } else {
                    const double rel = (s.cost - ccost) / s.model_cost_change;
                    if (rel > a.min_relative_decrease) {
                        s.cur = cand; s.cost = ccost; s.successful += 1;
                        if (a.lm) {                                                           // LevenbergMarquardtStrategy::StepAccepted
                            s.radius = s.radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3));
                            s.radius = fmin(a.max_radius, s.radius);
                            s.decrease_factor = 2.0;
                        } else {
                            if (rel < 0.25) s.radius *= 0.5;                                  // DoglegStrategy::StepAccepted
                            if (rel > 0.75) s.radius = fmax(s.radius, 3.0 * s.dogleg_step_norm);
                            s.mu = fmax(1e-8, 2.0 * s.mu / 10.0);
                        }
                        s.reuse = 0;
                    } else if (a.lm) {
                        s.radius /= s.decrease_factor; s.decrease_factor *= 2.0; s.reuse = 0;   // StepRejected: new damping, refactor
                    } else {
                        s.radius *= 0.5; s.reuse = 1;                                         // StepRejected
                    }
                }
            }
        }
        PB_SYNC();
        if (tid == 0) s.cand_pending = 0;
        PB_SYNC();
    }
    if (s.done) { finalize(a, s); return false; }
    PM_STAMP(82);

    const double* H = s.cur ? a.H1 : a.H0;
    const double* hdv = s.cur ? a.hd1 : a.hd0;
    const double* g = s.cur ? a.g1 : a.g0;
    const double* xc = s.cur ? a.x1 : a.x0;
    const bool mirrored = FAST && s.cur == cand0;    // the current point is the one the mirrors describe (the candidate was accepted)
    // gradient max norm = | x - Plus(x, -g) |_inf
    // (FAST: the rotation entries -- a quaternion update with its sin / cos each -- are taken by the lanes of the last wavefront, which has
    // no entry of its own when n <= 448, instead of by lanes scattered over all of them: a maximum does not care who contributes what)
    auto grad_max = [&](const double* gv, const double* xv) {
        double gmx = 0;
        auto rot = [&](const int sl) {
            const int k = 15 * sl + 3;
            const double d[3] = {-gv[k], -gv[k + 1], -gv[k + 2]};
            double q[4], qn[4];
            for (int c = 0; c < 4; ++c) q[c] = xv[3 * W + 4 * sl + c];
            d_quat_plus(q, d, qn);
            for (int c = 0; c < 4; ++c) gmx = fmax(gmx, fabs(q[c] - qn[c]));
        };
        for (int k = tid; k < n; k += TR_THREADS) {
            if (k < 15 * W && (k % 15) >= 3 && (k % 15) < 6) {
                if (!FAST && (k % 15) == 3) rot(k / 15);
            } else gmx = fmax(gmx, fabs(gv[k]));
        }
        if (FAST && tid >= TR_THREADS - 64) for (int sl = tid - (TR_THREADS - 64); sl < W; sl += 64) rot(sl);
        return gmx;
    };
    double gm;
    if (FAST) { const double* xl = s.cur ? m->x1 : m->x0; gm = mirrored ? grad_max(m->g, xl) : grad_max(g, xl); }
    else gm = grad_max(g, xc);
    PM_STAMP(83);
    gm = block_max<FAST>(gm, red);
    PM_STAMP(84);
    if (tid == 0) {
        s.grad_max_norm = gm;
        const bool out_of_time = a.stop_word && *reinterpret_cast<const volatile int*>(a.stop_word) == s.solve_id;      // Ceres: MaxSolverTimeReached
        if (s.iteration >= a.max_iterations || out_of_time) { s.done = 1; s.termination = GLIO_TERM_NO_CONVERGENCE; }
        else if (gm <= a.gradient_tolerance) { s.done = 1; s.termination = GLIO_TERM_GRADIENT_TOL; }
        else if (s.radius <= a.min_radius) { s.done = 1; s.termination = GLIO_TERM_MIN_RADIUS; }
        else s.iteration += 1;
    }
    PB_SYNC();
    if (s.done) { finalize(a, s); return false; }
    // The solve goes on: only now is the host told to enqueue the next kernel group (it then has this whole step, ~70 us,
    // to do so).  Announcing the group at its start instead made the host queue one group beyond the last useful one
    // every time: ~9 empty launches (~20 us) between back-to-back solves.
    // (a plain store to host-coherent memory: it leaves the GPU at once; a system-scope fence here would only stall this workgroup
    // ~1 us until the write is acknowledged across PCIe)
    // (a relaxed system-scope atomic store, NOT a volatile one: the compiler follows a volatile store with s_waitcnt vmcnt(0), i.e. this
    // wavefront would sit out the PCIe round trip, ~1 us, right here)
    if (tid == 0) __hip_atomic_store(a.progress, (s.solve_id << 16) | s.group, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (a.lm && tid == 0) s.mu = 1.0 / s.radius;      // (Hs + D^2 / radius) y = gs
    PB_SYNC();
    PM_STAMP(85);

    if (!s.reuse) {
        auto work_vectors = [&](const double* sv, const double* hv, const double* gv) {
            for (int i = tid; i < n; i += TR_THREADS) {
                double d = sv[i] * sv[i] * hv[i];
                d = d < 1e-6 ? 1e-6 : (d > 1e32 ? 1e32 : d);
                const double dd = sqrt(d);
                diag[i] = dd;
                const double gs = sv[i] * gv[i];
                grad[i] = gs / dd;
                u[i] = sv[i] * (gs / dd) / dd;
                if (FAST) { m->diag[i] = dd; m->grad[i] = gs / dd; }
            }
        };
        if (FAST) { if (mirrored) work_vectors(m->scale, m->hd, m->g); else work_vectors(m->scale, hdv, g); }
        else if (a.hd0) work_vectors(scale, hdv, g);
        else {
            for (int i = tid; i < n; i += TR_THREADS) {
                double d = scale[i] * scale[i] * H[(size_t)i * n + i];
                d = d < 1e-6 ? 1e-6 : (d > 1e32 ? 1e32 : d);
                const double dd = sqrt(d);
                diag[i] = dd;
                const double gs = scale[i] * g[i];
                grad[i] = gs / dd;
                u[i] = scale[i] * (gs / dd) / dd;
            }
        }
    }
    if (FAST && m->helpers && tid < m->helpers) {
        // (relaxed polling: the reader of the helpers' sums issues the acquire fence; the barrier below puts every helper's read of the record
        //  before the store that replaces it)
        // BOUNDED: HIP promises no forward progress between the workgroups of a launch -- on a device with fewer free CUs than the host assumed (CU masks,
        // partitions, a co-tenant) a helper may not even have started.  After hpolls polls the step stops waiting and sums the blocks itself (*hprod = 0);
        // a helper that starts late reads the rewritten record, leaves sums nobody reads and ends -- the next launch cannot start before it has.
        int polls = 0;
        while ((hword >> 2) != m->hseq && polls < m->hpolls) { ++polls; __builtin_amdgcn_s_sleep(1); hword = __hip_atomic_load(&m->hdone[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        if ((hword >> 2) != m->hseq || !(hword & 1)) *m->hprod = 0;
        if ((hword >> 2) != m->hseq || !(hword & 2)) *m->hfat = 0;
    }
    PB_SYNC();
    PM_STAMP(86);
    if (tid == 0) { *a.status = s; if (out && out->full) *out->full = s; }
    if (out) { out->cur = s.cur; out->reuse = s.reuse; out->mu = s.mu; }
    PB_SYNC();
    return true;
}
#undef PB_SYNC
__global__ __launch_bounds__(TR_THREADS) void k_tr_prepare(const TrArgs a) { (void)tr_prepare_body(a); }

// ------------------------------------------------------------------------------------------------
// K7b  k_tr_scale: one wavefront per row of H (multi-workgroup): t = H u, L = S H S + mu D^2, rhs row
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tr_scale(const TrArgs a) {
    const SolverStatus* st = a.status;
    if (st->done || st->reuse) return;
    const int n = a.n;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (a.arrow_flag && blockIdx.x == 0 && threadIdx.x == 0) *a.arrow_flag = 0;
    if (i > n) return;
    const double* H = st->cur ? a.H1 : a.H0;
    const double* g = st->cur ? a.g1 : a.g0;
    const double* scale = V_SCALE(a); const double* diag = V_DIAG(a); const double* u = V_U(a);
    const int W = a.W, nd = a.n_ddt;
    if (i == n) {                      // right-hand side S g as the carried row
        for (int j = lane; j < n; j += 64) a.L[(size_t)n * n + tr_perm(j, W, nd, a.perm_mode)] = scale[j] * g[j];
        return;
    }
    const double mu = st->mu, si = scale[i];
    const double* hrow = H + (size_t)i * n;
    const int pi = tr_perm(i, W, nd, a.perm_mode);
    double s = 0;
    for (int j = lane; j < n; j += 64) {
        const double h = hrow[j];
        s += h * u[j];
        if (j <= i) {
            double v = si * h * scale[j];
            if (i == j) v += mu * diag[i] * diag[i];
            const int pj = tr_perm(j, W, nd, a.perm_mode);
            a.L[(size_t)max(pi, pj) * n + min(pi, pj)] = v;
        }
    }
    s = wave_sum(s);
    if (lane == 0) V_T(a)[i] = s;
}

// ------------------------------------------------------------------------------------------------
// K7c  k_tr_factor
// ------------------------------------------------------------------------------------------------
// `Builder`: how the scaled, regularised matrix S H S + mu D^2 (lower triangle in tr_perm order, right-hand side S g as
// row n) is rebuilt in a.L when the structured factorisation broke down.  The default reads the dense H; k_chain_step, which
// never builds a dense H, passes a builder that gathers the entries from the factor blocks.
struct DenseBuilder { static constexpr bool kHasDenseH = true; __device__ void operator()(const TrArgs&, double) const {} };
template <class Builder = DenseBuilder>
__device__ __forceinline__ void tr_factor_body(const TrArgs& a, const Builder& builder = Builder()) {
    const int tid = threadIdx.x;
    const int n = a.n;
    double* Bp = reinterpret_cast<double*>(tr_lds);                       // 16 x bp_stride(n)
    double* part = Bp + TR_NB * bp_stride(n);                             // 16 units x 256 partial accumulators
    double* sD = part + 16 * 256;                                         // (TR_NB + 1) x TR_PS
    double* ylds = sD + (TR_NB + 1) * TR_PS;                              // n
    double* red = ylds + n + (n & 1);                                     // 32
    int* flag = reinterpret_cast<int*>(red + 32);
    double* smu = red + 40;
    // (in the dynamic carve: k_tr_finish may be given all 160 KB as dynamic LDS, a static variable would not fit beside it)
    double& f_mu = red[41];
    int* fi = reinterpret_cast<int*>(red + 42);
    int &f_done = fi[0], &f_reuse = fi[1], &f_cur = fi[2];
    if (tid == 0) { const SolverStatus st0 = status_load_l2(a.status); f_done = st0.done; f_reuse = st0.reuse; f_cur = st0.cur; f_mu = st0.mu; }
    __syncthreads();
    if (f_done || f_reuse) return;
    const double* H = f_cur ? a.H1 : a.H0;
    const double* g = f_cur ? a.g1 : a.g0;
    const double* scale = V_SCALE(a); const double* diag = V_DIAG(a); const double* grad = V_GRAD(a);
    const double* u = V_U(a); const double* t = V_T(a);
    if (Builder::kHasDenseH && a.fused_chain && !(a.arrow_flag && *a.arrow_flag == 2)) {
        // the chain kernel broke down (non-positive pivot): nobody has written t = H u or the scaled matrix; do k_tr_scale's
        // work here (one wavefront per row), then the dense factorisation below takes over
        double* tw = V_T(a);
        for (int i = tid >> 6; i < n; i += TR_WAVES) {
            const double* hrow = H + (size_t)i * n;
            double sacc = 0;
            for (int j = tid & 63; j < n; j += 64) sacc += hrow[j] * u[j];
            sacc = wave_sum(sacc);
            if ((tid & 63) == 0) tw[i] = sacc;
        }
        __syncthreads();
    }
    // Cauchy step length alpha = |g~|^2 / (u^T H u)
    double p = 0, q2 = 0;
    for (int i = tid; i < n; i += TR_THREADS) { p += u[i] * t[i]; q2 += grad[i] * grad[i]; }
    { double v2[2] = {p, q2}; block_sum_n<2>(v2, red); p = v2[0]; q2 = v2[1]; }
    if (tid == 0) { a.status->alpha = q2 / p; *smu = f_mu; }
    __syncthreads();
    bool solved = false;
    if (a.arrow_flag && *a.arrow_flag == 2) {          // the structured factorisation already solved this system
        for (int j = tid; j < n; j += TR_THREADS) ylds[j] = a.arrow_z[j];
        __syncthreads();
        solved = true;
    }
    for (int attempt = 0; attempt < (a.lm ? 1 : 12) && !solved; ++attempt) {
        const double mu = *smu;
        if (!a.lm && !(mu < 1.0)) break;
        if (!Builder::kHasDenseH) { builder(a, mu); __syncthreads(); }
        else if (attempt > 0 || a.fused_chain) {    // breakdown (or nobody built it yet): S H S + mu D^2 with the current mu (rare)
            for (int i = tid >> 6; i < n; i += TR_WAVES) {
                const double si = scale[i];
                const double* hrow = H + (size_t)i * n;
                const int pi = tr_perm(i, a.W, a.n_ddt, a.perm_mode);
                for (int j = tid & 63; j <= i; j += 64) {
                    double v = si * hrow[j] * scale[j];
                    if (i == j) v += mu * diag[i] * diag[i];
                    const int pj = tr_perm(j, a.W, a.n_ddt, a.perm_mode);
                    a.L[(size_t)max(pi, pj) * n + min(pi, pj)] = v;
                }
            }
            for (int j = tid; j < n; j += TR_THREADS) a.L[(size_t)n * n + tr_perm(j, a.W, a.n_ddt, a.perm_mode)] = scale[j] * g[j];
            __syncthreads();
        }
        const bool ok = chol_left_looking(a.L, n, Bp, part, sD, flag);
        double bad = 1;
        if (ok) {
            for (int j = tid; j < n; j += TR_THREADS) ylds[j] = a.L[(size_t)n * n + j];
            __syncthreads();
            back_substitute(a.L, n, ylds, sD);
            bad = 0;
            for (int j = tid; j < n; j += TR_THREADS) if (!isfinite(ylds[j])) bad = 1;
        }
        bad = block_max(bad, red);
        if (bad == 0) { solved = true; break; }
        __syncthreads();
        if (tid == 0) *smu = mu * 10.0;
        __syncthreads();
    }
    if (solved) {
        double* gn = V_GN(a); double* y = V_Y(a);
        for (int i = tid; i < n; i += TR_THREADS) { const double yi = ylds[tr_perm(i, a.W, a.n_ddt, a.perm_mode)]; y[i] = yi; gn[i] = -diag[i] * yi; }
    }
    __syncthreads();
    if (tid == 0) {
        // Ceres 1.14 DoglegStrategy::ComputeGaussNewtonStep only RAISES mu (the next solve starts from the last successful
        // value; StepAccepted lowers it).  If every mu < max_mu failed, the strategy reports LINEAR_SOLVER_FAILURE and the
        // minimizer counts an invalid step (tr_dogleg_body: StepIsInvalid, five in a row end the solve with FAILURE).
        if (solved) { a.status->mu_used = *smu; if (!a.lm) a.status->mu = *smu; a.status->lin_fail = 0; }
        else { if (!a.lm) a.status->mu = *smu; a.status->lin_fail = 1; }
    }
}

// ------------------------------------------------------------------------------------------------
// K7d  k_tr_dogleg
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void tr_dogleg_body(const TrArgs& a) {
    double* red = reinterpret_cast<double*>(tr_lds);                       // the factorisation's panel buffer is free again
    SolverStatus& s = *reinterpret_cast<SolverStatus*>(red + 32);
    const int tid = threadIdx.x;
    const int n = a.n, W = a.W;
    if (tid == 0) s = status_load_l2(a.status);
    __syncthreads();
    if (s.done) return;
    const double* g = s.cur ? a.g1 : a.g0;
    const double* scale = V_SCALE(a); const double* diag = V_DIAG(a); const double* grad = V_GRAD(a); const double* gn = V_GN(a);
    const double* y = V_Y(a); const double* t = V_T(a);
    double* wvec = V_W(a);
    double gg = 0, nn = 0, gd = 0;
    for (int i = tid; i < n; i += TR_THREADS) { gg += grad[i] * grad[i]; nn += gn[i] * gn[i]; gd += grad[i] * gn[i]; }
    { double v3[3] = {gg, nn, gd}; block_sum_n<3>(v3, red); gg = v3[0]; nn = v3[1]; gd = v3[2]; }
    const double gnorm = sqrt(gg), gnn = sqrt(nn), radius = s.radius, alpha = s.alpha;
    double ca, cb, snorm;       // step (D-space) = ca * grad + cb * gn
    if (a.lm) { ca = 0.0; cb = 1.0; snorm = gnn; }        // Levenberg-Marquardt: the damped step itself
    else if (gnn <= radius) { ca = 0.0; cb = 1.0; snorm = gnn; }
    else if (gnorm * alpha >= radius) { ca = -(radius / gnorm); cb = 0.0; snorm = radius; }
    else {
        const double b_dot_a = -alpha * gd;
        const double a_sq = alpha * alpha * gg;
        const double b_minus_a_sq = nn - 2 * b_dot_a + a_sq;
        const double c = b_dot_a - a_sq;
        const double d = sqrt(c * c + b_minus_a_sq * (radius * radius - a_sq));
        const double beta = (c <= 0) ? (d - c) / b_minus_a_sq : (radius * radius - a_sq) / (d + c);
        ca = -alpha * (1.0 - beta); cb = beta; snorm = -1.0;
    }
    // step_s = step / D = ca * (g~/D) + cb * (-y); w = S step_s.
    // Hs step_s = ca * Hs (g~/D) - cb * Hs y, with Hs (g~/D) = S H u = S t and Hs y = S g - mu D^2 y
    // -> model cost change = -(gs.step_s + step_s^T Hs step_s / 2) without another matrix pass.
    const double mu = s.mu_used;
    double sn2 = 0, lin = 0, quad = 0;
    for (int i = tid; i < n; i += TR_THREADS) {
        const double sv = ca * grad[i] + cb * gn[i];
        sn2 += sv * sv;
        const double step_s = sv / diag[i];
        wvec[i] = scale[i] * step_s;
        const double gs = scale[i] * g[i];
        const double hs_step = ca * (scale[i] * t[i]) - cb * (gs - mu * diag[i] * diag[i] * y[i]);
        lin += gs * step_s;
        quad += step_s * hs_step;
    }
    { double v3[3] = {sn2, lin, quad}; block_sum_n<3>(v3, red); sn2 = v3[0]; lin = v3[1]; quad = v3[2]; }
    if (snorm < 0) snorm = sqrt(sn2);
    const double mcc = -(lin + 0.5 * quad);
    const bool valid = mcc > 0.0 && !s.lin_fail;
    if (tid == 0) {
        s.dogleg_step_norm = snorm;
        if (!valid) {
            s.invalid += 1;
            if (s.invalid >= 5) { s.done = 1; s.termination = GLIO_TERM_FAILURE; }
            if (a.lm) { s.radius /= s.decrease_factor; s.decrease_factor *= 2.0; }
            else s.mu *= 10.0;
            s.lin_fail = 0;
            s.reuse = 0;                        // StepIsInvalid: consumes an iteration, no candidate
            s.cand_pending = 0;
        } else {
            s.invalid = 0;
            s.model_cost_change = mcc;
            s.cand_pending = 1;
        }
    }
    __syncthreads();
    if (s.done) { finalize(a, s); return; }
    if (valid) {        // candidate = x (+) delta into the other buffer
        const double* xc = s.cur ? a.x1 : a.x0;
        double* xn = s.cur ? a.x0 : a.x1;
        for (int k = tid; k < 3 * W; k += TR_THREADS) { const int sl = k / 3, c = k % 3; xn[k] = xc[k] + wvec[15 * sl + c]; }
        for (int sl = tid; sl < W; sl += TR_THREADS) {
            double q[4], qn[4];
            const double d[3] = {wvec[15 * sl + 3], wvec[15 * sl + 4], wvec[15 * sl + 5]};
            for (int c = 0; c < 4; ++c) q[c] = xc[3 * W + 4 * sl + c];
            d_quat_plus(q, d, qn);
            for (int c = 0; c < 4; ++c) xn[3 * W + 4 * sl + c] = qn[c];
        }
        for (int k = tid; k < 9 * W; k += TR_THREADS) { const int sl = k / 9, c = k % 9; xn[7 * W + k] = xc[7 * W + k] + wvec[15 * sl + 6 + c]; }
        for (int k = tid; k < a.n_ddt; k += TR_THREADS) xn[16 * W + k] = xc[16 * W + k] + wvec[15 * W + k];
    }
    __syncthreads();
    if (tid == 0) *a.status = s;
}


// K7c + K7d in one launch (one workgroup): the linear solve (or the structured solver's result) and the dogleg /
// Levenberg-Marquardt step that consumes it
__global__ __launch_bounds__(TR_THREADS) void k_tr_finish(const TrArgs a) {
    tr_factor_body(a);
    __syncthreads();
    tr_dogleg_body(a);
}

// ------------------------------------------------------------------------------------------------
// K7c'  structured ("arrow") factorisation of M = S H S + mu D^2 in the elimination order of tr_perm():
//   [ d: clock-drift epochs, diagonal | s: speed-bias blocks, block tridiagonal | p: poses, dense ]
// Block Cholesky with the same pivots a dense Cholesky in this order would meet, but only the non-zero
// structure is touched and the only long dense factorisation left is the 6W x 6W pose block:
//   k_arrow_forward (one workgroup per 16 columns of [M_ep | b_e]):  L_dd = sqrt(diag); the speed-bias chain
//       L_ii, L_{i+1,i} (9x9 blocks, wavefront 0, register/readlane factorisation of the 18x9 panel) and, one
//       block behind it, Y = L_ee^-1 [M_ep | b_e] for the workgroup's columns (wavefront 1) -- the chain is
//       recomputed by every workgroup (cheap) so that no grid synchronisation is needed;
//   k_arrow_schur   (16x16 tiles):  S_pp = M_pp - Y^T Y, b_p' = b_p - Y^T y_e;
//   k_arrow_solve   (one workgroup):  blocked MFMA Cholesky of S_pp (chol_left_looking), back substitution
//       through p, the speed-bias chain and the epochs.
// Any non-positive pivot raises the flag and k_tr_factor falls back to the dense factorisation (with its mu
// retries), so the outcome is the dense one in every case.
// ------------------------------------------------------------------------------------------------
struct ArrowArgs {
    int W, n, nd, np, K, ldY;
    const double* A;
    const int2* ep_slots; const int* ep_off; const int* ep_list;
    double* Y; double* Lblk; double* Sp; double* z;
    int* flag;
    const SolverStatus* status;
    long long* dbg;
    int lds_chol;             // pose block factored in LDS (packed) instead of through global memory
};
// development aid: wall-clock stamps of the phases of the arrow kernels (build with GLIO_DEV_STAMPS=1, scripts/arrow_time.py)
#ifdef GLIO_DEV_STAMPS
#define AR_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) a.dbg[k] = wall_clock64(); } while (0)
#define WV_STAMP(k) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) a.dbg[k] = wall_clock64(); } while (0)      /* by the first lane of any wavefront */
#else
#define AR_STAMP(k) do { } while (0)
#define WV_STAMP(k) do { } while (0)
#endif
#define AR_LB 190          /* one chain block in LDS / global: 18 rows x stride 10, then 9 reciprocal pivots (+1 pad) */
#define AR_YS 17           /* row stride of the 16-column Y slices in LDS */

__host__ __device__ __forceinline__ size_t arrow_forward_lds_doubles(int W, int nd) {
    return (size_t)(nd + (nd & 1)) + (size_t)nd * AR_YS + (size_t)nd * 18 + (size_t)9 * W * AR_YS + 3 * AR_LB + (size_t)W * 180 +
           (size_t)nd + 2 + (size_t)(W + 2) / 2 + 1 + (size_t)nd + 2 + 2 + 92 +      // ... + the 9x9 (stride 10) update scratch +
           4 * AR_LB + 92;                                                      // the bottom chain's panels, the meeting block, its scratch
}

// One step of a speed-bias chain, executed by one wavefront.  On entry lanes 0-8 hold (in av) the rows of the current
// diagonal block D_i, already corrected by the previous step.  Lanes 9-17 load the off-diagonal panel towards the NEXT
// block of this chain: DOWN = false (top chain, i ascending): rows of B_i = M[s_{i+1}, s_i]; DOWN = true (bottom chain,
// i descending): rows of B_{i-1}^T = M[s_{i-1}, s_i].  Nine register steps (pivot and multipliers by v_readlane) factor
// the 18 x 9 panel; it is stored to Lc (rows 0-8 L_ii, rows 9-17 L_{nb,i}, then the reciprocal pivots) and, from
// workgroup 0, to global memory for the back substitution.  The rank-9 correction C = L_{nb,i} L_{nb,i}^T of the next
// diagonal block is computed by 45 lanes (one (r, j <= r) pair each) into Cs; on exit av = D_nb - C.
template <bool DOWN>
__device__ __forceinline__ void arrow_chain_step(const int i, const int nb, const bool has_nb, double (&av)[9], const double* Blk, double* Lc,
                                                 double* Lg, double* Cs, const int lane, const int pr, const int pj, bool& bad) {
    const int r = lane;
    double nx[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) nx[j] = (has_nb && r < 9) ? Blk[nb * 180 + r * 10 + j] : 0.0;
    if (r >= 9 && r < 18) {
#pragma unroll
        for (int j = 0; j < 9; ++j) av[j] = has_nb ? (DOWN ? Blk[nb * 180 + (9 + j) * 10 + (r - 9)] : Blk[i * 180 + r * 10 + j]) : 0.0;
    }
    double rpv = 0.0;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        double djj = readlane_d(av[j], j);
        if (!(djj > 0.0) || !isfinite(djj)) { bad = true; djj = 1.0; }
        const double rdj = rsqrt(djj);
        const double lij = (lane == j) ? djj * rdj : av[j] * rdj;
        if (lane == j) rpv = rdj;
        av[j] = lij;
#pragma unroll
        for (int c = j + 1; c < 9; ++c) av[c] -= lij * readlane_d(lij, c);      // lanes < c only touch entries above the
    }                                                                            // diagonal, which nobody reads: no masking
    if (r < 18) {
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const double v = (r < 9 && j > r) ? 0.0 : av[j];
            Lc[r * 10 + j] = v;
            if (Lg) Lg[r * 10 + j] = v;
        }
        if (r < 9) { Lc[180 + r] = rpv; if (Lg) Lg[180 + r] = rpv; }
    }
    GLIO_WAVE_LDS_SYNC();
    if (has_nb) {
        const double* X = Lc + 90;
        if (lane < 45) {
            double xa[9], xb[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) { xa[k] = X[pr * 10 + k]; xb[k] = X[pj * 10 + k]; }
            double sacc = 0.0;
#pragma unroll
            for (int k = 0; k < 9; ++k) sacc += xa[k] * xb[k];
            Cs[pr * 10 + pj] = sacc;
        }
        GLIO_WAVE_LDS_SYNC();
        if (r < 9) {
#pragma unroll
            for (int j = 0; j < 9; ++j) nx[j] -= Cs[r * 10 + j];
        }
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) av[j] = (r < 9 && j <= r) ? nx[j] : 0.0;
}

// Forward substitution of one chain block for 16 right-hand-side columns, by one wavefront: lane = (column c, k-group g);
// Y_i = L_ii^-1 (R_i - sum over the (up to two) already eliminated neighbours nbr of L_{i,nbr} Y_nbr), L_{i,nbr} = rows 9-17
// of the neighbour's stored panel.
__device__ __forceinline__ void arrow_y_update(const int i, const int n_nbr, const int nbr0, const double* P0, const int nbr1, const double* P1,
                                               const double* Lc, double* Ys, const int lane) {
    const int c = lane & 15, g = lane >> 4;
    double tv[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) tv[r] = g == 0 ? Ys[(9 * i + r) * AR_YS + c] : 0.0;
    for (int q = 0; q < n_nbr; ++q) {
        const int nbr = q == 0 ? nbr0 : nbr1;
        const double* Lp = (q == 0 ? P0 : P1) + 90;
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            const int k = g + 4 * kk;
            if (k < 9) {
                const double yk = Ys[(9 * nbr + k) * AR_YS + c];
#pragma unroll
                for (int r = 0; r < 9; ++r) tv[r] -= Lp[r * 10 + k] * yk;
            }
        }
    }
    if (n_nbr > 0) {
#pragma unroll
        for (int r = 0; r < 9; ++r) { tv[r] += __shfl_xor(tv[r], 16, 64); tv[r] += __shfl_xor(tv[r], 32, 64); }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {          // column-oriented forward substitution
        tv[k] *= Lc[180 + k];
#pragma unroll
        for (int r = k + 1; r < 9; ++r) tv[r] -= Lc[r * 10 + k] * tv[k];
    }
    if (g == 0) {
#pragma unroll
        for (int r = 0; r < 9; ++r) Ys[(9 * i + r) * AR_YS + c] = tv[r];
    }
}

__global__ __launch_bounds__(256) void k_arrow_forward(const ArrowArgs a) {
    if (a.status->done || a.status->reuse) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int W = a.W, n = a.n, nd = a.nd, np = a.np;
    const int c0 = blockIdx.x * 16, ncol = min(16, np + 1 - c0);
    const double* A = a.A;
    const double* rhs = A + (size_t)n * n;
    double* rd = reinterpret_cast<double*>(tr_lds);
    double* Yd = rd + nd + (nd & 1);
    double* Vs = Yd + (size_t)nd * AR_YS;
    double* Ys = Vs + (size_t)nd * 18;
    double* Lb = Ys + (size_t)9 * W * AR_YS;
    double* Blk = Lb + 3 * AR_LB;                            // [W][18][10]: rows 0-8 D_i, rows 9-17 B_i = M[s_{i+1}, s_i]
    int2* eps = reinterpret_cast<int2*>(Blk + (size_t)W * 180);         // [nd]
    int* eoff = reinterpret_cast<int*>(eps + nd + 2);                   // [W+1]
    int* elist = eoff + ((W + 2) & ~1) + 2;                             // [2 nd]
    int* bad_lds = elist + 2 * nd + 2;
    double* Cs = Lb - 0;                 // set below
    Cs = reinterpret_cast<double*>(tr_lds) + arrow_forward_lds_doubles(W, nd) - 92 - (4 * AR_LB + 92);
    int* vrows_lds = bad_lds + 1;        // 1 = some epoch couples to a non-velocity speed-bias row
    if (tid == 0) { *bad_lds = 0; *vrows_lds = 0; }
    AR_STAMP(0);
    for (int e = tid; e < nd; e += 256) eps[e] = a.ep_slots[e];
    for (int i = tid; i <= W; i += 256) eoff[i] = a.ep_off[i];
    __syncthreads();
    for (int t = tid; t < eoff[W]; t += 256) elist[t] = a.ep_list[t];
    // (1) epochs: pivots
    for (int e = tid; e < nd; e += 256) {
        const double m = A[(size_t)e * n + e];
        if (!(m > 0.0) || !isfinite(m)) { *bad_lds = 1; rd[e] = 0.0; } else rd[e] = rsqrt(m);
    }
    __syncthreads();
    AR_STAMP(1);
    // (2) epoch rows of Y for this workgroup's columns, and the epoch columns of the speed-bias rows.
    //     All global reads of this kernel are issued as batches of independent loads (16 per lane) before anything
    //     is done with them: the kernel is latency-, not bandwidth-bound.
    const size_t prow = (size_t)(nd + 9 * W);            // first pose row of A
    for (int e = tid; e < nd; e += 256) {
        double v[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) { const int col = c0 + c; v[c] = c < ncol ? (col < np ? A[(prow + col) * n + e] : rhs[e]) : 0.0; }
        const double re = rd[e];
#pragma unroll
        for (int c = 0; c < 16; ++c) Yd[e * AR_YS + c] = v[c] * re;
    }
    for (int e = tid; e < nd; e += 256) {
        const int2 sl = eps[e];
        double v[18];
#pragma unroll
        for (int q = 0; q < 18; ++q) { const int s1 = q < 9 ? sl.x : sl.y; v[q] = s1 >= 0 ? A[(size_t)(nd + 9 * s1 + (q < 9 ? q : q - 9)) * n + e] : 0.0; }
        const double re = rd[e];
#pragma unroll
        for (int q = 0; q < 18; ++q) { Vs[e * 18 + q] = v[q] * re; if (v[q] != 0.0 && (q % 9) >= 3) *vrows_lds = 1; }
    }
    AR_STAMP(2);
    // (3) raw speed-bias rows of [M_ep | b_e] and raw chain blocks -> LDS
    for (int k = tid; k < 9 * W; k += 256) {
        double v[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) { const int col = c0 + c; v[c] = c < ncol ? (col < np ? A[(prow + col) * n + nd + k] : rhs[nd + k]) : 0.0; }
#pragma unroll
        for (int c = 0; c < 16; ++c) Ys[k * AR_YS + c] = v[c];
    }
    AR_STAMP(30);
    for (int q = tid; q < W * 18; q += 256) {            // one lane per row of a block: 9 contiguous doubles
        const int i = q / 18, r = q - 18 * i;
        const bool live = r < 9 || i + 1 < W;
        const double* src = A + (size_t)(nd + 9 * i + r) * n + nd + 9 * i;      // r >= 9 runs into the rows of s_{i+1}
        double v[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) v[j] = (live && (r >= 9 || j <= r)) ? src[j] : 0.0;
#pragma unroll
        for (int j = 0; j < 9; ++j) Blk[i * 180 + r * 10 + j] = v[j];
    }
    AR_STAMP(31);
    __syncthreads();
    AR_STAMP(32);
    //     ... minus the epoch contribution (LDS only).  One work item per matrix entry, accumulated in a register over
    //     the epochs of its keyframe in list order (fixed order -> reproducible).  A clock-drift epoch couples to the
    //     velocity rows only (first 3 of the 9 speed-bias components): when phase (2) saw no other non-zero entry the
    //     items are restricted to those rows.
    const int qn = *vrows_lds ? 9 : 3;
    for (int item = tid; item < W * qn * 16; item += 256) {
        const int c = item & 15, kq = item >> 4, sl = kq / qn, q = kq - sl * qn, k = 9 * sl + q;
        double v = Ys[k * AR_YS + c];
        for (int t = eoff[sl]; t < eoff[sl + 1]; ++t) {
            const int e = elist[t];
            v -= Vs[e * 18 + (eps[e].x == sl ? 0 : 9) + q] * Yd[e * AR_YS + c];
        }
        Ys[k * AR_YS + c] = v;
    }
    AR_STAMP(33);
    for (int item = tid; item < W * 2 * qn * qn; item += 256) {
        const int i = item / (2 * qn * qn), w = item - i * 2 * qn * qn, half = w / (qn * qn), u = w - half * qn * qn, rr = u / qn, j = u - rr * qn;
        const bool rowD = half == 0;
        if ((rowD && j > rr) || (!rowD && i + 1 >= W)) continue;
        double v = Blk[i * 180 + (rowD ? rr : 9 + rr) * 10 + j];
        for (int t = eoff[i]; t < eoff[i + 1]; ++t) {
            const int e = elist[t];
            const int2 sl = eps[e];
            const int side = sl.x == i ? 0 : 9;
            if (rowD) v -= Vs[e * 18 + side + rr] * Vs[e * 18 + side + j];
            else if ((sl.x == i ? sl.y : sl.x) == i + 1) v -= Vs[e * 18 + (9 - side) + rr] * Vs[e * 18 + side + j];
        }
        Blk[i * 180 + (rowD ? rr : 9 + rr) * 10 + j] = v;
    }
    __syncthreads();
    AR_STAMP(3);
    // (4) the chain, eliminated FROM BOTH ENDS (twisted factorisation): blocks 0 .. m-1 top-down by wavefront 0, blocks
    //     W-1 .. m+1 bottom-up by wavefront 2, the meeting block m = W/2 last -- half the sequential depth.  Wavefronts 1
    //     and 3 follow their chain one block behind with the forward substitution of this workgroup's 16 columns.
    const int mid = W / 2, nT = mid, nB = W - 1 - mid, T = nT > nB ? nT : nB;
    double* LbT = Lb;                               // 3 rotating panels of the top chain
    double* LbB = Cs + 92;                          // 3 rotating panels of the bottom chain
    double* Lm = LbB + 3 * AR_LB;                   // the meeting block
    double* CsB = Lm + AR_LB;                       // correction of D_mid from the bottom chain (Cs: from the top chain)
    double av[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) av[j] = 0.0;
    if (lane < 9) {
        if (wv == 0 && nT > 0) { for (int j = 0; j < 9; ++j) av[j] = Blk[lane * 10 + j]; }
        if (wv == 2 && nB > 0) { for (int j = 0; j < 9; ++j) av[j] = Blk[(W - 1) * 180 + lane * 10 + j]; }
    }
    bool bad = false;
    int pr = 0, pj = 0;                             // lane -> pair (pr, pj <= pr) of the rank-9 update, lanes 0..44
    { int p = lane < 45 ? lane : 0; while ((pr + 1) * (pr + 2) / 2 <= p) ++pr; pj = p - pr * (pr + 1) / 2; }
    for (int it = 0; it <= T + 1; ++it) {
        if (wv == 0) {
            if (it < nT) arrow_chain_step<false>(it, it + 1, true, av, Blk, LbT + (it % 3) * AR_LB, blockIdx.x == 0 ? a.Lblk + (size_t)it * AR_LB : nullptr,
                                                 Cs, lane, pr, pj, bad);
            else if (it == T) {                     // both chains are done: D_mid minus both corrections, factored alone
                if (lane < 9) {
#pragma unroll
                    for (int j = 0; j < 9; ++j) {
                        double v = j <= lane ? Blk[mid * 180 + lane * 10 + j] : 0.0;
                        if (j <= lane && nT > 0) v -= Cs[lane * 10 + j];
                        if (j <= lane && nB > 0) v -= CsB[lane * 10 + j];
                        av[j] = v;
                    }
                }
                arrow_chain_step<false>(mid, mid, false, av, Blk, Lm, blockIdx.x == 0 ? a.Lblk + (size_t)mid * AR_LB : nullptr, Cs, lane, pr, pj, bad);
            }
        } else if (wv == 2) {
            if (it < nB) {
                const int i = W - 1 - it;
                arrow_chain_step<true>(i, i - 1, true, av, Blk, LbB + (it % 3) * AR_LB, blockIdx.x == 0 ? a.Lblk + (size_t)i * AR_LB : nullptr, CsB, lane, pr, pj, bad);
            }
        } else if (wv == 1) {
            if (it >= 1 && it <= nT) {
                const int i = it - 1;
                arrow_y_update(i, i > 0 ? 1 : 0, i - 1, LbT + ((i + 2) % 3) * AR_LB, 0, nullptr, LbT + (i % 3) * AR_LB, Ys, lane);
            } else if (it == T + 1) {
                const double* Pt = nT > 0 ? LbT + ((nT - 1) % 3) * AR_LB : nullptr;
                const double* Pb = nB > 0 ? LbB + ((nB - 1) % 3) * AR_LB : nullptr;
                if (nT > 0 && nB > 0) arrow_y_update(mid, 2, mid - 1, Pt, mid + 1, Pb, Lm, Ys, lane);
                else if (nT > 0) arrow_y_update(mid, 1, mid - 1, Pt, 0, nullptr, Lm, Ys, lane);
                else if (nB > 0) arrow_y_update(mid, 1, mid + 1, Pb, 0, nullptr, Lm, Ys, lane);
                else arrow_y_update(mid, 0, 0, nullptr, 0, nullptr, Lm, Ys, lane);
            }
        } else {
            if (it >= 1 && it <= nB) {
                const int sB = it - 1, i = W - 1 - sB;
                arrow_y_update(i, sB > 0 ? 1 : 0, i + 1, LbB + ((sB + 2) % 3) * AR_LB, 0, nullptr, LbB + (sB % 3) * AR_LB, Ys, lane);
            }
        }
        __syncthreads();
    }
    if (bad && lane == 0) *bad_lds = 1;
    AR_STAMP(4);
    __syncthreads();
    // (5) publish
    for (int idx = tid; idx < a.K * 16; idx += 256) {
        const int k = idx >> 4, c = idx & 15;
        if (c < ncol) a.Y[(size_t)k * a.ldY + c0 + c] = k < nd ? Yd[k * AR_YS + c] : Ys[(k - nd) * AR_YS + c];
    }
    AR_STAMP(5);
    if (tid == 0 && *bad_lds) atomicOr(a.flag, 1);
}

__global__ __launch_bounds__(256) void k_arrow_schur(const ArrowArgs a) {
    if (a.status->done || a.status->reuse) return;
    const int T = (a.np + 1 + 15) / 16;
    const int ta = blockIdx.x / T, tb = blockIdx.x % T;
    if (tb > ta) return;
    __shared__ double Ya[32][17], Yb[32][17];
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int np = a.np, K = a.K;
    double acc = 0.0;
    for (int k0 = 0; k0 < K; k0 += 32) {
        for (int idx = tid; idx < 512; idx += 256) {
            const int rr = idx >> 4, cc = idx & 15, k = k0 + rr;
            Ya[rr][cc] = (k < K && 16 * ta + cc <= np) ? a.Y[(size_t)k * a.ldY + 16 * ta + cc] : 0.0;
            Yb[rr][cc] = (k < K && 16 * tb + cc <= np) ? a.Y[(size_t)k * a.ldY + 16 * tb + cc] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) acc += Ya[kk][ty] * Yb[kk][tx];
        __syncthreads();
    }
    const int ai = 16 * ta + ty, bi = 16 * tb + tx;
    if (ai <= np && bi < np && bi <= ai) {
        const int o = a.nd + 9 * a.W;
        const double base = ai < np ? a.A[(size_t)(o + ai) * a.n + o + bi] : a.A[(size_t)a.n * a.n + o + bi];
        a.Sp[(size_t)ai * np + bi] = base - acc;
    }
}

__host__ __device__ __forceinline__ size_t arrow_solve_extra_doubles(int W, int K) { return (size_t)K + (K & 1) + (size_t)W * AR_LB; }

__global__ __launch_bounds__(TR_THREADS) void k_arrow_solve(const ArrowArgs a) {
    if (a.status->done || a.status->reuse) return;
    if (*a.flag & 1) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int W = a.W, n = a.n, nd = a.nd, np = a.np, K = a.K;
    double *Bp = nullptr, *part = nullptr, *Pk = nullptr, *sD;
    if (a.lds_chol) { Pk = reinterpret_cast<double*>(tr_lds); sD = Pk + pk_doubles(np); }
    else { Bp = reinterpret_cast<double*>(tr_lds); part = Bp + TR_NB * bp_stride(np); sD = part + 16 * 256; }
    double* ylds = sD + (TR_NB + 1) * TR_PS;
    double* red = ylds + np + (np & 1);
    int* flag = reinterpret_cast<int*>(red + 32);
    double* wb = red + 48;                        // [K] w, overwritten by z_d / z_s
    double* Lb = wb + K + (K & 1);                // [W][AR_LB]
    AR_STAMP(8);
    for (int j = tid; j < W * AR_LB; j += TR_THREADS) Lb[j] = a.Lblk[j];
    if (a.lds_chol) {
        for (int i = wv; i <= np; i += TR_WAVES) {
            const double* src = a.Sp + (size_t)i * np;
            double* dst = Pk + pk_off(i);
            for (int j = lane; j <= i && j < np; j += 64) dst[j] = src[j];
        }
        __syncthreads();
        const bool good = chol_packed_lds(Pk, np, sD, flag);
        if (!good) { if (tid == 0) atomicOr(a.flag, 1); return; }
        AR_STAMP(9);
        for (int j = tid; j < np; j += TR_THREADS) ylds[j] = Pk[pk_off(np) + j];
        __syncthreads();
        backsub_packed_lds(Pk, np, ylds, sD);     // ylds = z_p
    } else {
        const bool good = chol_left_looking(a.Sp, np, Bp, part, sD, flag);
        if (!good) { if (tid == 0) atomicOr(a.flag, 1); return; }
        AR_STAMP(9);
        for (int j = tid; j < np; j += TR_THREADS) ylds[j] = a.Sp[(size_t)np * np + j];
        __syncthreads();
        back_substitute(a.Sp, np, ylds, sD);      // ylds = z_p
    }
    AR_STAMP(10);
    // w = y_e - Y_p z_p  (one wavefront per row)
    for (int k = tid; k < K; k += TR_THREADS) {
        const double* yr = a.Y + (size_t)k * a.ldY;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int c = 0;
        for (; c + 4 <= np; c += 4) { s0 += yr[c] * ylds[c]; s1 += yr[c + 1] * ylds[c + 1]; s2 += yr[c + 2] * ylds[c + 2]; s3 += yr[c + 3] * ylds[c + 3]; }
        for (; c < np; ++c) s0 += yr[c] * ylds[c];
        wb[k] = yr[np] - ((s0 + s1) + (s2 + s3));
    }
    __syncthreads();
    AR_STAMP(11);
    // speed-bias chain in reverse elimination order: the meeting block first, then the two halves independently
    // (wavefront 0: blocks mid-1 .. 0, wavefront 1: blocks mid+1 .. W-1):  L_ii^T z_i = w_i - L_{nbr,i}^T z_nbr
    const int mid = W / 2;
    auto chain_back = [&](const int i, const int nbr) {
        const double* Lc = Lb + i * AR_LB;
        double v = lane < 9 ? wb[nd + 9 * i + lane] : 0.0;
        if (nbr >= 0 && lane < 9) {
#pragma unroll
            for (int k = 0; k < 9; ++k) v -= Lc[(9 + k) * 10 + lane] * wb[nd + 9 * nbr + k];
        }
        const double rp = lane < 9 ? Lc[180 + lane] : 1.0;
#pragma unroll
        for (int k = 8; k >= 0; --k) {
            const double zk = readlane_d(v, k) * readlane_d(rp, k);
            if (lane == k) v = zk;
            else if (lane < k) v -= Lc[k * 10 + lane] * zk;
        }
        if (lane < 9) wb[nd + 9 * i + lane] = v;
        GLIO_WAVE_LDS_SYNC();
    };
    if (wv == 0) chain_back(mid, -1);
    __syncthreads();
    if (wv == 0) { for (int i = mid - 1; i >= 0; --i) chain_back(i, i + 1); }
    else if (wv == 1) { for (int i = mid + 1; i < W; ++i) chain_back(i, i - 1); }
    __syncthreads();
    AR_STAMP(12);
    // epochs: z_e = (w_e - sum_s L_se z_s) / L_ee
    for (int e = tid; e < nd; e += TR_THREADS) {
        const double m = a.A[(size_t)e * n + e];
        const double rde = rsqrt(m);
        const int2 sl = a.ep_slots[e];
        double v = wb[e];
        if (sl.x >= 0) {
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                v -= a.A[(size_t)(nd + 9 * sl.x + q) * n + e] * rde * wb[nd + 9 * sl.x + q];
                v -= a.A[(size_t)(nd + 9 * sl.y + q) * n + e] * rde * wb[nd + 9 * sl.y + q];
            }
        }
        a.z[e] = v * rde;
    }
    double bad = 0.0;
    for (int k = tid; k < 9 * W; k += TR_THREADS) { const double v = wb[nd + k]; a.z[nd + k] = v; if (!isfinite(v)) bad = 1.0; }
    for (int c = tid; c < np; c += TR_THREADS) { const double v = ylds[c]; a.z[K + c] = v; if (!isfinite(v)) bad = 1.0; }
    bad = block_max(bad, red);
    AR_STAMP(13);
    if (tid == 0) { if (bad == 0.0) *a.flag = 2; else atomicOr(a.flag, 1); }
}


// ------------------------------------------------------------------------------------------------
// K7c''  keyframe-chain factorisation.  When the prior is block diagonal by keyframe (what the reference's own
// marginalization produces, quirk Q7) and every IMU / GNSS factor couples neighbouring keyframes, M = S H S + mu D^2 in the
// order [clock-drift epochs | keyframe 0 (15) | keyframe 1 | ...] is a diagonal block followed by a block-TRIDIAGONAL
// chain of 15 x 15 blocks: no dense pose block at all.  ONE kernel, one workgroup: the blocks (minus the epoch
// contribution) are staged in LDS; wavefront 0 eliminates keyframes 0 .. m-1, wavefront 2 keyframes W-1 .. m+1 (twisted
// factorisation), the meeting keyframe m = W/2 last.  A chain step holds the 31 x 15 panel [D_i; B_i; rhs_i] one row per
// lane, factors it in 15 register steps (v_readlane multipliers, the pivots one step ahead as a scalar recurrence) -- the
// right-hand side rides along as row 30 -- and takes the rank-15 update of the next block from the stored panel as
// C = X X^T on the matrix core (4 x v_mfma_f64_16x16x4).
// Back substitution runs the two half chains in parallel again.  The result goes where the arrow kernels put theirs
// (a.z, flag 2); a non-positive pivot raises flag 1 and k_tr_factor falls back to the dense factorisation.
// ------------------------------------------------------------------------------------------------
#define KC_NB 15
#define KC_RS 17                       /* row stride of a staged block: odd, so that the 31 row-lanes hit distinct LDS banks */
#define KC_BLK (31 * KC_RS + 17)       /* 31 rows + 15 reciprocal pivots (+2) = 544 doubles per keyframe */
#define KC_THREADS 512

__host__ __device__ __forceinline__ size_t chain_lds_doubles(int W, int nd) {
    return (size_t)(nd + (nd & 1)) * 2 + (size_t)nd * 30 + (size_t)W * KC_BLK + 2 * 288 + (size_t)15 * W + (W & 1) + (size_t)nd + 2 + (size_t)(W + 2) / 2 + 1 +
           (size_t)nd + 2 + 2 * ((size_t)nd + 2) + 12 + (size_t)nd + 2;
}

// k_chain_solve<true>: the blocks live in global memory; the LDS holds everything else plus the four-front panels (all E slots, two C00 tiles)
__host__ __device__ __forceinline__ size_t chain_lds_doubles_g(int W, int nd);

// 1 / sqrt(d) for a pivot already known to be positive, finite and far from the denormal range (it is a diagonal entry of
// the Jacobi-scaled, mu-regularised matrix): the hardware estimate y0 (relative error <= 2^-24.2 on gfx950, measured:
// scripts/probe/rsq_accuracy.hip) and ONE third-order step, y = y0 (1 + e / 2 + 3 e^2 / 8) with e = 1 - d y0^2 -- residual
// ~ 5 e^3 / 16 < 2^-70, so the result is as good as the two Newton steps used before (max 1.25 vs 1.24 ulp over 4 M
// arguments) with FOUR dependent operations behind the estimate instead of six, on the critical path of every pivot
// (and without the library's class checks and rescaling: ~12 dependent operations).
__device__ __forceinline__ double pivot_rsqrt(const double d) {
    const double y0 = __builtin_amdgcn_rsq(d);
    const double e = fma(-d * y0, y0, 1.0);
    return fma(y0 * e, fma(0.375, e, 0.5), y0);
}

// Ordering point between the lanes of one wavefront for what the chain functions hand over through the blocks.  G = false: the blocks live in LDS
// (GLIO_WAVE_LDS_SYNC).  G = true: the blocks live in GLOBAL memory (windows whose blocks do not fit the LDS, k_chain_solve<true>): the stores must
// have been acknowledged (s_waitcnt vmcnt(0)) before another lane reads them back through the compute unit's L1, which is what a workgroup-scope
// release / acquire pair over all address spaces compiles to.
template <bool G>
__device__ __forceinline__ void chain_wave_sync() {
    if (G) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
    else GLIO_WAVE_LDS_SYNC();
}

template <bool DOWN, bool G = false>
__device__ __forceinline__ void chain_step15(const int i, const int nb, const bool has_nb, double (&av)[KC_NB], double* Blk, double* Cs, const int lane, bool& bad,
                                             long long* ph = nullptr, const double* Nrows = nullptr) {
#ifdef GLIO_DEV_STAMPS
#define CS_CLK(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define CS_PH(k) do { long long t_; CS_CLK(t_); if (ph) ph[k] += t_ - tprev; tprev = t_; } while (0)
    long long tprev; CS_CLK(tprev);
#else
#define CS_PH(k) do { } while (0)
#endif
    const int r = lane;
    // lane-varying conditions as one integer limit per lane (j < lim): `a && b || c` on lane-varying operands compiles to
    // exec-mask branches per element, which cost more than the arithmetic they guard
    const int lim = r == 30 ? KC_NB : (r < KC_NB ? r + 1 : 0);         // columns of the next block this lane carries
    const int tri = r < KC_NB ? r + 1 : KC_NB;                          // columns of its own row that are not above the diagonal
    double* Bi = Blk + (size_t)i * KC_BLK;
    double nx[KC_NB];
    {
        const double* Bn = Blk + (size_t)(has_nb ? nb : i) * KC_BLK;
        // unconditional reads at clamped (always valid) addresses, selected afterwards: the 30 ds_reads go out back to back
        const int row = r < KC_NB ? r : 30;
        const int rb = (r >= KC_NB && r < 30) ? r : KC_NB;
        double ld[KC_NB];
#pragma unroll
        for (int j = 0; j < KC_NB; ++j) nx[j] = Bn[row * KC_RS + j];
#pragma unroll
        for (int j = 0; j < KC_NB; ++j) ld[j] = DOWN ? Bn[(KC_NB + j) * KC_RS + (rb - KC_NB)] : (Nrows ? Nrows[(rb - KC_NB) * KC_RS + j] : Bi[rb * KC_RS + j]);
        const bool keep_nx = has_nb & (lim > 0);
        const bool is_b = (r >= KC_NB) & (r < 30);
#pragma unroll
        for (int j = 0; j < KC_NB; ++j) nx[j] = keep_nx ? nx[j] : 0.0;
#pragma unroll
        for (int j = 0; j < KC_NB; ++j) av[j] = is_b ? (has_nb ? ld[j] : 0.0) : av[j];
    }
    CS_PH(0);
    double rpv = 0.0;
    // The pivots form a scalar recurrence d_{j+1} = a_{j+1,j+1} - (a_{j+1,j} / sqrt(d_j))^2 that is carried in the uniform
    // domain one step ahead of the vector update: the next 1/sqrt starts three dependent operations after the previous
    // one instead of waiting for the column scaling, the rank-1 update and a v_readlane round trip.  A non-positive or
    // non-finite pivot only raises `bad` (the dense fallback redoes the solve); it is not patched on the critical path.
    double djj = readlane_d(av[0], 0);
#pragma unroll
    for (int j = 0; j < KC_NB; ++j) {
        bad |= !((djj > 0.0) & (djj < 1e300));                       // also true for NaN
        const double rdj = pivot_rsqrt(djj);
        if (j + 1 < KC_NB) {
            const double lnx = readlane_d(av[j], j + 1) * rdj;        // L[j+1][j]
            djj = fma(-lnx, lnx, readlane_d(av[j + 1], j + 1));       // next pivot (both operands were final before this step)
        }
        const double lij = av[j] * rdj;                              // lane j: d / sqrt(d) = sqrt(d)
        rpv = lane == j ? rdj : rpv;
        av[j] = lij;
#pragma unroll
        for (int c = j + 1; c < KC_NB; ++c) av[c] -= lij * readlane_d(lij, c);
    }
    CS_PH(1);
    if (r < 31) {
#pragma unroll
        for (int j = 0; j < KC_NB; ++j) Bi[r * KC_RS + j] = j < tri ? av[j] : 0.0;
        if (r < KC_NB) Bi[31 * KC_RS + r] = rpv;
    }
    chain_wave_sync<G>();
    CS_PH(2);
    if (has_nb) {
        // C = X X^T on the matrix core: X = rows 15..30 of the factored panel (15 rows towards the next keyframe + the
        // right-hand side row), 16 x 15.  v_mfma_f64_16x16x4 takes A[i][k] from lane (i + 16 k) and B[k][j] from lane
        // (j + 16 k): for X X^T both are X[lane % 16][kb + lane / 16], so one LDS read per lane feeds both operands; four
        // instructions cover k = 0..15 (k = 15 is padding: zero).  C: lane l, register q -> row (l >> 4) + 4 q, column l & 15.
        // (Measured slower, round 3: pulling the operands straight out of the row-per-lane registers with 30 ds_bpermute instead of reading
        // the stored panel back -- the pulls cost more than the store-wait-load round trip they avoid: 69.8 vs 68.0 us per step.)
        {
            const int xi = lane & 15, xg = lane >> 4;
            const double* xrow = Bi + (KC_NB + xi) * KC_RS + xg;
            double xv[4];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) xv[kb] = xrow[4 * kb];
            xv[3] = xg == 3 ? 0.0 : xv[3];
            v4f64 cacc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[kb], xv[kb], cacc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) Cs[(xg + 4 * q) * KC_RS + xi] = cacc[q];
        }
        GLIO_WAVE_LDS_SYNC();
        CS_PH(3);
        {
            const int row = r < KC_NB ? r : 15;              // lanes that hold nothing read row 15 too and discard it
            double cv[KC_NB];
#pragma unroll
            for (int j = 0; j < KC_NB; ++j) cv[j] = Cs[row * KC_RS + j];
#pragma unroll
            for (int j = 0; j < KC_NB; ++j) nx[j] -= j < lim ? cv[j] : 0.0;
        }
    }
#pragma unroll
    for (int j = 0; j < KC_NB; ++j) av[j] = j < lim ? nx[j] : 0.0;
    CS_PH(4);
}

// Preparation of a factored chain block for the back substitution, by a wavefront that has nothing else to do while the
// two chain wavefronts work on the next keyframes: with L (rows 0..14), X = B L^-T (rows 15..29, row = neighbour's unknown)
// and y = L^-1 r (row 30) in place, lane c < 15 solves L^T m = X[c][:] and lane 15 solves L^T m = y (15 sequential steps
// each, L broadcast from LDS); rows 15..29 then hold M = L^-T X^T (row = OWN unknown) and row 30 holds w = L^-T y, so that
// the back substitution of this keyframe is one matrix-vector product, z = w - M z_neighbour, instead of a 15-step
// triangular solve on the critical path.
template <bool G = false>
__device__ __forceinline__ void chain_prepare_back(double* Bi, const int lane) {
    const int c = lane < 16 ? lane : 0;
    const double* xr = Bi + (c < KC_NB ? KC_NB + c : 30) * KC_RS;
    double m[KC_NB];
#pragma unroll
    for (int k = 0; k < KC_NB; ++k) m[k] = xr[k];
#pragma unroll
    for (int k = KC_NB - 1; k >= 0; --k) {
        double s0 = m[k], s1 = 0.0;
#pragma unroll
        for (int j = k + 1; j < KC_NB; ++j) { if ((j - k) & 1) s0 -= Bi[j * KC_RS + k] * m[j]; else s1 -= Bi[j * KC_RS + k] * m[j]; }
        m[k] = (s0 + s1) * Bi[31 * KC_RS + k];
    }
    chain_wave_sync<G>();                                  // every lane has read its right-hand side before anyone overwrites the rows
    if (lane < KC_NB) {
#pragma unroll
        for (int r = 0; r < KC_NB; ++r) Bi[(KC_NB + r) * KC_RS + lane] = m[r];
    } else if (lane == KC_NB) {
#pragma unroll
        for (int r = 0; r < KC_NB; ++r) Bi[30 * KC_RS + r] = m[r];
    }
}

// ---- four fronts (k_chain_step, W >= KC_F4_MIN_W).  The middle keyframe s = W / 2 is a separator: the chain falls into a left segment 0 .. s-1
// and a right segment s+1 .. W-1, and each segment is eliminated from BOTH of its ends at once.  The outer fronts (A: 0 upwards, D: W-1
// downwards) are the steps above.  The inner fronts start beside the separator (B: s-1 downwards, C: s+1 upwards); eliminating their blocks
// fills in a coupling E_i = M[s, i] between the separator and the front's current block, which the front carries along as 15 more rows of
// its panel (lanes 32..46): [D_i; N_i; rhs_i; E_i] is 46 x 15, still one row per lane, the same 15 pivots.  Its Schur complement
//   [Xn; y; Xe] [Xn; y; Xe]^T  =  three 16 x 16 tiles:   C00 = [Xn; y][Xn; y]^T  -> D and rhs of the next block (as in chain_step15),
//                                                        C10 = Xe [Xn; y]^T      -> minus the next block's E rows (columns 0..14) and the
//                                                                                   separator's rhs (column 15, summed over the steps),
//                                                        C11 = Xe Xe^T           -> the separator's diagonal block, summed over the steps in
//                                                                                   the accumulator registers of the matrix core.
// The two fronts of a segment meet at m (mL, mR): the outer wavefront eliminates it with the inner front's last E rows as its coupling
// rows (chain_step15 with Nrows; its "next" block is the separator).  The separator is factored last, by wavefront 0.  Critical path for
// W = 20: 5 steps + meeting block + separator instead of 10 steps + middle block.  Every hand-over is a release/acquire pair on an LDS word.
#define KC_F4_MIN_W 12
#define KC_ERS 15                      /* row stride of a slot of E rows (odd: the 15 row-lanes hit distinct banks) */
#define KC_ES (15 * KC_ERS + 1)        /* one slot of E rows: 15 x 15 (+1: the matrix-core operand read of the padded k = 15 stays inside) */
#define KC_TILE (16 * KC_RS)           /* one 16 x 16 tile of a Schur complement, row stride KC_RS */
struct ChainSplit { int s, mL, mR, nA, nB, nC, nD; };
__host__ __device__ __forceinline__ ChainSplit chain_f4_split(const int W) {
    ChainSplit c;
    c.s = W / 2;
    c.nB = 4 * (c.s - 1) / 9;                 // inner fronts take a little less than half: their steps are the heavier ones
    c.nC = 4 * (W - 1 - c.s - 1) / 9;
    c.mL = c.s - 1 - c.nB; c.mR = c.s + 1 + c.nC;
    c.nA = c.mL; c.nD = W - 1 - c.mR;
    return c;
}
// One step of an inner front.  av: lanes 0..14 and 30 as in chain_step15, lanes 32..46 the E rows of block i.  Ei: where the factored E rows
// (Xe) of block i go; Cs0 / Cs1: this wavefront's scratch tiles for C00 / C10; c11, racc: the separator's sums (see above).
template <bool DOWN, bool G = false>
__device__ __forceinline__ void chain_step15e(const int i, const int nb, double (&av)[KC_NB], double* Blk, double* Ei, double* Cs0, double* Cs1, const int cs1_stride,
                                              const int lane, bool& bad, v4f64& c11, v4f64& racc) {
    const int r = lane;
    const int lim = r == 30 ? KC_NB : (r < KC_NB ? r + 1 : 0);
    const int tri = r < KC_NB ? r + 1 : KC_NB;
    const bool is_e = (r >= 32) & (r < 32 + KC_NB);
    double* Bi = Blk + (size_t)i * KC_BLK;
    double nx[KC_NB];
    {
        const double* Bn = Blk + (size_t)nb * KC_BLK;
        const int row = r < KC_NB ? r : 30;
        const int rb = (r >= KC_NB && r < 30) ? r : KC_NB;
        double ld[KC_NB];
#pragma unroll
        for (int j = 0; j < KC_NB; ++j) nx[j] = Bn[row * KC_RS + j];
#pragma unroll
        for (int j = 0; j < KC_NB; ++j) ld[j] = DOWN ? Bn[(KC_NB + j) * KC_RS + (rb - KC_NB)] : Bi[rb * KC_RS + j];
        const bool keep_nx = lim > 0;
        const bool is_b = (r >= KC_NB) & (r < 30);
#pragma unroll
        for (int j = 0; j < KC_NB; ++j) nx[j] = keep_nx ? nx[j] : 0.0;
#pragma unroll
        for (int j = 0; j < KC_NB; ++j) av[j] = is_b ? ld[j] : av[j];
    }
    double rpv = 0.0;
    double djj = readlane_d(av[0], 0);
#pragma unroll
    for (int j = 0; j < KC_NB; ++j) {
        bad |= !((djj > 0.0) & (djj < 1e300));
        const double rdj = pivot_rsqrt(djj);
        if (j + 1 < KC_NB) {
            const double lnx = readlane_d(av[j], j + 1) * rdj;
            djj = fma(-lnx, lnx, readlane_d(av[j + 1], j + 1));
        }
        const double lij = av[j] * rdj;
        rpv = lane == j ? rdj : rpv;
        av[j] = lij;
#pragma unroll
        for (int c = j + 1; c < KC_NB; ++c) av[c] -= lij * readlane_d(lij, c);
    }
    if (r < 31) {
#pragma unroll
        for (int j = 0; j < KC_NB; ++j) Bi[r * KC_RS + j] = j < tri ? av[j] : 0.0;
        if (r < KC_NB) Bi[31 * KC_RS + r] = rpv;
    } else if (is_e) {
#pragma unroll
        for (int j = 0; j < KC_NB; ++j) Ei[(r - 32) * KC_ERS + j] = av[j];
    }
    chain_wave_sync<G>();
    {
        const int xi = lane & 15, xg = lane >> 4;
        const double* x0row = Bi + (KC_NB + xi) * KC_RS + xg;
        const double* x1row = Ei + (xi < KC_NB ? xi : 0) * KC_ERS + xg;
        double x0[4], x1[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) { x0[kb] = x0row[4 * kb]; x1[kb] = x1row[4 * kb]; }
        x0[3] = xg == 3 ? 0.0 : x0[3];
        x1[3] = xg == 3 ? 0.0 : x1[3];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) x1[kb] = xi == KC_NB ? 0.0 : x1[kb];
        v4f64 c00 = {0.0, 0.0, 0.0, 0.0}, c10 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[kb], x0[kb], c00, 0, 0, 0);
            c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[kb], x0[kb], c10, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[kb], x1[kb], c11, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            Cs0[(xg + 4 * q) * KC_RS + xi] = c00[q];
            if (xg + 4 * q < KC_NB && xi < KC_NB) Cs1[(xg + 4 * q) * cs1_stride + xi] = c10[q];       // (the scratch is the next block's E rows: 15 x 15 only)
            racc[q] += xi == KC_NB ? c10[q] : 0.0;           // column 15 of C10 = Xe y^T: rows (xg + 4 q) of the separator's right-hand side
        }
    }
    chain_wave_sync<G>();
    {
        const double* cp = is_e ? Cs1 + (r - 32) * cs1_stride : Cs0 + (r < KC_NB ? r : 15) * KC_RS;
        double cv[KC_NB];
#pragma unroll
        for (int j = 0; j < KC_NB; ++j) cv[j] = cp[j];
#pragma unroll
        for (int j = 0; j < KC_NB; ++j) av[j] = is_e ? -cv[j] : (j < lim ? nx[j] - cv[j] : 0.0);
    }
}

// chain_prepare_back for a block of an inner front: lanes 16..30 transform its E rows the same way (Ei then holds Me = L^-T Xe^T, row = OWN
// unknown), so that z_i = w_i - M_i z_neighbour - Me_i z_s
template <bool G = false>
__device__ __forceinline__ void chain_prepare_back_e(double* Bi, double* Ei, const int lane) {
    const int c = lane < 31 ? lane : 15;
    const double* xr = c < KC_NB ? Bi + (KC_NB + c) * KC_RS : (c == KC_NB ? Bi + 30 * KC_RS : Ei + (c - 16) * KC_ERS);
    double m[KC_NB];
#pragma unroll
    for (int k = 0; k < KC_NB; ++k) m[k] = xr[k];
#pragma unroll
    for (int k = KC_NB - 1; k >= 0; --k) {
        double s0 = m[k], s1 = 0.0;
#pragma unroll
        for (int j = k + 1; j < KC_NB; ++j) { if ((j - k) & 1) s0 -= Bi[j * KC_RS + k] * m[j]; else s1 -= Bi[j * KC_RS + k] * m[j]; }
        m[k] = (s0 + s1) * Bi[31 * KC_RS + k];
    }
    chain_wave_sync<G>();
    if (lane < KC_NB) {
#pragma unroll
        for (int r = 0; r < KC_NB; ++r) Bi[(KC_NB + r) * KC_RS + lane] = m[r];
    } else if (lane == KC_NB) {
#pragma unroll
        for (int r = 0; r < KC_NB; ++r) Bi[30 * KC_RS + r] = m[r];
    } else if (lane < 31) {
#pragma unroll
        for (int r = 0; r < KC_NB; ++r) Ei[r * KC_ERS + (lane - 16)] = m[r];
    }
}

// Back substitution of one factored (NOT prepared) block by the triangular solve: L_ii^T z_i = y_i - X^T z_nbr (nbr < 0: no neighbour term).
// zout: where z_i goes (default: its place in zb).
template <bool G = false>
__device__ __forceinline__ void chain_back_solve(const double* Blk, const int i, const int nbr, double* zb, double* zout, const int lane) {
    const double* Bi = Blk + (size_t)i * KC_BLK;
    const int ln = lane < KC_NB ? lane : 0;
    double lcol[KC_NB], bcol[KC_NB];             // column `lane` of L_ii and of L_{nbr,i}: fetched before the dependent chain starts
#pragma unroll
    for (int k = 0; k < KC_NB; ++k) { lcol[k] = Bi[k * KC_RS + ln]; bcol[k] = Bi[(KC_NB + k) * KC_RS + ln]; }
    const double rp = lane < KC_NB ? Bi[31 * KC_RS + lane] : 1.0;
    double v = lane < KC_NB ? Bi[30 * KC_RS + lane] : 0.0;
    if (nbr >= 0) {
        double s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
        for (int k = 0; k < KC_NB; k += 3) { s0 += bcol[k] * zb[15 * nbr + k]; s1 += bcol[k + 1] * zb[15 * nbr + k + 1]; s2 += bcol[k + 2] * zb[15 * nbr + k + 2]; }
        v -= (s0 + s1) + s2;
    }
#pragma unroll
    for (int k = KC_NB - 1; k >= 0; --k) {
        const double zk = readlane_d(v, k) * readlane_d(rp, k);
        if (lane == k) v = zk;
        else if (lane < k) v -= lcol[k] * zk;
    }
    if (lane < KC_NB) (zout ? zout : zb + 15 * i)[lane] = v;
    GLIO_WAVE_LDS_SYNC();
}

// LDS of the four-front elimination: the E slots (es0: the first k0 of them, es1: the rest) and the inner fronts' C00 tiles
struct ChainF4Mem {
    double* es0; double* es1; int k0; double* cs1a; double* cs3a;
    __device__ __forceinline__ double* eslot(const int k) const { return k < k0 ? es0 + (size_t)k * KC_ES : es1 + (size_t)(k - k0) * KC_ES; }
};

// The four-front elimination itself (see above), by the eight wavefronts of the workgroup: fronts A, B, D, C on wavefronts 0, 1, 2, 3; wavefronts
// 4..7 follow one front each and prepare its factored blocks for the back substitution.  NO workgroup barrier inside: every hand-over is a
// release / acquire pair on a word of s_prog (zeroed by the caller, behind a barrier).  Ends with z of the separator in zb (wavefront 0); the
// caller's barrier follows.  `bad` collects non-positive pivots per thread.
template <bool G>
__device__ __forceinline__ void chain_f4_factor(const int W, const ChainSplit& cs, double* Blk, double* CsT, double* CsB, const ChainF4Mem& M, double* zb, int* s_prog,
                                                const int lane, const int wv, bool& bad, long long* ph, long long* dbg) {
#ifdef GLIO_DEV_STAMPS
#define F4_STAMP(k) do { if (dbg && blockIdx.x == 0 && lane == 0) dbg[k] = wall_clock64(); } while (0)
#else
#define F4_STAMP(k) do { } while (0)
#endif
    const int s = cs.s, mL = cs.mL, mR = cs.mR;
    double av[KC_NB];
#pragma unroll
    for (int j = 0; j < KC_NB; ++j) av[j] = 0.0;
    if (lane < KC_NB || lane == 30) {
        const int row = lane < KC_NB ? lane : 30;
        const int first = wv == 0 ? 0 : (wv == 1 ? s - 1 : (wv == 2 ? W - 1 : s + 1));
        if (wv < 4) { for (int j = 0; j < KC_NB; ++j) av[j] = Blk[(size_t)first * KC_BLK + row * KC_RS + j]; }
    } else if (lane >= 32 && lane < 32 + KC_NB) {
        // the inner fronts' first E rows: the separator's own coupling to its neighbour.  B: E = B_{s-1} (rows = unknowns of s);
        // C: E = B_s^T (B_s has the unknowns of s + 1 as rows)
        const int e = lane - 32;
        if (wv == 1) { for (int j = 0; j < KC_NB; ++j) av[j] = Blk[(size_t)(s - 1) * KC_BLK + (KC_NB + e) * KC_RS + j]; }
        if (wv == 3) { for (int j = 0; j < KC_NB; ++j) av[j] = Blk[(size_t)s * KC_BLK + (KC_NB + j) * KC_RS + e]; }
    }
    // (Workgroup-scope atomics on the __shared__ words: ds_write / ds_read.  A cast to `volatile int*` drops the address space -- the accesses
    //  became FLAT, system scope, each followed by s_waitcnt vmcnt(0).)  With the blocks in global memory the fences cover all address spaces.
    auto publish = [&](const int f, const int v) {
        if (G) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        if (lane == 0) __hip_atomic_store(&s_prog[f], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto await = [&](const int f, const int v, const int nap) {
        while (__hip_atomic_load(&s_prog[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < v) { if (nap == 1) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(2); }
        if (G) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    };
    // the fronts are the critical path: on the SIMD they share with a wavefront that prepares blocks for the back substitution they issue first
    if (wv < 4) __builtin_amdgcn_s_setprio(3);
    const int lim = lane == 30 ? KC_NB : (lane < KC_NB ? lane + 1 : 0);
    const int crow = lane < KC_NB ? lane : 15;
    // av -= rows of a C00-shaped tile (rows 0..14: the next block's diagonal block, row 15: its right-hand side)
    auto minus_c00 = [&](const double* tile) {
        double c[KC_NB];
#pragma unroll
        for (int j = 0; j < KC_NB; ++j) c[j] = tile[crow * KC_RS + j];
#pragma unroll
        for (int j = 0; j < KC_NB; ++j) av[j] = j < lim ? av[j] - c[j] : 0.0;
    };
    // What an inner front leaves behind after its last step.  The meeting block's E rows go where the outer wavefront's step reads its
    // coupling rows (left: rows 15..29 of block mL, the B_mL this front read in the step just finished; right: rows 15..29 of the separator's
    // block, the B_s this front read before its first step) -- the last step had its C10 written there, here it is negated in place.  The
    // separator's sums are subtracted from its staged block in place, front C first (front B waits for it): the left meeting step reads the
    // separator's rows after both.
    auto inner_done = [&](double* Em, const v4f64& c11, const v4f64& racc) {
        if (lane >= 32 && lane < 32 + KC_NB) {
#pragma unroll
            for (int j = 0; j < KC_NB; ++j) Em[(lane - 32) * KC_RS + j] = av[j];
        }
        double* Bs = Blk + (size_t)s * KC_BLK;
        const int xi = lane & 15, xg = lane >> 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = xg + 4 * q;
            if (e < KC_NB && xi <= e) Bs[e * KC_RS + xi] -= c11[q];
            else if (e < KC_NB && xi == KC_NB) Bs[30 * KC_RS + e] -= racc[q];
        }
    };
    double* EmL = Blk + (size_t)mL * KC_BLK + KC_NB * KC_RS;
    double* EmR = Blk + (size_t)s * KC_BLK + KC_NB * KC_RS;
    if (wv == 0) {
        for (int it = 0; it < cs.nA; ++it) { chain_step15<false, G>(it, it + 1, true, av, Blk, CsT, lane, bad, ph); publish(0, it + 1); }
        F4_STAMP(100);
        await(1, cs.nB, 1);
        F4_STAMP(101);
        minus_c00(M.cs1a);
        chain_step15<false, G>(mL, s, true, av, Blk, CsT, lane, bad);                   // (coupling rows: EmL; next block: the separator)
        publish(0, cs.nA + 1);
        F4_STAMP(111);
        await(2, cs.nD + 1, 1);
        F4_STAMP(112);
        minus_c00(CsB);
        chain_step15<false, G>(s, s, false, av, Blk, CsT, lane, bad);
        F4_STAMP(102);
        chain_back_solve<G>(Blk, s, -1, zb, nullptr, lane);
        F4_STAMP(103);
    } else if (wv == 2) {
        for (int it = 0; it < cs.nD; ++it) { const int i = W - 1 - it; chain_step15<true, G>(i, i - 1, true, av, Blk, CsB, lane, bad); publish(2, it + 1); }
        F4_STAMP(114);
        await(3, cs.nC, 1);
        minus_c00(M.cs3a);
        // (next block = itself: the rows a step prefetches for its successor are not used here -- wavefront 0 forms the separator's block -- and
        //  the separator's staged rows are still being updated by front B)
        chain_step15<false, G>(mR, mR, true, av, Blk, CsB, lane, bad, nullptr, EmR);
        publish(2, cs.nD + 1);
        F4_STAMP(115);
    } else if (wv == 1) {
        v4f64 c11 = {0.0, 0.0, 0.0, 0.0}, racc = {0.0, 0.0, 0.0, 0.0};
        for (int it = 0; it < cs.nB; ++it) {
            const int i = s - 1 - it;
            const bool last = it == cs.nB - 1;
            chain_step15e<true, G>(i, i - 1, av, Blk, M.eslot(it), M.cs1a, last ? EmL : M.eslot(it + 1), last ? KC_RS : KC_ERS, lane, bad, c11, racc);
            if (last) { await(3, cs.nC, 1); inner_done(EmL, c11, racc); }
            publish(1, it + 1);
        }
        F4_STAMP(113);
    } else if (wv == 3) {
        v4f64 c11 = {0.0, 0.0, 0.0, 0.0}, racc = {0.0, 0.0, 0.0, 0.0};
        for (int it = 0; it < cs.nC; ++it) {
            const int i = s + 1 + it;
            const bool last = it == cs.nC - 1;
            chain_step15e<false, G>(i, i + 1, av, Blk, M.eslot(cs.nB + it), M.cs3a, last ? EmR : M.eslot(cs.nB + it + 1), last ? KC_RS : KC_ERS, lane, bad, c11, racc);
            if (last) inner_done(EmR, c11, racc);
            publish(3, it + 1);
        }
        F4_STAMP(116);
    } else if (wv == 4) {
        for (int k = 0; k < cs.nA; ++k) { await(0, k + 1, 2); chain_prepare_back<G>(Blk + (size_t)k * KC_BLK, lane); }
    } else if (wv == 5) {
        for (int k = 0; k < cs.nB; ++k) { await(1, k + 1, 2); chain_prepare_back_e<G>(Blk + (size_t)(s - 1 - k) * KC_BLK, M.eslot(k), lane); }
    } else if (wv == 6) {
        for (int k = 0; k < cs.nD; ++k) { await(2, k + 1, 2); chain_prepare_back<G>(Blk + (size_t)(W - 1 - k) * KC_BLK, lane); }
    } else if (wv == 7) {
        for (int k = 0; k < cs.nC; ++k) { await(3, k + 1, 2); chain_prepare_back_e<G>(Blk + (size_t)(s + 1 + k) * KC_BLK, M.eslot(cs.nB + k), lane); }
    }
    if (wv < 4) __builtin_amdgcn_s_setprio(0);
#undef F4_STAMP
}

// Back substitution behind chain_f4_factor (and the caller's barrier): z of the separator is known; every wavefront 0..3 starts from its segment's
// meeting block (the inner ones compute it privately, into their dead C00 tile).  The meeting blocks are solved by the triangular back
// substitution like the separator: preparing them as matrix-vector products would have to happen after the last elimination step, on the
// critical path.  All other blocks: z_i = w_i - M_i z_neighbour - Me_i z_s on what chain_prepare_back(_e) left.
template <bool G>
__device__ __forceinline__ void chain_f4_backsub(const int W, const ChainSplit& cs, const double* Blk, const ChainF4Mem& M, double* zb, const int lane, const int wv) {
    const int s = cs.s, mL = cs.mL, mR = cs.mR;
    auto mv = [&](const int i, const double* znb, const double* Me, double* zout) {
        const double* Bi = Blk + (size_t)i * KC_BLK;
        const double* zs = zb + 15 * s;
        if (lane < KC_NB) {
            double mrow[KC_NB], zn[KC_NB];
#pragma unroll
            for (int k = 0; k < KC_NB; ++k) { mrow[k] = Bi[(KC_NB + lane) * KC_RS + k]; zn[k] = znb[k]; }
            double s0 = Bi[30 * KC_RS + lane], s1 = 0, s2 = 0;
#pragma unroll
            for (int k = 0; k < KC_NB; k += 3) { s0 -= mrow[k] * zn[k]; s1 -= mrow[k + 1] * zn[k + 1]; s2 -= mrow[k + 2] * zn[k + 2]; }
            if (Me) {
#pragma unroll
                for (int k = 0; k < KC_NB; ++k) { mrow[k] = Me[lane * KC_ERS + k]; zn[k] = zs[k]; }
#pragma unroll
                for (int k = 0; k < KC_NB; k += 3) { s0 -= mrow[k] * zn[k]; s1 -= mrow[k + 1] * zn[k + 1]; s2 -= mrow[k + 2] * zn[k + 2]; }
            }
            zout[lane] = (s0 + s1) + s2;
        }
        GLIO_WAVE_LDS_SYNC();
    };
    // one half chain: blocks i0, i0 + di, ... (cnt of them), each with its predecessor in the sequence as neighbour (the first one: zfirst)
    auto run = [&](const int i0, const int di, const int cnt, const double* zfirst, const bool with_e, const int slot0, const int dslot) {
        if (!G) {
            for (int k = 0; k < cnt; ++k) {
                const int i = i0 + k * di;
                mv(i, k == 0 ? zfirst : zb + 15 * (i - di), with_e ? M.eslot(slot0 + k * dslot) : nullptr, zb + 15 * i);
            }
            return;
        }
        // blocks in global memory: the rows of M_i do not depend on z -- those of the next block are in flight while this one is multiplied
        // (one global round trip per block on the critical path otherwise)
        const int ln = lane < KC_NB ? lane : 0;
        double cur[KC_NB + 1], nxt[KC_NB + 1];
        auto fetch = [&](const int i, double (&m)[KC_NB + 1]) {
            const double* Bi = Blk + (size_t)i * KC_BLK;
#pragma unroll
            for (int k = 0; k < KC_NB; ++k) m[k] = Bi[(KC_NB + ln) * KC_RS + k];
            m[KC_NB] = Bi[30 * KC_RS + ln];
        };
        auto compute = [&](const int k, const double (&m)[KC_NB + 1]) {
            const int i = i0 + k * di;
            const double* znb = k == 0 ? zfirst : zb + 15 * (i - di);
            const double* Me = with_e ? M.eslot(slot0 + k * dslot) : nullptr;
            const double* zs = zb + 15 * s;
            if (lane < KC_NB) {
                double zn[KC_NB], mrow[KC_NB];
#pragma unroll
                for (int q = 0; q < KC_NB; ++q) zn[q] = znb[q];
                double s0 = m[KC_NB], s1 = 0, s2 = 0;
#pragma unroll
                for (int q = 0; q < KC_NB; q += 3) { s0 -= m[q] * zn[q]; s1 -= m[q + 1] * zn[q + 1]; s2 -= m[q + 2] * zn[q + 2]; }
                if (Me) {
#pragma unroll
                    for (int q = 0; q < KC_NB; ++q) { mrow[q] = Me[lane * KC_ERS + q]; zn[q] = zs[q]; }
#pragma unroll
                    for (int q = 0; q < KC_NB; q += 3) { s0 -= mrow[q] * zn[q]; s1 -= mrow[q + 1] * zn[q + 1]; s2 -= mrow[q + 2] * zn[q + 2]; }
                }
                zb[15 * i + lane] = (s0 + s1) + s2;
            }
            GLIO_WAVE_LDS_SYNC();
        };
        // (two buffers used alternately, the loop unrolled by hand: a copy `cur = nxt` at the end of an iteration would wait for the prefetch)
        if (cnt > 0) fetch(i0, cur);
        for (int k = 0; k < cnt; k += 2) {
            if (k + 1 < cnt) fetch(i0 + (k + 1) * di, nxt);
            compute(k, cur);
            if (k + 1 < cnt) {
                if (k + 2 < cnt) fetch(i0 + (k + 2) * di, cur);
                compute(k + 1, nxt);
            }
        }
    };
    if (wv == 0) {
        chain_back_solve<G>(Blk, mL, s, zb, nullptr, lane);
        run(mL - 1, -1, mL, zb + 15 * mL, false, 0, 0);
    } else if (wv == 1) {
        chain_back_solve<G>(Blk, mL, s, zb, M.cs1a, lane);
        run(mL + 1, 1, s - 1 - mL, M.cs1a, true, s - 2 - mL, -1);
    } else if (wv == 2) {
        chain_back_solve<G>(Blk, mR, s, zb, nullptr, lane);
        run(mR + 1, 1, W - 1 - mR, zb + 15 * mR, false, 0, 0);
    } else if (wv == 3) {
        chain_back_solve<G>(Blk, mR, s, zb, M.cs3a, lane);
        run(mR - 1, -1, mR - 1 - s, M.cs3a, true, cs.nB + mR - 2 - s, -1);
    }
}

// Minus the clock-drift epochs' contribution to the blocks of keyframe i (both chain kernels): with V_q the scaled column of epoch q restricted to
// the rows of keyframe i that carry a coupling (Va, na rows), to those of keyframe i+1 when q couples (i, i+1) (Vb) and y_q its right-hand side,
//   D_i -= Va Va^T,  B_i -= Vb Va^T,  rhs_i -= y Va^T  =  [Va; Vb; y] Va^T summed over the epochs touching i in list order:
// a (2 na + 1) x L by L x na product per keyframe -- on the matrix core, four epochs per v_mfma_f64_16x16x4 with the entries themselves as the
// accumulator (2 na + 1 <= 16).  One wavefront per keyframe: a lane derives the offsets of ITS epoch's rows once, reads one A and one B
// operand per instruction and owns four entries of the result.  (As a flat list of entries, each re-reading the index lists and its 2 L
// operands, this phase was ~1000 wavefront-level LDS reads and 5 us; the products are fused into the accumulation here, so the result
// differs from that form in the last bits -- both chain kernels call this function, which keeps them bit-identical to each other.)
__device__ __forceinline__ void chain_epoch_corrections_mfma(const int W, const int na, const int* misc_rows, const int* eoff, const int* elist, const int2* eps,
                                                             const double* Vs, const double* yd, double* Blk, const int lane, const int wv, const int nwaves, long long* dbg = nullptr) {
#ifdef GLIO_DEV_STAMPS
#define EC_STAMP(k) do { if (dbg && threadIdx.x == 0) dbg[k] = wall_clock64(); } while (0)
#else
#define EC_STAMP(k) do { } while (0)
#endif
    EC_STAMP(104);
    const int ci = lane & 15, ck = lane >> 4;                  // operand roles: row of A / column of B, and the epoch within the group of four
    const int arow = ci < na ? misc_rows[ci] : (ci < 2 * na ? misc_rows[ci - na] : 0);
    const int bcol = ci < na ? misc_rows[ci] : 0;
    const bool colv = ci < na;
    // the four entries of this lane inside a keyframe's block: column ci, rows ck + 4 q
    int eoffs[4]; bool own0[4], isB[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int rr = ck + 4 * q;
        int rowoff; bool ok;
        if (rr < na) { const int r = misc_rows[rr]; rowoff = r; ok = colv && bcol <= r; }                           // D_i, lower triangle
        else if (rr < 2 * na) { rowoff = KC_NB + misc_rows[rr - na]; ok = colv; }                                    // B_i (rows of keyframe i + 1)
        else { rowoff = 30; ok = colv && rr == 2 * na; }                                                           // right-hand side row
        own0[q] = ok; isB[q] = rr >= na && rr < 2 * na;
        eoffs[q] = rowoff * KC_RS + bcol;
    }
    // one group of four epochs of keyframe i: the operands of this lane (A negated: the accumulator loses the product)
    auto operands = [&](const int i, const int t, const int t1, double& a, double& b) {
        const bool live = t < t1;
        const int e = live ? elist[t] : 0;
        const int2 sl = eps[e];
        const int side = sl.x == i ? 0 : 15;
        const int base = e * 30 + side;
        const int oth = (sl.x == i ? sl.y : sl.x) == i + 1 ? e * 30 + (15 - side) : -1;
        const double va = Vs[base + arow], vo = Vs[(oth >= 0 ? oth : base) + arow], vy = yd[e], vb = Vs[base + bcol];
        const double av = ci < na ? va : (ci < 2 * na ? (oth >= 0 ? vo : 0.0) : (ci == 2 * na ? vy : 0.0));
        a = live ? -av : 0.0; b = (live && colv) ? vb : 0.0;
    };
    // Three keyframes of a wavefront in flight: every stage below is a batch of independent LDS reads (their dependent chain -- list entry,
    // slot pair, operands -- is otherwise paid once per keyframe and group)
    constexpr int NK = 3;
    EC_STAMP(105);
    for (int i0 = wv; i0 < W; i0 += NK * nwaves) {
        int ii[NK], t0[NK], t1[NK];
        bool two = true;
#pragma unroll
        for (int u = 0; u < NK; ++u) {
            ii[u] = i0 + u * nwaves;
            const bool act = ii[u] < W;
            ii[u] = act ? ii[u] : i0;               // (an inactive one aliases the first: nothing of it is stored, every address stays inside the caller's blocks)
            t0[u] = eoff[ii[u]]; t1[u] = act ? eoff[ii[u] + 1] : t0[u];
            two = two && (t1[u] - t0[u] <= 8);
        }
        v4f64 acc[NK]; double* blk[NK];
#pragma unroll
        for (int u = 0; u < NK; ++u) {
            blk[u] = Blk + (size_t)ii[u] * KC_BLK;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[u][q] = (own0[q] && !(isB[q] && ii[u] + 1 >= W)) ? blk[u][eoffs[q]] : 0.0;
        }
        EC_STAMP(106);
        if (two) {
            // explicit stages over the six (keyframe, group) pairs: list entries, slot pairs, operands -- each a batch of independent reads
            // (written per pair, with its conditional read of the list, the compiler walks the six dependent chains one after the other)
            double a[NK][2], b[NK][2];
            int ee[NK][2]; bool lv[NK][2];
#pragma unroll
            for (int u = 0; u < NK; ++u)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int t = t0[u] + 4 * c + ck;
                    lv[u][c] = t < t1[u];
                    ee[u][c] = elist[lv[u][c] ? t : 0];            // (entry 0 of the list is always allocated)
                }
            int2 sl[NK][2];
#pragma unroll
            for (int u = 0; u < NK; ++u)
#pragma unroll
                for (int c = 0; c < 2; ++c) { ee[u][c] = lv[u][c] ? ee[u][c] : 0; sl[u][c] = eps[ee[u][c]]; }
            double va[NK][2], vo[NK][2], vy[NK][2], vb[NK][2]; bool ho[NK][2];
#pragma unroll
            for (int u = 0; u < NK; ++u)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int e = ee[u][c], i = ii[u];
                    const int side = sl[u][c].x == i ? 0 : 15;
                    const int base = e * 30 + side;
                    const int oth = (sl[u][c].x == i ? sl[u][c].y : sl[u][c].x) == i + 1 ? e * 30 + (15 - side) : -1;
                    ho[u][c] = oth >= 0;
                    va[u][c] = Vs[base + arow]; vo[u][c] = Vs[(oth >= 0 ? oth : base) + arow]; vy[u][c] = yd[e]; vb[u][c] = Vs[base + bcol];
                }
#pragma unroll
            for (int u = 0; u < NK; ++u)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const double av = ci < na ? va[u][c] : (ci < 2 * na ? (ho[u][c] ? vo[u][c] : 0.0) : (ci == 2 * na ? vy[u][c] : 0.0));
                    a[u][c] = lv[u][c] ? -av : 0.0; b[u][c] = (lv[u][c] && colv) ? vb[u][c] : 0.0;
                }
            EC_STAMP(107);
#pragma unroll
            for (int u = 0; u < NK; ++u) {
                acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][0], b[u][0], acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][1], b[u][1], acc[u], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int u = 0; u < NK; ++u)
                for (int tb = t0[u]; tb < t1[u]; tb += 4) {
                    double a, b;
                    operands(ii[u], tb + ck, t1[u], a, b);
                    acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
                }
        }
        EC_STAMP(108);
#pragma unroll
        for (int u = 0; u < NK; ++u) {
            if (t1[u] <= t0[u]) continue;                  // (also the padding keyframes of the last round)
#pragma unroll
            for (int q = 0; q < 4; ++q) if (own0[q] && !(isB[q] && ii[u] + 1 >= W)) blk[u][eoffs[q]] = acc[u][q];
        }
        EC_STAMP(109);
    }
#undef EC_STAMP
}

// dynamic LDS a fat helper carves (chain_step_fat_helper), in bytes: the launch's grant must cover it
__host__ __device__ __forceinline__ size_t chain_fat_helper_lds_bytes(int W, int nd) {
    return (size_t)(352 * 2 + 84 + 48 * 6 + 2 * KC_BLK + 16 * KC_RS + (size_t)nd * 30 + 4 * (nd + 2) + (size_t)nd * 15 + (nd & 1) + 3 * (nd + 2) + 16) * 8 +
           (size_t)(nd + 2) * 8 + (size_t)(((W + 2) & ~1) + 2 + 2 * nd + 2 + 24) * 4 + 80 * 2 + 64;
}
struct ChainArgs {
    int W, n, nd;
    const int2* ep_slots; const int* ep_off; const int* ep_list;
    double* z; int* flag;
    const SolverStatus* status;
    long long* dbg;
    int force_fail;           // test hook: report a breakdown although there is none (exercises the dense fallback)
    int fast;                 // k_chain_step: bit 0 = the tail (Cauchy length, dogleg, candidate) from LDS, bit 1 = the front (candidate's diag/g/cost,
                              // state machine) from LDS; 0 = the generic bodies that talk through the global work vectors (GLIO_CHAIN_FAST, default 3)
    int fronts4;              // 1 = separator + four fronts (chain_f4_split), 0 = two fronts
    double* blk;              // k_chain_solve<true>: [W][KC_BLK] the staged blocks in global memory (windows whose blocks do not fit the LDS)
    // k_chain_step's helper workgroups (grid = 1 + helpers): workgroup 1 + i sums the candidate's block entries of keyframe i over their six sources
    // while workgroup 0 runs the front and the state machine; hsum [W][GLIO_CS_STRIDE], hdone [W] = 4 * hseq + 2 * fat + produced
    int helpers; int hseq; double* hsum; int* hdone; int hpolls;      // hpolls: how often workgroup 0 polls a completion word before it gives the helpers up
    // "fat" helpers (round 6): helper i also forms, ASSUMING the pending candidate is accepted, everything of keyframe i that the step builds between its
    // state machine and the chain -- the scaled, regularised and epoch-corrected block, its right-hand side row, its rows of t = H u, and for the
    // clock-drift epochs whose first keyframe is i the eliminated column V, 1 / sqrt(m), y and t.  Workgroup 0 checks the assumptions (accepted, the mu
    // an acceptance leaves, every helper delivered) and then only LOADS: fat_blk [W][KC_BLK] (360 entries + 15 rows of t used), fat_ep [nd][34].
    int fat; double* fat_blk; double* fat_ep;
};
#define KC_FAT_ENT 360                 /* per keyframe: 345 chain-layout entries + 15 right-hand side entries; the 15 rows of t follow at 360 */
#define KC_FAT_EP 34                   /* per epoch: V (30), 1 / sqrt(m), y, t, pad */

// The kernel also does the work of k_tr_prepare (state machine, scaling vectors) and of k_tr_scale for this structure: it
// reads H and g directly, forms M = S H S + mu D^2 and the right-hand side S g while staging them (same arithmetic, same
// order of operations as k_tr_scale), and computes t = H u from the staged blocks -- two launches and a round trip of the
// scaled matrix through global memory less per iteration.  The dense fallback (k_tr_finish on flag 1) rebuilds what it needs.
// G = true: the staged blocks live in global memory (a.blk) instead of LDS -- windows of more keyframes than the LDS holds (C5: 50 keyframes).  Same
// arithmetic; the elimination runs on four fronts when a.fronts4 is set (chain_f4_factor).
template <bool G>
__global__ __launch_bounds__(KC_THREADS) void k_chain_solve(const ChainArgs a, const TrArgs tr) {
    static_assert(KC_THREADS == TR_THREADS, "tr_prepare_body runs with the chain kernel's workgroup");
    TrDecision dec;
    if (!tr_prepare_body(tr, &dec)) return;
    if (dec.reuse) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int W = a.W, n = a.n, nd = a.nd;
    if (tid == 0) *a.flag = 0;
    // chain order p = [epochs | keyframe 0 (15) | ...]  <->  natural order i = [keyframes | epochs]
    const int np15 = 15 * W;
    const double* Hn = dec.cur ? tr.H1 : tr.H0;
    const double* gn = dec.cur ? tr.g1 : tr.g0;
    // (scale, diag, u were written by tr_prepare_body above: safe to read back because the work vectors occupy whole 128 B
    // lines -- glio_ctx::vstride -- so no line holding them was fetched before they were written)
    const double* scv = V_SCALE(tr);
    const double* dgv = V_DIAG(tr);
    const double mu = dec.mu;
    auto nat = [&](const int p) { return p < nd ? np15 + p : p - nd; };
    // entries of S H S + mu D^2 as k_tr_scale forms them (same operations in the same order); everything is loaded
    // unconditionally and selected afterwards, so that a thread's loads go out as one batch
    auto Rld = [&](const int p) { const int i = nat(p); return scv[i] * gn[i]; };
    double* rd = reinterpret_cast<double*>(tr_lds);
    double* yd = rd + nd + (nd & 1);                       // forward-substituted right-hand side of the epochs
    double* Vs = yd + nd + (nd & 1);                       // [nd][30]: epoch column restricted to its two keyframes, scaled
    double* Blk = G ? a.blk : Vs + (size_t)nd * 30;        // [W][KC_BLK]
    double* CsT = G ? Vs + (size_t)nd * 30 : Blk + (size_t)W * KC_BLK;
    double* CsB = CsT + 288;
    double* zb = CsB + 288;                                // [15 W]
    int2* eps = reinterpret_cast<int2*>(zb + 15 * W + (W & 1));
    int* eoff = reinterpret_cast<int*>(eps + nd + 2);
    int* elist = eoff + ((W + 2) & ~1) + 2;
    int* esd = elist + 2 * nd + 2;                         // [2 nd] per list entry: offset into Vs of this keyframe's rows
    int* eoth = esd + 2 * nd + 2;                          // [2 nd] ... of the next keyframe's rows, or -1
    int* misc = eoth + 2 * nd + 2;                         // [0] bad, [1] number of active local rows, [2..17] their indices
    double* wd = reinterpret_cast<double*>(misc + 24);     // [nd] u / s of the epochs (for t = H u)
    double* wdr = reinterpret_cast<double*>(esd);          // [nd] (u / s) sqrt(m): lives where the index lists go AFTER t = H u
    __shared__ int rowmask;                            // cleared here, two barriers before the first atomicOr into it
    __shared__ int s_prog[4];
    const bool f4 = G && a.fronts4 != 0;
    const ChainSplit cs = chain_f4_split(W);
    ChainF4Mem f4m;
    f4m.es0 = f4m.es1 = wd + nd + (nd & 1); f4m.k0 = 0;
    f4m.cs1a = f4m.es1 + (size_t)(cs.nB + cs.nC) * KC_ES; f4m.cs3a = f4m.cs1a + KC_TILE;
    if (tid < 4) s_prog[tid] = 0;
    if (tid < 18) misc[tid] = 0;
    if (tid == 0) rowmask = 0;
    AR_STAMP(40);
    for (int e = tid; e < nd; e += KC_THREADS) eps[e] = a.ep_slots[e];
    for (int i = tid; i <= W; i += KC_THREADS) eoff[i] = a.ep_off[i];
    __syncthreads();
    for (int t = tid; t < eoff[W]; t += KC_THREADS) elist[t] = a.ep_list[t];
    for (int e = tid; e < nd; e += KC_THREADS) {
        const int ie = np15 + e;
        const double se = scv[ie], he = Hn[(size_t)ie * n + ie], de = dgv[ie];
        const double m = se * he * se + mu * de * de;
        if (!(m > 0.0) || !isfinite(m)) { misc[0] = 1; rd[e] = 0.0; } else rd[e] = rsqrt(m);
    }
    __syncthreads();
    AR_STAMP(41);
    // issued with the same batch of loads: w = u / s (staged where the back substitution will put z later) and the scalars
    // of the row of t = H u this thread will compute
    double pre_s = 1.0, pre_d = 0.0, pre_h = 0.0;
    {
        const double* uv = V_U(tr);
        for (int k = tid; k < np15; k += KC_THREADS) zb[k] = uv[k] / scv[k];
        // wdr = w_e / r_e (r = 1 / sqrt(m)): the epoch term of t = H u un-scales the stored columns V = S c S r with it
        for (int e = tid; e < nd; e += KC_THREADS) { const double we = uv[np15 + e] / scv[np15 + e]; wd[e] = we; wdr[e] = (1.0 / rd[e]) * we; }
        if (tid < np15 + nd) { pre_s = scv[tid]; pre_d = dgv[tid]; pre_h = tid >= np15 ? Hn[(size_t)tid * n + tid] : 0.0; }
    }
    // epoch columns (scaled) and the raw blocks, every global read issued as a batch of independent loads
    for (int e = tid; e < nd; e += KC_THREADS) {
        const int2 sl = eps[e];
        double v[30], sr[30];
        const double sce = scv[np15 + e];
#pragma unroll
        for (int q = 0; q < 30; ++q) {
            const int s1 = q < 15 ? sl.x : sl.y;
            const int i = 15 * (s1 >= 0 ? s1 : 0) + (q < 15 ? q : q - 15);
            v[q] = Hn[(size_t)i * n + np15 + e]; sr[q] = scv[i];
        }
#pragma unroll
        for (int q = 0; q < 30; ++q) { const int s1 = q < 15 ? sl.x : sl.y; v[q] = s1 >= 0 ? sr[q] * v[q] * sce : 0.0; }
        const double re = rd[e];
        int mk = 0;
#pragma unroll
        for (int q = 0; q < 30; ++q) { Vs[e * 30 + q] = v[q] * re; if (v[q] != 0.0) mk |= 1 << (q % 15); }
        if (mk) atomicOr(&rowmask, mk);
        yd[e] = Rld(e) * re;
    }
    for (int q = tid; q < W * 31; q += KC_THREADS) {
        const int i = q / 31, r = q - 31 * i;
        double v[KC_NB];
        if (r < 30) {
            const bool live = r < KC_NB || i + 1 < W;
            const int irow = live ? 15 * i + r : 15 * i, icol = 15 * i;                // r >= 15 runs into the rows of keyframe i+1
            const double si = scv[irow], di = dgv[irow];
            double h[KC_NB], sj[KC_NB];
#pragma unroll
            for (int j = 0; j < KC_NB; ++j) { h[j] = Hn[(size_t)irow * n + icol + j]; sj[j] = scv[icol + j]; }
#pragma unroll
            for (int j = 0; j < KC_NB; ++j) {
                double w = si * h[j] * sj[j];
                w += (irow == icol + j) ? mu * di * di : 0.0;
                v[j] = (live && (r >= KC_NB || j <= r)) ? w : 0.0;
            }
        } else {
#pragma unroll
            for (int j = 0; j < KC_NB; ++j) v[j] = Rld(nd + 15 * i + j);
        }
#pragma unroll
        for (int j = 0; j < KC_NB; ++j) Blk[(size_t)i * KC_BLK + r * KC_RS + j] = v[j];
    }
    __syncthreads();
    // t = H u from the staged (scaled) blocks: t_i = (sum_j (S H S)_ij w_j) / s_i, one row per thread (its scalars were
    // fetched with the blocks, so this phase touches LDS only)
    for (int row = tid; row < np15 + nd; row += KC_THREADS) {
        double acc = 0.0;
        const bool first = row == tid;
        const double s_row = first ? pre_s : scv[row], d_row = first ? pre_d : dgv[row];
        if (row < np15) {
            const int i = row / 15, r = row - 15 * i;
            const double* Bi = Blk + (size_t)i * KC_BLK;
#pragma unroll
            for (int j = 0; j < KC_NB; ++j) {
                double v = j <= r ? Bi[r * KC_RS + j] : Bi[j * KC_RS + r];
                if (j == r) v -= mu * d_row * d_row;
                acc += v * zb[15 * i + j];
            }
            if (i + 1 < W) {
#pragma unroll
                for (int j = 0; j < KC_NB; ++j) acc += Bi[(KC_NB + j) * KC_RS + r] * zb[15 * (i + 1) + j];
            }
            if (i > 0) {
                const double* Bp = Blk + (size_t)(i - 1) * KC_BLK;
#pragma unroll
                for (int j = 0; j < KC_NB; ++j) acc += Bp[(KC_NB + r) * KC_RS + j] * zb[15 * (i - 1) + j];
            }
            for (int tt = eoff[i]; tt < eoff[i + 1]; ++tt) {
                const int e = elist[tt];
                const int side = eps[e].x == i ? 0 : 15;
                acc += Vs[e * 30 + side + r] * wdr[e];
            }
            V_T(tr)[row] = acc / s_row;
        } else {
            const int e = row - np15;
            const int2 sl = eps[e];
            const double se = s_row;
            acc = se * (first ? pre_h : Hn[(size_t)row * n + row]) * se * wd[e];
            if (sl.x >= 0) {
                const double ire = 1.0 / rd[e];             // one division per epoch, not one per entry
#pragma unroll
                for (int q = 0; q < 30; ++q) acc += (Vs[e * 30 + q] * ire) * zb[15 * (q < 15 ? sl.x : sl.y) + (q < 15 ? q : q - 15)];
            }
            V_T(tr)[row] = acc / se;
        }
    }
    __syncthreads();
    AR_STAMP(42);
    if (tid == 0) { int na = 0; for (int q = 0; q < 15; ++q) if (rowmask >> q & 1) misc[2 + na++] = q; misc[1] = na; }
    __syncthreads();
    if (2 * misc[1] + 1 <= 16) chain_epoch_corrections_mfma(W, misc[1], misc + 2, eoff, elist, eps, Vs, yd, Blk, lane, wv, KC_THREADS / 64);
    else {
    // per list entry (keyframe i, epoch e): offset of the epoch's rows of keyframe i in Vs, and of keyframe i+1 (or -1)
    for (int i = wv; i < W; i += KC_THREADS / 64)
        for (int t = eoff[i] + lane; t < eoff[i + 1]; t += 64) {
            const int e = elist[t];
            const int2 sl = eps[e];
            const int side = sl.x == i ? 0 : 15;
            esd[t] = e * 30 + side;
            eoth[t] = (sl.x == i ? sl.y : sl.x) == i + 1 ? e * 30 + (15 - side) : -1;
        }
    __syncthreads();
    // minus the epoch contribution: one item per touched entry, accumulated over the epochs of its keyframe in list order
    {
        const int na = misc[1];
        const int per = 2 * na * na + na;            // D entries, B entries, rhs entries per keyframe
        for (int item = tid; item < W * per; item += KC_THREADS) {
            const int i = item / per, w = item - i * per;
            int r, j, kindI;                          // kindI 0: D_i[r][j], 1: B_i[r][j] (rows of keyframe i+1), 2: rhs_i[j]
            if (w < na * na) { kindI = 0; r = misc[2 + w / na]; j = misc[2 + w % na]; if (j > r) continue; }
            else if (w < 2 * na * na) { kindI = 1; const int u = w - na * na; r = misc[2 + u / na]; j = misc[2 + u % na]; if (i + 1 >= W) continue; }
            else { kindI = 2; r = 0; j = misc[2 + w - 2 * na * na]; }
            double* dst = Blk + (size_t)i * KC_BLK + (kindI == 0 ? r : (kindI == 1 ? KC_NB + r : 30)) * KC_RS + j;
            double v = *dst;
            // four epochs per round: the index reads, then the operand reads, go out as independent batches (the
            // subtractions stay in list order, so the result does not depend on the batching)
            const int t1 = eoff[i + 1];
            for (int t = eoff[i]; t < t1; t += 4) {
                int base[4], ob[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int tt = t + q < t1 ? t + q : t1 - 1;
                    base[q] = esd[tt];
                    ob[q] = kindI == 1 ? eoth[tt] : (kindI == 2 ? elist[tt] : 0);
                }
                double xa[4], xb[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    xb[q] = Vs[base[q] + j];
                    xa[q] = kindI == 0 ? Vs[base[q] + r] : (kindI == 1 ? Vs[(ob[q] >= 0 ? ob[q] : base[q]) + r] : yd[ob[q]]);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool live = (t + q < t1) & !(kindI == 1 && ob[q] < 0);
                    v -= live ? xa[q] * xb[q] : 0.0;
                }
            }
            *dst = v;
        }
    }
    }
    __syncthreads();
    AR_STAMP(43);
    // the chain from both ends
    const int mid = W / 2, nT = mid, nB = W - 1 - mid, T = nT > nB ? nT : nB;
    double av[KC_NB];
#pragma unroll
    for (int j = 0; j < KC_NB; ++j) av[j] = 0.0;
    if (lane < KC_NB || lane == 30) {
        const int row = lane < KC_NB ? lane : 30;
        if (wv == 0 && nT > 0) { for (int j = 0; j < KC_NB; ++j) av[j] = Blk[row * KC_RS + j]; }
        if (wv == 2 && nB > 0) { for (int j = 0; j < KC_NB; ++j) av[j] = Blk[(size_t)(W - 1) * KC_BLK + row * KC_RS + j]; }
    }
    bool bad = false;
    long long ph[5] = {0, 0, 0, 0, 0};
    if (f4) chain_f4_factor<G>(W, cs, Blk, CsT, CsB, f4m, zb, s_prog, lane, wv, bad, ph, a.dbg);
    else
    for (int it = 0; it <= T; ++it) {
        if (wv == 0) {
            if (it < nT) chain_step15<false, G>(it, it + 1, true, av, Blk, CsT, lane, bad, ph);
            else if (it == T) {
                {
                    const int row = lane < KC_NB ? lane : 30, crow = lane < KC_NB ? lane : 15;
                    const int lim = lane == 30 ? KC_NB : (lane < KC_NB ? lane + 1 : 0);
                    double b0[KC_NB], c0[KC_NB], c1[KC_NB];
#pragma unroll
                    for (int j = 0; j < KC_NB; ++j) { b0[j] = Blk[(size_t)mid * KC_BLK + row * KC_RS + j]; c0[j] = CsT[crow * KC_RS + j]; c1[j] = CsB[crow * KC_RS + j]; }
#pragma unroll
                    for (int j = 0; j < KC_NB; ++j) {
                        double v = b0[j];
                        if (nT > 0) v -= c0[j];
                        if (nB > 0) v -= c1[j];
                        av[j] = j < lim ? v : 0.0;
                    }
                }
                chain_step15<false, G>(mid, mid, false, av, Blk, CsT, lane, bad);
            }
        } else if (wv == 2) {
            if (it < nB) { const int i = W - 1 - it; chain_step15<true, G>(i, i - 1, true, av, Blk, CsB, lane, bad); }
        } else if (wv == 1) {                       // (SIMD 1 and 3: not the SIMDs the two chain wavefronts issue on)
            if (it >= 1 && it - 1 < nT) chain_prepare_back<G>(Blk + (size_t)(it - 1) * KC_BLK, lane);
        } else if (wv == 3) {
            if (it >= 1 && it - 1 < nB) chain_prepare_back<G>(Blk + (size_t)(W - it) * KC_BLK, lane);
        }
        __syncthreads();
    }
    AR_STAMP(44);
#ifdef GLIO_DEV_STAMPS
    if (tid == 0) for (int k = 0; k < 5; ++k) a.dbg[60 + k] = ph[k];
#endif
    if ((bad || a.force_fail) && lane == 0) misc[0] = 1;
    __syncthreads();
    if (misc[0]) { if (tid == 0) atomicOr(a.flag, 1); return; }
    // back substitution: meeting keyframe, then the two halves in parallel:  L_ii^T z_i = y_i - L_{nbr,i}^T z_nbr
    auto back = [&](const int i, const int nbr) {
        const double* Bi = Blk + (size_t)i * KC_BLK;
        const int ln = lane < KC_NB ? lane : 0;
        double lcol[KC_NB], bcol[KC_NB];             // column `lane` of L_ii and of L_{nbr,i}: fetched before the dependent chain starts
#pragma unroll
        for (int k = 0; k < KC_NB; ++k) { lcol[k] = Bi[k * KC_RS + ln]; bcol[k] = Bi[(KC_NB + k) * KC_RS + ln]; }
        const double rp = lane < KC_NB ? Bi[31 * KC_RS + lane] : 1.0;
        double v = lane < KC_NB ? Bi[30 * KC_RS + lane] : 0.0;
        if (nbr >= 0) {
            double s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
            for (int k = 0; k < KC_NB; k += 3) { s0 += bcol[k] * zb[15 * nbr + k]; s1 += bcol[k + 1] * zb[15 * nbr + k + 1]; s2 += bcol[k + 2] * zb[15 * nbr + k + 2]; }
            v -= (s0 + s1) + s2;
        }
#pragma unroll
        for (int k = KC_NB - 1; k >= 0; --k) {
            const double zk = readlane_d(v, k) * readlane_d(rp, k);
            if (lane == k) v = zk;
            else if (lane < k) v -= lcol[k] * zk;
        }
        if (lane < KC_NB) zb[15 * i + lane] = v;
        GLIO_WAVE_LDS_SYNC();
    };
    auto back_mv = [&](const int i, const int nbr) {          // z_i = w_i - M_i z_neighbour (blocks transformed by chain_prepare_back)
        const double* Bi = Blk + (size_t)i * KC_BLK;
        if (lane < KC_NB) {
            double mrow[KC_NB], zn[KC_NB];
#pragma unroll
            for (int k = 0; k < KC_NB; ++k) { mrow[k] = Bi[(KC_NB + lane) * KC_RS + k]; zn[k] = zb[15 * nbr + k]; }
            double s0 = Bi[30 * KC_RS + lane], s1 = 0, s2 = 0;
#pragma unroll
            for (int k = 0; k < KC_NB; k += 3) { s0 -= mrow[k] * zn[k]; s1 -= mrow[k + 1] * zn[k + 1]; s2 -= mrow[k + 2] * zn[k + 2]; }
            zb[15 * i + lane] = (s0 + s1) + s2;
        }
        GLIO_WAVE_LDS_SYNC();
    };
    if (f4) {
        __syncthreads();
        chain_f4_backsub<G>(W, cs, Blk, f4m, zb, lane, wv);
    } else {
    if (wv == 0) back(mid, -1);
    __syncthreads();
    if (wv == 0) { for (int i = mid - 1; i >= 0; --i) back_mv(i, i + 1); }
    else if (wv == 1) { for (int i = mid + 1; i < W; ++i) back_mv(i, i - 1); }
    }
    __syncthreads();
    AR_STAMP(45);
    double bd2 = 0.0;
    for (int e = tid; e < nd; e += KC_THREADS) {
        const int2 sl = eps[e];
        double v = yd[e];
        if (sl.x >= 0) {
#pragma unroll
            for (int q = 0; q < 15; ++q) { v -= Vs[e * 30 + q] * zb[15 * sl.x + q]; v -= Vs[e * 30 + 15 + q] * zb[15 * sl.y + q]; }
        }
        v *= rd[e];
        a.z[e] = v;
        if (!isfinite(v)) bd2 = 1.0;
    }
    for (int k = tid; k < 15 * W; k += KC_THREADS) { const double v = zb[k]; a.z[nd + k] = v; if (!isfinite(v)) bd2 = 1.0; }
    if (bd2 != 0.0) misc[0] = 1;
    __syncthreads();
    AR_STAMP(46);
    if (tid == 0) { if (misc[0]) atomicOr(a.flag, 1); else *a.flag = 2; }
}

// ------------------------------------------------------------------------------------------------
// k_chain_step: the WHOLE trust-region step of the keyframe-chain path in one launch (one workgroup) and without a dense
// H.  It gathers what it needs straight from the factor blocks that k_linearize_all left (K3 partials, IMU / GNSS pair
// blocks, clock-drift blocks, prior H) -- every entry as the same sum, in the same order, that k_assemble would have
// written into the dense matrix, so the numbers downstream are bit-identical to the assemble + k_chain_solve + k_tr_finish
// sequence it replaces -- and then runs, in this order: the state machine (tr_prepare_body, fed with diag(H), g and the
// cost of the candidate), the scaling + epoch elimination + twisted chain factorisation + back substitution of
// k_chain_solve, and the Cauchy / dogleg step + candidate of k_tr_finish.  A steady-state iteration is two launches:
// [k_linearize_all, k_chain_step].  On a non-positive pivot the same workgroup rebuilds S H S + mu D^2 densely from the
// blocks (ChainBuilder) and factors it with the generic blocked Cholesky and Ceres' mu retries -- the outcome is the dense
// one in every case, as before.
// ------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ size_t tr_step_lds_doubles(int n) {
    size_t d = (size_t)TR_NB * bp_stride(n);
    d += 16 * 256;
    d += (TR_NB + 1) * TR_PS;
    d += n + (n & 1);
    d += 32 + 16;
    return d;
}
// byte offset of the gather tables inside k_chain_step's dynamic LDS: behind both carves that use the front of the array
__host__ __device__ __forceinline__ size_t chain_step_tabs_offset(int W, int nd, int n) {
    const size_t a = chain_lds_doubles(W, nd) * 8, b = tr_step_lds_doubles(n) * 8;
    return ((a > b ? a : b) + 15) & ~(size_t)15;
}
// mirrors: the LDS copies of g, g~, t and the state that the LDS-resident front and tail (ChainArgs::fast) work on; a window whose blocks leave no
// room for them still takes this kernel, with the generic bodies
__host__ __device__ __forceinline__ size_t chain_step_lds_bytes(int W, int nd, int n, bool mirrors = true) {
    return chain_step_tabs_offset(W, nd, n) + (size_t)W * GLIO_LIDAR_ACC * 8 + (mirrors ? 5 : 2) * (size_t)(n + (n & 1)) * 8 + (mirrors ? (size_t)(n + W + ((n + W) & 1)) * 8 : 0) + (size_t)nd * 128 +
           (size_t)(8 + 15) * W * 2 + 3 * 346 * 2 + 64;
}
__host__ __device__ __forceinline__ size_t chain_lds_doubles_g(int W, int nd) {
    const ChainSplit c = chain_f4_split(W);
    return chain_lds_doubles(W, nd) - (size_t)W * KC_BLK + (size_t)(nd + (nd & 1)) + (size_t)(c.nB + c.nC) * KC_ES + 2 * KC_TILE + 8;
}
// Where the four-front panels live (k_chain_step): nothing new is allocated for the E slots that fit into the LDS copy of the clock-drift blocks
// (dds, dead once t = H u is formed); the rest and the inner fronts' two C00 tiles start at the gather index tables (wr30 / wj / wlx, dead after
// the block gather) and run past the regular end of the carve.  The host sizes the launch with the same function.
struct ChainF4Layout { size_t off_dds, off_r1, total; int k0; };
__host__ __device__ __forceinline__ ChainF4Layout chain_f4_layout(const int W, const int nd, const int n, const bool mir) {
    const size_t n2 = n + (n & 1), nx = n + W, nx2 = nx + (nx & 1);
    ChainF4Layout L;
    L.off_dds = chain_step_tabs_offset(W, nd, n) + (size_t)W * GLIO_LIDAR_ACC * 8 + (mir ? 5 : 2) * n2 * 8 + (mir ? nx2 * 8 : 0);
    const size_t off_stab = L.off_dds + ((size_t)nd * 15 + (nd & 1)) * 8;
    const size_t off_wr30 = off_stab + ((size_t)(8 + 15) * W + ((8 + 15) * W & 1)) * 2;
    L.off_r1 = (off_wr30 + 15) & ~(size_t)15;
    const ChainSplit c = chain_f4_split(W);
    const int nE = c.nB + c.nC, fit = (int)(((size_t)nd * 15) / KC_ES);
    L.k0 = fit < nE ? fit : nE;
    L.total = L.off_r1 + ((size_t)(nE - L.k0) * KC_ES + 2 * KC_TILE) * 8;
    return L;
}
struct GatherArgs {
    const double* lidar_partials; size_t lidar_pstride; int lidar_nb;
    const PairBlock* imu_blocks; const PairBlock* gnss_blocks; const DdtBlock* ddt_blocks;
    int gnss_stride, ddt_stride, n_imu, n_groups, has_prior, np;
    const double* pH; const double* pg; const double* pcost;
    const short* tabs;            // [W] ChainKf, then [15 W] prior index: built on the host from the graph's structure (glio_chain_tabs_upload)
    const double* chain_src;      // [2][W][GLIO_CS_SOURCES][GLIO_CS_STRIDE] chain-layout contributions written by the factor roles
    double* hd0; double* hd1; double* g0; double* g1; double* c0; double* c1;
};
// where the blocks of keyframe i come from (the chain structure leaves at most these):
//   e0 IMU edge (i, i+1) [its aa part feeds D_i, its ba part B_i], e1 IMU edge (i-1, i) [bb part feeds D_i],
//   k0 / k1 the GNSS groups touching i in ascending index, o0 / o1 = 1 when i is the group's slot_a,
//   kp the GNSS group of the pair (i, i+1) [ba part feeds B_i]
struct GatherTabs {
    const ChainKf* kd;                // [W]
    const short* pidx;                // [15 W] prior row of a pose parameter or -1
    double* lid;                      // [W][28] K3 partials of one buffer summed in index order
};
__device__ __forceinline__ int kc_lidar_sym_index(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }
__device__ __forceinline__ int kc_dop_local12(int slot_is_j, int lc) {
    int k;
    if (lc < 3) k = lc; else if (lc >= 6 && lc < 9) k = 3 + (lc - 6); else return -1;
    return slot_is_j ? 6 + k : k;
}
// chain-layout index of an entry: r30 < 15 -> D_i[r30][j] (j <= r30), r30 >= 15 -> B_i[r30 - 15][j]
__device__ __forceinline__ int kc_w(const int r30, const int j) { return r30 < 15 ? r30 * (r30 + 1) / 2 + j : 120 + (r30 - 15) * 15 + j; }
// One entry of the block-tridiagonal H as the sum of the LiDAR block and the five chain-layout slices of keyframe i, added in
// k_assemble's order (LiDAR, IMU edge (i, i+1), IMU edge (i-1, i), GNSS groups by ascending index, prior; a slice of an absent
// source is zero), so the sum is bit-identical to the dense matrix' entry.
__device__ __forceinline__ double kc_gather_entry(const GatherArgs& G, const GatherTabs& T, const int which, const int W, const int i, const int r30, const int j) {
    const double* p = G.chain_src + (((size_t)which * W + i) * GLIO_CS_SOURCES) * GLIO_CS_STRIDE + kc_w(r30, j);
    const double v0 = p[0], v1 = p[GLIO_CS_STRIDE], v2 = p[2 * GLIO_CS_STRIDE], v3 = p[3 * GLIO_CS_STRIDE], v4 = p[4 * GLIO_CS_STRIDE];
    const bool lid = (r30 < 6) & (j < 6);
    double s = 0;
    s += lid ? T.lid[i * GLIO_LIDAR_ACC + kc_lidar_sym_index(j < r30 ? j : r30, j < r30 ? r30 : j)] : 0.0;
    s += v0; s += v1; s += v2; s += v3; s += v4;
    return s;
}
// gradient entry of pose parameter (sc, lc), k_assemble's order
__device__ __forceinline__ double kc_gather_grad(const GatherArgs& G, const GatherTabs& T, const int which, const int W, const int sc, const int lc) {
    const PairBlock* imu = G.imu_blocks + (size_t)which * W;
    const PairBlock* gn = G.gnss_blocks + (size_t)which * G.gnss_stride;
    const double* pg = G.pg + (size_t)which * G.np;
    const ChainKf d = T.kd[sc];
    const bool lid = lc < 6;
    const int pi = T.pidx[15 * sc + lc];
    const double v0 = imu[d.e0 >= 0 ? d.e0 : 0].g[lc];
    const double v1 = imu[d.e1 >= 0 ? d.e1 : 0].g[15 + lc];
    const double vg0 = gn[d.k0 >= 0 ? d.k0 : 0].g[(d.o0 ? 0 : 15) + lc];
    const double vg1 = gn[d.k1 >= 0 ? d.k1 : 0].g[(d.o1 ? 0 : 15) + lc];
    const double vp = pg[pi >= 0 ? pi : 0];
    const double vl = T.lid[sc * GLIO_LIDAR_ACC + 21 + (lid ? lc : 0)];
    double s = 0;
    s += lid ? vl : 0.0;
    s += d.e0 >= 0 ? v0 : 0.0;
    s += d.e1 >= 0 ? v1 : 0.0;
    s += d.k0 >= 0 ? vg0 : 0.0;
    s += d.k1 >= 0 ? vg1 : 0.0;
    s += pi >= 0 ? vp : 0.0;
    return s;
}
// K3 partials of buffer `which` -> T.lid, each entry the sum of its nb partials in index order (what k_assemble adds up)
__device__ __forceinline__ void kc_reduce_lidar(const GatherArgs& G, const GatherTabs& T, const int which, const int W) {
    const double* lp = G.lidar_partials + (size_t)which * G.lidar_pstride;
    const int lnb = G.lidar_nb;
    const int total = W * GLIO_LIDAR_ACC;
    if (lnb <= 24 && total <= 2 * (int)blockDim.x) {
        // the usual geometry (560 entries, 24 partials each): a thread takes two ADJACENT entries and reads them as 16-byte loads -- 24 loads
        // for the pair in one round (partial blocks are 28 doubles = 14 x 16 bytes apart, the pair starts at an even index)
        static_assert(GLIO_LIDAR_ACC % 2 == 0, "pairs of entries stay inside a 28-double block");
        const int pr = threadIdx.x;
        if (2 * pr < total) {
            const int it0 = 2 * pr;
            const int s0 = it0 / GLIO_LIDAR_ACC;
            const double* p0 = lp + (size_t)s0 * lnb * GLIO_LIDAR_ACC + (it0 - s0 * GLIO_LIDAR_ACC);
            v2f64 va[24];
#pragma unroll
            for (int q = 0; q < 24; ++q) va[q] = *reinterpret_cast<const v2f64*>(p0 + (size_t)(q < lnb ? q : 0) * GLIO_LIDAR_ACC);
            double sa = 0, sb = 0;
#pragma unroll
            for (int q = 0; q < 24; ++q) { sa += q < lnb ? va[q][0] : 0.0; sb += q < lnb ? va[q][1] : 0.0; }
            T.lid[it0] = sa; T.lid[it0 + 1] = sb;
        }
        return;
    }
    for (int it = threadIdx.x; it < W * GLIO_LIDAR_ACC; it += blockDim.x) {
        const int slot = it / GLIO_LIDAR_ACC, idx = it - slot * GLIO_LIDAR_ACC;
        const double* p = lp + (size_t)slot * lnb * GLIO_LIDAR_ACC + idx;
        double sacc = 0;
        for (int k0 = 0; k0 < lnb; k0 += 24) {
            double vb[24];
#pragma unroll
            for (int q = 0; q < 24; ++q) vb[q] = p[(size_t)(k0 + q < lnb ? k0 + q : k0) * GLIO_LIDAR_ACC];
#pragma unroll
            for (int q = 0; q < 24; ++q) sacc += k0 + q < lnb ? vb[q] : 0.0;
        }
        T.lid[it] = sacc;
    }
}

// rebuilds S H S + mu D^2 (tr_perm order, mode 1) and the right-hand side row in a.L from the factor blocks: the dense
// fallback of k_chain_step.  Entries outside the block-tridiagonal part and the epoch columns are zero by the chain property.
struct ChainBuilder {
    static constexpr bool kHasDenseH = false;
    const GatherArgs* G; const GatherTabs* T; int cur;
    __device__ void operator()(const TrArgs& a, const double mu) const {
        const int n = a.n, W = a.W, nd = a.n_ddt, np15 = 15 * W, tid = threadIdx.x;
        const double* scale = V_SCALE(a); const double* diag = V_DIAG(a);
        const double* g = cur ? a.g1 : a.g0;
        const DdtBlock* dd = G->ddt_blocks + (size_t)cur * G->ddt_stride;
        for (size_t k = tid; k < (size_t)(n + 1) * n; k += TR_THREADS) a.L[k] = 0.0;
        __syncthreads();
        for (int q = tid; q < W * 30 * 15; q += TR_THREADS) {
            const int i = q / 450, rem = q - 450 * i, r30 = rem / 15, j = rem - 15 * r30;
            if (r30 >= 15 && i + 1 >= W) continue;
            if (r30 < 15 && j > r30) continue;
            const int irow = 15 * i + r30, icol = 15 * i + j;          // r30 >= 15 runs into keyframe i + 1
            double v = scale[irow] * kc_gather_entry(*G, *T, cur, W, i, r30, j) * scale[icol];
            if (irow == icol) v += mu * diag[irow] * diag[irow];
            a.L[(size_t)(nd + irow) * n + nd + icol] = v;
        }
        // epoch columns (the pair (sa, sa + 1) of the epoch's group) and diagonal; epochs come first in the elimination order
        for (int q = tid; q < nd * 31; q += TR_THREADS) {
            const int e = q / 31, r = q - 31 * e;
            const int ie = np15 + e;
            if (r == 30) { a.L[(size_t)e * n + e] = scale[ie] * dd[e].h * scale[ie] + mu * diag[ie] * diag[ie]; continue; }
            if (!dd[e].used) continue;
            const int sa = G->gnss_blocks[(size_t)cur * G->gnss_stride + dd[e].group].slot_a;
            const int k12 = kc_dop_local12(r >= 15, r < 15 ? r : r - 15);
            if (k12 < 0) continue;
            const int irow = 15 * sa + r;
            a.L[(size_t)(nd + irow) * n + e] = scale[irow] * dd[e].c[k12] * scale[ie];
        }
        for (int j = tid; j < n; j += TR_THREADS) a.L[(size_t)n * n + tr_perm(j, W, nd, 1)] = scale[j] * g[j];
    }
};

// Helper workgroup of k_chain_step for keyframe i (blockIdx.x = 1 + i).  The block gather of the step reads 276 KB that the factor roles left in six
// places -- what ONE compute unit can pull in is what that phase costs (6.2 us).  The sums themselves depend on nothing the main workgroup decides
// (only on which buffer holds the candidate: the status record as the previous kernel left it), so W otherwise idle compute units form them
// while workgroup 0 gathers diag / g / cost and runs the state machine; it then reads 55 KB of sums.  Every entry is the same sum in the same
// order as gather_store forms it (LiDAR partial sum first, then the five slices), so the step stays bit-identical.  (Measured: all helpers on
// workgroup 0's XCD -- grid 1 + 8 W, seven of eight workgroups idle -- read back no faster and delayed the front by 12 us.)  The helper ALWAYS reports
// (4 * hseq + 2 * fat + produced): workgroup 0 waits for all of them before its state machine rewrites the status record they read.
__device__ __forceinline__ void chain_step_helper(const ChainArgs& a, const TrArgs& tr, const GatherArgs& G, const int i) {
    __shared__ double h_lid[GLIO_LIDAR_ACC];
    __shared__ int h_go, h_cand;
    const int tid = threadIdx.x, W = a.W;
    if (tid == 0) {
        const SolverStatus st = *tr.status;
        h_go = (!st.done && st.cand_pending) ? 1 : 0;
        h_cand = 1 - st.cur;
    }
    __syncthreads();
    const int go = h_go, cand = h_cand;
    if (go) {
        if (tid < GLIO_LIDAR_ACC) {          // the keyframe's K3 partials, added in index order (kc_reduce_lidar)
            const int lnb = G.lidar_nb;
            const double* p = G.lidar_partials + (size_t)cand * G.lidar_pstride + (size_t)i * lnb * GLIO_LIDAR_ACC + tid;
            double sacc = 0;
            for (int k0 = 0; k0 < lnb; k0 += 24) {
                double vb[24];
#pragma unroll
                for (int q = 0; q < 24; ++q) vb[q] = p[(size_t)(k0 + q < lnb ? k0 + q : k0) * GLIO_LIDAR_ACC];
#pragma unroll
                for (int q = 0; q < 24; ++q) sacc += k0 + q < lnb ? vb[q] : 0.0;
            }
            h_lid[tid] = sacc;
        }
        double v[5];
        int lix = -1;
        const int w = tid < 345 ? tid : 0;
        {
            int r30, j;
            if (w < 120) { int r = 0; while ((r + 1) * (r + 2) / 2 <= w) ++r; r30 = r; j = w - r * (r + 1) / 2; }
            else { r30 = 15 + (w - 120) / 15; j = (w - 120) % 15; }
            lix = (r30 < 6 && j < 6) ? kc_lidar_sym_index(j, r30) : -1;
            const double* p = G.chain_src + (((size_t)cand * W + i) * GLIO_CS_SOURCES) * GLIO_CS_STRIDE + w;
#pragma unroll
            for (int sidx = 0; sidx < 5; ++sidx) v[sidx] = p[sidx * GLIO_CS_STRIDE];
        }
        __syncthreads();
        if (tid < 345) {
            double h = 0;
            h += lix >= 0 ? h_lid[lix] : 0.0;
            h += v[0]; h += v[1]; h += v[2]; h += v[3]; h += v[4];
            a.hsum[(size_t)i * GLIO_CS_STRIDE + tid] = h;
        }
    }
    __syncthreads();                                   // (one agent-scope release, by the store below: see chain_step_fat_helper)
    if (tid == 0) __hip_atomic_store(&a.hdone[i], 4 * a.hseq + go, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// The FAT helper of keyframe i (ChainArgs::fat): besides the sums of chain_step_helper it forms what workgroup 0 would build from them between its state
// machine and the chain -- under the assumption that the pending candidate is ACCEPTED (then the current point is the candidate and, with the dogleg
// strategy, mu becomes max(1e-8, mu / 5): DoglegStrategy::StepAccepted).  Every quantity is the SAME expression on the same operands in the same order as in
// workgroup 0's own phases (work_vectors of the state machine; "round 1", the epoch columns, the block store, t = H u and the epoch corrections of
// k_chain_step), restricted to keyframe i -- so the step's numbers do not depend on who formed them.  It needs keyframes i - 1 and i + 1 as far as they
// enter: their scale / diagonal / gradient rows (u of the neighbours), the coupling block B_{i-1}, the clock-drift epochs that touch keyframe i.
// Everything lives in this workgroup's own dynamic LDS.  Not taken (fat = 0 in the completion word, workgroup 0 builds as before): first step of a solve
// (no scale yet), Levenberg-Marquardt (mu follows the radius), an epoch without keyframes, a non-positive epoch pivot.
__device__ __forceinline__ void chain_step_fat_helper(const ChainArgs& a, const TrArgs& tr, const GatherArgs& G, const int i) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, W = a.W, nd = a.nd, np15 = 15 * W;
#ifdef GLIO_DEV_STAMPS
#define FH_STAMP(k) do { if (a.dbg && i == W / 2 && tid == 0) a.dbg[304 + (k)] = wall_clock64(); } while (0)
#else
#define FH_STAMP(k) do { } while (0)
#endif
    double* L = reinterpret_cast<double*>(tr_lds);
    double* f_h = L;                       // [352] block entries of keyframe i summed over their six sources (= hsum)
    double* f_hp = f_h + 352;              // [352] the same of keyframe i - 1
    double* f_lid = f_hp + 352;            // [3][28] K3 partial sums of keyframes i - 1, i, i + 1
    double* f_S = f_lid + 84;              // [3][16] Jacobi scale of their rows
    double* f_D = f_S + 48;                // D = sqrt(clamp(S^2 H_rr))
    double* f_gr = f_D + 48;               // g~ = S g / D
    double* f_G = f_gr + 48;               // g
    double* f_zb = f_G + 48;               // u / S with u = S g~ / D
    double* f_blk = f_zb + 48 + 48;        // [KC_BLK] the block of keyframe i as the chain takes it
    double* f_blk2 = f_blk + KC_BLK;       // the same before the epochs' corrections: what t = H u is formed from, while wavefront 0 corrects f_blk
    double* f_bp = f_blk2 + KC_BLK;        // [15][KC_RS] rows 15..29 of block i - 1 (B_{i-1})
    double* f_Vs = f_bp + 16 * KC_RS;      // [nd][30]
    double* f_rd = f_Vs + (size_t)nd * 30; // [nd] each: 1 / sqrt(m), y, u / S, (u / S) sqrt(m), scale, D, g~ of the epoch unknowns
    double* f_yd = f_rd + nd + 2;
    double* f_wd = f_yd + nd + 2;
    double* f_wdr = f_wd + nd + 2;
    double* f_dds = f_wdr + nd + 2;        // [nd][15] clock-drift blocks (only the epochs touching keyframe i are filled)
    double* f_Se = f_dds + (size_t)nd * 15 + (nd & 1);
    double* f_De = f_Se + nd + 2;
    double* f_gre = f_De + nd + 2;
    double* f_t = f_gre + nd + 2;          // [16] rows of t of keyframe i
    int2* f_eps = reinterpret_cast<int2*>(f_t + 16);
    int* f_eoff = reinterpret_cast<int*>(f_eps + nd + 2);
    int* f_elist = f_eoff + ((W + 2) & ~1) + 2;
    int* f_misc = f_elist + 2 * nd + 2;    // [0] failure, [1] rows with an epoch coupling, [2..] their indices, [20] row mask
    short* f_tab = reinterpret_cast<short*>(f_misc + 24);     // ChainKf of i - 1, i, i + 1 (24 shorts), then pidx of their 45 rows
    __shared__ int h_go, h_cand, h_fat, h_ph0;
    __shared__ double h_mu;
    const short* gtab = G.tabs;
    if (tid == 0) {
        const SolverStatus st = *tr.status;
        h_go = (!st.done && st.cand_pending) ? 1 : 0;
        h_cand = 1 - st.cur;
        h_fat = (h_go && !tr.lm && (a.fast & 3) == 3) ? 1 : 0;
        h_ph0 = st.phase == 0 ? 1 : 0;
        // mu as the state machine will leave it: the first step of a solve keeps the record's (and takes the point whatever it is, forming the Jacobi
        // scale from its diagonal); a later one is assumed accepted: DoglegStrategy::StepAccepted, tr_prepare_body
        h_mu = st.phase == 0 ? st.mu : fmax(1e-8, 2.0 * st.mu / 10.0);
    }
    // the structure tables travel with the status: descriptors of the three keyframes, the prior index of their rows, the epoch tables
    if (tid < 24) { const int k = tid >> 3, kk = i - 1 + k; f_tab[tid] = (kk >= 0 && kk < W) ? gtab[8 * kk + (tid & 7)] : (short)-1; }
    if (tid >= 64 && tid < 64 + 45) { const int q = tid - 64, k = q / 15, kk = i - 1 + k; f_tab[24 + q] = (kk >= 0 && kk < W) ? gtab[8 * W + 15 * kk + (q - 15 * k)] : (short)-1; }
    for (int e = tid; e < nd; e += KC_THREADS) f_eps[e] = a.ep_slots[e];
    for (int k = tid; k <= W; k += KC_THREADS) f_eoff[k] = a.ep_off[k];
    for (int t = tid; t < 2 * nd; t += KC_THREADS) f_elist[t] = a.ep_list[t];
    if (tid < 24) f_misc[tid] = 0;
    for (int k = tid; k < 2 * KC_BLK; k += KC_THREADS) f_blk[k] = 0.0;       // the block and its copy for t = H u (f_blk2): zero above the diagonal, pivots row unused
    __syncthreads();
    FH_STAMP(1);
    const int go = h_go, cand = h_cand;
    bool fat = h_fat != 0;
    const bool ph0 = h_ph0 != 0;
    const double mu = h_mu;
    if (go) {
        // ---- one round of loads: K3 partials of three keyframes, the slices of keyframes i and i - 1, the diagonal slices of i + 1, gradients, scale, epochs
        const int lnb = G.lidar_nb;
        double lsum = 0;
        if (tid < 84) {
            const int k = tid / GLIO_LIDAR_ACC, kk = i - 1 + k, ent = tid - GLIO_LIDAR_ACC * k;
            if (kk >= 0 && kk < W) {
                const double* p = G.lidar_partials + (size_t)cand * G.lidar_pstride + (size_t)kk * lnb * GLIO_LIDAR_ACC + ent;
                for (int k0 = 0; k0 < lnb; k0 += 24) {
                    double vb[24];
#pragma unroll
                    for (int q = 0; q < 24; ++q) vb[q] = p[(size_t)(k0 + q < lnb ? k0 + q : k0) * GLIO_LIDAR_ACC];
#pragma unroll
                    for (int q = 0; q < 24; ++q) lsum += k0 + q < lnb ? vb[q] : 0.0;
                }
            }
        }
        double v[5], vp[5];
        int lix = -1;
        const int w = tid < 345 ? tid : 0;
        {
            int r30, j;
            if (w < 120) { int r = 0; while ((r + 1) * (r + 2) / 2 <= w) ++r; r30 = r; j = w - r * (r + 1) / 2; }
            else { r30 = 15 + (w - 120) / 15; j = (w - 120) % 15; }
            lix = (r30 < 6 && j < 6) ? kc_lidar_sym_index(j, r30) : -1;
            const double* p = G.chain_src + (((size_t)cand * W + i) * GLIO_CS_SOURCES) * GLIO_CS_STRIDE + w;
            const double* pp = G.chain_src + (((size_t)cand * W + (i > 0 ? i - 1 : 0)) * GLIO_CS_SOURCES) * GLIO_CS_STRIDE + w;
#pragma unroll
            for (int sidx = 0; sidx < 5; ++sidx) { v[sidx] = p[sidx * GLIO_CS_STRIDE]; vp[sidx] = pp[sidx * GLIO_CS_STRIDE]; }
        }
        // rows of the three keyframes (thread q of the second wavefront pair: 45 rows): scale, gradient parts, the diagonal slices of keyframe i + 1
        double rs = 0, g5[5] = {0, 0, 0, 0, 0}, dn[5] = {0, 0, 0, 0, 0};
        int rk = 0, rlc = 0, rkk = -1, rpi = -1;
        ChainKf rd_; rd_.e0 = rd_.e1 = rd_.k0 = rd_.k1 = rd_.kp = -1; rd_.o0 = rd_.o1 = 0; rd_.pad_ = 0;
        if (tid >= 384 && tid < 384 + 45) {
            const int q = tid - 384;
            rk = q / 15; rlc = q - 15 * rk; rkk = i - 1 + rk;
            if (rkk >= 0 && rkk < W) {
                rd_ = reinterpret_cast<const ChainKf*>(f_tab)[rk];
                rpi = f_tab[24 + q];
                const PairBlock* imu = G.imu_blocks + (size_t)cand * W;
                const PairBlock* gnb = G.gnss_blocks + (size_t)cand * G.gnss_stride;
                rs = V_SCALE(tr)[15 * rkk + rlc];
                g5[0] = imu[rd_.e0 >= 0 ? rd_.e0 : 0].g[rlc];
                g5[1] = imu[rd_.e1 >= 0 ? rd_.e1 : 0].g[15 + rlc];
                g5[2] = gnb[rd_.k0 >= 0 ? rd_.k0 : 0].g[(rd_.o0 ? 0 : 15) + rlc];
                g5[3] = gnb[rd_.k1 >= 0 ? rd_.k1 : 0].g[(rd_.o1 ? 0 : 15) + rlc];
                g5[4] = (G.pg + (size_t)cand * G.np)[rpi >= 0 ? rpi : 0];
                if (rk == 2) {
                    const double* p = G.chain_src + (((size_t)cand * W + rkk) * GLIO_CS_SOURCES) * GLIO_CS_STRIDE + kc_w(rlc, rlc);
#pragma unroll
                    for (int sidx = 0; sidx < 5; ++sidx) dn[sidx] = p[sidx * GLIO_CS_STRIDE];
                }
            }
        }
        // the epochs that touch keyframe i: their blocks and the scale of their unknown
        const int t0e = f_eoff[i], t1e = f_eoff[i + 1];
        const DdtBlock* dd = G.ddt_blocks + (size_t)cand * G.ddt_stride;
        for (int q = tid; q < (t1e - t0e) * 15; q += KC_THREADS) {
            const int e = f_elist[t0e + q / 15], c15 = q % 15;
            f_dds[e * 15 + c15] = reinterpret_cast<const double*>(dd + e)[c15];
            if (c15 == 0) f_Se[e] = V_SCALE(tr)[np15 + e];
        }
        if (tid < 84) f_lid[tid] = lsum;
        __syncthreads();
    FH_STAMP(2);
        // ---- the sums of keyframe i (what chain_step_helper leaves in hsum) and of keyframe i - 1
        if (tid < 345) {
            double h = 0;
            h += lix >= 0 ? f_lid[GLIO_LIDAR_ACC + lix] : 0.0;
            h += v[0]; h += v[1]; h += v[2]; h += v[3]; h += v[4];
            f_h[tid] = h;
            a.hsum[(size_t)i * GLIO_CS_STRIDE + tid] = h;
            double hp = 0;
            hp += lix >= 0 ? f_lid[lix] : 0.0;
            hp += vp[0]; hp += vp[1]; hp += vp[2]; hp += vp[3]; hp += vp[4];
            f_hp[tid] = hp;
        }
        __syncthreads();
    FH_STAMP(3);
        if (fat) {
            // ---- diag(H) and g of the 45 rows (the front of k_chain_step), then the state machine's work vectors and "round 1"
            if (tid >= 384 && tid < 384 + 45) {
                double hv = 0, gg = 0;
                if (rkk >= 0 && rkk < W) {
                    const bool lidp = rlc < 6;
                    if (rk == 2) {
                        double sacc = 0;
                        sacc += lidp ? f_lid[2 * GLIO_LIDAR_ACC + kc_lidar_sym_index(lidp ? rlc : 0, lidp ? rlc : 0)] : 0.0;
                        sacc += dn[0]; sacc += dn[1]; sacc += dn[2]; sacc += dn[3]; sacc += dn[4];
                        hv = sacc;
                    } else hv = (rk == 1 ? f_h : f_hp)[kc_w(rlc, rlc)];
                    const double vl = f_lid[rk * GLIO_LIDAR_ACC + 21 + (lidp ? rlc : 0)];
                    double gacc = 0;
                    gacc += lidp ? vl : 0.0;
                    gacc += rd_.e0 >= 0 ? g5[0] : 0.0;
                    gacc += rd_.e1 >= 0 ? g5[1] : 0.0;
                    gacc += rd_.k0 >= 0 ? g5[2] : 0.0;
                    gacc += rd_.k1 >= 0 ? g5[3] : 0.0;
                    gacc += rpi >= 0 ? g5[4] : 0.0;
                    gg = gacc;
                    if (ph0) rs = tr.jacobi_scaling ? 1.0 / (1.0 + sqrt(hv)) : 1.0;          // (the first step forms the scale: tr_prepare_body, phase 0)
                    double d = rs * rs * hv;
                    d = d < 1e-6 ? 1e-6 : (d > 1e32 ? 1e32 : d);
                    const double dd_ = sqrt(d);
                    const double gs = rs * gg;
                    const double grd = gs / dd_;
                    f_S[16 * rk + rlc] = rs; f_D[16 * rk + rlc] = dd_; f_gr[16 * rk + rlc] = grd; f_G[16 * rk + rlc] = gg;
                    f_zb[16 * rk + rlc] = (rs * grd / dd_) / rs;
                } else { f_S[16 * rk + rlc] = 0.0; f_D[16 * rk + rlc] = 1.0; f_gr[16 * rk + rlc] = 0.0; f_G[16 * rk + rlc] = 0.0; f_zb[16 * rk + rlc] = 0.0; }
            }
            // the epoch unknowns: work vectors, then epoch_scalars
            for (int t = t0e + tid; t < t1e; t += KC_THREADS) {
                const int e = f_elist[t];
                const double he = f_dds[e * 15 + 12], ge = f_dds[e * 15 + 13];
                const double se = ph0 ? (tr.jacobi_scaling ? 1.0 / (1.0 + sqrt(he)) : 1.0) : f_Se[e];
                f_Se[e] = se;
                double d = se * se * he;
                d = d < 1e-6 ? 1e-6 : (d > 1e32 ? 1e32 : d);
                const double de = sqrt(d);
                const double gs = se * ge;
                const double grd = gs / de;
                f_De[e] = de; f_gre[e] = grd;
                const double we = (se * grd / de) / se;
                f_wd[e] = we;
                const double m = se * he * se + mu * de * de;
                double re;
                if (!(m > 0.0) || !isfinite(m)) { f_misc[0] = 1; re = 0.0; } else re = rsqrt(m);
                f_rd[e] = re;
                f_wdr[e] = (1.0 / re) * we;
                f_yd[e] = (se * ge) * re;
                if (f_eps[e].x < 0) f_misc[0] = 1;
            }
            if (i == 0) for (int e = tid; e < nd; e += KC_THREADS) if (f_eps[e].x < 0 || f_eps[e].y != f_eps[e].x + 1) f_misc[0] = 1;      // (an epoch nobody owns)
            __syncthreads();
    FH_STAMP(4);
            // ---- the epoch columns V = S c S / sqrt(m) of the epochs touching keyframe i (all 30 rows: the neighbour's rows feed B)
            for (int it = tid; it < (t1e - t0e) * 30; it += KC_THREADS) {
                const int e = f_elist[t0e + it / 30], q = it % 30;
                const int lc = q < 15 ? q : q - 15;
                const int k12 = kc_dop_local12(q >= 15, lc);
                const int2 sl = f_eps[e];
                const int used = reinterpret_cast<const int*>(f_dds + e * 15 + 14)[1];
                const int s1 = q >= 15 ? sl.y : sl.x;
                const bool ok = (used != 0) & (s1 >= 0) & (k12 >= 0);
                const int kx = s1 - (i - 1);                                    // which of the three keyframes the row belongs to
                const double sr = (kx >= 0 && kx < 3) ? f_S[16 * kx + lc] : 0.0;
                const double cv = f_dds[e * 15 + (k12 >= 0 ? k12 : 0)];
                const double vv = ok ? sr * cv * f_Se[e] : 0.0;
                f_Vs[e * 30 + q] = vv * f_rd[e];
            }
            // ---- the block of keyframe i: S H S + mu D^2, zeros above the diagonal of D_i, the right-hand side row; rows 15..29 of block i - 1
            // (independent of the epoch columns above: no barrier between the two)
            if (tid < 345) {
                int r30, j;
                if (tid < 120) { int r = 0; while ((r + 1) * (r + 2) / 2 <= tid) ++r; r30 = r; j = tid - r * (r + 1) / 2; }
                else { r30 = 15 + (tid - 120) / 15; j = (tid - 120) % 15; }
                const bool live = r30 < 15 || i + 1 < W;
                const double srow = r30 < 15 ? f_S[16 + r30] : f_S[32 + r30 - 15];
                double wv_ = (live ? srow : 0.0) * f_h[tid] * f_S[16 + j];
                wv_ += (r30 == j) ? mu * f_D[16 + r30] * f_D[16 + r30] : 0.0;
                f_blk[r30 * KC_RS + j] = live ? wv_ : 0.0;
                f_blk2[r30 * KC_RS + j] = live ? wv_ : 0.0;
                if (tid >= 120 && i > 0) {                 // B_{i-1}: rows of keyframe i, columns of keyframe i - 1
                    const int r = r30 - 15;
                    double wp = f_S[16 + r] * f_hp[tid] * f_S[j];
                    f_bp[r * KC_RS + j] = wp;
                }
            }
            if (tid >= 448 && tid < 448 + 15) { const int j = tid - 448; f_blk[30 * KC_RS + j] = f_S[16 + j] * f_G[16 + j]; }
            // the rows that can carry an epoch coupling are the position and velocity rows (kc_dop_local12): the corrections run over that fixed set -- a row
            // whose V is zero for every epoch of this keyframe receives a sum of zeros, as in workgroup 0's data-derived set
            if (tid == 0) { const int rows6[6] = {0, 1, 2, 6, 7, 8}; for (int q = 0; q < 6; ++q) f_misc[2 + q] = rows6[q]; f_misc[1] = 6; }
            __syncthreads();
    FH_STAMP(5);
            // ---- three things at once: wavefront 0 takes the epochs' contribution off f_blk (matrix core, as in workgroup 0); wavefront 1 forms the rows of
            // t = H u of keyframe i from the uncorrected copy; wavefront 2 the rows of t of the epochs whose first keyframe is i
            if (wv == 0 && t1e > t0e) chain_epoch_corrections_mfma(W, f_misc[1], f_misc + 2, f_eoff, f_elist, f_eps, f_Vs, f_yd, f_blk - (size_t)i * KC_BLK, lane, i, W);
            if (tid >= 64 && tid < 64 + 15) {
                const int r = tid - 64;
                double acc = 0.0;
                const double s_row = f_S[16 + r], d_row = f_D[16 + r];
#pragma unroll
                for (int j = 0; j < KC_NB; ++j) {
                    double vv = j <= r ? f_blk2[r * KC_RS + j] : f_blk2[j * KC_RS + r];
                    if (j == r) vv -= mu * d_row * d_row;
                    acc += vv * f_zb[16 + j];
                }
                if (i + 1 < W) {
#pragma unroll
                    for (int j = 0; j < KC_NB; ++j) acc += f_blk2[(KC_NB + j) * KC_RS + r] * f_zb[32 + j];
                }
                if (i > 0) {
#pragma unroll
                    for (int j = 0; j < KC_NB; ++j) acc += f_bp[r * KC_RS + j] * f_zb[j];
                }
                for (int t = t0e; t < t1e; ++t) {
                    const int e = f_elist[t];
                    const int sd = f_eps[e].x == i ? 0 : 15;
                    acc = acc + f_Vs[e * 30 + sd + r] * f_wdr[e];
                }
                f_t[r] = acc / s_row;
            }
            if (tid >= 128 && tid < 128 + (t1e - t0e)) {
                const int e = f_elist[t0e + tid - 128];
                const int2 sl = f_eps[e];
                if (sl.x == i) {
                    const double se = f_Se[e];
                    double acc = se * f_dds[e * 15 + 12] * se * f_wd[e];
                    const double ire = 1.0 / f_rd[e];
#pragma unroll
                    for (int q = 0; q < 30; ++q) acc += (f_Vs[e * 30 + q] * ire) * f_zb[16 * ((q < 15 ? sl.x : sl.y) - (i - 1)) + (q < 15 ? q : q - 15)];
                    a.fat_ep[(size_t)e * KC_FAT_EP + 32] = acc / se;
                }
            }
            __syncthreads();
    FH_STAMP(6);
            if (f_misc[0]) fat = false;
            if (fat) {
                double* ob = a.fat_blk + (size_t)i * KC_BLK;
                if (tid < 345) {
                    int r30, j;
                    if (tid < 120) { int r = 0; while ((r + 1) * (r + 2) / 2 <= tid) ++r; r30 = r; j = tid - r * (r + 1) / 2; }
                    else { r30 = 15 + (tid - 120) / 15; j = (tid - 120) % 15; }
                    ob[tid] = f_blk[r30 * KC_RS + j];
                } else if (tid < 360) ob[tid] = f_blk[30 * KC_RS + (tid - 345)];
                else if (tid < 375) ob[tid] = f_t[tid - 360];
                for (int it = tid; it < (t1e - t0e) * 32; it += KC_THREADS) {
                    const int e = f_elist[t0e + it / 32], q = it % 32;
                    if (f_eps[e].x != i) continue;
                    a.fat_ep[(size_t)e * KC_FAT_EP + q] = q < 30 ? f_Vs[e * 30 + q] : (q == 30 ? f_rd[e] : f_yd[e]);
                }
            }
        }
    } else fat = false;
    // ONE agent-scope release: the barrier puts every wavefront's stores before thread 0's release store, whose write-back of this XCD's L2 carries them all
    // (a __threadfence() by every wavefront in front of the barrier and the release store behind it were two write-backs: 1.5 us each, measured)
    __syncthreads();
    FH_STAMP(7);
    if (tid == 0) __hip_atomic_store(&a.hdone[i], 4 * a.hseq + (fat ? 2 : 0) + go, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#undef FH_STAMP
}

__global__ __launch_bounds__(KC_THREADS) void k_chain_step(const ChainArgs a, const TrArgs tr, const GatherArgs G) {
    static_assert(KC_THREADS == TR_THREADS, "tr_prepare_body runs with the chain kernel's workgroup");
    if (blockIdx.x > 0) {
        if (a.fat) chain_step_fat_helper(a, tr, G, (int)blockIdx.x - 1); else chain_step_helper(a, tr, G, (int)blockIdx.x - 1);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int W = a.W, n = a.n, nd = a.nd;
    const int np15 = 15 * W;
    // ---- carve (dynamic LDS).  The gather tables sit BEHIND the region tr_factor_body / tr_dogleg_body overlay at the
    // front of tr_lds: they are still needed by the dense fallback.
    double* rd = reinterpret_cast<double*>(tr_lds);
    double* yd = rd + nd + (nd & 1);
    double* Vs = yd + nd + (nd & 1);
    double* Blk = Vs + (size_t)nd * 30;
    double* CsT = Blk + (size_t)W * KC_BLK;
    double* CsB = CsT + 288;
    double* zb = CsB + 288;
    int2* eps = reinterpret_cast<int2*>(zb + 15 * W + (W & 1));
    int* eoff = reinterpret_cast<int*>(eps + nd + 2);
    int* elist = eoff + ((W + 2) & ~1) + 2;
    int* esd = elist + 2 * nd + 2;
    int* eoth = esd + 2 * nd + 2;
    int* misc = eoth + 2 * nd + 2;
    double* wd = reinterpret_cast<double*>(misc + 24);
    double* wdr = reinterpret_cast<double*>(esd);        // [nd] (u / s) sqrt(m) for t = H u: lives where the index lists go afterwards
    double* gt_base = reinterpret_cast<double*>(tr_lds + chain_step_tabs_offset(W, nd, n));
    double* lid = gt_base;                               // [W][28]
    const int n2 = n + (n & 1), nx = n + W, nx2 = nx + (nx & 1);
    double* sS = lid + W * GLIO_LIDAR_ACC;               // [n] Jacobi scale       (staged copies of the work vectors)
    double* sDg = sS + n2;                               // [n] D = sqrt(clamp(diag))
    const bool mir = a.fast != 0;                        // (host: 0 when the mirrors below do not fit beside the blocks)
    double* sG = sDg + n2;                               // [n] g of the current point            -+
    double* sGr = sG + n2;                               // [n] g~ = S g / D                       | only with mir
    double* sT = sGr + n2;                               // [n] t = H u                            |
    double* sX = sT + n2;                                // [16 W + nd] the current point         -+
    double* dds = mir ? sX + nx2 : sDg + n2;             // [nd][15] clock-drift blocks: c[12], h, g, (group, used)
    // scratch of the front: state buffer 0 sits in sX, buffer 1 in [CsT, CsB) (free until the chain), the candidate's diag(H) in sDg (the
    // state machine reads entry i before it overwrites it with D_i)
    double* xm0 = sX; double* xm1 = CsT; double* sHd = sDg;
    __shared__ double s_cost_lds;
    double* sCost = &s_cost_lds;
    short* stab = reinterpret_cast<short*>(dds + (size_t)nd * 15 + (nd & 1));
    short* wr30 = stab + (8 + 15) * W + ((8 + 15) * W & 1);      // [345] chain-layout index -> row (0..29), column, LiDAR packed index or -1
    short* wj = wr30 + 346;
    short* wlx = wj + 346;
    GatherTabs T;
    T.lid = lid; T.kd = reinterpret_cast<const ChainKf*>(stab); T.pidx = stab + 8 * W;
    __shared__ int rowmask;
    __shared__ int s_pending, s_cand, s_done;
    __shared__ int s_prog[4];
    __shared__ SolverStatus s_in, s_full;
    AR_STAMP(40);
#ifdef GLIO_DEV_STAMPS
    if (tid == 0) a.dbg[120] = clock64();          // shader-clock counter beside the 100 MHz wall clock: the clock the workgroup actually runs at
#endif
    // ---- round 0: status, the host-built gather tables, the structure tables of the epochs
    if (tid == 0) { s_in = *tr.status; s_done = s_in.done; s_pending = s_in.cand_pending; s_cand = 1 - s_in.cur; }
    if (tid < 18) misc[tid] = 0;            // [0] breakdown flag, [1] number of rows with an epoch coupling, [2..] their indices
    if (tid == 0) rowmask = 0;
    if ((a.fast & 2) && n <= KC_THREADS && nx2 <= 2 * 288) {        // what the state machine will want, in the same round trip: both state buffers and the Jacobi scale
        for (int k = tid; k < nx; k += KC_THREADS) { xm0[k] = tr.x0[k]; xm1[k] = tr.x1[k]; }
        for (int k = tid; k < n; k += KC_THREADS) sS[k] = V_SCALE(tr)[k];
        for (int t = tid; t < 2 * nd; t += KC_THREADS) elist[t] = a.ep_list[t];
    }
    for (int k = tid; k < (8 + 15) * W; k += KC_THREADS) stab[k] = G.tabs[k];
    for (int w = tid; w < 345; w += KC_THREADS) {
        int r30, j;
        if (w < 120) { int r = 0; while ((r + 1) * (r + 2) / 2 <= w) ++r; r30 = r; j = w - r * (r + 1) / 2; }
        else { r30 = 15 + (w - 120) / 15; j = (w - 120) % 15; }
        wr30[w] = (short)r30; wj[w] = (short)j;
        wlx[w] = (short)((r30 < 6 && j < 6) ? kc_lidar_sym_index(j, r30) : -1);      // j <= r30 in the lower triangle
    }
    for (int e = tid; e < nd; e += KC_THREADS) eps[e] = a.ep_slots[e];
    for (int i = tid; i <= W; i += KC_THREADS) eoff[i] = a.ep_off[i];
    GLIO_BLOCK_LDS_SYNC();              // (LDS-only barriers wherever the hand-over is through LDS: see glio_device.h)
    if (s_done) return;
    const int cand = s_cand;
    AR_STAMP(41);
    // The keyframe blocks: 345 entries per keyframe (lower triangle of D_i, all of B_i), each the LiDAR block + the five slices the factor
    // roles wrote in this layout (coalesced, same index in every slice), seven items per thread in flight.  gather_load issues the loads of
    // one batch, gather_store adds them up (k_assemble's order), scales (S H S + mu D^2) and stores.  (Measured without gain: gathering
    // into LDS before the state machine and scaling afterwards; holding the first batch in registers across the state machine; skipping
    // the slices that are zero by the graph's structure, ~45 % of the 276 KB -- the per-slice branches cost what the loads save.)
    // An item is a PAIR of adjacent entries (w, w + 1) of a slice, read as one 16-byte load: 35 dwordx4 loads per thread in ONE round instead of
    // 70 dwordx2 loads in two.  (Slices are 352 doubles apart and 16-byte aligned; entry 345, read with 344, is padding.)  The phase stays at
    // ~6.5 us either way, and also with one packed table word and real branches in the store loop: it is neither the load instructions nor the
    // LDS reads of the store loop -- what one CU can pull from the other XCDs' results (276 KB) sets it.
    constexpr int KB = 7;
    constexpr int GPAIRS = 173;                          // pairs per keyframe slice: entries 0..345
    const int gtotal = W * GPAIRS;
    auto gather_load = [&](const double* src, const int q0, v2f64 (&v)[KB][5]) {
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            const int q = q0 + u * KC_THREADS;
            const int qq = q < gtotal ? q : tid;
            const int i = qq / GPAIRS, w = 2 * (qq - GPAIRS * i);
            const double* p = src + (size_t)i * GLIO_CS_SOURCES * GLIO_CS_STRIDE + w;
#pragma unroll
            for (int sidx = 0; sidx < 5; ++sidx) v[u][sidx] = *reinterpret_cast<const v2f64*>(p + sidx * GLIO_CS_STRIDE);
        }
    };
    auto gather_store = [&](const int q0, const v2f64 (&v)[KB][5], const double mu_) {
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            const int q = q0 + u * KC_THREADS;
            if (q >= gtotal) continue;
            const int i = q / GPAIRS, w0 = 2 * (q - GPAIRS * i);
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int w = w0 + h2;
                if (w >= 345) continue;
                const int r30 = wr30[w], j = wj[w], lix = wlx[w];
                const bool live = r30 < 15 || i + 1 < W;
                const int irow = 15 * i + r30, icol = 15 * i;
                double h = 0;
                h += lix >= 0 ? lid[i * GLIO_LIDAR_ACC + (lix >= 0 ? lix : 0)] : 0.0;
                h += v[u][0][h2]; h += v[u][1][h2]; h += v[u][2][h2]; h += v[u][3][h2]; h += v[u][4][h2];
                double wv_ = (live ? sS[irow] : 0.0) * h * sS[icol + j];
                wv_ += (irow == icol + j) ? mu_ * sDg[irow] * sDg[irow] : 0.0;
                Blk[(size_t)i * KC_BLK + r30 * KC_RS + j] = live ? wv_ : 0.0;
            }
        }
    };
    // ---- diag(H), g and the cost of the candidate, for the state machine
    const bool ff = (a.fast & 2) && s_pending && n <= KC_THREADS && nx2 <= 2 * 288;      // the front from LDS (one item per thread; state buffer 1 fits [CsT, CsB))
    if (ff) {
        // every global load of this phase goes out before the first barrier: the slices' diagonal entries, the gradient parts, the costs
        // and the clock-drift blocks travel together with the K3 partials instead of one round trip after the other
        double* hd = cand ? G.hd1 : G.hd0;
        double* gv = cand ? G.g1 : G.g0;
        const DdtBlock* dd = G.ddt_blocks + (size_t)cand * G.ddt_stride;
        const PairBlock* imu = G.imu_blocks + (size_t)cand * W;
        const PairBlock* gnb = G.gnss_blocks + (size_t)cand * G.gnss_stride;
        const int c = tid < n ? tid : 0;
        const bool pose = c < np15;
        const int sc = pose ? c / 15 : 0, lc = pose ? c - 15 * sc : 0, e = pose ? 0 : c - np15;
        const ChainKf d = T.kd[sc];
        const int pi = T.pidx[15 * sc + lc];
        double e5[5], g5[5];
        {
            const double* p = G.chain_src + (((size_t)cand * W + sc) * GLIO_CS_SOURCES) * GLIO_CS_STRIDE + kc_w(lc, lc);
#pragma unroll
            for (int k = 0; k < 5; ++k) e5[k] = p[k * GLIO_CS_STRIDE];
            g5[0] = imu[d.e0 >= 0 ? d.e0 : 0].g[lc];
            g5[1] = imu[d.e1 >= 0 ? d.e1 : 0].g[15 + lc];
            g5[2] = gnb[d.k0 >= 0 ? d.k0 : 0].g[(d.o0 ? 0 : 15) + lc];
            g5[3] = gnb[d.k1 >= 0 ? d.k1 : 0].g[(d.o1 ? 0 : 15) + lc];
            g5[4] = (G.pg + (size_t)cand * G.np)[pi >= 0 ? pi : 0];
        }
        const double dh = nd > 0 ? dd[e].h : 0.0, dg = nd > 0 ? dd[e].g : 0.0;
        double ci0 = 0, cg0 = 0, cp0 = 0;
        if (wv == KC_THREADS / 64 - 1) {
            ci0 = lane < G.n_imu ? imu[lane].cost : 0.0;
            cg0 = lane < G.n_groups ? gnb[lane].cost : 0.0;
            if (lane == 0 && G.has_prior) cp0 = G.pcost[cand];
        }
        for (int k = tid; k < nd * 15; k += KC_THREADS) dds[k] = reinterpret_cast<const double*>(dd)[k];
        AR_STAMP(74);
        kc_reduce_lidar(G, T, cand, W);
        AR_STAMP(75);
        GLIO_BLOCK_LDS_SYNC();
        AR_STAMP(76);
        if (tid < n) {
            double hv, gg_;
            if (pose) {
                const bool lidp = lc < 6;
                double sacc = 0;        // kc_gather_entry's order: LiDAR, the five slices
                sacc += lidp ? T.lid[sc * GLIO_LIDAR_ACC + kc_lidar_sym_index(lidp ? lc : 0, lidp ? lc : 0)] : 0.0;
                sacc += e5[0]; sacc += e5[1]; sacc += e5[2]; sacc += e5[3]; sacc += e5[4];
                hv = sacc;
                const double vl = T.lid[sc * GLIO_LIDAR_ACC + 21 + (lidp ? lc : 0)];
                double gacc = 0;        // kc_gather_grad's order
                gacc += lidp ? vl : 0.0;
                gacc += d.e0 >= 0 ? g5[0] : 0.0;
                gacc += d.e1 >= 0 ? g5[1] : 0.0;
                gacc += d.k0 >= 0 ? g5[2] : 0.0;
                gacc += d.k1 >= 0 ? g5[3] : 0.0;
                gacc += pi >= 0 ? g5[4] : 0.0;
                gg_ = gacc;
            } else { hv = dh; gg_ = dg; }
            hd[tid] = hv; gv[tid] = gg_;
            sHd[tid] = hv; sG[tid] = gg_;
        }
        if (wv == KC_THREADS / 64 - 1) {          // total cost: the same lane assignment and wave reduction as k_assemble
            double cs = 0;
            for (int k = lane; k < W; k += 64) cs += T.lid[k * GLIO_LIDAR_ACC + 27];
            for (int k = lane; k < G.n_imu; k += 64) cs += k == lane ? ci0 : imu[k].cost;
            for (int k = lane; k < G.n_groups; k += 64) cs += k == lane ? cg0 : gnb[k].cost;
            if (lane == 0 && G.has_prior) cs += cp0;
            cs = wave_sum(cs);
            if (lane == 0) { *(cand ? G.c1 : G.c0) = cs; *sCost = cs; }
        }
        // (no fence: nothing of this is read back from global memory in this launch)
    } else if (s_pending) {
        kc_reduce_lidar(G, T, cand, W);
        __syncthreads();
        double* hd = cand ? G.hd1 : G.hd0;
        double* gv = cand ? G.g1 : G.g0;
        const DdtBlock* dd = G.ddt_blocks + (size_t)cand * G.ddt_stride;
        for (int c = tid; c < n; c += KC_THREADS) {
            if (c < np15) {
                const int sc = c / 15, lc = c - 15 * sc;
                hd[c] = kc_gather_entry(G, T, cand, W, sc, lc, lc);      // diagonal entry of D
                gv[c] = kc_gather_grad(G, T, cand, W, sc, lc);
            } else { hd[c] = dd[c - np15].h; gv[c] = dd[c - np15].g; }
        }
        if (tid < 64) {          // total cost: the same lane assignment and wave reduction as k_assemble
            const PairBlock* imu = G.imu_blocks + (size_t)cand * W;
            const PairBlock* gn = G.gnss_blocks + (size_t)cand * G.gnss_stride;
            double cs = 0;
            for (int k = tid; k < W; k += 64) cs += T.lid[k * GLIO_LIDAR_ACC + 27];
            for (int k = tid; k < G.n_imu; k += 64) cs += imu[k].cost;
            for (int k = tid; k < G.n_groups; k += 64) cs += gn[k].cost;
            if (tid == 0 && G.has_prior) cs += G.pcost[cand];
            cs = wave_sum(cs);
            if (tid == 0) *(cand ? G.c1 : G.c0) = cs;
        }
        __threadfence();
    }
    if (ff) GLIO_BLOCK_LDS_SYNC(); else __syncthreads();
    // ---- the helper workgroups: every one of them has read the status record (and left its sums) before the state machine rewrites that record.
    // With the LDS-resident front the state machine checks their completion words itself, just before that store (PrepMirror); the generic body
    // has no such hook: wait here.
    __shared__ int s_hprod, s_hfat;
    if (tid == 0) s_hfat = 0;
    if (a.helpers) {
        if (tid == 0) { s_hprod = 1; s_hfat = (a.fat && ff) ? 1 : 0; }
        GLIO_BLOCK_LDS_SYNC();
        if (!ff) {
            if (tid < a.helpers) {
                int v, polls = 0;          // (bounded like the state machine's wait: see tr_prepare_body)
                while (((v = __hip_atomic_load(&a.hdone[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 2) != a.hseq && polls < a.hpolls) { ++polls; __builtin_amdgcn_s_sleep(1); }
                if ((v >> 2) != a.hseq || !(v & 1)) s_hprod = 0;
            }
            GLIO_BLOCK_LDS_SYNC();
        }
    }
    AR_STAMP(42);
    // ---- state machine (reads the vectors just written: nothing of them was loaded earlier in this kernel)
    TrDecision dec;
    dec.full = &s_full;
    if (ff) {
        PrepMirror pm;
        pm.st_in = &s_in; pm.x0 = xm0; pm.x1 = xm1; pm.hd = sHd; pm.g = sG; pm.cost = sCost; pm.scale = sS; pm.diag = sDg; pm.grad = sGr; pm.dbg = a.dbg;
        pm.hdone = a.hdone; pm.hseq = a.hseq; pm.helpers = a.helpers; pm.hprod = &s_hprod; pm.hpolls = a.hpolls; pm.hfat = &s_hfat;
        if (!tr_prepare_body<true>(tr, &dec, &pm)) return;
    } else if (!tr_prepare_body(tr, &dec)) return;
    AR_STAMP(43);
    bool fast_tail = false;
    if (!dec.reuse) {
    if (tid == 0) *a.flag = 0;
    if (!(s_pending && dec.cur == cand)) { kc_reduce_lidar(G, T, dec.cur, W); }      // (invalid step earlier: the current point's blocks again)
    const double* gn = dec.cur ? tr.g1 : tr.g0;
    const DdtBlock* ddg = G.ddt_blocks + (size_t)dec.cur * G.ddt_stride;
    const double mu = dec.mu;
    // ---- fat helpers delivered and their assumptions hold (the candidate was accepted, mu is what an acceptance leaves): the blocks, the epoch columns and
    // t = H u are LOADED -- 60 KB of finished numbers instead of four LDS phases (epoch columns, block store, t = H u, epoch corrections: ~15 us)
    const bool fat_path = a.fat && a.helpers && s_hprod && s_hfat && ff && s_pending && dec.cur == cand && !tr.lm &&
                          __double_as_longlong(mu) == __double_as_longlong(s_in.phase == 0 ? s_in.mu : fmax(1e-8, 2.0 * s_in.mu / 10.0));
    if (fat_path) {
        if (tid == 0 && a.dbg) a.dbg[300] += 1;            // (steps that took the fat helpers' products: glio_debug_arrow_stamps, slot 300)
        const double* xm = cand ? xm1 : xm0;
        for (int k = tid; k < nx; k += KC_THREADS) sX[k] = xm[k];
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        constexpr int FP = KC_FAT_ENT / 2;                   // pairs per keyframe
        constexpr int KF = 8, KE = 5;
        // ONE round of loads (blocks, rows of t, epoch products: every load is issued before the first LDS store), for windows that fit the batch; else loops
        if (W * FP <= KF * KC_THREADS && np15 + nd <= KC_THREADS && nd * 32 <= KE * KC_THREADS) {
            v2f64 hv[KF]; double tvv, ev[KE];
#pragma unroll
            for (int u = 0; u < KF; ++u) {
                const int q = tid + u * KC_THREADS;
                const int qq = q < W * FP ? q : tid;
                const int i = qq / FP, w = 2 * (qq - FP * i);
                hv[u] = *reinterpret_cast<const v2f64*>(a.fat_blk + (size_t)i * KC_BLK + w);
            }
            {
                const int row = tid < np15 + nd ? tid : 0;
                if (row < np15) { const int i = row / 15; tvv = a.fat_blk[(size_t)i * KC_BLK + KC_FAT_ENT + (row - 15 * i)]; }
                else tvv = a.fat_ep[(size_t)(row - np15) * KC_FAT_EP + 32];
            }
#pragma unroll
            for (int u = 0; u < KE; ++u) {
                const int it = tid + u * KC_THREADS;
                const int itc = it < nd * 32 ? it : 0;
                ev[u] = nd > 0 ? a.fat_ep[(size_t)(itc >> 5) * KC_FAT_EP + (itc & 31)] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < KF; ++u) {
                const int q = tid + u * KC_THREADS;
                if (q >= W * FP) continue;
                const int i = q / FP, w0 = 2 * (q - FP * i);
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int w = w0 + h2;
                    const int off = w < 345 ? wr30[w] * KC_RS + wj[w] : 30 * KC_RS + (w - 345);
                    Blk[(size_t)i * KC_BLK + off] = hv[u][h2];
                }
            }
            if (tid < np15 + nd) { V_T(tr)[tid] = tvv; if (mir) sT[tid] = tvv; }
#pragma unroll
            for (int u = 0; u < KE; ++u) {
                const int it = tid + u * KC_THREADS;
                if (it >= nd * 32) continue;
                const int e = it >> 5, q = it & 31;
                if (q < 30) Vs[e * 30 + q] = ev[u]; else if (q == 30) rd[e] = ev[u]; else yd[e] = ev[u];
            }
        } else {
        for (int q0 = tid; q0 < W * FP; q0 += KF * KC_THREADS) {
            v2f64 hv[KF];
#pragma unroll
            for (int u = 0; u < KF; ++u) {
                const int q = q0 + u * KC_THREADS;
                const int qq = q < W * FP ? q : tid;
                const int i = qq / FP, w = 2 * (qq - FP * i);
                hv[u] = *reinterpret_cast<const v2f64*>(a.fat_blk + (size_t)i * KC_BLK + w);
            }
#pragma unroll
            for (int u = 0; u < KF; ++u) {
                const int q = q0 + u * KC_THREADS;
                if (q >= W * FP) continue;
                const int i = q / FP, w0 = 2 * (q - FP * i);
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int w = w0 + h2;
                    const int off = w < 345 ? wr30[w] * KC_RS + wj[w] : 30 * KC_RS + (w - 345);
                    Blk[(size_t)i * KC_BLK + off] = hv[u][h2];
                }
            }
        }
        for (int row = tid; row < np15 + nd; row += KC_THREADS) {
            double tv;
            if (row < np15) { const int i = row / 15; tv = a.fat_blk[(size_t)i * KC_BLK + KC_FAT_ENT + (row - 15 * i)]; }
            else tv = a.fat_ep[(size_t)(row - np15) * KC_FAT_EP + 32];
            V_T(tr)[row] = tv; if (mir) sT[row] = tv;
        }
        for (int it = tid; it < nd * 32; it += KC_THREADS) {
            const int e = it >> 5, q = it & 31;
            const double vv = a.fat_ep[(size_t)e * KC_FAT_EP + q];
            if (q < 30) Vs[e * 30 + q] = vv; else if (q == 30) rd[e] = vv; else yd[e] = vv;
        }
        }
        // the strict upper triangle of D_i is read as zero by the chain steps
        for (int q0 = tid; q0 < W * 105; q0 += KC_THREADS) {
            const int i = q0 / 105, w = q0 - 105 * i;
            const int r = wr30[w], c = wj[w];
            Blk[i * KC_BLK + c * KC_RS + (r + 1)] = 0.0;
        }
        GLIO_BLOCK_LDS_SYNC();
    } else {
    auto nat = [&](const int p) { return p < nd ? np15 + p : p - nd; };
    static_assert(sizeof(DdtBlock) == 15 * sizeof(double), "DdtBlock staged as 15 doubles");
    auto epoch_scalars = [&](const int e, const double we) {          // w_e = u / s, r_e = 1 / sqrt(m_e), w_e / r_e
        wd[e] = we;
        const int ie = np15 + e;
        const double se = sS[ie], he = dds[e * 15 + 12], de = sDg[ie];
        const double m = se * he * se + mu * de * de;
        double re;
        if (!(m > 0.0) || !isfinite(m)) { misc[0] = 1; re = 0.0; } else re = rsqrt(m);
        rd[e] = re;
        wdr[e] = (1.0 / re) * we;
    };
    if (ff && dec.cur == cand) {
        // round 1 without a global load: scale, D, g~, g and the clock-drift blocks are in LDS already; u = S g~ / D as the state machine
        // formed it, w = u / S and the epochs' scalars in the same pass (one barrier instead of two)
        const double* xm = cand ? xm1 : xm0;
        for (int k = tid; k < nx; k += KC_THREADS) sX[k] = xm[k];
        for (int k = tid; k < np15; k += KC_THREADS) zb[k] = (sS[k] * sGr[k] / sDg[k]) / sS[k];
        for (int e = tid; e < nd; e += KC_THREADS) epoch_scalars(e, (sS[np15 + e] * sGr[np15 + e] / sDg[np15 + e]) / sS[np15 + e]);
    } else {
        // round 1: the work vectors, the clock-drift blocks (coalesced, 15 doubles each) and the epoch list into LDS
        const double* xg = dec.cur ? tr.x1 : tr.x0;
        for (int k = tid; k < n; k += KC_THREADS) { sS[k] = V_SCALE(tr)[k]; sDg[k] = V_DIAG(tr)[k]; }
        if (mir) {
            for (int k = tid; k < n; k += KC_THREADS) { sGr[k] = V_GRAD(tr)[k]; sG[k] = gn[k]; }
            for (int k = tid; k < nx; k += KC_THREADS) sX[k] = xg[k];
        }
        for (int k = tid; k < nd * 15; k += KC_THREADS) dds[k] = reinterpret_cast<const double*>(ddg)[k];
        for (int t = tid; t < eoff[W]; t += KC_THREADS) elist[t] = a.ep_list[t];
        const double* uv = V_U(tr);
        for (int k = tid; k < np15; k += KC_THREADS) zb[k] = uv[k];
        for (int e = tid; e < nd; e += KC_THREADS) wd[e] = uv[np15 + e];
        GLIO_BLOCK_LDS_SYNC();
        for (int k = tid; k < np15; k += KC_THREADS) zb[k] = zb[k] / sS[k];
        for (int e = tid; e < nd; e += KC_THREADS) epoch_scalars(e, wd[e] / sS[np15 + e]);
    }
    auto Rld = [&](const int p) { const int i = nat(p); return mir ? sS[i] * sG[i] : sS[i] * gn[i]; };
    GLIO_BLOCK_LDS_SYNC();
    AR_STAMP(90);
    AR_STAMP(91);
    int mk_rows = 0;
    // one (epoch, row) pair per item, five items of a thread at a time: what depends on (e, row) only goes out as one batch of LDS reads, the
    // scale of the keyframe row (which needs the epoch's slot pair) as a second (item by item the compiler walks the chains one after the other)
    for (int it0 = tid; it0 < nd * 30; it0 += 5 * KC_THREADS) {
        int2 sl[5]; int used[5], k12[5], lc[5], ee[5]; double cv[5], se[5], re[5]; bool in[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int it = it0 + u * KC_THREADS;
            in[u] = it < nd * 30;
            const int itc = in[u] ? it : tid;
            const int e = itc / 30, q = itc - 30 * e;
            ee[u] = e; lc[u] = q < 15 ? q : q - 15;
            k12[u] = kc_dop_local12(q >= 15, lc[u]);
            sl[u] = eps[e];
            used[u] = reinterpret_cast<const int*>(dds + e * 15 + 14)[1];
            cv[u] = dds[e * 15 + (k12[u] >= 0 ? k12[u] : 0)]; se[u] = sS[np15 + e]; re[u] = rd[e];
            lc[u] |= (q >= 15) << 8;                              // (bit 8: the row belongs to the epoch's second keyframe)
        }
        double sr[5]; bool ok[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int s1 = (lc[u] >> 8) ? sl[u].y : sl[u].x;
            ok[u] = (used[u] != 0) & (s1 >= 0) & (k12[u] >= 0);
            sr[u] = sS[15 * (s1 >= 0 ? s1 : 0) + (lc[u] & 255)];
        }
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            if (!in[u]) continue;
            const double v = ok[u] ? sr[u] * cv[u] * se[u] : 0.0;
            Vs[it0 + u * KC_THREADS] = v * re[u];
            if (v != 0.0) mk_rows |= 1 << ((lc[u] & 255) % 15);
        }
    }
    AR_STAMP(92);
    {   // rows that carry an epoch coupling: one LDS atomic per wavefront
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mk_rows |= __shfl_xor(mk_rows, off, 64);
        if (lane == 0 && mk_rows) atomicOr(&rowmask, mk_rows);
    }
    AR_STAMP(93);
    for (int e = tid; e < nd; e += KC_THREADS) yd[e] = Rld(e) * rd[e];
    AR_STAMP(44);
    if (a.helpers && s_hprod && s_pending && dec.cur == cand) {
        // the sums the helper workgroups left (same entries, same order of additions): 55 KB in one round of 16-byte loads instead of 276 KB.
        // (Requesting them before the state machine and holding 36 registers across it was measured: the phases in between got slower by what this
        //  one gained.)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // (every wavefront: the completion words were read by the first W threads only)
        constexpr int KH = 7;
        for (int q0 = tid; q0 < gtotal; q0 += KH * KC_THREADS) {
            v2f64 hv[KH];
#pragma unroll
            for (int u = 0; u < KH; ++u) {
                const int q = q0 + u * KC_THREADS;
                const int qq = q < gtotal ? q : tid;
                const int i = qq / GPAIRS, w = 2 * (qq - GPAIRS * i);
                hv[u] = *reinterpret_cast<const v2f64*>(a.hsum + (size_t)i * GLIO_CS_STRIDE + w);
            }
#pragma unroll
            for (int u = 0; u < KH; ++u) {
                const int q = q0 + u * KC_THREADS;
                if (q >= gtotal) continue;
                const int i = q / GPAIRS, w0 = 2 * (q - GPAIRS * i);
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int w = w0 + h2;
                    if (w >= 345) continue;
                    const int r30 = wr30[w], j = wj[w];
                    const bool live = r30 < 15 || i + 1 < W;
                    const int irow = 15 * i + r30, icol = 15 * i;
                    const double h = hv[u][h2];
                    double wv_ = (live ? sS[irow] : 0.0) * h * sS[icol + j];
                    wv_ += (irow == icol + j) ? mu * sDg[irow] * sDg[irow] : 0.0;
                    Blk[(size_t)i * KC_BLK + r30 * KC_RS + j] = live ? wv_ : 0.0;
                }
            }
        }
    } else {
        const double* src = G.chain_src + (size_t)dec.cur * W * GLIO_CS_SOURCES * GLIO_CS_STRIDE;
        for (int q0 = tid; q0 < gtotal; q0 += KB * KC_THREADS) {
            v2f64 gb[KB][5];
            gather_load(src, q0, gb);
            gather_store(q0, gb, mu);
        }
    }
    // the strict upper triangle of D_i is read as zero by the chain steps
    for (int q0 = tid; q0 < W * 105; q0 += 5 * KC_THREADS) {             // (five items per round: their table reads as one batch)
        int off[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int q = q0 + u * KC_THREADS < W * 105 ? q0 + u * KC_THREADS : q0;
            const int i = q / 105, w = q - 105 * i;
            const int r = wr30[w], c = wj[w];           // pair (r + 1, c) with c <= r  ->  entry [c][r + 1] above the diagonal
            off[u] = i * KC_BLK + c * KC_RS + (r + 1);
        }
#pragma unroll
        for (int u = 0; u < 5; ++u) Blk[off[u]] = 0.0;
    }
    for (int q = tid; q < W * 15; q += KC_THREADS) { const int i = q / 15, j = q - 15 * i; Blk[(size_t)i * KC_BLK + 30 * KC_RS + j] = Rld(nd + 15 * i + j); }
    GLIO_BLOCK_LDS_SYNC();
    AR_STAMP(45);
    // (the list of rows with an epoch coupling, for the corrections below: by the last thread, which has no row of t when n < 512)
    if (tid == KC_THREADS - 1) { int na = 0; for (int q = 0; q < 15; ++q) if (rowmask >> q & 1) misc[2 + na++] = q; misc[1] = na; }
    for (int row = tid; row < np15 + nd; row += KC_THREADS) {
        double acc = 0.0;
        const double s_row = sS[row], d_row = sDg[row];
        if (row < np15) {
            const int i = row / 15, r = row - 15 * i;
            const double* Bi = Blk + (size_t)i * KC_BLK;
#pragma unroll
            for (int j = 0; j < KC_NB; ++j) {
                double v = j <= r ? Bi[r * KC_RS + j] : Bi[j * KC_RS + r];
                if (j == r) v -= mu * d_row * d_row;
                acc += v * zb[15 * i + j];
            }
            if (i + 1 < W) {
#pragma unroll
                for (int j = 0; j < KC_NB; ++j) acc += Bi[(KC_NB + j) * KC_RS + r] * zb[15 * (i + 1) + j];
            }
            if (i > 0) {
                const double* Bp = Blk + (size_t)(i - 1) * KC_BLK;
#pragma unroll
                for (int j = 0; j < KC_NB; ++j) acc += Bp[(KC_NB + r) * KC_RS + j] * zb[15 * (i - 1) + j];
            }
            // the epochs of this keyframe, eight at a time: the list entries, then their slot pairs, then the operands go out as three
            // batches of independent LDS reads (one after the other they are three dependent round trips PER EPOCH); added in list order
            const int t0e = eoff[i], t1e = eoff[i + 1];
            for (int tb = t0e; tb < t1e; tb += 8) {
                int ee[8], sd[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) ee[q] = elist[tb + q < t1e ? tb + q : t1e - 1];
#pragma unroll
                for (int q = 0; q < 8; ++q) sd[q] = eps[ee[q]].x == i ? 0 : 15;
                double xv[8], xw[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { xv[q] = Vs[ee[q] * 30 + sd[q] + r]; xw[q] = wdr[ee[q]]; }
#pragma unroll
                for (int q = 0; q < 8; ++q) acc = tb + q < t1e ? acc + xv[q] * xw[q] : acc;
            }
            { const double tv = acc / s_row; V_T(tr)[row] = tv; if (mir) sT[row] = tv; }
        } else {
            const int e = row - np15;
            const int2 sl = eps[e];
            const double se = s_row;
            acc = se * dds[e * 15 + 12] * se * wd[e];
            if (sl.x >= 0) {
                const double ire = 1.0 / rd[e];             // one division per epoch, not one per entry
#pragma unroll
                for (int q = 0; q < 30; ++q) acc += (Vs[e * 30 + q] * ire) * zb[15 * (q < 15 ? sl.x : sl.y) + (q < 15 ? q : q - 15)];
            }
            { const double tv = acc / se; V_T(tr)[row] = tv; if (mir) sT[row] = tv; }
        }
    }
    AR_STAMP(94);
    GLIO_BLOCK_LDS_SYNC();
    AR_STAMP(46);
    AR_STAMP(95);
    // Minus the epochs' contribution D_i -= V V^T, B_i -= V' V^T, rhs_i -= y V over the epochs touching keyframe i, in list order: on the
    // matrix core, one wavefront per keyframe (chain_epoch_corrections_mfma).  The flat entry list below stays for row sets that do not
    // fit a 16-row tile.  (Measured without gain before that: threads that hold the keyframe's epoch list in registers and take a share of
    // its entries -- the flat form is bound by its ~1000 wavefront-level LDS operand reads, not by the index chains.)
    if (2 * misc[1] + 1 <= 16) {
        AR_STAMP(96);
        chain_epoch_corrections_mfma(W, misc[1], misc + 2, eoff, elist, eps, Vs, yd, Blk, lane, wv, KC_THREADS / 64, a.dbg);
        AR_STAMP(110);
    } else {
    for (int i = wv; i < W; i += KC_THREADS / 64)
        for (int t = eoff[i] + lane; t < eoff[i + 1]; t += 64) {
            const int e = elist[t];
            const int2 sl = eps[e];
            const int side = sl.x == i ? 0 : 15;
            esd[t] = e * 30 + side;
            eoth[t] = (sl.x == i ? sl.y : sl.x) == i + 1 ? e * 30 + (15 - side) : -1;
        }
    GLIO_BLOCK_LDS_SYNC();
    AR_STAMP(96);
    // (the usual row set -- position and velocity, six rows -- as a compile-time constant: the index arithmetic of an item is five divisions
    // by na and per, ~150 instructions with run-time divisors)
    auto corrections = [&](const auto na_c) {
        const int na = na_c;
        const int per = 2 * na * na + na;
        for (int item = tid; item < W * per; item += KC_THREADS) {
            const int i = item / per, w = item - i * per;
            int r, j, kindI;
            if (w < na * na) { kindI = 0; r = misc[2 + w / na]; j = misc[2 + w % na]; if (j > r) continue; }
            else if (w < 2 * na * na) { kindI = 1; const int u = w - na * na; r = misc[2 + u / na]; j = misc[2 + u % na]; if (i + 1 >= W) continue; }
            else { kindI = 2; r = 0; j = misc[2 + w - 2 * na * na]; }
            double* dst = Blk + (size_t)i * KC_BLK + (kindI == 0 ? r : (kindI == 1 ? KC_NB + r : 30)) * KC_RS + j;
            double v = *dst;
            const int t1 = eoff[i + 1];
            for (int t = eoff[i]; t < t1; t += 4) {
                int base[4], ob[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int tt = t + q < t1 ? t + q : t1 - 1;
                    base[q] = esd[tt];
                    ob[q] = kindI == 1 ? eoth[tt] : (kindI == 2 ? elist[tt] : 0);
                }
                double xa[4], xb[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    xb[q] = Vs[base[q] + j];
                    xa[q] = kindI == 0 ? Vs[base[q] + r] : (kindI == 1 ? Vs[(ob[q] >= 0 ? ob[q] : base[q]) + r] : yd[ob[q]]);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool live = (t + q < t1) & !(kindI == 1 && ob[q] < 0);
                    v -= live ? xa[q] * xb[q] : 0.0;
                }
            }
            *dst = v;
        }
    };
    if (misc[1] == 6) corrections(std::integral_constant<int, 6>{}); else corrections(misc[1]);
    }
    }   // !fat_path
    if (tid < 4) s_prog[tid] = 0;
    GLIO_BLOCK_LDS_SYNC();
    AR_STAMP(47);
    // four fronts: the split and where its panels live (chain_f4_layout)
    const bool f4 = a.fronts4 != 0;
    const ChainSplit cs = chain_f4_split(W);
    ChainF4Mem f4m;
    {
        const ChainF4Layout f4l = chain_f4_layout(W, nd, n, mir);
        f4m.es0 = dds; f4m.es1 = reinterpret_cast<double*>(tr_lds + f4l.off_r1); f4m.k0 = f4l.k0;
        f4m.cs1a = f4m.es1 + (size_t)(cs.nB + cs.nC - f4l.k0) * KC_ES;      // C00 of front B
        f4m.cs3a = f4m.cs1a + KC_TILE;                                      // C00 of front C
    }
    // the chain from both ends
    const int mid = W / 2, nT = mid, nB = W - 1 - mid, Tn = nT > nB ? nT : nB;
    double av[KC_NB];
#pragma unroll
    for (int j = 0; j < KC_NB; ++j) av[j] = 0.0;
    if (lane < KC_NB || lane == 30) {
        const int row = lane < KC_NB ? lane : 30;
        if (wv == 0 && nT > 0) { for (int j = 0; j < KC_NB; ++j) av[j] = Blk[row * KC_RS + j]; }
        if (wv == 2 && nB > 0) { for (int j = 0; j < KC_NB; ++j) av[j] = Blk[(size_t)(W - 1) * KC_BLK + row * KC_RS + j]; }
    }
    bool bad = false;
    long long ph[5] = {0, 0, 0, 0, 0};
    // back substitution of the middle keyframe (the one block whose triangular solve is on the critical path)
    auto back = [&](const int i, const int nbr, double* zout = nullptr) { chain_back_solve<false>(Blk, i, nbr, zb, zout, lane); };

    // The two fronts run WITHOUT workgroup barriers between their steps (they meet only at the middle keyframe); each publishes
    // its progress in LDS, and the wavefronts that prepare the factored blocks for the back substitution (on the two SIMDs the
    // fronts do not issue on) follow it by polling.  (Workgroup-scope atomics on the __shared__ words: ds_write / ds_read.  A cast to
    // `volatile int*` drops the address space -- the accesses became FLAT, system scope, each followed by s_waitcnt vmcnt(0).)
    if (f4) {
        chain_f4_factor<false>(W, cs, Blk, CsT, CsB, f4m, zb, s_prog, lane, wv, bad, ph, a.dbg);
    } else if (wv == 0) {
        for (int it = 0; it < nT; ++it) {
            chain_step15<false>(it, it + 1, true, av, Blk, CsT, lane, bad, ph);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            if (lane == 0) __hip_atomic_store(&s_prog[0], it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        // the middle keyframe as soon as the other front has arrived (its last update sits in CsB): no workgroup barrier here, the wavefronts
        // that prepare the back substitution finish their last blocks while this step runs
        AR_STAMP(100);
        while (__hip_atomic_load(&s_prog[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < nB) __builtin_amdgcn_s_sleep(1);
        AR_STAMP(101);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        {
            const int row = lane < KC_NB ? lane : 30, crow = lane < KC_NB ? lane : 15;
            const int lim = lane == 30 ? KC_NB : (lane < KC_NB ? lane + 1 : 0);
            double b0[KC_NB], c0[KC_NB], c1[KC_NB];
#pragma unroll
            for (int j = 0; j < KC_NB; ++j) { b0[j] = Blk[(size_t)mid * KC_BLK + row * KC_RS + j]; c0[j] = CsT[crow * KC_RS + j]; c1[j] = CsB[crow * KC_RS + j]; }
#pragma unroll
            for (int j = 0; j < KC_NB; ++j) {
                double v = b0[j];
                if (nT > 0) v -= c0[j];
                if (nB > 0) v -= c1[j];
                av[j] = j < lim ? v : 0.0;
            }
        }
        chain_step15<false>(mid, mid, false, av, Blk, CsT, lane, bad);
        AR_STAMP(102);
        back(mid, -1);          // (garbage in, garbage out when a pivot broke down: nobody reads zb then)
        AR_STAMP(103);
    } else if (wv == 2) {
        for (int it = 0; it < nB; ++it) {
            const int i = W - 1 - it;
            chain_step15<true>(i, i - 1, true, av, Blk, CsB, lane, bad);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            if (lane == 0) __hip_atomic_store(&s_prog[1], it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else if (wv == 1) {
        for (int k = 0; k < nT; ++k) {
            while (__hip_atomic_load(&s_prog[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < k + 1) __builtin_amdgcn_s_sleep(2);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            chain_prepare_back(Blk + (size_t)k * KC_BLK, lane);
        }
    } else if (wv == 3) {
        for (int k = 0; k < nB; ++k) {
            while (__hip_atomic_load(&s_prog[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < k + 1) __builtin_amdgcn_s_sleep(2);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            chain_prepare_back(Blk + (size_t)(W - 1 - k) * KC_BLK, lane);
        }
    }
    GLIO_BLOCK_LDS_SYNC();
    AR_STAMP(48);
#ifdef GLIO_DEV_STAMPS
    if (tid == 0) for (int k = 0; k < 5; ++k) a.dbg[60 + k] = ph[k];
#endif
    if ((bad || a.force_fail) && lane == 0) misc[0] = 1;
    GLIO_BLOCK_LDS_SYNC();
    AR_STAMP(97);
    if (!misc[0]) {
        // z_i = w_i - M_i z_neighbour on the blocks chain_prepare_back transformed
        auto back_mv = [&](const int i, const int nbr) {
            const double* Bi = Blk + (size_t)i * KC_BLK;
            if (lane < KC_NB) {
                double mrow[KC_NB], zn[KC_NB];
#pragma unroll
                for (int k = 0; k < KC_NB; ++k) { mrow[k] = Bi[(KC_NB + lane) * KC_RS + k]; zn[k] = zb[15 * nbr + k]; }
                double s0 = Bi[30 * KC_RS + lane], s1 = 0, s2 = 0;
#pragma unroll
                for (int k = 0; k < KC_NB; k += 3) { s0 -= mrow[k] * zn[k]; s1 -= mrow[k + 1] * zn[k + 1]; s2 -= mrow[k + 2] * zn[k + 2]; }
                zb[15 * i + lane] = (s0 + s1) + s2;
            }
            GLIO_WAVE_LDS_SYNC();
        };
        if (f4) chain_f4_backsub<false>(W, cs, Blk, f4m, zb, lane, wv);
        else if (wv == 0) { for (int i = mid - 1; i >= 0; --i) back_mv(i, i + 1); }
        else if (wv == 1) { for (int i = mid + 1; i < W; ++i) back_mv(i, i - 1); }
        AR_STAMP(98);
        GLIO_BLOCK_LDS_SYNC();
        AR_STAMP(99);
        double bd2 = 0.0;
        for (int e = tid; e < nd; e += KC_THREADS) {
            const int2 sl = eps[e];
            double v = yd[e];
            if (sl.x >= 0) {
#pragma unroll
                for (int q = 0; q < 15; ++q) { v -= Vs[e * 30 + q] * zb[15 * sl.x + q]; v -= Vs[e * 30 + 15 + q] * zb[15 * sl.y + q]; }
            }
            v *= rd[e];
            a.z[e] = v;
            wd[e] = v;                  // (the fast tail reads the solution from LDS)
            if (!isfinite(v)) bd2 = 1.0;
        }
        for (int k = tid; k < 15 * W; k += KC_THREADS) { const double v = zb[k]; a.z[nd + k] = v; if (!isfinite(v)) bd2 = 1.0; }
        if (bd2 != 0.0) misc[0] = 1;
        GLIO_BLOCK_LDS_SYNC();
    }
    fast_tail = (a.fast & 1) && !misc[0];
    if (tid == 0) { *a.flag = misc[0] ? 1 : 2; }
    if (fast_tail) GLIO_BLOCK_LDS_SYNC(); else { __threadfence(); __syncthreads(); }      // (the generic tail reads t, z and the record back from global memory)
    AR_STAMP(49);
    }   // !dec.reuse
    if (fast_tail) {
        // ---- tr_factor_body (structured solve succeeded) + tr_dogleg_body on the LDS copies: the same arithmetic on the same numbers in the same
        // order (thread i takes entry i, the sums go through block_sum_n), without the ~8 global round trips the generic bodies make through
        // the work vectors.  The vectors and the status record are still WRITTEN to global memory (a rejected step re-enters through the
        // generic dogleg body in the next launch), just not read back here.
        double* red = CsT;                                     // 8 x 8 doubles
        double* sW = Blk;                                      // the step (the chain blocks are dead)
        const SolverStatus& s0 = s_full;
        double p = 0, q2 = 0, gg = 0, nn = 0, gd = 0;
        for (int i = tid; i < n; i += TR_THREADS) {
            const double gr = sGr[i];
            const double ui = sS[i] * gr / sDg[i];
            p += ui * sT[i]; q2 += gr * gr;
            const double yi = i < np15 ? zb[i] : wd[i - np15];
            const double gni = -sDg[i] * yi;
            V_Y(tr)[i] = yi; V_GN(tr)[i] = gni;
            gg += gr * gr; nn += gni * gni; gd += gr * gni;
        }
        { double v5[5] = {p, q2, gg, nn, gd}; block_sum_n<5, true>(v5, red); p = v5[0]; q2 = v5[1]; gg = v5[2]; nn = v5[3]; gd = v5[4]; }
        AR_STAMP(70);
        const double alpha = q2 / p;
        const double gnorm = sqrt(gg), gnn = sqrt(nn), radius = s0.radius;
        double ca, cb, snorm;       // step (D-space) = ca * grad + cb * gn
        if (tr.lm) { ca = 0.0; cb = 1.0; snorm = gnn; }
        else if (gnn <= radius) { ca = 0.0; cb = 1.0; snorm = gnn; }
        else if (gnorm * alpha >= radius) { ca = -(radius / gnorm); cb = 0.0; snorm = radius; }
        else {
            const double b_dot_a = -alpha * gd;
            const double a_sq = alpha * alpha * gg;
            const double b_minus_a_sq = nn - 2 * b_dot_a + a_sq;
            const double c = b_dot_a - a_sq;
            const double d = sqrt(c * c + b_minus_a_sq * (radius * radius - a_sq));
            const double beta = (c <= 0) ? (d - c) / b_minus_a_sq : (radius * radius - a_sq) / (d + c);
            ca = -alpha * (1.0 - beta); cb = beta; snorm = -1.0;
        }
        const double mu_u = s0.mu;                             // the factorisation succeeded with the record's mu: mu_used = mu
        double sn2 = 0, lin = 0, quad = 0;
        for (int i = tid; i < n; i += TR_THREADS) {
            const double yi = i < np15 ? zb[i] : wd[i - np15];
            const double gni = -sDg[i] * yi;
            const double sv = ca * sGr[i] + cb * gni;
            sn2 += sv * sv;
            const double step_s = sv / sDg[i];
            const double wi = sS[i] * step_s;
            V_W(tr)[i] = wi; sW[i] = wi;
            const double gs = sS[i] * sG[i];
            const double hs_step = ca * (sS[i] * sT[i]) - cb * (gs - mu_u * sDg[i] * sDg[i] * yi);
            lin += gs * step_s;
            quad += step_s * hs_step;
        }
        AR_STAMP(71);
        { double v3[3] = {sn2, lin, quad}; block_sum_n<3, true>(v3, red + 40); sn2 = v3[0]; lin = v3[1]; quad = v3[2]; }
        AR_STAMP(72);
        if (snorm < 0) snorm = sqrt(sn2);
        const double mcc = -(lin + 0.5 * quad);
        const bool valid = mcc > 0.0;                          // (lin_fail = 0: the linear solve succeeded)
        bool ended = false;
        SolverStatus sn = s0;                                  // registers; thread 0's copy is the one written back
        sn.alpha = alpha; sn.mu_used = mu_u; sn.lin_fail = 0;
        sn.dogleg_step_norm = snorm;
        if (!valid) {
            sn.invalid += 1;
            if (sn.invalid >= 5) { sn.done = 1; sn.termination = GLIO_TERM_FAILURE; ended = true; }
            if (tr.lm) { sn.radius /= sn.decrease_factor; sn.decrease_factor *= 2.0; }
            else sn.mu *= 10.0;
            sn.reuse = 0;
            sn.cand_pending = 0;
        } else {
            sn.invalid = 0;
            sn.model_cost_change = mcc;
            sn.cand_pending = 1;
        }
        if (ended) { __syncthreads(); finalize(tr, sn); AR_STAMP(50); AR_STAMP(51); return; }
        if (valid) {        // candidate = x (+) delta into the other buffer
            double* xn = sn.cur ? tr.x0 : tr.x1;
            for (int k = tid; k < 3 * W; k += TR_THREADS) { const int sl = k / 3, c = k % 3; xn[k] = sX[k] + sW[15 * sl + c]; }
            if (tid >= TR_THREADS - 64) for (int sl = tid - (TR_THREADS - 64); sl < W; sl += 64) {      // (the last wavefront: beside the others' sums, not after them)
                double q[4], qn[4];
                const double d[3] = {sW[15 * sl + 3], sW[15 * sl + 4], sW[15 * sl + 5]};
                for (int c = 0; c < 4; ++c) q[c] = sX[3 * W + 4 * sl + c];
                d_quat_plus(q, d, qn);
                for (int c = 0; c < 4; ++c) xn[3 * W + 4 * sl + c] = qn[c];
            }
            for (int k = tid; k < 9 * W; k += TR_THREADS) { const int sl = k / 9, c = k % 9; xn[7 * W + k] = sX[7 * W + k] + sW[15 * sl + 6 + c]; }
            for (int k = tid; k < tr.n_ddt; k += TR_THREADS) xn[16 * W + k] = sX[16 * W + k] + sW[15 * W + k];
        }
        AR_STAMP(73);
#ifdef GLIO_DEV_STAMPS
        if (tid == 0) a.dbg[121] = clock64();
#endif
        if (tid == 0) *tr.status = sn;
        AR_STAMP(50); AR_STAMP(51);
        return;
    }
    // ---- the linear solve's result (or the dense fallback) and the dogleg step, as k_tr_finish
    ChainBuilder cb;
    cb.G = &G; cb.T = &T; cb.cur = dec.cur;
    tr_factor_body(tr, cb);
    __syncthreads();
    AR_STAMP(50);
    tr_dogleg_body(tr);
    AR_STAMP(51);
}

size_t glio_tr_step_lds_bytes(int n) { return tr_step_lds_doubles(n) * sizeof(double); }

// The gather tables of k_chain_step, from the host's mirror of the factor graph (IMU edge slots, GNSS groups, prior index):
// per keyframe a ChainKf descriptor, then the prior index.  Rebuilt when a set_* call changed the structure; the copy is
// asynchronous on the context's stream, ahead of the kernels that read it.
void glio_chain_tabs_upload(glio_ctx* c) {
    const int W = c->W;
    // Two pinned copies used in turn, each with the event of its last copy: the tables are rebuilt once or twice per keyframe (every set_* call and the
    // marginalization change the structure), and waiting for the STREAM here meant waiting for whatever was queued on it (the window's searches, the
    // status upload of the solve) with the host and then the GPU idle.  The copy two uploads back has long left its buffer; the wait below is a formality.
    static_assert(sizeof(ChainKf) == 8 * sizeof(short), "ChainKf is eight shorts");
    const int half = c->chain_tabs_half & 1;
    c->chain_tabs_half ^= 1;
    if (!c->ev_tabs[half]) hipEventCreateWithFlags(&c->ev_tabs[half], hipEventDisableTiming);
    else hipEventSynchronize(c->ev_tabs[half]);
    short* hbuf = c->h_chain_tabs + (size_t)half * (8 + 15) * W;
    ChainKf* kd = reinterpret_cast<ChainKf*>(hbuf);
    short* pidx = hbuf + 8 * W;
    for (int i = 0; i < W; ++i) { ChainKf d; d.e0 = d.e1 = d.k0 = d.k1 = d.kp = -1; d.o0 = d.o1 = 0; d.pad_ = 0; kd[i] = d; }
    for (int k = 0; k < c->n_imu; ++k) {
        const int si = c->h_imu_slot[k];
        if (si < 0 || si + 1 >= W) continue;
        kd[si].e0 = (short)k; kd[si + 1].e1 = (short)k;
    }
    for (int k = 0; k < c->n_groups; ++k) {          // ascending group index: the order k_assemble adds the groups in
        const int sa = c->h_groups[k].slot_i, sb = c->h_groups[k].slot_j;
        for (int side = 0; side < 2; ++side) {
            const int sl = side == 0 ? sa : sb;
            if (sl < 0 || sl >= W) continue;
            if (kd[sl].k0 < 0) { kd[sl].k0 = (short)k; kd[sl].o0 = side == 0; }
            else if (kd[sl].k1 < 0) { kd[sl].k1 = (short)k; kd[sl].o1 = side == 0; }
        }
        if (sb == sa + 1 && sa >= 0 && sa < W) kd[sa].kp = (short)k;
    }
    for (int k = 0; k < 15 * W; ++k) pidx[k] = (short)c->h_prior_index[k];
    // a stream of keyframes rebuilds the SAME tables call after call (IMU edge on every pair, the GNSS groups of the same pairs, the prior on the same
    // blocks): then the device copy and the zeroed slices already are what this upload would leave
    const short* prev = c->h_chain_tabs + (size_t)(half ^ 1) * (8 + 15) * W;
    if (c->chain_tabs_on_device && memcmp(prev, hbuf, (size_t)(8 + 15) * W * sizeof(short)) == 0) { c->chain_tabs_dirty = 0; return; }
    c->chain_tabs_on_device = 1;
    hipMemcpyAsync(c->d_chain_tabs, hbuf, (size_t)(8 + 15) * W * sizeof(short), hipMemcpyHostToDevice, c->stream);
    hipEventRecord(c->ev_tabs[half], c->stream);
    // the structure changed: slices of sources that no longer exist must read as zero
    hipMemsetAsync(c->d_chain_src, 0, 2 * (size_t)W * GLIO_CS_SOURCES * GLIO_CS_STRIDE * 8, c->stream);
    c->chain_tabs_dirty = 0;
}

// which factorisation the trust-region step of this context takes for a state with n_ddt clock-drift unknowns:
int glio_chain_kind(const glio_ctx* c, int n_ddt);
// 2 = keyframe chain (k_chain_step, no dense H), 1 = arrow, 0 = dense
int glio_solver_path(const glio_ctx* c, int n_ddt) {
    const int n = 15 * c->W + n_ddt;
    const int np = 6 * c->W, K = n - np;
    const size_t lds_fwd = arrow_forward_lds_doubles(c->W, n_ddt) * 8;
    const size_t slv_tail = ((size_t)(TR_NB + 1) * TR_PS + np + (np & 1) + 48 + arrow_solve_extra_doubles(c->W, K)) * 8;
    const size_t lds_pk = pk_doubles(np) * 8 + slv_tail;
    const bool lds_chol = lds_pk <= 160 * 1024;
    const size_t lds_slv = lds_chol ? lds_pk : glio_tr_step_lds_bytes(np) + arrow_solve_extra_doubles(c->W, K) * 8;
    const bool arrow = c->arrow.mode >= 1 && c->arrow.gnss_ok && c->arrow.prior_ok && c->arrow.max_epoch < n_ddt && lds_fwd <= 160 * 1024 && lds_slv <= 160 * 1024;
    return glio_chain_kind(c, n_ddt) ? 2 : (arrow ? 1 : 0);
}
// which chain kernel: 0 none (graph not a chain / nothing fits), 1 = k_chain_step (one launch, blocks in LDS), 2 = the legacy sequence assemble +
// k_chain_solve<false> + k_tr_finish (debug mode 3), 3 = the same sequence with k_chain_solve<true>: the blocks in global memory, for windows of
// more keyframes than the LDS holds (C5: 50 keyframes, where the arrow factorisation costs ~420 us per step)
int glio_chain_kind(const glio_ctx* c, int n_ddt) {
    const int n = 15 * c->W + n_ddt;
    if (!(c->arrow.mode >= 1 && c->arrow.gnss_chain && c->arrow.prior_chain && c->arrow.max_epoch < n_ddt)) return 0;
    if (c->arrow.mode == 4) return 0;                          // (debug: the arrow factorisation although the graph is a chain)
    if (c->arrow.mode == 3) return chain_lds_doubles(c->W, n_ddt) * 8 + 8 * 1024 <= 158 * 1024 ? 2 : 0;
    if (chain_step_lds_bytes(c->W, n_ddt, n, false) + 2 * 1024 <= 158 * 1024) return 1;
    if (c->W >= KC_F4_MIN_W && chain_lds_doubles_g(c->W, n_ddt) * 8 + 8 * 1024 <= 158 * 1024) return 3;
    return 0;
}
// the linearisation must also build the dense H (k_assemble) unless the step is k_chain_step
int glio_solver_needs_dense_H(const glio_ctx* c, int n_ddt) { return glio_chain_kind(c, n_ddt) != 1; }

// GLIO_CHAIN_FAST (bit 0: tail, bit 1: front of k_chain_step from LDS; default both).  0 = the generic bodies: the A/B switch of
// scripts/chain_step_time.py and of test_chain_step_fast_paths_are_bit_identical.
static int g_chain_fast = -1;
extern "C" int glio_debug_chain_fast(int mask) { const int old = g_chain_fast; g_chain_fast = mask; return old; }
static int chain_fast_mask() {
    if (g_chain_fast < 0) { const char* e = getenv("GLIO_CHAIN_FAST"); g_chain_fast = e ? atoi(e) & 3 : 3; }
    return g_chain_fast;
}

// test hook (CPU-callable: host arithmetic only): where the four-front panels of k_chain_step would live for a window of W keyframes and nd clock-drift
// epochs.  out = {regular dynamic LDS bytes, off_dds, bytes available at off_dds (the clock-drift copy), k0 (E slots placed there), off_r1, total bytes,
// E slots in all, 1 if the launch would take four fronts}
extern "C" int glio_debug_chain_f4_layout(int W, int nd, int mirrors, long long* out) {
    if (W < 2 || nd < 0 || !out) return -1;
    const int n = 15 * W + nd;
    const bool mir = mirrors != 0;
    const ChainF4Layout L = chain_f4_layout(W, nd, n, mir);
    const ChainSplit c = chain_f4_split(W);
    const size_t regular = chain_step_lds_bytes(W, nd, n, mir);
    const size_t with4 = L.total > regular ? L.total : regular;
    out[0] = (long long)regular; out[1] = (long long)L.off_dds; out[2] = (long long)nd * 15 * 8; out[3] = L.k0; out[4] = (long long)L.off_r1; out[5] = (long long)L.total;
    out[6] = c.nB + c.nC; out[7] = (W >= KC_F4_MIN_W && with4 + 256 <= 158 * 1024) ? 1 : 0;
    out[8] = c.s; out[9] = c.mL; out[10] = c.mR; out[11] = c.nA; out[12] = c.nB; out[13] = c.nC; out[14] = c.nD;
    out[15] = KC_ES * 8; out[16] = KC_TILE * 8;
    return 0;
}

// GLIO_CHAIN_FRONTS = 2: k_chain_step keeps the two-front elimination for every window (A/B switch and cross-check of the four-front order)
static int g_chain_fronts = -1;
extern "C" int glio_debug_chain_fronts(int fronts) { const int old = g_chain_fronts; g_chain_fronts = fronts; return old; }
static int chain_fronts_mode() {
    if (g_chain_fronts < 0) { const char* e = getenv("GLIO_CHAIN_FRONTS"); g_chain_fronts = e ? atoi(e) : 4; }
    return g_chain_fronts;
}

void glio_launch_tr_step(glio_ctx* c, int n_ddt) {
    TrArgs a;
    a.W = c->W; a.n = 15 * c->W + n_ddt; a.n_ddt = n_ddt; a.max_iterations = c->opts.max_iterations;
    a.stop_word = c->opts.max_solver_time_s > 0.0 ? c->d_progress + 2 : nullptr;
    a.min_relative_decrease = c->opts.min_relative_decrease; a.function_tolerance = c->opts.function_tolerance;
    a.gradient_tolerance = c->opts.gradient_tolerance; a.parameter_tolerance = c->opts.parameter_tolerance;
    a.min_radius = c->opts.min_trust_region_radius; a.initial_radius = c->opts.initial_trust_region_radius;
    a.max_radius = c->opts.max_trust_region_radius; a.lm = c->opts.trust_region_strategy == GLIO_STRATEGY_LM;
    a.jacobi_scaling = c->opts.jacobi_scaling;
    a.x0 = c->d_x[0]; a.x1 = c->d_x[1]; a.xout = c->d_xout;
    a.H0 = c->d_H[0]; a.H1 = c->d_H[1]; a.g0 = c->d_g[0]; a.g1 = c->d_g[1]; a.c0 = c->d_cost[0]; a.c1 = c->d_cost[1];
    a.L = c->d_L; a.vec = c->d_vec; a.vstride = c->vstride;
    a.status = c->d_status; a.progress = c->d_progress;
    a.status_host = reinterpret_cast<SolverStatus*>(c->d_result); a.xout_host = reinterpret_cast<double*>(c->d_result + 512);
    // structured factorisation when the factor graph is a chain (IMU / Doppler edges between neighbours only, at most
    // one speed-bias block in the prior) and its workspaces fit the LDS; the dense kernel stays as the fallback
    const int np = 6 * c->W, K = a.n - np;
    const size_t lds_fwd = arrow_forward_lds_doubles(c->W, n_ddt) * 8;
    const size_t slv_tail = ((size_t)(TR_NB + 1) * TR_PS + np + (np & 1) + 48 + arrow_solve_extra_doubles(c->W, K)) * 8;
    const size_t lds_pk = pk_doubles(np) * 8 + slv_tail;
    const bool lds_chol = lds_pk <= 160 * 1024;
    const size_t lds_slv = lds_chol ? lds_pk : glio_tr_step_lds_bytes(np) + arrow_solve_extra_doubles(c->W, K) * 8;
    const int path = glio_solver_path(c, n_ddt);
    const bool chain = path == 2, arrow = path >= 1;       // (arrow is only consulted when !chain)
    const int ckind = glio_chain_kind(c, n_ddt);
    const bool legacy_chain = chain && ckind >= 2;   // assemble + k_chain_solve + k_tr_finish (cross-check of k_chain_step; with global blocks: long windows)
    const size_t lds_chain = ckind == 3 ? chain_lds_doubles_g(c->W, n_ddt) * 8 : chain_lds_doubles(c->W, n_ddt) * 8;
    a.hd0 = nullptr; a.hd1 = nullptr;
    a.perm_mode = chain ? 1 : 0;
    c->arrow.last_path = chain ? 2 : (arrow ? 1 : 0);
    a.arrow_flag = (arrow || chain) ? c->arrow.d_flag : nullptr; a.arrow_z = c->arrow.d_z;
    a.fused_chain = chain ? 1 : 0;
    if (!chain) {
        hipLaunchKernelGGL(k_tr_prepare, dim3(1), dim3(TR_THREADS), 0, c->stream, a);
        hipLaunchKernelGGL(k_tr_scale, dim3((a.n + 1 + 3) / 4), dim3(256), 0, c->stream, a);
    }
    if (chain) {
        ChainArgs r;
        r.W = c->W; r.n = a.n; r.nd = n_ddt; r.ep_slots = c->arrow.d_ep_slots; r.ep_off = c->arrow.d_ep_off; r.ep_list = c->arrow.d_ep_list;
        r.z = c->arrow.d_z; r.flag = c->arrow.d_flag; r.status = c->d_status; r.dbg = c->arrow.d_dbg; r.force_fail = c->arrow.mode == 2; r.fast = chain_fast_mask(); r.blk = nullptr; r.helpers = 0; r.hseq = 0; r.hsum = nullptr; r.hdone = nullptr; r.hpolls = 0; r.fat = 0; r.fat_blk = nullptr; r.fat_ep = nullptr;
        if (chain_step_lds_bytes(c->W, n_ddt, a.n, true) + 2 * 1024 > 158 * 1024) r.fast = 0;      // no room for the LDS mirrors: generic bodies
        size_t lds_step = chain_step_lds_bytes(c->W, n_ddt, a.n, r.fast != 0);
        {   // separator + four fronts when the window is long enough for it to pay and its panels fit (chain_f4_layout)
            const ChainF4Layout L = chain_f4_layout(c->W, n_ddt, a.n, r.fast != 0);
            const size_t with4 = L.total > lds_step ? L.total : lds_step;
            // (158 KB is what hipFuncSetAttribute grants this kernel as dynamic LDS: 160 KB less 2 KB for its 1.4 KB of static LDS)
            r.fronts4 = (c->W >= KC_F4_MIN_W && chain_fronts_mode() != 2 && with4 + 256 <= 158 * 1024) ? 1 : 0;
            if (r.fronts4) lds_step = with4;
            c->arrow.last_fronts = r.fronts4 ? 4 : 2;
        }
        if (!legacy_chain) {
            GatherArgs G;
            G.lidar_partials = c->d_lidar_partials; G.lidar_pstride = glio_partials_stride(c); G.lidar_nb = c->last_k3_nb;
            G.imu_blocks = c->d_imu_blocks; G.gnss_blocks = c->d_gnss_blocks; G.ddt_blocks = c->d_ddt_blocks;
            G.gnss_stride = c->W * c->W; G.ddt_stride = c->n_ddt_max > 0 ? c->n_ddt_max : 1;
            G.n_imu = c->n_imu; G.n_groups = c->n_groups; G.has_prior = c->prior_n > 0; G.np = c->prior_n;
            G.pH = c->d_prior_H; G.pg = c->d_prior_g; G.pcost = c->d_prior_cost;
            G.tabs = c->d_chain_tabs; G.chain_src = c->d_chain_src;
            G.hd0 = c->d_hdiag[0]; G.hd1 = c->d_hdiag[1]; G.g0 = c->d_g[0]; G.g1 = c->d_g[1]; G.c0 = c->d_cost[0]; G.c1 = c->d_cost[1];
            a.hd0 = c->d_hdiag[0]; a.hd1 = c->d_hdiag[1];
            a.fused_chain = 2;
            // helper workgroups (one per keyframe) pre-sum the candidate's block entries: GLIO_CHAIN_HELPERS=0 switches them off (A/B)
            static const bool helpers_off = getenv("GLIO_CHAIN_HELPERS") && atoi(getenv("GLIO_CHAIN_HELPERS")) == 0;
            // Workgroup 0 WAITS for the helpers' completion words inside the launch, and every workgroup of this launch reserves the whole dynamic LDS
            // grant, i.e. a CU of its own: the hand-over needs 1 + W CUs that the launch can actually get.  HIP promises no forward progress between
            // workgroups, so the helpers are used only when the device has room to spare (>= 2 (1 + W) CUs) AND the wait is bounded (hpolls below): a
            // deployment that masks CUs or partitions the device cannot hang the step, it only loses the helpers' 4 us (GLIO_CHAIN_HELPERS=0 skips the
            // attempt).  INTEGRATION.md section 5.
            const bool room = c->n_cu >= 2 * (1 + c->W);          // (the context's own device: queried at glio_create, not a process-wide value)
            r.helpers = (helpers_off || !room) ? 0 : c->W; r.hsum = c->arrow.d_chain_sum; r.hdone = c->arrow.d_chain_done;
            r.hseq = c->arrow.chain_seq; c->arrow.chain_seq = c->arrow.chain_seq % (1 << 28) + 1;
            // the wait for the helpers is bounded (a poll is a device-scope load + s_sleep: ~0.5-1 us; the helpers report ~5 us into the launch): ~0.2 ms at
            // most, then the step sums the blocks itself.  GLIO_CHAIN_HELPER_POLLS=0 gives them up at once (test: same bits out)
            static const int hpolls = getenv("GLIO_CHAIN_HELPER_POLLS") ? atoi(getenv("GLIO_CHAIN_HELPER_POLLS")) : 256;
            r.hpolls = hpolls;
            // fat helpers (GLIO_CHAIN_FAT=0: the helpers only sum): the speculative build of every keyframe's block, its rows of t and its epochs' columns
            static const bool fat_on = !(getenv("GLIO_CHAIN_FAT") && atoi(getenv("GLIO_CHAIN_FAT")) == 0);
            r.fat = (fat_on && r.helpers && chain_fat_helper_lds_bytes(c->W, n_ddt) <= lds_step) ? 1 : 0; r.fat_blk = c->arrow.d_blk; r.fat_ep = c->arrow.d_fat_ep;
            hipLaunchKernelGGL(k_chain_step, dim3(1 + r.helpers), dim3(KC_THREADS), lds_step, c->stream, r, a, G);
            return;                                   // the one launch is the whole step
        }
        if (ckind == 3) {
            r.blk = c->arrow.d_blk; r.fronts4 = chain_fronts_mode() != 2 ? 1 : 0;
            c->arrow.last_fronts = r.fronts4 ? 4 : 2;
            hipLaunchKernelGGL(k_chain_solve<true>, dim3(1), dim3(KC_THREADS), lds_chain, c->stream, r, a);
        } else hipLaunchKernelGGL(k_chain_solve<false>, dim3(1), dim3(KC_THREADS), lds_chain, c->stream, r, a);
    } else if (arrow) {
        ArrowArgs r;
        r.W = c->W; r.n = a.n; r.nd = n_ddt; r.np = np; r.K = K; r.ldY = np + 2;
        r.A = c->d_L; r.ep_slots = c->arrow.d_ep_slots; r.ep_off = c->arrow.d_ep_off; r.ep_list = c->arrow.d_ep_list;
        r.Y = c->arrow.d_Y; r.Lblk = c->arrow.d_Lblk; r.Sp = c->arrow.d_Sp; r.z = c->arrow.d_z; r.flag = c->arrow.d_flag;
        r.status = c->d_status; r.dbg = c->arrow.d_dbg; r.lds_chol = lds_chol ? 1 : 0;
        const int T = (np + 1 + 15) / 16;
        hipLaunchKernelGGL(k_arrow_forward, dim3(T), dim3(256), lds_fwd, c->stream, r);
        hipLaunchKernelGGL(k_arrow_schur, dim3(T * T), dim3(256), 0, c->stream, r);
        hipLaunchKernelGGL(k_arrow_solve, dim3(1), dim3(TR_THREADS), lds_slv, c->stream, r);
    }
    hipLaunchKernelGGL(k_tr_finish, dim3(1), dim3(TR_THREADS), glio_tr_step_lds_bytes(a.n), c->stream, a);
}

// ------------------------------------------------------------------------------------------------
// Marginalization (MarginalizationInfo::Marginalize, reference GLIO/src/MarginalizationFactor.cpp:128-202):
// A, b over [dropped m = 15 | kept n] -> Amm^+ by a 15x15 symmetric eigen-decomposition with the reference's
// eps = 1e-8 -> Schur complement S = Arr - Arm Amm^+ Amr, bs = br - Arm Amm^+ bm -> S = L L^T with the blocked
// MFMA Cholesky; J0 = L^T, r0 = L^-1 bs.  (The reference takes J0 = sqrt(Lambda) V^T from a second
// eigen-decomposition; any J0 with J0^T J0 = S and J0^T r0 = bs is the same prior for its only consumer,
// MarginalizationFactor::Evaluate.  A rank-deficient S -- where the reference would truncate -- is reported.)
// ------------------------------------------------------------------------------------------------
// Symmetric eigen-decomposition of a 16 x 16 matrix (15 x 15 padded with a zero row/column) by ONE wavefront:
// parallel cyclic Jacobi with the round-robin ordering -- every round applies 8 disjoint rotations at once,
// A' = J^T A J and V' = V J, each lane producing 4 entries of A' and of V' from the previous buffers.
// buf: 2 x (256 A + 256 V) doubles + 48 doubles of per-index rotation data.  Result: eigenvalues on the diagonal
// of the returned A buffer, eigenvectors in the columns of the returned V buffer.
template <bool G = false>            // G: the buffer lives in GLOBAL memory (k_marg_inv keeps its LDS under 6 KB; this path runs once per stream)
__device__ __forceinline__ int jacobi16_wave(double* buf, const int lane) {
    double* coef = buf + 1024;                     // alpha[16], beta[16]
    int* partner = reinterpret_cast<int*>(coef + 32);
    int cur = 0;
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double* A0 = buf + cur * 512;
        double off = 0, dg = 0;
        for (int e = lane; e < 256; e += 64) { const int i = e >> 4, j = e & 15; const double v = A0[e]; if (i == j) dg += v * v; else off += v * v; }
        off = wave_sum(off); dg = wave_sum(dg);
        if (off <= 1e-30 * (dg + 1e-300)) break;
        for (int r = 0; r < 15; ++r) {
            double* A = buf + cur * 512; double* V = A + 256;
            double* An = buf + (cur ^ 1) * 512; double* Vn = An + 256;
            if (lane < 8) {
                const int p0 = lane == 0 ? 15 : (r + lane) % 15, q0 = lane == 0 ? r : (r - lane + 15) % 15;
                const int p = p0 < q0 ? p0 : q0, q = p0 < q0 ? q0 : p0;
                const double apq = A[p * 16 + q];
                double c = 1.0, sn = 0.0;
                if (apq != 0.0) {
                    const double theta = (A[q * 16 + q] - A[p * 16 + p]) / (2.0 * apq);
                    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    c = 1.0 / sqrt(t * t + 1.0); sn = t * c;
                }
                coef[p] = c; coef[16 + p] = -sn; partner[p] = q;
                coef[q] = c; coef[16 + q] = sn; partner[q] = p;
            }
            chain_wave_sync<G>();
            for (int e = lane; e < 256; e += 64) {
                const int i = e >> 4, j = e & 15, pi = partner[i], pj = partner[j];
                const double ai = coef[i], bi = coef[16 + i], aj = coef[j], bj = coef[16 + j];
                An[e] = ai * (aj * A[i * 16 + j] + bj * A[i * 16 + pj]) + bi * (aj * A[pi * 16 + j] + bj * A[pi * 16 + pj]);
                Vn[e] = aj * V[i * 16 + j] + bj * V[i * 16 + pj];
            }
            chain_wave_sync<G>();
            cur ^= 1;
        }
    }
    return cur;
}

// A: pos x pos row-major (pos = 15 + n), b: pos.  Out: J0 (n x n row-major), r0 (n), *ok.
__global__ __launch_bounds__(TR_THREADS) void k_marg_schur(const double* A, const double* b, const int n, double* Lwork, double* Twork,
                                                           double* J0, double* r0, int* ok) {
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, m = 15, pos = m + n;
    double* Bp = reinterpret_cast<double*>(tr_lds);
    double* part = Bp + TR_NB * bp_stride(n);
    double* sD = part + 16 * 256;
    double* ylds = sD + (TR_NB + 1) * TR_PS;
    double* red = ylds + n + (n & 1);
    int* flag = reinterpret_cast<int*>(red + 32);
    double* ebuf = part;                       // free until the factorisation starts: 2 x (A, V) 16 x 16 + rotation data
    double* Ainv = part + 1100;
    int& ecur = flag[2];
    if (tid < 256) {
        const int i = tid >> 4, j = tid & 15;
        ebuf[tid] = (i < 15 && j < 15) ? 0.5 * (A[(size_t)i * pos + j] + A[(size_t)j * pos + i]) : 0.0;
        ebuf[256 + tid] = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();
    // Amm^+ : the reference inverts the eigenvalues above 1e-8 and drops the rest (MarginalizationFactor.cpp:175-182).  When NO
    // eigenvalue is dropped that is the plain inverse, and lambda_min = 1 / |Amm^-1|_2 >= 1 / |Amm^-1|_F can be certified from the
    // inverse itself: one wavefront factors Amm = L L^T in registers and inverts L (3 us); if the factorisation is clean and
    // 1 / |Amm^-1|_F > 1e-7 the result stands, otherwise (the first window of a stream is rank deficient by 3) the
    // eigen-decomposition (16 x 16 parallel Jacobi, ~100 us) runs as before.
    int& fast = flag[4];
    double* Linv = ebuf + 512;                     // 16 x 16, free until the Jacobi sweeps start
    if (tid < 64) {
        const int lane = tid;
        double a[16], x[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j] = lane < 15 ? (j < 15 ? ebuf[lane * 16 + j] : 0.0) : ((lane == 15 && j == 15) ? 1.0 : 0.0);
        bool bad = false;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            double djj = readlane_d(a[j], j);
            if (!(djj > 0.0) || !isfinite(djj)) { bad = true; djj = 1.0; }
            const double rd = 1.0 / sqrt(djj);
            const double lij = (lane == j) ? djj * rd : a[j] * rd;
            a[j] = lij;
#pragma unroll
            for (int c2 = j + 1; c2 < 16; ++c2) a[c2] -= lij * readlane_d(lij, c2);
        }
        // lane c: column c of L^-1 (forward substitution of e_c); L[i][k] sits in lane i, register k
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double sacc = (i == lane) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < i; ++k) sacc -= readlane_d(a[k], i) * x[k];
            x[i] = sacc / readlane_d(a[i], i);
        }
        if (lane < 16) {
#pragma unroll
            for (int k = 0; k < 16; ++k) Linv[k * 16 + lane] = x[k];
        }
        if (lane == 0) fast = bad ? 0 : 1;
    }
    __syncthreads();
    if (fast) {
        if (tid < 225) {
            const int i = tid / 15, j = tid % 15;
            double sacc = 0;
            for (int k = (i > j ? i : j); k < 15; ++k) sacc += Linv[k * 16 + i] * Linv[k * 16 + j];
            Ainv[tid] = sacc;
        }
        __syncthreads();
        if (tid == 0) {
            double f2 = 0;
            for (int k = 0; k < 225; ++k) f2 += Ainv[k] * Ainv[k];
            if (!(f2 > 0.0) || !isfinite(f2) || !(1.0 / sqrt(f2) > 1e-7)) fast = 0;
        }
        __syncthreads();
    }
    if (!fast) {
        if (tid < 64) { const int cur = jacobi16_wave(ebuf, tid); if (tid == 0) ecur = cur; }
        __syncthreads();
        if (tid < 225) {
            const double* Am = ebuf + ecur * 512; const double* Vm = Am + 256;
            const int i = tid / 15, j = tid % 15;
            double sacc = 0;
            for (int k = 0; k < 16; ++k) { const double w = Am[k * 17]; sacc += Vm[i * 16 + k] * (w > 1e-8 ? 1.0 / w : 0.0) * Vm[j * 16 + k]; }
            Ainv[tid] = sacc;
        }
        __syncthreads();
    }
    for (int e = tid; e < n * 15; e += TR_THREADS) {          // T = Arm Amm^+
        const int i = e / 15, j = e % 15;
        double sacc = 0;
        for (int k = 0; k < 15; ++k) sacc += A[(size_t)(m + i) * pos + k] * Ainv[k * 15 + j];
        Twork[e] = sacc;
    }
    __syncthreads();
    for (int i = tid >> 6; i <= n; i += TR_WAVES) {            // S (lower) and the carried right-hand side
        for (int j = tid & 63; j < n; j += 64) {
            if (i < n) {
                if (j > i) continue;
                double sacc = A[(size_t)(m + i) * pos + m + j];
                for (int k = 0; k < 15; ++k) sacc -= Twork[i * 15 + k] * A[(size_t)k * pos + m + j];
                Lwork[(size_t)i * n + j] = sacc;
            } else {
                double sacc = b[m + j];
                for (int k = 0; k < 15; ++k) sacc -= Twork[j * 15 + k] * b[k];
                Lwork[(size_t)n * n + j] = sacc;
            }
        }
    }
    __syncthreads();
    for (int j = tid; j < n; j += TR_THREADS) ylds[j] = fmax(1e-8, 1e-9 * Lwork[(size_t)j * n + j]);
    __syncthreads();
    // The Schur complement of block-diagonal pieces (LiDAR blocks of the dropped keyframe, its IMU edge, a prior that is block
    // diagonal by keyframe -- what this marginalization itself produces, quirk Q7) is block diagonal in the kept order
    // [T1 Q1 SB1 (15) | T2 Q2 (6) | T3 Q3 (6) | ...], with EXACT zeros between the blocks (sums of zeros).  Then its root is the
    // roots of the blocks: one wavefront per block, the same register steps and the same null-pivot rule as the dense routine,
    // instead of eight 16-column panels over the whole 123 x 123 matrix.  Any non-zero outside the blocks: dense routine.
    int& bdiag = flag[3];
    if (tid == 0) bdiag = (n >= 15 && (n - 15) % 6 == 0) ? 1 : 0;
    __syncthreads();
    {
        int viol = 0;
        for (int e = tid; e < n * n; e += TR_THREADS) {
            const int i = e / n, j = e - n * i;
            if (j >= i) continue;
            const int bi = i < 15 ? 0 : 1 + (i - 15) / 6, bj = j < 15 ? 0 : 1 + (j - 15) / 6;
            if (bi != bj && Lwork[e] != 0.0) viol = 1;
        }
        if (viol) bdiag = 0;
    }
    __syncthreads();
    bool good;
    if (bdiag) {
        const int lane = tid & 63, wv = tid >> 6, nblk = 1 + (n - 15) / 6;
        if (tid == 0) *flag = 0;
        __syncthreads();
        for (int bI = wv; bI < nblk; bI += TR_WAVES) {
            const int o = bI == 0 ? 0 : 15 + 6 * (bI - 1), bs = bI == 0 ? 15 : 6;
            const bool isrow = lane < bs, isrhs = lane == TR_NB;
            double a[TR_NB];
#pragma unroll
            for (int j = 0; j < TR_NB; ++j) {
                double v = (lane < TR_NB && lane == j) ? 1.0 : 0.0;                       // identity padding of the unused rows / columns
                if (isrow && j < bs) v = j <= lane ? Lwork[(size_t)(o + lane) * n + o + j] : 0.0;
                if (isrhs && j < bs) v = Lwork[(size_t)n * n + o + j];
                a[j] = v;
            }
            const double tolv = isrow ? ylds[o + lane] : 0.0;
            bool bad = false;
#pragma unroll
            for (int j = 0; j < TR_NB; ++j) {
                double djj = readlane_d(a[j], j);
                const bool null_pivot = j < bs && djj <= readlane_d(tolv, j);
                if (!null_pivot && (!(djj > 0.0) || !isfinite(djj))) { bad = true; djj = 1.0; }
                const double rd = null_pivot ? 0.0 : rsqrt(djj);
                const double lij = (lane == j) ? djj * rd : a[j] * rd;
                a[j] = lij;
#pragma unroll
                for (int c2 = j + 1; c2 < TR_NB; ++c2) a[c2] -= lij * readlane_d(lij, c2);
            }
            if (isrow) {
#pragma unroll
                for (int j = 0; j < TR_NB; ++j) if (j < bs && j <= lane) Lwork[(size_t)(o + lane) * n + o + j] = a[j];
            } else if (isrhs) {
#pragma unroll
                for (int j = 0; j < TR_NB; ++j) if (j < bs) Lwork[(size_t)n * n + o + j] = a[j];
            }
            if (bad && lane == 0) *flag = 1 + o;
        }
        __threadfence_block();
        __syncthreads();
        good = *flag == 0;
    } else {
        good = chol_left_looking<true>(Lwork, n, Bp, part, sD, flag, 0, ylds);
    }
    if (good) {
        for (int i = tid >> 6; i < n; i += TR_WAVES)
            for (int j = tid & 63; j < n; j += 64) J0[(size_t)i * n + j] = (j >= i) ? Lwork[(size_t)j * n + i] : 0.0;
        for (int j = tid; j < n; j += TR_THREADS) r0[j] = Lwork[(size_t)n * n + j];
    }
    if (tid == 0) *ok = good ? 1 : 0;
}

// ---- the same in three launches (round 6).  k_marg_schur runs everything in ONE workgroup over matrices that live in global memory: the n x n Schur
// complement alone is n^2 / 2 entries of 15 dependent global reads each for 512 threads -- 90 us for n = 123 where the arithmetic is a few microseconds.
// Split: (1) k_marg_inv, one workgroup: Amm^+ (fast inverse or the Jacobi eigen-decomposition) -> 225 doubles in global memory;
//        (2) k_marg_rows, n + 1 workgroups: row i of T = Arm Amm^+ in LDS, row i of S (and the carried right-hand side as row n), the null-pivot tolerance
//            of its diagonal entry, and a flag when a non-zero lies outside the block-diagonal pattern -- the same sums in the same order as above;
//        (3) k_marg_root, one workgroup: the root of S (per diagonal block by one wavefront each, or the dense blocked Cholesky), J0, r0.
// Same bits out (tests/test_hip_marg.py runs both forms).
__global__ __launch_bounds__(64) void k_marg_inv(const double* A, const int n, double* Ainv_out, int* viol_out, double* jbuf) {
    __builtin_amdgcn_s_setprio(3);      // (a short latency-bound kernel that shares its compute unit with the batch association's wide launches: its wavefronts issue first)
    // ONE wavefront and 6 KB of static LDS: in a keyframe call this kernel is launched while the batch association's searches hold every compute unit's LDS
    // but a few kilobytes -- with 16 KB of its own it waited for them to END (0.22 ms in profiles/r06_stream_cpp_timeline.txt).  The Jacobi eigen-decomposition
    // (rank-deficient Amm: the first window of a stream) works in global memory instead (jbuf: 1100 doubles).
    __shared__ double sA[256], sLinv[256], sAinv[232];
    __shared__ int s_fast;
    const int lane = threadIdx.x, m = 15, pos = m + n;
    if (lane == 0) *viol_out = 0;
    for (int e = lane; e < 256; e += 64) {
        const int i = e >> 4, j = e & 15;
        sA[e] = (i < 15 && j < 15) ? 0.5 * (A[(size_t)i * pos + j] + A[(size_t)j * pos + i]) : 0.0;
    }
    GLIO_WAVE_LDS_SYNC();
    {
        double a[16], x[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j] = lane < 15 ? (j < 15 ? sA[lane * 16 + j] : 0.0) : ((lane == 15 && j == 15) ? 1.0 : 0.0);
        bool bad = false;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            double djj = readlane_d(a[j], j);
            if (!(djj > 0.0) || !isfinite(djj)) { bad = true; djj = 1.0; }
            const double rd = 1.0 / sqrt(djj);
            const double lij = (lane == j) ? djj * rd : a[j] * rd;
            a[j] = lij;
#pragma unroll
            for (int c2 = j + 1; c2 < 16; ++c2) a[c2] -= lij * readlane_d(lij, c2);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double sacc = (i == lane) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < i; ++k) sacc -= readlane_d(a[k], i) * x[k];
            x[i] = sacc / readlane_d(a[i], i);
        }
        if (lane < 16) {
#pragma unroll
            for (int k = 0; k < 16; ++k) sLinv[k * 16 + lane] = x[k];
        }
        if (lane == 0) s_fast = bad ? 0 : 1;
    }
    GLIO_WAVE_LDS_SYNC();
    if (s_fast) {
        for (int t = lane; t < 225; t += 64) {
            const int i = t / 15, j = t % 15;
            double sacc = 0;
            for (int k = (i > j ? i : j); k < 15; ++k) sacc += sLinv[k * 16 + i] * sLinv[k * 16 + j];
            sAinv[t] = sacc;
        }
        GLIO_WAVE_LDS_SYNC();
        if (lane == 0) {
            double f2 = 0;
            for (int k = 0; k < 225; ++k) f2 += sAinv[k] * sAinv[k];
            if (!(f2 > 0.0) || !isfinite(f2) || !(1.0 / sqrt(f2) > 1e-7)) s_fast = 0;
        }
        GLIO_WAVE_LDS_SYNC();
    }
    if (!s_fast) {
        for (int e = lane; e < 256; e += 64) { jbuf[e] = sA[e]; jbuf[256 + e] = ((e >> 4) == (e & 15)) ? 1.0 : 0.0; }
        chain_wave_sync<true>();
        const int cur = jacobi16_wave<true>(jbuf, lane);
        const double* Am = jbuf + cur * 512; const double* Vm = Am + 256;
        for (int t = lane; t < 225; t += 64) {
            const int i = t / 15, j = t % 15;
            double sacc = 0;
            for (int k = 0; k < 16; ++k) { const double w = Am[k * 17]; sacc += Vm[i * 16 + k] * (w > 1e-8 ? 1.0 / w : 0.0) * Vm[j * 16 + k]; }
            sAinv[t] = sacc;
        }
        GLIO_WAVE_LDS_SYNC();
    }
    for (int t = lane; t < 225; t += 64) Ainv_out[t] = sAinv[t];
}
// blockIdx.x = i < n: row i of S (lower part) ; blockIdx.x = n: the carried right-hand side.  128 threads.
__global__ __launch_bounds__(128) void k_marg_rows(const double* __restrict__ A, const double* __restrict__ b, const int n, const double* __restrict__ Ainv,
                                                   double* __restrict__ Twork, double* __restrict__ Lwork, double* __restrict__ tol, int* __restrict__ viol) {
    __builtin_amdgcn_s_setprio(3);
    __shared__ double sA[225], sT[15], sArow[15];
    const int tid = threadIdx.x, m = 15, pos = m + n, i = blockIdx.x;
    for (int k = tid; k < 225; k += 128) sA[k] = Ainv[k];
    if (i < n && tid < 15) sArow[tid] = A[(size_t)(m + i) * pos + tid];
    __syncthreads();
    if (i < n) {
        if (tid < 15) {                                      // T[i][j] = sum_k A[m + i][k] Ainv[k][j], k ascending
            double sacc = 0;
            for (int k = 0; k < 15; ++k) sacc += sArow[k] * sA[k * 15 + tid];
            sT[tid] = sacc;
            Twork[i * 15 + tid] = sacc;
        }
        __syncthreads();
        int bad = 0;
        for (int j = tid; j <= i; j += 128) {
            double sacc = A[(size_t)(m + i) * pos + m + j];
            for (int k = 0; k < 15; ++k) sacc -= sT[k] * A[(size_t)k * pos + m + j];
            Lwork[(size_t)i * n + j] = sacc;
            if (j == i) tol[i] = fmax(1e-8, 1e-9 * sacc);
            else {
                const int bi = i < 15 ? 0 : 1 + (i - 15) / 6, bj = j < 15 ? 0 : 1 + (j - 15) / 6;
                if (bi != bj && sacc != 0.0) bad = 1;
            }
        }
        if (bad) atomicOr(viol, 1);
    } else {
        // bs[j] = b[m + j] - sum_k T[j][k] b[k]: T of row j recomputed here (this workgroup does not wait for the others)
        for (int j = tid; j < n; j += 128) {
            double sacc = b[m + j];
            for (int k = 0; k < 15; ++k) {
                double tj = 0;
                for (int l = 0; l < 15; ++l) tj += A[(size_t)(m + j) * pos + l] * sA[l * 15 + k];
                sacc -= tj * b[k];
            }
            Lwork[(size_t)n * n + j] = sacc;
        }
    }
}
// SMALL: the host knows the Schur complement to be block diagonal (no prior, or a prior that is block diagonal by keyframe: LiDAR blocks, the dropped
// keyframe's IMU edge and such a prior produce EXACT zeros between the blocks) -- the kernel then needs 3 KB of static LDS instead of the dense routine's
// panels (~100 KB: a request that, beside the batch association's searches, waits for a compute unit to EMPTY).  Should the pattern check of k_marg_rows
// have found a non-zero after all, it reports failure (ok = 0: the caller is left without a prior, as for a rank-deficient complement).
template <bool SMALL>
__global__ __launch_bounds__(TR_THREADS) void k_marg_root(const int n, double* Lwork, const double* __restrict__ tol, const int* __restrict__ viol, double* J0, double* r0, int* ok) {
    __builtin_amdgcn_s_setprio(3);      // (a short latency-bound kernel that shares its compute unit with the batch association's wide launches: its wavefronts issue first)
    const int tid = threadIdx.x;
    __shared__ double s_tol[SMALL ? 6 * GLIO_MAX_WINDOW + 16 : 2];
    __shared__ int s_flag[8];
    double* Bp = SMALL ? nullptr : reinterpret_cast<double*>(tr_lds);
    double* part = SMALL ? nullptr : Bp + TR_NB * bp_stride(n);
    double* sD = SMALL ? nullptr : part + 16 * 256;
    double* ylds = SMALL ? s_tol : sD + (TR_NB + 1) * TR_PS;
    double* red = SMALL ? nullptr : ylds + n + (n & 1);
    int* flag = SMALL ? s_flag : reinterpret_cast<int*>(red + 32);
    for (int j = tid; j < n; j += TR_THREADS) ylds[j] = tol[j];
    const bool bdiag = (n >= 15 && (n - 15) % 6 == 0) && *viol == 0;
    if (tid == 0) *flag = 0;
    __syncthreads();
    if (SMALL && !bdiag) { if (tid == 0) *ok = 0; return; }
    bool good;
    if (bdiag) {
        const int lane = tid & 63, wv = tid >> 6, nblk = 1 + (n - 15) / 6;
        for (int bI = wv; bI < nblk; bI += TR_WAVES) {
            const int o = bI == 0 ? 0 : 15 + 6 * (bI - 1), bs = bI == 0 ? 15 : 6;
            const bool isrow = lane < bs, isrhs = lane == TR_NB;
            double a[TR_NB];
#pragma unroll
            for (int j = 0; j < TR_NB; ++j) {
                double v = (lane < TR_NB && lane == j) ? 1.0 : 0.0;
                if (isrow && j < bs) v = j <= lane ? Lwork[(size_t)(o + lane) * n + o + j] : 0.0;
                if (isrhs && j < bs) v = Lwork[(size_t)n * n + o + j];
                a[j] = v;
            }
            const double tolv = isrow ? ylds[o + lane] : 0.0;
            bool bad = false;
#pragma unroll
            for (int j = 0; j < TR_NB; ++j) {
                double djj = readlane_d(a[j], j);
                const bool null_pivot = j < bs && djj <= readlane_d(tolv, j);
                if (!null_pivot && (!(djj > 0.0) || !isfinite(djj))) { bad = true; djj = 1.0; }
                const double rd = null_pivot ? 0.0 : rsqrt(djj);
                const double lij = (lane == j) ? djj * rd : a[j] * rd;
                a[j] = lij;
#pragma unroll
                for (int c2 = j + 1; c2 < TR_NB; ++c2) a[c2] -= lij * readlane_d(lij, c2);
            }
            // J0 = L^T and r0 straight from the registers: J0[j][o + lane] = L[o + lane][j] (the rest of J0 is zero: written below)
            if (isrow) {
#pragma unroll
                for (int j = 0; j < TR_NB; ++j) if (j < bs && j <= lane) Lwork[(size_t)(o + lane) * n + o + j] = a[j];
            } else if (isrhs) {
#pragma unroll
                for (int j = 0; j < TR_NB; ++j) if (j < bs) Lwork[(size_t)n * n + o + j] = a[j];
            }
            if (bad && lane == 0) *flag = 1 + o;
        }
        __threadfence_block();
        __syncthreads();
        good = *flag == 0;
    } else {
        if (SMALL) good = false;
        else good = chol_left_looking<true>(Lwork, n, Bp, part, sD, flag, 0, ylds);
    }
    if (good) {
        for (int i = tid >> 6; i < n; i += TR_WAVES)
            for (int j = tid & 63; j < n; j += 64) J0[(size_t)i * n + j] = (j >= i) ? Lwork[(size_t)j * n + i] : 0.0;
        for (int j = tid; j < n; j += TR_THREADS) r0[j] = Lwork[(size_t)n * n + j];
    }
    if (tid == 0) *ok = good ? 1 : 0;
}

// marginalization ordering: [T0 Q0 SB0 | T1 Q1 SB1 | T2 Q2 | ... ] -> index in the window state vector (15 per slot)
__device__ __forceinline__ int marg_state_index(int mi) { return mi < 30 ? mi : 15 * (2 + (mi - 30) / 6) + (mi - 30) % 6; }

struct MargAsmArgs {
    int W, pos, np, has_prior, imu_edge0;
    const double* lidar_blocks; const PairBlock* imu_blocks; const double* pH; const double* pg; const int* prior_index;
    double* A; double* b;
};
__global__ __launch_bounds__(256) void k_marg_assemble(const MargAsmArgs a) {
    __builtin_amdgcn_s_setprio(3);      // (a short latency-bound kernel that shares its compute unit with the batch association's wide launches: its wavefronts issue first)
    const int pos = a.pos;
    for (int r = blockIdx.x; r <= pos; r += gridDim.x) {
        for (int c = threadIdx.x; c < pos; c += blockDim.x) {
            const int sc_i = marg_state_index(c), sc = sc_i / 15, lc = sc_i % 15;
            double sacc = 0;
            if (r == pos) {          // b
                if (lc < 6) sacc += a.lidar_blocks[sc * GLIO_LIDAR_ACC + 21 + lc];
                if (a.imu_edge0 >= 0 && sc <= 1) sacc += a.imu_blocks[a.imu_edge0].g[15 * sc + lc];
                if (a.has_prior) { const int pi = a.prior_index[sc_i]; if (pi >= 0) sacc += a.pg[pi]; }
                a.b[c] = sacc;
                continue;
            }
            const int sr_i = marg_state_index(r), sr = sr_i / 15, lr = sr_i % 15;
            if (sr == sc && lr < 6 && lc < 6) {
                const int i = lr <= lc ? lr : lc, j = lr <= lc ? lc : lr;
                sacc += a.lidar_blocks[sr * GLIO_LIDAR_ACC + i * 6 - (i * (i - 1)) / 2 + (j - i)];
            }
            if (a.imu_edge0 >= 0 && sr <= 1 && sc <= 1) sacc += a.imu_blocks[a.imu_edge0].H[(15 * sr + lr) * GLIO_PAIR_DIM + 15 * sc + lc];
            if (a.has_prior) { const int pi = a.prior_index[sr_i], pj = a.prior_index[sc_i]; if (pi >= 0 && pj >= 0) sacc += a.pH[(size_t)pi * a.np + pj]; }
            a.A[(size_t)r * pos + c] = sacc;
        }
    }
}

int glio_launch_marginalize(glio_ctx* c, int imu_edge0, double** J0_dev, double** r0_dev, int** ok_dev) {
    const int W = c->W, n = 6 * (W - 1) + 9, pos = 15 + n;
    MargAsmArgs a;
    a.W = W; a.pos = pos; a.np = c->prior_n; a.has_prior = c->prior_n > 0; a.imu_edge0 = imu_edge0;
    a.lidar_blocks = c->d_lidar_blocks; a.imu_blocks = c->d_imu_blocks; a.pH = c->d_prior_H; a.pg = c->d_prior_g; a.prior_index = c->d_prior_index;
    a.A = c->d_H[0]; a.b = c->d_g[0];
    // (both dense H buffers are overwritten below -- the pos x pos A in d_H[0], J0 in d_H[1]: whatever a band-only assembly assumed about zeros outside
    //  the band no longer holds, for EVERY caller: glio_marginalize, glio_marginalize_keep, the timing hook)
    c->h_band_clean = 0;
    hipLaunchKernelGGL(k_marg_assemble, dim3(pos + 1), dim3(256), 0, c->stream, a);
    int* d_ok = reinterpret_cast<int*>(c->d_vec + 9 * (size_t)c->n_max);
    double* Twork = c->d_vec;                    // n x 15 <= 10 n_max doubles? n*15 <= 15W*... checked by the caller
    static const bool split = !(getenv("GLIO_MARG_SPLIT") && atoi(getenv("GLIO_MARG_SPLIT")) == 0);      // (0: the one-workgroup form, for A/B and tests)
    // scratch of the split form: Ainv (225), the tolerances (n), the pattern flag -- behind the (n + 1) x n work matrix in d_L, when it fits there
    const size_t lwork = (size_t)(n + 1) * n + 2, need = 226 + (size_t)n + 2 + 1100, have = (size_t)(c->n_max + 1) * c->n_max;
    if (split && lwork + need <= have) {
        double* Ainv = c->d_L + ((lwork + 1) & ~(size_t)1);
        double* tol = Ainv + 226;
        int* viol = reinterpret_cast<int*>(tol + n + 1);
        double* jbuf = tol + n + 2;                  // the Jacobi fallback's working matrices (global memory: k_marg_inv)
        // the complement is block diagonal by construction when the old prior is (or there is none): the root then needs no panels
        // (a prior the CALLER handed over is block diagonal only to glio_set_prior's tolerance: its complement takes the general root)
        const bool bd = (c->prior_n == 0 || (c->arrow.prior_chain && c->prior_device_made)) && n >= 15 && (n - 15) % 6 == 0;
        hipLaunchKernelGGL(k_marg_inv, dim3(1), dim3(64), 0, c->stream, c->d_H[0], n, Ainv, viol, jbuf);
        hipLaunchKernelGGL(k_marg_rows, dim3(n + 1), dim3(128), 0, c->stream, c->d_H[0], c->d_g[0], n, Ainv, Twork, c->d_L, tol, viol);
        if (bd) hipLaunchKernelGGL(k_marg_root<true>, dim3(1), dim3(TR_THREADS), 0, c->stream, n, c->d_L, tol, viol, c->d_H[1], c->d_g[1], d_ok);
        else hipLaunchKernelGGL(k_marg_root<false>, dim3(1), dim3(TR_THREADS), glio_tr_step_lds_bytes(n), c->stream, n, c->d_L, tol, viol, c->d_H[1], c->d_g[1], d_ok);
    } else
    hipLaunchKernelGGL(k_marg_schur, dim3(1), dim3(TR_THREADS), glio_tr_step_lds_bytes(n), c->stream, c->d_H[0], c->d_g[0], n, c->d_L, Twork,
                       c->d_H[1], c->d_g[1], d_ok);
    *J0_dev = c->d_H[1]; *r0_dev = c->d_g[1]; *ok_dev = d_ok;
    return 0;
}

// ---- test hook: solve A x = b for a dense SPD n x n matrix with the in-kernel blocked Cholesky
__global__ __launch_bounds__(TR_THREADS) void k_chol_test(double* L, int n, double* x, int* ok, int skip) {
    double* Bp = reinterpret_cast<double*>(tr_lds);
    double* part = Bp + TR_NB * bp_stride(n);
    double* sD = part + 16 * 256;
    double* ylds = sD + (TR_NB + 1) * TR_PS;
    int* flag = reinterpret_cast<int*>(ylds + n + (n & 1) + 32);
    const bool good = chol_left_looking(L, n, Bp, part, sD, flag, skip);
    if (good && !(skip & 8)) {
        for (int j = threadIdx.x; j < n; j += TR_THREADS) ylds[j] = L[(size_t)n * n + j];
        __syncthreads();
        back_substitute(L, n, ylds, sD);
        for (int j = threadIdx.x; j < n; j += TR_THREADS) x[j] = ylds[j];
    }
    if (threadIdx.x == 0) *ok = good ? 1 : 0;
}

extern "C" int glio_debug_chol_solve(glio_ctx* c, int n, const double* A, const double* b, double* x) {
    if (!c || n < 1 || n > c->n_max) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipSetDevice(c->device));
    GLIO_HIP_CHECK(hipMemcpy(c->d_L, A, (size_t)n * n * 8, hipMemcpyHostToDevice));
    GLIO_HIP_CHECK(hipMemcpy(c->d_L + (size_t)n * n, b, (size_t)n * 8, hipMemcpyHostToDevice));
    int* d_ok = reinterpret_cast<int*>(c->d_vec + 9 * (size_t)c->n_max);
    hipLaunchKernelGGL(k_chol_test, dim3(1), dim3(TR_THREADS), glio_tr_step_lds_bytes(n), c->stream, c->d_L, n, c->d_vec, d_ok, 0);
    GLIO_HIP_CHECK(hipGetLastError());
    int ok = 0;
    GLIO_HIP_CHECK(hipMemcpyAsync(x, c->d_vec, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    GLIO_HIP_CHECK(hipMemcpyAsync(&ok, d_ok, 4, hipMemcpyDeviceToHost, c->stream));
    GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));
    return ok ? GLIO_OK : GLIO_E_NUMERIC;
}

int glio_tr_step_configure(size_t max_lds) {
#define TR_CONF_(kernel, bytes) do { const hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
        if (e_ != hipSuccess) { (void)hipGetLastError(); glio_set_error("hipFuncSetAttribute(%s, %d B of dynamic LDS) failed: %s", #kernel, (int)(bytes), hipGetErrorString(e_)); return GLIO_E_HIP; } } while (0)
    TR_CONF_(k_tr_finish, max_lds);
    TR_CONF_(k_chol_test, max_lds);
    TR_CONF_(k_marg_schur, max_lds);
    TR_CONF_(k_marg_root<false>, max_lds);
    TR_CONF_(k_arrow_forward, max_lds);
    TR_CONF_(k_arrow_solve, max_lds);
    TR_CONF_(k_chain_solve<false>, max_lds - 1024);
    TR_CONF_(k_chain_solve<true>, max_lds - 1024);
    TR_CONF_(k_chain_step, max_lds - 2048);
#undef TR_CONF_
    return GLIO_OK;
}

extern "C" int glio_debug_read_vec(glio_ctx* c, int k, double* out, int n) {
    if (!c || k < 0 || k > 9 || n > c->n_max) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipMemcpy(out, c->d_vec + (size_t)k * c->vstride, (size_t)n * 8, hipMemcpyDeviceToHost));
    return GLIO_OK;
}

// timing hook: `reps` factorisations of an n x n identity-like matrix with selected phases skipped
// (skip bits: 1 = MFMA update, 2 = diagonal-block factor, 4 = row solves, 8 = back substitution)
extern "C" int glio_debug_chol_time(glio_ctx* c, int n, int reps, int skip, float* ms_out) {
    if (!c || n < 1 || n > c->n_max) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipSetDevice(c->device));
    std::vector<double> A((size_t)(n + 1) * n, 0.0);
    for (int i = 0; i < n; ++i) { A[(size_t)i * n + i] = 4.0 + i % 3; for (int j = 0; j < i; ++j) A[(size_t)i * n + j] = 1e-3 * ((i * 7 + j * 3) % 11); }
    for (int j = 0; j < n; ++j) A[(size_t)n * n + j] = 1.0;
    double* d_src = nullptr;
    GLIO_HIP_CHECK(hipMalloc((void**)&d_src, A.size() * 8));
    GLIO_HIP_CHECK(hipMemcpy(d_src, A.data(), A.size() * 8, hipMemcpyHostToDevice));
    int* d_ok = reinterpret_cast<int*>(c->d_vec + 9 * (size_t)c->n_max);
    float total = 0;
    for (int r = 0; r < reps + 1; ++r) {
        GLIO_HIP_CHECK(hipMemcpyAsync(c->d_L, d_src, A.size() * 8, hipMemcpyDeviceToDevice, c->stream));
        GLIO_HIP_CHECK(hipEventRecord(c->ev0, c->stream));
        hipLaunchKernelGGL(k_chol_test, dim3(1), dim3(TR_THREADS), glio_tr_step_lds_bytes(n), c->stream, c->d_L, n, c->d_vec, d_ok, skip);
        GLIO_HIP_CHECK(hipEventRecord(c->ev1, c->stream));
        GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));
        float ms = 0;
        GLIO_HIP_CHECK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        if (r > 0) total += ms;
    }
    hipFree(d_src);
    *ms_out = total / reps;
    return GLIO_OK;
}

// ---- test hook: the DPP / permlane-swap exchanges of glio_device.h against the __shfl_xor forms they replace, bit for bit.
// One wavefront; in[64 * rounds]; returns the number of mismatching (lane, check) pairs.
__global__ __launch_bounds__(64) void k_wave_reduce_check(const double* in, int rounds, int* bad) {
    const int lane = threadIdx.x;
    int nbad = 0;
    auto same = [](const double a, const double b) { return __double_as_longlong(a) == __double_as_longlong(b); };
    for (int r = 0; r < rounds; ++r) {
        const double v = in[64 * r + lane], w = in[64 * ((r + 1) % rounds) + lane];
        nbad += !same(wave_sum(v), wave_sum_shfl(v));
        { double m = fabs(v); for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64)); nbad += !same(wave_max(fabs(v)), m); }
        nbad += !same(lane_xor_sum<32>(v), v + __shfl_xor(v, 32, 64));
        nbad += !same(lane_xor_sum<16>(v), v + __shfl_xor(v, 16, 64));
        nbad += !same(lane_xor_sum<8>(v), v + __shfl_xor(v, 8, 64));
        nbad += !same(lane_xor_sum<4>(v), v + __shfl_xor(v, 4, 64));
        nbad += !same(lane_xor_sum<2>(v), v + __shfl_xor(v, 2, 64));
        nbad += !same(lane_xor_sum<1>(v), v + __shfl_xor(v, 1, 64));
        nbad += !same(lane_xor_row_d<8>(v), __shfl_xor(v, 8, 64)) + !same(lane_xor_row_d<4>(v), __shfl_xor(v, 4, 64));
        nbad += !same(lane_xor_row_d<2>(v), __shfl_xor(v, 2, 64)) + !same(lane_xor_row_d<1>(v), __shfl_xor(v, 1, 64));
        {   // the reduce-scatter step of the value-splitting butterflies (k3_reduce_store, butterfly64): keep + received
            double x, y;
            lane_swap32(v, w, x, y);
            const bool hi = (lane & 32) != 0;
            nbad += !same(x + y, (hi ? w : v) + __shfl_xor(hi ? v : w, 32, 64));
            lane_swap16(v, w, x, y);
            const bool hi16 = (lane & 16) != 0;
            nbad += !same(x + y, (hi16 ? w : v) + __shfl_xor(hi16 ? v : w, 16, 64));
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}
extern "C" int glio_debug_wave_reduce_check(glio_ctx* c, const double* values, int rounds, int* mismatches) {
    if (!c || !values || rounds < 1 || rounds > 64 || !mismatches) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipSetDevice(c->device));
    double* d_in = c->d_L;                                     // scratch big enough for 64 * 64 doubles (n_max >= 15)
    int* d_bad = reinterpret_cast<int*>(c->d_vec + 9 * (size_t)c->n_max);
    GLIO_HIP_CHECK(hipMemcpyAsync(d_in, values, (size_t)64 * rounds * 8, hipMemcpyHostToDevice, c->stream));
    GLIO_HIP_CHECK(hipMemsetAsync(d_bad, 0, 4, c->stream));
    hipLaunchKernelGGL(k_wave_reduce_check, dim3(1), dim3(64), 0, c->stream, d_in, rounds, d_bad);
    GLIO_HIP_CHECK(hipGetLastError());
    GLIO_HIP_CHECK(hipMemcpyAsync(mismatches, d_bad, 4, hipMemcpyDeviceToHost, c->stream));
    GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));
    return GLIO_OK;
}
