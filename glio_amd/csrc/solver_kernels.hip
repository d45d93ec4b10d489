// solver_kernels.hip -- K7: one trust-region iteration of ceres::Solve as configured by the reference
// (GLIO/src/Estimator.cpp:2424-2433: SPARSE_NORMAL_CHOLESKY, DOGLEG, 15 iterations, monotonic steps),
// restated on the dense device-resident normal equations and run entirely on the GPU:
//
//   three single-workgroup launches per iteration (512 lanes = 8 wavefronts on one CU), each reading and
//   writing the SolverStatus record in device memory:
//     k_tr_prepare  step evaluation of the candidate produced by the previous iteration: parameter /
//                   function tolerance, relative decrease -> accept (swap the double-buffered H,g,x) or
//                   reject, dogleg radius update (Ceres 1.14 TrustRegionMinimizer); loop-top checks
//                   (max iterations, gradient tolerance, min radius); Jacobi scaling, D = sqrt(clamp(diag)),
//                   Cauchy point
//     k_tr_factor   Gauss-Newton step from an in-LDS-panel blocked right-looking Cholesky of
//                   (S H S + mu D^2) with the right-hand side carried as an extra row; mu retry x10
//     k_tr_dogleg   traditional dogleg interpolation, model cost change, candidate x (+) delta
//                   (an invalid step, model cost change <= 0, consumes an iteration without a candidate)
//   The host enqueues max_iterations+1 such groups interleaved with the linearisation kernels and never
//   reads anything back until the end; kernels exit immediately once status.done is set, and the
//   linearisation kernels also when no candidate is pending.
//
// The Cholesky works on (n+1) x n doubles in L2-resident global memory with a 16-column panel staged
// in LDS (<= 120 KB at n = 826); its trailing update is the one GEMM-shaped piece of the whole path and
// runs on the matrix cores (v_mfma_f64_16x16x4_f64) -- latency/LDS-bound dense fp64, not roofline material.
#include "glio_device.h"

#define TR_THREADS 512
#define TR_WAVES (TR_THREADS / 64)
#define TR_NB 16
#define TR_PS (TR_NB + 2)      // padded LDS row stride of the panel (conflict-free MFMA operand reads)

struct TrArgs {
    int W, n, n_ddt, max_iterations;
    double min_relative_decrease, function_tolerance, gradient_tolerance, parameter_tolerance;
    double min_radius, initial_radius;
    int jacobi_scaling;
    double* x0; double* x1; double* xout;
    const double* H0; const double* H1; const double* g0; const double* g1; const double* c0; const double* c1;
    double* L; double* vec; int vstride;
    SolverStatus* status;
};

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
__device__ __forceinline__ double block_max(double v, double* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    double s = 0;
#pragma unroll
    for (int k = 0; k < TR_WAVES; ++k) s = fmax(s, red[k]);
    return s;
}
// NOTE: no __restrict__ on anything in this file: every buffer here is handed between lanes of the
// workgroup across s_barrier, and noalias lets LLVM move such loads above the barrier (observed: stale
// sD / y reads in back_substitute, deterministic per binary).
// y = H u, H n x n row-major in global memory; one wavefront per row, lanes across columns
__device__ __forceinline__ void matvec(const double* H, const double* u, double* y, int n) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int r = 4 * wv; r < n; r += 4 * TR_WAVES) {       // 4 rows per wavefront pass: independent loads + reductions
        double s[4] = {0, 0, 0, 0};
        for (int c = lane; c < n; c += 64) {
            const double uc = u[c];
#pragma unroll
            for (int q = 0; q < 4; ++q) if (r + q < n) s[q] += H[(size_t)(r + q) * n + c] * uc;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] = wave_sum(s[q]);
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (r + q < n) y[r + q] = s[q];
        }
    }
    __syncthreads();
}

typedef double v4f64 __attribute__((ext_vector_type(4)));

// In-place blocked Cholesky of the leading n x n block of the (n+1) x n row-major matrix A (lower
// triangle); row n is carried along (forward substitution of the right-hand side).  Returns false on
// a non-positive / non-finite pivot.  lds: panel [(n+1 rounded to 16) x TR_PS], sD [TR_NB x TR_PS].
//
// Per 16-column panel:
//   (1,2) the 16x16 diagonal block is factored by wavefront 0: lane i keeps row i in registers and the
//         pivot column is broadcast with v_readlane (no LDS round trip, no s_barrier inside);
//   (3)   the rows below are solved one lane per row (forward substitution against the LDS copy of
//         L_kk, uniform-address LDS reads = broadcasts) and staged in the LDS panel P;
//   (4)   the trailing update C -= P P^T runs on the matrix cores: one wavefront per 16x16 tile,
//         4 x v_mfma_f64_16x16x4_f64 (A[i][k] = -P[I+i][k], B[k][j] = P[J+j][k]: both operands are the
//         same "lane l -> row l&15, column l>>4" LDS read; TR_PS = 18 makes it bank-conflict free),
//         accumulator initialised from the tile in L2 and written straight back.
__device__ bool chol_augmented(double* A, const int n, double* panel, double* sD, int* flag) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ld = n;
    if (tid == 0) *flag = 0;
    for (int k0 = 0; k0 < n; k0 += TR_NB) {
        const int nb = min(TR_NB, n - k0);
        // (1) diagonal block -> LDS
        if (tid < TR_NB * TR_NB) {
            const int i = tid / TR_NB, j = tid % TR_NB;
            sD[i * TR_PS + j] = (i < nb && j <= i) ? A[(size_t)(k0 + i) * ld + k0 + j] : (i == j ? 1.0 : 0.0);
        }
        __syncthreads();
        // (2) factor it in wavefront 0
        if (wv == 0) {
            double a[TR_NB];
#pragma unroll
            for (int j = 0; j < TR_NB; ++j) a[j] = (lane < TR_NB) ? sD[lane * TR_PS + j] : 0.0;
            bool bad = false;
#pragma unroll
            for (int j = 0; j < TR_NB; ++j) {
                double djj = readlane_d(a[j], j);
                if (!(djj > 0.0) || !isfinite(djj)) { bad = true; djj = 1.0; }
                const double d = sqrt(djj);
                const double lij = (lane == j) ? d : a[j] / d;
                a[j] = lij;
#pragma unroll
                for (int c = j + 1; c < TR_NB; ++c) {
                    const double lcj = readlane_d(lij, c);
                    if (lane >= c) a[c] -= lij * lcj;
                }
            }
            if (lane < TR_NB) {
#pragma unroll
                for (int j = 0; j < TR_NB; ++j) {
                    if (j <= lane) {
                        sD[lane * TR_PS + j] = a[j];
                        if (lane < nb && j < nb) A[(size_t)(k0 + lane) * ld + k0 + j] = a[j];
                    }
                }
            }
            if (bad && lane == 0) *flag = 1 + k0;
        }
        __syncthreads();
        if (*flag) return false;
        // (3) rows below (incl. the carried row n): X L_kk^T = A_panel, one lane per row
        const int r0 = k0 + nb;
        const int m = n + 1 - r0;
        const int mpad = (m + 15) & ~15;
        for (int i = tid; i < mpad; i += TR_THREADS) {
            double xv[TR_NB];
            if (i < m) {
                double* row = A + (size_t)(r0 + i) * ld + k0;
#pragma unroll
                for (int j = 0; j < TR_NB; ++j) xv[j] = (j < nb) ? row[j] : 0.0;
#pragma unroll
                for (int j = 0; j < TR_NB; ++j) {
                    double s = xv[j];
#pragma unroll
                    for (int k = 0; k < j; ++k) s -= xv[k] * sD[j * TR_PS + k];
                    xv[j] = s / sD[j * TR_PS + j];
                }
#pragma unroll
                for (int j = 0; j < TR_NB; ++j) if (j < nb) row[j] = xv[j];
            } else {
#pragma unroll
                for (int j = 0; j < TR_NB; ++j) xv[j] = 0.0;
            }
#pragma unroll
            for (int j = 0; j < TR_NB; ++j) panel[i * TR_PS + j] = (j < nb) ? xv[j] : 0.0;
        }
        __syncthreads();
        // (4) trailing update on the matrix cores
        const int mc = n - r0;                 // trailing columns (the carried row is not a column)
        if (mc > 0) {
            const int Tb = mpad >> 4;
            const int ntiles = Tb * (Tb + 1) / 2;
            const int li = lane & 15, lk = lane >> 4;
            // SYRK_ILP independent tiles per wavefront iteration: their C loads and MFMA chains overlap
            constexpr int SYRK_ILP = 3;
            for (int id0 = wv; id0 < ntiles; id0 += TR_WAVES * SYRK_ILP) {
                double av[SYRK_ILP][4], bv[SYRK_ILP][4];
                v4f64 acc[SYRK_ILP];
                double* cbase[SYRK_ILP];
                bool okr[SYRK_ILP][4];
#pragma unroll
                for (int u = 0; u < SYRK_ILP; ++u) {
                    const int id = id0 + u * TR_WAVES;
                    const bool live = id < ntiles;
                    const int idc = live ? id : 0;
                    int ti = (int)((sqrtf(8.0f * (float)idc + 1.0f) - 1.0f) * 0.5f);
                    while ((ti + 1) * (ti + 2) / 2 <= idc) ++ti;
                    while (ti * (ti + 1) / 2 > idc) --ti;
                    const int tj = idc - ti * (ti + 1) / 2;
                    const double* pa = panel + (ti * 16 + li) * TR_PS + lk;
                    const double* pb = panel + (tj * 16 + li) * TR_PS + lk;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { av[u][q] = -pa[4 * q]; bv[u][q] = pb[4 * q]; }
                    // C tile: lane l, reg r -> row (l>>4) + 4 r, col l&15
                    cbase[u] = A + (size_t)(r0 + ti * 16 + lk) * ld + r0 + tj * 16 + li;
                    const int col = tj * 16 + li;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = ti * 16 + lk + 4 * r;
                        okr[u][r] = live && row < m && col < mc && col <= row;
                        acc[u][r] = okr[u][r] ? cbase[u][(size_t)(4 * r) * ld] : 0.0;
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int u = 0; u < SYRK_ILP; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][q], bv[u][q], acc[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < SYRK_ILP; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (okr[u][r]) cbase[u][(size_t)(4 * r) * ld] = acc[u][r];
            }
        }
        __syncthreads();
    }
    return true;
}

// Solve L^T z = y (L lower n x n in A, y in LDS, overwritten by z), blocked from the bottom up.
__device__ void back_substitute(const double* A, const int n, double* y, double* sD) {
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
#pragma unroll
            for (int k = TR_NB - 1; k >= 0; --k) {
                const double zk = readlane_d(yi, k) / sD[k * TR_PS + k];
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

__device__ __forceinline__ void finalize(const TrArgs& a, const SolverStatus& s) {
    const double* xc = s.cur ? a.x1 : a.x0;
    const int nx = 16 * a.W + a.n_ddt;
    for (int k = threadIdx.x; k < nx; k += TR_THREADS) a.xout[k] = xc[k];
    if (threadIdx.x == 0) *a.status = s;
}

extern __shared__ __attribute__((aligned(16))) unsigned char tr_lds[];

// workspace vectors (global, persist across the launches of one solve)
#define V_SCALE(a) ((a).vec + 0 * (a).vstride)
#define V_DIAG(a) ((a).vec + 1 * (a).vstride)
#define V_GRAD(a) ((a).vec + 2 * (a).vstride)   /* g~ = gs / D                      */
#define V_GN(a) ((a).vec + 3 * (a).vstride)     /* Gauss-Newton step in D-space      */
#define V_STEP(a) ((a).vec + 4 * (a).vstride)
#define V_W(a) ((a).vec + 5 * (a).vstride)      /* scale * step = delta              */
#define V_T1(a) ((a).vec + 6 * (a).vstride)
#define V_T2(a) ((a).vec + 7 * (a).vstride)

// ------------------------------------------------------------------------------------------------
// K7a  k_tr_prepare: step evaluation of the pending candidate, loop-top checks, Cauchy point
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TR_THREADS) void k_tr_prepare(const TrArgs a) {
    __shared__ double red[32];
    __shared__ SolverStatus s;
    const int tid = threadIdx.x;
    const int n = a.n, W = a.W, nx = 16 * W + a.n_ddt;
    if (tid == 0) s = *a.status;
    __syncthreads();
    if (s.done) return;
    double* scale = V_SCALE(a); double* diag = V_DIAG(a); double* grad = V_GRAD(a); double* t1 = V_T1(a); double* t2 = V_T2(a);

    if (s.cand_pending) {
        const int cand = 1 - s.cur;
        if (s.phase == 0) {
            const double* Hc = cand ? a.H1 : a.H0;
            for (int i = tid; i < n; i += TR_THREADS) scale[i] = a.jacobi_scaling ? 1.0 / (1.0 + sqrt(Hc[(size_t)i * n + i])) : 1.0;
            __syncthreads();
            if (tid == 0) {
                s.cur = cand;
                s.cost = *(cand ? a.c1 : a.c0);
                s.initial_cost = s.cost;
                s.phase = 1;
            }
        } else {
            const double* xc = s.cur ? a.x1 : a.x0;
            const double* xn = cand ? a.x1 : a.x0;
            double d2 = 0, x2 = 0;
            for (int k = tid; k < nx; k += TR_THREADS) { const double d = xc[k] - xn[k]; d2 += d * d; x2 += xc[k] * xc[k]; }
            d2 = block_sum(d2, red);
            x2 = block_sum(x2, red);
            if (tid == 0) {
                const double ccost = *(cand ? a.c1 : a.c0);
                const double step_norm = sqrt(d2), x_norm = sqrt(x2);
                if (step_norm <= a.parameter_tolerance * (x_norm + a.parameter_tolerance)) {
                    s.done = 1; s.termination = GLIO_TERM_PARAMETER_TOL;
                } else if (fabs(s.cost - ccost) <= a.function_tolerance * s.cost) {
                    s.done = 1; s.termination = GLIO_TERM_FUNCTION_TOL;
                } else {
                    const double rel = (s.cost - ccost) / s.model_cost_change;
                    if (rel > a.min_relative_decrease) {
                        s.cur = cand; s.cost = ccost; s.successful += 1;
                        if (rel < 0.25) s.radius *= 0.5;                                      // StepAccepted
                        if (rel > 0.75) s.radius = fmax(s.radius, 3.0 * s.dogleg_step_norm);
                        s.mu = fmax(1e-8, 2.0 * s.mu / 10.0);
                        s.reuse = 0;
                    } else {
                        s.radius *= 0.5; s.reuse = 1;                                         // StepRejected
                    }
                }
            }
        }
        __syncthreads();
        if (tid == 0) s.cand_pending = 0;
        __syncthreads();
    }
    if (s.done) { finalize(a, s); return; }

    const double* H = s.cur ? a.H1 : a.H0;
    const double* g = s.cur ? a.g1 : a.g0;
    const double* xc = s.cur ? a.x1 : a.x0;
    // gradient max norm = | x - Plus(x, -g) |_inf
    double gm = 0;
    for (int k = tid; k < n; k += TR_THREADS) {
        if (k < 15 * W && (k % 15) >= 3 && (k % 15) < 6) {
            if ((k % 15) == 3) {
                const int sl = k / 15;
                const double d[3] = {-g[k], -g[k + 1], -g[k + 2]};
                double q[4], qn[4];
                for (int c = 0; c < 4; ++c) q[c] = xc[3 * W + 4 * sl + c];
                d_quat_plus(q, d, qn);
                for (int c = 0; c < 4; ++c) gm = fmax(gm, fabs(q[c] - qn[c]));
            }
        } else gm = fmax(gm, fabs(g[k]));
    }
    gm = block_max(gm, red);
    if (tid == 0) {
        s.grad_max_norm = gm;
        if (s.iteration >= a.max_iterations) { s.done = 1; s.termination = GLIO_TERM_NO_CONVERGENCE; }
        else if (gm <= a.gradient_tolerance) { s.done = 1; s.termination = GLIO_TERM_GRADIENT_TOL; }
        else if (s.radius <= a.min_radius) { s.done = 1; s.termination = GLIO_TERM_MIN_RADIUS; }
        else s.iteration += 1;
    }
    __syncthreads();
    if (s.done) { finalize(a, s); return; }

    if (!s.reuse) {
        for (int i = tid; i < n; i += TR_THREADS) {
            double d = scale[i] * scale[i] * H[(size_t)i * n + i];
            d = d < 1e-6 ? 1e-6 : (d > 1e32 ? 1e32 : d);
            const double dd = sqrt(d);
            diag[i] = dd;
            const double gs = scale[i] * g[i];
            grad[i] = gs / dd;
            t1[i] = scale[i] * (gs / dd) / dd;          // u = S (g~ / D)
        }
        __syncthreads();
        matvec(H, t1, t2, n);                            // H u
        double p = 0, q2 = 0;
        for (int i = tid; i < n; i += TR_THREADS) { p += t1[i] * t2[i]; q2 += grad[i] * grad[i]; }
        p = block_sum(p, red);
        q2 = block_sum(q2, red);
        if (tid == 0) s.alpha = q2 / p;
    }
    __syncthreads();
    if (tid == 0) *a.status = s;
}

// ------------------------------------------------------------------------------------------------
// K7b  k_tr_factor: Gauss-Newton step, (S H S + mu D^2) y = S g by the blocked MFMA Cholesky
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TR_THREADS) void k_tr_factor(const TrArgs a) {
    const int tid = threadIdx.x;
    const int n = a.n;
    double* panel = reinterpret_cast<double*>(tr_lds);                    // ((n+1) padded to 16) x TR_PS
    double* sD = panel + (size_t)((n + 1 + 15) & ~15) * TR_PS;           // TR_NB x TR_PS
    double* ylds = sD + TR_NB * TR_PS;                                    // n
    double* red = ylds + n + (n & 1);                                     // 32
    int* flag = reinterpret_cast<int*>(red + 32);
    double* smu = red + 40;
    if (a.status->done || a.status->reuse) return;
    const double* H = a.status->cur ? a.H1 : a.H0;
    const double* g = a.status->cur ? a.g1 : a.g0;
    const double* scale = V_SCALE(a); const double* diag = V_DIAG(a); double* gn = V_GN(a);
    if (tid == 0) *smu = a.status->mu;
    __syncthreads();
    bool solved = false;
    for (int attempt = 0; attempt < 12; ++attempt) {
        const double mu = *smu;
        if (!(mu < 1.0)) break;
        for (int i = tid >> 6; i < n; i += TR_WAVES) {
            const double si = scale[i];
            const double* hrow = H + (size_t)i * n;
            double* lrow = a.L + (size_t)i * n;
            for (int j = tid & 63; j <= i; j += 64) {
                double v = si * hrow[j] * scale[j];
                if (i == j) v += mu * diag[i] * diag[i];
                lrow[j] = v;
            }
        }
        for (int j = tid; j < n; j += TR_THREADS) a.L[(size_t)n * n + j] = scale[j] * g[j];
        __syncthreads();
        const bool ok = chol_augmented(a.L, n, panel, sD, flag);
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
    if (solved) for (int i = tid; i < n; i += TR_THREADS) gn[i] = -diag[i] * ylds[i];
    __syncthreads();
    if (tid == 0) {
        if (solved) a.status->mu = fmax(1e-8, 2.0 * (*smu) / 10.0);
        else { a.status->mu = *smu; a.status->done = 1; a.status->termination = GLIO_TERM_FAILURE; }
    }
    if (!solved) {
        const double* xc = a.status->cur ? a.x1 : a.x0;
        for (int k = tid; k < 16 * a.W + a.n_ddt; k += TR_THREADS) a.xout[k] = xc[k];
    }
}

// ------------------------------------------------------------------------------------------------
// K7c  k_tr_dogleg: traditional dogleg interpolation, model cost change, candidate x (+) delta
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TR_THREADS) void k_tr_dogleg(const TrArgs a) {
    __shared__ double red[32];
    __shared__ SolverStatus s;
    const int tid = threadIdx.x;
    const int n = a.n, W = a.W;
    if (tid == 0) s = *a.status;
    __syncthreads();
    if (s.done) return;
    const double* H = s.cur ? a.H1 : a.H0;
    const double* g = s.cur ? a.g1 : a.g0;
    const double* scale = V_SCALE(a); const double* diag = V_DIAG(a); const double* grad = V_GRAD(a); const double* gn = V_GN(a);
    double* step = V_STEP(a); double* wvec = V_W(a); double* t2 = V_T2(a);
    double gg = 0, nn = 0, gd = 0;
    for (int i = tid; i < n; i += TR_THREADS) { gg += grad[i] * grad[i]; nn += gn[i] * gn[i]; gd += grad[i] * gn[i]; }
    gg = block_sum(gg, red); nn = block_sum(nn, red); gd = block_sum(gd, red);
    const double gnorm = sqrt(gg), gnn = sqrt(nn), radius = s.radius, alpha = s.alpha;
    double ca, cb, snorm;       // step = ca * grad + cb * gn
    if (gnn <= radius) { ca = 0.0; cb = 1.0; snorm = gnn; }
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
    double sn2 = 0;
    for (int i = tid; i < n; i += TR_THREADS) {
        const double sv = ca * grad[i] + cb * gn[i];
        sn2 += sv * sv;
        step[i] = sv / diag[i];
        wvec[i] = scale[i] * step[i];
    }
    sn2 = block_sum(sn2, red);
    if (snorm < 0) snorm = sqrt(sn2);
    // model cost change = -(g.w + w^T H w / 2), w = S step
    matvec(H, wvec, t2, n);
    double lin = 0, quad = 0;
    for (int i = tid; i < n; i += TR_THREADS) { lin += g[i] * wvec[i]; quad += wvec[i] * t2[i]; }
    lin = block_sum(lin, red); quad = block_sum(quad, red);
    const double mcc = -(lin + 0.5 * quad);
    const bool valid = mcc > 0.0;
    if (tid == 0) {
        s.dogleg_step_norm = snorm;
        if (!valid) {
            s.invalid += 1;
            if (s.invalid >= 5) { s.done = 1; s.termination = GLIO_TERM_FAILURE; }
            s.mu *= 10.0; s.reuse = 0;          // StepIsInvalid: consumes an iteration, no candidate
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

size_t glio_tr_step_lds_bytes(int n) {
    size_t d = (size_t)((n + 1 + 15) & ~15) * TR_PS;
    d += TR_NB * TR_PS;
    d += n + (n & 1);
    d += 32 + 16;
    return d * sizeof(double);
}

void glio_launch_tr_step(glio_ctx* c, int n_ddt) {
    TrArgs a;
    a.W = c->W; a.n = 15 * c->W + n_ddt; a.n_ddt = n_ddt; a.max_iterations = c->opts.max_iterations;
    a.min_relative_decrease = c->opts.min_relative_decrease; a.function_tolerance = c->opts.function_tolerance;
    a.gradient_tolerance = c->opts.gradient_tolerance; a.parameter_tolerance = c->opts.parameter_tolerance;
    a.min_radius = c->opts.min_trust_region_radius; a.initial_radius = c->opts.initial_trust_region_radius;
    a.jacobi_scaling = c->opts.jacobi_scaling;
    a.x0 = c->d_x[0]; a.x1 = c->d_x[1]; a.xout = c->d_xout;
    a.H0 = c->d_H[0]; a.H1 = c->d_H[1]; a.g0 = c->d_g[0]; a.g1 = c->d_g[1]; a.c0 = c->d_cost[0]; a.c1 = c->d_cost[1];
    a.L = c->d_L; a.vec = c->d_vec; a.vstride = c->n_max;
    a.status = c->d_status;
    hipLaunchKernelGGL(k_tr_prepare, dim3(1), dim3(TR_THREADS), 0, c->stream, a);
    hipLaunchKernelGGL(k_tr_factor, dim3(1), dim3(TR_THREADS), glio_tr_step_lds_bytes(a.n), c->stream, a);
    hipLaunchKernelGGL(k_tr_dogleg, dim3(1), dim3(TR_THREADS), 0, c->stream, a);
}

// ---- test hook: solve (A + 0) x = b for a dense SPD n x n matrix with the in-kernel blocked Cholesky
__global__ __launch_bounds__(TR_THREADS) void k_chol_test(double* L, int n, double* x, int* ok) {
    double* panel = reinterpret_cast<double*>(tr_lds);
    double* sD = panel + (size_t)((n + 1 + 15) & ~15) * TR_PS;
    double* ylds = sD + TR_NB * TR_PS;
    int* flag = reinterpret_cast<int*>(ylds + n + (n & 1) + 32);
    const bool good = chol_augmented(L, n, panel, sD, flag);
    if (good) {
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
    hipLaunchKernelGGL(k_chol_test, dim3(1), dim3(TR_THREADS), glio_tr_step_lds_bytes(n), c->stream, c->d_L, n, c->d_vec, d_ok);
    GLIO_HIP_CHECK(hipGetLastError());
    int ok = 0;
    GLIO_HIP_CHECK(hipMemcpyAsync(x, c->d_vec, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    GLIO_HIP_CHECK(hipMemcpyAsync(&ok, d_ok, 4, hipMemcpyDeviceToHost, c->stream));
    GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));
    return ok ? GLIO_OK : GLIO_E_NUMERIC;
}

void glio_tr_step_configure(size_t max_lds) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_tr_factor), hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_chol_test), hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds);
}

extern "C" int glio_debug_read_vec(glio_ctx* c, int k, double* out, int n) {
    if (!c || k < 0 || k > 9 || n > c->n_max) return GLIO_E_ARG;
    GLIO_HIP_CHECK(hipMemcpy(out, c->d_vec + (size_t)k * c->n_max, (size_t)n * 8, hipMemcpyDeviceToHost));
    return GLIO_OK;
}
