// batch_solve_kernels.hip -- the damped block-banded solve of the batch stage by BLOCK CYCLIC REDUCTION, parallel over the
// chip, replacing the one-workgroup sequential banded Cholesky (k_batch_factor + k_batch_backsolve, 6.5 ms at K = 2000)
// on the critical path of every iteration of the sharded stage (Estimator::optimizeBatchWithLandMark's normal equations,
// reference GLIO/src/Estimator.cpp:3004-3076,3275-3284: there Ceres' SPARSE_NORMAL_CHOLESKY).
//
// The K keyframes (6 unknowns each, half band `band` keyframes) are grouped into S = ceil(K / sb) SUPER-BLOCKS of
// sb = M / 6 >= band keyframes (M = 36 for band <= 6, 72 for band <= 12): in that blocking the matrix is block
// TRIDIAGONAL with dense M x M blocks.  Odd-even (nested-dissection) elimination: a level eliminates every other active
// node -- all of them at once, one workgroup each -- then the kept nodes take their Schur updates and become the next
// level's chain; ceil(log2 S) + 1 levels.  It is an exact Cholesky in the nested-dissection order (no pivoting, SPD), so the
// result equals the banded factorisation's up to rounding.
//
//   k_bcr_init    super-blocks D_s, couplings C_s = A[s+1][s] and right-hand sides from the band buffer [H | g] (+ damping)
//   k_bcr_elim    node p with active neighbours a < p < b: the (3M+1) x M panel [A_pp; A_ap; A_bp; y_p^T] one row per lane,
//                 factored in M register steps (pivot and multipliers through LDS): L_p, U_a = A_ap L^-T, U_b = A_bp L^-T,
//                 w = L^-1 y_p
//   k_bcr_update  kept node q: A_qq -= U U^T of its eliminated neighbours, y_q -= U w, and the new coupling
//                 A[b][a] = -U_b U_a^T of the node eliminated to its right
//   k_bcr_back    z_p = L^-T (w - U_a^T z_a - U_b^T z_b), last level first
// Every sum has a fixed order: two runs are bit-identical.
#include <vector>

#include "glio_device.h"

#define BCR_THREADS 256

struct BcrElim { int node, a, b, ea, eb; };          // active neighbours (-1: none), edges A[node][a] (= C[ea], rows node) and A[b][node] (= C[eb])
struct BcrKept { int node, pl, pr, enew, pad_; };    // eliminated neighbours to the left / right (-1: none), the new edge A[b(pr)][node] or -1

struct BcrDev {
    int M, sb, S, K, band, levels;
    double* D;            // [S][M*M] row-major, symmetric (both triangles kept)
    double* C;            // [2 S][M*M]  edge e: A[hi][lo], rows hi
    double* y;            // [S][M]
    double* z;            // [S][M]
    double* L;            // [S][M*M] lower
    double* Ua; double* Ub;   // [S][M*M] rows = unknowns of the neighbour
    double* w;            // [S][M]
    BcrElim* elim; BcrKept* kept;
    std::vector<int> h_elim_off, h_kept_off;       // per level [levels + 1]
    int* fail;
};

// ------------------------------------------------------------------------------------------------ init
__global__ __launch_bounds__(BCR_THREADS) void k_bcr_init(const double* __restrict__ Hg, const int K, const int band, const double lambda,
                                                          const double* __restrict__ dadd, const int M, const int sb, const int S, double* __restrict__ D,
                                                          double* __restrict__ C, double* __restrict__ y) {
    const int s = blockIdx.x, bw = band + 1;
    const long long nH = (long long)K * bw * 36;
    // H(ka, kb)[r][c] for |ka - kb| <= band from the upper band storage
    auto Hent = [&](const int ka, const int r, const int kb, const int c) -> double {
        if (ka >= K || kb >= K) return (ka == kb && r == c) ? 1.0 : 0.0;          // padding keyframes of the last super-block: identity
        const int d = kb - ka;
        if (d > band || d < -band) return 0.0;
        if (d >= 0) return Hg[((size_t)ka * bw + d) * 36 + r * 6 + c];
        return Hg[((size_t)kb * bw + (-d)) * 36 + c * 6 + r];
    };
    for (int e = threadIdx.x; e < M * M; e += BCR_THREADS) {
        const int i = e / M, j = e - M * i;
        const int ka = s * sb + i / 6, r = i % 6, kb = s * sb + j / 6, c = j % 6;
        double v = Hent(ka, r, kb, c);
        if (i == j && ka < K) v += dadd ? dadd[(size_t)ka * 6 + r] : lambda * v + 1e-12;
        D[(size_t)s * M * M + e] = v;
        if (s + 1 < S) {        // C_s = A[s+1][s]: rows in super-block s+1, columns in s
            const int kr = (s + 1) * sb + i / 6;
            C[(size_t)s * M * M + e] = (kr < K && kb < K) ? Hent(kr, r, kb, c) : 0.0;
        }
    }
    for (int i = threadIdx.x; i < M; i += BCR_THREADS) {
        const int k = s * sb + i / 6;
        y[(size_t)s * M + i] = k < K ? Hg[nH + (size_t)k * 6 + i % 6] : 0.0;
    }
}

// ------------------------------------------------------------------------------------------------ elimination
template <int M>
__global__ __launch_bounds__(BCR_THREADS) void k_bcr_elim(const BcrElim* __restrict__ tab, const double* __restrict__ D, const double* __restrict__ C,
                                                          const double* __restrict__ y, double* __restrict__ L, double* __restrict__ Ua,
                                                          double* __restrict__ Ub, double* __restrict__ w, int* fail) {
    __shared__ double colbuf[M];
    __shared__ double s_rd;
    __shared__ int s_bad;
    const BcrElim t = tab[blockIdx.x];
    const int tid = threadIdx.x;
    constexpr int ROWS = 3 * M + 1;
    static_assert(ROWS <= BCR_THREADS, "one panel row per thread");
    const size_t MM = (size_t)M * M;
    // ---- the panel row of this thread
    double a[M];
    const int row = tid;
    if (tid == 0) s_bad = 0;
    {
        const bool hasA = t.a >= 0, hasB = t.b >= 0;
#pragma unroll
        for (int c = 0; c < M; ++c) a[c] = 0.0;
        if (row < M) {
            const double* src = D + (size_t)t.node * MM + (size_t)row * M;
#pragma unroll
            for (int c = 0; c < M; ++c) a[c] = src[c];
        } else if (row < 2 * M) {             // row r of A[a][node] = column r of C[ea] (rows node, columns a)
            if (hasA) {
                const double* src = C + (size_t)t.ea * MM + (row - M);
#pragma unroll
                for (int c = 0; c < M; ++c) a[c] = src[(size_t)c * M];
            }
        } else if (row < 3 * M) {             // row r of A[b][node] = row r of C[eb]
            if (hasB) {
                const double* src = C + (size_t)t.eb * MM + (size_t)(row - 2 * M) * M;
#pragma unroll
                for (int c = 0; c < M; ++c) a[c] = src[c];
            }
        } else if (row == 3 * M) {
            const double* src = y + (size_t)t.node * M;
#pragma unroll
            for (int c = 0; c < M; ++c) a[c] = src[c];
        }
    }
    __syncthreads();
    // ---- M register steps; the pivot's reciprocal root and the multipliers L[c][j] go through LDS
#pragma unroll
    for (int j = 0; j < M; ++j) {
        if (row == j) {
            const double piv = a[j];
            if (!(piv > 0.0) || !isfinite(piv)) { s_bad = 1; s_rd = 1.0; } else s_rd = rsqrt(piv);
        }
        __syncthreads();
        const double rd = s_rd;
        a[j] = (row == j) ? a[j] * rd : a[j] * rd;          // L[row][j] (row j: sqrt(piv))
        if (row > j && row < M) colbuf[row] = a[j];
        __syncthreads();
        if (row > j) {
#pragma unroll
            for (int c = j + 1; c < M; ++c) a[c] -= a[j] * colbuf[c];
        }
    }
    if (tid == 0 && s_bad) atomicOr(fail, 1);
    // ---- store: L (lower, zeros above), U_a, U_b row-major, w
    if (row < M) {
        double* dst = L + (size_t)t.node * MM + (size_t)row * M;
#pragma unroll
        for (int c = 0; c < M; ++c) dst[c] = c <= row ? a[c] : 0.0;
    } else if (row < 2 * M) {
        double* dst = Ua + (size_t)t.node * MM + (size_t)(row - M) * M;
#pragma unroll
        for (int c = 0; c < M; ++c) dst[c] = a[c];
    } else if (row < 3 * M) {
        double* dst = Ub + (size_t)t.node * MM + (size_t)(row - 2 * M) * M;
#pragma unroll
        for (int c = 0; c < M; ++c) dst[c] = a[c];
    } else if (row == 3 * M) {
        double* dst = w + (size_t)t.node * M;
#pragma unroll
        for (int c = 0; c < M; ++c) dst[c] = a[c];
    }
}

// ------------------------------------------------------------------------------------------------ Schur updates of the kept nodes
template <int M>
__global__ __launch_bounds__(BCR_THREADS) void k_bcr_update(const BcrKept* __restrict__ tab, double* __restrict__ D, double* __restrict__ C,
                                                            double* __restrict__ y, const double* __restrict__ Ua, const double* __restrict__ Ub,
                                                            const double* __restrict__ w) {
    extern __shared__ double bcr_lds[];
    constexpr int LD = M + 1;
    double* X1 = bcr_lds;                 // U_b of the node eliminated to the left  (this node is its right neighbour)
    double* X2 = X1 + M * LD;             // U_a of the node eliminated to the right (this node is its left neighbour)
    double* X3 = X2 + M * LD;             // U_b of the node eliminated to the right (for the new coupling)
    double* w1 = X3 + M * LD;
    double* w2 = w1 + M;
    const BcrKept t = tab[blockIdx.x];
    const int tid = threadIdx.x;
    const size_t MM = (size_t)M * M;
    for (int e = tid; e < M * M; e += BCR_THREADS) {
        const int r = e / M, c = e - M * r;
        X1[r * LD + c] = t.pl >= 0 ? Ub[(size_t)t.pl * MM + e] : 0.0;
        X2[r * LD + c] = t.pr >= 0 ? Ua[(size_t)t.pr * MM + e] : 0.0;
        X3[r * LD + c] = (t.pr >= 0 && t.enew >= 0) ? Ub[(size_t)t.pr * MM + e] : 0.0;
    }
    for (int k = tid; k < M; k += BCR_THREADS) { w1[k] = t.pl >= 0 ? w[(size_t)t.pl * M + k] : 0.0; w2[k] = t.pr >= 0 ? w[(size_t)t.pr * M + k] : 0.0; }
    __syncthreads();
    double* Dq = D + (size_t)t.node * MM;
    for (int e = tid; e < M * M; e += BCR_THREADS) {
        const int r = e / M, c = e - M * r;
        double s1 = 0, s2 = 0, s3 = 0;
#pragma unroll 6
        for (int k = 0; k < M; ++k) { s1 += X1[r * LD + k] * X1[c * LD + k]; s2 += X2[r * LD + k] * X2[c * LD + k]; s3 += X3[r * LD + k] * X2[c * LD + k]; }
        Dq[e] = (Dq[e] - s1) - s2;
        if (t.enew >= 0) C[(size_t)t.enew * MM + e] = -s3;          // A[b][a] = - U_b U_a^T  (rows b, columns a = this node)
    }
    for (int r = tid; r < M; r += BCR_THREADS) {
        double s1 = 0, s2 = 0;
        for (int k = 0; k < M; ++k) { s1 += X1[r * LD + k] * w1[k]; s2 += X2[r * LD + k] * w2[k]; }
        y[(size_t)t.node * M + r] = (y[(size_t)t.node * M + r] - s1) - s2;
    }
}

// ------------------------------------------------------------------------------------------------ back substitution
template <int M>
__global__ __launch_bounds__(128) void k_bcr_back(const BcrElim* __restrict__ tab, const double* __restrict__ L, const double* __restrict__ Ua,
                                                  const double* __restrict__ Ub, const double* __restrict__ w, double* __restrict__ z) {
    extern __shared__ double bcr_lds[];
    constexpr int LD = M + 1;
    double* Ls = bcr_lds;            // [M][LD]
    double* tv = Ls + M * LD;        // [M]
    double* za = tv + M;
    double* zb = za + M;
    const BcrElim t = tab[blockIdx.x];
    const int tid = threadIdx.x;
    const size_t MM = (size_t)M * M;
    for (int e = tid; e < M * M; e += 128) { const int r = e / M, c = e - M * r; Ls[r * LD + c] = L[(size_t)t.node * MM + e]; }
    for (int k = tid; k < M; k += 128) { za[k] = t.a >= 0 ? z[(size_t)t.a * M + k] : 0.0; zb[k] = t.b >= 0 ? z[(size_t)t.b * M + k] : 0.0; }
    __syncthreads();
    if (tid < M) {
        double v = w[(size_t)t.node * M + tid];
        double s1 = 0, s2 = 0;
        if (t.a >= 0) { const double* U = Ua + (size_t)t.node * MM + tid; for (int r = 0; r < M; ++r) s1 += U[(size_t)r * M] * za[r]; }
        if (t.b >= 0) { const double* U = Ub + (size_t)t.node * MM + tid; for (int r = 0; r < M; ++r) s2 += U[(size_t)r * M] * zb[r]; }
        tv[tid] = (v - s1) - s2;
    }
    __syncthreads();
    // L^T z = t, last unknown first: z_r = t_r / L_rr, then t_k -= L[r][k] z_r for k < r
    for (int r = M - 1; r >= 0; --r) {
        if (tid == r) tv[r] = tv[r] / Ls[r * LD + r];
        __syncthreads();
        if (tid < r) tv[tid] -= Ls[r * LD + tid] * tv[r];
        __syncthreads();
    }
    if (tid < M) z[(size_t)t.node * M + tid] = tv[tid];
}

__global__ void k_bcr_delta(const double* __restrict__ z, const int K, const int M, const int sb, double* __restrict__ delta) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= K * 6) return;
    const int k = e / 6, r = e - 6 * k;
    delta[e] = -z[(size_t)(k / sb) * M + (k % sb) * 6 + r];
}

// ------------------------------------------------------------------------------------------------ host
void* glio_bcr_create(int K, int band) {
    if (band > 12) return nullptr;                         // wider bands keep the sequential banded kernels
    BcrDev* b = new BcrDev();
    b->K = K; b->band = band; b->M = band <= 6 ? 36 : 72; b->sb = b->M / 6; b->S = (K + b->sb - 1) / b->sb;
    const int S = b->S, M = b->M;
    const size_t MM = (size_t)M * M;
    auto A = [](void** p, size_t bytes) { return hipMalloc(p, bytes > 0 ? bytes : 16) == hipSuccess; };
    bool ok = A((void**)&b->D, S * MM * 8) && A((void**)&b->C, 2 * (size_t)S * MM * 8) && A((void**)&b->y, (size_t)S * M * 8) &&
              A((void**)&b->z, (size_t)S * M * 8) && A((void**)&b->L, S * MM * 8) && A((void**)&b->Ua, S * MM * 8) && A((void**)&b->Ub, S * MM * 8) &&
              A((void**)&b->w, (size_t)S * M * 8) && A((void**)&b->fail, 16);
    // the elimination schedule: active list (chain order), edge between consecutive active nodes
    std::vector<int> act(S), edge(S > 1 ? S - 1 : 0);
    for (int s = 0; s < S; ++s) act[s] = s;
    for (int s = 0; s + 1 < S; ++s) edge[s] = s;
    int next_edge = S - 1;
    std::vector<BcrElim> elim; std::vector<BcrKept> kept;
    b->h_elim_off.push_back(0); b->h_kept_off.push_back(0);
    while (!act.empty()) {
        const int n = (int)act.size();
        std::vector<int> nact, nedge;
        const int e0 = (int)elim.size();
        for (int p = 0; p < n; p += 2) {                   // even positions are eliminated
            BcrElim t;
            t.node = act[p]; t.a = p > 0 ? act[p - 1] : -1; t.b = p + 1 < n ? act[p + 1] : -1;
            t.ea = p > 0 ? edge[p - 1] : -1; t.eb = p + 1 < n ? edge[p] : -1;
            elim.push_back(t);
        }
        for (int p = 1; p < n; p += 2) {                   // odd positions are kept
            BcrKept t;
            t.node = act[p]; t.pl = act[p - 1]; t.pr = p + 1 < n ? act[p + 1] : -1; t.pad_ = 0;
            t.enew = (p + 2 < n) ? next_edge++ : -1;       // new coupling with the next kept node through the node eliminated between them
            kept.push_back(t);
            nact.push_back(act[p]);
            if (t.enew >= 0) nedge.push_back(t.enew);
        }
        (void)e0;
        b->h_elim_off.push_back((int)elim.size()); b->h_kept_off.push_back((int)kept.size());
        act.swap(nact); edge.swap(nedge);
    }
    b->levels = (int)b->h_elim_off.size() - 1;
    ok = ok && A((void**)&b->elim, elim.size() * sizeof(BcrElim)) && A((void**)&b->kept, (kept.size() + 1) * sizeof(BcrKept));
    if (!ok) { glio_set_error("batch solver: device allocation failed"); return nullptr; }
    hipMemcpy(b->elim, elim.data(), elim.size() * sizeof(BcrElim), hipMemcpyHostToDevice);
    if (!kept.empty()) hipMemcpy(b->kept, kept.data(), kept.size() * sizeof(BcrKept), hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_bcr_update<72>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((3 * 72 * 73 + 2 * 72) * 8));
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_bcr_back<72>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((72 * 73 + 3 * 72) * 8));
    (void)hipGetLastError();
    return b;
}

void glio_bcr_destroy(void* h) {
    BcrDev* b = static_cast<BcrDev*>(h);
    if (!b) return;
    void* ptrs[] = {b->D, b->C, b->y, b->z, b->L, b->Ua, b->Ub, b->w, b->elim, b->kept, b->fail};
    for (void* p : ptrs) if (p) hipFree(p);
    delete b;
}

// (H + lambda diag H) x = g, delta = -x; everything enqueued on `stream`.  *fail_dev (int) is set non-zero on a non-positive pivot.
template <int M>
static void bcr_run(BcrDev* b, const double* Hg, double lambda, const double* dadd, double* delta, hipStream_t stream) {
    const int S = b->S;
    hipMemsetAsync(b->fail, 0, 4, stream);
    hipLaunchKernelGGL(k_bcr_init, dim3(S), dim3(BCR_THREADS), 0, stream, Hg, b->K, b->band, lambda, dadd, M, b->sb, S, b->D, b->C, b->y);
    const size_t lds_up = (size_t)(3 * M * (M + 1) + 2 * M) * 8, lds_back = (size_t)(M * (M + 1) + 3 * M) * 8;
    for (int l = 0; l < b->levels; ++l) {
        const int ne = b->h_elim_off[l + 1] - b->h_elim_off[l], nk = b->h_kept_off[l + 1] - b->h_kept_off[l];
        if (ne > 0) hipLaunchKernelGGL((k_bcr_elim<M>), dim3(ne), dim3(BCR_THREADS), 0, stream, b->elim + b->h_elim_off[l], b->D, b->C, b->y, b->L, b->Ua, b->Ub, b->w, b->fail);
        if (nk > 0) hipLaunchKernelGGL((k_bcr_update<M>), dim3(nk), dim3(BCR_THREADS), lds_up, stream, b->kept + b->h_kept_off[l], b->D, b->C, b->y, b->Ua, b->Ub, b->w);
    }
    for (int l = b->levels - 1; l >= 0; --l) {
        const int ne = b->h_elim_off[l + 1] - b->h_elim_off[l];
        if (ne > 0) hipLaunchKernelGGL((k_bcr_back<M>), dim3(ne), dim3(128), lds_back, stream, b->elim + b->h_elim_off[l], b->L, b->Ua, b->Ub, b->w, b->z);
    }
    hipLaunchKernelGGL(k_bcr_delta, dim3((b->K * 6 + 255) / 256), dim3(256), 0, stream, b->z, b->K, M, b->sb, delta);
}
void glio_bcr_solve_shift(void* h, const double* Hg, double lambda, const double* dadd, double* delta, int** fail_dev, hipStream_t stream) {
    BcrDev* b = static_cast<BcrDev*>(h);
    if (b->M == 36) bcr_run<36>(b, Hg, lambda, dadd, delta, stream); else bcr_run<72>(b, Hg, lambda, dadd, delta, stream);
    *fail_dev = b->fail;
}
void glio_bcr_solve(void* h, const double* Hg, double lambda, double* delta, int** fail_dev, hipStream_t stream) { glio_bcr_solve_shift(h, Hg, lambda, nullptr, delta, fail_dev, stream); }
int glio_bcr_levels(void* h) { return static_cast<BcrDev*>(h)->levels; }
