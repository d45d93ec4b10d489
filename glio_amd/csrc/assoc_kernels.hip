// assoc_kernels.hip -- K1 + K2: scan-to-map surf correspondence search on gfx950.
//
// Replaces  kd_tree_surf_local_map->setInputCloud(surf_local_map_ds)      (GLIO/src/Estimator.cpp:2056)
//      and  Estimator::findCorrespondingSurfFeatures(idx, Q2, T2)         (GLIO/src/Estimator.cpp:3633-3708)
//
// K1  voxel hash build.  The reference accepts a point only if its 5th nearest neighbour is closer than
//     sqrt(kd_max_radius) (the yaml value is compared with the SQUARED distance, quirk Q1), so every
//     neighbour that can matter lies within r = sqrt(1.5) = 1.2247 m.  With a cell edge >= r the 27 cells
//     around the query contain all of them: the 27-cell scan returns the EXACT 5 nearest neighbours
//     whenever the gate passes, and whenever it fails the point is rejected by the reference too.
//     Build = open-addressing insert (64-bit key CAS) + per-cell count, range allocation with one
//     atomic per occupied cell, scatter.  Cell order in memory is arbitrary; results are not, because
//     candidates are ranked by (float distance, original map index).
// K2  one lane per scan point: double transform stored as float (transformPoint :1490-1498), float
//     L2 distances without fused multiply-add (FLANN L2_Simple<float>), 5x3 column-pivoted
//     Householder least squares in double (:3661), plane gate (:3667-3674), float pd / weight
//     (:3678-3679), then an order-preserving compaction into the slot's correspondence arrays
//     (vec_surf_cur_pts / vec_surf_normal / vec_surf_scores, :3682-3692).
//
// Roofline: gather-bound.  Algorithmic bytes per query = 16 (query) + 5*16 (true neighbours) + 40
// (output record) = 136 B; the 27-cell candidate scan is served by L2.
//
// The kernels of this file, in the order a call uses them:
//   K1   k_hash_clear / k_hash_insert / k_cell_alloc / k_scatter (+ the _multi forms: 64 keyframe hashes per launch): points ordered by (cell, octant)
//   bin  k_qbin_count / _alloc / _scatter: presort of an uploaded cloud into 5 m blocks (once per cloud); k_qbin_tile: per call, queries grouped by cell
//        inside tiles of 1024 presorted points -> units of <= 16 queries of one cell (a single launch row) or (tile, cell) segments (merged window);
//        k_gbin_alloc / k_gbin_scatter: the segments of ALL rows laid out cell-major, units of <= 64 queries
//   5-NN k_knn5_near<64> (merged window: near block + certificate, one unit per wavefront), k_knn5_rest (what it hands on: single queries by 16-lane
//        groups, sixteen-query groups by the 27-cell tiled search); k_knn5_tile (single rows: the 27-cell tiled search of round 4);
//        k_knn5 (debug mode 1: one 16-lane group per query, no binning), k_knn5_near<16> (debug mode 3: near block row by row)
//   fit  k_plane_fit<BATCH>: one lane per query, gates, record, per-workgroup kept counts; k_compact / k_scan_pairs + k_compact_pairs: order-preserving
//   batch association (glio_bassoc_*): resident per-keyframe hashes, pair-major records appended on the device
#include <cfloat>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <vector>

#include "glio_device.h"

// Bit-exact contract with the reference arithmetic: no fused multiply-add anywhere in this file (the
// float distances of FLANN and the double transform / plane fit of the reference are separate IEEE
// operations; a fused product changes the last bit and with it the ranking of exact-tie neighbours).
#pragma clang fp contract(off)

// q * v through Eigen's _transformVector, compiled here so that it obeys the pragma above
__device__ __forceinline__ void a_qrot(const double q[4], const double v[3], double o[3]) {
    double uv[3], uuv[3];
    uv[0] = q[2] * v[2] - q[3] * v[1]; uv[1] = q[3] * v[0] - q[1] * v[2]; uv[2] = q[1] * v[1] - q[2] * v[0];
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    uuv[0] = q[2] * uv[2] - q[3] * uv[1]; uuv[1] = q[3] * uv[0] - q[1] * uv[2]; uuv[2] = q[1] * uv[1] - q[2] * uv[0];
    o[0] = v[0] + q[0] * uv[0] + uuv[0];
    o[1] = v[1] + q[0] * uv[1] + uuv[1];
    o[2] = v[2] + q[0] * uv[2] + uuv[2];
}

struct AssocWork {
    float cell;                   // cell edge
    float inv_cell;
    int table_cap;                // power of two
    int cap_eff;                  // power of two <= table_cap sized for the current map (2 n points: load factor <= 0.5)
    unsigned long long* d_keys;   // [cap] EMPTY = ~0ull
    int4* d_ent;                  // [cap] {key lo, key hi, first point, points}: what the queries read, one 16 B load per probe
    unsigned* d_cnt8;             // [cap][8] points per (cell, octant)
    uint4* d_sub;                 // [cap] exclusive prefix of the octant counts, 8 x u16
    int* d_pt_slot;               // [max_map] 8 * table slot + octant of point i
    int* d_pt_rank;               // [max_map] its arrival rank inside that octant
    float4* d_map_raw;            // [max_map] upload staging
    const float4* last_src;       // the points the current map was built from (d_map_raw, or the local map's own output array)
    int* d_total;                 // [1]
    // per-query dense results (capacity = cap of a slot)
    float4* d_q_pt; float4* d_q_plane; double* d_q_score; int* d_q_flag; int* d_q_pos;
    int* d_nn;                    // optional [cap][5] neighbour indices (tests)
    int* d_nn5;                   // [cap][8] per query: positions of the 5-NN in the sorted map, the fifth distance (bits), 2 pad (searches -> k_plane_fit)
    int* d_bcount; int* d_boff;   // per-workgroup kept counts and their exclusive scan
    int* d_count_tmp;
    int* h_count;                 // pinned
    int* h_counts_win;            // pinned [GLIO_MAX_WINDOW]: counts of an asynchronous window association, picked up by glio_assoc_finish_pending
    hipEvent_t ev_counts;         // ... which waits for THIS point of the stream (the counts' copy), not for what the caller enqueued behind the searches since
    int* h_sel; int* d_sel; size_t sel_cap; hipEvent_t ev_sel; int sel_in_flight;      // glio_assoc_select_window: its pinned block [offsets | changed | indices], the device copy, the event of the last upload
    int counts_pending;
    double* d_win; double* h_win; // [W][7] poses + [W] counts of the window association (h_win pinned)
    struct KnnBinHost* kb;        // query binning buffers of the tiled search
    float4* d_ps;                 // [W][cap] presorted copies of the resident scans (w = index in the scan)
    float3 origin;
    double last_pose0[7];         // (q, t) slot 0 was last associated with: the timing hook replays THAT association
    int have_pose0;
};

#define KEY_EMPTY (~0ull)

__device__ __forceinline__ unsigned long long pack_key(int ix, int iy, int iz) {
    // 21 bits per axis, biased
    return ((unsigned long long)(unsigned)(ix + (1 << 20)) << 42) | ((unsigned long long)(unsigned)(iy + (1 << 20)) << 21) |
           (unsigned long long)(unsigned)(iz + (1 << 20));
}
__device__ __forceinline__ unsigned hash_key(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (unsigned)k;
}
__device__ __forceinline__ int cell_of(float v, float inv_cell) { return (int)floorf(v * inv_cell); }
// Home slot: the eight cells of a 2x2x2 block sit in eight consecutive 16 B entries (one 128 B line), so the 27 probes
// of a query touch at most 8 lines of the table instead of 27; collisions continue linearly.
__device__ __forceinline__ unsigned home_slot(int cx, int cy, int cz, int cap) {
    const unsigned h = hash_key(pack_key(cx >> 1, cy >> 1, cz >> 1));
    return ((h << 3) | (unsigned)((cx & 1) | ((cy & 1) << 1) | ((cz & 1) << 2))) & (unsigned)(cap - 1);
}

// K1 orders the points of a cell by OCTANT (which half of the cell along x, y, z): oct = (hx & 1) | (hy & 1) << 1 | (hz & 1) << 2 with
// h* = floor(v * 2 inv_cell) -- consistent with the cell (floor(2 y) >> 1 == floor(y), and 2 inv_cell scales exactly in binary floating point).
// The near-block search (k_knn5_near) stages only the octants that touch the query cell.  cnt8 [cap][8]: per-octant counts; the atomic that
// counts a point also hands it its rank inside its octant, so the scatter needs no second atomic.  sub [cap]: the exclusive prefix of the eight
// octant counts as 8 x u16 (all ones when the cell holds more than SUB_MAX points: such cells are searched by the 27-cell kernel only).
#define SUB_MAX 60000
__device__ __forceinline__ int oct_of(const float4 p, const float inv_cell) {
    const float ih = inv_cell * 2.0f;
    return ((int)floorf(p.x * ih) & 1) | (((int)floorf(p.y * ih) & 1) << 1) | (((int)floorf(p.z * ih) & 1) << 2);
}
// (with_ent = 0 when a range allocation follows: k_cell_alloc writes ent and sub of EVERY slot from the keys, clearing them first only moved 32 of 72 B per slot)
__device__ __forceinline__ void hash_clear_slot(const int i, const int cap, unsigned long long* keys, unsigned* cnt8, int* total, int4* ent, uint4* sub, const int with_ent) {
    if (i < cap) {
        keys[i] = KEY_EMPTY;
        reinterpret_cast<uint4*>(cnt8)[2 * (size_t)i] = make_uint4(0, 0, 0, 0); reinterpret_cast<uint4*>(cnt8)[2 * (size_t)i + 1] = make_uint4(0, 0, 0, 0);
        if (with_ent) { ent[i] = make_int4(-1, -1, 0, 0); sub[i] = make_uint4(0, 0, 0, 0); }
    }
    if (i == 0) *total = 0;
}
__device__ __forceinline__ void hash_insert_point(const float4 p, const int i, const float inv_cell, unsigned long long* keys, unsigned* cnt8, int* pt_slot, int* pt_rank,
                                                  const int cap) {
    const int cx = cell_of(p.x, inv_cell), cy = cell_of(p.y, inv_cell), cz = cell_of(p.z, inv_cell);
    const unsigned long long key = pack_key(cx, cy, cz);
    unsigned s = home_slot(cx, cy, cz, cap);
    for (;;) {
        unsigned long long prev = keys[s];                         // a stale EMPTY only costs the CAS; a key, once seen, stays
        if (prev != key) prev = atomicCAS(&keys[s], KEY_EMPTY, key);
        if (prev == KEY_EMPTY || prev == key) break;
        s = (s + 1) & (cap - 1);
    }
    const int so = (int)(s * 8u) + oct_of(p, inv_cell);
    pt_rank[i] = (int)atomicAdd(&cnt8[so], 1u);
    pt_slot[i] = so;
}
// The same per TILE of HI_THREADS consecutive points, aggregated in LDS first (round 5).  Every returning device-scope atomic is a round trip to the memory side and
// they pass at ~4 per ns whatever the launch: the per-point form spent 187 us inserting the 12 x 64 k points of one keyframe's batch association.  A tile groups
// its points by cell in an LDS table (LDS atomics), then issues ONE global key insertion per distinct cell and ONE counting atomic per distinct (cell, octant);
// the points take their ranks from the LDS counts.  Clouds that arrive spatially ordered (the presorted keyframe clouds, the voxel-sorted local map) put ~10-25
// points into a cell of a tile; a cloud in random order gains nothing and loses nothing.
#define HI_THREADS 512
#define HI_SLOTS 1024
struct HashInsertLds { unsigned long long key[HI_SLOTS]; unsigned cnt[HI_SLOTS][8]; unsigned base[HI_SLOTS][8]; int gslot[HI_SLOTS]; };
__device__ __forceinline__ void hash_insert_tile(HashInsertLds& L, const float4* __restrict__ pts, const int n, const int tile0, const float inv_cell, unsigned long long* keys,
                                                 unsigned* cnt8, int* pt_slot, int* pt_rank, const int cap) {
    const int tid = threadIdx.x;
    for (int q = tid; q < HI_SLOTS; q += HI_THREADS) {
        L.key[q] = KEY_EMPTY;
#pragma unroll
        for (int o = 0; o < 8; ++o) L.cnt[q][o] = 0;
    }
    __syncthreads();
    const int i = tile0 + tid;
    int ls = 0, oc = 0;
    unsigned lrank = 0;
    if (i < n) {
        const float4 p = pts[i];
        const unsigned long long key = pack_key(cell_of(p.x, inv_cell), cell_of(p.y, inv_cell), cell_of(p.z, inv_cell));
        unsigned q = hash_key(key) & (HI_SLOTS - 1);
        for (;;) {
            const unsigned long long prev = atomicCAS(&L.key[q], KEY_EMPTY, key);
            if (prev == KEY_EMPTY || prev == key) break;
            q = (q + 1) & (HI_SLOTS - 1);
        }
        ls = (int)q; oc = oct_of(p, inv_cell);
        lrank = atomicAdd(&L.cnt[q][oc], 1u);
    }
    __syncthreads();
    for (int q = tid; q < HI_SLOTS; q += HI_THREADS) {
        const unsigned long long key = L.key[q];
        if (key == KEY_EMPTY) continue;
        const int cx = (int)((key >> 42) & 0x1fffffu) - (1 << 20), cy = (int)((key >> 21) & 0x1fffffu) - (1 << 20), cz = (int)(key & 0x1fffffu) - (1 << 20);
        unsigned g = home_slot(cx, cy, cz, cap);
        for (;;) {
            unsigned long long prev = keys[g];                         // a stale EMPTY only costs the CAS; a key, once seen, stays
            if (prev != key) prev = atomicCAS(&keys[g], KEY_EMPTY, key);
            if (prev == KEY_EMPTY || prev == key) break;
            g = (g + 1) & (cap - 1);
        }
        L.gslot[q] = (int)g;
#pragma unroll
        for (int o = 0; o < 8; ++o) { const unsigned c = L.cnt[q][o]; if (c) L.base[q][o] = atomicAdd(&cnt8[g * 8u + o], c); }
    }
    __syncthreads();
    if (i < n) { pt_slot[i] = L.gslot[ls] * 8 + oc; pt_rank[i] = (int)(L.base[ls][oc] + lrank); }
}
// range allocation: one atomic per 1024-slot workgroup (block-wide exclusive scan of the cell counts; same-address atomics
// execute one after the other at the memory side, a wavefront-granular version spent 14 us on 2048 of them)
__device__ __forceinline__ void cell_alloc_slot(const int i, const int cap, const unsigned* __restrict__ cnt8, int* total, const unsigned long long* __restrict__ keys,
                                                int4* __restrict__ ent, uint4* __restrict__ sub, int* s_w, int* s_base) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (i < cap) {
        const uint4 a = reinterpret_cast<const uint4*>(cnt8)[2 * (size_t)i], b = reinterpret_cast<const uint4*>(cnt8)[2 * (size_t)i + 1];
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    }
    unsigned pre[8];
    unsigned cs = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { pre[k] = cs; cs += o[k]; }
    const int c = (int)cs;
    int incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int k = 0; k < 16; ++k) { const int v = s_w[k]; s_w[k] = t; t += v; }
        *s_base = t > 0 ? atomicAdd(total, t) : 0;
    }
    __syncthreads();
    const int st = *s_base + s_w[wv] + incl - c;
    if (i < cap) {
        const unsigned long long k = keys[i];
        ent[i] = make_int4((int)(unsigned)(k & 0xffffffffull), (int)(unsigned)(k >> 32), st, c);
        sub[i] = c > SUB_MAX ? make_uint4(~0u, ~0u, ~0u, ~0u)
                             : make_uint4(pre[0] | (pre[1] << 16), pre[2] | (pre[3] << 16), pre[4] | (pre[5] << 16), pre[6] | (pre[7] << 16));
    }
}
// (orig = the index the point has in the caller's cloud: i itself, or what a presorted cloud carries in .w)
__device__ __forceinline__ void scatter_point(float4 p, const int i, const int* __restrict__ pt_slot, const int* __restrict__ pt_rank, const unsigned* __restrict__ cnt8,
                                              const int4* __restrict__ ent, float4* __restrict__ sorted, const int orig) {
    const int so = pt_slot[i], s = so >> 3, oc = so & 7;
    const uint4 a = reinterpret_cast<const uint4*>(cnt8)[2 * (size_t)s], b = reinterpret_cast<const uint4*>(cnt8)[2 * (size_t)s + 1];
    const unsigned o[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    unsigned pre = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) pre += k < oc ? o[k] : 0u;
    p.w = __int_as_float(orig);       // original index rides in .w (map intensity is not used by the path)
    sorted[ent[s].z + (int)pre + pt_rank[i]] = p;
}

__global__ void k_hash_clear(unsigned long long* keys, unsigned* cnt8, int cap, int* total, int4* ent, uint4* sub, int with_ent) {
    hash_clear_slot(blockIdx.x * blockDim.x + threadIdx.x, cap, keys, cnt8, total, ent, sub, with_ent);
}
__global__ __launch_bounds__(HI_THREADS) void k_hash_insert(const float4* __restrict__ pts, int n, float inv_cell, unsigned long long* keys, unsigned* cnt8, int* pt_slot, int* pt_rank, int cap) {
    __shared__ HashInsertLds L;
    hash_insert_tile(L, pts, n, blockIdx.x * HI_THREADS, inv_cell, keys, cnt8, pt_slot, pt_rank, cap);
}
__global__ __launch_bounds__(1024) void k_cell_alloc(const unsigned* __restrict__ cnt8, int cap, int* total, const unsigned long long* __restrict__ keys,
                                                     int4* __restrict__ ent, uint4* __restrict__ sub) {
    __shared__ int s_w[16], s_base;
    cell_alloc_slot(blockIdx.x * 1024 + threadIdx.x, cap, cnt8, total, keys, ent, sub, s_w, &s_base);
}
__global__ void k_scatter(const float4* __restrict__ pts, int n, const int* __restrict__ pt_slot, const int* __restrict__ pt_rank, const unsigned* __restrict__ cnt8,
                          const int4* __restrict__ ent, float4* __restrict__ sorted) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) scatter_point(pts[i], i, pt_slot, pt_rank, cnt8, ent, sorted, i);
}

// ---- 5x3 column-pivoted Householder least squares, all indices compile-time (stays in registers)
__device__ __forceinline__ void plane_qr_solve(double A[5][3], double b[5], double x[3]) {
    int perm[3] = {0, 1, 2};
    double maxnorm = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double s = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) s += A[i][j] * A[i][j];
        s = sqrt(s);
        maxnorm = fmax(maxnorm, s);
    }
    const double thr_helper = (maxnorm * DBL_EPSILON) * (maxnorm * DBL_EPSILON) / 5.0;
    int nonzero = 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int best = k;
        double bestsq = -1.0;
#pragma unroll
        for (int j = k; j < 3; ++j) {
            double s = 0;
#pragma unroll
            for (int i = k; i < 5; ++i) s += A[i][j] * A[i][j];
            if (s > bestsq) { bestsq = s; best = j; }
        }
        if (nonzero == 3 && bestsq < thr_helper * (double)(5 - k)) nonzero = k;
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            if (best == j) {
#pragma unroll
                for (int i = 0; i < 5; ++i) { const double t = A[i][k]; A[i][k] = A[i][j]; A[i][j] = t; }
                const int t = perm[k]; perm[k] = perm[j]; perm[j] = t;
            }
        }
        double tail = 0;
#pragma unroll
        for (int i = k + 1; i < 5; ++i) tail += A[i][k] * A[i][k];
        const double c0 = A[k][k];
        double beta, tau, v[5];
        if (tail <= DBL_MIN) {
            tau = 0; beta = c0;
#pragma unroll
            for (int i = 0; i < 5; ++i) v[i] = 0;
        } else {
            beta = sqrt(c0 * c0 + tail);
            if (c0 >= 0) beta = -beta;
#pragma unroll
            for (int i = 0; i < 5; ++i) v[i] = (i > k) ? A[i][k] / (c0 - beta) : 0.0;
            tau = (beta - c0) / beta;
        }
        v[k] = 1.0;
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            double s = 0;
#pragma unroll
            for (int i = k; i < 5; ++i) s += v[i] * A[i][j];
            s *= tau;
#pragma unroll
            for (int i = k; i < 5; ++i) A[i][j] -= s * v[i];
        }
        {
            double s = 0;
#pragma unroll
            for (int i = k; i < 5; ++i) s += v[i] * b[i];
            s *= tau;
#pragma unroll
            for (int i = k; i < 5; ++i) b[i] -= s * v[i];
        }
        A[k][k] = beta;
#pragma unroll
        for (int i = k + 1; i < 5; ++i) A[i][k] = 0;
    }
    double y[3] = {0, 0, 0};
#pragma unroll
    for (int i = 2; i >= 0; --i) {
        if (i < nonzero) {
            double s = b[i];
#pragma unroll
            for (int j = i + 1; j < 3; ++j) if (j < nonzero) s -= A[i][j] * y[j];
            y[i] = s / A[i][i];
        }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int c = 0; c < 3; ++c) if (perm[j] == c) x[c] = y[j];
    }
}

// Query binning of the tiled neighbour search (k_qbin_* / k_knn5_tile): per launch row y (a scan, a window slot or a keyframe
// pair) the queries are grouped by the voxel-hash cell they fall in; a UNIT is up to TK_LANES queries of one cell.
#define NK_ROW_SHIFT 26    /* merged window: w of a grouped query = launch row << 26 | index in the scan */
struct KnnBin {
    // presort of one uploaded cloud (k_qbin_count / _alloc / _scatter, launch row 0 only):
    unsigned long long* keys; int* cnt; int* cstart;      // [capq] open-addressing table of the occupied cells (left EMPTY / 0 by k_qbin_alloc)
    int* qslot; int* qrank; float4* qtmp;                 // [cap] table slot, arrival rank and transformed point (w = its index) of point i
    // per-call grouping (k_qbin_tile) and search (k_knn5_tile), every launch row y:
    float4* qs;                                           // [Y][w_stride] queries grouped by cell (world xyz, w = index in the scan); the presort's output pointer
    int4* units;                                          // [Y][unit_stride] (first grouped position, queries, cell key lo, hi): the search probes from the record alone
    int* counters;                                        // [Y][4] grouped queries, units, units the near-block search handed on (zeroed again by k_plane_fit)
    int4* fails;                                          // [Y][unit_stride] (first grouped query, queries | mask << 16, cell key lo, hi): up to 16 queries of a cell the near-block
                                                          // search (k_knn5_near) hands on to the 27-cell tiled search
    int* failq;                                           // [Y][w_stride_q] grouped positions of single uncertified queries (searched by k_knn5<true>, one 16-lane group each)
    int use_fails;                                        // k_knn5_tile: 1 = run over `fails` instead of every unit
    unsigned long long* dbg;                              // nullptr, or [8] statistics of the near-block search (glio_debug_knn_stats)
    int capq, unit_stride;
    // merged-window mode (glob = 1): the queries of ALL launch rows grouped by cell together.  k_qbin_tile emits (tile, cell) segments and counts them per
    // cell in a global table, k_gbin_alloc lays the cells out and cuts them into units of <= 64 queries, k_gbin_scatter copies the segments into qs2; the
    // searches then run as ONE launch row over `units` / `fails` / `failq` used flat and the counter block gctr (same layout as a row of `counters`).
    int glob;
    float4* qs2;                                          // [Y * w_stride] queries grouped by cell across the rows (w = row << 26 | index in the scan)
    unsigned long long* gkeys; int* gcnt; int* gstart;    // [gcap] cell table (left EMPTY / 0 by k_gbin_alloc), first position of the cell in qs2
    int4* segs;                                           // [Y * w_stride] (table slot or -1, offset inside the cell or absolute position, position in qs incl. row offset, queries)
    int* gctr;                                            // [8] queries laid out, units, handed-on units, handed-on queries, segments, cells, orphan queries, -
    int gcap, qcap_total;
};

struct AssocArgs {
    KnnBin kb;
    double q[4], t[3];
    float inv_cell, cell;
    double kd_max_radius, weight_gate;      // doubles in the reference: float quantities are promoted for the comparison
    double surf_dist_thres, lidar_const;
    int n, table_cap, unit_scores;
    // window mode (glio_assoc_run_window): blockIdx.y = keyframe slot; pose and query count of the slot come from device
    // arrays, every per-query array is addressed at slot * stride.  nullptr / 0 for a single scan.
    const double* win_poses;                // [W][7] = q (w, x, y, z), t
    const int* win_counts;                  // [W]
    int q_stride, w_stride, b_stride;       // scan + correspondence arrays, dense work arrays, per-workgroup counts
    int w_stride_q;                         // row stride of kb.failq (= the binning capacity per row)
    int ring_base, ring_W;                  // window mode: slot k reads the scan row (ring_base + k) % ring_W
    // pair mode (glio_bassoc_run): blockIdx.y = pair inside the chunk; pair (ci, cj) queries the cloud of keyframe ci
    // (local frame, posed with poses[ci]) against the voxel hash of keyframe cj
    const struct FrameDesc* frames;         // [K]
    const int* pair_ci; const int* pair_cj; // [n_pairs]
    const double* poses;                    // [K][7] = t, q (w, x, y, z)
    int pair0;                              // first pair of this launch
    // pair mode, shared query binning: the pairs of a chunk that query the SAME keyframe cloud (the 2 search_range pairs of a keyframe call; 12 consecutive pairs
    // of batch.pair_list) have the same queries at the same pose -- they are grouped by cell ONCE.  A bin row = one distinct ci of the chunk:
    const int* pair_row;                    // [n_pairs] bin row (inside its chunk) of a pair, or nullptr: every launch row bins for itself
    const int* row_pair;                    // [n_pairs] at pair0 + r: a pair (absolute index) that queries bin row r's cloud
    int bin_pass;                           // 1: the launch is the binning of the chunk's bin rows (blockIdx.y = bin row), 0: blockIdx.y = pair of the chunk
    // pair mode, tables in the search frames' OWN frames (bassoc local mode): the voxel hash of keyframe cj was built once from its local cloud; its points are
    // re-posed every run (FrameDesc::sorted, global frame, in the table's order).  Only the GROUPING of the queries changes: a query's cell is the cell of its
    // position in cj's frame; distances, ranking, the gate and the fit work on the same global-frame floats as ever.
    int local_tables;
};
struct FrameDesc { const int4* ent; const uint4* sub; const float4* sorted; int n, cap_eff; };
struct AssocSlot { double q[4], t[3]; int n; size_t qoff, woff, boff; const int4* ent; const uint4* sub; const float4* map; size_t locoff; int table_cap;
                   int brow;         // brow: the launch row whose grouped queries / units this row searches (itself, unless pairs share their binning)
                   double lq[4], lt[3]; int local; };      // local tables: the search frame's pose (the queries are grouped by their cell in ITS frame)
__device__ __forceinline__ AssocSlot assoc_slot(const AssocArgs& a) {
    AssocSlot s;
    s.ent = nullptr; s.sub = nullptr; s.map = nullptr; s.locoff = 0; s.table_cap = a.table_cap;
    s.brow = blockIdx.y;
    s.local = 0;
    if (a.frames) {
        int p = a.pair0 + blockIdx.y;
        if (a.pair_row) {
            if (a.bin_pass) p = a.row_pair[p];          // the binning of bin row blockIdx.y: any pair that queries its cloud (same cloud, same pose)
            else s.brow = a.pair_row[p];
        }
        const int ci = a.pair_ci[p], cj = a.pair_cj[p];
        const double* P = a.poses + 7 * ci;
#pragma unroll
        for (int i = 0; i < 3; ++i) s.t[i] = P[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) s.q[i] = P[3 + i];
        const FrameDesc fj = a.frames[cj];
        s.n = a.frames[ci].n;
        s.qoff = (size_t)ci * a.q_stride; s.woff = (size_t)blockIdx.y * a.w_stride; s.boff = (size_t)blockIdx.y * a.b_stride;
        s.ent = fj.ent; s.sub = fj.sub; s.map = fj.sorted; s.table_cap = fj.cap_eff;
        s.locoff = (size_t)cj * a.q_stride;
        if (a.local_tables) {
            const double* Pj = a.poses + 7 * cj;
            s.local = 1;
#pragma unroll
            for (int i = 0; i < 3; ++i) s.lt[i] = Pj[i];
#pragma unroll
            for (int i = 0; i < 4; ++i) s.lq[i] = Pj[3 + i];
        }
    } else if (a.win_poses) {
        const int k = blockIdx.y;
        const double* P = a.win_poses + 7 * k;
#pragma unroll
        for (int i = 0; i < 4; ++i) s.q[i] = P[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) s.t[i] = P[4 + i];
        s.n = a.win_counts[k];
        s.qoff = (size_t)((k + a.ring_base) % a.ring_W) * a.q_stride; s.woff = (size_t)k * a.w_stride; s.boff = (size_t)k * a.b_stride;      // scans: ring rows
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) s.q[i] = a.q[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) s.t[i] = a.t[i];
        s.n = a.n; s.qoff = 0; s.woff = 0; s.boff = 0;
    }
    return s;
}

// 16 lanes per query (4 queries per wavefront, 16 per workgroup): the lanes of a group probe the 27 cells
// in two rounds, stride through the candidate points of each cell together, keep private top-5 lists and
// merge them with shuffles; lane 0 of the group fits the plane.  This turns ~30 dependent L2 round trips
// per query into a handful and gives the launch 16x the lanes to hide them with.
#ifndef AQ_LANES
#define AQ_LANES 16      /* 8 measured -7 % on the one-call window association but +23 % on the 32k-query pair launches */
#endif
#define AQ_PER_BLOCK (256 / AQ_LANES)

__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int src) {
    const int lo = __shfl((int)(v & 0xffffffffull), src, 64), hi = __shfl((int)(v >> 32), src, 64);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
    const int lo = __shfl_xor((int)(v & 0xffffffffull), m, 64), hi = __shfl_xor((int)(v >> 32), m, 64);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}

// K2 runs as two kernels.
// k_knn5: exact 5 nearest neighbours.  AQ_LANES lanes per query: the lanes of a group probe the 27 cells of the
//   neighbourhood in rounds, stride through the candidate points of each cell together, keep private top-5 lists (ranked
//   by float distance, then original map index) and merge them with shuffles.  Output per query: the positions of the
//   five neighbours in the cell-sorted map and the fifth distance.
// k_plane_fit<BATCH>: ONE lane per query (the 5x3 QR in double, the gates and the record need no cooperation, and inside
//   k_knn5 they would run with 1 lane in AQ_LANES active).  BATCH = true: findGlobalCorrespondingSurfFeaturesAdd_Batch
//   (Estimator.cpp:3808-3892): the "map" is ANOTHER keyframe's cloud in the global frame, `loc` the same cloud in that
//   keyframe's own frame: a second plane is fitted to the local coordinates of the same five neighbours and the record is
//   [unit local normal | local centroid] (6 doubles, o_nc) with score 2.5 w instead of the weighted global plane.
#define AQ_ROUNDS ((27 + AQ_LANES - 1) / AQ_LANES)
// Candidate lists are kept as ONE 64-bit key per entry, (float distance bits << 32) | original map index: for non-negative
// floats the unsigned order of the bits is the order of the values, so "smaller distance, then smaller index" is a single
// u64 compare and a list entry moves as one register pair (the kernel is VALU-issue bound: ~1600 vector instructions per
// wavefront of four queries, rocprofv3 SQ_INSTS_VALU).
__device__ __forceinline__ void knn5_insert_key(const unsigned long long key, const int m, unsigned long long bk[5], int bp[5]) {
    if (key < bk[4]) {
        bk[4] = key; bp[4] = m;
#pragma unroll
        for (int k = 4; k > 0; --k) {
            const bool sw = bk[k] < bk[k - 1];
            const unsigned long long tk = sw ? bk[k - 1] : bk[k];
            bk[k - 1] = sw ? bk[k] : bk[k - 1];
            bk[k] = tk;
            const int tp = sw ? bp[k - 1] : bp[k];
            bp[k - 1] = sw ? bp[k] : bp[k - 1];
            bp[k] = tp;
        }
    }
}
__device__ __forceinline__ void knn5_insert(const float px, const float py, const float pz, const float4 mp, const int m,
                                            unsigned long long bk[5], int bp[5]) {
    // plain operators, NOT the __f*_rn intrinsics: those are header functions compiled with
    // contraction allowed and fuse after inlining; here the file-scope pragma keeps them separate
    const float ex = px - mp.x, ey = py - mp.y, ez = pz - mp.z;
    float d = ex * ex;
    d = d + ey * ey;
    d = d + ez * ez;
    knn5_insert_key(((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(mp.w), m, bk, bp);
}

// LIST = false: query i of the launch row = point i of the scan (transformed here).  LIST = true: the queries the near-block search could not
// certify -- `failq` holds their positions in the grouped query array (world frame already, w = index in the scan); the workgroups stride over the list.
// (bx of gdx workgroups of blockDim.x threads work on the launch row blockIdx.y; s_tab: one 33-entry cell table per group of the workgroup)
template <bool LIST>
__device__ __forceinline__ void knn5_group_body(const AssocArgs& a, const float4* __restrict__ scan, const float4* __restrict__ map, const int4* __restrict__ ent,
                                                int* __restrict__ o_nn5, const int bx, const int gdx, int2 (*s_tab)[33]) {
    // per query: the 27 cells as (exclusive prefix of candidate counts, first point); entries 27..31 hold the total
    static_assert(AQ_ROUNDS * AQ_LANES <= 32 && AQ_LANES >= 2, "the cell table of a query has 32 entries");
    const int lane = threadIdx.x & 63, j = threadIdx.x & (AQ_LANES - 1), g = threadIdx.x / AQ_LANES;
    const int gbase = lane & ~(AQ_LANES - 1);                 // first lane of this group inside the wavefront
    const int per_block = blockDim.x / AQ_LANES;
    const AssocSlot sl = assoc_slot(a);
    const int n_q = LIST ? a.kb.counters[4 * blockIdx.y + 3] : sl.n;
    if (bx * per_block >= n_q) return;
    const bool glob = LIST && a.kb.glob != 0;
    scan += sl.qoff;
    if (!glob) { o_nn5 += 8 * sl.woff; }
    if (sl.ent) { ent = sl.ent; map = sl.map; }
    const int table_cap = sl.table_cap;
  for (int i0 = bx * per_block; i0 < n_q; i0 += gdx * per_block) {
    size_t i = (size_t)(i0 + g);
    const bool qlive = i0 + g < n_q;
    float px, py, pz;
    if (LIST) {
        const int fq = qlive ? a.kb.failq[(size_t)blockIdx.y * a.w_stride_q + i] : 0;
        const float4 qp = glob ? a.kb.qs2[fq] : a.kb.qs[(size_t)sl.brow * a.w_stride + fq];
        const int qw = __float_as_int(qp.w);
        px = qp.x; py = qp.y; pz = qp.z;
        i = glob ? (size_t)((unsigned)qw >> NK_ROW_SHIFT) * a.w_stride + (qw & ((1 << NK_ROW_SHIFT) - 1)) : (size_t)qw;
    } else {
        const float4 pl = scan[qlive ? i : 0];
        // transformPoint: double math, float store
        const double pin[3] = {(double)pl.x, (double)pl.y, (double)pl.z};
        double po[3];
        a_qrot(sl.q, pin, po);
        px = (float)(po[0] + sl.t[0]); py = (float)(po[1] + sl.t[1]); pz = (float)(po[2] + sl.t[2]);
    }
    const int cx = cell_of(px, a.inv_cell), cy = cell_of(py, a.inv_cell), cz = cell_of(pz, a.inv_cell);
    // ---- probe: lane j looks up cells j and j + AQ_LANES of the 27-neighbourhood (one 16 B entry per probe)
    int cs[AQ_ROUNDS], cc[AQ_ROUNDS];
#pragma unroll
    for (int h = 0; h < AQ_ROUNDS; ++h) {
        cs[h] = 0; cc[h] = 0;
        const int c = j + AQ_LANES * h;
        if (c < 27 && qlive) {
            const int dx = c % 3 - 1, dy = (c / 3) % 3 - 1, dz = c / 9 - 1;
            const unsigned long long key = pack_key(cx + dx, cy + dy, cz + dz);
            const int klo = (int)(unsigned)(key & 0xffffffffull), khi = (int)(unsigned)(key >> 32);
            unsigned s = home_slot(cx + dx, cy + dy, cz + dz, table_cap);
            for (;;) {
                const int4 e = ent[s];
                if (e.x == klo && e.y == khi) { cs[h] = e.z; cc[h] = e.w; break; }
                if ((e.x & e.y) == -1) break;                 // KEY_EMPTY
                s = (s + 1) & (table_cap - 1);
            }
        }
    }
    // ---- flatten: exclusive prefix of the counts in cell order (round 0 lanes 0..15, then round 1), so that the group
    // walks ONE list of `tot` candidates with all its lanes busy instead of 27 mostly empty cells one after the other
    int tot = 0;
#pragma unroll
    for (int h = 0; h < AQ_ROUNDS; ++h) {
        int incl = cc[h];
#pragma unroll
        for (int off = 1; off < AQ_LANES; off <<= 1) {
            const int t0 = __shfl_up(incl, off, AQ_LANES);
            if (j >= off) incl += t0;
        }
        s_tab[g][h * AQ_LANES + j] = make_int2(tot + incl - cc[h], cs[h]);      // cells >= 27 are empty: their prefix is the total
        tot += __shfl(incl, gbase + AQ_LANES - 1, 64);
    }
    for (int c = AQ_ROUNDS * AQ_LANES + j; c < 32; c += AQ_LANES) s_tab[g][c] = make_int2(tot, 0);   // padding of the search table
    GLIO_WAVE_LDS_SYNC();
    // ---- candidates: private top-5 per lane, ranked by (float distance, original index)
    unsigned long long bk[5] = {~0ull, ~0ull, ~0ull, ~0ull, ~0ull};
    int bp[5] = {-1, -1, -1, -1, -1};          // position in the sorted map
    const int2* tab = s_tab[g];
    auto locate = [&](const int f) {           // largest cell whose prefix is <= f holds candidate f
        int c = 0;
#pragma unroll
        for (int step = 16; step > 0; step >>= 1) if (tab[c + step].x <= f) c += step;
        const int2 e = tab[c];
        return e.y + (f - e.x);
    };
    for (int f = j; f < tot; f += 2 * AQ_LANES) {
        const int f2 = f + AQ_LANES;
        const bool two = f2 < tot;
        const int m1 = locate(f), m2 = two ? locate(f2) : m1;
        const float4 p1 = map[m1], p2 = map[m2];
        knn5_insert(px, py, pz, p1, m1, bk, bp);
        if (two) knn5_insert(px, py, pz, p2, m2, bk, bp);
    }
    // ---- merge the private lists of the group: five rounds of group-wide argmin on the key (distance bits, index)
    float md4 = FLT_MAX; int mp5[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const unsigned long long mykey = bk[0];
        unsigned long long kmin = mykey;
#pragma unroll
        for (int off = AQ_LANES / 2; off > 0; off >>= 1) {
            const unsigned long long o = shfl_xor_u64(kmin, off);
            kmin = o < kmin ? o : kmin;
        }
        const bool win = mykey == kmin && bp[0] >= 0;
        int pos = win ? bp[0] : -1;
#pragma unroll
        for (int off = AQ_LANES / 2; off > 0; off >>= 1) pos = max(pos, __shfl_xor(pos, off, 64));
        mp5[r] = pos;
        if (r == 4) md4 = pos >= 0 ? __uint_as_float((unsigned)(kmin >> 32)) : FLT_MAX;
        if (win) {                                   // pop the head of the winner's list
#pragma unroll
            for (int k = 0; k < 4; ++k) { bk[k] = bk[k + 1]; bp[k] = bp[k + 1]; }
            bk[4] = ~0ull; bp[4] = -1;
        }
    }
    if (j == 0 && qlive) {
        int4* rec = reinterpret_cast<int4*>(o_nn5 + 8 * i);         // one 32 B record per query: five positions, the fifth distance
        rec[0] = make_int4(mp5[0], mp5[1], mp5[2], mp5[3]); rec[1] = make_int4(mp5[4], __float_as_int(md4), 0, 0);
    }
    if (!LIST) break;
    GLIO_WAVE_LDS_SYNC();                       // (the next query of this group rewrites its cell table)
  }
}
__global__ __launch_bounds__(256) void k_knn5(const AssocArgs a, const float4* __restrict__ scan, const float4* __restrict__ map,
                                              const int4* __restrict__ ent, int* __restrict__ o_nn5) {
    __shared__ int2 s_tab[AQ_PER_BLOCK][33];
    knn5_group_body<false>(a, scan, map, ent, o_nn5, blockIdx.x, gridDim.x, s_tab);
}


// ------------------------------------------------------------------------------------------------
// K2, tiled.  k_knn5 above spends ~1600 vector instructions per wavefront of FOUR queries, most of them on per-query
// overhead (27 hash probes, prefix, candidate location, a five-round shuffle merge); it is VALU-issue bound.  Queries that
// fall in the same cell share their 27-cell candidate set, so the search runs per UNIT = up to 16 queries of one cell:
//   presort (once per uploaded cloud: k_qbin_count / _alloc / _scatter in the cloud's OWN frame, 5 m blocks) makes every run of
//           1024 consecutive points spatially compact, whatever order the caller's cloud came in;
//   k_qbin_tile   per 1024 consecutive presorted points: transformPoint, cell, grouping by cell in an LDS hash table (LDS
//                 atomics only; one global atomic per workgroup for its range of the unit list), grouped queries written out;
//   k_knn5_tile   per unit, 32 lanes (16 queries x 2 halves of the candidate list): the 27 cells are probed ONCE, their points
//                 staged in LDS (in pairs, structure of arrays), then every lane scans its half of the staged list for its own
//                 query: 6 packed VALU per candidate PAIR for the two float distances (v_pk_add / v_pk_mul / v_pk_fma) + 2 x 7 for
//                 the keys and a six-deep sorted insertion on 32-bit keys (v_and_or_b32, v_min_u32 + 5 v_med3_u32).
// The 32-bit key is (distance bits with the low 8 bits replaced by the LDS slot): unsigned order = distance order up to
// 2^-15 relative.  The exact ranking rule of the reference-equivalent search -- float distance (UNFUSED, as FLANN's L2 computes it),
// then original map index -- is restored afterwards: the six selected candidates are re-evaluated exactly (64-bit keys) and merged
// into the lane's running top five.  The selection distances use fused multiply-adds (a few ulp off the unfused value: at most one
// truncation bucket), so the selection provably contains the exact top five of what the lane scanned when the fifth exact distance
// lies at least TWO buckets under the sixth selected key (anything not selected has a selection bucket at least as large as that
// key's, hence an exact bucket at most one below it).  When it does not (near-ties inside 2^-14), the lane rescans with the unfused
// distance and exact keys.  The two halves are merged
// at the end.  The order of the queries inside a cell and of the units in memory is arbitrary; the results are not: every query
// is independent and written at its own index.
#define TK_Q 16
#define TK_LANES 32
#ifndef TK_CAP
#define TK_CAP 256         /* staged candidates per chunk: the 27 cells of a C2 unit hold ~160 (p90 193, max 330): one chunk for nearly every unit (round 4; 128 before) */
#endif
#define TK_SEL 6           /* selection by truncated key; exact when the fifth exact distance lies in a lower bucket than the SIXTH selected key */
#ifndef TK_THREADS
#define TK_THREADS 64      /* ONE wavefront per workgroup (two units): a workgroup of four made every unit wait for the slowest of eight, and 43 KB of LDS per workgroup sent a quarter of the workgroups into a second round (r04 per-workgroup stamps) */
#endif
#define TK_UNITS (TK_THREADS / TK_LANES)
#define TK_MASK ((unsigned)(TK_CAP - 1))
#define QT_THREADS 1024
#define QT_SLOTS 2048
#define PRESORT_CELL 5.0f

// (per tile of QC_THREADS points the blocks are grouped in LDS first -- one global key insertion and ONE returning counting atomic per distinct block of the
//  tile instead of two per point: returning device-scope atomics pass at ~4 per ns, this kernel took 20 us for 64 k points; a LiDAR scan in ring order puts
//  most of a tile into a handful of blocks)
#define QC_THREADS 512
#define QC_SLOTS 1024
__global__ __launch_bounds__(QC_THREADS) void k_qbin_count(const AssocArgs a, const float4* __restrict__ scan) {
    __shared__ unsigned long long s_key[QC_SLOTS];
    __shared__ int s_cnt[QC_SLOTS], s_base[QC_SLOTS], s_g[QC_SLOTS];
    const AssocSlot sl = assoc_slot(a);
    const int tid = threadIdx.x, i = blockIdx.x * QC_THREADS + tid;
    if ((int)(blockIdx.x * QC_THREADS) >= sl.n) return;
    for (int q = tid; q < QC_SLOTS; q += QC_THREADS) { s_key[q] = KEY_EMPTY; s_cnt[q] = 0; }
    __syncthreads();
    const int capq = a.kb.capq;
    unsigned long long* keys = a.kb.keys + (size_t)blockIdx.y * capq;
    float px = 0, py = 0, pz = 0;
    int ls = 0, lrank = 0;
    if (i < sl.n) {
        const float4 pl = scan[sl.qoff + i];
        const double pin[3] = {(double)pl.x, (double)pl.y, (double)pl.z};
        double po[3];
        a_qrot(sl.q, pin, po);
        px = (float)(po[0] + sl.t[0]); py = (float)(po[1] + sl.t[1]); pz = (float)(po[2] + sl.t[2]);
        const unsigned long long key = pack_key(cell_of(px, a.inv_cell), cell_of(py, a.inv_cell), cell_of(pz, a.inv_cell));
        unsigned q = hash_key(key) & (QC_SLOTS - 1);
        for (;;) {
            const unsigned long long prev = atomicCAS(&s_key[q], KEY_EMPTY, key);
            if (prev == KEY_EMPTY || prev == key) break;
            q = (q + 1) & (QC_SLOTS - 1);
        }
        ls = (int)q;
        lrank = atomicAdd(&s_cnt[q], 1);
    }
    __syncthreads();
    for (int q = tid; q < QC_SLOTS; q += QC_THREADS) {
        const unsigned long long key = s_key[q];
        if (key == KEY_EMPTY) continue;
        const int cx = (int)((key >> 42) & 0x1fffffu) - (1 << 20), cy = (int)((key >> 21) & 0x1fffffu) - (1 << 20), cz = (int)(key & 0x1fffffu) - (1 << 20);
        unsigned s = home_slot(cx, cy, cz, capq);
        for (;;) {
            unsigned long long prev = keys[s];                         // a stale EMPTY only costs the CAS; a key, once seen, stays
            if (prev != key) prev = atomicCAS(&keys[s], KEY_EMPTY, key);
            if (prev == KEY_EMPTY || prev == key) break;
            s = (s + 1) & (capq - 1);
        }
        s_g[q] = (int)s;
        s_base[q] = atomicAdd(&a.kb.cnt[(size_t)blockIdx.y * capq + s], s_cnt[q]);
    }
    __syncthreads();
    if (i < sl.n) {
        a.kb.qslot[sl.woff + i] = s_g[ls];
        a.kb.qrank[sl.woff + i] = s_base[ls] + lrank;
        a.kb.qtmp[sl.woff + i] = make_float4(px, py, pz, __int_as_float(i));
    }
}

__global__ __launch_bounds__(1024) void k_qbin_alloc(const AssocArgs a) {
    __shared__ int s_w[16], s_base;
    const int capq = a.kb.capq;
    const int s = blockIdx.x * 1024 + threadIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t row = (size_t)blockIdx.y * capq;
    const int c = s < capq ? a.kb.cnt[row + s] : 0;
    int ic = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(ic, off, 64);
        if (lane >= off) ic += v;
    }
    if (lane == 63) s_w[wv] = ic;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int k = 0; k < 16; ++k) { const int v = s_w[k]; s_w[k] = t; t += v; }
        s_base = t > 0 ? atomicAdd(&a.kb.counters[4 * blockIdx.y], t) : 0;
    }
    __syncthreads();
    if (c > 0) {
        a.kb.cstart[row + s] = s_base + s_w[wv] + ic - c;
        a.kb.keys[row + s] = KEY_EMPTY;
        a.kb.cnt[row + s] = 0;
    }
}

__global__ __launch_bounds__(256) void k_qbin_scatter(const AssocArgs a) {
    const AssocSlot sl = assoc_slot(a);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= sl.n) return;
    const int pos = a.kb.cstart[(size_t)blockIdx.y * a.kb.capq + a.kb.qslot[sl.woff + i]] + a.kb.qrank[sl.woff + i];
    a.kb.qs[sl.woff + pos] = a.kb.qtmp[sl.woff + i];
}


// ---- merged-window grouping (KnnBin::glob): the (tile, cell) segments k_qbin_tile found, counted per cell in a global table.
#define GB_UNIT 64         /* queries per unit of the merged window (k_knn5_near<64>) */
// One segment = `count` queries of cell `key` at qs[qs_pos ..].  The cell gets a table slot (bounded probe: 32 steps) and the segment an offset inside the
// cell's run; a segment that finds no slot (more distinct cells than the table was sized for) is laid out by itself from the END of qs2 downwards and
// becomes its own unit(s) -- slower, never wrong.  (No fill counter: every same-address atomic is executed one after the other at the memory side,
// ~5 ns each -- a counter bumped per cell and read per segment made this kernel 284 us instead of 25.)
__device__ __forceinline__ void gbin_segment(const KnnBin& kb, const int seg, const unsigned long long key, const int count, const int qs_pos) {
    const unsigned mask = (unsigned)(kb.gcap - 1);
    unsigned s = hash_key(key) & mask;
    int slot = -1;
    for (int tries = 0; tries < 32; ++tries) {
        unsigned long long prev = kb.gkeys[s];
        if (prev == KEY_EMPTY) {
            prev = atomicCAS(&kb.gkeys[s], KEY_EMPTY, key);
            if (prev == KEY_EMPTY) prev = key;
        }
        if (prev == key) { slot = (int)s; break; }
        s = (s + 1) & mask;
    }
    int off;
    if (slot >= 0) off = atomicAdd(&kb.gcnt[slot], count);
    else {
        off = kb.qcap_total - atomicAdd(&kb.gctr[6], count) - count;
        const int nu = (count + GB_UNIT - 1) / GB_UNIT, ub = atomicAdd(&kb.gctr[1], nu);
        for (int u = 0; u < nu; ++u)
            kb.units[ub + u] = make_int4(off + GB_UNIT * u, min(GB_UNIT, count - GB_UNIT * u), (int)(unsigned)(key & 0xffffffffull), (int)(unsigned)(key >> 32));
    }
    kb.segs[seg] = make_int4(slot, off, qs_pos, count);
}
// the cells in table order: first position in qs2 (exclusive scan of the counts), units of <= 64 queries; the table is left EMPTY / 0 for the next call
__global__ __launch_bounds__(1024) void k_gbin_alloc(const KnnBin kb) {
    __shared__ int s_wc[16], s_wu[16], s_pb, s_ub;
    const int s = blockIdx.x * 1024 + threadIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = s < kb.gcap ? kb.gcnt[s] : 0, nu = (c + GB_UNIT - 1) / GB_UNIT;
    int ic = c, iu = nu;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(ic, off, 64), u = __shfl_up(iu, off, 64);
        if (lane >= off) { ic += v; iu += u; }
    }
    if (lane == 63) { s_wc[wv] = ic; s_wu[wv] = iu; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int tc = 0, tu = 0;
        for (int k = 0; k < 16; ++k) { const int v = s_wc[k], u = s_wu[k]; s_wc[k] = tc; s_wu[k] = tu; tc += v; tu += u; }
        s_pb = tc > 0 ? atomicAdd(&kb.gctr[0], tc) : 0;
        s_ub = tu > 0 ? atomicAdd(&kb.gctr[1], tu) : 0;
    }
    __syncthreads();
    if (c > 0) {
        const int pb = s_pb + s_wc[wv] + ic - c, ub = s_ub + s_wu[wv] + iu - nu;
        const unsigned long long key = kb.gkeys[s];
        kb.gstart[s] = pb;
        for (int u = 0; u < nu; ++u)
            kb.units[ub + u] = make_int4(pb + GB_UNIT * u, min(GB_UNIT, c - GB_UNIT * u), (int)(unsigned)(key & 0xffffffffull), (int)(unsigned)(key >> 32));
        kb.gkeys[s] = KEY_EMPTY; kb.gcnt[s] = 0;
    }
}
// segment after segment (32 lanes each) from the tile-grouped rows into the cell-major array
__global__ __launch_bounds__(256) void k_gbin_scatter(const KnnBin kb) {
    const int n_segs = kb.gctr[4], l32 = threadIdx.x & 31;
    for (int i = blockIdx.x * 8 + (threadIdx.x >> 5); i < n_segs; i += gridDim.x * 8) {
        const int4 sg = kb.segs[i];
        const int dst = sg.x >= 0 ? kb.gstart[sg.x] + sg.y : sg.y;
        for (int l = l32; l < sg.w; l += 32) kb.qs2[dst + l] = kb.qs[sg.z + l];
    }
}

// per launch row y and tile of 1024 consecutive presorted points: units + grouped world-frame queries (w = original index)
__global__ __launch_bounds__(QT_THREADS) void k_qbin_tile(const AssocArgs a, const float4* __restrict__ ps) {
    __shared__ unsigned long long s_key[QT_SLOTS];
    __shared__ int s_cnt[QT_SLOTS];
    __shared__ int s_wc[16], s_wu[16], s_ubase;
    const AssocSlot sl = assoc_slot(a);
    const int tile0 = blockIdx.x * QT_THREADS, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tile0 >= sl.n) return;
    for (int s = tid; s < QT_SLOTS; s += QT_THREADS) { s_key[s] = KEY_EMPTY; s_cnt[s] = 0; }
    __syncthreads();
    const int i = tile0 + tid;
    const bool live = i < sl.n;
    float px = 0, py = 0, pz = 0, pw = 0;
    int slot = 0, rank = 0;
    if (live) {
        const float4 pl = ps[sl.qoff + i];
        const double pin[3] = {(double)pl.x, (double)pl.y, (double)pl.z};
        double po[3];
        a_qrot(sl.q, pin, po);
        px = (float)(po[0] + sl.t[0]); py = (float)(po[1] + sl.t[1]); pz = (float)(po[2] + sl.t[2]); pw = pl.w;
        float kx = px, ky = py, kz = pz;            // the position the query is grouped by
        if (sl.local) {
            // in the search frame's own frame (its table was built there): R_j^T (p_world - t_j).  Only the grouping depends on it -- a cell edge exceeds
            // the search radius by 2.5 cm, rounding here is ~1e-6 m
            const double dw[3] = {(po[0] + sl.t[0]) - sl.lt[0], (po[1] + sl.t[1]) - sl.lt[1], (po[2] + sl.t[2]) - sl.lt[2]};
            const double qc[4] = {sl.lq[0], -sl.lq[1], -sl.lq[2], -sl.lq[3]};
            double pl2[3];
            a_qrot(qc, dw, pl2);
            kx = (float)pl2[0]; ky = (float)pl2[1]; kz = (float)pl2[2];
        }
        const unsigned long long key = pack_key(cell_of(kx, a.inv_cell), cell_of(ky, a.inv_cell), cell_of(kz, a.inv_cell));
        unsigned s = hash_key(key) & (QT_SLOTS - 1);
        for (;;) {
            const unsigned long long prev = atomicCAS(&s_key[s], KEY_EMPTY, key);
            if (prev == KEY_EMPTY || prev == key) break;
            s = (s + 1) & (QT_SLOTS - 1);
        }
        rank = atomicAdd(&s_cnt[s], 1);
        slot = (int)s;
    }
    __syncthreads();
    // exclusive scan of (queries, units) over the 2048 slots, two slots per thread
    const bool glob = a.kb.glob != 0;
    const int c0 = s_cnt[2 * tid], c1 = s_cnt[2 * tid + 1];
    const int u0 = glob ? (c0 > 0) : (c0 + TK_Q - 1) / TK_Q, u1 = glob ? (c1 > 0) : (c1 + TK_Q - 1) / TK_Q;      // units of this row, or segments of the merged window
    int ic = c0 + c1, iu = u0 + u1;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(ic, off, 64), u = __shfl_up(iu, off, 64);
        if (lane >= off) { ic += v; iu += u; }
    }
    if (lane == 63) { s_wc[wv] = ic; s_wu[wv] = iu; }
    __syncthreads();
    if (tid == 0) {
        int tc = 0, tu = 0;
        for (int k = 0; k < 16; ++k) { const int v = s_wc[k], u = s_wu[k]; s_wc[k] = tc; s_wu[k] = tu; tc += v; tu += u; }
        s_ubase = glob ? atomicAdd(&a.kb.gctr[4], tu) : atomicAdd(&a.kb.counters[4 * blockIdx.y + 1], tu);
    }
    __syncthreads();
    const int st0 = s_wc[wv] + ic - (c0 + c1), ut0 = s_ubase + s_wu[wv] + iu - (u0 + u1);
    s_cnt[2 * tid] = st0; s_cnt[2 * tid + 1] = st0 + c0;
    int4* un = a.kb.units + (size_t)blockIdx.y * a.kb.unit_stride;
    const unsigned long long k0 = s_key[2 * tid], k1 = s_key[2 * tid + 1];
    if (glob) {
        if (c0 > 0) gbin_segment(a.kb, ut0, k0, c0, (int)(sl.woff + tile0 + st0));
        if (c1 > 0) gbin_segment(a.kb, ut0 + u0, k1, c1, (int)(sl.woff + tile0 + st0 + c0));
        pw = __int_as_float((int)((unsigned)blockIdx.y << NK_ROW_SHIFT) | __float_as_int(pw));      // the row travels with the query
    } else {
        for (int u = 0; u < u0; ++u) un[ut0 + u] = make_int4(tile0 + st0 + TK_Q * u, min(TK_Q, c0 - TK_Q * u), (int)(unsigned)(k0 & 0xffffffffull), (int)(unsigned)(k0 >> 32));
        for (int u = 0; u < u1; ++u) un[ut0 + u0 + u] = make_int4(tile0 + st0 + c0 + TK_Q * u, min(TK_Q, c1 - TK_Q * u), (int)(unsigned)(k1 & 0xffffffffull), (int)(unsigned)(k1 >> 32));
    }
    __syncthreads();
    if (live) a.kb.qs[sl.woff + tile0 + s_cnt[slot] + rank] = make_float4(px, py, pz, pw);
}

// sorted insertion of `key` into t[0] <= ... <= t[6]: new t[k] = med3(t[k-1], key, t[k]) (from the OLD values), new t[0] = min
__device__ __forceinline__ unsigned tk_med3(const unsigned a, const unsigned b, const unsigned c) {
    return max(min(a, b), min(max(a, b), c));
}
__device__ __forceinline__ void tk_insert(unsigned t[TK_SEL], const unsigned key) {
#pragma unroll
    for (int k = TK_SEL - 1; k > 0; --k) t[k] = tk_med3(t[k - 1], key, t[k]);
    t[0] = min(t[0], key);
}

#ifdef GLIO_DEV_STAMPS
__device__ long long g_knn_stamps[8];      // workgroup 0, wavefront 0 of the last launch: [0] probe + prefix, [1] staging, [2] scan, [3] exact re-ranking, [4] merge + store, [5] units
#define KN_T(var) const long long var = wall_clock64()
#define KN_ACC(k, t1, t0) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_knn_stamps[k] += (t1) - (t0); } while (0)
extern "C" int glio_debug_knn_stamps(long long* out8) { return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_knn_stamps), 64) == hipSuccess ? 0 : -2; }
// per workgroup (launch row 0) of the last launch: [start, end] device clock (100 MHz), candidates of its first unit, XCC / CU id
__device__ long long g_knn_wg[4096][12];
extern "C" int glio_debug_knn_wg(long long* out, int n) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_knn_wg), (size_t)n * 96) == hipSuccess ? 0 : -2; }
#else
#define KN_T(var) do { } while (0)
#define KN_ACC(k, t1, t0) do { } while (0)
#endif
typedef float tk_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void knn5_tile_body(const AssocArgs& a, const float4* __restrict__ map, const int4* __restrict__ ent,
                                               int* __restrict__ o_nn5, const int bx, const int gdx) {
    // Candidates are staged in PAIRS, structure-of-arrays inside the pair: [x0 x1 | y0 y1 | z0 z1], so that a lane ranks two candidates with
    // packed single-precision instructions (v_pk_add_f32 / v_pk_mul_f32: separate IEEE operations per half, no contraction -- the same bits as the
    // scalar form) -- the search is VALU-issue bound (r04 counters: 8.8 M wavefront-VALU per 64 k scan = half the kernel's duration on 1024 SIMDs).
    __shared__ tk_v2f s_xyz[TK_UNITS][TK_CAP / 2][3];
    __shared__ int s_idx[TK_UNITS][TK_CAP];          // original map index (the tie-break of the ranking)
    // (no array of map positions: the running five carry the candidate's ORDINAL in the unit's 27-cell list and the winners are located once at the
    //  end -- a kilobyte of LDS per unit buys four more wavefronts per CU, and the search is bound by latency x occupancy, r04 counters)
    __shared__ int2 s_tab[TK_UNITS][33];
    const int lane = threadIdx.x & 63, l32 = threadIdx.x & 31, j = threadIdx.x & (TK_Q - 1), h = (threadIdx.x >> 4) & 1, g = threadIdx.x / TK_LANES;
    const int gbase = lane & ~(TK_LANES - 1);
    const AssocSlot sl = assoc_slot(a);
    if (a.kb.glob == 0) {
        if (sl.n <= 0) return;
        o_nn5 += 8 * sl.woff;
    }
    if (sl.ent) { ent = sl.ent; map = sl.map; }
    const int table_cap = sl.table_cap;
    // use_fails: the launch ranks only what the near-block search (k_knn5_near) handed on -- entries (unit, mask of its queries); else every unit, every query
    const bool by_list = a.kb.use_fails != 0;
    // (units and grouped queries: the bin row's; the handed-on lists and their counts: this launch row's own)
    const int n_units = by_list ? a.kb.counters[4 * blockIdx.y + 2] : a.kb.counters[4 * sl.brow + 1];
    const int4* units = a.kb.units + (size_t)sl.brow * a.kb.unit_stride;
    const int4* fails = a.kb.fails + (size_t)blockIdx.y * a.kb.unit_stride;
    const bool glob = a.kb.glob != 0;
    const float4* qs = glob ? a.kb.qs2 : a.kb.qs + (size_t)sl.brow * a.w_stride;
#ifdef GLIO_DEV_STAMPS
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { for (int k = 0; k < 8; ++k) g_knn_stamps[k] = 0; }
    const long long wg_t0 = wall_clock64();
    int wg_tot = 0, wg_chunks = 0, wg_unsafe = 0, wg_probe = 0;
    long long wg_ph[5] = {0, 0, 0, 0, 0};
#define WG_PH(k, t1, t0) wg_ph[k] += (t1) - (t0)
#else
#define WG_PH(k, t1, t0) do { } while (0)
#endif
    for (int u0 = bx * TK_UNITS; u0 < n_units; u0 += gdx * TK_UNITS) {
        KN_T(tk0);
        const int uid = u0 + g;
        int fmask = 0xffff;
        const bool ulive = uid < n_units;
        int4 un = ulive ? (by_list ? fails[uid] : units[uid]) : make_int4(0, 0, 0, 0);
        if (by_list) { fmask = (int)((unsigned)un.y >> 16); un.y &= 0xffff; }
        const bool qlive = j < un.y && ((fmask >> j) & 1);
        const float4 qp = ulive ? qs[un.x + (qlive ? j : 0)] : make_float4(0, 0, 0, 0);
        const float px = qp.x, py = qp.y, pz = qp.z;
        const int qw = __float_as_int(qp.w);
        const size_t qi = glob ? (size_t)((unsigned)qw >> NK_ROW_SHIFT) * a.w_stride + (qw & ((1 << NK_ROW_SHIFT) - 1)) : (size_t)qw;
        // the unit's cell from its record (not from the query: the probe below then does not wait for the query's load)
        const unsigned long long ukey = ((unsigned long long)(unsigned)un.w << 32) | (unsigned)un.z;
        const int cx = (int)((ukey >> 42) & 0x1fffffu) - (1 << 20), cy = (int)((ukey >> 21) & 0x1fffffu) - (1 << 20), cz = (int)(ukey & 0x1fffffu) - (1 << 20);
        // ---- probe the 27 cells once per unit: lane c < 27 takes cell c
        int cs = 0, cc = 0;
        if (l32 < 27 && ulive) {
            const int dx = l32 % 3 - 1, dy = (l32 / 3) % 3 - 1, dz = l32 / 9 - 1;
            const unsigned long long key = pack_key(cx + dx, cy + dy, cz + dz);
            const int klo = (int)(unsigned)(key & 0xffffffffull), khi = (int)(unsigned)(key >> 32);
            unsigned s = home_slot(cx + dx, cy + dy, cz + dz, table_cap);
            for (;;) {
                const int4 e = ent[s];
#ifdef GLIO_DEV_STAMPS
                ++wg_probe;
#endif
                if (e.x == klo && e.y == khi) { cs = e.z; cc = e.w; break; }
                if ((e.x & e.y) == -1) break;
                s = (s + 1) & (table_cap - 1);
            }
        }
        int incl = cc;
#pragma unroll
        for (int off = 1; off < TK_LANES; off <<= 1) {
            const int t0 = __shfl_up(incl, off, TK_LANES);
            if (l32 >= off) incl += t0;
        }
        s_tab[g][l32] = make_int2(incl - cc, cs);
        const int tot = __shfl(incl, gbase + TK_LANES - 1, 64);
        if (l32 == 0) s_tab[g][32] = make_int2(tot, 0);
        const int tot_w = max(tot, __shfl_xor(tot, 32, 64));
        GLIO_WAVE_LDS_SYNC();
        const int2* tab = s_tab[g];
        auto locate = [&](const int f) {
            int c = 0;
#pragma unroll
            for (int step = 16; step > 0; step >>= 1) if (tab[c + step].x <= f) c += step;
            const int2 e = tab[c];
            return e.y + (f - e.x);
        };
        auto staged = [&](const int slot) {            // the staged candidate `slot` as the float4 the exact ranking takes (w = original index)
            const tk_v2f* pr = s_xyz[g][slot >> 1];
            const int e = slot & 1;
            return make_float4(e ? pr[0].y : pr[0].x, e ? pr[1].y : pr[1].x, e ? pr[2].y : pr[2].x, __int_as_float(s_idx[g][slot]));
        };
        unsigned long long bk[5] = {~0ull, ~0ull, ~0ull, ~0ull, ~0ull};
        int bp[5] = {-1, -1, -1, -1, -1};
        const tk_v2f P0 = {px, px}, P1 = {py, py}, P2 = {pz, pz};
        KN_T(tk1); KN_ACC(0, tk1, tk0); KN_ACC(5, 1, 0); WG_PH(0, tk1, tk0);
        for (int base = 0; base < tot_w; base += TK_CAP) {
            KN_T(tc0);
            const int n_c = min(max(tot - base, 0), TK_CAP);                       // staged candidates of this unit
            const int n_w = min(tot_w - base, TK_CAP), n_w8 = (n_w + 7) & ~7;       // scan length of the wavefront
            // ---- stage: beyond the unit's own list a far point (never selected).  All the loads of a lane are issued before the first LDS write.
            for (int r0 = 0; r0 < TK_CAP / TK_LANES; r0 += 4) {          // four loads of a lane in flight (eight cost 16 more VGPRs = one wavefront per SIMD less)
                if (TK_LANES * r0 >= n_w8) break;
                float4 pt[4]; int ps[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = l32 + TK_LANES * (r0 + r);
                    ps[r] = (f < n_c) ? locate(base + f) : -1;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) pt[r] = ps[r] >= 0 ? map[ps[r]] : make_float4(3e18f, 3e18f, 3e18f, 0.f);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = l32 + TK_LANES * (r0 + r);
                    if (f < n_w8) {
                        float* pr = reinterpret_cast<float*>(s_xyz[g][f >> 1]);
                        pr[f & 1] = pt[r].x; pr[2 + (f & 1)] = pt[r].y; pr[4 + (f & 1)] = pt[r].z;
                        s_idx[g][f] = __float_as_int(pt[r].w);
                    }
                }
            }
            GLIO_WAVE_LDS_SYNC();
            KN_T(tc1); KN_ACC(1, tc1, tc0); WG_PH(1, tc1, tc0);
            // ---- scan: every lane ranks its half of the staged PAIRS (pairs of parity h) for its own query
            unsigned tk[TK_SEL];
#pragma unroll
            for (int k = 0; k < TK_SEL; ++k) tk[k] = ~0u;
            for (int pp = h; pp < (n_w8 >> 1); pp += 4) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const tk_v2f* pr = s_xyz[g][pp + 2 * u];
#ifdef TK_SCALAR_DIST
                    const tk_v2f X = pr[0], Y = pr[1], Z = pr[2];
                    tk_v2f d;
                    { const float ex = px - X.x, ey = py - Y.x, ez = pz - Z.x; float t = ex * ex; t = t + ey * ey; t = t + ez * ez; d.x = t; }
                    { const float ex = px - X.y, ey = py - Y.y, ez = pz - Z.y; float t = ex * ex; t = t + ey * ey; t = t + ez * ez; d.y = t; }
#else
                    // SELECTION distances with fused multiply-adds (v_pk_fma_f32: 6 packed instructions per candidate pair instead of 8).  They differ
                    // from the reference's unfused float arithmetic by a few ulp (<= 3 x 2^-24 relative), which is far below a truncation bucket
                    // (2^-15): a candidate's exact bucket is at most ONE away from its bucket here.  The exact rule is restored below -- the selected
                    // six are re-ranked with the unfused distance, and the selection is trusted only when the fifth exact distance lies at least TWO
                    // buckets under the sixth selected key (`safe`); otherwise the rescan uses the unfused form.
                    const tk_v2f ex = P0 - pr[0], ey = P1 - pr[1], ez = P2 - pr[2];
                    tk_v2f d = ex * ex;
#ifdef TK_UNFUSED_SELECT
                    d = d + ey * ey;
                    d = d + ez * ez;
#else
                    d = __builtin_elementwise_fma(ey, ey, d);
                    d = __builtin_elementwise_fma(ez, ez, d);
#endif
#endif
                    const unsigned f0 = (unsigned)(2 * (pp + 2 * u));
                    tk_insert(tk, (__float_as_uint(d.x) & ~TK_MASK) | f0);
                    tk_insert(tk, (__float_as_uint(d.y) & ~TK_MASK) | (f0 + 1u));
                }
            }
            KN_T(tc2); KN_ACC(2, tc2, tc1); WG_PH(2, tc2, tc1);
            // ---- exact re-ranking of the selection, merged into the lane's running five
            bool all_in = false;
#pragma unroll
            for (int k = 0; k < TK_SEL; ++k) {
                const unsigned t = tk[k];
                const int slot = (int)(t & TK_MASK);
                const bool real = t != ~0u && slot < n_c;
                if (real) knn5_insert(px, py, pz, staged(slot), base + slot, bk, bp);
                if (k == TK_SEL - 1) all_in = !real;                                // fewer than TK_SEL candidates: all of them were merged
            }
#ifdef TK_UNFUSED_SELECT
            const bool safe = all_in || ((unsigned)(bk[4] >> 32) & ~TK_MASK) < (tk[TK_SEL - 1] & ~TK_MASK);
#else
            // an unselected candidate's selection bucket is >= the sixth key's, its exact bucket >= that minus one: the exact five are among the
            // selected when the fifth exact distance lies in a bucket below THAT (the compare cannot wrap: distances are finite floats)
            const bool safe = all_in || ((unsigned)(bk[4] >> 32) & ~TK_MASK) + (TK_MASK + 1u) < (tk[TK_SEL - 1] & ~TK_MASK);
#endif
            if (__any(qlive && !safe)) {
                // near-ties at the selection boundary: the candidates that can still belong to the exact five are those whose bucket is not above the
                // last selected key's (everything else has a larger truncated, hence a larger exact, distance than all six selected).  A second pass
                // over the lane's half with the cheap packed distances; the exact 64-bit insertion runs only for those few (round 3 re-ranked EVERY
                // candidate of the half exactly: 12 us for the wavefronts that hit it -- the tail of the kernel, r04 per-workgroup stamps).
#ifdef TK_UNFUSED_SELECT
                const unsigned edge = tk[TK_SEL - 1] & ~TK_MASK;
#else
                // (the six selected have EXACT buckets up to one above the sixth key's selection bucket, so that is where a member of the exact five
                //  can still sit; unfused distances from here on)
                const unsigned edge = (tk[TK_SEL - 1] & ~TK_MASK) + (TK_MASK + 1u);
#endif
                if (qlive && !safe) {
                    for (int pp = h; pp < ((n_c + 1) >> 1); pp += 2) {
                        const tk_v2f* pr = s_xyz[g][pp];
                        const tk_v2f ex = P0 - pr[0], ey = P1 - pr[1], ez = P2 - pr[2];
                        tk_v2f d = ex * ex;
                        d = d + ey * ey;
                        d = d + ez * ez;
                        // (the common iteration is the packed distance and one compare: the selection's own slots are told apart inside the rare branch)
                        if (min(__float_as_uint(d.x), __float_as_uint(d.y)) <= (edge | TK_MASK)) {
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const int f = 2 * pp + e;
                                bool skip = f >= n_c || (__float_as_uint(e ? d.y : d.x) & ~TK_MASK) > edge;
#pragma unroll
                                for (int k = 0; k < TK_SEL; ++k) skip = skip || (tk[k] != ~0u && (int)(tk[k] & TK_MASK) == f);
                                if (!skip) knn5_insert(px, py, pz, staged(f), base + f, bk, bp);
                            }
                        }
                    }
                }
            }
            GLIO_WAVE_LDS_SYNC();
            KN_T(tc3); KN_ACC(3, tc3, tc2); WG_PH(3, tc3, tc2);
#ifdef GLIO_DEV_STAMPS
            ++wg_chunks; if (__any(qlive && !safe)) ++wg_unsafe;
#endif
        }
        KN_T(tk2);
        // ---- the other half's five
        unsigned long long ok[5]; int op[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) { ok[k] = shfl_xor_u64(bk[k], 16); op[k] = __shfl_xor(bp[k], 16, 64); }
#pragma unroll
        for (int k = 0; k < 5; ++k) knn5_insert_key(ok[k], op[k], bk, bp);
        if (qlive && h == 0) {
            int ps5[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) ps5[k] = bp[k] >= 0 ? locate(bp[k]) : -1;      // ordinal in the 27-cell list -> position in the sorted map
            int4* rec = reinterpret_cast<int4*>(o_nn5 + 8 * qi);
            rec[0] = make_int4(ps5[0], ps5[1], ps5[2], ps5[3]);
            rec[1] = make_int4(ps5[4], __float_as_int(bp[4] >= 0 ? __uint_as_float((unsigned)(bk[4] >> 32)) : FLT_MAX), 0, 0);
        }
        GLIO_WAVE_LDS_SYNC();                        // (the next unit of this wavefront rewrites the cell table the winners were located with)
        KN_T(tk3); KN_ACC(4, tk3, tk2); WG_PH(4, tk3, tk2);
#ifdef GLIO_DEV_STAMPS
        wg_tot = tot_w;
#endif
    }
#ifdef GLIO_DEV_STAMPS
    if (blockIdx.y == 0 && blockIdx.x < 4096 && threadIdx.x == 0) {
        unsigned xcc = 0, hwid = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        long long* rec = g_knn_wg[blockIdx.x];
        rec[0] = wg_t0; rec[1] = wall_clock64(); rec[2] = wg_tot; rec[3] = ((long long)xcc << 32) | hwid;
        rec[4] = wg_chunks; rec[5] = wg_unsafe; rec[6] = wg_probe;
        for (int k = 0; k < 5; ++k) rec[7 + k] = wg_ph[k];
    }
#endif
}
__global__ __launch_bounds__(TK_THREADS) __attribute__((amdgpu_waves_per_eu(5))) void k_knn5_tile(const AssocArgs a, const float4* __restrict__ map, const int4* __restrict__ ent,
                                                   int* __restrict__ o_nn5) {
    knn5_tile_body(a, map, ent, o_nn5, blockIdx.x, gridDim.x);
}
// What the near-block search (k_knn5_near) handed on, in ONE launch (both lists are short and each of their wavefronts is a chain of dependent
// round trips: two launches one after the other cost two such chains): workgroups [0, gq) take the single queries (one 16-lane group per query,
// four per wavefront), the rest the units (27-cell tiled search, query mask).
__global__ __launch_bounds__(64) void k_knn5_rest(const AssocArgs a, const int gq, const float4* __restrict__ scan, const float4* __restrict__ map, const int4* __restrict__ ent,
                                                  int* __restrict__ o_nn5) {
    __shared__ int2 s_tab[64 / AQ_LANES][33];
    if ((int)blockIdx.x < gq) knn5_group_body<true>(a, scan, map, ent, o_nn5, blockIdx.x, gq, s_tab);
    else knn5_tile_body(a, map, ent, o_nn5, (int)blockIdx.x - gq, (int)gridDim.x - gq);
}


// ------------------------------------------------------------------------------------------------
// K2, near block first (round 5).  k_knn5_tile ranks every query against all the points of 27 cells of edge 1.25 m -- 130-160 candidates, 84 % of
// them outside the search radius (r04 counters: 10.6 M pairs per 64 k scan, 44 lane-operations each), although on a 0.4 m voxel map the fifth
// neighbour lies 0.4-0.6 m away.  K1 now orders the points of every cell by octant, so a unit (<= 16 queries of ONE cell) can stage just the
// NEAR BLOCK = the 4 x 4 x 4 half-cells centred on its cell (its own eight octants + the ring of octants that touch it: ~57 candidates on the C2
// map) and rank it with ONE lane per query (no half lists, no merge).  A result is accepted only with a certificate of exactness:
//     every map point outside the block is farther from the query than b = half + (distance of the query to the nearest face of its cell),
//     so when the fifth distance found inside the block is < (b - margin)^2 the five found ARE the global five (ranked by float distance, then
//     original index, exactly as k_knn5_tile ranks them), and sqd[4] < kd_max_radius is decided as well (b <= 1.25 m: b^2 can exceed 1.5 only
//     for a query in the very centre of its cell; the gate itself is applied downstream on the exact distance either way).
// Queries without a certificate (sparse surroundings: 0.1-3 % on the C2 stream), units whose block does not fit the staging area (dense maps) and
// cells above SUB_MAX points are handed on -- (unit, query mask) in `fails` -- to k_knn5_tile, which ranks them over the full 27 cells as before.
// Both kernels produce the same bytes for any query either can answer; tests/test_hip_assoc.py compares the three search modes.
// margin: the cell of a point is floor(fl(v * inv_cell)); the products are off by <= 2^-24 relative, so cell faces sit within 3 ulp(|v|) of where
// the arithmetic below puts them: 1e-6 |v| + 1e-5 m covers that four times over.
#define NK_Q 16            /* lanes that own the 16 rows of the block (three runs each) */
#ifndef NK_UNIT_FAILS
#define NK_UNIT_FAILS 6    /* uncertified queries (of 16) from which they are re-searched together by the tiled kernel */
#endif
__device__ __forceinline__ void nk_insert(unsigned t[6], const unsigned key) {
#pragma unroll
    for (int k = 5; k > 0; --k) t[k] = tk_med3(t[k - 1], key, t[k]);
    t[0] = min(t[0], key);
}
__device__ __forceinline__ void nk_cex(unsigned long long& x, unsigned long long& y) {
    const bool sw = y < x;
    const unsigned long long lo = sw ? y : x, hi = sw ? x : y;
    x = lo; y = hi;
}
__device__ __forceinline__ unsigned nk_pref(const uint4 sb, const int cc, const int o) {      // exclusive prefix of octant o (o = 8: the cell's count)
    const unsigned w = (o >> 1) == 0 ? sb.x : (o >> 1) == 1 ? sb.y : (o >> 1) == 2 ? sb.z : sb.w;
    return o >= 8 ? (unsigned)cc : ((w >> (16 * (o & 1))) & 0xffffu);
}
#ifdef GLIO_DEV_STAMPS
#define NK_STAMP_WAVES 65536
__device__ long long g_near_stamps[NK_STAMP_WAVES][8];   // per workgroup of the LAST launch (no atomics: same-address atomics queue at the memory side and distort what they measure):
                                                        // [0] probe, [1] runs + marks, [2] staging, [3] scan, [4] re-ranking .. hand-on, [5] iterations, [6] start, [7] end (100 MHz)
extern "C" int glio_debug_near_stamps(long long* out, int n_waves) {
    if (n_waves > NK_STAMP_WAVES) n_waves = NK_STAMP_WAVES;
    return (hipDeviceSynchronize() == hipSuccess && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_near_stamps), (size_t)n_waves * 64) == hipSuccess) ? 0 : -2;
}
#define NK_T(var) const long long var = wall_clock64()
#define NK_ACC(k, t1, t0) nk_ph[k] += (t1) - (t0)
#else
#define NK_T(var) do { } while (0)
#define NK_ACC(k, t1, t0) do { } while (0)
#endif
// UL = lanes (= queries) per unit: 16 -> four units of <= 16 queries per wavefront, 128 staging slots each (a scan or a keyframe pair alone: ~10-20
// queries fall in a cell); 64 -> ONE unit of <= 64 queries per wavefront, 256 slots (the merged window association: the queries of all W scans are
// grouped by cell together, ~350 per cell at C2, so the probes, the run table and the staging of a cell are paid once per 64 queries and every lane of
// the scan works).
#ifndef NK_ATTR
#define NK_ATTR
#endif
#ifndef NK_WPB
#define NK_WPB 4           /* independent wavefronts per workgroup (no workgroup barrier): one-wavefront workgroups left the SIMDs at 3.6 resident wavefronts of the 6 the registers allow */
#endif
template <int UL>
__global__ __launch_bounds__(64 * NK_WPB) NK_ATTR void k_knn5_near(const AssocArgs a, const float4* __restrict__ map, const int4* __restrict__ ent, const uint4* __restrict__ sub,
                                                  int* __restrict__ o_nn5) {
    constexpr int NU = 64 / UL, CAP = UL == 64 ? 256 : 128, PER = CAP / UL;
    constexpr unsigned MASK = (unsigned)(CAP - 1);
    static_assert(UL == 16 || UL == 64, "units of 16 or 64 lanes");
    // candidates staged in QUADS, structure of arrays: [x0 x1 x2 x3][y0..y3][z0..z3] -- three 16 B LDS reads (broadcast inside a unit) feed four
    // distance evaluations done two at a time with packed fp32 instructions
    __shared__ float4 s_q4w[NK_WPB][NU][CAP / 4][3];
    __shared__ int s_idxw[NK_WPB][NU][CAP];           // run << 24 | original map index (the tie-break); before the staging: the run marks
    __shared__ int s_deltaw[NK_WPB][NU][64];          // per run: map position - staged slot
    const int wv = threadIdx.x >> 6;
    float4 (*s_q4)[CAP / 4][3] = s_q4w[wv];
    int (*s_idx)[CAP] = s_idxw[wv];
    int (*s_delta)[64] = s_deltaw[wv];
    const int lane = threadIdx.x & 63, j = lane & (UL - 1), g = lane / UL, gbase = lane & ~(UL - 1);
    const bool glob = a.kb.glob != 0;
    const AssocSlot sl = assoc_slot(a);
    if (!glob && sl.n <= 0) return;
    if (!glob) { o_nn5 += 8 * sl.woff; }
    if (sl.ent) { ent = sl.ent; sub = sl.sub; map = sl.map; }
    const int table_cap = sl.table_cap;
    int* ctr = a.kb.counters + 4 * blockIdx.y;
    const int n_units = a.kb.counters[4 * sl.brow + 1];
    const int4* units = a.kb.units + (size_t)sl.brow * a.kb.unit_stride;
    int4* fails = a.kb.fails + (size_t)blockIdx.y * a.kb.unit_stride;
    int* failq = a.kb.failq + (size_t)blockIdx.y * a.w_stride_q;
    const float4* qs = glob ? a.kb.qs2 : a.kb.qs + (size_t)sl.brow * a.w_stride;
    const float cell = a.cell, half = 0.5f * a.cell;
#ifdef GLIO_DEV_STAMPS
    long long nk_ph[8] = {0, 0, 0, 0, 0, 0, wall_clock64(), 0};
#endif
    for (int u0 = (blockIdx.x * NK_WPB + wv) * NU; u0 < n_units; u0 += gridDim.x * NK_WPB * NU) {
        NK_T(nt0);
        const int uid = u0 + g;
        const bool ulive = uid < n_units;
        const int4 un = ulive ? units[uid] : make_int4(0, 0, 0, 0);
        const bool qlive = j < un.y;
        const float4 qp = ulive ? qs[un.x + (qlive ? j : 0)] : make_float4(0, 0, 0, 0);
        const float px = qp.x, py = qp.y, pz = qp.z;
        const int qw = __float_as_int(qp.w);
        const size_t qi = glob ? (size_t)((unsigned)qw >> NK_ROW_SHIFT) * a.w_stride + (qw & ((1 << NK_ROW_SHIFT) - 1)) : (size_t)qw;
        const unsigned long long ukey = ((unsigned long long)(unsigned)un.w << 32) | (unsigned)un.z;
        const int cx = (int)((ukey >> 42) & 0x1fffffu) - (1 << 20), cy = (int)((ukey >> 21) & 0x1fffffu) - (1 << 20), cz = (int)(ukey & 0x1fffffu) - (1 << 20);
        // ---- probe the 27 cells once per unit; the record of a cell = (first point, points, octant prefix)
        int* cellw = reinterpret_cast<int*>(&s_q4[g][0][0]);      // [27][6], dead before the staging writes the quads
#pragma unroll
        for (int hh = 0; hh < (27 + UL - 1) / UL; ++hh) {
            const int c = j + UL * hh;
            if (c < 27) {
                int cs = 0, cc = 0;
                uint4 sb = make_uint4(0, 0, 0, 0);
                if (ulive) {
                    const int dx = c % 3 - 1, dy = (c / 3) % 3 - 1, dz = c / 9 - 1;
                    const unsigned long long key = pack_key(cx + dx, cy + dy, cz + dz);
                    const int klo = (int)(unsigned)(key & 0xffffffffull), khi = (int)(unsigned)(key >> 32);
                    unsigned s = home_slot(cx + dx, cy + dy, cz + dz, table_cap);
                    for (;;) {
                        const int4 e = ent[s];
                        const uint4 sv = sub[s];                  // (issued with the entry: one round trip per probe step)
                        if (e.x == klo && e.y == khi) { cs = e.z; cc = e.w; sb = sv; break; }
                        if ((e.x & e.y) == -1) break;
                        s = (s + 1) & (table_cap - 1);
                    }
                }
                int2* cw = reinterpret_cast<int2*>(cellw + 6 * c);
                cw[0] = make_int2(cs, cc); cw[1] = make_int2((int)sb.x, (int)sb.y); cw[2] = make_int2((int)sb.z, (int)sb.w);
            }
        }
        {   // own staging slots: no run starts here yet
            int4* mk = reinterpret_cast<int4*>(&s_idx[g][PER * j]);
#pragma unroll
            for (int i = 0; i < PER / 4; ++i) mk[i] = make_int4(0, 0, 0, 0);
        }
        GLIO_WAVE_LDS_SYNC();
        NK_T(nt1); NK_ACC(0, nt1, nt0); NK_ACC(5, 1, 0);
        // ---- runs: lane j < 16 owns the row (y, z) = (j & 3, j >> 2) of the 4 x 4 x 4 half-cell block; along x the row crosses three cells:
        // [upper half of cell -1][both halves of cell 0: contiguous][lower half of cell +1] = three runs of the octant-ordered map
        int rst[3] = {0, 0, 0}, rcn[3] = {0, 0, 0};
        bool big = false;
        if (j < NK_Q) {
            const int hb = (j & 3) + 1, hc = (j >> 2) + 1;
            const int rowc = 3 * (hb >> 1) + 9 * (hc >> 1), ob = ((hb & 1) << 1) | ((hc & 1) << 2);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int2* cw = reinterpret_cast<const int2*>(cellw + 6 * (rowc + k));
                const int2 r0 = cw[0], r1 = cw[1], r2 = cw[2];
                const uint4 sb = make_uint4((unsigned)r1.x, (unsigned)r1.y, (unsigned)r2.x, (unsigned)r2.y);
                big = big || (sb.x == ~0u);
                const int o0 = ob | (k == 0 ? 1 : 0), o1 = ob + (k == 2 ? 1 : 2);
                const unsigned p0 = nk_pref(sb, r0.y, o0), p1 = nk_pref(sb, r0.y, o1);
                rst[k] = r0.x + (int)p0; rcn[k] = (int)(p1 - p0);
            }
        }
        const int mine = rcn[0] + rcn[1] + rcn[2];
        int incl = mine;
#pragma unroll
        for (int off = 1; off < NK_Q; off <<= 1) {
            const int t0 = __shfl_up(incl, off, NK_Q);
            if ((j & (NK_Q - 1)) >= off) incl += t0;
        }
        const int tot = __shfl(incl, gbase + NK_Q - 1, 64);
        const unsigned long long bigb = __ballot(big);
        const bool over = tot > CAP || ((bigb >> gbase) & (UL == 64 ? ~0ull : 0xffffull)) != 0;
        const int tot_e = (over || !ulive) ? 0 : tot;
        if (tot_e > 0 && j < NK_Q) {
            int pf = incl - mine;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (rcn[k] > 0) s_idx[g][pf] = 3 * j + k + 1;
                s_delta[g][3 * j + k] = rst[k] - pf;
                pf += rcn[k];
            }
        }
        GLIO_WAVE_LDS_SYNC();
        NK_T(nt2); NK_ACC(1, nt2, nt1);
        // ---- stage: lane j fills the slots [PER j, PER j + PER): the run of a slot = the last mark at or before it (max-scan; marks grow with the slot)
        {
            int rn[PER];
            {
                const int4* mk = reinterpret_cast<const int4*>(&s_idx[g][PER * j]);
                int run = 0;
#pragma unroll
                for (int i = 0; i < PER / 4; ++i) {
                    const int4 m = mk[i];
                    run = max(run, m.x); rn[4 * i] = run; run = max(run, m.y); rn[4 * i + 1] = run;
                    run = max(run, m.z); rn[4 * i + 2] = run; run = max(run, m.w); rn[4 * i + 3] = run;
                }
                int inc = run;
#pragma unroll
                for (int off = 1; off < UL; off <<= 1) {
                    const int t0 = __shfl_up(inc, off, UL);
                    if (j >= off) inc = max(inc, t0);
                }
                int carry = __shfl_up(inc, 1, UL);
                if (j == 0) carry = 0;
#pragma unroll
                for (int i = 0; i < PER; ++i) rn[i] = max(rn[i], carry) - 1;
            }
            float4 pt[PER];
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int f = PER * j + i;
                pt[i] = make_float4(3e18f, 3e18f, 3e18f, 0.f);
                if (f < tot_e) pt[i] = map[f + s_delta[g][rn[i]]];
            }
#pragma unroll
            for (int i = 0; i < PER / 4; ++i) {
                float4* qd = s_q4[g][(PER / 4) * j + i];
                qd[0] = make_float4(pt[4 * i].x, pt[4 * i + 1].x, pt[4 * i + 2].x, pt[4 * i + 3].x);
                qd[1] = make_float4(pt[4 * i].y, pt[4 * i + 1].y, pt[4 * i + 2].y, pt[4 * i + 3].y);
                qd[2] = make_float4(pt[4 * i].z, pt[4 * i + 1].z, pt[4 * i + 2].z, pt[4 * i + 3].z);
                reinterpret_cast<int4*>(&s_idx[g][PER * j])[i] =
                    make_int4((rn[4 * i] << 24) | (__float_as_int(pt[4 * i].w) & 0xffffff), (rn[4 * i + 1] << 24) | (__float_as_int(pt[4 * i + 1].w) & 0xffffff),
                              (rn[4 * i + 2] << 24) | (__float_as_int(pt[4 * i + 2].w) & 0xffffff), (rn[4 * i + 3] << 24) | (__float_as_int(pt[4 * i + 3].w) & 0xffffff));
            }
        }
        GLIO_WAVE_LDS_SYNC();
        NK_T(nt3); NK_ACC(2, nt3, nt2);
        // ---- scan: every lane ranks the staged block for its own query; selection by truncated key (distance bits, low bits = slot) with fused distances
        int tot_w = tot_e;
#pragma unroll
        for (int off = UL; off < 64; off <<= 1) tot_w = max(tot_w, __shfl_xor(tot_w, off, 64));
        const int n_q = (tot_w + 3) >> 2;
        unsigned tk[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) tk[k] = ~0u;
        const tk_v2f P0 = {px, px}, P1 = {py, py}, P2 = {pz, pz};
        // (the next quad is read while this one is ranked: the three 16 B LDS reads of an iteration otherwise sit in front of its 47 vector instructions)
        float4 Xn = s_q4[g][0][0], Yn = s_q4[g][0][1], Zn = s_q4[g][0][2];
        for (int qd = 0; qd < n_q; ++qd) {
            const float4 X = Xn, Y = Yn, Z = Zn;
            { const int nx = min(qd + 1, CAP / 4 - 1); Xn = s_q4[g][nx][0]; Yn = s_q4[g][nx][1]; Zn = s_q4[g][nx][2]; }
            const tk_v2f xa = {X.x, X.y}, xb = {X.z, X.w}, ya = {Y.x, Y.y}, yb = {Y.z, Y.w}, za = {Z.x, Z.y}, zb = {Z.z, Z.w};
            const tk_v2f exa = P0 - xa, eya = P1 - ya, eza = P2 - za, exb = P0 - xb, eyb = P1 - yb, ezb = P2 - zb;
            tk_v2f da = exa * exa, db = exb * exb;
            da = __builtin_elementwise_fma(eya, eya, da); db = __builtin_elementwise_fma(eyb, eyb, db);
            da = __builtin_elementwise_fma(eza, eza, da); db = __builtin_elementwise_fma(ezb, ezb, db);
            const unsigned f0 = (unsigned)(4 * qd);
            nk_insert(tk, (__float_as_uint(da.x) & ~MASK) | f0);
            nk_insert(tk, (__float_as_uint(da.y) & ~MASK) | (f0 + 1u));
            nk_insert(tk, (__float_as_uint(db.x) & ~MASK) | (f0 + 2u));
            nk_insert(tk, (__float_as_uint(db.y) & ~MASK) | (f0 + 3u));
        }
        NK_T(nt4); NK_ACC(3, nt4, nt3);
        // ---- exact re-ranking of the six selected: unfused float distance (FLANN's L2), then original index; the slot rides in the low byte (indices
        // are unique, so it never decides)
        unsigned long long K[6];
        const float* flat = reinterpret_cast<const float*>(&s_q4[g][0][0]);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const unsigned t = tk[k];
            const int slot = (int)(t & MASK);
            const bool real = t != ~0u && slot < tot_e;
            const int fo = 12 * (slot >> 2) + (slot & 3);
            const float ex = px - flat[fo], ey = py - flat[fo + 4], ez = pz - flat[fo + 8];
            float d = ex * ex;
            d = d + ey * ey;
            d = d + ez * ez;
            const unsigned lo = ((unsigned)(s_idx[g][slot] & 0xffffff) << 8) | (unsigned)slot;
            K[k] = real ? (((unsigned long long)__float_as_uint(d) << 32) | lo) : ~0ull;
        }
        nk_cex(K[0], K[5]); nk_cex(K[1], K[3]); nk_cex(K[2], K[4]);
        nk_cex(K[1], K[2]); nk_cex(K[3], K[4]);
        nk_cex(K[0], K[3]); nk_cex(K[2], K[5]);
        nk_cex(K[0], K[1]); nk_cex(K[2], K[3]); nk_cex(K[4], K[5]);
        nk_cex(K[1], K[2]); nk_cex(K[3], K[4]);
        const bool has5 = K[4] != ~0ull;
        // the selection distances are fused (a few ulp off): a candidate's exact bucket is at most ONE away from its selection bucket, so nothing
        // unselected can precede the fifth exact key when that lies at least two buckets under the sixth selected key
        const bool all_in = tk[5] == ~0u || (int)(tk[5] & MASK) >= tot_e;
        const bool safe = all_in || ((unsigned)(K[4] >> 32) & ~MASK) + (MASK + 1u) < (tk[5] & ~MASK);
        // ---- certificate: distance from the query to the boundary of the staged block
        const float ox = (float)cx * cell, oy = (float)cy * cell, oz = (float)cz * cell;
        float b = fminf(px - ox, (ox + cell) - px);
        b = fminf(b, fminf(py - oy, (oy + cell) - py));
        b = fminf(b, fminf(pz - oz, (oz + cell) - pz));
        const float bm = (b + half) - (1e-5f + 1e-6f * fmaxf(fabsf(px), fmaxf(fabsf(py), fabsf(pz))));
        const float d5 = __uint_as_float((unsigned)(K[4] >> 32));
        const bool cert = qlive && has5 && safe && bm > 0.f && d5 < bm * bm * 0.99999f;
        if (cert) {
            int ps5[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int slot = (int)((unsigned)K[k] & MASK);
                ps5[k] = slot + s_delta[g][s_idx[g][slot] >> 24];
            }
            int4* rec = reinterpret_cast<int4*>(o_nn5 + 8 * qi);            // one 32 B record per query (two 16 B stores into one line; six dword stores into three before)
            rec[0] = make_int4(ps5[0], ps5[1], ps5[2], ps5[3]); rec[1] = make_int4(ps5[4], __float_as_int(d5), 0, 0);
        }
        // ---- hand the rest on, one atomic per wavefront and list.  Per 16 queries: all of them when the block did not fit, or many uncertified -> the
        // 27-cell tiled search as (first grouped query, queries | mask << 16, cell key); single queries -> the one-group-per-query search (a tiled unit
        // costs the same for 1 or 16 queries)
        const unsigned long long fb = __ballot(qlive && !cert);
        const int m16 = (int)((fb >> (lane & ~15)) & 0xffffull);
        const bool by_unit = over || __popc(m16) >= NK_UNIT_FAILS;
        const unsigned long long ub = __ballot((lane & 15) == 0 && m16 != 0 && by_unit);
        if (ub) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&ctr[2], __popcll(ub));
            base = __shfl(base, 0, 64);
            if ((lane & 15) == 0 && m16 != 0 && by_unit)
                fails[base + __popcll(ub & ((1ull << lane) - 1ull))] = make_int4(un.x + j, min(16, un.y - j) | (m16 << 16), un.z, un.w);
        }
        if (a.kb.dbg) {
            const unsigned long long lb = __ballot(qlive), cb = __ballot(cert), ob = __ballot(j == 0 && ulive && over), ul = __ballot(j == 0 && ulive);
            int st = j == 0 ? tot_e : 0;
#pragma unroll
            for (int off = UL; off < 64; off <<= 1) st += __shfl_xor(st, off, 64);
            if (lane == 0) {
                atomicAdd(&a.kb.dbg[0], (unsigned long long)__popcll(ul)); atomicAdd(&a.kb.dbg[1], (unsigned long long)__popcll(ob));
                atomicAdd(&a.kb.dbg[2], (unsigned long long)__popcll(ub)); atomicAdd(&a.kb.dbg[3], (unsigned long long)__popcll(fb));
                atomicAdd(&a.kb.dbg[4], (unsigned long long)__popcll(lb)); atomicAdd(&a.kb.dbg[5], (unsigned long long)__popcll(cb));
                atomicAdd(&a.kb.dbg[6], (unsigned long long)st); atomicAdd(&a.kb.dbg[7], (unsigned long long)n_q);
            }
        }
        const unsigned long long qb = __ballot(qlive && !cert && !by_unit);
        if (qb) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&ctr[3], __popcll(qb));
            base = __shfl(base, 0, 64);
            if (qlive && !cert && !by_unit) failq[base + __popcll(qb & ((1ull << lane) - 1ull))] = un.x + j;
        }
        GLIO_WAVE_LDS_SYNC();
        NK_T(nt5); NK_ACC(4, nt5, nt4);
    }
#ifdef GLIO_DEV_STAMPS
    {
        const unsigned w = (blockIdx.y * gridDim.x + blockIdx.x) * NK_WPB + wv;
        nk_ph[7] = wall_clock64();
        if (lane == 0 && w < NK_STAMP_WAVES) { for (int k = 0; k < 8; ++k) g_near_stamps[w][k] = nk_ph[k]; }
    }
#endif
}

#define PF_BLOCK 256
template <bool BATCH>
__global__ __launch_bounds__(PF_BLOCK) void k_plane_fit(const AssocArgs a, const float4* __restrict__ scan, const float4* __restrict__ map,
                                                        const int* __restrict__ nn5,
                                                        float4* __restrict__ o_pt, float4* __restrict__ o_plane, double* __restrict__ o_score,
                                                        int* __restrict__ o_flag, int* __restrict__ o_lpos, int* __restrict__ o_bcount,
                                                        int* __restrict__ o_nn, const float4* __restrict__ loc, double* __restrict__ o_nc) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * PF_BLOCK + threadIdx.x;
    if (a.kb.counters && blockIdx.x == 0 && threadIdx.x == 0) *reinterpret_cast<int4*>(a.kb.counters + 4 * blockIdx.y) = make_int4(0, 0, 0, 0);
    if (a.kb.counters && a.kb.gctr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 8) a.kb.gctr[threadIdx.x] = 0;
    const AssocSlot sl = assoc_slot(a);
    if (blockIdx.x * PF_BLOCK >= sl.n) return;
    scan += sl.qoff; nn5 += 8 * sl.woff;
    o_pt += sl.woff; if (!BATCH) o_plane += sl.woff; o_score += sl.woff; o_flag += sl.woff; o_lpos += sl.woff; o_bcount += sl.boff;
    if (sl.map) { map = sl.map; loc += sl.locoff; o_nc += 6 * sl.woff; }
    const bool qlive = i < sl.n;
    const float4 pl = scan[qlive ? i : 0];
    const double pin[3] = {(double)pl.x, (double)pl.y, (double)pl.z};
    double po[3];
    a_qrot(sl.q, pin, po);
    const float px = (float)(po[0] + sl.t[0]), py = (float)(po[1] + sl.t[1]), pz = (float)(po[2] + sl.t[2]);
    int mp5[5], mi[5];
    float md4 = FLT_MAX;
    float4 nbp[5];
    if (qlive) {
        const int4 r0 = reinterpret_cast<const int4*>(nn5 + 8 * (size_t)i)[0], r1 = reinterpret_cast<const int4*>(nn5 + 8 * (size_t)i)[1];
        mp5[0] = r0.x; mp5[1] = r0.y; mp5[2] = r0.z; mp5[3] = r0.w; mp5[4] = r1.x; md4 = __int_as_float(r1.y);
#pragma unroll
        for (int k = 0; k < 5; ++k) { nbp[k] = map[mp5[k] >= 0 ? mp5[k] : 0]; mi[k] = __float_as_int(nbp[k].w); }
    } else {
#pragma unroll
        for (int k = 0; k < 5; ++k) { mp5[k] = -1; mi[k] = -1; nbp[k] = make_float4(0, 0, 0, 0); }
    }
    float md[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {               // same float expression as in k_knn5: identical bits
        const float ex = px - nbp[k].x, ey = py - nbp[k].y, ez = pz - nbp[k].z;
        float d = ex * ex;
        d = d + ey * ey;
        d = d + ez * ez;
        md[k] = mp5[k] >= 0 ? d : FLT_MAX;
    }
    md[4] = md4;
    const int j = 0;
    // ---- plane fit + gates by lane 0 of the group
    int valid = 0;
    if (j == 0 && qlive) {
        if (o_nn) {
#ifdef GLIO_ASSOC_DEBUG_P
            o_nn[5 * (size_t)i + 0] = __float_as_int(px); o_nn[5 * (size_t)i + 1] = __float_as_int(py); o_nn[5 * (size_t)i + 2] = __float_as_int(pz);
            o_nn[5 * (size_t)i + 3] = __float_as_int(md[3]); o_nn[5 * (size_t)i + 4] = __float_as_int(md[4]);
#else
#pragma unroll
            for (int k = 0; k < 5; ++k) o_nn[5 * (size_t)i + k] = (mp5[k] >= 0 && (double)md[k] < a.kd_max_radius) ? mi[k] : -1;
#endif
        }
        float4 oplane = make_float4(0, 0, 0, 0);
        double oscore = 0;
        if (mp5[4] >= 0 && (double)md[4] < a.kd_max_radius) {                    // Estimator.cpp:3651
            double A[5][3], A0[5][3], b[5], nrm[3];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const float4 mp = nbp[k];
                A[k][0] = A0[k][0] = (double)mp.x; A[k][1] = A0[k][1] = (double)mp.y; A[k][2] = A0[k][2] = (double)mp.z;
                b[k] = -1.0;
            }
            plane_qr_solve(A, b, nrm);
            double nloc[3] = {0, 0, 0}, cen[3] = {0, 0, 0};
            if (BATCH) {                                                     // :3843-3859
                double Al[5][3], bl[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const float4 lp = loc[mi[k]];
                    Al[k][0] = (double)lp.x; Al[k][1] = (double)lp.y; Al[k][2] = (double)lp.z;
                    cen[0] += Al[k][0]; cen[1] += Al[k][1]; cen[2] += Al[k][2];
                    bl[k] = -1.0;
                }
                plane_qr_solve(Al, bl, nloc);
                const double nln = sqrt(nloc[0] * nloc[0] + nloc[1] * nloc[1] + nloc[2] * nloc[2]);
                nloc[0] /= nln; nloc[1] /= nln; nloc[2] /= nln;
            }
            const double nn = sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
            const double normInverse = 1.0 / nn;
            nrm[0] /= nn; nrm[1] /= nn; nrm[2] /= nn;
            bool ok = true;
#pragma unroll
            for (int k = 0; k < 5; ++k)
                if (fabs(nrm[0] * A0[k][0] + nrm[1] * A0[k][1] + nrm[2] * A0[k][2] + normInverse) > a.surf_dist_thres) ok = false;
            if (ok) {
                const float pd = (float)(nrm[0] * (double)px + nrm[1] * (double)py + nrm[2] * (double)pz + normInverse);
                float r2 = px * px;
                r2 = r2 + py * py;
                r2 = r2 + pz * pz;
                const float rr = sqrtf(sqrtf(r2));
                const float weight = (float)(1.0 - 0.9 * (double)fabsf(pd) / (double)rr);
                if ((double)weight > a.weight_gate) {
                    valid = 1;
                    oplane.x = (float)((double)weight * nrm[0]);
                    oplane.y = (float)((double)weight * nrm[1]);
                    oplane.z = (float)((double)weight * nrm[2]);
                    oplane.w = (float)((double)weight * normInverse);
                    oscore = a.unit_scores ? 1.0 : a.lidar_const * (double)weight;
                    if (BATCH) {
                        double* nc = o_nc + 6 * (size_t)i;
                        nc[0] = nloc[0]; nc[1] = nloc[1]; nc[2] = nloc[2];
                        nc[3] = cen[0] / 5.; nc[4] = cen[1] / 5.; nc[5] = cen[2] / 5.;
                    }
                }
            }
        }
        o_flag[i] = valid;
        if (valid) { o_pt[i] = pl; if (!BATCH) o_plane[i] = oplane; o_score[i] = oscore; }
    }
    // ---- order-preserving positions inside the workgroup + workgroup count
    __shared__ int wcount[PF_BLOCK / 64];
    const unsigned long long bal = __ballot(valid);
    const int wv = threadIdx.x >> 6;
    if (lane == 0) wcount[wv] = __popcll(bal);
    __syncthreads();
    if (j == 0 && qlive) {
        int before = __popcll(bal & ((1ull << lane) - 1ull));
        for (int w2 = 0; w2 < wv; ++w2) before += wcount[w2];
        o_lpos[i] = before;
    }
    if (threadIdx.x == 0) { int tot = 0; for (int w2 = 0; w2 < PF_BLOCK / 64; ++w2) tot += wcount[w2]; o_bcount[blockIdx.x] = tot; }
}

// order-preserving compaction: single-workgroup exclusive scan of the flags, then scatter
// order-preserving compaction: k_plane_fit left, per workgroup of PF_BLOCK queries, the kept count (bcount) and every kept
// query's position inside the workgroup (lpos).  Each workgroup here sums the counts of the workgroups before it (a few
// hundred integers) instead of waiting for a separate scan kernel; the last one writes the slot's total.
__global__ __launch_bounds__(PF_BLOCK) void k_compact(const int* __restrict__ flag, const int* __restrict__ lpos, const int* __restrict__ bcount, int n,
                                                      const float4* __restrict__ q_pt, const float4* __restrict__ q_plane, const double* __restrict__ q_score,
                                                      float4* __restrict__ o_pt, float4* __restrict__ o_plane, double* __restrict__ o_score, int* __restrict__ total,
                                                      const int* __restrict__ win_counts, const int q_stride, const int w_stride, const int b_stride) {
    __shared__ int s_part[PF_BLOCK / 64], s_off;
    const int i = blockIdx.x * PF_BLOCK + threadIdx.x;
    if (win_counts) {
        const size_t k = blockIdx.y;
        n = win_counts[k];
        flag += k * w_stride; lpos += k * w_stride; bcount += k * b_stride; q_pt += k * w_stride; q_plane += k * w_stride; q_score += k * w_stride;
        o_pt += k * q_stride; o_plane += k * q_stride; o_score += k * q_stride; total += k;
        if (n == 0 && blockIdx.x == 0 && threadIdx.x == 0) *total = 0;
    }
    if ((int)(blockIdx.x * PF_BLOCK) >= n) return;
    int part = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += PF_BLOCK) part += bcount[b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int k = 0; k < PF_BLOCK / 64; ++k) t += s_part[k];
        s_off = t;
        if ((int)((blockIdx.x + 1) * PF_BLOCK) >= n) *total = t + bcount[blockIdx.x];
    }
    __syncthreads();
    if (i >= n || !flag[i]) return;
    const int p = s_off + lpos[i];
    o_pt[p] = q_pt[i]; o_plane[p] = q_plane[i]; o_score[p] = q_score[i];
}

// ------------------------------------------------------------------------------------------------
static int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

struct KnnBinHost { KnnBin d; int rows, cap, capq_max; };
static unsigned long long* g_knn_dbg = nullptr;      // statistics of the near-block search (glio_debug_knn_stats)
static int g_bassoc_local = -1;       // batch association, tables in the search frames' own frames: -1 = GLIO_BASSOC_LOCAL_TABLES (default on), test knob glio_debug_set_bassoc_local
static int g_gbin_cap = 0;            // test knob (glio_debug_set_gbin_cap): cell-table size of the merged-window grouping, 0 = from the map
static int knn_mode_from_env() { const char* e = getenv("GLIO_KNN_MODE"); const int m = e ? atoi(e) : 0; return m >= 0 && m <= 3 ? m : 0; }      // (A/B of whole programs, e.g. host_demo_stream)
static int g_knn_mode = knn_mode_from_env();           // 0 = window calls: the queries of all slots grouped by cell together, near block first (k_knn5_near<64>), the rest by
                                     //     k_knn5_rest; single rows: the 27-cell tiled search (k_knn5_tile);  1 = one 16-lane group per query (k_knn5);
                                     // 2 = every query by the 27-cell tiled search (the round-4 path);  3 = near block first (k_knn5_near<16>), every launch
                                     //     row by itself; glio_debug_set_knn_mode
// gcells > 0: also the buffers of the merged-window grouping, its cell table sized for gcells cells (0: rows are always searched one by one)
static KnnBinHost* knn_bin_create(int rows, int cap, int gcells = 0) {
    KnnBinHost* h = new KnnBinHost();
    memset(h, 0, sizeof *h);
    h->rows = rows; h->cap = cap; h->capq_max = next_pow2(2 * (cap > 512 ? cap : 512));
    KnnBin& d = h->d;
    d.unit_stride = cap / TK_Q + cap + 16;
    // the global-atomic binning tables (keys .. qtmp) serve the presort of ONE uploaded cloud at a time: one row; the per-call grouping
    // (k_qbin_tile) needs only the grouped queries, the unit list and the counters of every launch row
    const size_t tq = (size_t)h->capq_max, wq1 = (size_t)cap, wq = (size_t)rows * cap;
    bool ok = hipMalloc((void**)&d.keys, tq * 8) == hipSuccess && hipMalloc((void**)&d.cnt, tq * 4) == hipSuccess && hipMalloc((void**)&d.cstart, tq * 4) == hipSuccess &&
              hipMalloc((void**)&d.qslot, wq1 * 4) == hipSuccess && hipMalloc((void**)&d.qrank, wq1 * 4) == hipSuccess && hipMalloc((void**)&d.qtmp, wq1 * 16) == hipSuccess &&
              hipMalloc((void**)&d.qs, wq * 16) == hipSuccess && hipMalloc((void**)&d.units, (size_t)rows * d.unit_stride * 16) == hipSuccess &&
              hipMalloc((void**)&d.counters, (size_t)rows * 16) == hipSuccess && hipMalloc((void**)&d.fails, (size_t)rows * d.unit_stride * 16) == hipSuccess &&
              hipMalloc((void**)&d.failq, wq * 4) == hipSuccess;
    ok = ok && hipMemset(d.keys, 0xff, tq * 8) == hipSuccess && hipMemset(d.cnt, 0, tq * 4) == hipSuccess && hipMemset(d.counters, 0, (size_t)rows * 16) == hipSuccess;
    if (ok && gcells > 0 && rows > 1 && rows < 64 && cap <= (1 << NK_ROW_SHIFT) && wq < (1ull << 31)) {
        int gc = next_pow2(gcells > 4096 ? gcells : 4096);
        const int gmax = next_pow2((int)(2 * wq > (1u << 30) ? (1u << 30) : 2 * wq));
        if (gc > gmax) gc = gmax;
        d.gcap = gc; d.qcap_total = (int)wq;
        ok = hipMalloc((void**)&d.qs2, wq * 16) == hipSuccess && hipMalloc((void**)&d.segs, wq * 16) == hipSuccess && hipMalloc((void**)&d.gkeys, (size_t)gc * 8) == hipSuccess &&
             hipMalloc((void**)&d.gcnt, (size_t)gc * 4) == hipSuccess && hipMalloc((void**)&d.gstart, (size_t)gc * 4) == hipSuccess && hipMalloc((void**)&d.gctr, 32) == hipSuccess;
        ok = ok && hipMemset(d.gkeys, 0xff, (size_t)gc * 8) == hipSuccess && hipMemset(d.gcnt, 0, (size_t)gc * 4) == hipSuccess && hipMemset(d.gctr, 0, 32) == hipSuccess;
    }
    if (!ok) { glio_set_error("hipMalloc failed for the query binning buffers"); return nullptr; }
    return h;
}
static void knn_bin_destroy(KnnBinHost* h) {
    if (!h) return;
    void* p[] = {h->d.keys, h->d.cnt, h->d.cstart, h->d.qslot, h->d.qrank, h->d.qtmp, h->d.qs, h->d.units, h->d.counters, h->d.fails, h->d.failq,
                 h->d.qs2, h->d.segs, h->d.gkeys, h->d.gcnt, h->d.gstart, h->d.gctr};
    for (void* q : p) if (q) hipFree(q);
    delete h;
}
// presort one cloud (n points at `cloud`, in its own frame) into `ps` (w = original index): the same binning kernels with the
// identity pose and 5 m blocks, so that every run of 1024 consecutive points of `ps` is spatially compact
static void enqueue_presort(hipStream_t stream, KnnBinHost* kb, const float4* cloud, int n, float4* ps) {
    if (n <= 0) return;
    AssocArgs a;
    memset(&a, 0, sizeof a);
    a.q[0] = 1.0; a.inv_cell = 1.0f / PRESORT_CELL; a.n = n;
    a.kb = kb->d; a.kb.qs = ps;
    a.kb.capq = next_pow2(2 * (n > 512 ? n : 512));
    if (a.kb.capq > kb->capq_max) a.kb.capq = kb->capq_max;
    hipLaunchKernelGGL(k_qbin_count, dim3((n + QC_THREADS - 1) / QC_THREADS), dim3(QC_THREADS), 0, stream, a, cloud);
    hipLaunchKernelGGL(k_qbin_alloc, dim3(a.kb.capq / 1024), dim3(1024), 0, stream, a);
    hipLaunchKernelGGL(k_qbin_scatter, dim3((n + 255) / 256), dim3(256), 0, stream, a);
    hipMemsetAsync(kb->d.counters, 0, 16, stream);
}
// exact 5-NN of every query of the launch rows [0, rows): the caller's AssocArgs select the row geometry (assoc_slot);
// `scan` = the clouds as uploaded, `ps` = their presorted copies
static void enqueue_knn(hipStream_t stream, AssocArgs& a, KnnBinHost* kb, int rows, int maxn, const float4* scan, const float4* ps, const float4* map,
                        const int4* ent, const uint4* sub, int map_n_max, int* nn5, int bin_rows = 0) {
    // bin_rows > 0 (pair mode, a.pair_row / a.row_pair set): only that many launch rows are grouped by cell; every pair searches its bin row's units
    if (bin_rows <= 0 || !a.pair_row) { bin_rows = rows; a.pair_row = nullptr; a.row_pair = nullptr; }
    a.bin_pass = 0;
    if (g_knn_mode == 1 || !kb) {
        memset(&a.kb, 0, sizeof a.kb);
        hipLaunchKernelGGL(k_knn5, dim3((maxn + AQ_PER_BLOCK - 1) / AQ_PER_BLOCK, rows), dim3(256), 0, stream, a, scan, map, ent, nn5);
        return;
    }
    a.kb = kb->d; a.w_stride_q = kb->cap; a.kb.dbg = g_knn_dbg;
    // merged window (mode 0, launch rows that share one map): the queries of all rows grouped by cell together, units of 64
    if (g_knn_mode == 0 && map_n_max <= (1 << 24) && rows > 1 && map && kb->d.qs2 && !a.frames) {
        int ge = g_gbin_cap > 0 ? next_pow2(g_gbin_cap) : next_pow2(map_n_max > 4096 ? map_n_max : 4096);      // (queried cells of the C2 window: ~10 k against a 70 k-point map)
        if (ge > kb->d.gcap) ge = kb->d.gcap;
        a.kb.glob = 1; a.kb.gcap = ge;
        hipLaunchKernelGGL(k_qbin_tile, dim3((maxn + QT_THREADS - 1) / QT_THREADS, rows), dim3(QT_THREADS), 0, stream, a, ps);
        hipLaunchKernelGGL(k_gbin_alloc, dim3((ge + 1023) / 1024), dim3(1024), 0, stream, a.kb);
        const long long tot = (long long)rows * maxn;
        hipLaunchKernelGGL(k_gbin_scatter, dim3((unsigned)((tot / 512 > 4096 ? 4096 : tot / 512) + 1)), dim3(256), 0, stream, a.kb);
        a.kb.counters = kb->d.gctr;                                // one launch row: the counter block of the merged window
        long long gn = tot / 32 + 64;                               // units of 64 at half fill; beyond that the workgroups stride
        if (gn > 65536) gn = 65536;
        hipLaunchKernelGGL(k_knn5_near<64>, dim3((unsigned)((gn + NK_WPB - 1) / NK_WPB), 1), dim3(64 * NK_WPB), 0, stream, a, map, ent, sub, nn5);
        long long gq = tot / 64 + 64;                               // (one pass for up to 1/16 of the queries, 4 per workgroup; 2048 x 2 handed-on units)
        if (gq > 16384) gq = 16384;
        a.kb.use_fails = 1;
        hipLaunchKernelGGL(k_knn5_rest, dim3((unsigned)gq + 2048, 1), dim3(64), 0, stream, a, (int)gq, scan, map, ent, nn5);
        a.kb.use_fails = 0; a.kb.glob = 0; a.kb.counters = kb->d.counters; a.kb.gcap = kb->d.gcap;
        return;
    }
    a.bin_pass = a.pair_row ? 1 : 0;
    hipLaunchKernelGGL(k_qbin_tile, dim3((maxn + QT_THREADS - 1) / QT_THREADS, bin_rows), dim3(QT_THREADS), 0, stream, a, ps);
    a.bin_pass = 0;
    // capacity for maxn / 8 units per row (k_qbin_tile makes at most n / 16 + cells): one wavefront-workgroup per unit pair up to 4096 per row, beyond
    // that the workgroups stride
    // (sizing the grid to what the chip holds at once -- ~4096 workgroups striding over the units -- measured 390 us instead of 303 us for the one-call
    //  window association of 20 rows: the dispatcher is not the limit, and fresh workgroups overlap their probe / staging latencies better)
    int gx = (maxn + 8 * TK_UNITS - 1) / (8 * TK_UNITS);
    if (gx > 4096) gx = 4096;
    // (the tie-break index rides in 24 bits of the near-block search's exact key: maps beyond 2^24 points take the 27-cell search alone)
    // A launch row by itself (a lone scan, C3, the keyframe pairs of the batch association): ~10-20 queries fall in a cell, the near-block kernel then
    // carries four quarter-filled units per wavefront and needs a second launch for what it hands on -- measured equal or slower than the 27-cell
    // tiled search alone (C2 scan 46 vs 40 us, C3 57-63 vs 57, 192 pairs 2.3-2.5 vs 2.33 ms), so mode 0 keeps the tiled search there; mode 3 forces
    // the near-block path row by row (tests: the same bytes).
    if (g_knn_mode == 3 && map_n_max <= (1 << 24)) {
        int gn = (maxn + 31) / 32;
        if (gn > 4096) gn = 4096;
        hipLaunchKernelGGL(k_knn5_near<16>, dim3((gn + NK_WPB - 1) / NK_WPB, rows), dim3(64 * NK_WPB), 0, stream, a, map, ent, sub, nn5);
        // the lists are short (0.1-4 % of the queries on the C2 stream): striding workgroups
        int gq = (maxn + 15) / 16, gt = gx;
        if (gq > 1024) gq = 1024;
        if (gt > 256) gt = 256;
        a.kb.use_fails = 1;
        hipLaunchKernelGGL(k_knn5_rest, dim3(gq + gt, rows), dim3(64), 0, stream, a, gq, scan, map, ent, nn5);
        a.kb.use_fails = 0;
        return;
    }
    hipLaunchKernelGGL(k_knn5_tile, dim3(gx, rows), dim3(TK_THREADS), 0, stream, a, map, ent, nn5);
}

int glio_assoc_create(glio_ctx* c) {
    AssocWork* w = new AssocWork();
    memset(w, 0, sizeof *w);
    const float r = sqrtf((float)c->opts.kd_max_radius);
    w->cell = fmaxf(1.25f, r * 1.0001f);
    w->inv_cell = 1.0f / w->cell;
    const int mm = c->opts.max_map_points > 0 ? c->opts.max_map_points : 1;
    w->table_cap = next_pow2(2 * mm);
    const int cap = c->cap;
#define AALLOC(ptr, bytes) do { if (hipMalloc((void**)&(ptr), (size_t)(bytes)) != hipSuccess) { glio_set_error("hipMalloc failed in assoc_create"); return GLIO_E_HIP; } } while (0)
    w->cap_eff = w->table_cap;
    AALLOC(w->d_keys, (size_t)w->table_cap * 8); AALLOC(w->d_ent, (size_t)w->table_cap * 16); AALLOC(w->d_cnt8, (size_t)w->table_cap * 32);
    AALLOC(w->d_sub, (size_t)w->table_cap * 16);
    AALLOC(w->d_pt_slot, (size_t)mm * 4); AALLOC(w->d_pt_rank, (size_t)mm * 4); AALLOC(w->d_map_raw, (size_t)mm * 16); AALLOC(c->d_map_sorted, (size_t)mm * 16);
    AALLOC(w->d_total, 4); AALLOC(w->d_count_tmp, 4);
    // dense per-query work arrays for ALL W slots (72 B per query): the window association runs the slots in one launch
    const size_t wc = (size_t)cap * c->W, wb = (size_t)(cap / AQ_PER_BLOCK + 2) * c->W;
    AALLOC(w->d_q_pt, wc * 16); AALLOC(w->d_q_plane, wc * 16); AALLOC(w->d_q_score, wc * 8);
    AALLOC(w->d_q_flag, wc * 4); AALLOC(w->d_q_pos, wc * 4); AALLOC(w->d_nn, (size_t)cap * 5 * 4);
    AALLOC(w->d_nn5, wc * 32);
    AALLOC(w->d_bcount, wb * 4); AALLOC(w->d_boff, wb * 4);
    AALLOC(w->d_win, (size_t)c->W * 64);
    if (hipHostMalloc((void**)&w->h_count, 16) != hipSuccess) return GLIO_E_HIP;
    if (hipHostMalloc((void**)&w->h_counts_win, GLIO_MAX_WINDOW * 4) != hipSuccess) return GLIO_E_HIP;
    w->counts_pending = 0;
    if (hipHostMalloc((void**)&w->h_win, (size_t)c->W * 64) != hipSuccess) return GLIO_E_HIP;
    w->kb = knn_bin_create(c->W, cap, 4 * mm);
    if (!w->kb) return GLIO_E_HIP;
    AALLOC(w->d_ps, wc * 16);
    c->assoc = w;
    c->map_n = 0;
    return GLIO_OK;
}

void glio_assoc_destroy(glio_ctx* c) {
    AssocWork* w = c->assoc;
    if (!w) return;
    void* ptrs[] = {w->d_keys, w->d_ent, w->d_cnt8, w->d_sub, w->d_pt_slot, w->d_pt_rank, w->d_map_raw, c->d_map_sorted, w->d_total,
                    w->d_count_tmp, w->d_q_pt, w->d_q_plane, w->d_q_score, w->d_q_flag, w->d_q_pos, w->d_nn, w->d_nn5, w->d_bcount, w->d_boff, w->d_win, w->d_ps};
    for (void* p : ptrs) if (p) hipFree(p);
    knn_bin_destroy(w->kb);
    hipHostFree(w->h_count);
    if (w->h_counts_win) hipHostFree(w->h_counts_win);
    if (w->h_win) hipHostFree(w->h_win);
    if (w->ev_counts) hipEventDestroy(w->ev_counts);
    if (w->ev_sel) hipEventDestroy(w->ev_sel);
    if (w->h_sel) hipHostFree(w->h_sel);
    if (w->d_sel) hipFree(w->d_sel);
    delete w;
    c->assoc = nullptr;
}

// a scan was uploaded to / moved between slots: keep its presorted copy in step (enqueued on the context stream)
void glio_assoc_presort_row(glio_ctx* c, hipStream_t stream, size_t row_offset, int n) {      // glio_set_scan_ahead: the presort of a ring row on a stream of the caller's
    AssocWork* w = c->assoc;
    if (w) enqueue_presort(stream, w->kb, c->d_scan + row_offset, n, w->d_ps + row_offset);
}
void glio_assoc_scan_uploaded(glio_ctx* c, int slot, int n) {
    AssocWork* w = c->assoc;
    if (w) { const size_t row = (size_t)glio_scan_row(c, slot) * c->cap; enqueue_presort(c->stream, w->kb, c->d_scan + row, n, w->d_ps + row); }
}
static void enqueue_build(glio_ctx* c, int n, const float4* src = nullptr) {
    AssocWork* w = c->assoc;
    if (!src) src = w->d_map_raw;                          // (src: the map's points as they lie on the device -- the local map's output needs no copy into d_map_raw)
    w->last_src = src;                                     // (what a re-build of the same map -- the timing hook -- reads)
    int cap = next_pow2(2 * (n > 512 ? n : 512));          // sized for THIS map: a smaller table stays in L2
    if (cap > w->table_cap) cap = w->table_cap;
    w->cap_eff = cap;
    hipLaunchKernelGGL(k_hash_clear, dim3((cap + 255) / 256), dim3(256), 0, c->stream, w->d_keys, w->d_cnt8, cap, w->d_total, w->d_ent, w->d_sub, n == 0 ? 1 : 0);
    if (n == 0) return;
    hipLaunchKernelGGL(k_hash_insert, dim3((n + HI_THREADS - 1) / HI_THREADS), dim3(HI_THREADS), 0, c->stream, src, n, w->inv_cell, w->d_keys, w->d_cnt8, w->d_pt_slot, w->d_pt_rank, cap);
    hipLaunchKernelGGL(k_cell_alloc, dim3((cap + 1023) / 1024), dim3(1024), 0, c->stream, w->d_cnt8, cap, w->d_total, w->d_keys, w->d_ent, w->d_sub);
    hipLaunchKernelGGL(k_scatter, dim3((n + 255) / 256), dim3(256), 0, c->stream, src, n, w->d_pt_slot, w->d_pt_rank, w->d_cnt8, w->d_ent, c->d_map_sorted);
}

int glio_assoc_build_map(glio_ctx* c, const void* map_points, int n, int stride, int ioff) {
    AssocWork* w = c->assoc;
    if (!w) return GLIO_E_STATE;
    { const int ru = glio_upload_points(c->stream, &c->raw_stage, map_points, n, stride, ioff, w->d_map_raw); if (ru != GLIO_OK) return ru; }
    enqueue_build(c, n);
    GLIO_HIP_CHECK(hipGetLastError());
    GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->map_n = n;
    return GLIO_OK;
}

// K1 from points that are already on the device (the local map built by localmap_kernels.hip)
int glio_assoc_build_map_dev(glio_ctx* c, const float4* d_pts, int n) {
    AssocWork* w = c->assoc;
    if (!w) return GLIO_E_STATE;
    if (n > c->opts.max_map_points) { glio_set_error("local map has %d points, max_map_points is %d", n, c->opts.max_map_points); return GLIO_E_ARG; }
    enqueue_build(c, n, d_pts);          // (straight from the caller's device array: it stays untouched until the next local-map build, which this stream orders behind)
    GLIO_HIP_CHECK(hipGetLastError());
    c->map_n = n;
    return GLIO_OK;
}

static void enqueue_assoc(glio_ctx* c, int slot, const double q[4], const double t[3], int n, int want_nn) {
    AssocWork* w = c->assoc;
    AssocArgs a;
    memset(&a, 0, sizeof a);
    for (int k = 0; k < 4; ++k) a.q[k] = q[k];
    for (int k = 0; k < 3; ++k) a.t[k] = t[k];
    a.inv_cell = w->inv_cell; a.cell = w->cell; a.kd_max_radius = c->opts.kd_max_radius; a.weight_gate = c->opts.weight_gate;
    a.surf_dist_thres = c->opts.surf_dist_thres; a.lidar_const = c->opts.lidar_const;
    a.n = n; a.table_cap = w->cap_eff; a.unit_scores = c->opts.unit_scores;
    a.win_poses = nullptr; a.win_counts = nullptr; a.q_stride = a.w_stride = a.b_stride = 0;
    const size_t off = (size_t)slot * c->cap, row = (size_t)glio_scan_row(c, slot) * c->cap;      // correspondences by window slot, scans by ring row
    const int nblk = (n + PF_BLOCK - 1) / PF_BLOCK;
    if (n > 0) {
        enqueue_knn(c->stream, a, w->kb, 1, n, c->d_scan + row, w->d_ps + row, c->d_map_sorted, w->d_ent, w->d_sub, c->map_n, w->d_nn5);
        hipLaunchKernelGGL(k_plane_fit<false>, dim3(nblk), dim3(PF_BLOCK), 0, c->stream, a, c->d_scan + row, c->d_map_sorted, w->d_nn5,
                           w->d_q_pt, w->d_q_plane, w->d_q_score, w->d_q_flag, w->d_q_pos, w->d_bcount, want_nn ? w->d_nn : nullptr,
                           (const float4*)nullptr, (double*)nullptr);
    }
    if (n > 0) {
        hipLaunchKernelGGL(k_compact, dim3((n + PF_BLOCK - 1) / PF_BLOCK), dim3(PF_BLOCK), 0, c->stream, w->d_q_flag, w->d_q_pos, w->d_bcount, n, w->d_q_pt,
                           w->d_q_plane, w->d_q_score, c->d_pts + off, c->d_planes + off, c->d_scores + off, c->d_count + slot, (const int*)nullptr, 0, 0, 0);
    } else {
        hipMemsetAsync(c->d_count + slot, 0, 4, c->stream);
    }
}

// all W slots in four launches (tile binning, search, plane fit, compaction): blockIdx.y = slot
static int enqueue_assoc_window(glio_ctx* c, const double* quats, const double* trans) {
    AssocWork* w = c->assoc;
    const int W = c->W;
    int maxn = 0;
    for (int s = 0; s < W; ++s) {
        for (int k = 0; k < 4; ++k) w->h_win[7 * s + k] = quats[4 * s + k];
        for (int k = 0; k < 3; ++k) w->h_win[7 * s + 4 + k] = trans[3 * s + k];
        reinterpret_cast<int*>(w->h_win + 7 * W)[s] = c->h_scan_count[s];
        if (c->h_scan_count[s] > maxn) maxn = c->h_scan_count[s];
    }
    GLIO_HIP_CHECK(hipMemcpyAsync(w->d_win, w->h_win, (size_t)W * 60, hipMemcpyHostToDevice, c->stream));
    AssocArgs a;
    memset(&a, 0, sizeof a);
    a.inv_cell = w->inv_cell; a.cell = w->cell; a.kd_max_radius = c->opts.kd_max_radius; a.weight_gate = c->opts.weight_gate;
    a.surf_dist_thres = c->opts.surf_dist_thres; a.lidar_const = c->opts.lidar_const;
    a.n = 0; a.table_cap = w->cap_eff; a.unit_scores = c->opts.unit_scores;
    a.win_poses = w->d_win; a.win_counts = reinterpret_cast<const int*>(w->d_win + 7 * W);
    a.q_stride = c->cap; a.w_stride = c->cap; a.b_stride = c->cap / AQ_PER_BLOCK + 2;
    a.ring_base = c->scan_base; a.ring_W = W;
    if (maxn > 0) {
        enqueue_knn(c->stream, a, w->kb, W, maxn, c->d_scan, w->d_ps, c->d_map_sorted, w->d_ent, w->d_sub, c->map_n, w->d_nn5);
        hipLaunchKernelGGL(k_plane_fit<false>, dim3((maxn + PF_BLOCK - 1) / PF_BLOCK, W), dim3(PF_BLOCK), 0, c->stream, a, c->d_scan, c->d_map_sorted,
                           w->d_nn5, w->d_q_pt, w->d_q_plane, w->d_q_score, w->d_q_flag, w->d_q_pos, w->d_bcount, (int*)nullptr,
                           (const float4*)nullptr, (double*)nullptr);
    }
    hipLaunchKernelGGL(k_compact, dim3(maxn > 0 ? (maxn + PF_BLOCK - 1) / PF_BLOCK : 1, W), dim3(PF_BLOCK), 0, c->stream, w->d_q_flag, w->d_q_pos, w->d_bcount, 0, w->d_q_pt,
                       w->d_q_plane, w->d_q_score, c->d_pts, c->d_planes, c->d_scores, c->d_count, a.win_counts, a.q_stride, a.w_stride, a.b_stride);
    return GLIO_OK;
}

int glio_assoc_finish_pending(glio_ctx* c);
int glio_assoc_run(glio_ctx* c, int slot, const double q[4], const double t[3], int* out_count) {
    AssocWork* w = c->assoc;
    if (!w) return GLIO_E_STATE;
    { const int rp = glio_assoc_finish_pending(c); if (rp != GLIO_OK) return rp; }
    if (c->map_n <= 0) { glio_set_error("no map set"); return GLIO_E_STATE; }
    const int n = c->h_scan_count[slot];
    if (slot == 0) { for (int k = 0; k < 4; ++k) w->last_pose0[k] = q[k]; for (int k = 0; k < 3; ++k) w->last_pose0[4 + k] = t[k]; w->have_pose0 = 1; }
    enqueue_assoc(c, slot, q, t, n, 1);
    GLIO_HIP_CHECK(hipGetLastError());
    GLIO_HIP_CHECK(hipMemcpyAsync(w->h_count, c->d_count + slot, 4, hipMemcpyDeviceToHost, c->stream));
    GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->h_count[slot] = w->h_count[0];
    if (out_count) *out_count = w->h_count[0];
    return GLIO_OK;
}

// featureSelection (Estimator.cpp:3894-3992) keeps a random subset of a slot's correspondences in draw order; the draws are
// the caller's (host RNG), the device only gathers: record k <- record sel[k]
__global__ void k_gather_corr(const int* __restrict__ sel, int n, const float4* __restrict__ pts, const float4* __restrict__ planes,
                              const double* __restrict__ scores, float4* __restrict__ o_pts, float4* __restrict__ o_planes, double* __restrict__ o_scores) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int s = sel[k];
    o_pts[k] = pts[s]; o_planes[k] = planes[s]; o_scores[k] = scores[s];
}
int glio_assoc_select(glio_ctx* c, int slot, const int32_t* indices, int n) {
    AssocWork* w = c->assoc;
    if (!w) return GLIO_E_STATE;
    const int cur = c->h_count[slot];
    if (n < 0 || n > cur) { glio_set_error("selection of %d out of %d correspondences", n, cur); return GLIO_E_ARG; }
    for (int k = 0; k < n; ++k) if (indices[k] < 0 || indices[k] >= cur) { glio_set_error("selection index %d out of range", indices[k]); return GLIO_E_ARG; }
    const size_t off = (size_t)slot * c->cap;
    if (n > 0) {
        GLIO_HIP_CHECK(hipMemcpyAsync(w->d_q_pos, indices, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_gather_corr, dim3((n + 255) / 256), dim3(256), 0, c->stream, w->d_q_pos, n, c->d_pts + off, c->d_planes + off, c->d_scores + off,
                           w->d_q_pt, w->d_q_plane, w->d_q_score);
        GLIO_HIP_CHECK(hipMemcpyAsync(c->d_pts + off, w->d_q_pt, (size_t)n * 16, hipMemcpyDeviceToDevice, c->stream));
        GLIO_HIP_CHECK(hipMemcpyAsync(c->d_planes + off, w->d_q_plane, (size_t)n * 16, hipMemcpyDeviceToDevice, c->stream));
        GLIO_HIP_CHECK(hipMemcpyAsync(c->d_scores + off, w->d_q_score, (size_t)n * 8, hipMemcpyDeviceToDevice, c->stream));
    }
    c->h_count[slot] = n;
    GLIO_HIP_CHECK(hipMemcpyAsync(c->d_count + slot, &c->h_count[slot], 4, hipMemcpyHostToDevice, c->stream));
    GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));
    return GLIO_OK;
}

// featureSelection for the WHOLE window in one call (the released configuration selects in every slot of every keyframe call, Estimator.cpp:2222-2223: five
// glio_assoc_select calls = five pageable uploads, fifteen device copies and five stream synchronisations, ~0.2 ms of a 0.95 ms call).  One pinned block
// [offsets W + 1 | changed W | indices] -> one copy, a gather of every changed slot into the dense work arrays, a put-back that also installs the counts;
// nothing is waited for: the solve is ordered behind it on the stream.
__global__ void k_gather_corr_win(const int* __restrict__ blk, const int W, const int cap, const float4* __restrict__ pts, const float4* __restrict__ planes,
                                  const double* __restrict__ scores, float4* __restrict__ o_pts, float4* __restrict__ o_planes, double* __restrict__ o_scores) {
    const int s = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
    const int o0 = blk[s], n = blk[s + 1] - o0;
    if (!blk[W + 1 + s] || k >= n) return;
    const size_t off = (size_t)s * cap;
    const int src = blk[2 * W + 1 + o0 + k];
    o_pts[off + k] = pts[off + src]; o_planes[off + k] = planes[off + src]; o_scores[off + k] = scores[off + src];
}
__global__ void k_put_corr_win(const int* __restrict__ blk, const int W, const int cap, const float4* __restrict__ s_pts, const float4* __restrict__ s_planes,
                               const double* __restrict__ s_scores, float4* __restrict__ pts, float4* __restrict__ planes, double* __restrict__ scores, int* __restrict__ count) {
    const int s = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blk[s + 1] - blk[s];
    if (!blk[W + 1 + s]) return;
    if (k == 0) count[s] = n;
    if (k >= n) return;
    const size_t off = (size_t)s * cap;
    pts[off + k] = s_pts[off + k]; planes[off + k] = s_planes[off + k]; scores[off + k] = s_scores[off + k];
}
int glio_assoc_select_window(glio_ctx* c, const int32_t* offsets, const int32_t* indices, const uint8_t* changed) {
    AssocWork* w = c->assoc;
    if (!w) return GLIO_E_STATE;
    const int W = c->W;
    if (offsets[0] != 0) { glio_set_error("offsets[0] must be 0"); return GLIO_E_ARG; }
    int maxn = 0, any = 0;
    for (int s = 0; s < W; ++s) {
        const int n = offsets[s + 1] - offsets[s], cur = c->h_count[s];
        if (n < 0) { glio_set_error("offsets not ascending at slot %d", s); return GLIO_E_ARG; }
        if (changed && !changed[s]) continue;
        if (n > cur) { glio_set_error("selection of %d out of %d correspondences (slot %d)", n, cur, s); return GLIO_E_ARG; }
        for (int k = offsets[s]; k < offsets[s + 1]; ++k) if (indices[k] < 0 || indices[k] >= cur) { glio_set_error("selection index %d out of range (slot %d)", indices[k], s); return GLIO_E_ARG; }
        if (n > maxn) maxn = n;
        any = 1;
    }
    if (!any) return GLIO_OK;
    const size_t words = (size_t)2 * W + 1 + (size_t)offsets[W];
    if (words > w->sel_cap) {
        if (w->sel_in_flight) { GLIO_HIP_CHECK(hipEventSynchronize(w->ev_sel)); w->sel_in_flight = 0; }
        if (w->h_sel) hipHostFree(w->h_sel);
        if (w->d_sel) hipFree(w->d_sel);
        w->h_sel = nullptr; w->d_sel = nullptr; w->sel_cap = 0;
        const size_t cap = words + words / 2 + 1024;
        GLIO_HIP_CHECK(hipHostMalloc((void**)&w->h_sel, cap * 4)); GLIO_HIP_CHECK(hipMalloc((void**)&w->d_sel, cap * 4));
        w->sel_cap = cap;
    }
    if (!w->ev_sel) GLIO_HIP_CHECK(hipEventCreateWithFlags(&w->ev_sel, hipEventDisableTiming));
    if (w->sel_in_flight) { GLIO_HIP_CHECK(hipEventSynchronize(w->ev_sel)); w->sel_in_flight = 0; }     // (the previous call's block has left the pinned copy)
    for (int s = 0; s <= W; ++s) w->h_sel[s] = offsets[s];
    for (int s = 0; s < W; ++s) w->h_sel[W + 1 + s] = changed ? (changed[s] ? 1 : 0) : 1;
    if (offsets[W] > 0) memcpy(w->h_sel + 2 * W + 1, indices, (size_t)offsets[W] * 4);
    GLIO_HIP_CHECK(hipMemcpyAsync(w->d_sel, w->h_sel, words * 4, hipMemcpyHostToDevice, c->stream));
    GLIO_HIP_CHECK(hipEventRecord(w->ev_sel, c->stream));
    w->sel_in_flight = 1;
    const dim3 grid((unsigned)((std::max(maxn, 1) + 255) / 256), (unsigned)W);
    if (maxn > 0)
        hipLaunchKernelGGL(k_gather_corr_win, grid, dim3(256), 0, c->stream, w->d_sel, W, c->cap, c->d_pts, c->d_planes, c->d_scores, w->d_q_pt, w->d_q_plane, w->d_q_score);
    hipLaunchKernelGGL(k_put_corr_win, grid, dim3(256), 0, c->stream, w->d_sel, W, c->cap, w->d_q_pt, w->d_q_plane, w->d_q_score, c->d_pts, c->d_planes, c->d_scores, c->d_count);
    GLIO_HIP_CHECK(hipGetLastError());
    for (int s = 0; s < W; ++s) if (!changed || changed[s]) c->h_count[s] = offsets[s + 1] - offsets[s];
    return GLIO_OK;
}

// all W slots back to back on the stream, ONE host synchronisation: the per-slot sync of glio_assoc_run (count
// read-back) costs as much as half a K2 launch
int glio_assoc_finish_pending(glio_ctx* c);
int glio_assoc_run_window(glio_ctx* c, const double* quats, const double* trans, int32_t* out_counts) {
    AssocWork* w = c->assoc;
    if (!w) return GLIO_E_STATE;
    { const int rp = glio_assoc_finish_pending(c); if (rp != GLIO_OK) return rp; }      // (counts of an earlier asynchronous call must not overwrite this call's later)
    if (c->map_n <= 0) { glio_set_error("no map set"); return GLIO_E_STATE; }
    for (int k = 0; k < 4; ++k) w->last_pose0[k] = quats[k];
    for (int k = 0; k < 3; ++k) w->last_pose0[4 + k] = trans[k];
    w->have_pose0 = 1;
    { const int rc = enqueue_assoc_window(c, quats, trans); if (rc != GLIO_OK) return rc; }
    GLIO_HIP_CHECK(hipGetLastError());
    GLIO_HIP_CHECK(hipMemcpyAsync(c->h_count, c->d_count, (size_t)c->W * 4, hipMemcpyDeviceToHost, c->stream));
    GLIO_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (out_counts) for (int s = 0; s < c->W; ++s) out_counts[s] = c->h_count[s];
    return GLIO_OK;
}

// The same without waiting: the searches are enqueued, the counts travel to pinned memory behind them, the call returns.  The host is free
// for the window's factor tables (glio_set_imu / glio_set_gnss: their marshalling then runs while the GPU searches); glio_assoc_finish_pending --
// called by glio_associate_window_counts and by every entry point that needs the correspondences -- waits and takes the counts over.
int glio_assoc_run_window_async(glio_ctx* c, const double* quats, const double* trans) {
    AssocWork* w = c->assoc;
    if (!w) return GLIO_E_STATE;
    if (c->map_n <= 0) { glio_set_error("no map set"); return GLIO_E_STATE; }
    // an earlier asynchronous call may still be reading the pinned staging block (poses, scan counts) this one rewrites: take it over first
    { const int rp = glio_assoc_finish_pending(c); if (rp != GLIO_OK) return rp; }
    for (int k = 0; k < 4; ++k) w->last_pose0[k] = quats[k];
    for (int k = 0; k < 3; ++k) w->last_pose0[4 + k] = trans[k];
    w->have_pose0 = 1;
    { const int rc = enqueue_assoc_window(c, quats, trans); if (rc != GLIO_OK) return rc; }
    GLIO_HIP_CHECK(hipGetLastError());
    GLIO_HIP_CHECK(hipMemcpyAsync(w->h_counts_win, c->d_count, (size_t)c->W * 4, hipMemcpyDeviceToHost, c->stream));
    if (!w->ev_counts) GLIO_HIP_CHECK(hipEventCreateWithFlags(&w->ev_counts, hipEventDisableTiming));
    GLIO_HIP_CHECK(hipEventRecord(w->ev_counts, c->stream));
    w->counts_pending = 1;
    return GLIO_OK;
}
int glio_assoc_finish_pending(glio_ctx* c) {
    AssocWork* w = c->assoc;
    if (!w || !w->counts_pending) return GLIO_OK;
    // the counts' copy, not the whole stream: the factor tables the caller has set meanwhile (k_unstage, the clock-drift blocks' memset) are still being
    // installed behind the searches, and the solve's first launches can be enqueued while they are (0.03 ms of idle GPU per keyframe otherwise)
    GLIO_HIP_CHECK(hipEventSynchronize(w->ev_counts));
    for (int s = 0; s < c->W; ++s) c->h_count[s] = w->h_counts_win[s];
    w->counts_pending = 0;
    return GLIO_OK;
}

// test hook: neighbour indices of the last association (original map indices, -1 where the gate failed)
extern "C" int glio_debug_last_nn(glio_ctx* c, int32_t* out, int n) {
    if (!c || !c->assoc) return GLIO_E_STATE;
    GLIO_HIP_CHECK(hipMemcpy(out, c->assoc->d_nn, (size_t)n * 5 * 4, hipMemcpyDeviceToHost));
    return GLIO_OK;
}

void glio_assoc_time_hooks(glio_ctx* c, int which, int reps, float* ms) {
    *ms = 0;
    AssocWork* w = c->assoc;
    if (!w || c->map_n <= 0) return;
    // the workload is slot 0's resident scan at the pose it was last associated with (identity if it never was): the
    // queries must sit where the map is, or the probes hit empty cells and the kernel looks faster than it is
    double q[4] = {1, 0, 0, 0}, t[3] = {0, 0, 0};
    if (w->have_pose0) { for (int k = 0; k < 4; ++k) q[k] = w->last_pose0[k]; for (int k = 0; k < 3; ++k) t[k] = w->last_pose0[4 + k]; }
    for (int pass = 0; pass < 2; ++pass) {
        const int r = pass == 0 ? 1 : reps;
        if (pass == 1) hipEventRecord(c->ev0, c->stream);
        for (int k = 0; k < r; ++k) {
            if (which == GLIO_KERNEL_MAP_BUILD) enqueue_build(c, c->map_n, w->last_src);
            else {
                // it overwrites slot 0's correspondences with the same result
                enqueue_assoc(c, 0, q, t, c->h_scan_count[0], 0);
            }
        }
        if (pass == 1) hipEventRecord(c->ev1, c->stream);
        hipStreamSynchronize(c->stream);
    }
    hipEventElapsedTime(ms, c->ev0, c->ev1);
    *ms /= reps;
}


// ================================================================================================
// Batch association: findGlobalCorrespondingSurfFeaturesAdd_Batch for a list of keyframe pairs.
// Every keyframe gets its OWN voxel hash of its cloud in the global frame, built once per run and kept resident
// (~2.9 MB per 32k-point keyframe: 2000 keyframes = 5.8 GB of the 288 GB); pair (ci, cj) then queries the points of ci
// against hash[cj], BA_CHUNK pairs per launch (blockIdx.y = pair of the chunk; frame descriptors, pair lists and poses
// live in device tables).  The kept records of consecutive pairs are appended at a device-side running offset (no host
// round trip between pairs), which yields exactly the pair-major constraint arrays K8 (batch_kernels.hip) consumes.
// ================================================================================================
struct FrameHash {
    int n, table_cap, cap_eff;
    int4* d_ent; uint4* d_sub;        // resident: what the pair searches read
    float4* d_sorted;                 // the cloud in the table's order, GLOBAL frame (poses of the last run)
    // local mode: the table was built from the keyframe's LOCAL cloud (once per cloud, valid whatever the poses do); d_sorted_local is the cloud in that
    // table's order, d_sorted its re-posed copy of the run
    float4* d_sorted_local; int local_valid;
};
struct glio_bassoc {
    int device; hipStream_t stream;
    int prep_nb, prep_todo[64], prep_n[64], prep_tc[64];      // glio_bassoc_prepare_async: the first batch of search frames whose descriptors are on the device and whose tables are cleared
    hipEvent_t ev_scan;             // glio_bassoc_set_frame_from_scan: the point of the context's stream the copy of its scan waits for (no host wait)
    int K, cap; long long max_con;
    float inv_cell, cell;
    float4* d_local;                // [K][cap] keyframe-local clouds
    float4* d_local_ps;             // [K][cap] the same, presorted (w = index in the cloud)
    float4* d_global;               // [cap] staging: one cloud in the global frame
    int* h_n;                       // [K]
    FrameHash* frames;              // [K]
    int* d_total;                   // scratch of the hash build: [BA_FB]
    unsigned long long* d_bkeys; unsigned* d_bcnt8; int* d_bslot; int* d_brank;      // build scratch of one batch of BA_FB keyframes: [BA_FB][tc], [BA_FB][tc][8], [BA_FB][cap] x 2
    struct FrameBuild* d_fb; struct FrameBuild* h_fb;       // [2 K] build descriptors of the keyframes of a run, batch after batch (device / pinned); local mode: [0, K) re-posing, [K, 2 K) builds
    float4* d_sorted_local;         // [K][cap] slab behind FrameHash::d_sorted_local
    // local mode: everything a run sends ahead of its kernels -- poses [K][7], re-posing descriptors [K], frame descriptors [K], four pair arrays [4 K] each --
    // as ONE pinned block and ONE copy (they were eight copies of a few hundred bytes, ~9 us of stream time each, in front of the searches of every keyframe call)
    char* h_inbox; char* d_inbox; size_t inbox_bytes;
    // dense per-query results of the pair in flight
    float4* d_q_cp; double* d_q_nc; double* d_q_score; int* d_q_flag; int* d_q_pos; int* d_bcount; int* d_boff;
    int* d_nn5;
    KnnBinHost* kb;                 // query binning buffers of the tiled search, BA_CHUNK rows
    // compacted output, pair major
    float4* d_cp; double* d_nc; double* d_score;
    long long* d_run;               // [1] running total
    long long* d_pair_off; int max_pairs;     // [max_pairs + 1]
    FrameDesc* d_frames; int* d_pair_ci; int* d_pair_cj;      // [K], [max_pairs] x 2: what the chunked launches index by blockIdx.y
    int* d_pair_row; int* d_row_pair;                          // [max_pairs] x 2: shared query binning (AssocArgs::pair_row / row_pair)
    int b_stride;                             // per-pair stride of the per-workgroup count arrays
    long long* h_pair_off;          // pinned
    long long* h_tail;              // pinned [2]: running total, overflow flag of the last run
    double* h_poses; int32_t* h_pairs; FrameDesc* h_fd;      // pinned staging of a run's inputs ([K][7], [2][max_pairs], [K]): the asynchronous run returns before they are read
    GlioRawStage raw_stage;         // staging of strided clouds (glio_bassoc_set_frame_strided)
    int pending_pairs;              // pairs of an asynchronous run that is still on the stream (-1: none)
    int done_pairs, done_overflow;  // ... of a run the stream was waited for (by any entry point) but whose counts glio_bassoc_finish has not fetched yet (-1: none):
                                    // they stay in h_pair_off / h_tail until it does (or until the next run replaces them)
    hipEvent_t ev_fb; int fb_in_flight;   // the last upload of build descriptors from the pinned h_fb (rewritten only behind it)
    // glio_bassoc_select_tail_draws_async: the caller's raw draws (pinned + device, [n_pairs * res_num] 64-bit numbers), the new offsets of the pairs
    unsigned long long* h_raws; unsigned long long* d_raws; long long raws_cap; hipEvent_t ev_raws; int raws_in_flight;
    long long* d_sel_off;             // [max_pairs + 1] where every pair's kept records go (absolute), last = the new running total
    long long pending_first;          // records held when the pending asynchronous run was enqueued (the tail starts here)
    int pending_selected;             // the pending run's pairs already went through the on-stream selection with this many records per pair at most (0: not)
    double* d_poses;                // [K][7]
    // feature selection scratch (grow-only): the gathered records and their source indices
    float4* d_sel_cp; double* d_sel_nc; double* d_sel_score; long long* d_sel_idx; long long sel_cap;      // (d_sel_idx: [1 + cap], word 0 = the new running total)
    long long* h_sel; long long h_sel_cap; hipEvent_t ev_sel; int sel_in_flight;                           // pinned copy of the same block; the event of its last upload
};

__global__ void k_transform_cloud(const float4* __restrict__ in, int n, const double* __restrict__ pose, float4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // transformCloud, Estimator.cpp:1517-1546
    if (i >= n) return;
    const double t[3] = {pose[0], pose[1], pose[2]}, q[4] = {pose[3], pose[4], pose[5], pose[6]};
    const float4 p = in[i];
    const double pin[3] = {(double)p.x, (double)p.y, (double)p.z};
    double po[3];
    a_qrot(q, pin, po);
    out[i] = make_float4((float)(po[0] + t[0]), (float)(po[1] + t[1]), (float)(po[2] + t[2]), p.w);
}

// ---- the per-keyframe voxel hashes of the batch association, BA_FB keyframes per launch (blockIdx.y = keyframe of the batch).  One keyframe at a time
// the five build kernels are launches of 128 workgroups that last 4-20 us each: 10 000 of them for 2000 keyframes, 80 of the 370 ms the association of
// every pair of a C4-sized batch took.  Same device code as the single-frame kernels above, addressed through a descriptor per keyframe.
#define BA_FB 64
struct FrameBuild {
    unsigned long long* keys; int4* ent; uint4* sub; unsigned* cnt8; int* pt_slot; int* pt_rank; float4* sorted;
    const float4* local; float4* global; const double* pose; int* total; int n, tc;
};
__global__ void k_hash_clear_multi(const FrameBuild* __restrict__ fb) {
    const FrameBuild f = fb[blockIdx.y];
    hash_clear_slot(blockIdx.x * blockDim.x + threadIdx.x, f.tc, f.keys, f.cnt8, f.total, f.ent, f.sub, f.n == 0 ? 1 : 0);
}
__global__ void k_transform_cloud_multi(const FrameBuild* __restrict__ fb) {
    const FrameBuild f = fb[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // transformCloud, Estimator.cpp:1517-1546
    if (i >= f.n) return;
    const double t[3] = {f.pose[0], f.pose[1], f.pose[2]}, q[4] = {f.pose[3], f.pose[4], f.pose[5], f.pose[6]};
    const float4 p = f.local[i];                                  // (the PRESORTED copy of the keyframe cloud: w = index in the cloud; consecutive points are neighbours,
    const double pin[3] = {(double)p.x, (double)p.y, (double)p.z};    //  which is what lets the tile insert below aggregate its atomics)
    double po[3];
    a_qrot(q, pin, po);
    f.global[i] = make_float4((float)(po[0] + t[0]), (float)(po[1] + t[1]), (float)(po[2] + t[2]), p.w);
}
__global__ __launch_bounds__(HI_THREADS) void k_hash_insert_multi(const FrameBuild* __restrict__ fb, const float inv_cell) {
    __shared__ HashInsertLds L;
    const FrameBuild f = fb[blockIdx.y];
    if ((int)(blockIdx.x * HI_THREADS) >= f.n) return;
    hash_insert_tile(L, f.global, f.n, blockIdx.x * HI_THREADS, inv_cell, f.keys, f.cnt8, f.pt_slot, f.pt_rank, f.tc);
}
__global__ __launch_bounds__(1024) void k_cell_alloc_multi(const FrameBuild* __restrict__ fb) {
    __shared__ int s_w[16], s_base;
    const FrameBuild f = fb[blockIdx.y];
    if (blockIdx.x * 1024 >= f.tc) return;
    cell_alloc_slot(blockIdx.x * 1024 + threadIdx.x, f.tc, f.cnt8, f.total, f.keys, f.ent, f.sub, s_w, &s_base);
}
__global__ void k_scatter_multi(const FrameBuild* __restrict__ fb) {
    const FrameBuild f = fb[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < f.n) scatter_point(f.global[i], i, f.pt_slot, f.pt_rank, f.cnt8, f.ent, f.sorted, __float_as_int(f.global[i].w));
}

// Compaction of the kept records, pair major, for a CHUNK of pairs per launch (blockIdx.y / wavefront = pair of the chunk):
// the per-workgroup kept counts of every pair are scanned by one wavefront, one thread then threads the running total
// through the chunk in pair order, so the constraint arrays come out exactly as with one launch per pair.
#ifndef BA_CHUNK
#define BA_CHUNK 128        /* pairs per launch: 32 -> 128 took the association of 24 000 pairs from 289 to 272 ms (fewer tails); 436 MB of per-query work arrays at 32k points */
#endif
__global__ __launch_bounds__(1024) void k_scan_pairs(const int* __restrict__ bcount, const int b_stride, const FrameDesc* __restrict__ frames,
                                                     const int* __restrict__ pair_ci, const int pair0, const int np, int* __restrict__ boff,
                                                     long long* run, long long* pair_off, const long long max_con, int* overflow) {
    __shared__ int tot[BA_CHUNK];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int q = wv; q < np; q += 16) {
        const int n = frames[pair_ci[pair0 + q]].n, nblk = (n + PF_BLOCK - 1) / PF_BLOCK;
        int carry = 0;
        for (int i0 = 0; i0 < nblk; i0 += 64) {
            const int i = i0 + lane;
            const int v = i < nblk ? bcount[(size_t)q * b_stride + i] : 0;
            int incl = v;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
            if (i < nblk) boff[(size_t)q * b_stride + i] = carry + incl - v;
            carry += __shfl(incl, 63, 64);
        }
        if (lane == 0) tot[q] = carry;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long r = *run;
        for (int q = 0; q < np; ++q) {
            long long t = tot[q];
            if (r + t > max_con) { *overflow = 1; t = 0; }
            pair_off[pair0 + q] = r;
            r += t;
        }
        pair_off[pair0 + np] = r;             // overwritten by the next chunk with the same value
        *run = r;
    }
}
__global__ void k_compact_pairs(const int* __restrict__ flag, const int* __restrict__ lpos, const int* __restrict__ boff, const int w_stride,
                                const int b_stride, const FrameDesc* __restrict__ frames, const int* __restrict__ pair_ci, const int pair0,
                                const long long* __restrict__ pair_off, const float4* __restrict__ q_cp, const double* __restrict__ q_nc,
                                const double* __restrict__ q_score, float4* __restrict__ o_cp, double* __restrict__ o_nc,
                                double* __restrict__ o_score) {
    const int q = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = frames[pair_ci[pair0 + q]].n;
    const size_t w = (size_t)q * w_stride + i;
    if (i >= n || !flag[w]) return;
    const long long p0 = pair_off[pair0 + q];
    if (pair_off[pair0 + q + 1] == p0) return;                   // overflow: nothing is written for this pair
    const long long p = p0 + boff[(size_t)q * b_stride + i / PF_BLOCK] + lpos[w];
    o_cp[p] = q_cp[w]; o_score[p] = q_score[w];
#pragma unroll
    for (int k = 0; k < 6; ++k) o_nc[6 * p + k] = q_nc[6 * w + k];
}

#define BA_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { glio_set_error("%s failed: %s", #expr, hipGetErrorString(e_)); return GLIO_E_HIP; } } while (0)

extern "C" {

// statistics of the near-block search since the last call (test / profiling hook): out[8] = units, units whose block did not fit, units handed on as units,
// uncertified queries, queries, certified queries, staged candidates, scan quads (per wavefront); enable = 1 starts counting, 0 stops
int glio_debug_knn_stats(int enable, unsigned long long* out8) {
    if (out8) {
        if (!g_knn_dbg) { memset(out8, 0, 64); }
        else { if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out8, g_knn_dbg, 64, hipMemcpyDeviceToHost) != hipSuccess) return GLIO_E_HIP; }
    }
    if (enable && !g_knn_dbg) { if (hipMalloc((void**)&g_knn_dbg, 64) != hipSuccess) return GLIO_E_HIP; }
    if (g_knn_dbg) hipMemset(g_knn_dbg, 0, 64);
    if (!enable && g_knn_dbg) { hipFree(g_knn_dbg); g_knn_dbg = nullptr; }
    return GLIO_OK;
}

// test knob: size of the cell table of the merged-window grouping (0 = sized from the map); a tiny table forces the orphan path
int glio_debug_set_gbin_cap(int cap) { g_gbin_cap = cap < 0 ? 0 : cap; return GLIO_OK; }
// test knob: 0 = every run hashes its search frames at their poses (the only mode before round 6), 1 = tables in the frames' own frames where the rule allows
int glio_debug_set_bassoc_local(int on) { g_bassoc_local = on ? 1 : 0; return GLIO_OK; }
int glio_debug_set_knn_mode(int mode) {
    if (mode < 0 || mode > 3) return GLIO_E_ARG;
    g_knn_mode = mode;
    return GLIO_OK;
}

int glio_bassoc_create(int device, int K, int max_points_per_frame, int64_t max_constraints, glio_bassoc** out) {
    if (!out || K < 2 || max_points_per_frame < 1 || max_constraints < 1) return GLIO_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { glio_set_error("no HIP device %d", device); return GLIO_E_HIP; }
    BA_CHECK(hipSetDevice(device));
    glio_bassoc* b = new glio_bassoc();
    memset(b, 0, sizeof *b);
    b->device = device; b->K = K; b->cap = max_points_per_frame; b->max_con = max_constraints;
    {   // LOW priority: the pair searches are wide, throughput-bound launches that a keyframe call enqueues beside its own latency-bound kernels (marginalization,
        // local map, the solve's one-CU steps on the context's stream) -- those must get their CUs first (GLIO_BASSOC_PRIORITY=0: default priority, for A/B)
        int least = 0, greatest = 0;
        const char* e = getenv("GLIO_BASSOC_PRIORITY");
        if ((!e || atoi(e) != 0) && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
            BA_CHECK(hipStreamCreateWithPriority(&b->stream, hipStreamNonBlocking, least));
        else BA_CHECK(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    }
    b->cell = fmaxf(1.25f, sqrtf(1.5f) * 1.0001f);
    b->inv_cell = 1.0f / b->cell;
    const size_t cap = (size_t)b->cap;
    BA_CHECK(hipMalloc((void**)&b->d_local, (size_t)K * cap * 16)); BA_CHECK(hipMalloc((void**)&b->d_global, (size_t)BA_FB * cap * 16));
    BA_CHECK(hipMalloc((void**)&b->d_local_ps, (size_t)K * cap * 16));
    BA_CHECK(hipMalloc((void**)&b->d_sorted_local, (size_t)K * cap * 16));
    b->inbox_bytes = (size_t)K * (7 * 8 + sizeof(FrameBuild) + sizeof(FrameDesc)) + (size_t)2 * 4 * 16 * K + 64;      // (pair arrays: 2 x up to 16 K pairs)
    BA_CHECK(hipMalloc((void**)&b->d_inbox, b->inbox_bytes)); BA_CHECK(hipHostMalloc((void**)&b->h_inbox, b->inbox_bytes));
    b->h_n = new int[K]();
    b->frames = new FrameHash[K]();
    const int tc = next_pow2(2 * b->cap);
    for (int k = 0; k < K; ++k) {
        FrameHash& f = b->frames[k];
        f.table_cap = tc;
        f.d_sorted_local = b->d_sorted_local + (size_t)k * cap; f.local_valid = 0;
        BA_CHECK(hipMalloc((void**)&f.d_ent, (size_t)tc * 16)); BA_CHECK(hipMalloc((void**)&f.d_sub, (size_t)tc * 16)); BA_CHECK(hipMalloc((void**)&f.d_sorted, cap * 16));
        // (a keyframe that never serves as a search frame keeps an EMPTY table: the probes of a stray pair end at once)
        BA_CHECK(hipMemsetAsync(f.d_ent, 0xff, (size_t)tc * 16, b->stream));
    }
    BA_CHECK(hipMalloc((void**)&b->d_total, (size_t)BA_FB * 4));
    BA_CHECK(hipMalloc((void**)&b->d_bkeys, (size_t)BA_FB * tc * 8)); BA_CHECK(hipMalloc((void**)&b->d_bcnt8, (size_t)BA_FB * tc * 32));
    BA_CHECK(hipMalloc((void**)&b->d_bslot, (size_t)BA_FB * cap * 4)); BA_CHECK(hipMalloc((void**)&b->d_brank, (size_t)BA_FB * cap * 4));
    BA_CHECK(hipMalloc((void**)&b->d_fb, (size_t)2 * K * sizeof(FrameBuild))); BA_CHECK(hipHostMalloc((void**)&b->h_fb, (size_t)2 * K * sizeof(FrameBuild)));
    // dense per-query work arrays for a chunk of BA_CHUNK pairs (104 B per query and pair)
    const size_t wc = cap * BA_CHUNK;
    b->b_stride = (int)(cap / PF_BLOCK + 2);
    BA_CHECK(hipMalloc((void**)&b->d_q_cp, wc * 16)); BA_CHECK(hipMalloc((void**)&b->d_q_nc, wc * 48)); BA_CHECK(hipMalloc((void**)&b->d_q_score, wc * 8));
    BA_CHECK(hipMalloc((void**)&b->d_q_flag, wc * 4)); BA_CHECK(hipMalloc((void**)&b->d_q_pos, wc * 4));
    BA_CHECK(hipMalloc((void**)&b->d_nn5, wc * 32));
    BA_CHECK(hipMalloc((void**)&b->d_bcount, (size_t)b->b_stride * BA_CHUNK * 4)); BA_CHECK(hipMalloc((void**)&b->d_boff, (size_t)b->b_stride * BA_CHUNK * 4));
    BA_CHECK(hipMalloc((void**)&b->d_frames, (size_t)K * sizeof(FrameDesc)));
    BA_CHECK(hipMalloc((void**)&b->d_cp, (size_t)max_constraints * 16)); BA_CHECK(hipMalloc((void**)&b->d_nc, (size_t)max_constraints * 48));
    BA_CHECK(hipMalloc((void**)&b->d_score, (size_t)max_constraints * 8));
    BA_CHECK(hipMalloc((void**)&b->d_run, 16)); BA_CHECK(hipMalloc((void**)&b->d_poses, (size_t)K * 7 * 8));
    b->kb = knn_bin_create(BA_CHUNK, b->cap);
    if (!b->kb) return GLIO_E_HIP;
    BA_CHECK(hipHostMalloc((void**)&b->h_tail, 16)); BA_CHECK(hipHostMalloc((void**)&b->h_poses, (size_t)K * 7 * 8)); BA_CHECK(hipHostMalloc((void**)&b->h_fd, (size_t)K * sizeof(FrameDesc)));
    b->h_tail[0] = b->h_tail[1] = 0; b->pending_pairs = -1; b->done_pairs = -1; b->done_overflow = 0;
    BA_CHECK(hipMemsetAsync(b->d_run, 0, 16, b->stream));
    BA_CHECK(hipStreamSynchronize(b->stream));
    *out = b;
    return GLIO_OK;
}

void glio_bassoc_destroy(glio_bassoc* b) {
    if (!b) return;
    hipSetDevice(b->device);
    hipStreamSynchronize(b->stream);
    for (int k = 0; k < b->K; ++k) {
        FrameHash& f = b->frames[k];
        void* p[] = {f.d_ent, f.d_sub, f.d_sorted};
        for (void* q : p) if (q) hipFree(q);
    }
    void* p[] = {b->d_bkeys, b->d_bcnt8, b->d_bslot, b->d_brank, b->d_nn5, b->d_local, b->d_local_ps, b->d_sorted_local, b->d_global, b->d_total, b->d_q_cp, b->d_q_nc, b->d_q_score, b->d_q_flag, b->d_q_pos, b->d_bcount, b->d_boff,
                 b->d_cp, b->d_nc, b->d_score, b->d_run, b->d_poses, b->d_pair_off, b->d_frames, b->d_pair_ci, b->d_pair_cj, b->d_pair_row, b->d_row_pair,
                 b->d_sel_cp, b->d_sel_nc, b->d_sel_score, b->d_sel_idx};
    for (void* q : p) if (q) hipFree(q);
    knn_bin_destroy(b->kb);
    if (b->d_fb) hipFree(b->d_fb);
    if (b->h_fb) hipHostFree(b->h_fb);
    if (b->d_inbox) hipFree(b->d_inbox);
    if (b->h_inbox) hipHostFree(b->h_inbox);
    if (b->h_pair_off) hipHostFree(b->h_pair_off);
    if (b->h_pairs) hipHostFree(b->h_pairs);
    if (b->h_tail) hipHostFree(b->h_tail);
    if (b->h_poses) hipHostFree(b->h_poses);
    if (b->h_fd) hipHostFree(b->h_fd);
    if (b->raw_stage.d) hipFree(b->raw_stage.d);
    delete[] b->h_n; delete[] b->frames;
    if (b->ev_scan) hipEventDestroy(b->ev_scan);
    if (b->ev_sel) hipEventDestroy(b->ev_sel);
    if (b->ev_fb) hipEventDestroy(b->ev_fb);
    if (b->ev_raws) hipEventDestroy(b->ev_raws);
    if (b->h_raws) hipHostFree(b->h_raws);
    if (b->d_raws) hipFree(b->d_raws);
    if (b->d_sel_off) hipFree(b->d_sel_off);
    if (b->h_sel) hipHostFree(b->h_sel);
    hipStreamDestroy(b->stream);
    delete b;
}

static int bassoc_drain(glio_bassoc* b);
int glio_bassoc_set_frame(glio_bassoc* b, int k, const float* scan_xyzi, int n) { return glio_bassoc_set_frame_strided(b, k, scan_xyzi, n, 16, 12); }
int glio_bassoc_set_frame_strided(glio_bassoc* b, int k, const void* scan, int n, int stride_bytes, int intensity_offset) {
    if (!b || k < 0 || k >= b->K || n < 0 || n > b->cap || (n > 0 && !scan)) { glio_set_error("bad keyframe cloud (k %d, n %d)", k, n); return GLIO_E_ARG; }
    if (!glio_point_layout_ok(stride_bytes, intensity_offset)) { glio_set_error("bad point layout (stride %d, intensity at %d)", stride_bytes, intensity_offset); return GLIO_E_ARG; }
    BA_CHECK(hipSetDevice(b->device));
    { const int rf = bassoc_drain(b); if (rf != GLIO_OK) return rf; }      // (an asynchronous run may still read the clouds)
    { const int ru = glio_upload_points(b->stream, &b->raw_stage, scan, n, stride_bytes, intensity_offset, b->d_local + (size_t)k * b->cap); if (ru != GLIO_OK) return ru; }
    enqueue_presort(b->stream, b->kb, b->d_local + (size_t)k * b->cap, n, b->d_local_ps + (size_t)k * b->cap);
    BA_CHECK(hipGetLastError());
    BA_CHECK(hipStreamSynchronize(b->stream));
    b->h_n[k] = n;
    b->frames[k].local_valid = 0;
    return GLIO_OK;
}

// one association run: hashes of every search frame at the given poses, then the pairs in the caller's order.  append: the records go behind what the
// object already holds (the per-keyframe calls of batchFeatureAssociation, Estimator.cpp:3413-3432, accumulate gl_vec_surf_* this way); wait = 0: everything
// is enqueued on the object's stream and the call returns (inputs are staged in pinned memory first) -- glio_bassoc_finish picks the counts up.
// An asynchronous run has two states behind it.  DRAINED: the stream was waited for (every entry point that touches what the run reads or writes does that
// first) -- the run's counts and its overflow flag then lie in the pinned h_pair_off / h_tail.  COLLECTED: glio_bassoc_finish handed them to the caller.
// bassoc_drain only moves a run from "on the stream" to "drained": the counts of a run are reported to glio_bassoc_finish and to nobody else, and its overflow
// is reported there too (advisor finding of round 5: side calls used to consume both).
static int bassoc_drain(glio_bassoc* b) {
    if (b->pending_pairs < 0) return GLIO_OK;
    BA_CHECK(hipStreamSynchronize(b->stream));
    b->done_pairs = b->pending_pairs;
    b->done_overflow = *reinterpret_cast<int*>(&b->h_tail[1]) != 0;
    b->pending_pairs = -1;
    return GLIO_OK;
}
static int bassoc_finish(glio_bassoc* b, int64_t* pair_count_out, int64_t* total_out) {
    { const int rd = bassoc_drain(b); if (rd != GLIO_OK) return rd; }
    if (b->done_pairs < 0) { if (total_out) *total_out = b->h_tail[0]; return GLIO_OK; }
    const int n_pairs = b->done_pairs;
    b->done_pairs = -1;
    if (b->done_overflow) { b->done_overflow = 0; glio_set_error("more constraints than max_constraints (%lld)", b->max_con); return GLIO_E_ARG; }
    if (pair_count_out) for (int p = 0; p < n_pairs; ++p) pair_count_out[p] = b->h_pair_off[p + 1] - b->h_pair_off[p];
    if (total_out) *total_out = b->h_tail[0];
    return GLIO_OK;
}
// build descriptors of one batch of search frames (keyframes todo[0 .. nb), descriptor slots t0 ..): where the keyframe's hash goes, its build scratch
static void bassoc_fill_batch(glio_bassoc* b, const int* todo, const size_t t0, const int nb, int* max_tc_out, int* max_n_out) {
    int max_tc = 0, max_n = 0;
    for (int q = 0; q < nb; ++q) {
        const int k = todo[q];
        FrameHash& f = b->frames[k];
        const int n = b->h_n[k];
        int tc = next_pow2(2 * (n > 512 ? n : 512));
        if (tc > f.table_cap) tc = f.table_cap;
        f.n = n; f.cap_eff = tc;
        FrameBuild& d = b->h_fb[t0 + q];
        d.keys = b->d_bkeys + (size_t)q * f.table_cap; d.cnt8 = b->d_bcnt8 + (size_t)q * f.table_cap * 8;
        d.pt_slot = b->d_bslot + (size_t)q * b->cap; d.pt_rank = b->d_brank + (size_t)q * b->cap;
        d.ent = f.d_ent; d.sub = f.d_sub; d.sorted = f.d_sorted;
        d.local = b->d_local_ps + (size_t)k * b->cap; d.global = b->d_global + (size_t)q * b->cap; d.pose = b->d_poses + 7 * k; d.total = b->d_total + q; d.n = n; d.tc = tc;
        if (tc > max_tc) max_tc = tc;
        if (n > max_n) max_n = n;
    }
    *max_tc_out = max_tc; *max_n_out = max_n;
}
// the pinned descriptor block h_fb is the source of asynchronous copies: it is rewritten only after the last of them has left it
static int bassoc_fb_reusable(glio_bassoc* b) {
    if (b->fb_in_flight) { BA_CHECK(hipEventSynchronize(b->ev_fb)); b->fb_in_flight = 0; }
    return GLIO_OK;
}
static int bassoc_fb_uploaded(glio_bassoc* b) {
    if (!b->ev_fb) BA_CHECK(hipEventCreateWithFlags(&b->ev_fb, hipEventDisableTiming));
    BA_CHECK(hipEventRecord(b->ev_fb, b->stream));
    b->fb_in_flight = 1;
    return GLIO_OK;
}
// ---- LOCAL mode.  A run hashes every search frame at its pose of the run: 12 builds (clear, tile insert, cell allocation, scatter: 127 us of the 490 us chain)
// in every keyframe call, for clouds that never change.  In local mode a keyframe's table is built ONCE, from its local cloud; a run re-poses the table's
// points (one streaming launch for all search frames) and groups every pair's queries by their cell in the search frame's own frame (k_qbin_tile,
// AssocArgs::local_tables) -- the pairs of a keyframe can then no longer share one grouping.  Used for runs with fewer than 16 pairs per search frame
// (GLIO_BASSOC_LOCAL_RATIO), which covers a keyframe call's 2 x search_range pairs AND the batch stage's pair list (12 per keyframe: its first run builds each
// table once in either mode and measured equal, 50.9 against 51.2 ms for 4656 pairs of 32 k points; every later run over the same clouds -- the rounds'
// re-association -- skips the builds: 48.0 against 50.0 ms).  Same records, bit for bit: the candidates of a query
// are the 27 cells around its cell in either frame, both cover the ball of the search radius, and distances, ranking and gate read the same global floats.
static bool bassoc_local_mode(const glio_bassoc* b, const int n_pairs, const int n_need, const double* poses, const std::vector<char>& need) {
    if (g_bassoc_local < 0) g_bassoc_local = (getenv("GLIO_BASSOC_LOCAL_TABLES") && atoi(getenv("GLIO_BASSOC_LOCAL_TABLES")) == 0) ? 0 : 1;
    static const int ratio = getenv("GLIO_BASSOC_LOCAL_RATIO") ? atoi(getenv("GLIO_BASSOC_LOCAL_RATIO")) : 16;
    if (!g_bassoc_local || g_knn_mode != 0 || n_need == 0 || n_pairs >= ratio * n_need) return false;
    if (poses)       // (the grouping inverts the pose with the conjugate: unit quaternions only; anything else takes the global mode)
        for (int k = 0; k < b->K; ++k) if (need[k]) {
            const double* q = poses + 7 * k + 3;
            const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
            if (!(fabs(n2 - 1.0) < 1e-9)) return false;
        }
    return true;
}
// the local tables that are missing among `need`: descriptors h_fb[K + ..] (the caller has made the pinned block reusable), upload, build.  *wrote: the block was written
static int bassoc_build_local(glio_bassoc* b, const std::vector<char>& need, bool* wrote) {
    std::vector<int> todo;
    for (int k = 0; k < b->K; ++k) if (need[k] && !b->frames[k].local_valid) todo.push_back(k);
    for (size_t t0 = 0; t0 < todo.size(); t0 += BA_FB) {
        const int nb = (int)std::min<size_t>(BA_FB, todo.size() - t0);
        int max_tc = 0, max_n = 0;
        bassoc_fill_batch(b, todo.data() + t0, (size_t)b->K + t0, nb, &max_tc, &max_n);
        for (int q = 0; q < nb; ++q) {
            FrameBuild& d = b->h_fb[(size_t)b->K + t0 + q];
            const int k = todo[t0 + q];
            d.global = b->d_local_ps + (size_t)k * b->cap;            // (what is hashed: the presorted LOCAL cloud itself; nothing is transformed)
            d.sorted = b->frames[k].d_sorted_local;
        }
        const FrameBuild* dfb = b->d_fb + b->K + t0;
        BA_CHECK(hipMemcpyAsync(b->d_fb + b->K + t0, b->h_fb + b->K + t0, (size_t)nb * sizeof(FrameBuild), hipMemcpyHostToDevice, b->stream));
        *wrote = true;
        hipLaunchKernelGGL(k_hash_clear_multi, dim3((max_tc + 255) / 256, nb), dim3(256), 0, b->stream, dfb);
        if (max_n > 0) {
            hipLaunchKernelGGL(k_hash_insert_multi, dim3((max_n + HI_THREADS - 1) / HI_THREADS, nb), dim3(HI_THREADS), 0, b->stream, dfb, b->inv_cell);
            hipLaunchKernelGGL(k_cell_alloc_multi, dim3((max_tc + 1023) / 1024, nb), dim3(1024), 0, b->stream, dfb);
            hipLaunchKernelGGL(k_scatter_multi, dim3((max_n + 255) / 256, nb), dim3(256), 0, b->stream, dfb);
        }
        for (int q = 0; q < nb; ++q) b->frames[todo[t0 + q]].local_valid = 1;
    }
    return GLIO_OK;
}
// What a run does before it needs the poses: the build descriptors of its (first batch of) search frames go to the device and their hash tables are
// cleared.  A caller that knows its pairs before it knows its poses (batchFeatureAssociation of a keyframe call: the pairs follow from the keyframe
// count, the poses from the solve) calls this first; the run that follows with the same search frames skips both (0.02 ms off the start of the
// searches).  Anything else in between just makes the run do them itself.
extern "C" int glio_bassoc_prepare_async(glio_bassoc* b, int n_pairs, const int32_t* pair_ci, const int32_t* pair_cj) {
    if (!b || n_pairs < 0 || (n_pairs > 0 && (!pair_ci || !pair_cj))) return GLIO_E_ARG;
    BA_CHECK(hipSetDevice(b->device));
    { const int rf = bassoc_drain(b); if (rf != GLIO_OK) return rf; }          // (an earlier asynchronous run still uses the build scratch)
    b->prep_nb = 0;
    std::vector<char> need(b->K, 0);
    for (int p = 0; p < n_pairs; ++p) { if (pair_cj[p] < 0 || pair_cj[p] >= b->K) { glio_set_error("bad pair %d", p); return GLIO_E_ARG; } need[pair_cj[p]] = 1; }
    int todo[BA_FB], nb = 0, n_need = 0;
    for (int k = 0; k < b->K; ++k) if (need[k]) { if (nb < BA_FB) todo[nb++] = k; ++n_need; }
    if (nb == 0) return GLIO_OK;
    int max_tc = 0, max_n = 0;
    { const int rw = bassoc_fb_reusable(b); if (rw != GLIO_OK) return rw; }
    if (bassoc_local_mode(b, n_pairs, n_need, nullptr, need)) {
        // local mode: nothing of a run's tables depends on the poses -- the missing ones (the newest keyframe's) are BUILT now, the run only re-poses
        bool wrote = false;
        { const int rb = bassoc_build_local(b, need, &wrote); if (rb != GLIO_OK) return rb; }
        if (wrote) { const int rw = bassoc_fb_uploaded(b); if (rw != GLIO_OK) return rw; }
        BA_CHECK(hipGetLastError());
        return GLIO_OK;
    }
    bassoc_fill_batch(b, todo, 0, nb, &max_tc, &max_n);
    BA_CHECK(hipMemcpyAsync(b->d_fb, b->h_fb, (size_t)nb * sizeof(FrameBuild), hipMemcpyHostToDevice, b->stream));
    { const int rw = bassoc_fb_uploaded(b); if (rw != GLIO_OK) return rw; }
    hipLaunchKernelGGL(k_hash_clear_multi, dim3((max_tc + 255) / 256, nb), dim3(256), 0, b->stream, static_cast<const FrameBuild*>(b->d_fb));
    BA_CHECK(hipGetLastError());
    for (int q = 0; q < nb; ++q) { b->prep_todo[q] = todo[q]; b->prep_n[q] = b->h_n[todo[q]]; b->prep_tc[q] = b->h_fb[q].tc; b->frames[todo[q]].local_valid = 0; }
    b->prep_nb = nb;
    return GLIO_OK;
}

static int bassoc_run(glio_bassoc* b, const double* poses, int n_pairs, const int32_t* pair_ci, const int32_t* pair_cj, bool append, bool wait,
                      int64_t* pair_count_out, int64_t* total_out) {
    GLIO_TRACE("K2 glio_bassoc_run (batch association)");
    if (!b || !poses || n_pairs < 0 || (n_pairs > 0 && (!pair_ci || !pair_cj))) return GLIO_E_ARG;
    BA_CHECK(hipSetDevice(b->device));
    { const int rf = bassoc_drain(b); if (rf != GLIO_OK) return rf; }          // (an earlier asynchronous run still reads the staging buffers)
    // a drained run nobody collected: this run replaces its counts in the pinned buffers -- but its overflow must not get lost with them
    if (b->done_pairs >= 0) {
        const int ov = b->done_overflow;
        b->done_pairs = -1; b->done_overflow = 0;
        if (ov) { glio_set_error("more constraints than max_constraints (%lld) in the previous asynchronous run", b->max_con); return GLIO_E_ARG; }
    }
    for (int p = 0; p < n_pairs; ++p)
        if (pair_ci[p] < 0 || pair_ci[p] >= b->K || pair_cj[p] < 0 || pair_cj[p] >= b->K || pair_ci[p] == pair_cj[p]) { glio_set_error("bad pair %d", p); return GLIO_E_ARG; }
    if (n_pairs > b->max_pairs) {
        if (b->d_pair_off) { hipFree(b->d_pair_off); hipHostFree(b->h_pair_off); hipFree(b->d_pair_ci); hipFree(b->d_pair_cj); hipFree(b->d_pair_row); hipFree(b->d_row_pair); hipHostFree(b->h_pairs);
                             b->d_pair_off = nullptr; b->h_pair_off = nullptr; b->h_pairs = nullptr; b->d_pair_ci = b->d_pair_cj = b->d_pair_row = b->d_row_pair = nullptr; }
        b->max_pairs = n_pairs + 64;
        BA_CHECK(hipMalloc((void**)&b->d_pair_off, (size_t)(b->max_pairs + 2) * 8));
        BA_CHECK(hipMalloc((void**)&b->d_pair_ci, (size_t)b->max_pairs * 4)); BA_CHECK(hipMalloc((void**)&b->d_pair_cj, (size_t)b->max_pairs * 4));
        BA_CHECK(hipMalloc((void**)&b->d_pair_row, (size_t)b->max_pairs * 4)); BA_CHECK(hipMalloc((void**)&b->d_row_pair, (size_t)b->max_pairs * 4));
        BA_CHECK(hipHostMalloc((void**)&b->h_pair_off, (size_t)(b->max_pairs + 2) * 8));
        BA_CHECK(hipHostMalloc((void**)&b->h_pairs, (size_t)b->max_pairs * 16));
    }
    const long long first_before = append ? b->h_tail[0] : 0;      // the records held before this run (every earlier run was drained above: h_tail is settled)
    // (1) every keyframe that occurs as a search frame: cloud -> global frame -> voxel hash
    std::vector<char> need(b->K, 0);
    for (int p = 0; p < n_pairs; ++p) need[pair_cj[p]] = 1;
    int n_need = 0;
    for (int k = 0; k < b->K; ++k) n_need += need[k] ? 1 : 0;
    const bool local = bassoc_local_mode(b, n_pairs, n_need, poses, need);
    // what the kernels of this run index: the object's tables, or (local mode) the run's one block
    const double* d_poses_run = b->d_poses;
    const FrameDesc* d_frames_run = b->d_frames;
    const int* d_ci_run = b->d_pair_ci; const int* d_cj_run = b->d_pair_cj;
    if (local) {
        // [poses K x 7 | re-posing descriptors K | frame descriptors K | pair_ci | pair_cj]: filled, sent once (the previous run was drained above: the pinned
        // block is free)
        const size_t o_fb = (size_t)b->K * 56, o_fd = o_fb + (size_t)b->K * sizeof(FrameBuild), o_ci = o_fd + (size_t)b->K * sizeof(FrameDesc), o_cj = o_ci + (size_t)n_pairs * 4;
        const size_t used = o_cj + (size_t)n_pairs * 4;
        if (used > b->inbox_bytes) { glio_set_error("batch association: run block too small"); return GLIO_E_STATE; }
        double* hp = reinterpret_cast<double*>(b->h_inbox);
        FrameBuild* hfb = reinterpret_cast<FrameBuild*>(b->h_inbox + o_fb);
        FrameDesc* hfd = reinterpret_cast<FrameDesc*>(b->h_inbox + o_fd);
        int32_t* hci = reinterpret_cast<int32_t*>(b->h_inbox + o_ci); int32_t* hcj = reinterpret_cast<int32_t*>(b->h_inbox + o_cj);
        d_poses_run = reinterpret_cast<const double*>(b->d_inbox);
        d_frames_run = reinterpret_cast<const FrameDesc*>(b->d_inbox + o_fd);
        d_ci_run = reinterpret_cast<const int*>(b->d_inbox + o_ci); d_cj_run = reinterpret_cast<const int*>(b->d_inbox + o_cj);
        b->prep_nb = 0;
        { const int rw = bassoc_fb_reusable(b); if (rw != GLIO_OK) return rw; }
        bool wrote = false;
        { const int rb = bassoc_build_local(b, need, &wrote); if (rb != GLIO_OK) return rb; }      // (only what glio_bassoc_prepare_async has not built already)
        if (wrote) { const int rw = bassoc_fb_uploaded(b); if (rw != GLIO_OK) return rw; }
        memcpy(hp, poses, (size_t)b->K * 56);
        // every search frame's points at its pose of this run, in its table's order: one launch (k_transform_cloud_multi reads .local, writes .global)
        int nt = 0, max_n = 0;
        for (int k = 0; k < b->K; ++k) {
            FrameDesc& fd = hfd[k];
            fd.ent = b->frames[k].d_ent; fd.sub = b->frames[k].d_sub; fd.sorted = b->frames[k].d_sorted; fd.n = b->h_n[k];
            fd.cap_eff = need[k] ? b->frames[k].cap_eff : b->frames[k].table_cap;
            if (!need[k]) continue;
            FrameBuild& d = hfb[nt++];
            memset(&d, 0, sizeof d);
            d.local = b->frames[k].d_sorted_local; d.global = b->frames[k].d_sorted; d.pose = d_poses_run + 7 * k; d.n = b->h_n[k];
            if (d.n > max_n) max_n = d.n;
        }
        for (int p = 0; p < n_pairs; ++p) { hci[p] = pair_ci[p]; hcj[p] = pair_cj[p]; }
        BA_CHECK(hipMemcpyAsync(b->d_inbox, b->h_inbox, used, hipMemcpyHostToDevice, b->stream));
        if (append) BA_CHECK(hipMemsetAsync(b->d_run + 1, 0, 8, b->stream));
        else BA_CHECK(hipMemsetAsync(b->d_run, 0, 16, b->stream));
        if (max_n > 0) hipLaunchKernelGGL(k_transform_cloud_multi, dim3((max_n + 255) / 256, nt), dim3(256), 0, b->stream, reinterpret_cast<const FrameBuild*>(b->d_inbox + o_fb));
    } else {
        memcpy(b->h_poses, poses, (size_t)b->K * 7 * 8);
        BA_CHECK(hipMemcpyAsync(b->d_poses, b->h_poses, (size_t)b->K * 7 * 8, hipMemcpyHostToDevice, b->stream));
        if (append) BA_CHECK(hipMemsetAsync(b->d_run + 1, 0, 8, b->stream));
        else BA_CHECK(hipMemsetAsync(b->d_run, 0, 16, b->stream));
        std::vector<int> todo;
        for (int k = 0; k < b->K; ++k) if (need[k]) { todo.push_back(k); b->frames[k].local_valid = 0; }      // (the global build overwrites the table)
        bool fb_checked = false, fb_written = false;
        for (size_t t0 = 0; t0 < todo.size(); t0 += BA_FB) {
            const int nb = (int)std::min<size_t>(BA_FB, todo.size() - t0);
            int max_tc = 0, max_n = 0;
            // glio_bassoc_prepare_async has sent this batch's descriptors and cleared its tables already (same keyframes, same sizes, nothing run since)?  The
            // answer follows from what was prepared alone -- never from how long ago (advisor finding of round 5: a wall-clock test made the kernel sequence
            // depend on host timing).  A prepared batch's descriptors are NOT rebuilt: they are on the device as the preparation wrote them.
            bool prepared = t0 == 0 && b->prep_nb == nb;
            for (int q = 0; prepared && q < nb; ++q) {
                const int k = todo[q], n = b->h_n[k];
                int tc = next_pow2(2 * (n > 512 ? n : 512));
                if (tc > b->frames[k].table_cap) tc = b->frames[k].table_cap;
                prepared = b->prep_todo[q] == k && b->prep_n[q] == n && b->prep_tc[q] == tc;
            }
            b->prep_nb = 0;
            const FrameBuild* dfb = b->d_fb + t0;
            if (prepared) {
                for (int q = 0; q < nb; ++q) { if (b->prep_tc[q] > max_tc) max_tc = b->prep_tc[q]; if (b->prep_n[q] > max_n) max_n = b->prep_n[q]; }
            } else {
                if (!fb_checked) { const int rw = bassoc_fb_reusable(b); if (rw != GLIO_OK) return rw; fb_checked = true; }   // (once per run: its batches write distinct parts of h_fb)
                bassoc_fill_batch(b, todo.data() + t0, t0, nb, &max_tc, &max_n);
                BA_CHECK(hipMemcpyAsync(b->d_fb + t0, b->h_fb + t0, (size_t)nb * sizeof(FrameBuild), hipMemcpyHostToDevice, b->stream));
                fb_written = true;
                hipLaunchKernelGGL(k_hash_clear_multi, dim3((max_tc + 255) / 256, nb), dim3(256), 0, b->stream, dfb);
            }
            if (max_n == 0) continue;
            hipLaunchKernelGGL(k_transform_cloud_multi, dim3((max_n + 255) / 256, nb), dim3(256), 0, b->stream, dfb);
            hipLaunchKernelGGL(k_hash_insert_multi, dim3((max_n + HI_THREADS - 1) / HI_THREADS, nb), dim3(HI_THREADS), 0, b->stream, dfb, b->inv_cell);
            hipLaunchKernelGGL(k_cell_alloc_multi, dim3((max_tc + 1023) / 1024, nb), dim3(1024), 0, b->stream, dfb);
            hipLaunchKernelGGL(k_scatter_multi, dim3((max_n + 255) / 256, nb), dim3(256), 0, b->stream, dfb);
        }
        if (fb_written) { const int rw = bassoc_fb_uploaded(b); if (rw != GLIO_OK) return rw; }
    }
    // (2) the pairs, in the caller's (ci, cj) order, BA_CHUNK pairs per launch (blockIdx.y = pair of the chunk)
    int* d_overflow = reinterpret_cast<int*>(b->d_run + 1);
    if (n_pairs > 0) {
        int maxn = 0;
        for (int p = 0; p < n_pairs; ++p) if (b->h_n[pair_ci[p]] > maxn) maxn = b->h_n[pair_ci[p]];
        if (!local) {
        for (int k = 0; k < b->K; ++k) {
            FrameDesc& fd = b->h_fd[k];
            fd.ent = b->frames[k].d_ent; fd.sub = b->frames[k].d_sub; fd.sorted = b->frames[k].d_sorted; fd.n = b->h_n[k];
            fd.cap_eff = need[k] ? b->frames[k].cap_eff : b->frames[k].table_cap;
        }
        for (int p = 0; p < n_pairs; ++p) { b->h_pairs[p] = pair_ci[p]; b->h_pairs[b->max_pairs + p] = pair_cj[p]; }
        }
        // shared query binning: per chunk, the distinct source keyframes in order of first appearance (their pairs usually follow each other: a linear look-back
        // over the chunk's rows, at most BA_CHUNK of them)
        std::vector<int> chunk_rows((size_t)(n_pairs + BA_CHUNK - 1) / BA_CHUNK, 0);
        {
            int32_t* h_row = b->h_pairs + 2 * (size_t)b->max_pairs; int32_t* h_rp = b->h_pairs + 3 * (size_t)b->max_pairs;
            for (int p0 = 0; p0 < n_pairs; p0 += BA_CHUNK) {
                const int np = n_pairs - p0 < BA_CHUNK ? n_pairs - p0 : BA_CHUNK;
                int nr = 0;
                for (int q = 0; q < np; ++q) {
                    const int ci = pair_ci[p0 + q];
                    int r = nr - 1;
                    while (r >= 0 && pair_ci[h_rp[p0 + r]] != ci) --r;
                    if (r < 0) { r = nr++; h_rp[p0 + r] = p0 + q; }
                    h_row[p0 + q] = r;
                }
                for (int r = nr; r < np; ++r) h_rp[p0 + r] = p0;
                chunk_rows[(size_t)p0 / BA_CHUNK] = nr;
            }
        }
        static const bool share_env = !(getenv("GLIO_BASSOC_SHARED_BINS") && atoi(getenv("GLIO_BASSOC_SHARED_BINS")) == 0);      // (0: every pair bins for itself -- A/B and tests)
        const bool share_bins = share_env && !local;       // (local tables: a pair's queries are grouped by their cell in ITS search frame)
        if (!local) {
        BA_CHECK(hipMemcpyAsync(b->d_frames, b->h_fd, (size_t)b->K * sizeof(FrameDesc), hipMemcpyHostToDevice, b->stream));
        BA_CHECK(hipMemcpyAsync(b->d_pair_ci, b->h_pairs, (size_t)n_pairs * 4, hipMemcpyHostToDevice, b->stream));
        BA_CHECK(hipMemcpyAsync(b->d_pair_cj, b->h_pairs + b->max_pairs, (size_t)n_pairs * 4, hipMemcpyHostToDevice, b->stream));
        BA_CHECK(hipMemcpyAsync(b->d_pair_row, b->h_pairs + 2 * (size_t)b->max_pairs, (size_t)n_pairs * 4, hipMemcpyHostToDevice, b->stream));
        BA_CHECK(hipMemcpyAsync(b->d_row_pair, b->h_pairs + 3 * (size_t)b->max_pairs, (size_t)n_pairs * 4, hipMemcpyHostToDevice, b->stream));
        }
        AssocArgs a;
        memset(&a, 0, sizeof a);
        a.inv_cell = b->inv_cell; a.cell = b->cell; a.kd_max_radius = 1.5; a.weight_gate = 0.3; a.surf_dist_thres = 0.18; a.lidar_const = 2.5;   // :3839,3874,3863,3885
        a.unit_scores = 0;
        a.q_stride = b->cap; a.w_stride = b->cap; a.b_stride = b->b_stride;
        a.frames = d_frames_run; a.pair_ci = d_ci_run; a.pair_cj = d_cj_run; a.poses = d_poses_run;
        a.local_tables = local ? 1 : 0;
        for (int p0 = 0; p0 < n_pairs; p0 += BA_CHUNK) {
            const int np = n_pairs - p0 < BA_CHUNK ? n_pairs - p0 : BA_CHUNK;
            a.pair0 = p0;
            if (maxn > 0) {
                a.pair_row = share_bins ? b->d_pair_row : nullptr; a.row_pair = share_bins ? b->d_row_pair : nullptr;
                enqueue_knn(b->stream, a, b->kb, np, maxn, b->d_local, b->d_local_ps, (const float4*)nullptr, (const int4*)nullptr, (const uint4*)nullptr, b->cap, b->d_nn5,
                            share_bins ? chunk_rows[(size_t)p0 / BA_CHUNK] : 0);
                a.pair_row = nullptr; a.row_pair = nullptr;          // (the fit below addresses everything by the pair)
                hipLaunchKernelGGL(k_plane_fit<true>, dim3((maxn + PF_BLOCK - 1) / PF_BLOCK, np), dim3(PF_BLOCK), 0, b->stream, a, b->d_local, (const float4*)nullptr,
                                   b->d_nn5, b->d_q_cp, (float4*)nullptr, b->d_q_score, b->d_q_flag, b->d_q_pos, b->d_bcount, (int*)nullptr,
                                   b->d_local, b->d_q_nc);
            }
            hipLaunchKernelGGL(k_scan_pairs, dim3(1), dim3(1024), 0, b->stream, b->d_bcount, b->b_stride, d_frames_run, d_ci_run, p0, np, b->d_boff,
                               b->d_run, b->d_pair_off, (long long)b->max_con, d_overflow);
            if (maxn > 0)
                hipLaunchKernelGGL(k_compact_pairs, dim3((maxn + 255) / 256, np), dim3(256), 0, b->stream, b->d_q_flag, b->d_q_pos, b->d_boff, b->cap, b->b_stride,
                                   d_frames_run, d_ci_run, p0, b->d_pair_off, b->d_q_cp, b->d_q_nc, b->d_q_score, b->d_cp, b->d_nc, b->d_score);
        }
    }
    BA_CHECK(hipGetLastError());
    if (n_pairs > 0) BA_CHECK(hipMemcpyAsync(b->h_pair_off, b->d_pair_off, (size_t)(n_pairs + 1) * 8, hipMemcpyDeviceToHost, b->stream));
    BA_CHECK(hipMemcpyAsync(b->h_tail, b->d_run, 16, hipMemcpyDeviceToHost, b->stream));
    b->pending_pairs = n_pairs;
    b->pending_first = first_before;
    b->pending_selected = 0;
    return wait ? bassoc_finish(b, pair_count_out, total_out) : GLIO_OK;
}
int glio_bassoc_run(glio_bassoc* b, const double* poses, int n_pairs, const int32_t* pair_ci, const int32_t* pair_cj, int64_t* pair_count_out, int64_t* total_out) {
    return bassoc_run(b, poses, n_pairs, pair_ci, pair_cj, false, true, pair_count_out, total_out);
}
int glio_bassoc_run_append(glio_bassoc* b, const double* poses, int n_pairs, const int32_t* pair_ci, const int32_t* pair_cj, int64_t* pair_count_out, int64_t* total_out) {
    return bassoc_run(b, poses, n_pairs, pair_ci, pair_cj, true, true, pair_count_out, total_out);
}
int glio_bassoc_run_append_async(glio_bassoc* b, const double* poses, int n_pairs, const int32_t* pair_ci, const int32_t* pair_cj) {
    return bassoc_run(b, poses, n_pairs, pair_ci, pair_cj, true, false, nullptr, nullptr);
}
int glio_bassoc_finish(glio_bassoc* b, int64_t* pair_count_out, int64_t* total_out) {
    if (!b) return GLIO_E_ARG;
    BA_CHECK(hipSetDevice(b->device));
    return bassoc_finish(b, pair_count_out, total_out);
}
int glio_bassoc_reset(glio_bassoc* b) {
    if (!b) return GLIO_E_ARG;
    BA_CHECK(hipSetDevice(b->device));
    { const int rf = bassoc_drain(b); if (rf != GLIO_OK) return rf; }
    b->done_pairs = -1; b->done_overflow = 0;              // (a reset drops the records, and with them what an uncollected run had to say)
    BA_CHECK(hipMemsetAsync(b->d_run, 0, 16, b->stream));
    b->h_tail[0] = b->h_tail[1] = 0;
    return GLIO_OK;
}
// surf_frames[k] from a scan that is already resident in a sliding-window context (slot of its ring), minus the LiDAR offset: no second upload
__global__ void k_copy_offset(const float4* __restrict__ in, int n, float ox, float oy, float oz, float4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float4 p = in[i]; out[i] = make_float4(p.x - ox, p.y - oy, p.z - oz, p.w); }
}
int glio_bassoc_set_frame_from_scan(glio_bassoc* b, int k, glio_ctx* c, int slot, const float lidar_offset[3]) {
    if (!b || !c || !lidar_offset || k < 0 || k >= b->K || slot < 0 || slot >= c->W || c->device != b->device) return GLIO_E_ARG;
    const int n = c->h_scan_count[slot];
    if (n > b->cap) { glio_set_error("scan of %d points, the batch association holds %d per keyframe", n, b->cap); return GLIO_E_ARG; }
    BA_CHECK(hipSetDevice(b->device));
    { const int rf = bassoc_drain(b); if (rf != GLIO_OK) return rf; }
    // the upload of the scan (and its presort) ran on the context's stream: this stream waits for that point, the host does not
    if (!b->ev_scan) BA_CHECK(hipEventCreateWithFlags(&b->ev_scan, hipEventDisableTiming));
    BA_CHECK(hipEventRecord(b->ev_scan, c->stream));
    BA_CHECK(hipStreamWaitEvent(b->stream, b->ev_scan, 0));
    // the context presorted this scan when it was uploaded: its presorted copy (same points, w = index in the scan) is taken over with the same offset instead
    // of presorting the cloud a second time (any spatially compact order serves: results are written at the original indices)
    if (n > 0) {
        const size_t row = (size_t)glio_scan_row(c, slot) * c->cap;
        hipLaunchKernelGGL(k_copy_offset, dim3((n + 255) / 256), dim3(256), 0, b->stream, c->d_scan + row, n, lidar_offset[0], lidar_offset[1], lidar_offset[2],
                           b->d_local + (size_t)k * b->cap);
        if (c->assoc && c->assoc->d_ps)
            hipLaunchKernelGGL(k_copy_offset, dim3((n + 255) / 256), dim3(256), 0, b->stream, c->assoc->d_ps + row, n, lidar_offset[0], lidar_offset[1], lidar_offset[2],
                               b->d_local_ps + (size_t)k * b->cap);
        else enqueue_presort(b->stream, b->kb, b->d_local + (size_t)k * b->cap, n, b->d_local_ps + (size_t)k * b->cap);
        // ... and the context must not overwrite the row (glio_set_scan after W slides of the ring), nor be destroyed, before these copies have read it
        if (!c->ev_ext_read) BA_CHECK(hipEventCreateWithFlags(&c->ev_ext_read, hipEventDisableTiming));
        BA_CHECK(hipEventRecord(c->ev_ext_read, b->stream));
        c->ext_read_pending = 1;
    }
    BA_CHECK(hipGetLastError());
    b->h_n[k] = n;
    b->frames[k].local_valid = 0;
    return GLIO_OK;
}

// globalFeatureSelectionAdd_Batch (Estimator.cpp:4057-4116) ON THE STREAM, right behind the searches of an asynchronous run: no host round trip between
// the searches and the selection (waiting for the counts, drawing, uploading the indices: ~50 us at the end of every keyframe call).  The draws are still
// the caller's: `raws` holds res_num 64-bit numbers per pair, drawn before the counts are known; pair p keeps all of its `count` records when
// count <= res_num, else the first res_num of a uniform shuffle of 0 .. count - 2 (the LAST record is never drawn, random_generator.hpp:79-93): step i swaps
// position i with position i + raws[p * res_num + i] mod (count - 1 - i) -- glio::batchSelectionDraws with rand_below(n) = raw mod n.
// k_bassoc_draw: one thread per pair forms the kept source indices (absolute) and the kept count; k_bassoc_draw_off: their prefix;
// gather into the staging arrays, put back (the kernels of glio_bassoc_select_range).
__global__ __launch_bounds__(64) void k_bassoc_draw(const long long* __restrict__ pair_off, const int n_pairs, const int res_num, const unsigned long long* __restrict__ raws,
                                                    long long* __restrict__ src /* [n_pairs][res_num] */, int* __restrict__ kept) {
    // one wavefront per pair (a thread-private map indexed at run time would live in scratch memory: a dependent global round trip per look-up -- the first
    // version of this kernel took ~80 us for 12 pairs)
    const int p = blockIdx.x, lane = threadIdx.x;
    const long long o0 = pair_off[p], count = pair_off[p + 1] - o0;
    long long* out = src + (size_t)p * res_num;
    if (count <= res_num) { for (long long k = lane; k < count; k += 64) out[k] = o0 + k; if (lane == 0) kept[p] = (int)count; return; }
    // partial Fisher-Yates without the index array: a small map of the positions whose content differs from their index (<= res_num entries), ONE ENTRY PER
    // LANE; a look-up is a ballot and a shuffle.  The 64-bit remainders (the slow part: ~150 instructions each) do not depend on the shuffle's state: lane i
    // takes draw i's.  (One lane doing all of it -- remainders and linear searches of an LDS map -- took 37 us for 12 pairs at the end of every keyframe call.)
    const int niter = (int)(res_num < count - 1 ? res_num : count - 1);
    long long jmine = 0;
    if (lane < niter) jmine = lane + (long long)(raws[(size_t)p * res_num + lane] % (unsigned long long)(count - 1 - lane));
    long long mk = -1, mv = 0, outv = 0;
    int nm = 0;
    for (int i = 0; i < niter; ++i) {
        const long long j = (long long)shfl_u64((unsigned long long)jmine, i);
        const unsigned long long bi = __ballot(lane < nm && mk == i), bj = __ballot(lane < nm && mk == j);
        long long vi = i, vj = j;
        if (bi) vi = (long long)shfl_u64((unsigned long long)mv, __ffsll((long long)bi) - 1);
        if (bj) vj = (long long)shfl_u64((unsigned long long)mv, __ffsll((long long)bj) - 1);
        if (bj) { if (lane == __ffsll((long long)bj) - 1) mv = vi; }
        else { if (lane == nm) { mk = j; mv = vi; } ++nm; }
        if (lane == i) outv = o0 + vj;
    }
    if (lane == 0) kept[p] = niter;
    if (lane < niter) out[lane] = outv;
}
// the four kernels of the on-stream selection as ONE workgroup (a keyframe call's 2 x search_range pairs: <= 16, one wavefront each): draws, the prefix of the kept
// counts, the kept records into registers, a barrier (every source has been read), the records to their places.  Same draws as k_bassoc_draw, same places as
// k_bassoc_draw_off / _gather / _put (6 + 6 + 5 + 5 us of launches at the end of every keyframe call's association).
__global__ __launch_bounds__(1024) void k_bassoc_draw_all(const long long* __restrict__ pair_off, const int n_pairs, const int res_num, const unsigned long long* __restrict__ raws,
                                                          const long long first, long long* __restrict__ sel_off, long long* __restrict__ run,
                                                          float4* __restrict__ cp, double* __restrict__ nc, double* __restrict__ score) {
    __shared__ int s_kept[16];
    __shared__ long long s_off[17];
    const int p = threadIdx.x >> 6, lane = threadIdx.x & 63;
    long long sidx = -1;
    int mine = 0;
    if (p < n_pairs) {
        const long long o0 = pair_off[p], count = pair_off[p + 1] - o0;
        if (count <= res_num) { mine = (int)count; if (lane < count) sidx = o0 + lane; }
        else {
            const int niter = (int)(res_num < count - 1 ? res_num : count - 1);
            long long jmine = 0;
            if (lane < niter) jmine = lane + (long long)(raws[(size_t)p * res_num + lane] % (unsigned long long)(count - 1 - lane));
            long long mk = -1, mv = 0;
            int nm = 0;
            for (int i = 0; i < niter; ++i) {
                const long long j = (long long)shfl_u64((unsigned long long)jmine, i);
                const unsigned long long bi = __ballot(lane < nm && mk == i), bj = __ballot(lane < nm && mk == j);
                long long vi = i, vj = j;
                if (bi) vi = (long long)shfl_u64((unsigned long long)mv, __ffsll((long long)bi) - 1);
                if (bj) vj = (long long)shfl_u64((unsigned long long)mv, __ffsll((long long)bj) - 1);
                if (bj) { if (lane == __ffsll((long long)bj) - 1) mv = vi; }
                else { if (lane == nm) { mk = j; mv = vi; } ++nm; }
                if (lane == i) sidx = o0 + vj;
            }
            mine = niter;
        }
        if (lane == 0) s_kept[p] = mine;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long r = first;
        for (int q = 0; q < n_pairs; ++q) { s_off[q] = r; sel_off[q] = r; r += s_kept[q]; }
        s_off[n_pairs] = r; sel_off[n_pairs] = r;
        run[0] = r;
    }
    float4 rc = make_float4(0, 0, 0, 0);
    double rn[6] = {0, 0, 0, 0, 0, 0}, rs = 0;
    const bool live = p < n_pairs && lane < mine;
    if (live) {
        rc = cp[sidx];
#pragma unroll
        for (int c = 0; c < 6; ++c) rn[c] = nc[6 * sidx + c];
        rs = score[sidx];
    }
    __syncthreads();            // every kept record is in registers (the places written below may be another pair's sources), the offsets are known
    if (live) {
        const long long d = s_off[p] + lane;
        cp[d] = rc;
#pragma unroll
        for (int c = 0; c < 6; ++c) nc[6 * d + c] = rn[c];
        score[d] = rs;
    }
}
__global__ void k_bassoc_draw_off(const int* __restrict__ kept, const int n_pairs, const long long first, long long* __restrict__ sel_off, long long* __restrict__ run) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    long long r = first;
    for (int p = 0; p < n_pairs; ++p) { sel_off[p] = r; r += kept[p]; }
    sel_off[n_pairs] = r;
    run[0] = r;
}
__global__ void k_bassoc_draw_gather(const long long* __restrict__ src, const int* __restrict__ kept, const long long* __restrict__ sel_off, const int res_num, const long long first,
                                     const float4* __restrict__ cp, const double* __restrict__ nc, const double* __restrict__ score,
                                     float4* __restrict__ o_cp, double* __restrict__ o_nc, double* __restrict__ o_score) {
    const int p = blockIdx.x, k = threadIdx.x;
    if (k >= kept[p]) return;
    const long long sidx = src[(size_t)p * res_num + k], d = sel_off[p] - first + k;
    o_cp[d] = cp[sidx];
#pragma unroll
    for (int c = 0; c < 6; ++c) o_nc[6 * d + c] = nc[6 * sidx + c];
    o_score[d] = score[sidx];
}
__global__ void k_bassoc_draw_put(const long long* __restrict__ sel_off, const int n_pairs, const long long first, const float4* __restrict__ s_cp, const double* __restrict__ s_nc,
                                  const double* __restrict__ s_score, float4* __restrict__ cp, double* __restrict__ nc, double* __restrict__ score) {
    const long long n = sel_off[n_pairs] - first, k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    cp[first + k] = s_cp[k];
#pragma unroll
    for (int c = 0; c < 6; ++c) nc[6 * (first + k) + c] = s_nc[6 * k + c];
    score[first + k] = s_score[k];
}
int glio_bassoc_select_tail_draws_async(glio_bassoc* b, int res_num, const uint64_t* raws) {
    if (!b || res_num < 1 || res_num > 64 || !raws) return GLIO_E_ARG;
    if (b->pending_pairs < 0) { glio_set_error("glio_bassoc_select_tail_draws_async: no asynchronous run is pending"); return GLIO_E_STATE; }
    BA_CHECK(hipSetDevice(b->device));
    const int n_pairs = b->pending_pairs;
    if (n_pairs == 0) return GLIO_OK;
    const long long need = (long long)n_pairs * res_num;
    if (need > b->raws_cap) {
        if (b->raws_in_flight) { BA_CHECK(hipEventSynchronize(b->ev_raws)); b->raws_in_flight = 0; }
        if (b->h_raws) hipHostFree(b->h_raws);
        if (b->d_raws) hipFree(b->d_raws);
        if (b->d_sel_off) hipFree(b->d_sel_off);
        b->h_raws = nullptr; b->d_raws = nullptr; b->d_sel_off = nullptr; b->raws_cap = 0;
        const long long cap = need + need / 2 + 1024;
        BA_CHECK(hipHostMalloc((void**)&b->h_raws, (size_t)cap * 8)); BA_CHECK(hipMalloc((void**)&b->d_raws, (size_t)cap * 16));      // (device: the raws, then the source indices)
        BA_CHECK(hipMalloc((void**)&b->d_sel_off, (size_t)(cap + 2) * 8 + (size_t)cap * 4));                                            // (the offsets, then the kept counts)
        b->raws_cap = cap;
    }
    // the staging arrays of the selection (grow-only, shared with glio_bassoc_select_range)
    if (need > b->sel_cap || !b->d_sel_idx) {
        BA_CHECK(hipStreamSynchronize(b->stream));
        void** old[] = {(void**)&b->d_sel_cp, (void**)&b->d_sel_nc, (void**)&b->d_sel_score, (void**)&b->d_sel_idx};
        for (void** q : old) { if (*q) hipFree(*q); *q = nullptr; }
        b->sel_cap = 0;
        const int64_t cap = need + need / 2 + 1024;
        BA_CHECK(hipMalloc((void**)&b->d_sel_cp, (size_t)cap * 16)); BA_CHECK(hipMalloc((void**)&b->d_sel_nc, (size_t)cap * 48));
        BA_CHECK(hipMalloc((void**)&b->d_sel_score, (size_t)cap * 8)); BA_CHECK(hipMalloc((void**)&b->d_sel_idx, (size_t)(cap + 1) * 8));
        b->sel_cap = cap;
    }
    if (!b->ev_raws) BA_CHECK(hipEventCreateWithFlags(&b->ev_raws, hipEventDisableTiming));
    if (b->raws_in_flight) { BA_CHECK(hipEventSynchronize(b->ev_raws)); b->raws_in_flight = 0; }
    memcpy(b->h_raws, raws, (size_t)need * 8);
    BA_CHECK(hipMemcpyAsync(b->d_raws, b->h_raws, (size_t)need * 8, hipMemcpyHostToDevice, b->stream));
    BA_CHECK(hipEventRecord(b->ev_raws, b->stream));
    b->raws_in_flight = 1;
    long long* d_src = reinterpret_cast<long long*>(b->d_raws + b->raws_cap);
    int* d_kept = reinterpret_cast<int*>(b->d_sel_off + b->raws_cap + 2);
    if (n_pairs <= 16) {
        hipLaunchKernelGGL(k_bassoc_draw_all, dim3(1), dim3(64 * n_pairs), 0, b->stream, b->d_pair_off, n_pairs, res_num, b->d_raws, b->pending_first, b->d_sel_off, b->d_run,
                           b->d_cp, b->d_nc, b->d_score);
    } else {
    hipLaunchKernelGGL(k_bassoc_draw, dim3(n_pairs), dim3(64), 0, b->stream, b->d_pair_off, n_pairs, res_num, b->d_raws, d_src, d_kept);
    hipLaunchKernelGGL(k_bassoc_draw_off, dim3(1), dim3(64), 0, b->stream, d_kept, n_pairs, b->pending_first, b->d_sel_off, b->d_run);
    hipLaunchKernelGGL(k_bassoc_draw_gather, dim3(n_pairs), dim3(64), 0, b->stream, d_src, d_kept, b->d_sel_off, res_num, b->pending_first, b->d_cp, b->d_nc, b->d_score,
                       b->d_sel_cp, b->d_sel_nc, b->d_sel_score);
    hipLaunchKernelGGL(k_bassoc_draw_put, dim3((unsigned)((need + 255) / 256)), dim3(256), 0, b->stream, b->d_sel_off, n_pairs, b->pending_first, b->d_sel_cp, b->d_sel_nc, b->d_sel_score,
                       b->d_cp, b->d_nc, b->d_score);
    }
    BA_CHECK(hipGetLastError());
    // the running total as the selection left it (the copy the run enqueued carried the total BEFORE the selection; the pair counts stay the FOUND ones)
    BA_CHECK(hipMemcpyAsync(b->h_tail, b->d_run, 8, hipMemcpyDeviceToHost, b->stream));
    b->pending_selected = res_num;
    return GLIO_OK;
}

int glio_bassoc_results_dev(glio_bassoc* b, const float** cp_dev, const double** nc_dev, const double** score_dev) {
    if (!b) return GLIO_E_ARG;
    // the hand-over to another stream's consumer (glio_batch_set_constraints_pairs_dev): whatever this stream still does to the arrays is waited for here
    if (hipSetDevice(b->device) != hipSuccess || hipStreamSynchronize(b->stream) != hipSuccess) { glio_set_error("glio_bassoc_results_dev: stream"); return GLIO_E_HIP; }
    if (cp_dev) *cp_dev = reinterpret_cast<const float*>(b->d_cp);
    if (nc_dev) *nc_dev = b->d_nc;
    if (score_dev) *score_dev = b->d_score;
    return GLIO_OK;
}

// globalFeatureSelectionAdd_Batch / globalFeatureSelection_Batch (Estimator.cpp:4057-4116, 3994-4055): keep, per keyframe pair, the
// records the caller drew (the reference seeds from std::random_device -- random_generator.hpp:58 -- so the draws stay with the
// caller, glio_amd/batch.py::batch_selection_draws restates the rules); the gather runs on the device and the pair-major arrays
// glio_batch_set_constraints_pairs_dev consumes are compacted in place.  src_index [n_keep]: indices into the CURRENT arrays,
// in the order the kept records shall have (pair after pair).
__global__ void k_bassoc_gather(const long long* __restrict__ idx, const long long n, const float4* __restrict__ cp, const double* __restrict__ nc,
                                const double* __restrict__ score, float4* __restrict__ o_cp, double* __restrict__ o_nc, double* __restrict__ o_score) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const long long sidx = idx[k];
    o_cp[k] = cp[sidx];
#pragma unroll
    for (int c = 0; c < 6; ++c) o_nc[6 * k + c] = nc[6 * sidx + c];
    o_score[k] = score[sidx];
}
int glio_bassoc_select(glio_bassoc* b, int64_t n_keep, const int64_t* src_index, int64_t n_current) { return glio_bassoc_select_range(b, 0, n_keep, src_index, n_current); }
// the gathered records back to [first, first + n) of the arrays, and the running total (word 0 of the uploaded block) to its place
__global__ void k_bassoc_put(const long long first, const long long n, const float4* __restrict__ s_cp, const double* __restrict__ s_nc, const double* __restrict__ s_score,
                             float4* __restrict__ cp, double* __restrict__ nc, double* __restrict__ score, const long long* __restrict__ blk, long long* __restrict__ run) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) run[0] = blk[0];
    if (k >= n) return;
    cp[first + k] = s_cp[k];
#pragma unroll
    for (int c = 0; c < 6; ++c) nc[6 * (first + k) + c] = s_nc[6 * k + c];
    score[first + k] = s_score[k];
}

// the same over the TAIL [first, n_current) only (the records one keyframe's batchFeatureAssociation appended): they are replaced by the n_keep records
// src_index names (absolute indices >= first); the object then holds first + n_keep records
int glio_bassoc_select_range(glio_bassoc* b, int64_t first, int64_t n_keep, const int64_t* src_index, int64_t n_current) {
    if (!b || first < 0 || n_keep < 0 || n_current < first || n_current > b->max_con || n_keep > n_current - first || (n_keep > 0 && !src_index)) return GLIO_E_ARG;
    BA_CHECK(hipSetDevice(b->device));
    { const int rf = bassoc_drain(b); if (rf != GLIO_OK) return rf; }
    for (int64_t k = 0; k < n_keep; ++k) if (src_index[k] < first || src_index[k] >= n_current) { glio_set_error("selection index %lld out of range", (long long)src_index[k]); return GLIO_E_ARG; }
    // ONE pinned block [new running total | the indices] -> one copy, the gather into the staging arrays, one kernel that puts the records (and the total) in
    // place; nothing is waited for: every later consumer is on this stream or synchronises it (glio_bassoc_read, glio_bassoc_results_dev).  (It used to be
    // two pageable copies, the gather, three device copies and a synchronisation: ~60 us at the end of every keyframe call.)
    if (n_keep > b->sel_cap || !b->d_sel_idx) {
        // (pointers nulled and the capacity reset before the new allocations: a failed hipMalloc must not leave dangling pointers
        //  behind a capacity that claims room -- advisor finding of round 2)
        BA_CHECK(hipStreamSynchronize(b->stream));
        void** old[] = {(void**)&b->d_sel_cp, (void**)&b->d_sel_nc, (void**)&b->d_sel_score, (void**)&b->d_sel_idx};
        for (void** q : old) { if (*q) hipFree(*q); *q = nullptr; }
        b->sel_cap = 0;
        const int64_t cap = n_keep + n_keep / 2 + 1024;
        BA_CHECK(hipMalloc((void**)&b->d_sel_cp, (size_t)cap * 16)); BA_CHECK(hipMalloc((void**)&b->d_sel_nc, (size_t)cap * 48));
        BA_CHECK(hipMalloc((void**)&b->d_sel_score, (size_t)cap * 8)); BA_CHECK(hipMalloc((void**)&b->d_sel_idx, (size_t)(cap + 1) * 8));
        b->sel_cap = cap;
    }
    if (b->sel_in_flight) { BA_CHECK(hipEventSynchronize(b->ev_sel)); b->sel_in_flight = 0; }      // (the previous selection's block has left the pinned copy)
    if (n_keep + 1 > b->h_sel_cap) {
        if (b->h_sel) hipHostFree(b->h_sel);
        b->h_sel = nullptr; b->h_sel_cap = 0;
        const int64_t cap = n_keep + n_keep / 2 + 1024;
        BA_CHECK(hipHostMalloc((void**)&b->h_sel, (size_t)cap * 8));
        b->h_sel_cap = cap;
    }
    if (!b->ev_sel) BA_CHECK(hipEventCreateWithFlags(&b->ev_sel, hipEventDisableTiming));
    b->h_tail[0] = first + n_keep;
    b->h_sel[0] = first + n_keep;
    if (n_keep > 0) memcpy(b->h_sel + 1, src_index, (size_t)n_keep * 8);
    BA_CHECK(hipMemcpyAsync(b->d_sel_idx, b->h_sel, (size_t)(n_keep + 1) * 8, hipMemcpyHostToDevice, b->stream));
    BA_CHECK(hipEventRecord(b->ev_sel, b->stream));
    b->sel_in_flight = 1;
    if (n_keep > 0)
        hipLaunchKernelGGL(k_bassoc_gather, dim3((unsigned)((n_keep + 255) / 256)), dim3(256), 0, b->stream, b->d_sel_idx + 1, (long long)n_keep, b->d_cp, b->d_nc, b->d_score,
                           b->d_sel_cp, b->d_sel_nc, b->d_sel_score);
    hipLaunchKernelGGL(k_bassoc_put, dim3((unsigned)((std::max<int64_t>(n_keep, 1) + 255) / 256)), dim3(256), 0, b->stream, (long long)first, (long long)n_keep, b->d_sel_cp, b->d_sel_nc,
                       b->d_sel_score, b->d_cp, b->d_nc, b->d_score, b->d_sel_idx, b->d_run);
    BA_CHECK(hipGetLastError());
    return GLIO_OK;
}

int glio_bassoc_read(glio_bassoc* b, int64_t first, int64_t n, float* cp, double* norm_cent, double* score) {
    if (!b || first < 0 || n < 0 || first + n > b->max_con) return GLIO_E_ARG;
    BA_CHECK(hipSetDevice(b->device));
    { const int rf = bassoc_drain(b); if (rf != GLIO_OK) return rf; }
    BA_CHECK(hipStreamSynchronize(b->stream));                       // (a selection may still be compacting the arrays)
    if (n == 0) return GLIO_OK;
    if (cp) BA_CHECK(hipMemcpy(cp, b->d_cp + first, (size_t)n * 16, hipMemcpyDeviceToHost));
    if (norm_cent) BA_CHECK(hipMemcpy(norm_cent, b->d_nc + 6 * first, (size_t)n * 48, hipMemcpyDeviceToHost));
    if (score) BA_CHECK(hipMemcpy(score, b->d_score + first, (size_t)n * 8, hipMemcpyDeviceToHost));
    return GLIO_OK;
}

}  // extern "C"
