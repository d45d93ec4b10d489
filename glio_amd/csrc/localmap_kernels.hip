// localmap_kernels.hip -- device-resident local surf map.
//
// Replaces, for the sliding-window path, buildLocalMapWithLandMark + downSampleCloud + the kd-tree input
// (reference GLIO/src/Estimator.cpp:3529-3631, 2056): the last `local_map_width` keyframe clouds are kept ON THE
// DEVICE in the map frame (transformCloud(surf_frames[idx], q_po * q_bl, q_po * t_bl + t_po), :3569-3574), so that per
// keyframe only ONE new scan (16 B/point) crosses PCIe instead of the whole down-sampled map; the concatenation is
// voxel-averaged on the device (pcl::VoxelGrid semantics: bounding box, floor(p * inv_leaf) - min_b, centroid per
// voxel, output ordered by linear voxel index) and handed to K1 without leaving HBM.
//
// The ring holds width x cap x 16 B (50 x 64k points = 52 MB).  The voxel table is PERSISTENT and keyed by absolute voxel
// coordinates; sums are kept in 2^-20 m fixed point (int64 atomics), i.e. they are exact integers: pushing a keyframe adds
// its points, evicting the oldest subtracts them again and the table is exactly what a rebuild from scratch would give
// (bit for bit, whatever the atomic order) -- insert/evict costs two keyframes of traffic instead of re-streaming the
// whole ring (3.3 M points).  Emptied voxels stay as tombstones (count 0) until they fill half the table, then the table
// is rebuilt once.  PCL's float accumulation (whose order std::sort leaves unspecified) is matched to ~1e-6 m.
#include <cfloat>
#include <algorithm>
#include <cstring>
#include <chrono>

#include "glio_device.h"

// the transform must round like the reference's scalar code (and the oracle): no FMA contraction in this file
#pragma clang fp contract(off)

struct LocalMap {
    int width, cap;                 // keyframes in the ring, points per keyframe
    float leaf;
    float4* d_ring;                 // [width][cap] clouds in the map frame
    int* h_n;                       // [width] points per ring entry
    int* d_n;                       // [width] the same, on the device (grid.y = ring slot)
    int head, count;                // ring: entries [head, head+count) modulo width, oldest first
    long long pushed;               // keyframes pushed so far
    // voxel grid workspace
    int table_cap;                  // power of two >= 2 * max voxels
    unsigned long long* d_keys;     // voxel linear index or EMPTY
    long long* d_sum;               // [table_cap][4] fixed-point sums x,y,z,intensity
    int* d_cnt;                     // [table_cap]
    int* d_bbox;                    // [6] ordered-int encoded min xyz / max xyz of the whole ring
    int* d_slot_bbox;               // [width][6] the same per ring slot
    int* d_nkeys;                   // [1] voxel keys ever claimed in the table (live + tombstones)
    int nkeys_seen;                 // host copy after the last build
    int* d_nvox;                    // [1]
    unsigned long long* d_vkey; unsigned long long* d_vkey_sorted; int* d_vslot; int* d_vslot_sorted;
    void* d_sort_tmp; size_t sort_tmp_bytes;
    float4* d_out;                  // [max voxels] down-sampled map, ordered by voxel index
    int max_vox;
    int* h_pin;                     // pinned scalar read-back
    unsigned* d_bm; int* d_bm_pre; int* d_bm_blk; int* d_bm_over; int bm_dirty; int force_sort;      // occupancy bitmap over the bounding box (ordered output without a sort, see k_bm_*)
    unsigned long long* h_pub; unsigned long long* d_h_pub; unsigned pub_seq;      // the same three scalars published by a kernel into mapped host memory (k_lm_publish)
    // accumulation = 1 (glio_localmap_set_accumulation): centroids as pcl::VoxelGrid forms them -- FLOAT sums over a voxel's points in the order of the
    // concatenated cloud (keyframes oldest first, points in scan order: the order a stable sort by voxel index leaves, and what the oracle's restatement
    // does) -- instead of the exact fixed-point sums.  Same voxels, same output order; the centroids then equal the oracle's bit for bit.
    int accumulation;
    int* d_fill; int* d_slot_start; int* d_plist;      // [table_cap], [table_cap], [width * cap]: per-voxel fill counters, list starts, point lists
};

#define LM_EMPTY (~0ull)
#define LM_FIX 1048576.0            /* 2^20: fixed-point scale of the voxel sums */

__device__ __forceinline__ int f2ord(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }
static inline float h_ord2f(int i) { const int u = i >= 0 ? i : i ^ 0x7fffffff; float f; memcpy(&f, &u, 4); return f; }     // the same on the host

__global__ void k_lm_transform(const float4* __restrict__ in, int n, const double q0, const double q1, const double q2, const double q3,
                               const double t0, const double t1, const double t2, float4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // transformCloud, Estimator.cpp:1517-1546 (double q*v + t, float store)
    if (i >= n) return;
    const float4 p = in[i];
    const double v[3] = {(double)p.x, (double)p.y, (double)p.z};
    // Eigen: v + w * (2 u x v) + u x (2 u x v), products kept separate (no contraction) as in assoc_kernels.hip
    double uv[3] = {q2 * v[2] - q3 * v[1], q3 * v[0] - q1 * v[2], q1 * v[1] - q2 * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    const double uuv[3] = {q2 * uv[2] - q3 * uv[1], q3 * uv[0] - q1 * uv[2], q1 * uv[1] - q2 * uv[0]};
    out[i] = make_float4((float)((v[0] + q0 * uv[0] + uuv[0]) + t0), (float)((v[1] + q0 * uv[1] + uuv[1]) + t1),
                         (float)((v[2] + q0 * uv[2] + uuv[2]) + t2), p.w);
}

// the same from a cloud that is already on the device in the LiDAR frame: p_body = p - off in FLOAT (what a caller's float cloud minus the
// extrinsic translation gives), then transformCloud as above
__global__ void k_lm_transform_off(const float4* __restrict__ in, int n, const float ox, const float oy, const float oz, const double q0, const double q1, const double q2,
                                   const double q3, const double t0, const double t1, const double t2, float4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = in[i];
    const float bx = p.x - ox, by = p.y - oy, bz = p.z - oz;
    const double v[3] = {(double)bx, (double)by, (double)bz};
    double uv[3] = {q2 * v[2] - q3 * v[1], q3 * v[0] - q1 * v[2], q1 * v[1] - q2 * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    const double uuv[3] = {q2 * uv[2] - q3 * uv[1], q3 * uv[0] - q1 * uv[2], q1 * uv[1] - q2 * uv[0]};
    out[i] = make_float4((float)((v[0] + q0 * uv[0] + uuv[0]) + t0), (float)((v[1] + q0 * uv[1] + uuv[1]) + t1),
                         (float)((v[2] + q0 * uv[2] + uuv[2]) + t2), p.w);
}

__global__ void k_lm_clear(unsigned long long* keys, long long* sum, int* cnt, int cap, int* nkeys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap) { keys[i] = LM_EMPTY; cnt[i] = 0; sum[4 * (size_t)i] = 0; sum[4 * (size_t)i + 1] = 0; sum[4 * (size_t)i + 2] = 0; sum[4 * (size_t)i + 3] = 0; }
    if (i == 0) *nkeys = 0;
}
// (also records the slot's point count on the device: the build used to upload all `width` counts from pageable memory -- an API call of ~8 us in a
//  chain of launches that is bound by the host's launch rate)
__global__ void k_lm_bbox_init(int* bbox, int* n_slot, const int n) {
    const int i = threadIdx.x;
    if (i < 3) bbox[i] = 0x7fffffff; else if (i < 6) bbox[i] = (int)0x80000000; else if (i == 6) *n_slot = n;
}
// bounding box of one ring slot (ordered-int atomics)
__global__ __launch_bounds__(1024) void k_lm_bbox(const float4* __restrict__ pts, int n, int* bbox) {
    __shared__ int s_mn[16][3], s_mx[16][3];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, mx[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
    for (; i < n; i += gridDim.x * blockDim.x) {
        const float4 p = pts[i];
        const int o[3] = {f2ord(p.x), f2ord(p.y), f2ord(p.z)};
#pragma unroll
        for (int c = 0; c < 3; ++c) { mn[c] = min(mn[c], o[c]); mx[c] = max(mx[c], o[c]); }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { mn[c] = min(mn[c], __shfl_xor(mn[c], off, 64)); mx[c] = max(mx[c], __shfl_xor(mx[c], off, 64)); }
    }
    // one set of six atomics per 1024-thread workgroup (they all hit the same six words)
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { s_mn[threadIdx.x >> 6][c] = mn[c]; s_mx[threadIdx.x >> 6][c] = mx[c]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int c = threadIdx.x % 3;
        const int nw = (blockDim.x + 63) >> 6;
        if (threadIdx.x < 3) { int v = 0x7fffffff; for (int w = 0; w < nw; ++w) v = min(v, s_mn[w][c]); atomicMin(bbox + c, v); }
        else { int v = (int)0x80000000; for (int w = 0; w < nw; ++w) v = max(v, s_mx[w][c]); atomicMax(bbox + 3 + c, v); }
    }
}
// union of the slot boxes (slots with n = 0 are skipped)
__global__ void k_lm_bbox_union(const int* __restrict__ slot_bbox, const int* __restrict__ ns, int width, int* bbox, int* bm_over, int* nvox) {
    const int c = threadIdx.x;
    if (c == 6) *bm_over = 0;
    if (c == 7) *nvox = 0;                 // (k_lm_list counts the voxels from zero: was a hipMemsetAsync of its own)
    if (c >= 6) return;
    int v = c < 3 ? 0x7fffffff : (int)0x80000000;
    for (int k = 0; k < width; ++k) if (ns[k] > 0) v = c < 3 ? min(v, slot_bbox[6 * k + c]) : max(v, slot_bbox[6 * k + c]);
    bbox[c] = v;
}
__device__ __forceinline__ unsigned lm_hash(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (unsigned)k;
}
__device__ __forceinline__ unsigned long long lm_key(int ix, int iy, int iz) {        // absolute voxel coordinates, 21 bits each, biased
    return ((unsigned long long)(unsigned)(ix + (1 << 20)) << 42) | ((unsigned long long)(unsigned)(iy + (1 << 20)) << 21) | (unsigned long long)(unsigned)(iz + (1 << 20));
}
// add (sign = +1) or remove (sign = -1) the points of one keyframe
__global__ void k_lm_accumulate(const float4* __restrict__ pts, int n, float inv_leaf, int sign, unsigned long long* keys, long long* sum, int* cnt,
                                int cap, int* nkeys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    const unsigned long long key = lm_key((int)floorf(p.x * inv_leaf), (int)floorf(p.y * inv_leaf), (int)floorf(p.z * inv_leaf));
    unsigned s = lm_hash(key) & (cap - 1);
    int probes = 0;
    for (;;) {
        const unsigned long long k = sign > 0 ? atomicCAS(keys + s, LM_EMPTY, key) : keys[s];
        if (k == LM_EMPTY) { if (sign > 0) atomicAdd(nkeys, 1); break; }       // (a removal always finds its key)
        if (k == key) break;
        s = (s + 1) & (cap - 1);
        if (++probes >= cap) { atomicOr(nkeys, 0x40000000); return; }          // table full: reported by glio_localmap_build
    }
    const long long f[4] = {llrint((double)p.x * LM_FIX), llrint((double)p.y * LM_FIX), llrint((double)p.z * LM_FIX), llrint((double)p.w * LM_FIX)};
#pragma unroll
    for (int c = 0; c < 4; ++c) atomicAdd(reinterpret_cast<unsigned long long*>(sum + 4 * (size_t)s + c), (unsigned long long)(sign > 0 ? f[c] : -f[c]));
    atomicAdd(cnt + s, sign);
}
// list the live voxels with their pcl::VoxelGrid linear index (relative to the ring's bounding box) for the ordered output
__global__ __launch_bounds__(1024) void k_lm_list(const unsigned long long* __restrict__ keys, const int* __restrict__ cnt, int cap, float inv_leaf,
                                                  const int* __restrict__ bbox, int* nvox, unsigned long long* vkey, int* vslot, int max_vox,
                                                  unsigned* __restrict__ bm, const unsigned long long bm_bits, int* bm_over) {
    // one list position per live voxel: positions come from a block-wide count (ballot per wavefront, 16 wavefront totals through LDS)
    // and ONE atomic per 1024 slots -- an atomic per voxel (~1e5 on one address) made this kernel 100-500 us
    __shared__ int s_w[16], s_base;
    const int s = blockIdx.x * 1024 + threadIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool live = s < cap && cnt[s] > 0;
    const unsigned long long bal = __ballot(live);
    if (lane == 0) s_w[wv] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int k = 0; k < 16; ++k) { const int v = s_w[k]; s_w[k] = t; t += v; }
        s_base = t > 0 ? atomicAdd(nvox, t) : 0;
    }
    __syncthreads();
    if (!live) return;
    int min_b[3], div_b[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { min_b[c] = (int)floorf(ord2f(bbox[c]) * inv_leaf); div_b[c] = (int)floorf(ord2f(bbox[3 + c]) * inv_leaf) - min_b[c] + 1; }
    const unsigned long long k = keys[s];
    const int ix = (int)((k >> 42) & 0x1fffff) - (1 << 20), iy = (int)((k >> 21) & 0x1fffff) - (1 << 20), iz = (int)(k & 0x1fffff) - (1 << 20);
    const unsigned long long lin = (unsigned long long)((long long)(ix - min_b[0]) + (long long)(iy - min_b[1]) * div_b[0] + (long long)(iz - min_b[2]) * div_b[0] * (long long)div_b[1]);
    const int v = s_base + s_w[wv] + __popcll(bal & ((1ull << lane) - 1ull));
    if (v < max_vox) { vkey[v] = lin; vslot[v] = s; }
    // the voxel's bit in the occupancy bitmap of the bounding box (k_bm_*: its rank among the set bits is its place in the ordered output)
    if (bm) { if (lin < bm_bits) atomicOr(&bm[lin >> 5], 1u << (unsigned)(lin & 31ull)); else *bm_over = 1; }
}
// ------------------------------------------------------------------------------------------------
// Ordered output WITHOUT a sort.  pcl::VoxelGrid emits the voxels by ascending linear index; the indices are distinct and bounded by the cell count
// of the bounding box, so a voxel's place in the output is the number of occupied cells below it: the rank of its bit among the set bits of an occupancy
// bitmap over the box.  k_lm_list sets the bits; k_bm_scan1 / k_bm_scan2 turn the per-word popcounts into running counts (per word inside a block of
// 1024 words, per block); k_bm_rank writes every voxel's table slot at its rank; the emit kernel clears the words it used.  Three launches behind
// the build's synchronisation instead of the nine of a three-pass radix sort (~58 -> ~15 us for 1e5 voxels).  A bounding box of more than 2^27
// cells (16 MB of bits) takes the radix sort below, as does GLIO_LM_SORT=1.
// ------------------------------------------------------------------------------------------------
#define BM_MAX_BITS (1ull << 27)
__global__ __launch_bounds__(1024) void k_bm_scan1(const unsigned* __restrict__ bm, const int nwords, int* __restrict__ pre, int* __restrict__ blk) {
    __shared__ int s_w[16];
    const int w = blockIdx.x * 1024 + threadIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = w < nwords ? __popc(bm[w]) : 0;
    int incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int v = threadIdx.x < 16 ? s_w[threadIdx.x] : 0;
        int inc = v;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) { const int o = __shfl_up(inc, off, 64); if ((int)threadIdx.x >= off) inc += o; }
        if (threadIdx.x < 16) s_w[threadIdx.x] = inc - v;
        if (threadIdx.x == 15) blk[blockIdx.x] = inc;
    }
    __syncthreads();
    if (w < nwords) pre[w] = s_w[wv] + incl - c;
}
// exclusive scan of the block totals in place (one workgroup; nblk <= 4096: four per thread)
__global__ __launch_bounds__(1024) void k_bm_scan2(int* __restrict__ blk, const int nblk) {
    __shared__ int s_w[16];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    int a[4], tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { a[k] = 4 * t + k < nblk ? blk[4 * t + k] : 0; tot += a[k]; }
    int incl = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    if (t < 64) {
        const int v = t < 16 ? s_w[t] : 0;
        int inc = v;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) { const int o = __shfl_up(inc, off, 64); if (t >= off) inc += o; }
        if (t < 16) s_w[t] = inc - v;
    }
    __syncthreads();
    int run = s_w[wv] + incl - tot;
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (4 * t + k < nblk) blk[4 * t + k] = run; run += a[k]; }
}
__global__ void k_bm_rank(const unsigned long long* __restrict__ vkey, const int* __restrict__ vslot, const int nv, const unsigned* __restrict__ bm,
                          const int* __restrict__ pre, const int* __restrict__ blk, int* __restrict__ vslot_sorted) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    const unsigned long long lin = vkey[v];
    const int w = (int)(lin >> 5);
    const unsigned below = bm[w] & ((1u << (unsigned)(lin & 31ull)) - 1u);
    vslot_sorted[blk[w >> 10] + pre[w] + __popc(below)] = vslot[v];
}
// (the words of the bitmap this build set, zero again for the next one: by the emit kernels, which run over the same nv voxels)
__device__ __forceinline__ void bm_clear_word(unsigned* __restrict__ bm, const unsigned long long* __restrict__ vkey, const int v, const unsigned long long bm_bits) {
    if (bm) { const unsigned long long lin = vkey[v]; if (lin < bm_bits) bm[lin >> 5] = 0u; }
}
// ------------------------------------------------------------------------------------------------
// Ordered output: the live voxels sorted by their pcl::VoxelGrid linear index.  A least-significant-digit radix sort on 8-bit
// digits, hand-written for this case: the keys are bounded by the ring's bounding box (typically < 2^24), so the host -- which
// reads the voxel count back anyway -- asks for only ceil(bits / 8) passes (3 instead of the 8 a generic 64-bit sort runs).
//   k_rs_hist     one wavefront per tile of 1024 pairs: 256-bin digit histogram (LDS atomics) -> hist[tile][digit]
//   k_rs_scan     exclusive scan over (digit, tile) in that order: where each tile's run of each digit starts
//   k_rs_scatter  the same wavefront per tile walks its 16 chunks of 64 in order; inside a chunk a pair's rank among the lanes
//                 with the same digit comes from eight ballots (one per digit bit): stable, no LDS traffic for the ranking
#define RS_TILE 1024
__global__ __launch_bounds__(64) void k_rs_hist(const unsigned long long* __restrict__ key, const int n, const int shift, const int nt, int* __restrict__ hist) {
    __shared__ int h[256];
    const int lane = threadIdx.x, t0 = blockIdx.x * RS_TILE;
    for (int d = lane; d < 256; d += 64) h[d] = 0;
    GLIO_WAVE_LDS_SYNC();
    unsigned long long kk[RS_TILE / 64];
#pragma unroll
    for (int q = 0; q < RS_TILE / 64; ++q) { const int e = t0 + 64 * q + lane; kk[q] = e < n ? key[e] : ~0ull; }
#pragma unroll
    for (int q = 0; q < RS_TILE / 64; ++q) if (t0 + 64 * q + lane < n) atomicAdd(&h[(int)((kk[q] >> shift) & 255ull)], 1);
    GLIO_WAVE_LDS_SYNC();
    for (int d = lane; d < 256; d += 64) hist[blockIdx.x * 256 + d] = h[d];           // [tile][digit]: coalesced here, in the scan and in the scatter
}
#define RS_SCAN_PER 64          /* tiles of one (digit, quarter) held in registers: up to 256 tiles = 262144 voxels */
__global__ __launch_bounds__(1024) void k_rs_scan(int* __restrict__ hist, const int nt) {
    // 256 digits x 4 quarters of the tiles: every thread sums its quarter (coalesced over the digits), the quarters and then the digits are chained
    // through LDS, and the thread rewrites its quarter as running offsets.  The quarter is read ONCE, all loads in flight together, and kept in
    // registers for the rewrite (it used to be two loops of dependent round trips: 10 us per pass for 103 tiles, three passes per map build).
    __shared__ int part[4][256], dbase[256];
    const int d = threadIdx.x & 255, q = threadIdx.x >> 8;
    const int per = (nt + 3) / 4, ta = min(nt, q * per), tb = min(nt, ta + per);
    const bool in_regs = per <= RS_SCAN_PER;
    int v[RS_SCAN_PER];
    int s = 0;
    if (in_regs) {
#pragma unroll
        for (int k = 0; k < RS_SCAN_PER; ++k) v[k] = ta + k < tb ? hist[(ta + k) * 256 + d] : 0;
#pragma unroll
        for (int k = 0; k < RS_SCAN_PER; ++k) s += v[k];
    } else {
        for (int t = ta; t < tb; ++t) s += hist[t * 256 + d];
    }
    part[q][d] = s;
    __syncthreads();
    if (threadIdx.x < 256) dbase[d] = part[0][d] + part[1][d] + part[2][d] + part[3][d];
    __syncthreads();
    if (threadIdx.x < 64) {            // exclusive scan of the 256 digit totals by one wavefront: four per lane, then across the lanes
        const int l = threadIdx.x;
        const int a0 = dbase[4 * l], a1 = dbase[4 * l + 1], a2 = dbase[4 * l + 2], a3 = dbase[4 * l + 3];
        const int tot = a0 + a1 + a2 + a3;
        int incl = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if (l >= off) incl += o; }
        const int ex = incl - tot;
        dbase[4 * l] = ex; dbase[4 * l + 1] = ex + a0; dbase[4 * l + 2] = ex + a0 + a1; dbase[4 * l + 3] = ex + a0 + a1 + a2;
    }
    __syncthreads();
    int run = dbase[d];
    for (int k = 0; k < q; ++k) run += part[k][d];
    if (in_regs) {
#pragma unroll
        for (int k = 0; k < RS_SCAN_PER; ++k) if (ta + k < tb) { hist[(ta + k) * 256 + d] = run; run += v[k]; }
    } else {
        for (int t = ta; t < tb; ++t) { const int x = hist[t * 256 + d]; hist[t * 256 + d] = run; run += x; }
    }
}
__global__ __launch_bounds__(64) void k_rs_scatter(const unsigned long long* __restrict__ key, const int* __restrict__ val, const int n, const int shift, const int nt,
                                                   const int* __restrict__ hist, unsigned long long* __restrict__ okey, int* __restrict__ oval) {
    __shared__ int base[256];
    const int lane = threadIdx.x, t0 = blockIdx.x * RS_TILE;
    for (int d = lane; d < 256; d += 64) base[d] = hist[blockIdx.x * 256 + d];
    GLIO_WAVE_LDS_SYNC();
    // all 16 chunks of the tile are fetched first (16 independent loads per lane in flight), then ranked chunk by chunk
    unsigned long long kk[RS_TILE / 64]; int vv[RS_TILE / 64];
#pragma unroll
    for (int q = 0; q < RS_TILE / 64; ++q) {
        const int e = t0 + 64 * q + lane;
        kk[q] = e < n ? key[e] : 0ull;
        vv[q] = e < n ? val[e] : 0;
    }
#pragma unroll
    for (int q = 0; q < RS_TILE / 64; ++q) {
        const int e = t0 + 64 * q + lane;
        const bool live = e < n;
        const unsigned long long k = kk[q];
        const int dg = (int)((k >> shift) & 255ull);
        unsigned long long same = __ballot(live);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bal = __ballot((dg >> b) & 1);
            same &= ((dg >> b) & 1) ? bal : ~bal;
        }
        const int rank = __popcll(same & ((1ull << lane) - 1ull));
        const int pos = live ? base[dg] + rank : 0;
        GLIO_WAVE_LDS_SYNC();
        if (live && rank == 0) base[dg] += __popcll(same);           // the first lane of every digit group advances its run
        GLIO_WAVE_LDS_SYNC();
        if (live) { okey[pos] = k; oval[pos] = vv[q]; }
    }
}

__global__ void k_lm_emit(const int* __restrict__ vslot_sorted, int nv, const long long* __restrict__ sum, const int* __restrict__ cnt,
                          float4* __restrict__ out, unsigned* __restrict__ bm, const unsigned long long* __restrict__ vkey, const unsigned long long bm_bits) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    bm_clear_word(bm, vkey, v, bm_bits);
    const int s = vslot_sorted[v];
    const double c = (double)cnt[s];
    out[v] = make_float4((float)((double)sum[4 * (size_t)s] / LM_FIX / c), (float)((double)sum[4 * (size_t)s + 1] / LM_FIX / c),
                         (float)((double)sum[4 * (size_t)s + 2] / LM_FIX / c), (float)((double)sum[4 * (size_t)s + 3] / LM_FIX / c));
}

// ---- float accumulation in concatenation order (accumulation = 1)
// exclusive scan of the live voxels' counts in OUTPUT order -> where each voxel's point list starts (one workgroup: a contiguous chunk per thread)
__global__ __launch_bounds__(1024) void k_lm_starts(const int* __restrict__ vslot_sorted, const int nv, const int* __restrict__ cnt, int* __restrict__ slot_start) {
    __shared__ int part[1024];
    const int tid = threadIdx.x, chunk = (nv + 1023) / 1024, v0 = tid * chunk, v1 = min(nv, v0 + chunk);
    int sum = 0;
    for (int v = v0; v < v1; ++v) sum += cnt[vslot_sorted[v]];
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) { int t = 0; for (int k = 0; k < 1024; ++k) { const int x = part[k]; part[k] = t; t += x; } }
    __syncthreads();
    int run = part[tid];
    for (int v = v0; v < v1; ++v) { const int sl = vslot_sorted[v]; slot_start[sl] = run; run += cnt[sl]; }
}
// every ring point -> its voxel's list (arbitrary position; the entry is the point's index in the concatenated cloud: age * cap + j)
__global__ void k_lm_scatter_idx(const float4* __restrict__ ring, const int* __restrict__ ns, const int cap, const int width, const int head, const int count,
                                 const float inv_leaf, const unsigned long long* __restrict__ keys, const int table_cap, const int* __restrict__ slot_start,
                                 int* __restrict__ fill, int* __restrict__ plist) {
    const int slot = blockIdx.y, age = (slot - head + width) % width;
    if (age >= count) return;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ns[slot]) return;
    const float4 p = ring[(size_t)slot * cap + j];
    const unsigned long long key = lm_key((int)floorf(p.x * inv_leaf), (int)floorf(p.y * inv_leaf), (int)floorf(p.z * inv_leaf));
    unsigned s = lm_hash(key) & (table_cap - 1);
    while (keys[s] != key) s = (s + 1) & (table_cap - 1);              // (every ring point's voxel is in the table)
    const int pos = atomicAdd(fill + s, 1);
    plist[slot_start[s] + pos] = age * cap + j;
}
// one thread per voxel: its list sorted by concatenated index (insertion sort in the thread's own segment: a voxel holds tens of points), then the
// float sums in that order and the centroid = sum / (float) count, as pcl::VoxelGrid
__global__ void k_lm_emit_float(const int* __restrict__ vslot_sorted, const int nv, const int* __restrict__ cnt, const int* __restrict__ slot_start,
                                int* __restrict__ plist, const float4* __restrict__ ring, const int cap, const int width, const int head, float4* __restrict__ out,
                                unsigned* __restrict__ bm, const unsigned long long* __restrict__ vkey, const unsigned long long bm_bits) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    bm_clear_word(bm, vkey, v, bm_bits);
    const int s = vslot_sorted[v], n = cnt[s];
    int* L = plist + slot_start[s];
    for (int a = 1; a < n; ++a) {
        const int x = L[a];
        int b = a - 1;
        while (b >= 0 && L[b] > x) { L[b + 1] = L[b]; --b; }
        L[b + 1] = x;
    }
    float ax = 0.f, ay = 0.f, az = 0.f, aw = 0.f;
    for (int a = 0; a < n; ++a) {
        const int idx = L[a], age = idx / cap, j = idx - age * cap;
        const float4 p = ring[(size_t)((head + age) % width) * cap + j];
        ax += p.x; ay += p.y; az += p.z; aw += p.w;
    }
    const float c = (float)n;
    out[v] = make_float4(ax / c, ay / c, az / c, aw / c);
}

#define LM_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { glio_set_error("%s failed: %s", #expr, hipGetErrorString(e_)); return GLIO_E_HIP; } } while (0)
static int lm_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

void glio_localmap_destroy(glio_ctx* c) {
    LocalMap* m = c->localmap;
    if (!m) return;
    void* p[] = {m->d_slot_bbox, m->d_nkeys, m->d_n, m->d_ring, m->d_keys, m->d_sum, m->d_cnt, m->d_bbox, m->d_nvox, m->d_vkey, m->d_vkey_sorted, m->d_vslot, m->d_vslot_sorted, m->d_sort_tmp, m->d_out, m->d_fill, m->d_slot_start, m->d_plist, m->d_bm, m->d_bm_pre, m->d_bm_blk, m->d_bm_over};
    for (void* q : p) if (q) hipFree(q);
    if (m->h_pin) hipHostFree(m->h_pin);
    if (m->h_pub) hipHostFree(m->h_pub);
    delete[] m->h_n;
    delete m;
    c->localmap = nullptr;
}

extern "C" {

int glio_localmap_config(glio_ctx* c, int width, float leaf, int max_points_per_keyframe) {
    if (!c || width < 1 || !(leaf > 0.f) || max_points_per_keyframe < 1) return GLIO_E_ARG;
    LM_CHECK(hipSetDevice(c->device));
    const int keep_accumulation = c->localmap ? c->localmap->accumulation : 0;      // (the centroid arithmetic is a property of the context, not of one ring)
    glio_localmap_destroy(c);
    LocalMap* m = new LocalMap();
    memset(m, 0, sizeof *m);
    m->width = width; m->cap = max_points_per_keyframe; m->leaf = leaf;
    m->max_vox = c->opts.max_map_points > 0 ? c->opts.max_map_points : 1;
    m->table_cap = lm_pow2(2 * m->max_vox);
    m->h_n = new int[width]();
    LM_CHECK(hipMalloc((void**)&m->d_n, (size_t)width * 4));
    LM_CHECK(hipMalloc((void**)&m->d_ring, (size_t)width * m->cap * 16));
    LM_CHECK(hipMalloc((void**)&m->d_keys, (size_t)m->table_cap * 8)); LM_CHECK(hipMalloc((void**)&m->d_sum, (size_t)m->table_cap * 32));
    LM_CHECK(hipMalloc((void**)&m->d_cnt, (size_t)m->table_cap * 4)); LM_CHECK(hipMalloc((void**)&m->d_bbox, 32)); LM_CHECK(hipMalloc((void**)&m->d_nvox, 4));
    LM_CHECK(hipMalloc((void**)&m->d_slot_bbox, (size_t)width * 24)); LM_CHECK(hipMalloc((void**)&m->d_nkeys, 4));
    LM_CHECK(hipMalloc((void**)&m->d_vkey, (size_t)m->max_vox * 8)); LM_CHECK(hipMalloc((void**)&m->d_vkey_sorted, (size_t)m->max_vox * 8));
    LM_CHECK(hipMalloc((void**)&m->d_vslot, (size_t)m->max_vox * 4)); LM_CHECK(hipMalloc((void**)&m->d_vslot_sorted, (size_t)m->max_vox * 4));
    LM_CHECK(hipMalloc((void**)&m->d_out, (size_t)m->max_vox * 16));
    m->sort_tmp_bytes = (size_t)256 * ((m->max_vox + RS_TILE - 1) / RS_TILE) * 4;          // digit histogram [256][tiles] of the radix sort
    LM_CHECK(hipMalloc(&m->d_sort_tmp, m->sort_tmp_bytes + 16));
    LM_CHECK(hipHostMalloc((void**)&m->h_pin, 64));
    LM_CHECK(hipHostMalloc((void**)&m->h_pub, 64));
    memset(m->h_pub, 0, 64);
    LM_CHECK(hipHostGetDevicePointer((void**)&m->d_h_pub, m->h_pub, 0));
    {   // occupancy bitmap of the bounding box + its running counts (GLIO_LM_SORT=1: the radix sort only)
        const char* e = getenv("GLIO_LM_SORT");
        m->force_sort = e && atoi(e) != 0;
        LM_CHECK(hipMalloc((void**)&m->d_bm_over, 4));
        LM_CHECK(hipMemsetAsync(m->d_bm_over, 0, 4, c->stream));
        if (!m->force_sort) {
            const size_t words = (size_t)(BM_MAX_BITS / 32);
            LM_CHECK(hipMalloc((void**)&m->d_bm, words * 4)); LM_CHECK(hipMalloc((void**)&m->d_bm_pre, words * 4)); LM_CHECK(hipMalloc((void**)&m->d_bm_blk, 4096 * 4));
            LM_CHECK(hipMemsetAsync(m->d_bm, 0, words * 4, c->stream));
        }
    }
    hipLaunchKernelGGL(k_lm_clear, dim3((m->table_cap + 255) / 256), dim3(256), 0, c->stream, m->d_keys, m->d_sum, m->d_cnt, m->table_cap, m->d_nkeys);
    LM_CHECK(hipMemsetAsync(m->d_n, 0, (size_t)width * 4, c->stream));
    LM_CHECK(hipStreamSynchronize(c->stream));
    c->localmap = m;
    if (keep_accumulation) return glio_localmap_set_accumulation(c, keep_accumulation);
    return GLIO_OK;
}

int glio_localmap_push(glio_ctx* c, const float* cloud_xyzi, int n, const double q[4], const double t[3]) { return glio_localmap_push_strided(c, cloud_xyzi, n, 16, 12, q, t); }
int glio_localmap_push_strided(glio_ctx* c, const void* cloud_xyzi, int n, int stride_bytes, int intensity_offset, const double q[4], const double t[3]) {
    if (!c || !c->localmap) { glio_set_error("glio_localmap_config first"); return GLIO_E_STATE; }
    LocalMap* m = c->localmap;
    if (n < 0 || n > m->cap || (n > 0 && !cloud_xyzi) || !q || !t) return GLIO_E_ARG;
    if (!glio_point_layout_ok(stride_bytes, intensity_offset)) { glio_set_error("bad point layout (stride %d, intensity at %d)", stride_bytes, intensity_offset); return GLIO_E_ARG; }
    LM_CHECK(hipSetDevice(c->device));
    const float inv_leaf = 1.0f / m->leaf;
    int slot;
    if (m->count < m->width) { slot = (m->head + m->count) % m->width; ++m->count; }
    else {                                                                        // recent_surf_keyframes.pop_front() (:3585):
        slot = m->head; m->head = (m->head + 1) % m->width;                       // take the oldest keyframe's points out of the table
        const int no = m->h_n[slot];
        if (no > 0) hipLaunchKernelGGL(k_lm_accumulate, dim3((no + 255) / 256), dim3(256), 0, c->stream, m->d_ring + (size_t)slot * m->cap, no, inv_leaf, -1,
                                       m->d_keys, m->d_sum, m->d_cnt, m->table_cap, m->d_nkeys);
    }
    float4* dst = m->d_ring + (size_t)slot * m->cap;
    hipLaunchKernelGGL(k_lm_bbox_init, dim3(1), dim3(64), 0, c->stream, m->d_slot_bbox + 6 * slot, m->d_n + slot, n);
    if (n > 0) {
        // stage the raw scan in the destination itself, transform in place, then add it to the voxel table
        { const int ru = glio_upload_points(c->stream, &c->raw_stage, cloud_xyzi, n, stride_bytes, intensity_offset, dst); if (ru != GLIO_OK) return ru; }
        hipLaunchKernelGGL(k_lm_transform, dim3((n + 255) / 256), dim3(256), 0, c->stream, dst, n, q[0], q[1], q[2], q[3], t[0], t[1], t[2], dst);
        hipLaunchKernelGGL(k_lm_bbox, dim3(std::min(64, (n + 1023) / 1024)), dim3(1024), 0, c->stream, dst, n, m->d_slot_bbox + 6 * slot);
        hipLaunchKernelGGL(k_lm_accumulate, dim3((n + 255) / 256), dim3(256), 0, c->stream, dst, n, inv_leaf, +1, m->d_keys, m->d_sum, m->d_cnt, m->table_cap, m->d_nkeys);
    }
    LM_CHECK(hipGetLastError());
    LM_CHECK(hipStreamSynchronize(c->stream));            // the caller's buffer may be reused after return
    m->h_n[slot] = n;
    ++m->pushed;
    return GLIO_OK;
}

// The newest keyframe's cloud is usually on the device already -- glio_set_scan put it into window slot `slot` for the association.  This pushes THAT
// copy (LiDAR frame) into the ring: body-frame point = scan point - lidar_offset (float, the extrinsic translation with R_lb = I as a caller's float
// cloud would hold it), pose (q, t) of the body in the world.  No second upload of the same megabyte, no synchronisation.
int glio_localmap_push_scan(glio_ctx* c, int scan_slot, const float lidar_offset[3], const double q[4], const double t[3]) {
    if (!c || !c->localmap) { glio_set_error("glio_localmap_config first"); return GLIO_E_STATE; }
    LocalMap* m = c->localmap;
    if (scan_slot < 0 || scan_slot >= c->W || !lidar_offset || !q || !t) return GLIO_E_ARG;
    const int n = c->h_scan_count[scan_slot];
    if (n < 0 || n > m->cap) { glio_set_error("scan of slot %d has %d points, the ring takes %d", scan_slot, n, m->cap); return GLIO_E_ARG; }
    LM_CHECK(hipSetDevice(c->device));
    const float inv_leaf = 1.0f / m->leaf;
    int slot;
    if (m->count < m->width) { slot = (m->head + m->count) % m->width; ++m->count; }
    else {
        slot = m->head; m->head = (m->head + 1) % m->width;
        const int no = m->h_n[slot];
        if (no > 0) hipLaunchKernelGGL(k_lm_accumulate, dim3((no + 255) / 256), dim3(256), 0, c->stream, m->d_ring + (size_t)slot * m->cap, no, inv_leaf, -1,
                                       m->d_keys, m->d_sum, m->d_cnt, m->table_cap, m->d_nkeys);
    }
    float4* dst = m->d_ring + (size_t)slot * m->cap;
    hipLaunchKernelGGL(k_lm_bbox_init, dim3(1), dim3(64), 0, c->stream, m->d_slot_bbox + 6 * slot, m->d_n + slot, n);
    if (n > 0) {
        hipLaunchKernelGGL(k_lm_transform_off, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->d_scan + (size_t)glio_scan_row(c, scan_slot) * c->cap, n, lidar_offset[0], lidar_offset[1],
                           lidar_offset[2], q[0], q[1], q[2], q[3], t[0], t[1], t[2], dst);
        hipLaunchKernelGGL(k_lm_bbox, dim3(std::min(64, (n + 1023) / 1024)), dim3(1024), 0, c->stream, dst, n, m->d_slot_bbox + 6 * slot);
        hipLaunchKernelGGL(k_lm_accumulate, dim3((n + 255) / 256), dim3(256), 0, c->stream, dst, n, inv_leaf, +1, m->d_keys, m->d_sum, m->d_cnt, m->table_cap, m->d_nkeys);
    }
    LM_CHECK(hipGetLastError());
    m->h_n[slot] = n;
    ++m->pushed;
    return GLIO_OK;
}

// The build needs three scalars on the host in its middle (voxel count, table fill, bounding box: the sort's geometry).  Three device-to-host copies and
// a stream synchronisation cost ~40 us of idle GPU there; one thread writing them into MAPPED host memory -- four 8-byte words, then a tag that carries the
// build's sequence number and a checksum of the words (writes to host memory have been seen out of order under load: glio_device.h, glio_result_mix) --
// and the host polling the tag cost ~10.  No match within 2 ms: the copies and the synchronisation, as before.
__global__ void k_lm_publish(const int* __restrict__ nvox, const int* __restrict__ nkeys, const int* __restrict__ bbox, const int* __restrict__ bm_over,
                             unsigned long long* out, const unsigned seq) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned long long w[4];
    const unsigned nk = (unsigned)nkeys[0] | (bm_over[0] ? 0x20000000u : 0u);          // (bit 30: voxel table overflow; bit 29: a voxel outside the bitmap)
    w[0] = (unsigned long long)(unsigned)nvox[0] | ((unsigned long long)nk << 32);
#pragma unroll
    for (int k = 0; k < 3; ++k) w[1 + k] = (unsigned long long)(unsigned)bbox[2 * k] | ((unsigned long long)(unsigned)bbox[2 * k + 1] << 32);
    unsigned long long cs = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) cs += glio_result_mix(w[k], (unsigned long long)k);
#pragma unroll
    for (int k = 0; k < 4; ++k) __hip_atomic_store(out + k, w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __hip_atomic_store(out + 4, (cs & 0xffffffff00000000ull) | seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
static bool lm_wait_published(LocalMap* m, int* out8) {
    const unsigned seq = m->pub_seq;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 1;; ++spins) {
        const unsigned long long tag = __atomic_load_n(m->h_pub + 4, __ATOMIC_ACQUIRE);
        if ((unsigned)(tag & 0xffffffffull) == seq) {
            unsigned long long w[4], cs = 0;
            for (int k = 0; k < 4; ++k) { w[k] = __atomic_load_n(m->h_pub + k, __ATOMIC_RELAXED); cs += glio_result_mix(w[k], (unsigned long long)k); }
            if ((cs & 0xffffffff00000000ull) == (tag & 0xffffffff00000000ull)) {
                out8[0] = (int)(unsigned)(w[0] & 0xffffffffull); out8[1] = (int)(unsigned)(w[0] >> 32);
                for (int k = 0; k < 3; ++k) { out8[2 + 2 * k] = (int)(unsigned)(w[1 + k] & 0xffffffffull); out8[3 + 2 * k] = (int)(unsigned)(w[1 + k] >> 32); }
                return true;
            }
        }
        if ((spins & 0xff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) return false;
    }
}

int glio_localmap_build(glio_ctx* c, int* out_points) {
    GLIO_TRACE("K1 glio_localmap_build (voxel grid + hash)");
    if (!c || !c->localmap) { glio_set_error("glio_localmap_config first"); return GLIO_E_STATE; }
    LocalMap* m = c->localmap;
    LM_CHECK(hipSetDevice(c->device));
    const float inv_leaf = 1.0f / m->leaf;
    // (the slots' point counts are on the device already: every push records its own, k_lm_bbox_init)
    if (m->nkeys_seen > m->table_cap / 2) {               // too many tombstones: rebuild the table from the ring once
        hipLaunchKernelGGL(k_lm_clear, dim3((m->table_cap + 255) / 256), dim3(256), 0, c->stream, m->d_keys, m->d_sum, m->d_cnt, m->table_cap, m->d_nkeys);
        for (int k = 0; k < m->count; ++k) {
            const int slot = (m->head + k) % m->width, n = m->h_n[slot];
            if (n > 0) hipLaunchKernelGGL(k_lm_accumulate, dim3((n + 255) / 256), dim3(256), 0, c->stream, m->d_ring + (size_t)slot * m->cap, n, inv_leaf, +1,
                                          m->d_keys, m->d_sum, m->d_cnt, m->table_cap, m->d_nkeys);
        }
    }
    if (m->d_bm && m->bm_dirty) {          // a build that ended between k_lm_list and its emit kernel (an error return) left bits behind
        LM_CHECK(hipMemsetAsync(m->d_bm, 0, (size_t)(BM_MAX_BITS / 32) * 4, c->stream));
        m->bm_dirty = 0;
    }
    hipLaunchKernelGGL(k_lm_bbox_union, dim3(1), dim3(64), 0, c->stream, m->d_slot_bbox, m->d_n, m->width, m->d_bbox, m->d_bm_over, m->d_nvox);
    hipLaunchKernelGGL(k_lm_list, dim3((m->table_cap + 1023) / 1024), dim3(1024), 0, c->stream, m->d_keys, m->d_cnt, m->table_cap, inv_leaf, m->d_bbox,
                       m->d_nvox, m->d_vkey, m->d_vslot, m->max_vox, m->d_bm, (unsigned long long)BM_MAX_BITS, m->d_bm_over);
    LM_CHECK(hipGetLastError());
    if (m->d_bm) m->bm_dirty = 1;
    m->pub_seq = m->pub_seq == 0xffffffffu ? 1u : m->pub_seq + 1u;
    hipLaunchKernelGGL(k_lm_publish, dim3(1), dim3(64), 0, c->stream, m->d_nvox, m->d_nkeys, m->d_bbox, m->d_bm_over, m->d_h_pub, m->pub_seq);
    LM_CHECK(hipGetLastError());
    if (!lm_wait_published(m, m->h_pin)) {
        LM_CHECK(hipMemcpyAsync(m->h_pin, m->d_nvox, 4, hipMemcpyDeviceToHost, c->stream));
        LM_CHECK(hipMemcpyAsync(m->h_pin + 1, m->d_nkeys, 4, hipMemcpyDeviceToHost, c->stream));
        LM_CHECK(hipMemcpyAsync(m->h_pin + 2, m->d_bbox, 24, hipMemcpyDeviceToHost, c->stream));
        LM_CHECK(hipMemcpyAsync(m->h_pin + 8, m->d_bm_over, 4, hipMemcpyDeviceToHost, c->stream));
        LM_CHECK(hipStreamSynchronize(c->stream));
        if (m->h_pin[8]) m->h_pin[1] |= 0x20000000;
    }
    const bool bm_over = (m->h_pin[1] & 0x20000000) != 0;
    m->h_pin[1] &= ~0x20000000;
    const int nv = m->h_pin[0];
    if (m->h_pin[1] & 0x40000000) { glio_set_error("local map voxel table overflow (raise max_map_points)"); return GLIO_E_ARG; }
    m->nkeys_seen = m->h_pin[1];
    if (nv > m->max_vox) { glio_set_error("local map has %d voxels, max_map_points is %d", nv, m->max_vox); return GLIO_E_ARG; }
    if (nv > 0) {
        // number of key bits from the bounding box, exactly as k_lm_list forms the linear index
        double span = 1.0;
        for (int c3 = 0; c3 < 3; ++c3) {
            const int lo = (int)floorf(h_ord2f(m->h_pin[2 + c3]) * inv_leaf), hi = (int)floorf(h_ord2f(m->h_pin[5 + c3]) * inv_leaf);
            span *= (double)(hi - lo + 1);
        }
        int* va = m->d_vslot;
        const unsigned long long* keys_final = m->d_vkey;          // every voxel's linear index, in whatever order: what the emit kernels clear the bitmap by
        const bool by_rank = m->d_bm && !bm_over && span <= (double)BM_MAX_BITS;
        if (by_rank) {
            const int nwords = (int)(((unsigned long long)span + 31ull) / 32ull), nblk = (nwords + 1023) / 1024;
            hipLaunchKernelGGL(k_bm_scan1, dim3(nblk), dim3(1024), 0, c->stream, m->d_bm, nwords, m->d_bm_pre, m->d_bm_blk);
            hipLaunchKernelGGL(k_bm_scan2, dim3(1), dim3(1024), 0, c->stream, m->d_bm_blk, nblk);
            hipLaunchKernelGGL(k_bm_rank, dim3((nv + 255) / 256), dim3(256), 0, c->stream, m->d_vkey, m->d_vslot, nv, m->d_bm, m->d_bm_pre, m->d_bm_blk, m->d_vslot_sorted);
            va = m->d_vslot_sorted;
        } else {
            int bits = 1;
            while (bits < 63 && (double)(1ull << bits) < span) ++bits;
            const int passes = (bits + 7) / 8, nt = (nv + RS_TILE - 1) / RS_TILE;
            unsigned long long* ka = m->d_vkey; unsigned long long* kb = m->d_vkey_sorted;
            int* vb = m->d_vslot_sorted;
            int* hist = reinterpret_cast<int*>(m->d_sort_tmp);
            for (int p = 0; p < passes; ++p) {
                hipLaunchKernelGGL(k_rs_hist, dim3(nt), dim3(64), 0, c->stream, ka, nv, 8 * p, nt, hist);
                hipLaunchKernelGGL(k_rs_scan, dim3(1), dim3(1024), 0, c->stream, hist, nt);
                hipLaunchKernelGGL(k_rs_scatter, dim3(nt), dim3(64), 0, c->stream, ka, va, nv, 8 * p, nt, hist, kb, vb);
                std::swap(ka, kb); std::swap(va, vb);
            }
            keys_final = ka;
        }
        LM_CHECK(hipGetLastError());
        if (m->accumulation == 1) {
            hipLaunchKernelGGL(k_lm_starts, dim3(1), dim3(1024), 0, c->stream, va, nv, m->d_cnt, m->d_slot_start);
            LM_CHECK(hipMemsetAsync(m->d_fill, 0, (size_t)m->table_cap * 4, c->stream));
            hipLaunchKernelGGL(k_lm_scatter_idx, dim3((m->cap + 255) / 256, m->width), dim3(256), 0, c->stream, m->d_ring, m->d_n, m->cap, m->width, m->head, m->count,
                               inv_leaf, m->d_keys, m->table_cap, m->d_slot_start, m->d_fill, m->d_plist);
            hipLaunchKernelGGL(k_lm_emit_float, dim3((nv + 255) / 256), dim3(256), 0, c->stream, va, nv, m->d_cnt, m->d_slot_start, m->d_plist, m->d_ring, m->cap, m->width,
                               m->head, m->d_out, m->d_bm, keys_final, (unsigned long long)BM_MAX_BITS);
        } else
        hipLaunchKernelGGL(k_lm_emit, dim3((nv + 255) / 256), dim3(256), 0, c->stream, va, nv, m->d_sum, m->d_cnt, m->d_out, m->d_bm, keys_final,
                           (unsigned long long)BM_MAX_BITS);     // (va: the sorted side after the last swap)
        m->bm_dirty = 0;
    } else m->bm_dirty = 0;
    const int rc = glio_assoc_build_map_dev(c, m->d_out, nv);          // K1: replaces setInputCloud(surf_local_map_ds) (:2056)
    if (rc) return rc;
    // (no wait here: the map's size is known since the synchronisation above, and everything that reads the map -- the searches, glio_localmap_read --
    //  is ordered behind the build on this stream or waits for it)
    if (out_points) *out_points = nv;
    return GLIO_OK;
}

// 0 (default): exact fixed-point voxel sums (insert / evict without re-streaming the ring, bit-reproducible whatever the atomic order);
// 1: pcl::VoxelGrid's own arithmetic -- float sums over the voxel's points in the order of the concatenated cloud (the oracle's restatement): the centroids
// then equal the oracle's BIT FOR BIT, at the price of one pass over the ring per build.  For A/B runs of the 1.5e-5 m difference between the two.
int glio_localmap_set_accumulation(glio_ctx* c, int mode) {
    if (!c || !c->localmap || (mode != 0 && mode != 1)) { glio_set_error("glio_localmap_config first; mode 0 or 1"); return GLIO_E_ARG; }
    LocalMap* m = c->localmap;
    LM_CHECK(hipSetDevice(c->device));
    if (mode == 1 && !m->d_plist) {
        LM_CHECK(hipMalloc((void**)&m->d_fill, (size_t)m->table_cap * 4)); LM_CHECK(hipMalloc((void**)&m->d_slot_start, (size_t)m->table_cap * 4));
        LM_CHECK(hipMalloc((void**)&m->d_plist, (size_t)m->width * m->cap * 4));
    }
    m->accumulation = mode;
    return GLIO_OK;
}

int glio_localmap_read(glio_ctx* c, float* out_xyzi, int capacity, int* out_n) {
    if (!c || !c->localmap || !out_n) return GLIO_E_ARG;
    LM_CHECK(hipSetDevice(c->device));
    const int n = c->map_n;
    *out_n = n;
    if (out_xyzi && n > 0) {
        if (capacity < n) return GLIO_E_ARG;
        LM_CHECK(hipStreamSynchronize(c->stream));
        LM_CHECK(hipMemcpy(out_xyzi, c->localmap->d_out, (size_t)n * 16, hipMemcpyDeviceToHost));
    }
    return GLIO_OK;
}

}  // extern "C"
