/*
 * orc_assoc.c -- CPU ORACLE (test infrastructure): Estimator::findCorrespondingSurfFeatures
 * (GLIO/src/Estimator.cpp:3633-3708) with the PCL kd-tree replaced by a brute-force exact 5-NN.
 *
 * Precision per step follows the reference (quirk Q2): the transform is computed in double and
 * stored as float (transformPoint, Estimator.cpp:1490-1498); nearest-neighbour squared distances are
 * float sums dx*dx+dy*dy+dz*dz in that order (FLANN L2_Simple<float> as used by pcl::KdTreeFLANN);
 * the 5x3 plane fit is double (colPivHouseholderQr, :3661); pd and weight are float (:3678-3679).
 * Ties in distance are broken by the lower map index (FLANN's own tie order is unspecified).
 * Compile with -ffp-contract=off so that no float multiply-add is fused.
 * PARITY UNPINNED -- see glio_oracle.h.
 */
#include <float.h>
#include <stdlib.h>
#include "glio_oracle.h"
#include "orc_math.h"

/* Eigen::ColPivHouseholderQR<Matrix<double,5,3>>::solve restated (Eigen 3.3 ColPivHouseholderQR.h:
 * column pivoting on the largest remaining column norm, Householder reflectors, solve through
 * nonzeroPivots()).  Column norms are recomputed directly at every step instead of being
 * down-dated, which can only change the pivot order on exact near-ties. */
void orc_plane_qr_solve(const double Ain[15], const double bin[5], double x[3]) {
    enum { M = 5, N = 3 };
    double A[M][N], b[M];
    int perm[N] = {0, 1, 2};
    for (int i = 0; i < M; ++i) { b[i] = bin[i]; for (int j = 0; j < N; ++j) A[i][j] = Ain[i * N + j]; }
    double maxnorm = 0;
    for (int j = 0; j < N; ++j) { double s = 0; for (int i = 0; i < M; ++i) s += A[i][j] * A[i][j]; s = sqrt(s); if (s > maxnorm) maxnorm = s; }
    const double thr_helper = (maxnorm * DBL_EPSILON) * (maxnorm * DBL_EPSILON) / (double)M;
    int nonzero = N;
    for (int k = 0; k < N; ++k) {
        int best = k; double bestsq = -1;
        for (int j = k; j < N; ++j) { double s = 0; for (int i = k; i < M; ++i) s += A[i][j] * A[i][j]; if (s > bestsq) { bestsq = s; best = j; } }
        if (nonzero == N && bestsq < thr_helper * (double)(M - k)) nonzero = k;
        if (best != k) {
            for (int i = 0; i < M; ++i) { double t = A[i][k]; A[i][k] = A[i][best]; A[i][best] = t; }
            int t = perm[k]; perm[k] = perm[best]; perm[best] = t;
        }
        /* makeHouseholderInPlace on A[k:,k] */
        double tail = 0;
        for (int i = k + 1; i < M; ++i) tail += A[i][k] * A[i][k];
        double c0 = A[k][k], beta, tau, v[M];
        if (tail <= DBL_MIN) { tau = 0; beta = c0; for (int i = k + 1; i < M; ++i) v[i] = 0; }
        else {
            beta = sqrt(c0 * c0 + tail);
            if (c0 >= 0) beta = -beta;
            for (int i = k + 1; i < M; ++i) v[i] = A[i][k] / (c0 - beta);
            tau = (beta - c0) / beta;
        }
        v[k] = 1.0;
        /* apply H = I - tau v v^T to remaining columns and to b */
        for (int j = k + 1; j < N; ++j) {
            double s = 0;
            for (int i = k; i < M; ++i) s += v[i] * A[i][j];
            s *= tau;
            for (int i = k; i < M; ++i) A[i][j] -= s * v[i];
        }
        {
            double s = 0;
            for (int i = k; i < M; ++i) s += v[i] * b[i];
            s *= tau;
            for (int i = k; i < M; ++i) b[i] -= s * v[i];
        }
        A[k][k] = beta;
        for (int i = k + 1; i < M; ++i) A[i][k] = 0;
    }
    double y[N] = {0, 0, 0};
    for (int i = nonzero - 1; i >= 0; --i) {
        double s = b[i];
        for (int j = i + 1; j < nonzero; ++j) s -= A[i][j] * y[j];
        y[i] = s / A[i][i];
    }
    for (int j = 0; j < N; ++j) x[perm[j]] = y[j];
}

/* exact brute-force 5-NN of one query over the whole map, ascending (distance, index).  The distances of a block of map
 * points are formed first (a loop the compiler vectorises: every d is its own chain dx*dx, + dy*dy, + dz*dz, so the
 * rounding does not depend on the vector width), then the block is scanned in index order. */
static void knn5_brute(const float* map, int M, float px, float py, float pz, float bd[5], int bi[5]) {
    enum { BLK = 64 };
    for (int k = 0; k < 5; ++k) { bd[k] = FLT_MAX; bi[k] = -1; }
    for (int m0 = 0; m0 < M; m0 += BLK) {
        const int mm = M - m0 < BLK ? M - m0 : BLK;
        float dd[BLK];
        for (int k = 0; k < mm; ++k) {
            const float* mp = map + 4 * (size_t)(m0 + k);
            const float dx = px - mp[0], dy = py - mp[1], dz = pz - mp[2];
            float d = dx * dx; d = d + dy * dy; d = d + dz * dz;
            dd[k] = d;
        }
        for (int kk = 0; kk < mm; ++kk) {
            const float d = dd[kk];
            if (d < bd[4]) {
                int k = 4;
                while (k > 0 && d < bd[k - 1]) { bd[k] = bd[k - 1]; bi[k] = bi[k - 1]; --k; }
                bd[k] = d; bi[k] = m0 + kk;
            }
        }
    }
}

/* ---- a grid index for the CPU baseline (bench.py): the same exact answer as the brute force for every query that passes the
 * radius gate.  Cells of edge sqrt(kd_max_radius) (the gate compares the SQUARED 5th distance with kd_max_radius, quirk Q1), so
 * every map point within the gate radius of a query lies in the 27 cells around the query's cell; candidates are ranked by
 * (float distance, map index) exactly as knn5_brute ranks them.  A query whose fifth candidate fails the gate is rejected by
 * the caller either way (with the brute force its true fifth neighbour is at least as far). */
typedef struct { double edge; float lo[3]; int dim[3]; int* start; int* order; } orc_grid;
static int grid_cell(const orc_grid* g, float x, float y, float z, int c[3]) {
    const float p[3] = {x, y, z};
    for (int k = 0; k < 3; ++k) { c[k] = (int)floor(((double)p[k] - (double)g->lo[k]) / g->edge); }
    return c[0] >= -1 && c[0] <= g->dim[0] && c[1] >= -1 && c[1] <= g->dim[1] && c[2] >= -1 && c[2] <= g->dim[2];
}
static int grid_build(orc_grid* g, const float* map, int M, double edge) {
    float hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    g->edge = edge; g->lo[0] = g->lo[1] = g->lo[2] = FLT_MAX; g->start = NULL; g->order = NULL;
    for (int m = 0; m < M; ++m) for (int k = 0; k < 3; ++k) { const float v = map[4 * (size_t)m + k]; if (v < g->lo[k]) g->lo[k] = v; if (v > hi[k]) hi[k] = v; }
    double cells = 1;
    for (int k = 0; k < 3; ++k) { g->dim[k] = M > 0 ? (int)floor(((double)hi[k] - (double)g->lo[k]) / edge) + 1 : 1; cells *= g->dim[k]; }
    if (M <= 0 || cells > 64e6) return 0;
    const size_t nc = (size_t)cells;
    g->start = (int*)calloc(nc + 1, sizeof(int));
    g->order = (int*)malloc(sizeof(int) * (size_t)M);
    int* cid = (int*)malloc(sizeof(int) * (size_t)M);
    for (int m = 0; m < M; ++m) {
        int c[3];
        grid_cell(g, map[4 * (size_t)m], map[4 * (size_t)m + 1], map[4 * (size_t)m + 2], c);
        for (int k = 0; k < 3; ++k) { if (c[k] < 0) c[k] = 0; if (c[k] >= g->dim[k]) c[k] = g->dim[k] - 1; }
        cid[m] = (c[0] * g->dim[1] + c[1]) * g->dim[2] + c[2];
        g->start[cid[m] + 1]++;
    }
    for (size_t c = 0; c < nc; ++c) g->start[c + 1] += g->start[c];
    int* fill = (int*)malloc(sizeof(int) * nc);
    memcpy(fill, g->start, sizeof(int) * nc);
    for (int m = 0; m < M; ++m) g->order[fill[cid[m]]++] = m;          /* ascending map index inside a cell */
    free(fill); free(cid);
    return 1;
}
static void knn5_grid(const orc_grid* g, const float* map, float px, float py, float pz, float bd[5], int bi[5]) {
    for (int k = 0; k < 5; ++k) { bd[k] = FLT_MAX; bi[k] = -1; }
    int c[3];
    if (!grid_cell(g, px, py, pz, c)) return;
    for (int dx = -1; dx <= 1; ++dx) for (int dy = -1; dy <= 1; ++dy) for (int dz = -1; dz <= 1; ++dz) {
        const int x = c[0] + dx, y = c[1] + dy, z = c[2] + dz;
        if (x < 0 || y < 0 || z < 0 || x >= g->dim[0] || y >= g->dim[1] || z >= g->dim[2]) continue;
        const int cell = (x * g->dim[1] + y) * g->dim[2] + z;
        for (int s = g->start[cell]; s < g->start[cell + 1]; ++s) {
            const int m = g->order[s];
            const float* mp = map + 4 * (size_t)m;
            const float ex = px - mp[0], ey = py - mp[1], ez = pz - mp[2];
            float d = ex * ex; d = d + ey * ey; d = d + ez * ez;
            if (d < bd[4] || (d == bd[4] && m < bi[4])) {
                int k = 4;
                while (k > 0 && (d < bd[k - 1] || (d == bd[k - 1] && m < bi[k - 1]))) { bd[k] = bd[k - 1]; bi[k] = bi[k - 1]; --k; }
                bd[k] = d; bi[k] = m;
            }
        }
    }
}
static int g_assoc_use_grid = 0;
/* bench.py's CPU baseline: 1 = index the map with the grid above (same records as the brute force; falls back to it when the
 * map's bounding box has more than 6.4e7 cells); parity tests keep 0 */
void orc_set_assoc_grid(int on) { g_assoc_use_grid = on; }

int orc_associate(const glio_opts* o, const float* map, int M, const float* scan, int n,
                  const double q[4], const double t[3], float* out_pts, float* out_planes,
                  double* out_scores, int32_t* out_src, int32_t* out_nn) {
    return orc_associate_mt(o, map, M, scan, n, q, t, out_pts, out_planes, out_scores, out_src, out_nn, 1);
}

/* The same with the nearest-neighbour phase (the O(n M) part) spread over `threads` OpenMP threads; the rest runs in scan
 * order as before, so the output is identical for every thread count.  Full-size parity checks (131 072 queries against a
 * map of > 10^6 points are 1.5e11 distance evaluations) use this on the GPU box's host cores. */
int orc_associate_mt(const glio_opts* o, const float* map, int M, const float* scan, int n,
                     const double q[4], const double t[3], float* out_pts, float* out_planes,
                     double* out_scores, int32_t* out_src, int32_t* out_nn, int threads) {
    int cnt = 0;
    int* all_bi = NULL; float* all_bd = NULL;
    orc_grid grid;
    const int use_grid = g_assoc_use_grid && grid_build(&grid, map, M, sqrt(o->kd_max_radius));
    if (threads > 1) {
        all_bi = (int*)malloc(sizeof(int) * 5 * (size_t)(n > 0 ? n : 1));
        all_bd = (float*)malloc(sizeof(float) * 5 * (size_t)(n > 0 ? n : 1));
#pragma omp parallel for schedule(dynamic, 256) num_threads(threads)
        for (int i = 0; i < n; ++i) {
            const float* pl = scan + 4 * (size_t)i;
            double pin[3] = {pl[0], pl[1], pl[2]}, pout[3];
            q_rot(q, pin, pout);
            const float px = (float)(pout[0] + t[0]), py = (float)(pout[1] + t[1]), pz = (float)(pout[2] + t[2]);
            if (use_grid) knn5_grid(&grid, map, px, py, pz, all_bd + 5 * (size_t)i, all_bi + 5 * (size_t)i);
            else knn5_brute(map, M, px, py, pz, all_bd + 5 * (size_t)i, all_bi + 5 * (size_t)i);
        }
    }
    for (int i = 0; i < n; ++i) {
        const float* pl = scan + 4 * (size_t)i;
        /* transformPoint: double math, float store */
        double pin[3] = {pl[0], pl[1], pl[2]}, pout[3];
        q_rot(q, pin, pout);
        const float px = (float)(pout[0] + t[0]), py = (float)(pout[1] + t[1]), pz = (float)(pout[2] + t[2]);
        /* exact 5-NN, ascending (dist, index) */
        float bd[5];
        int bi[5];
        if (all_bi) { for (int k = 0; k < 5; ++k) { bd[k] = all_bd[5 * (size_t)i + k]; bi[k] = all_bi[5 * (size_t)i + k]; } }
        else if (use_grid) knn5_grid(&grid, map, px, py, pz, bd, bi);
        else knn5_brute(map, M, px, py, pz, bd, bi);
        if (out_nn) for (int k = 0; k < 5; ++k) out_nn[5 * (size_t)i + k] = bi[k];
        if (!(bi[4] >= 0 && (double)bd[4] < o->kd_max_radius)) continue;              /* :3651 */
        double A[15], b[5] = {-1, -1, -1, -1, -1}, nrm[3];
        for (int k = 0; k < 5; ++k) for (int c = 0; c < 3; ++c) A[k * 3 + c] = (double)map[4 * (size_t)bi[k] + c];
        orc_plane_qr_solve(A, b, nrm);                                           /* :3661 */
        const double nn = v3_norm(nrm);
        const double normInverse = 1.0 / nn;                                     /* :3662 */
        nrm[0] /= nn; nrm[1] /= nn; nrm[2] /= nn;                                /* :3663 */
        int valid = 1;
        for (int k = 0; k < 5; ++k)
            if (fabs(nrm[0] * A[k * 3] + nrm[1] * A[k * 3 + 1] + nrm[2] * A[k * 3 + 2] + normInverse) > o->surf_dist_thres) { valid = 0; break; }
        if (!valid) continue;                                                    /* :3667-3674 */
        const float pd = (float)(nrm[0] * (double)px + nrm[1] * (double)py + nrm[2] * (double)pz + normInverse);  /* :3678 */
        const float rr = sqrtf(sqrtf(px * px + py * py + pz * pz));              /* float sqrt(sqrt(.)) */
        const float weight = (float)(1.0 - 0.9 * (double)fabsf(pd) / (double)rr);  /* :3679 */
        if (!((double)weight > o->weight_gate)) continue;                                /* :3681 */
        float* op = out_pts + 4 * (size_t)cnt;
        float* on = out_planes + 4 * (size_t)cnt;
        op[0] = pl[0]; op[1] = pl[1]; op[2] = pl[2]; op[3] = pl[3];              /* :3688 */
        on[0] = (float)((double)weight * nrm[0]);                                /* :3683-3686 */
        on[1] = (float)((double)weight * nrm[1]);
        on[2] = (float)((double)weight * nrm[2]);
        on[3] = (float)((double)weight * normInverse);
        out_scores[cnt] = o->unit_scores ? 1.0 : o->lidar_const * (double)weight;  /* :3692 */
        if (out_src) out_src[cnt] = i;
        ++cnt;
    }
    free(all_bi); free(all_bd);
    if (use_grid) { free(grid.start); free(grid.order); }
    return cnt;
}

/* ------------------------------------------------------------------------------------------------
 * findGlobalCorrespondingSurfFeaturesAdd_Batch for ONE (idx, search_idx) pair
 * (reference GLIO/src/Estimator.cpp:3808-3892; the _Batch twin at :3711-3806 is the same arithmetic).
 * scan_a = surf_frames[idx], scan_b = surf_frames[search_idx] (PointXYZI as 4 floats, keyframe-local);
 * (qa,ta) / (qb,tb) = pose_info_keyframe->points[idx / search_idx].  Both clouds go to the global frame with
 * transformCloud (:1517-1546: double q*v + t, float store); 5-NN of every point of a in global b; gate
 * sqd[4] < 1.5 (:3839); two 5x3 colPivHouseholderQr fits, global and keyframe-local coordinates of the same five
 * neighbours (:3855-3859); validity with the global plane, 0.18 (:3861-3869); pd / weight in float (:3872-3873);
 * kept if weight > 0.3 (:3874).  Output record (:3879-3887): the point of a in ITS local frame, score 2.5 w,
 * [unit local normal of b | centroid of the five neighbours in b's local frame].
 * ------------------------------------------------------------------------------------------------ */
int orc_associate_pair(const float* scan_a, int na, const double qa[4], const double ta[3],
                       const float* scan_b, int nb, const double qb[4], const double tb[3],
                       float* out_cp, double* out_norm_cent, double* out_score, int32_t* out_src) {
    float* gb = (float*)malloc(sizeof(float) * 4 * (size_t)(nb > 0 ? nb : 1));
    for (int m = 0; m < nb; ++m) {
        double pin[3] = {scan_b[4 * (size_t)m], scan_b[4 * (size_t)m + 1], scan_b[4 * (size_t)m + 2]}, po[3];
        q_rot(qb, pin, po);
        gb[4 * (size_t)m] = (float)(po[0] + tb[0]); gb[4 * (size_t)m + 1] = (float)(po[1] + tb[1]); gb[4 * (size_t)m + 2] = (float)(po[2] + tb[2]); gb[4 * (size_t)m + 3] = 0.f;
    }
    /* (bench.py's CPU baseline indexes the search cloud with the grid above: the same records as the scan over every point, see orc_set_assoc_grid) */
    orc_grid grid;
    const int use_grid = g_assoc_use_grid && grid_build(&grid, gb, nb, sqrt(1.5));
    int cnt = 0;
    for (int i = 0; i < na; ++i) {
        const float* pl = scan_a + 4 * (size_t)i;
        double pin[3] = {pl[0], pl[1], pl[2]}, pout[3];
        q_rot(qa, pin, pout);
        const float px = (float)(pout[0] + ta[0]), py = (float)(pout[1] + ta[1]), pz = (float)(pout[2] + ta[2]);
        float bd[5] = {FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX};
        int bi[5] = {-1, -1, -1, -1, -1};
        if (use_grid) knn5_grid(&grid, gb, px, py, pz, bd, bi);
        else for (int m = 0; m < nb; ++m) {
            const float* mp = gb + 4 * (size_t)m;
            const float dx = px - mp[0], dy = py - mp[1], dz = pz - mp[2];
            float d = dx * dx; d = d + dy * dy; d = d + dz * dz;
            if (d < bd[4]) {
                int k = 4;
                while (k > 0 && d < bd[k - 1]) { bd[k] = bd[k - 1]; bi[k] = bi[k - 1]; --k; }
                bd[k] = d; bi[k] = m;
            }
        }
        if (!(bi[4] >= 0 && (double)bd[4] < 1.5)) continue;                                  /* :3839 */
        double A[15], Al[15], b[5] = {-1, -1, -1, -1, -1}, bl[5] = {-1, -1, -1, -1, -1}, nrm[3], nl[3];
        double cx = 0, cy = 0, cz = 0;
        for (int k = 0; k < 5; ++k) {
            for (int c = 0; c < 3; ++c) {
                A[k * 3 + c] = (double)gb[4 * (size_t)bi[k] + c];
                Al[k * 3 + c] = (double)scan_b[4 * (size_t)bi[k] + c];
            }
            cx += Al[k * 3]; cy += Al[k * 3 + 1]; cz += Al[k * 3 + 2];                /* :3848-3850 */
        }
        orc_plane_qr_solve(A, b, nrm);                                                /* :3856 */
        const double nn = v3_norm(nrm);
        const double normInverse = 1.0 / nn;
        nrm[0] /= nn; nrm[1] /= nn; nrm[2] /= nn;
        orc_plane_qr_solve(Al, bl, nl);                                               /* :3858 */
        const double nln = v3_norm(nl);
        nl[0] /= nln; nl[1] /= nln; nl[2] /= nln;
        int valid = 1;
        for (int k = 0; k < 5; ++k)
            if (fabs(nrm[0] * A[k * 3] + nrm[1] * A[k * 3 + 1] + nrm[2] * A[k * 3 + 2] + normInverse) > 0.18) { valid = 0; break; }
        if (!valid) continue;
        const float pd = (float)(nrm[0] * (double)px + nrm[1] * (double)py + nrm[2] * (double)pz + normInverse);
        const float rr = sqrtf(sqrtf(px * px + py * py + pz * pz));
        const float weight = (float)(1.0 - 0.9 * (double)fabsf(pd) / (double)rr);
        if (!((double)weight > 0.3)) continue;                                        /* :3874, double comparison */
        float* op = out_cp + 4 * (size_t)cnt;
        op[0] = pl[0]; op[1] = pl[1]; op[2] = pl[2]; op[3] = pl[3];
        double* nc = out_norm_cent + 6 * (size_t)cnt;
        nc[0] = nl[0]; nc[1] = nl[1]; nc[2] = nl[2];
        nc[3] = cx / 5.; nc[4] = cy / 5.; nc[5] = cz / 5.;
        out_score[cnt] = 2.5 * (double)weight;                                        /* :3885 */
        if (out_src) out_src[cnt] = i;
        ++cnt;
    }
    free(gb);
    if (use_grid) { free(grid.start); free(grid.order); }
    return cnt;
}

/* ------------------------------------------------------------------------------------------------
 * pcl::VoxelGrid<PointXYZI>::applyFilter as used by downSampleCloud (Estimator.cpp:3618-3631,
 * ds_filter_surf_map, leaf = surf_ds_size 0.4 m :854): PCL 1.8 semantics restated (PCL itself is an external
 * dependency, not under /root/reference): bounding box -> min_b = floor(min * inv_leaf); voxel of a point =
 * (int)(floor(x * inv_leaf) - (float)min_b) per axis, linear index i + j*div0 + k*div0*div1; points grouped by index,
 * centroid of x,y,z,intensity accumulated in float; output ordered by voxel index.  PCL sorts with std::sort (order
 * inside a voxel unspecified); here points of a voxel are accumulated in input order.
 * Returns the number of voxels; out [n][4], out_idx [n] (optional) the linear voxel index of every output point.
 * ------------------------------------------------------------------------------------------------ */
typedef struct { int64_t idx; int src; } vg_item;
static int vg_cmp(const void* a, const void* b) {
    const vg_item* x = (const vg_item*)a; const vg_item* y = (const vg_item*)b;
    if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
    return x->src < y->src ? -1 : (x->src > y->src);
}
int orc_voxel_grid(const float* pts, int n, float leaf, float* out, int64_t* out_idx) {
    if (n <= 0) return 0;
    const float inv = 1.0f / leaf;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = 0; i < n; ++i) for (int c = 0; c < 3; ++c) { const float v = pts[4 * (size_t)i + c]; if (v < mn[c]) mn[c] = v; if (v > mx[c]) mx[c] = v; }
    int min_b[3], div_b[3];
    for (int c = 0; c < 3; ++c) { min_b[c] = (int)floorf(mn[c] * inv); div_b[c] = (int)floorf(mx[c] * inv) - min_b[c] + 1; }
    vg_item* it = (vg_item*)malloc(sizeof(vg_item) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        const float* p = pts + 4 * (size_t)i;
        const int i0 = (int)(floorf(p[0] * inv) - (float)min_b[0]);
        const int i1 = (int)(floorf(p[1] * inv) - (float)min_b[1]);
        const int i2 = (int)(floorf(p[2] * inv) - (float)min_b[2]);
        it[i].idx = (int64_t)i0 + (int64_t)i1 * div_b[0] + (int64_t)i2 * div_b[0] * (int64_t)div_b[1];
        it[i].src = i;
    }
    qsort(it, (size_t)n, sizeof(vg_item), vg_cmp);
    int nv = 0;
    for (int i = 0; i < n;) {
        int j = i;
        float acc[4] = {0, 0, 0, 0};
        while (j < n && it[j].idx == it[i].idx) { const float* p = pts + 4 * (size_t)it[j].src; for (int c = 0; c < 4; ++c) acc[c] += p[c]; ++j; }
        const float cnt = (float)(j - i);
        for (int c = 0; c < 4; ++c) out[4 * (size_t)nv + c] = acc[c] / cnt;
        if (out_idx) out_idx[nv] = it[i].idx;
        ++nv;
        i = j;
    }
    free(it);
    return nv;
}

/* transformCloud (Estimator.cpp:1517-1546): double q*v + t, float store; intensity copied */
void orc_transform_cloud(const float* in, int n, const double q[4], const double t[3], float* out) {
    for (int i = 0; i < n; ++i) {
        double pin[3] = {in[4 * (size_t)i], in[4 * (size_t)i + 1], in[4 * (size_t)i + 2]}, po[3];
        q_rot(q, pin, po);
        out[4 * (size_t)i] = (float)(po[0] + t[0]); out[4 * (size_t)i + 1] = (float)(po[1] + t[1]); out[4 * (size_t)i + 2] = (float)(po[2] + t[2]);
        out[4 * (size_t)i + 3] = in[4 * (size_t)i + 3];
    }
}
