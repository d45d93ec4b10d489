/*
 * glio_hip.h -- C-ABI of libglio_hip.so: the MI355X (gfx950) implementation of GLIO's
 * sliding-window hot path, Estimator::optimizeSlidingWindowWithLandMark
 * (reference: GLIO/src/Estimator.cpp:2046-2736).
 *
 * This is the drop-in boundary.  The reference has no plugin API; what it has are
 * (a) the Estimator member buffers the function reads and writes, and (b) the
 * ceres::CostFunction::Evaluate(double const* const*, double*, double**) contract of every factor.
 * Each entry point below names the reference interface it replaces.  Signatures carry plain
 * pointers and sizes only; every pointer is caller-owned HOST memory unless the name ends in
 * `_dev`.  All functions return 0 on success, a negative GLIO_E_* code otherwise; nothing throws.
 * A context owns one HIP stream and is not thread-safe; independent contexts (sliding window vs
 * batch thread, Estimator.cpp:5398-5404) may be used concurrently.
 */
#ifndef GLIO_HIP_H_
#define GLIO_HIP_H_

#include "glio_types.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
    GLIO_OK = 0,
    GLIO_E_ARG = -1,        /* bad argument / capacity exceeded */
    GLIO_E_HIP = -2,        /* HIP runtime error (no device, launch failure ...) */
    GLIO_E_STATE = -3,      /* call order violated (e.g. solve before any factor was set) */
    GLIO_E_NUMERIC = -4     /* solver failure (Cholesky breakdown with mu >= 1) */
};

typedef struct glio_ctx glio_ctx;

/* Library / device info.  `glio_device_count` < 1 means the HIP path is unusable: callers must fail. */
int glio_abi_version(void);
int glio_device_count(void);
const char* glio_last_error(void);
/* sizeof() of every POD struct, in the order opts,state,preint,prior,dd_psr,doppler,gnss_frame,summary,batch_tr_opts
 * (lets a foreign-language binding verify its struct layout); returns how many there are. */
int glio_struct_sizes(int32_t* out, int n);

/* yaml defaults: GLIO/config/config_urban_hk.yaml:60-104 + Estimator.cpp:70,2424-2430 */
void glio_opts_default(glio_opts* o);

/* Replaces: construction of the per-window Ceres problem + device residency of Estimator members. */
int glio_create(int device, const glio_opts* opts, glio_ctx** out);
void glio_destroy(glio_ctx* ctx);
/* Optional: run on a caller-provided hipStream_t (e.g. torch's current stream). NULL restores the own stream. */
int glio_set_stream(glio_ctx* ctx, void* hip_stream);
int glio_synchronize(glio_ctx* ctx);

/* ---- K1: local map.  Replaces kd_tree_surf_local_map->setInputCloud(surf_local_map_ds)
 * (Estimator.cpp:2056).  pts = PointXYZI[n] as 4 floats; builds the voxel hash on device. */
int glio_set_map(glio_ctx* ctx, const float* map_xyzi, int n);
/* ---- strided point input.  The reference hands its clouds around as pcl::PointCloud<PointType>, PointType = pcl::PointXYZI (GLIO/include/utils/common.h):
 * records of 32 bytes, x y z as floats at bytes 0-11, the intensity at byte 16.  Every cloud-taking entry point has a *_strided twin that takes
 * cloud->points.data() as it is: stride_bytes per record (>= 16, a multiple of 4), x y z at offset 0, intensity_offset in [12, stride - 4].  The raw
 * records travel in ONE copy and are unpacked on the device -- no packing pass over the scan / the map on the host.  (stride 16, offset 12 = the packed
 * float4 form the plain entry points take.)  Replaces the cloud hand-over of Estimator.cpp:2056 (map), :2198-2248 (scans), :3529-3631 (local map). */
int glio_set_map_strided(glio_ctx* ctx, const void* map_points, int n, int stride_bytes, int intensity_offset);

/* ---- device-resident local map.  Replaces buildLocalMapWithLandMark + downSampleCloud + setInputCloud
 * (Estimator.cpp:3529-3631, 2056) by a ring of the last `width` keyframe clouds kept on the device in the map frame:
 * per keyframe ONE scan crosses PCIe (glio_localmap_push), the concatenation is voxel-averaged (pcl::VoxelGrid
 * semantics, leaf = surf_ds_size) and hashed on the device (glio_localmap_build does what glio_set_map does).
 *   width = local_map_width (yaml: 50); q,t = q_po * q_bl, q_po * t_bl + t_po (:3569-3570). */
int glio_localmap_config(glio_ctx* ctx, int width, float leaf, int max_points_per_keyframe);
int glio_localmap_push(glio_ctx* ctx, const float* cloud_xyzi, int n, const double q[4], const double t[3]);
int glio_localmap_push_strided(glio_ctx* ctx, const void* cloud_points, int n, int stride_bytes, int intensity_offset, const double q[4], const double t[3]);
/* the same from the scan glio_set_scan already put into window slot `scan_slot` (LiDAR frame; body point = scan point - lidar_offset in float):
 * the newest keyframe's cloud crosses PCIe once for both the association and the map */
int glio_localmap_push_scan(glio_ctx* ctx, int scan_slot, const float lidar_offset[3], const double q[4], const double t[3]);
/* Voxel grid + hash of the ring's content; *out_points = the map's size.  The call waits ONCE in its middle (the voxel count sizes the rest) and returns with
 * the ordered output and the hash build still running on the context's stream: the searches are ordered behind them, glio_localmap_read waits. */
int glio_localmap_build(glio_ctx* ctx, int* out_points);
/* centroid arithmetic of the voxel grid: 0 (default) exact fixed-point sums; 1 = float sums in the order of the concatenated cloud, as the oracle's
 * restatement of pcl::VoxelGrid forms them (Estimator.cpp:3618-3631 through PCL): bit-identical to the ORACLE's map (stable order inside a voxel).  PCL itself
 * orders its point / voxel index vector with an unstable sort, so against a real PCL build the float sums may differ by an ulp -- the order inside a voxel is
 * not specified by PCL.  One extra pass over the ring per build.  The mode survives glio_localmap_config (re-applied to the new ring). */
int glio_localmap_set_accumulation(glio_ctx* ctx, int mode);
/* test hook: the down-sampled map (surf_local_map_ds), ordered by voxel index */
int glio_localmap_read(glio_ctx* ctx, float* out_xyzi, int capacity, int* out_n);

/* ---- K2: correspondences.  Replaces findCorrespondingSurfFeatures(idx, Q2, T2)
 * (Estimator.cpp:3633-3708) for window slot `slot`: uploads the scan (surf_frames[idx], PointXYZI[n],
 * LiDAR frame), runs exact 5-NN + plane fit + gates on device and leaves the compacted
 * vec_surf_cur_pts / vec_surf_normal / vec_surf_scores of that slot resident.  q,t = the LiDAR pose
 * Q2,T2 of Estimator.cpp:2216-2217.  *out_count receives vec_surf_res_cnt[slot]. */
int glio_associate(glio_ctx* ctx, int slot, const float* scan_xyzi, int n, const double q[4],
                   const double t[3], int* out_count);
/* Same, for a scan already resident from a previous glio_associate/glio_set_scan (re-association).
 * glio_set_scan returns when the caller's buffer has been read; the presort of the scan is enqueued behind the copy and not waited for. */
int glio_set_scan(glio_ctx* ctx, int slot, const float* scan_xyzi, int n);
int glio_set_scan_strided(glio_ctx* ctx, int slot, const void* scan_points, int n, int stride_bytes, int intensity_offset);
/* The NEXT keyframe's scan, sent while the current keyframe's call is still running (the reference hands surf_frames over from the front end before
 * optimizeSlidingWindowWithLandMark runs, Estimator.cpp:5372ff): it goes into the ring row that becomes slot W - 1 with the next glio_slide_window -- the row of
 * the CURRENT slot 0, whose scan is gone afterwards (no re-association of slot 0) -- on a stream of its own, beside the call's kernels, the presort behind it.
 * To be called once the window's association is through (after glio_associate_window_counts / the solve).  The next glio_slide_window takes the scan over; that
 * call then makes no glio_set_scan for slot W - 1.  Returns when the caller's buffer has been read. */
int glio_set_scan_ahead(glio_ctx* ctx, const float* scan_xyzi, int n);
int glio_set_scan_ahead_strided(glio_ctx* ctx, const void* scan_points, int n, int stride_bytes, int intensity_offset);
/* ... and, behind it, the next call's local map: glio_localmap_push_scan of the cloud just sent ahead at the new keyframe's pose + glio_localmap_build, on the same
 * stream beside the call's tail (the new keyframe's initial pose follows from this call's solve and the odometry: buildLocalMapWithLandMark pushes each keyframe
 * once, at the pose it has when it arrives, Estimator.cpp:3585-3616).  The next call then makes neither call; its glio_slide_window waits for the event. */
int glio_localmap_push_scan_ahead_and_build(glio_ctx* ctx, const float lidar_offset[3], const double q[4], const double t[3], int* out_points);
int glio_associate_resident(glio_ctx* ctx, int slot, const double q[4], const double t[3], int* out_count);
/* Slide the window by one keyframe: the resident scan of slot s+1 becomes that of slot s (the scans are a ring on the device: nothing is
 * copied, nothing waited for); slot W-1 is free for the new keyframe's glio_set_scan.  (surf_frames / keyframe_idx bookkeeping of
 * Estimator.cpp:4240-4300.) */
int glio_slide_window(glio_ctx* ctx);
/* The whole loop of Estimator.cpp:2198-2248 in one call: every slot's resident scan against the map with its own
 * LiDAR pose (quats [W][4], trans [W][3] = Q2, T2 per slot), one host synchronisation; out_counts [W]. */
int glio_associate_window(glio_ctx* ctx, const double* quats, const double* trans, int32_t* out_counts);
/* the same in two halves: _async enqueues the searches and returns (the host is free for glio_set_imu / glio_set_gnss while the GPU searches);
 * _counts waits and returns the per-slot counts (optional: every entry point that needs the correspondences waits by itself) */
int glio_associate_window_async(glio_ctx* ctx, const double* quats, const double* trans);
int glio_associate_window_counts(glio_ctx* ctx, int32_t* out_counts);
/* featureSelection (Estimator.cpp:3894-3992): keep records indices[0..n) of the slot, in that order (n = 0 empties the
 * slot, the reference's random_select == false case :3945,3981-3987).  The random draws stay with the caller (the
 * reference seeds from std::random_device, random_generator.hpp:58, so they are not reproducible anyway); the gather
 * runs on the device, nothing is read back.  glio_amd/sliding.py::feature_selection restates the draw procedure. */
int glio_select_correspondences(glio_ctx* ctx, int slot, const int32_t* indices, int n);
/* The same for every slot of the window in ONE call -- what the released configuration does in every keyframe call (featureSelection behind each slot's
 * search, Estimator.cpp:2222-2223; feature_res_num 100 of ~4 k records per slot): slot s keeps records indices[offsets[s] .. offsets[s+1]) of its own set, in
 * that order (offsets[0] = 0, W + 1 entries); changed[s] == 0 leaves slot s untouched (the early return of :3906-3909), changed == NULL changes all.  One
 * upload, two launches, no host wait: the solve that follows on the context is ordered behind it. */
int glio_select_correspondences_window(glio_ctx* ctx, const int32_t* offsets, const int32_t* indices, const uint8_t* changed);
/* Parity hook / featureSelection replacement: provide or read back a slot's correspondence arrays. */
int glio_set_correspondences(glio_ctx* ctx, int slot, const float* pts_xyzi, const float* planes,
                             const double* scores, int n);
int glio_get_correspondences(glio_ctx* ctx, int slot, float* pts_xyzi, float* planes, double* scores,
                             int capacity, int* out_count);

/* ---- factors of the window problem
 * glio_set_imu / glio_set_gnss may be called while the window's searches run (glio_associate_window_async): their tables travel as one pinned block
 * on a stream of the library's own into a device mirror, the call waits for that copy alone, and the kernel that installs the tables is enqueued on
 * the context's stream behind everything that still reads the old ones.  (GLIO_EARLY_UPLOAD=0 in the environment: copy and wait on the context's
 * stream, for A/B runs.) */
/* Replaces problem.AddResidualBlock(new ImuFactor(pre_integrations[idx+1]), NULL, ...)
 * (Estimator.cpp:2182-2192).  Edge `k` links slots slot_i and slot_i+1. */
int glio_set_imu(glio_ctx* ctx, int n_edges, const glio_preint* edges, const int32_t* slot_i);
/* Replaces problem.AddResidualBlock(new MarginalizationFactor(last_marginalization_info), NULL,
 * last_marginalization_parameter_blocks) (Estimator.cpp:2153-2158).  prior->n == 0 removes it. */
int glio_set_prior(glio_ctx* ctx, const glio_prior* prior);
/* Replaces addDDPsrResFactor (Estimator.cpp:1893-1897) and the tcdopplerFactor blocks
 * (Estimator.cpp:2329-2337); para_yaw_enu_local / para_anc_ecef are the constant blocks. */
int glio_set_gnss(glio_ctx* ctx, const glio_gnss_frame* frame, int n_dd, const glio_dd_psr* dd,
                  int n_dop, const glio_doppler* dop);

/* ---- the solve.  Replaces ceres::Solve(options, &problem, &summary) (Estimator.cpp:2424-2433):
 * state in = tmpTrans/tmpQuat/tmpSpeedBias(/para_rcv_ddt) before, out = after. */
int glio_solve(glio_ctx* ctx, glio_state* state_inout, glio_summary* summary);
/* One linearisation at `state`: dense H = J^T J (n x n row-major, n = 15 W + state->n_ddt),
 * g = J^T r, cost; after loss correction and local parameterisation, unscaled.  H/g may be NULL.
 * Exposed so that parity can be checked per linearisation (SURVEY.md section 7 "hard parts"). */
int glio_linearize(glio_ctx* ctx, const glio_state* state, double* H, double* g, double* cost);

/* ---- marginalization of the oldest keyframe.  Replaces the MarginalizationInfo block of
 * Estimator.cpp:2462-2607 (addResidualBlockInfo x {prior, IMU(0,1), every LidarPlaneNormFactor with Huber},
 * preMarginalize, marginalize, getParameterBlocks(addr_shift)) on the factors currently set in the context,
 * evaluated at `state` (the solution of glio_solve).  Outputs, caller-allocated for n = 6 (W-1) + 9 and
 * nb = 2 (W-1) + 1: lin_jac [n][n] row-major, lin_res [n], blocks (slot already shifted s -> s-1, kind, first
 * column, x0[9]) -- exactly the fields of glio_prior for the NEXT window.  lin_jac^T lin_jac and
 * lin_jac^T lin_res equal the reference's (it factors the Schur complement by eigen-decomposition, this
 * library by Cholesky: a different square root of the same matrix).  GLIO_E_NUMERIC if the Schur
 * complement is not positive definite. */
int glio_marginalize(glio_ctx* ctx, const glio_state* state, double* lin_jac, double* lin_res,
                     int32_t* blk_slot, int32_t* blk_kind, int32_t* blk_idx, double* blk_x0,
                     int32_t* out_n, int32_t* out_n_blocks);

/* The same, but the result is installed as THIS context's prior for the next window without leaving the device
 * (= glio_marginalize + glio_set_prior of its output, minus the two PCIe trips of the n x n matrix).  The caller then
 * slides its state arrays / scans / IMU edges by one keyframe as the reference does (Estimator.cpp:2584-2607, 4300ff). */
int glio_marginalize_keep(glio_ctx* ctx, const glio_state* state);
/* the same in two halves: _async enqueues everything and returns (the host is free, e.g. for glio_set_scan_ahead, while the GPU marginalizes); _finish waits and
 * reports GLIO_E_NUMERIC if the Schur complement was not positive definite (the context is then left without a prior).  Every entry point that reads the prior
 * finishes a pending marginalization by itself. */
int glio_marginalize_keep_async(glio_ctx* ctx, const glio_state* state);
int glio_marginalize_keep_finish(glio_ctx* ctx);

/* ---- single-factor evaluators with the exact Evaluate() pointer convention, computed on the GPU.
 * A ceres::CostFunction shim is a five-line wrapper around these. */
/* LidarPlaneNormFactor (LidarKeyframeFactor.h:73-122): parameters = {t[3], q[4]} */
int glio_eval_lidar_plane(glio_ctx* ctx, const float cp[4], const float plane[4], double score,
                          double const* const* parameters, double* residuals, double** jacobians);
/* ImuFactor::Evaluate (ImuFactor.h:21-171): parameters = {Pi,Qi,SBi,Pj,Qj,SBj} */
int glio_eval_imu(glio_ctx* ctx, const glio_preint* pre, double const* const* parameters,
                  double* residuals, double** jacobians);

/* dd_psr_factor_20::Evaluate (dd_psr_factor.hpp:25-171): parameters = {Pi[3], Pj[3], yaw[1], anc[3]}, 19 residuals
 * (rows >= n_sat-1 zero), jacobians[0..1] 19x3 row-major; jacobians[2..3] are not written (the reference leaves them) */
int glio_eval_dd_psr(glio_ctx* ctx, const glio_dd_psr* f, double const* const* parameters, double* residuals,
                     double** jacobians);
/* tcdopplerFactor (dopp_factor.hpp:24-75): parameters = {Pi[3], SBi[9], Pj[3], SBj[9], rcv_ddt[>epoch], yaw[1], anc[3]},
 * 1 residual; jacobians[0..3] 1x3 / 1x9 / 1x3 / 1x9, jacobians[4] receives d r / d rcv_ddt[epoch] as ONE double */
int glio_eval_doppler(glio_ctx* ctx, const glio_doppler* f, double const* const* parameters, double* residuals,
                      double** jacobians);
/* MarginalizationFactor::Evaluate (MarginalizationFactor.cpp:233-287): parameters[b] = kept block b (3, 4 or 9 doubles),
 * prior->n residuals, jacobians[b] n x size_b row-major */
int glio_eval_marginalization(glio_ctx* ctx, const glio_prior* prior, double const* const* parameters,
                              double* residuals, double** jacobians);
/* BinaryLidarPlaneNormFactor (LidarKeyframeFactor.h:124-164): parameters = {t1[3], q1[4], t2[3], q2[4]}, 1 residual */
int glio_eval_binary_plane(glio_ctx* ctx, const float cp[4], const double norm_cent[6], double score,
                           double const* const* parameters, double* residuals, double** jacobians);

/* ---- measurement hooks (bench.py): time `reps` launches of one kernel with HIP events on the
 * context's stream; returns average milliseconds per launch in *ms_out. */
enum { GLIO_KERNEL_LIDAR_LINEARIZE = 0, GLIO_KERNEL_FULL_LINEARIZE = 1, GLIO_KERNEL_TR_STEP = 2,
       GLIO_KERNEL_ASSOCIATE = 3, GLIO_KERNEL_MAP_BUILD = 4, GLIO_KERNEL_MARGINALIZE = 5,
       GLIO_KERNEL_STREAM_READ = 6 /* same bytes as LIDAR_LINEARIZE, no arithmetic: the practical ceiling */,
       GLIO_KERNEL_LINEARIZE_ALL = 7 /* the launch glio_solve uses: K3 workgroups beside the small-factor workgroups */,
       GLIO_KERNEL_TR_STEP_STEADY = 8 /* a LATER step of a solve (an accepted candidate pending, scale in place: call after a solve): what iterations 2.. cost;
                                         GLIO_KERNEL_TR_STEP is the first step of a solve (no candidate yet, the helpers' speculative build does not apply) */ };
int glio_time_kernel(glio_ctx* ctx, int which, int reps, float* ms_out);
/* time `reps` complete solves from the same initial state with HIP events (state is not modified) */
int glio_time_solve(glio_ctx* ctx, const glio_state* state, int reps, float* ms_out, glio_summary* last);


/* ================================================================================================
 * Batch stage (scan-to-multiscan), the one piece that shards over GPUs.
 * Replaces, inside Estimator::optimizeBatchWithLandMark (Estimator.cpp:2739-3410), the evaluation of all
 * BinaryLidarPlaneNormFactor residual blocks (LidarKeyframeFactor.h:124-164, built at Estimator.cpp:3004-3076,
 * no loss function :2768) and their J^T J / J^T r build.  Unknowns: K keyframe poses (t, q), local size 6 K;
 * H is block banded: block (k, k+d), d = 0..band, stored at Hg[(k*(band+1)+d)*36 ...] (row-major 6x6),
 * followed by g [K][6] and the cost (1 double): Hg has glio_batch_hg_size(K, band) doubles.
 * Each rank loads only ITS constraints.  Damped Gauss-Newton path (glio_batch_linearize_dev / _step_dev): the ranks' Hg buffers are
 * summed with one RCCL all-reduce by the caller, then every rank runs the same banded solve.  Trust-region path
 * (glio_batch_solve_tr2, below): everything is sharded, see there.
 * Pointers named *_dev are DEVICE pointers (e.g. torch tensors); the others are host memory. */
typedef struct glio_batch glio_batch;
int64_t glio_batch_hg_size(int K, int band);
int glio_batch_create(int device, int K, int band, int64_t max_constraints, glio_batch** out);
void glio_batch_destroy(glio_batch* b);
int glio_batch_set_stream(glio_batch* b, void* hip_stream);
/* constraints sorted by (ci, cj); cp [n][4] float (point in frame ci), norm_cent [n][6] double (plane normal and
 * centroid in frame cj), score [n]; |ci-cj| in 1..band */
int glio_batch_set_constraints(glio_batch* b, int64_t n, const int32_t* ci, const int32_t* cj, const float* cp,
                               const double* norm_cent, const double* score);
int glio_batch_set_constraints_dev(glio_batch* b, int64_t n, const int32_t* ci_host, const int32_t* cj_host,
                                   const float* cp_dev, const double* norm_cent_dev, const double* score_dev);
/* the same from a pair list (the output of glio_bassoc_run): pair p = (pair_ci[p], pair_cj[p]) owns pair_count[p]
 * consecutive records of the device arrays; pairs sorted by (ci, cj) */
int glio_batch_set_constraints_pairs_dev(glio_batch* b, int n_pairs, const int32_t* pair_ci, const int32_t* pair_cj,
                                         const int64_t* pair_count, const float* cp_dev, const double* norm_cent_dev,
                                         const double* score_dev);
/* (every *_dev entry point BORROWS device buffers and reads them on the library's own non-blocking stream: the caller's stream must have finished
 * producing them before the call)
 * the same for a constraint set that differs from the previous one only in the pairs marked in pair_changed (one byte per input pair; NULL = all):
 * the outer rounds of optimizeBatch re-search the first / last search_range keyframes and keep the stored interior constraints
 * (Estimator.cpp:3018-3030); the next solve then reads the constraints of the marked pairs only (glio_batch_solve_tr2). */
int glio_batch_update_constraints_pairs_dev(glio_batch* b, int n_pairs, const int32_t* pair_ci, const int32_t* pair_cj, const int64_t* pair_count,
                                            const float* cp_dev, const double* nc_dev, const double* score_dev, const uint8_t* pair_changed);
/* the same with every pair's record range given explicitly: [pair_offset[p], pair_offset[p] + pair_count[p]) (NULL = the pairs follow each other).  The caller
 * of the outer rounds keeps the stored interior constraints where they are and rewrites only the regions of the re-searched end keyframes. */
int glio_batch_update_constraints_pairs_at_dev(glio_batch* b, int n_pairs, const int32_t* pair_ci, const int32_t* pair_cj, const int64_t* pair_count,
                                               const int64_t* pair_offset, const float* cp_dev, const double* nc_dev, const double* score_dev, const uint8_t* pair_changed);
/* poses [K][7] = (t, q) host; Hg_dev device buffer of glio_batch_hg_size doubles (overwritten) */
int glio_batch_linearize_dev(glio_batch* b, const double* poses, double* Hg_dev);
/* damped Gauss-Newton step from a (reduced) Hg: solves (H + lambda diag(H)) d = -g with a block-banded Cholesky on
 * the device and returns poses (+) d; *model_decrease = -(g.d + d^T H d / 2).  poses_out may alias poses_in. */
int glio_batch_step_dev(glio_batch* b, const double* Hg_dev, double lambda, const double* poses_in, double* poses_out,
                        double* model_decrease);
/* ---- the rest of the batch problem and its trust-region solve (Estimator.cpp:2739-3410, sms_fusion_level 1).
 * glio_batch_set_small_factors: delta_q_factor_auto attitude constraints (Estimator.cpp:2831-2891; dq_const [n_dq][4] = const_diff,
 *   blocks q[dq_i], q[dq_j]) and dd_psr_factor_20 per GNSS epoch (Estimator.cpp:3197-3271; slot_i / slot_j = leftKey / rightKey,
 *   identity weight and the station position as addDDPsrResFactor_gl passes them, :1899-1911; `threshold` = DDpsr_threshold of the
 *   current outer round, :2764-2767).  Given whole on every rank; a rank keeps the factors whose first keyframe it owns.
 * glio_batch_set_imu: the ImuFactor chain between consecutive keyframes (Estimator.cpp:2990-3001; gl_tmpSpeedBias blocks :2809-2819):
 *   edges[k] = the pre-integration between keyframes k and k + 1 (the caller decides which interval that is),
 *   n_edges = K - 1, or 0 for the pose-only problem.  With the chain every keyframe has 15 unknowns (band <= 12: the reference's +-12 end windows).
 * glio_batch_set_shard / glio_batch_shard_range: this object is rank `rank` of `world`; it owns a contiguous range of whole
 *   super-blocks of keyframes (6, or 12 for bands > 6).  The constraints handed to glio_batch_set_constraints* must have their source
 *   keyframe (ci) in that range.  Call before glio_batch_set_small_factors.
 * glio_batch_solve_tr2: ceres::Solve of Estimator.cpp:3275-3284 (DOGLEG, opts->dogleg_type = SUBSPACE_DOGLEG, non-monotonic steps,
 *   max_num_iter), device resident -- the host feeds kernel groups and synchronises once per solve.  Returns, as Ceres does, the
 *   iterate of least cost (poses [K][7] in/out, speed_bias [K][9] in/out with the IMU chain) and that cost as final_cost.
 *   `allreduce` (NULL = one rank) is called with a device buffer, its length in doubles, the HIP stream it is produced and consumed on
 *   and `user`; it must leave the sum over the ranks in place ORDERED ON THAT STREAM (ncclAllReduce on it in C++;
 *   torch.distributed.all_reduce with that stream current in Python) -- the host does not wait for it.  Per trust-region iteration:
 *   the assembly buffer (band rows next to the range boundaries + diagonal + gradient + cost), the separator system of the block
 *   cyclic reduction, the Gauss-Newton step, and two 64-byte buffers of curvature sums; every rank the same sequence.
 *   The plane constraints are READ ONCE per call: BinaryLidarPlaneNormFactor's residual is linear in (R_b^T R_a, R_b^T (t_a - t_b)) and
 *   carries no loss function, so the first linearisation takes 12-dimensional moments per keyframe pair (centred at the call's poses) and
 *   every later one evaluates them -- the same sums, exact (GLIO_BATCH_MOMENTS=0 in the environment streams the constraints every time).
 *   A caller that changes the constraint set calls again (every outer round of optimizeBatch does, Estimator.cpp:3018-3030).
 * glio_batch_linearize_full: one linearisation through the same path (parity hook): diag(H) [n], g [n], cost, n = (6 | 15) K. */
typedef void (*glio_allreduce_fn)(double* dev, int64_t count, void* hip_stream, void* user);
int glio_batch_shard_range(int K, int band, int rank, int world, int32_t* lo, int32_t* hi);
int glio_batch_set_shard(glio_batch* b, int rank, int world);
int glio_batch_set_small_factors(glio_batch* b, const glio_gnss_frame* frame, int n_dq, const int32_t* dq_i, const int32_t* dq_j,
                                 const double* dq_const, int n_dd, const glio_dd_psr* dd);
/* LidarPoseFactorBatchRelativeAutoDiff (GLIO/include/factors/LidarPoseFactor.h:55-97), the relative-pose factors that ARE the scan-to-multiscan
 * constraints when sms_fusion_level == 0 (Estimator.cpp:2897-2955; the released default, config_urban_hk.yaml:63): blocks (P, Q) of keyframes rp_i[f]
 * and rp_j[f], rp_const [n_rp][7] = (delta_q w,x,y,z, delta_p) from the odometry poses.  Call BEFORE glio_batch_set_small_factors, which builds the
 * factor table (n_dq = n_dd = 0 is fine there). */
int glio_batch_set_relative_pose_factors(glio_batch* b, int n_rp, const int32_t* rp_i, const int32_t* rp_j, const double* rp_const);
int glio_batch_set_dd_threshold(glio_batch* b, double threshold);   /* the next round's DDpsr_threshold, factors stay on the device */
int glio_batch_set_imu(glio_batch* b, int n_edges, const glio_preint* edges, double gravity);
int glio_batch_add_small_dev(glio_batch* b, const double* poses, double* Hg_dev);      /* damped Gauss-Newton path, one rank */
int glio_batch_linearize_full(glio_batch* b, const double* poses, const double* speed_bias, glio_allreduce_fn allreduce, void* user,
                              double* diag, double* grad, double* cost);
int glio_batch_solve_tr2(glio_batch* b, double* poses, double* speed_bias, const glio_batch_tr_opts* opts, glio_allreduce_fn allreduce,
                         void* user, glio_summary* summary);
/* the pose-only problem (no IMU chain set) */
int glio_batch_solve_tr(glio_batch* b, double* poses, const glio_batch_tr_opts* opts, glio_allreduce_fn allreduce, void* user, glio_summary* summary);
/* hook calls, doubles handed to the hook, trust-region groups enqueued, elimination levels since the last call (reset on read) */
int glio_batch_debug_counters(glio_batch* b, int64_t* out4);
/* trust-region groups glio_batch_solve_tr2 keeps in flight: group g is enqueued when group g - lead has decided that the solve goes on; the groups behind the
 * deciding one exit at once on the device.  Every rank enqueues the same number of groups (a function of the device's decisions, not of host timing), so
 * the collective sequences match.  1 = wait for every group's decision before enqueuing the next (the round-4 loop); 0 = default (2). */
int glio_batch_debug_set_enqueue_lead(glio_batch* b, int lead);

/* For a C++ host that never includes HIP headers (glio_amd/host/glio_batch_backend.hpp): the reduced buffer
 * [H band | g | cost] as a device allocation, the batch stream to hand to ncclAllReduce between glio_batch_linearize_dev and
 * glio_batch_step_dev, a read-back of a few of its doubles (the cost is the last one), a stream synchronisation. */
int glio_batch_hg_alloc_dev(glio_batch* b, double** out_dev);
int glio_batch_hg_free_dev(glio_batch* b, double* dev);
int glio_batch_get_stream(glio_batch* b, void** out_hip_stream);
int glio_batch_read_dev(glio_batch* b, const double* dev, int64_t first, int64_t n, double* out_host);
int glio_batch_synchronize(glio_batch* b);
/* 0 = the sequential banded Cholesky (one workgroup), 1 = block cyclic reduction (default where the band permits) */
int glio_batch_debug_set_solver(glio_batch* b, int mode);
/* timing hook: average ms of `reps` banded solves (H + lambda diag H) x = g (HIP events on the batch stream) */
int glio_batch_time_solve(glio_batch* b, const double* Hg_dev, double lambda, int reps, float* ms_out);
/* timing hook: average ms of `reps` linearisation launches (HIP events on the batch stream) */
int glio_batch_time_linearize(glio_batch* b, const double* poses, double* Hg_dev, int reps, float* ms_out);


/* ================================================================================================
 * Batch association.  Replaces findGlobalCorrespondingSurfFeaturesAdd_Batch(idx, search_idx_start)
 * (Estimator.cpp:3808-3892; its twin findGlobalCorrespondingSurfFeatures_Batch :3711-3806 is the same arithmetic)
 * for a list of keyframe pairs: per pair (ci, cj) the points of surf_frames[ci] are matched against surf_frames[cj],
 * both placed with pose_info_keyframe (poses [K][7] = t, q): exact 5-NN (sqd[4] < 1.5), plane fit in global and in
 * cj-local coordinates, 0.18 validity, float pd / weight, weight > 0.3.  Output per kept point, appended pair after
 * pair in the caller's order (device resident, the layout glio_batch_set_constraints_pairs_dev takes):
 *   cp [4] float (the point in ci's frame), norm_cent [6] (unit normal and 5-point centroid in cj's frame),
 *   score = 2.5 weight.   The caller applies the search-window rule of Estimator.cpp:3009-3017 to make the pair list
 * (glio_amd/batch.py: search_window). */
typedef struct glio_bassoc glio_bassoc;
int glio_bassoc_create(int device, int K, int max_points_per_frame, int64_t max_constraints, glio_bassoc** out);
void glio_bassoc_destroy(glio_bassoc* b);
/* surf_frames[k]: PointXYZI[n] as 4 floats, keyframe-local; stays resident */
int glio_bassoc_set_frame(glio_bassoc* b, int k, const float* scan_xyzi, int n);
int glio_bassoc_set_frame_strided(glio_bassoc* b, int k, const void* scan_points, int n, int stride_bytes, int intensity_offset);
int glio_bassoc_run(glio_bassoc* b, const double* poses, int n_pairs, const int32_t* pair_ci, const int32_t* pair_cj,
                    int64_t* pair_count_out, int64_t* total_out);
/* batchFeatureAssociation() (Estimator.cpp:3413-3432), the call that ENDS every optimizeSlidingWindowWithLandMark (:2733): the keyframe
 * idx = size - search_range - 1 is matched against its 2 search_range neighbours and the records are ADDED to gl_vec_surf_* -- here: appended behind
 * what the object already holds (pair_count_out: the pairs of this call; total_out: everything held).  _async enqueues on the object's stream and
 * returns (the inputs are staged in pinned memory); glio_bassoc_finish waits and hands the counts over.  glio_bassoc_reset forgets the records.
 * Any other entry point of the object may be called in between (each waits for the run as far as it must): the run's counts -- and its
 * overflow error, if it produced more than max_constraints records -- are kept for glio_bassoc_finish and reported there ONCE; a second run
 * started before glio_bassoc_finish replaces the first run's counts, but returns its overflow error instead of starting. */
int glio_bassoc_run_append(glio_bassoc* b, const double* poses, int n_pairs, const int32_t* pair_ci, const int32_t* pair_cj,
                           int64_t* pair_count_out, int64_t* total_out);
int glio_bassoc_run_append_async(glio_bassoc* b, const double* poses, int n_pairs, const int32_t* pair_ci, const int32_t* pair_cj);
int glio_bassoc_finish(glio_bassoc* b, int64_t* pair_count_out, int64_t* total_out);
int glio_bassoc_reset(glio_bassoc* b);
/* globalFeatureSelectionAdd_Batch (Estimator.cpp:4057-4116) for the pairs of the asynchronous run in flight, ON ITS STREAM (call it right after
 * glio_bassoc_run_append_async, before glio_bassoc_finish): no host round trip between the searches and the selection.  The draws stay the caller's:
 * raws = res_num (<= 64) 64-bit numbers per pair of that run, drawn before the counts exist.  A pair with at most res_num records keeps them all; else it
 * keeps the first res_num of a uniform shuffle of its records but the last (random_generator.hpp:79-93 never draws it): step i swaps position i with
 * position i + raws[p * res_num + i] mod (count - 1 - i).  glio_bassoc_finish then reports the per-pair counts FOUND and the total HELD after the selection. */
int glio_bassoc_select_tail_draws_async(glio_bassoc* b, int res_num, const uint64_t* raws);
/* Optional, ahead of a run whose pairs are known before its poses (batchFeatureAssociation inside a keyframe call: the pairs follow from the keyframe count,
 * the poses from the solve): sends the build descriptors of the run's search frames and clears their hash tables now; the run that follows with the same
 * search frames skips both.  Anything else in between only makes the run do them itself.  Whether a preparation applies follows from what was prepared
 * (same search frames, same cloud sizes) and from nothing else -- not from how long ago it was made. */
int glio_bassoc_prepare_async(glio_bassoc* b, int n_pairs, const int32_t* pair_ci, const int32_t* pair_cj);
/* surf_frames[k] <- the scan resident in window slot `slot` of a sliding-window context on the same device, minus the LiDAR offset (a device copy:
 * the keyframe that just entered the window is not uploaded a second time).  The copy runs on the association's stream; the context's next
 * glio_set_scan and glio_destroy are ordered behind it on the device (no host wait). */
int glio_bassoc_set_frame_from_scan(glio_bassoc* b, int k, glio_ctx* ctx, int slot, const float lidar_offset[3]);
int glio_bassoc_results_dev(glio_bassoc* b, const float** cp_dev, const double** norm_cent_dev, const double** score_dev);
/* globalFeatureSelectionAdd_Batch / globalFeatureSelection_Batch (Estimator.cpp:4057-4116, 3994-4055; batch_feature_res_num: 25):
 * keep records src_index[0..n_keep) of the current n_current records, in that order (pair after pair; the caller updates its
 * per-pair counts).  The random draws stay with the caller (the reference seeds from std::random_device); the gather runs on
 * the device, in place.  glio_amd/batch.py::batch_selection_draws restates the draw rules. */
int glio_bassoc_select(glio_bassoc* b, int64_t n_keep, const int64_t* src_index, int64_t n_current);
/* the same over the tail [first, n_current) only -- what one keyframe's batchFeatureAssociation appended: src_index holds absolute indices >= first;
 * afterwards the object holds first + n_keep records.  Both calls copy src_index before they return and ENQUEUE the gather (no host wait);
 * glio_bassoc_read and glio_bassoc_results_dev wait for it, later runs and selections are ordered behind it. */
int glio_bassoc_select_range(glio_bassoc* b, int64_t first, int64_t n_keep, const int64_t* src_index, int64_t n_current);
int glio_bassoc_read(glio_bassoc* b, int64_t first, int64_t n, float* cp, double* norm_cent, double* score);

#ifdef __cplusplus
}
#endif
#endif /* GLIO_HIP_H_ */
