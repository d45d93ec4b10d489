/*
 * glio_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C, single-thread, fp64 restatement of the reference's sliding-window hot path
 * (GLIO/src/Estimator.cpp:2046-2736 and the factor headers under GLIO/include/factors/), written the
 * way the reference + Ceres 1.14 evaluate it: every factor returns residuals and GLOBAL Jacobians
 * through the ceres::CostFunction::Evaluate convention, the loss corrector and the quaternion
 * local parameterisation are applied afterwards, and the normal equations are summed densely.
 *
 * PARITY: the FACTOR LAYER of this restatement (every orc_eval_*, the pre-integration inputs, the invariants of
 * orc_marginalize) is pinned, since round 4, on the reference's own code: oracle/_ref/libglio_ref.so = the reference's
 * factor headers + MarginalizationFactor.cpp + gnss_utility.cpp compiled unmodified from /root/reference against the
 * stand-in headers of oracle/ref_shim/include (tests/test_oracle_ref.py: <= 1e-12 relative on 1000 random inputs per
 * factor; tests/golden/ref_factors.npz carries the reference's outputs to the GPU box).  The trust-region LOOP (orc_solver.c,
 * orc_batch2.c; Ceres is neither in the image nor under /root/reference) is held, iteration by iteration, to tests/np_ceres.py
 * (tests/test_oracle_tr_pins.py), and that restatement reproduces the progress tables the real library printed into its own
 * documentation -- cost to 7 digits, gradient, step, tr_ratio, radius per iteration for helloworld and Powell's function, the
 * radius column of curve_fitting's rejected and accepted steps (GraphGNSSLibV1.1/docs/source/nnls_tutorial.rst:139-143,378-394,
 * 411-432,508-523; tests/test_ceres_docs_kat.py).  STILL UNPINNED: the dogleg step itself (no printed table uses it), PCL's kd-tree /
 * VoxelGrid and Eigen's colPivHouseholderQr (orc_assoc.c).  Numbers that pass through those must be labelled
 * "reference-restatement (Ceres-1.14 semantics)", never "Ceres".
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this.
 */
#ifndef GLIO_ORACLE_H_
#define GLIO_ORACLE_H_

#include "../include/glio_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_problem {
    glio_opts opts;
    /* LiDAR plane correspondences, concatenated by window slot: slot s owns [off[s], off[s+1]) */
    const int32_t* lidar_offset;   /* [W+1] */
    const float* lidar_pts;        /* [N][4]  vec_surf_cur_pts (x,y,z,intensity), LiDAR frame */
    const float* lidar_planes;     /* [N][4]  vec_surf_normal: (w*n, w*d) */
    const double* lidar_scores;    /* [N]     vec_surf_scores = lidar_const*w */
    /* IMU edges: edge k links slot imu_slot[k] and imu_slot[k]+1 */
    int32_t n_imu;
    const glio_preint* imu;
    const int32_t* imu_slot;
    glio_prior prior;
    int32_t n_dd;
    const glio_dd_psr* dd;
    int32_t n_dop;
    const glio_doppler* dop;
    glio_gnss_frame frame;
} orc_problem;

void orc_opts_default(glio_opts* o);

/* ---- single-factor evaluators, Ceres Evaluate() pointer convention (jacobians / jacobians[i] may be NULL) */
/* LidarPlaneNormFactor (LidarKeyframeFactor.h:73-122): blocks t[3], q[4]; 1 residual */
int orc_eval_lidar_plane(const glio_opts* o, const float cp[4], const float plane[4], double score,
                         double const* const* parameters, double* residuals, double** jacobians);
/* ImuFactor (ImuFactor.h:21-171): blocks Pi3 Qi4 SBi9 Pj3 Qj4 SBj9; 15 residuals */
int orc_eval_imu(const glio_opts* o, const glio_preint* pre,
                 double const* const* parameters, double* residuals, double** jacobians);
/* sqrt_info = LLT(cov^-1).L^T  (ImuFactor.h:44-45); out 15x15 row-major */
int orc_imu_sqrt_info(const double* covariance, double* sqrt_info);
/* MarginalizationFactor::Evaluate (MarginalizationFactor.cpp:233-287): parameters[b] = block b */
int orc_eval_marg(const glio_prior* p, double const* const* parameters, double* residuals, double** jacobians);
/* dd_psr_factor_20::Evaluate (dd_psr_factor.hpp:25-171): blocks Pi3 Pj3 yaw1 anc3; 19 residuals */
int orc_eval_dd_psr(const glio_dd_psr* f, double const* const* parameters, double* residuals, double** jacobians);
/* tcdopplerFactor (dopp_factor.hpp:24-75): blocks Pi3 SBi9 Pj3 SBj9 ddt[n] yaw1 anc3; 1 residual.
 * jacobians[4] (if non-NULL) receives only d r / d ddt[epoch] as a single double. */
int orc_eval_doppler(const glio_doppler* f, double const* const* parameters, double* residuals, double** jacobians);
/* ecef2rotation (gnss_comm/src/gnss_utility.cpp:347-390,738-753): R_ecef_enu row-major */
void orc_ecef2rotation(const double ecef[3], double R[9]);

/* ---- Ceres pieces */
/* QuaternionParameterization::Plus (nnls_modeling.rst:1312-1327) */
void orc_quat_plus(const double q[4], const double delta[3], double out[4]);
/* x (+) delta over the whole window; delta has 15*W + n_ddt entries */
void orc_state_plus(const glio_state* x, int W, const double* delta, glio_state* out);

/* ---- window problem */
/* local (tangent) dimension 15*W + n_ddt */
int orc_local_dim(const orc_problem* p, const glio_state* x);
/* One linearisation: dense H = J^T J (n x n row-major), g = J^T r, cost = 1/2 sum rho(|r|^2),
 * all AFTER loss correction and local parameterisation, unscaled.  Any of H,g may be NULL. */
int orc_linearize(const orc_problem* p, const glio_state* x, double* H, double* g, double* cost);
/* bench.py's all-cores CPU baseline: OpenMP threads over the keyframes of the LiDAR factor loop (default 1 = the reference's
 * options.num_threads = 1; parity tests never change it) */
void orc_set_threads(int t);
/* Ceres-1.14 trust-region (traditional dogleg, dense normal Cholesky, Jacobi scaling) */
int orc_solve(const orc_problem* p, glio_state* x, glio_summary* summary);
/* the same, recording per iteration (candidate cost, radius the step was computed with, |x - candidate|): history [max_iterations][3] */
int orc_solve_history(const orc_problem* p, glio_state* x, glio_summary* summary, double* history);

/* ---- correspondence search (Estimator.cpp:3633-3708): brute-force exact 5-NN + plane fit.
 * out arrays have capacity n_scan; returns the number of correspondences kept, in scan order. */
int orc_associate(const glio_opts* o, const float* map_pts /*[M][4]*/, int M,
                  const float* scan /*[n][4]*/, int n, const double q[4], const double t[3],
                  float* out_pts /*[n][4]*/, float* out_planes /*[n][4]*/, double* out_scores,
                  int32_t* out_src_index /* may be NULL */, int32_t* out_nn /* [n][5] may be NULL */);
/* identical output; the brute-force nearest-neighbour phase runs on `threads` OpenMP threads (full-size parity checks) */
int orc_associate_mt(const glio_opts* o, const float* map_pts, int M, const float* scan, int n, const double q[4],
                     const double t[3], float* out_pts, float* out_planes, double* out_scores,
                     int32_t* out_src_index, int32_t* out_nn, int threads);
/* bench.py's CPU baseline only: 1 = the nearest-neighbour phase of orc_associate* uses a voxel grid of edge sqrt(kd_max_radius)
 * instead of the brute force (identical records for every query that passes the radius gate); parity tests keep the default 0 */
void orc_set_assoc_grid(int on);
/* colPivHouseholderQr().solve for the 5x3 system A n = b  (Estimator.cpp:3661) */
void orc_plane_qr_solve(const double A[15], const double b[5], double x[3]);

/* ---- marginalization (MarginalizationFactor.cpp:87-221 + Estimator.cpp:2462-2607) */
/* Builds the next prior from the CURRENT window (prior + IMU(0,1) + all LiDAR factors, slot 0
 * dropped).  Output arrays sized for n = 6(W-1)+9: lin_jac [n*n], lin_res [n], blk_* [2(W-1)+1],
 * blk_x0 [(2(W-1)+1)*9].  Slots in the output are ALREADY shifted (i -> i-1, Estimator.cpp:2584-2598).
 * Returns n. */
int orc_marginalize(const orc_problem* p, const glio_state* x, double* lin_jac, double* lin_res,
                    int32_t* blk_slot, int32_t* blk_kind, int32_t* blk_idx, double* blk_x0);

void orc_transform_cloud(const float* in, int n, const double q[4], const double t[3], float* out);
/* pcl::VoxelGrid<PointXYZI> as used by downSampleCloud (Estimator.cpp:3618-3631); returns the voxel count */
int orc_voxel_grid(const float* pts, int n, float leaf, float* out, int64_t* out_idx);

/* findGlobalCorrespondingSurfFeaturesAdd_Batch for one keyframe pair (Estimator.cpp:3808-3892); returns the count */
int orc_associate_pair(const float* scan_a, int na, const double qa[4], const double ta[3],
                       const float* scan_b, int nb, const double qb[4], const double tb[3],
                       float* out_cp, double* out_norm_cent, double* out_score, int32_t* out_src);

/* ---- batch stage (BinaryLidarPlaneNormFactor, LidarKeyframeFactor.h:124-164) */
/* residual + global jacobians, blocks t1[3] q1[4] t2[3] q2[4] */
int orc_eval_binary_plane(const float cp[4], const double norm_cent[6], double score,
                          double const* const* parameters, double* residuals, double** jacobians);
/* Banded normal equations of a batch of binary constraints.  Poses [K][7] = (t, q).  Constraint c:
 * keyframes (ci[c], cj[c]).  Hband: [K][band+1][36] upper block band (block (k, k+d) at [k][d]),
 * g: [K][6]. */
int orc_batch_linearize(int K, int band, const double* poses, int64_t n_con, const int32_t* ci,
                        const int32_t* cj, const float* cp /*[n][4]*/, const double* norm_cent /*[n][6]*/,
                        const double* score, double* Hband, double* g, double* cost);

/* ---- the batch problem on keyframe poses (Estimator::optimizeBatchWithLandMark, Estimator.cpp:2739-3410, sms_fusion_level 1
 * without the IMU chain): BinaryLidarPlaneNormFactor blocks (:3004-3076), delta_q_factor_auto attitude constraints (:2831-2891,
 * LidarKeyframeFactor.h:283-303), dd_psr_factor_20 per GNSS epoch between the bracketing keyframes with identity weight and the
 * station position (:3197-3271, :1899-1911); no loss function (:2768). */
typedef struct orc_batch_problem {
    int32_t K, band;
    int64_t n_con; const int32_t* ci; const int32_t* cj; const float* cp; const double* norm_cent; const double* score;
    int32_t n_dq; const int32_t* dq_i; const int32_t* dq_j; const double* dq_const;      /* [n_dq][4] const_diff = qi^-1 qj at construction (w,x,y,z) */
    int32_t n_dd; const glio_dd_psr* dd;                                                /* slot_i / slot_j = keyframe indices (leftKey, rightKey) */
    glio_gnss_frame frame;
    /* the IMU chain (Estimator.cpp:2990-3001): n_imu = 0 (pose-only problem, 6 unknowns per keyframe) or K - 1: imu[k] is the
     * pre-integration of the ImuFactor between keyframes k and k + 1 (the caller decides which interval that is, quirk Q11);
     * then every keyframe carries its speed-bias block (:2809-2819) and 15 unknowns */
    int32_t n_imu; int32_t pad_; const glio_preint* imu; double gravity;
    /* LidarPoseFactorBatchRelativeAutoDiff factors (sms_fusion_level 0, Estimator.cpp:2897-2955): keyframes (rp_i, rp_j), rp_const [n_rp][7] = delta_q (w,x,y,z), delta_p */
    int32_t n_rp; int32_t pad2_; const int32_t* rp_i; const int32_t* rp_j; const double* rp_const;
} orc_batch_problem;
/* delta_q_factor_auto (LidarKeyframeFactor.h:283-303): blocks qi[4], qj[4]; 3 residuals = 10000 (dq^-1 qi^-1 qj).vec; global Jacobians 3x4 */
int orc_eval_delta_q(const double dq_const[4], double const* const* parameters, double* residuals, double** jacobians);
/* LidarPoseFactorBatchRelativeAutoDiff (LidarPoseFactor.h:55-97): blocks P1[3] Q1[4] P2[3] Q2[4]; 6 residuals; global Jacobians 6 x {3,4,3,4} */
int orc_eval_relative_pose(const double dq[4], const double dp[3], double const* const* parameters, double* residuals, double** jacobians);
/* banded normal equations of ALL factors (layout as orc_batch_linearize) */
int orc_batch_linearize_full(const orc_batch_problem* p, const double* poses, double* Hband, double* g, double* cost);
/* Ceres-1.14 trust region on the batch problem (orc_batch2.c): Jacobi scaling, DOGLEG with o->dogleg_type (the reference:
 * SUBSPACE_DOGLEG, Estimator.cpp:3278), non-monotonic step acceptance (TrustRegionStepEvaluator), the returned point is the one
 * of least cost (Ceres copies x to the user's parameters only when x_cost < minimum_cost) and final_cost that minimum, dense
 * Cholesky.  poses [K][7] in/out; speed_bias [K][9] in/out when p->n_imu > 0 (NULL otherwise).
 * history (may be NULL): [max_iterations + 1][4] = candidate cost, radius, |x - candidate|, step quality per iteration (row 0: start). */
int orc_batch2_dim(const orc_batch_problem* p);
int orc_batch2_linearize(const orc_batch_problem* p, const double* poses, const double* speed_bias, double* H /*[n][n]*/, double* g, double* cost);
/* the same matrix as a symmetric LOWER BAND (what the solve itself works on: C4's K = 2000 does not fit a dense n x n): Hband [n][hbw + 1],
 * entry (i, j) with i - hbw <= j <= i at [i][j - i + hbw]; hbw = max(B * band + 5, 29 with the IMU chain) */
int orc_batch2_half_bandwidth(const orc_batch_problem* p);
int orc_batch2_linearize_banded(const orc_batch_problem* p, const double* poses, const double* speed_bias, double* Hband, double* g, double* cost);
int orc_batch2_solve(const orc_batch_problem* p, const glio_batch_tr_opts* o, double* poses, double* speed_bias, glio_summary* summary, double* history);
/* the pose-only problem (p->n_imu must be 0) */
int orc_batch_solve(const orc_batch_problem* p, const glio_batch_tr_opts* o, double* poses, glio_summary* summary);
/* pieces of the subspace dogleg, exposed for the pins in tests/: real parts of all roots of a polynomial (leading coefficient
 * first; Ceres' FindPolynomialRoots) and FindMinimumOnTrustRegionBoundary of the 2-D model (B row-major 2x2) */
int orc_poly_roots_real(const double* coeffs, int degree, double* roots_real, int* n_roots);
int orc_subspace_boundary_minimum(const double B[4], const double g[2], double radius, double out[2]);

#ifdef __cplusplus
}
#endif
#endif
