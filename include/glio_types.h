/*
 * glio_types.h -- plain-old-data buffer contracts for the GLIO sliding-window hot path.
 *
 * Every struct here restates (in flat C) a buffer that the reference keeps as an Eigen / PCL /
 * ROS-message member of `class Estimator` (reference: GLIO/src/Estimator.cpp) or of one of its
 * factor classes (reference: GLIO/include/factors/).  No HIP, torch or C++ types appear: this header
 * is what a cgo / ctypes / plain-C++ caller sees.  It is shared by the product library
 * (glio_amd/csrc, declared in glio_hip.h) and by the CPU oracle (oracle/glio_oracle.h) so that the
 * parity tests hand *identical bytes* to both sides.
 *
 * Conventions (all taken from the reference, file:line given per field):
 *   - quaternions are (w,x,y,z) doubles               Estimator.cpp:2103-2107
 *   - a keyframe state is T[3], Q[4], SpeedBias[9] = (v, ba, bg)   Estimator.cpp:345-348,2124-2126
 *   - points are PCL `PointXYZI` = 4 x float32 (x,y,z,intensity)   GLIO/include/utils/common.h
 *   - matrices are row-major unless said otherwise (Ceres Jacobian convention,
 *     GraphGNSSLibV1.1/docs/source/nnls_modeling.rst:75-140)
 */
#ifndef GLIO_TYPES_H_
#define GLIO_TYPES_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLIO_POSE_LOCAL 15      /* local (tangent) size of one keyframe: dt3 dtheta3 dv3 dba3 dbg3 */
#define GLIO_DD_MAX_SAT 20      /* dd_psr_factor_20: psr_size_20, dd_psr_factor.hpp:12 */
#define GLIO_MAX_WINDOW 64
enum { GLIO_STRATEGY_DOGLEG = 0, GLIO_STRATEGY_LM = 1 };

/* Options that shape the hot path.  Defaults (glio_opts_default) are the shipped yaml /
 * hard-coded values: GLIO/config/config_urban_hk.yaml:60-104, Estimator.cpp:70,2424-2430. */
typedef struct glio_opts {
    int32_t window;              /* slide_window_width (yaml:66) -- number of keyframes W */
    int32_t max_iterations;      /* options.max_num_iterations = 15, Estimator.cpp:2427 */
    int32_t max_points_per_scan; /* device capacity per keyframe slot (scan points / correspondences) */
    int32_t max_map_points;      /* device capacity of the local surf map */
    int32_t max_ddt_epochs;      /* number of para_rcv_ddt slots carried as unknowns (<= EPOCH_SIZE 5000) */
    int32_t jacobi_scaling;      /* Ceres default true, nnls_solving.rst:1402-1404 */
    double huber_delta;          /* lossKernel 1.0, Estimator.cpp:70,2092 */
    double doppler_huber_delta;  /* HuberLoss(1.0) on tcdoppler, Estimator.cpp:2335 */
    double q_lb[4];              /* LiDAR->IMU extrinsic rotation (w,x,y,z), yaml:90-93 */
    double t_lb[3];              /* LiDAR->IMU extrinsic translation, yaml:95-97 */
    double lidar_const;          /* yaml:70 */
    double surf_dist_thres;      /* yaml:71 (plane gate) */
    double kd_max_radius;        /* yaml:72 (a double, Estimator.cpp:364) -- compared with the SQUARED float 5th-NN distance, :3651 */
    double weight_gate;          /* 0.3 -- `weight > 0.3` compares the float weight with a double literal, Estimator.cpp:3681 */
    double gravity;              /* IMU/gravity yaml:11 ; g_vec = (0,0,-gravity) Preintegration.h:58 */
    /* trust region (Ceres 1.14 defaults, nnls_solving.rst:1056-1188) */
    double initial_trust_region_radius; /* 1e4 */
    double max_trust_region_radius;     /* 1e16 */
    double min_trust_region_radius;     /* 1e-32 */
    double min_relative_decrease;       /* 1e-3 */
    double function_tolerance;          /* 1e-6 */
    double gradient_tolerance;          /* 1e-10 */
    double parameter_tolerance;         /* 1e-8 */
    /* 0 = DOGLEG (sliding window / batch, Estimator.cpp:2425), 1 = LEVENBERG_MARQUARDT (the Ceres default the
     * front end runs with, LidarOdometry.cpp:521-530) */
    int32_t trust_region_strategy;
    /* 0 = vec_surf_scores = lidar_const * weight (Estimator.cpp:3692); 1 = unit scores: the front end's
     * LidarPlaneNormIncreFactor carries no score (LidarKeyframeFactor.h:222-257) */
    int32_t unit_scores;
    /* arithmetic of the LiDAR plane linearisation (K3).  GLIO_LIDAR_F64 (default): everything in double, 40 B per residual
     * (the reference's Jet<double,7> evaluation).  GLIO_LIDAR_F32_MFMA (BASELINE config C5, the W = 50 x 256 k stress
     * shape): the score travels as a float in the point's 4th component (32 B per residual), the residual is still
     * formed in double, the 1x6 Jacobian in float, and the per-keyframe J^T J / J^T r contraction of each 64-residual
     * chunk runs on the matrix core (v_mfma_f32_16x16x4_f32) with double accumulation across chunks. */
    int32_t lidar_precision;
    int32_t reserved_;
    /* options.max_solver_time_in_seconds (the front end runs with 0.015, LidarOdometry.cpp:524; 0 = no limit, the sliding window's
     * default).  Checked where Ceres checks it -- at the start of an iteration, after the tolerance tests of the previous one: the
     * host's clock raises a flag in mapped memory, the device-resident loop reads it in its state machine and ends the solve with
     * GLIO_TERM_NO_CONVERGENCE at the current (last accepted) point. */
    double max_solver_time_s;
} glio_opts;
enum { GLIO_LIDAR_F64 = 0, GLIO_LIDAR_F32_MFMA = 1 };

/* Window state = the Ceres parameter blocks of the sliding-window problem.
 * Reference: tmpTrans/tmpQuat/tmpSpeedBias (Estimator.cpp:345-348), para_rcv_ddt (:309).
 * All pointers are caller-owned host memory. */
typedef struct glio_state {
    double* trans;       /* [W][3] */
    double* quat;        /* [W][4]  (w,x,y,z) */
    double* speed_bias;  /* [W][9]  (v, ba, bg) */
    double* rcv_ddt;     /* [n_ddt] receiver clock-drift slots touched by Doppler factors (may be NULL) */
    int32_t n_ddt;
} glio_state;

/* One IMU pre-integration between consecutive keyframes: the members of `class Preintegration`
 * (GLIO/include/factors/Preintegration.h:237-256) that ImuFactor::Evaluate reads. */
typedef struct glio_preint {
    double delta_p[3];
    double delta_q[4];        /* (w,x,y,z), normalised by Propagate (Preintegration.h:190) */
    double delta_v[3];
    double linearized_ba[3];
    double linearized_bg[3];
    double sum_dt;
    double jacobian[225];     /* 15x15 row-major, order P,R,V,BA,BG (Preintegration.h:15-21) */
    double covariance[225];   /* 15x15 row-major */
} glio_preint;

/* Kind of a kept parameter block inside the marginalization prior. */
enum { GLIO_BLK_TRANS = 0, GLIO_BLK_QUAT = 1, GLIO_BLK_SPEEDBIAS = 2 };

/* Marginalization prior = `MarginalizationInfo` members read by MarginalizationFactor::Evaluate
 * (GLIO/src/MarginalizationFactor.cpp:233-287): linearized_jacobians (n x n), linearized_residuals
 * (n), keep_block_{size,idx,data}.  Block b refers to window slot blk_slot[b], kind blk_kind[b];
 * its local offset inside the n-vector is blk_idx[b] (= keep_block_idx - m). */
typedef struct glio_prior {
    int32_t n;                /* number of rows/cols (0 = no prior) */
    int32_t n_blocks;
    const double* lin_jac;    /* [n][n] row-major */
    const double* lin_res;    /* [n] */
    const int32_t* blk_slot;  /* [n_blocks] */
    const int32_t* blk_kind;  /* [n_blocks] GLIO_BLK_* */
    const int32_t* blk_idx;   /* [n_blocks] */
    const double* blk_x0;     /* [n_blocks][9] linearisation point (first 3/4/9 entries used) */
} glio_prior;

/* One double-differenced pseudorange factor (dd_psr_factor_20, dd_psr_factor.hpp:15-171).
 * Per-satellite arrays have n_sat entries taken from the user / reference-station GNSS_Raw arrays
 * (nlosExclusion/msg/GNSS_Raw.msg:5-20). */
typedef struct glio_dd_psr {
    int32_t slot_i, slot_j;   /* Pi, Pj parameter blocks (tmpTrans[leftKey], tmpTrans[rightKey]) */
    int32_t n_sat;            /* <= 20 */
    int32_t master;           /* mPrn: index of the master satellite */
    double ratio;             /* ts_ratio */
    double threshold;         /* DDpsrThreshold */
    double station[3];        /* Station_pos (ECEF) */
    double user_sat_pos[GLIO_DD_MAX_SAT][3];  /* gnss_data.GNSS_Raws[i].sat_pos_{x,y,z} */
    double ref_sat_pos[GLIO_DD_MAX_SAT][3];   /* ref_gnss_data.GNSS_Raws[i].sat_pos_{x,y,z} */
    double user_psr[GLIO_DD_MAX_SAT];         /* gnss_data...raw_pseudorange */
    double ref_psr[GLIO_DD_MAX_SAT];          /* ref_gnss_data...raw_pseudorange */
    double weight[(GLIO_DD_MAX_SAT - 1) * (GLIO_DD_MAX_SAT - 1)]; /* DD_W_matrix, row-major (n_sat-1)^2 packed with stride n_sat-1 */
} glio_dd_psr;

/* One tightly-coupled Doppler row (tcdopplerFactor, dopp_factor.hpp:19-85). */
typedef struct glio_doppler {
    int32_t slot_i, slot_j;   /* statePi/stateVi from slot_i, statePj/stateVj from slot_j */
    int32_t epoch;            /* index into glio_state.rcv_ddt */
    int32_t pad_;
    double ratio;             /* ts_ratio */
    double var;               /* residual divided by var (dopp_factor.hpp:72) */
    double sat_pos[3], sat_vel[3];
    double sv_ddt;            /* gnss_data.ddt */
    double doppler, lamda;    /* gnss_data.doppler * gnss_data.lamda */
    double lever_arm[3];      /* lever_arm_T */
    double R_ecef_local[9];   /* row-major, fixed at construction (Estimator.cpp:2326) */
} glio_doppler;

/* Constant GNSS frame blocks: para_yaw_enu_local[1], para_anc_ecef[3] (Estimator.cpp:307-308;
 * SetParameterBlockConstant at :2141,2145). */
typedef struct glio_gnss_frame {
    double yaw_enu_local;
    double anc_ecef[3];
} glio_gnss_frame;

/* Solver summary: the subset of ceres::Solver::Summary the caller can observe. */
typedef struct glio_summary {
    int32_t iterations;           /* step attempts executed (successful + unsuccessful) */
    int32_t successful_steps;
    int32_t termination;          /* GLIO_TERM_* */
    int32_t n_lidar_residuals;
    double initial_cost;
    double final_cost;
    double final_radius;
    double gradient_max_norm;
} glio_summary;

enum {
    GLIO_TERM_NO_CONVERGENCE = 0,   /* hit max_iterations */
    GLIO_TERM_FUNCTION_TOL = 1,
    GLIO_TERM_PARAMETER_TOL = 2,
    GLIO_TERM_GRADIENT_TOL = 3,
    GLIO_TERM_MIN_RADIUS = 4,
    GLIO_TERM_FAILURE = 5
};

/* Trust-region options of the batch solve (ceres::Solver::Options as set at Estimator.cpp:3275-3281: DOGLEG, non-monotonic
 * steps, max_num_iter from the yaml; the rest are Ceres 1.14 defaults).  Shared by libglio_hip (glio_batch_solve_tr) and the oracle. */
typedef struct glio_batch_tr_opts {
    int32_t max_iterations;                      /* options.max_num_iterations = max_num_iter (yaml:65: 100) */
    int32_t use_nonmonotonic_steps;              /* true, Estimator.cpp:3281 */
    int32_t max_consecutive_nonmonotonic_steps;  /* Ceres default 5 */
    int32_t jacobi_scaling;                      /* Ceres default true */
    double initial_trust_region_radius;          /* 1e4 */
    double max_trust_region_radius;              /* 1e16 */
    double min_trust_region_radius;              /* 1e-32 */
    double min_relative_decrease;                /* 1e-3 */
    double function_tolerance;                   /* 1e-6 */
    double gradient_tolerance;                   /* 1e-10 */
    double parameter_tolerance;                  /* 1e-8 */
    int32_t dogleg_type;                         /* GLIO_DOGLEG_SUBSPACE: options.dogleg_type = SUBSPACE_DOGLEG, Estimator.cpp:3278 */
    int32_t reserved_;
} glio_batch_tr_opts;
enum { GLIO_DOGLEG_TRADITIONAL = 0, GLIO_DOGLEG_SUBSPACE = 1 };

/* One scan-to-multiscan constraint of the batch stage (BinaryLidarPlaneNormFactor,
 * LidarKeyframeFactor.h:124-164; built Estimator.cpp:3048,3071). */
typedef struct glio_batch_opts {
    int32_t n_keyframes;
    int32_t band;            /* max |idx - search_idx| (2*search_range at the ends, Estimator.cpp:3009-3017) */
    int32_t max_iterations;
    int32_t pad_;
    double lm_lambda;        /* fixed Levenberg damping used by the restated batch solve */
} glio_batch_opts;

#ifdef __cplusplus
}
#endif
#endif /* GLIO_TYPES_H_ */
