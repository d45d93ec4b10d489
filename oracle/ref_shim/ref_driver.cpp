// ref_driver.cpp -- TEST INFRASTRUCTURE (oracle/_ref): a C-ABI around the REFERENCE'S OWN factor code.
//
// The classes evaluated here are compiled, unmodified, from /root/reference:
//     GLIO/include/factors/LidarKeyframeFactor.h   LidarPlaneNormFactor, BinaryLidarPlaneNormFactor, LidarPlaneNormIncreFactor,
//                                                  delta_q_factor_auto
//     GLIO/include/factors/LidarPoseFactor.h       LidarPoseFactorBatchRelativeAutoDiff
//     GLIO/include/factors/Preintegration.h        Preintegration (push_back / Propagate / MidPointIntegration / evaluate)
//     GLIO/include/factors/ImuFactor.h             ImuFactor::Evaluate
//     GLIO/include/factors/dd_psr_factor.hpp       dd_psr_factor_20::Evaluate
//     GLIO/include/factors/dopp_factor.hpp         tcdopplerFactor
//     GLIO/include/factors/MarginalizationFactor.h + GLIO/src/MarginalizationFactor.cpp   (its own translation unit, see Makefile)
//     GLIO/include/utils/math_tools.h, GLIO/include/utils/common.h
//     gnss_comm/src/gnss_utility.cpp               ecef2rotation (its own translation unit)
// against the in-tree stand-ins for Eigen / the Ceres modelling API / ROS / PCL under oracle/ref_shim/include (none of those
// libraries is in this image or under /root/reference).  This file only marshals plain arrays into the reference's constructors
// and calls Evaluate(); the one piece of orchestration it restates is the list of ResidualBlockInfo the estimator hands to
// MarginalizationInfo (Estimator.cpp:2462-2607 -- Estimator.cpp itself needs all of ROS/PCL/GTSAM and cannot be compiled).
//
// tests/test_oracle_ref.py compares oracle/orc_*.c with this library; nothing in glio_amd/ may link or load it.
#include <cstdint>
#include <cstring>
#include <memory>
#include <unordered_map>
#include <vector>

#include "factors/LidarKeyframeFactor.h"
#include "factors/LidarPoseFactor.h"
#include "factors/ImuFactor.h"
#include "factors/MarginalizationFactor.h"
#include "factors/dd_psr_factor.hpp"
#include "factors/dopp_factor.hpp"

#include "../../include/glio_types.h"

namespace {
Eigen::Vector3d v3(const double* p) { return Eigen::Vector3d(p[0], p[1], p[2]); }
Eigen::Quaterniond q4(const double* p) { return Eigen::Quaterniond(p[0], p[1], p[2], p[3]); }

// a Preintegration object carrying the numbers of a glio_preint (all members are public in the reference's class)
std::unique_ptr<Preintegration> make_preint(const glio_preint* g, double gravity) {
    glio_ref_shim::params()["/IMU/gravity"] = gravity;         // g_vec_ = -(0, 0, gravity), Preintegration.h:47-58
    std::unique_ptr<Preintegration> p(new Preintegration(Eigen::Vector3d::Zero(), Eigen::Vector3d::Zero(), v3(g->linearized_ba), v3(g->linearized_bg)));
    p->delta_p_ = v3(g->delta_p); p->delta_q_ = q4(g->delta_q); p->delta_v_ = v3(g->delta_v);
    p->sum_dt_ = g->sum_dt;
    for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) { p->jacobian_(i, j) = g->jacobian[i * 15 + j]; p->covariance_(i, j) = g->covariance[i * 15 + j]; }
    return p;
}
void export_preint(const Preintegration& p, glio_preint* g) {
    for (int k = 0; k < 3; ++k) { g->delta_p[k] = p.delta_p_(k); g->delta_v[k] = p.delta_v_(k); g->linearized_ba[k] = p.linearized_ba_(k); g->linearized_bg[k] = p.linearized_bg_(k); }
    g->delta_q[0] = p.delta_q_.w(); g->delta_q[1] = p.delta_q_.x(); g->delta_q[2] = p.delta_q_.y(); g->delta_q[3] = p.delta_q_.z();
    g->sum_dt = p.sum_dt_;
    for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) { g->jacobian[i * 15 + j] = p.jacobian_(i, j); g->covariance[i * 15 + j] = p.covariance_(i, j); }
}
nlosExclusion::GNSS_Raw_Array raws(int n, const double (*pos)[3], const double* psr) {
    nlosExclusion::GNSS_Raw_Array a;
    a.GNSS_Raws.resize(n);
    for (int i = 0; i < n; ++i) { a.GNSS_Raws[i].sat_pos_x = pos[i][0]; a.GNSS_Raws[i].sat_pos_y = pos[i][1]; a.GNSS_Raws[i].sat_pos_z = pos[i][2]; a.GNSS_Raws[i].raw_pseudorange = psr[i]; }
    return a;
}
// a MarginalizationInfo that plays "last_marginalization_info": the members MarginalizationFactor reads, filled from a glio_prior
MarginalizationInfo* info_from_prior(const glio_prior* p, std::vector<std::vector<double> >& x0_store) {
    MarginalizationInfo* info = new MarginalizationInfo();
    info->n = p->n; info->m = 0;
    info->linearized_jacobians.resize(p->n, p->n);
    info->linearized_residuals.resize(p->n);
    for (int i = 0; i < p->n; ++i) { info->linearized_residuals(i) = p->lin_res[i]; for (int j = 0; j < p->n; ++j) info->linearized_jacobians(i, j) = p->lin_jac[(size_t)i * p->n + j]; }
    x0_store.resize(p->n_blocks);
    for (int b = 0; b < p->n_blocks; ++b) {
        const int size = p->blk_kind[b] == GLIO_BLK_TRANS ? 3 : (p->blk_kind[b] == GLIO_BLK_QUAT ? 4 : 9);
        x0_store[b].assign(p->blk_x0 + 9 * b, p->blk_x0 + 9 * b + size);
        info->keep_block_size.push_back(size);
        info->keep_block_idx.push_back(p->blk_idx[b]);          // (+ m, and m = 0)
        info->keep_block_data.push_back(x0_store[b].data());
    }
    return info;
}
}  // namespace

extern "C" {

// yaml parameters the reference reads through ros::NodeHandle::param (Preintegration.h:47-51)
void ref_set_param(const char* name, double value) { glio_ref_shim::params()[name] = value; }

// ---- LiDAR factors (Ceres Evaluate pointer convention; jacobians / jacobians[i] may be NULL) -------------------------------------------
// LidarPlaneNormFactor::Create (LidarKeyframeFactor.h:73-122), blocks t[3], q[4]
int ref_eval_lidar_plane(const double cp[3], const double n[3], double d, double score, const double qlb[4], const double tlb[3],
                         double const* const* parameters, double* residuals, double** jacobians) {
    std::unique_ptr<ceres::CostFunction> f(LidarPlaneNormFactor::Create(v3(cp), v3(n), q4(qlb), v3(tlb), d, score));
    return f->Evaluate(parameters, residuals, jacobians) ? 0 : 1;
}
// BinaryLidarPlaneNormFactor::Create (LidarKeyframeFactor.h:124-164), blocks t1[3], q1[4], t2[3], q2[4]
int ref_eval_binary_plane(const double cp[3], const double norm_cent[6], double score, double const* const* parameters, double* residuals, double** jacobians) {
    Eigen::Matrix<double, 6, 1> pnc;
    for (int k = 0; k < 6; ++k) pnc(k) = norm_cent[k];
    std::unique_ptr<ceres::CostFunction> f(BinaryLidarPlaneNormFactor::Create(v3(cp), pnc, score));
    return f->Evaluate(parameters, residuals, jacobians) ? 0 : 1;
}
// LidarPlaneNormIncreFactor::Create (LidarKeyframeFactor.h:222-257), blocks q[4], t[3]
int ref_eval_plane_incre(const double cp[3], const double n[3], double d, double const* const* parameters, double* residuals, double** jacobians) {
    std::unique_ptr<ceres::CostFunction> f(LidarPlaneNormIncreFactor::Create(v3(cp), v3(n), d));
    return f->Evaluate(parameters, residuals, jacobians) ? 0 : 1;
}
// delta_q_factor_auto as instantiated at Estimator.cpp:2861: AutoDiffCostFunction<delta_q_factor_auto, 3, 4, 4>
int ref_eval_delta_q(const double dq[4], double const* const* parameters, double* residuals, double** jacobians) {
    ceres::AutoDiffCostFunction<delta_q_factor_auto, 3, 4, 4> f(new delta_q_factor_auto(q4(dq)));
    return f.Evaluate(parameters, residuals, jacobians) ? 0 : 1;
}
// LidarPoseFactorBatchRelativeAutoDiff::Create (LidarPoseFactor.h:55-97; Estimator.cpp:2922), blocks p1[3], q1[4], p2[3], q2[4]; 6 residuals
int ref_eval_relative_pose(const double dq[4], const double dp[3], double const* const* parameters, double* residuals, double** jacobians) {
    std::unique_ptr<ceres::CostFunction> f(LidarPoseFactorBatchRelativeAutoDiff::Create(q4(dq), v3(dp)));
    return f->Evaluate(parameters, residuals, jacobians) ? 0 : 1;
}

// ---- IMU -------------------------------------------------------------------------------------------------------------------------------
// Preintegration(acc0, gyr0, ba, bg) + push_back(dt, acc, gyr) for n samples (Preintegration.h:27-194): the reference's own midpoint
// propagation of delta_p/q/v, jacobian_ and covariance_.  Noise densities through ref_set_param("/IMU/acc_n", ...) etc.
int ref_preintegrate(const double acc0[3], const double gyr0[3], const double ba[3], const double bg[3], int n, const double* dt, const double* acc, const double* gyr,
                     glio_preint* out) {
    Preintegration p(v3(acc0), v3(gyr0), v3(ba), v3(bg));
    for (int k = 0; k < n; ++k) p.push_back(dt[k], v3(acc + 3 * k), v3(gyr + 3 * k));
    export_preint(p, out);
    return 0;
}
// ImuFactor::Evaluate (ImuFactor.h:21-171), blocks Pi3 Qi4 SBi9 Pj3 Qj4 SBj9; 15 residuals
int ref_eval_imu(const glio_preint* pre, double gravity, double const* const* parameters, double* residuals, double** jacobians) {
    std::unique_ptr<Preintegration> p = make_preint(pre, gravity);
    ImuFactor f(p.get());
    return f.Evaluate(parameters, residuals, jacobians) ? 0 : 1;
}

// ---- GNSS ------------------------------------------------------------------------------------------------------------------------------
// dd_psr_factor_20::Evaluate (dd_psr_factor.hpp:25-171), blocks Pi3 Pj3 yaw1 anc3; 19 residuals (rows beyond n_sat - 1 are the factor's zero padding)
int ref_eval_dd_psr(const glio_dd_psr* g, double const* const* parameters, double* residuals, double** jacobians) {
    const int ns = g->n_sat;
    Eigen::MatrixXd W(ns - 1, ns - 1);
    for (int i = 0; i < ns - 1; ++i) for (int j = 0; j < ns - 1; ++j) W(i, j) = g->weight[i * (ns - 1) + j];
    dd_psr_factor_20 f(raws(ns, g->user_sat_pos, g->user_psr), raws(ns, g->ref_sat_pos, g->ref_psr), W, g->master, g->ratio, v3(g->station), g->threshold);
    return f.Evaluate(parameters, residuals, jacobians) ? 0 : 1;
}
// tcdopplerFactor (dopp_factor.hpp:19-85) as the estimator wraps it (Estimator.cpp:3176-3178): AutoDiffCostFunction<tcdopplerFactor, 1, 3, 9, 3, 9, EPOCH_SIZE, 1, 3>.
// EPOCH_SIZE is 5000 there; the clock-drift block is REF_DDT_SLOTS wide here (the functor only reads para_rcv_ddt[epoch]; a 5000-wide Jet per
// evaluation would only make the test slow).  jacobians[4] is therefore 1 x REF_DDT_SLOTS.
#define REF_DDT_SLOTS 16
int ref_ddt_slots(void) { return REF_DDT_SLOTS; }
int ref_eval_doppler(const glio_doppler* g, double const* const* parameters, double* residuals, double** jacobians) {
    if (g->epoch < 0 || g->epoch >= REF_DDT_SLOTS) return 2;
    nlosExclusion::GNSS_Raw raw;
    raw.sat_pos_x = g->sat_pos[0]; raw.sat_pos_y = g->sat_pos[1]; raw.sat_pos_z = g->sat_pos[2];
    raw.vel_x = g->sat_vel[0]; raw.vel_y = g->sat_vel[1]; raw.vel_z = g->sat_vel[2];
    raw.ddt = g->sv_ddt; raw.doppler = g->doppler; raw.lamda = g->lamda;
    Eigen::Matrix3d R;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R(i, j) = g->R_ecef_local[3 * i + j];
    ceres::AutoDiffCostFunction<tcdopplerFactor, 1, 3, 9, 3, 9, REF_DDT_SLOTS, 1, 3> f(new tcdopplerFactor(g->epoch, "GPS", raw, g->ratio, v3(g->lever_arm), R, g->var));
    return f.Evaluate(parameters, residuals, jacobians) ? 0 : 1;
}
// gnss_comm::ecef2rotation (gnss_comm/src/gnss_utility.cpp:750), row-major out
void ref_ecef2rotation(const double ecef[3], double R[9]) {
    const Eigen::Matrix3d M = gnss_comm::ecef2rotation(v3(ecef));
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[3 * i + j] = M(i, j);
}

// ---- marginalization prior as a factor ---------------------------------------------------------------------------------------------------
// MarginalizationFactor::Evaluate (MarginalizationFactor.cpp:233-287); parameters[b] = kept block b; jacobians[b] row-major n x size_b
int ref_eval_marg(const glio_prior* p, double const* const* parameters, double* residuals, double** jacobians) {
    std::vector<std::vector<double> > x0;
    std::unique_ptr<MarginalizationInfo> info(info_from_prior(p, x0));
    MarginalizationFactor f(info.get());
    return f.Evaluate(parameters, residuals, jacobians) ? 0 : 1;
}

// ---- the marginalization step ------------------------------------------------------------------------------------------------------------
// The factor list of Estimator.cpp:2462-2568 handed to the reference's MarginalizationInfo: the previous prior (drop = its slot-0 blocks),
// ImuFactor(0, 1) with drop {0, 1, 2}, LidarPlaneNormFactor + HuberLoss for every correspondence of every slot (drop {0, 1} on slot 0 only);
// then PreMarginalize, Marginalize, the addr_shift of :2584-2598 and GetParameterBlocks.  `marg` stays true as the class initialises it
// (Estimator.cpp:356): no SpeedBiasPriorFactor.
// Outputs in the REFERENCE'S OWN block order (unordered_map iteration): lin_jac [n][n] row-major, lin_res [n], per kept block its slot AFTER
// the shift, kind, offset (keep_block_idx - m) and linearisation point.  Returns n (< 0: error).
int ref_marginalize(int W, const double* trans /*[W][3]*/, const double* quat /*[W][4]*/, const double* speed_bias /*[W][9]*/,
                    const double qlb[4], const double tlb[3], double huber_delta, double gravity,
                    const int32_t* lidar_offset /*[W+1]*/, const float* lidar_pts /*[N][4]*/, const float* lidar_planes /*[N][4]*/, const double* lidar_scores,
                    const glio_preint* imu01, const glio_prior* prior /* may be NULL or n == 0 */,
                    double* lin_jac, double* lin_res, int32_t* blk_slot, int32_t* blk_kind, int32_t* blk_idx, double* blk_x0, int32_t* n_blocks_out) {
    // the estimator's parameter arrays: tmpTrans[W][3], tmpQuat[W][4], tmpSpeedBias[W][9] (Estimator.cpp:345-348) -- fixed addresses for the whole call
    std::vector<double> T(trans, trans + 3 * W), Q(quat, quat + 4 * W), SB(speed_bias, speed_bias + 9 * W);
    std::vector<double*> tmpTrans(W), tmpQuat(W), tmpSpeedBias(W);
    for (int i = 0; i < W; ++i) { tmpTrans[i] = &T[3 * i]; tmpQuat[i] = &Q[4 * i]; tmpSpeedBias[i] = &SB[9 * i]; }
    auto block_addr = [&](int slot, int kind) -> double* { return kind == GLIO_BLK_TRANS ? tmpTrans[slot] : (kind == GLIO_BLK_QUAT ? tmpQuat[slot] : tmpSpeedBias[slot]); };

    MarginalizationInfo* marginalization_info = new MarginalizationInfo();
    std::vector<std::vector<double> > x0_store;
    MarginalizationInfo* last_marginalization_info = nullptr;
    if (prior && prior->n > 0) {
        last_marginalization_info = info_from_prior(prior, x0_store);
        std::vector<double*> last_marginalization_parameter_blocks;
        for (int b = 0; b < prior->n_blocks; ++b) last_marginalization_parameter_blocks.push_back(block_addr(prior->blk_slot[b], prior->blk_kind[b]));
        std::vector<int> drop_set;                                                                       // Estimator.cpp:2465-2471
        for (int i = 0; i < static_cast<int>(last_marginalization_parameter_blocks.size()); i++)
            if (last_marginalization_parameter_blocks[i] == tmpTrans[0] || last_marginalization_parameter_blocks[i] == tmpQuat[0] || last_marginalization_parameter_blocks[i] == tmpSpeedBias[0])
                drop_set.push_back(i);
        MarginalizationFactor* marginalization_factor = new MarginalizationFactor(last_marginalization_info);
        marginalization_info->AddResidualBlockInfo(new ResidualBlockInfo(marginalization_factor, NULL, last_marginalization_parameter_blocks, drop_set));
    }
    std::unique_ptr<Preintegration> pre = make_preint(imu01, gravity);
    {                                                                                                    // Estimator.cpp:2520-2534
        ImuFactor* imuFactor = new ImuFactor(pre.get());
        marginalization_info->AddResidualBlockInfo(new ResidualBlockInfo(imuFactor, NULL,
            std::vector<double*>{tmpTrans[0], tmpQuat[0], tmpSpeedBias[0], tmpTrans[1], tmpQuat[1], tmpSpeedBias[1]}, std::vector<int>{0, 1, 2}));
    }
    std::vector<std::unique_ptr<ceres::LossFunction> > losses;                                           // (the reference leaks them; MarginalizationInfo does not own them)
    for (int s = 0; s < W; ++s) {                                                                        // Estimator.cpp:2538-2568
        losses.emplace_back(new ceres::HuberLoss(huber_delta));
        ceres::LossFunction* lossFunction = losses.back().get();
        std::vector<double*> tmp{tmpTrans[s], tmpQuat[s]};
        for (int i = lidar_offset[s]; i < lidar_offset[s + 1]; ++i) {
            Eigen::Vector3d currentPt(lidar_pts[4 * i], lidar_pts[4 * i + 1], lidar_pts[4 * i + 2]);
            Eigen::Vector3d norm(lidar_planes[4 * i], lidar_planes[4 * i + 1], lidar_planes[4 * i + 2]);
            double normInverse = lidar_planes[4 * i + 3];
            ceres::CostFunction* costFunction = LidarPlaneNormFactor::Create(currentPt, norm, q4(qlb), v3(tlb), normInverse, lidar_scores[i]);
            std::vector<int> drop_set;
            if (s == 0) { drop_set.push_back(0); drop_set.push_back(1); }
            marginalization_info->AddResidualBlockInfo(new ResidualBlockInfo(costFunction, lossFunction, tmp, drop_set));
        }
    }
    marginalization_info->PreMarginalize();
    marginalization_info->Marginalize();

    std::unordered_map<long, double*> addr_shift;                                                        // Estimator.cpp:2584-2598
    for (int i = 1; i < W; ++i) {
        addr_shift[reinterpret_cast<long>(tmpTrans[i])] = tmpTrans[i - 1];
        addr_shift[reinterpret_cast<long>(tmpQuat[i])] = tmpQuat[i - 1];
        addr_shift[reinterpret_cast<long>(tmpSpeedBias[i])] = tmpSpeedBias[i - 1];
    }
    std::vector<double*> parameter_blocks = marginalization_info->GetParameterBlocks(addr_shift);

    const int n = marginalization_info->n, m = marginalization_info->m;
    for (int i = 0; i < n; ++i) { lin_res[i] = marginalization_info->linearized_residuals(i); for (int j = 0; j < n; ++j) lin_jac[(size_t)i * n + j] = marginalization_info->linearized_jacobians(i, j); }
    const int nb = (int)parameter_blocks.size();
    *n_blocks_out = nb;
    for (int b = 0; b < nb; ++b) {
        int slot = -1, kind = -1;
        for (int s = 0; s < W && slot < 0; ++s)
            for (int k = 0; k < 3; ++k) if (parameter_blocks[b] == block_addr(s, k)) { slot = s; kind = k; break; }
        if (slot < 0) return -2;
        blk_slot[b] = slot; blk_kind[b] = kind; blk_idx[b] = marginalization_info->keep_block_idx[b] - m;
        for (int k = 0; k < 9; ++k) blk_x0[9 * b + k] = k < marginalization_info->keep_block_size[b] ? marginalization_info->keep_block_data[b][k] : 0.0;
    }
    delete marginalization_info;             // (deletes the factors, among them the MarginalizationFactor over last_marginalization_info)
    delete last_marginalization_info;
    return n;
}

}  // extern "C"
