// glio_backend.hpp -- host-side C++ mirror of the reference interface for the sliding-window hot path.
//
// Plain C++14, no HIP / torch / Ceres / Eigen headers: this is what a GLIO maintainer compiles into
// GLIO/src/Estimator.cpp and links against libglio_hip.so.  Two layers:
//   (1) factor shims with the exact ceres::CostFunction::Evaluate() signature (reference
//       GLIO/include/factors/LidarKeyframeFactor.h:73-122, ImuFactor.h:12-175) -- derive them from
//       ceres::SizedCostFunction<...> when Ceres is present;
//   (2) SlidingWindowBackend: the buffers optimizeSlidingWindowWithLandMark() works on (tmpTrans/tmpQuat/
//       tmpSpeedBias, Ps/Rs/Vs/Bas/Bgs, Estimator.cpp:345-348) and its call sequence (:2046-2736).
#pragma once

#include <array>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/glio_hip.h"

namespace glio {

struct Error : std::runtime_error {
    explicit Error(const std::string& what) : std::runtime_error(what + ": " + glio_last_error()) {}
};
inline void check(int rc, const char* what) { if (rc != GLIO_OK) throw Error(what); }

// The marginalization result: last_marginalization_info (linearized_jacobians / residuals, keep_block_*) and
// last_marginalization_parameter_blocks (MarginalizationFactor.h, Estimator.cpp:2584-2606) with blocks named by
// (slot, kind) -- already shifted to the NEXT window's slots.
struct MarginalizationPrior {
    int n = 0;
    std::vector<double> linearized_jacobians, linearized_residuals, keep_block_data;
    std::vector<int32_t> keep_block_slot, keep_block_kind, keep_block_idx;
    glio_prior view() const {
        glio_prior p;
        p.n = n; p.n_blocks = (int32_t)keep_block_slot.size();
        p.lin_jac = linearized_jacobians.data(); p.lin_res = linearized_residuals.data();
        p.blk_slot = keep_block_slot.data(); p.blk_kind = keep_block_kind.data(); p.blk_idx = keep_block_idx.data();
        p.blk_x0 = keep_block_data.data();
        return p;
    }
};

// ---- (1) factor shims ------------------------------------------------------------------------------
// LidarPlaneNormFactor: residual 1, blocks {t[3], q[4]} (LidarKeyframeFactor.h:112-114)
class LidarPlaneNormFactorHip {
public:
    LidarPlaneNormFactorHip(glio_ctx* ctx, const float cp[4], const float plane[4], double score) : ctx_(ctx), score_(score) {
        for (int k = 0; k < 4; ++k) { cp_[k] = cp[k]; plane_[k] = plane[k]; }
    }
    bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const {
        return glio_eval_lidar_plane(ctx_, cp_, plane_, score_, parameters, residuals, jacobians) == GLIO_OK;
    }
private:
    glio_ctx* ctx_; float cp_[4], plane_[4]; double score_;
};
// ImuFactor: residual 15, blocks {Pi3,Qi4,SBi9,Pj3,Qj4,SBj9} (ImuFactor.h:12)
class ImuFactorHip {
public:
    ImuFactorHip(glio_ctx* ctx, const glio_preint& pre) : ctx_(ctx), pre_(pre) {}
    bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const {
        return glio_eval_imu(ctx_, &pre_, parameters, residuals, jacobians) == GLIO_OK;
    }
private:
    glio_ctx* ctx_; glio_preint pre_;
};

// ---- (2) the window ----------------------------------------------------------------------------------
// dd_psr_factor_20 (dd_psr_factor.hpp:15-171): 19 residuals, blocks {Pi3, Pj3, yaw1, anc3}
class DdPsrFactorHip {
public:
    DdPsrFactorHip(glio_ctx* ctx, const glio_dd_psr& f) : ctx_(ctx), f_(f) {}
    bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const {
        return glio_eval_dd_psr(ctx_, &f_, parameters, residuals, jacobians) == GLIO_OK;
    }
private:
    glio_ctx* ctx_; glio_dd_psr f_;
};
// tcdopplerFactor (dopp_factor.hpp:19-85): 1 residual, blocks {Pi3, SBi9, Pj3, SBj9, rcv_ddt[EPOCH_SIZE], yaw1, anc3};
// jacobians[4], when requested, must point to EPOCH_SIZE doubles in Ceres: the shim a maintainer writes zero-fills it and
// stores the single value glio_eval_doppler returns at [epoch]
class DopplerFactorHip {
public:
    DopplerFactorHip(glio_ctx* ctx, const glio_doppler& f) : ctx_(ctx), f_(f) {}
    bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const {
        return glio_eval_doppler(ctx_, &f_, parameters, residuals, jacobians) == GLIO_OK;
    }
private:
    glio_ctx* ctx_; glio_doppler f_;
};
// MarginalizationFactor (MarginalizationFactor.cpp:223-287): prior->n residuals, one block per kept parameter block
class MarginalizationFactorHip {
public:
    MarginalizationFactorHip(glio_ctx* ctx, const glio_prior* prior) : ctx_(ctx), prior_(prior) {}
    bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const {
        return glio_eval_marginalization(ctx_, prior_, parameters, residuals, jacobians) == GLIO_OK;
    }
private:
    glio_ctx* ctx_; const glio_prior* prior_;
};
// BinaryLidarPlaneNormFactor (LidarKeyframeFactor.h:124-164): 1 residual, blocks {t1 3, q1 4, t2 3, q2 4}
class BinaryLidarPlaneNormFactorHip {
public:
    BinaryLidarPlaneNormFactorHip(glio_ctx* ctx, const float cp[4], const double norm_cent[6], double score) : ctx_(ctx), score_(score) {
        for (int k = 0; k < 4; ++k) cp_[k] = cp[k];
        for (int k = 0; k < 6; ++k) nc_[k] = norm_cent[k];
    }
    bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const {
        return glio_eval_binary_plane(ctx_, cp_, nc_, score_, parameters, residuals, jacobians) == GLIO_OK;
    }
private:
    glio_ctx* ctx_; float cp_[4]; double nc_[6]; double score_;
};

// Estimator.cpp:2611-2726: copy the solution back into the estimator's members with the reference's sanity gates.
//   tmpTrans [W][3], tmpQuat [W][4] (w,x,y,z), tmpSpeedBias [W][9], tmp_rcv_dt [W][3] (may be null): the solved blocks
//   Ps / Vs [W][3], Qs [W][4] (the reference keeps Rs as matrices), para_speed_bias [W][9]                : in/out
//   Bas / Bgs [W][3], abs_poses [W][7] (q w,x,y,z then t, :2651-2666), rcv_dt [W][3]                       : out, any may be null
// A component failing its gate keeps its old value (Q14: |dp| < 100, |dq.vec| < 10, |dv| < 100, |db| < 22).  abs_poses takes the
// RAW solved quaternion, Rs the normalised one (:2661-2669).  The six bias gates are chained by dangling `else`s (the
// ROS_WARNs between them are commented out, :2689-2722), so only the FIRST bias component that passes is written (Q16).
// rcv_dt is copied unconditionally (:2645-2647; the blocks carry no residuals, Q6) -- and the reference writes it ONE KEYFRAME EARLIER than
// the other arrays: rcv_dt[i-1] <- tmp_rcv_dt[slot of i] while Ps[i], Vs[i] ... take slot i.  Row s of every array here is window slot s,
// so pass rcv_dt = &rcv_dt[first_idx - 1][0] where the others get &X[first_idx] (host_writeback_test.cpp does).
inline void writeBackState(int W, const double* tmpTrans, const double* tmpQuat, const double* tmpSpeedBias, const double* tmp_rcv_dt,
                           double* Ps, double* Qs, double* Vs, double* para_speed_bias, double* Bas, double* Bgs, double* abs_poses,
                           double* rcv_dt) {
    for (int i = 0; i < W; ++i) {
        const double* t = tmpTrans + 3 * i; const double* q = tmpQuat + 4 * i; const double* sb = tmpSpeedBias + 9 * i;
        double dp = 0, dv = 0;
        for (int k = 0; k < 3; ++k) { dp += (Ps[3 * i + k] - t[k]) * (Ps[3 * i + k] - t[k]); dv += (Vs[3 * i + k] - sb[k]) * (Vs[3 * i + k] - sb[k]); }
        // dq = normalized(tmpQuat)^-1 * Rs ; qnorm = |dq.vec|
        const double qn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        const double a[4] = {q[0] / qn, -q[1] / qn, -q[2] / qn, -q[3] / qn};
        const double* b = Qs + 4 * i;
        const double vx = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
        const double vy = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
        const double vz = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
        const double qnorm = std::sqrt(vx * vx + vy * vy + vz * vz);
        if (rcv_dt && tmp_rcv_dt) for (int k = 0; k < 3; ++k) rcv_dt[3 * i + k] = tmp_rcv_dt[3 * i + k];
        if (std::sqrt(dp) < 100) {
            for (int k = 0; k < 3; ++k) { Ps[3 * i + k] = t[k]; if (abs_poses) abs_poses[7 * i + 4 + k] = t[k]; }
        }
        if (qnorm < 10) {
            for (int k = 0; k < 4; ++k) { Qs[4 * i + k] = q[k] / qn; if (abs_poses) abs_poses[7 * i + k] = q[k]; }
        }
        if (std::sqrt(dv) < 100) for (int k = 0; k < 3; ++k) { Vs[3 * i + k] = sb[k]; para_speed_bias[9 * i + k] = sb[k]; }
        for (int k = 3; k < 9; ++k)
            if (std::fabs(para_speed_bias[9 * i + k] - sb[k]) < 22) {          // Q16: the first component that passes, and only that one
                para_speed_bias[9 * i + k] = sb[k];
                if (k < 6) { if (Bas) Bas[3 * i + k - 3] = sb[k]; } else if (Bgs) Bgs[3 * i + k - 6] = sb[k];
                break;
            }
    }
}

// featureSelection (Estimator.cpp:3894-3992), the draws only: which of `count` correspondences of a slot stay.  Returns false when nothing changes
// (count - 1 < feature_res_num: the early return at :3906-3909, quirk Q9 -- the set is kept WHOLE, not cut to feature_res_num); otherwise `kept` holds
// the original indices in draw order: feature_res_num draws, each uniform over the records still left (the reference builds a no-repeat random array
// over the remaining records and takes its LAST element, :3948-3957, then erases that record, :3964-3978 -- i.e. one uniform draw per kept record,
// whatever rand_set_num is); random_select == false empties the slot (:3945 never enters the loop, :3981-3987 install the empty sets).
// rand_below(n): uniform integer in [0, n) -- the reference seeds from std::random_device (random_generator.hpp:58), so the generator is the caller's.
// The Python twin is glio_amd/sliding.py::feature_selection_draws: same generator in, same indices out (tests/test_host_cpp.py).
template <typename RandBelow>
inline bool featureSelectionDraws(int64_t count, int feature_res_num, RandBelow&& rand_below, bool random_select, std::vector<int32_t>& kept) {
    kept.clear();
    if (count < 1 || count - 1 < (int64_t)feature_res_num) return false;
    if (!random_select) return true;
    // the d-th draw picks the k-th record STILL LEFT (k uniform below count - d): its original index is k advanced past every removed index <= it.
    // `gone` stays sorted: O(feature_res_num^2) whatever the count (erasing from a 64k-element list per draw would be O(count) each)
    std::vector<int32_t> gone;
    gone.reserve((size_t)feature_res_num);
    for (int d = 0; d < feature_res_num; ++d) {
        int64_t v = (int64_t)rand_below((uint64_t)(count - d));
        size_t pos = 0;
        while (pos < gone.size() && (int64_t)gone[pos] <= v) { ++v; ++pos; }
        gone.insert(gone.begin() + (std::ptrdiff_t)pos, (int32_t)v);
        kept.push_back((int32_t)v);
    }
    return true;
}

// how the caller's clouds lie in memory: records of stride_bytes, x y z as floats at offset 0, the intensity as a float at intensity_offset
struct PointLayout {
    int stride_bytes, intensity_offset;
    static PointLayout packed() { return {16, 12}; }
    static PointLayout pclXYZI() { return {32, 16}; }        // pcl::PointXYZI, the reference's PointType (GLIO/include/utils/common.h)
};

class SlidingWindowBackend {
public:
    explicit SlidingWindowBackend(const glio_opts& opts, int device = 0) : opts_(opts), W_(opts.window) {
        check(glio_create(device, &opts_, &ctx_), "glio_create");
        tmpTrans.assign(3 * W_, 0.0); tmpQuat.assign(4 * W_, 0.0); tmpSpeedBias.assign(9 * W_, 0.0);
        for (int i = 0; i < W_; ++i) tmpQuat[4 * i] = 1.0;
    }
    ~SlidingWindowBackend() { glio_destroy(ctx_); }
    SlidingWindowBackend(const SlidingWindowBackend&) = delete;
    SlidingWindowBackend& operator=(const SlidingWindowBackend&) = delete;

    glio_ctx* ctx() const { return ctx_; }
    int window() const { return W_; }

    // Estimator.cpp:2056  kd_tree_surf_local_map->setInputCloud(surf_local_map_ds)
    void setLocalMap(const float* xyzi, int n) { check(glio_set_map(ctx_, xyzi, n), "glio_set_map"); map_points_ = n; }
    // the same straight from cloud->points.data() of a pcl::PointCloud<pcl::PointXYZI> (32-byte records, intensity at byte 16): PointLayout::pclXYZI()
    void setLocalMap(const void* points, int n, PointLayout l) { check(glio_set_map_strided(ctx_, points, n, l.stride_bytes, l.intensity_offset), "glio_set_map_strided"); map_points_ = n; }
    // `if (surf_local_map_ds->points.size() > 50)` (Estimator.cpp:2221,2244): with a smaller map the reference adds NO LiDAR factor
    // for the keyframe ("Not enough feature points from the map")
    bool mapLargeEnough() const { return map_points_ > 50; }
    // Estimator.cpp:2216-2222  Q2 = Q*q_lb^-1, T2 = T - Q2*t_lb, findCorrespondingSurfFeatures(idx-1, Q2, T2)
    int findCorrespondingSurfFeatures(int slot, const float* scan_xyzi, int n) {
        if (!mapLargeEnough()) {
            check(glio_set_scan(ctx_, slot, scan_xyzi, n), "glio_set_scan");
            check(glio_select_correspondences(ctx_, slot, nullptr, 0), "glio_select_correspondences");    // the slot carries no residuals
            return 0;
        }
        double q2[4], t2[3];
        lidarPose(slot, q2, t2);
        int cnt = 0;
        check(glio_associate(ctx_, slot, scan_xyzi, n, q2, t2, &cnt), "glio_associate");
        return cnt;
    }
    // Estimator.cpp:2223  featureSelection(idx-1, Q2_, T2_), right behind the slot's findCorrespondingSurfFeatures: `count` = what the search kept for the
    // slot (findCorrespondingSurfFeatures' return value / windowCounts()[slot]); the gather runs on the device.  Returns the slot's residual count.
    template <typename RandBelow>
    int featureSelection(int slot, int count, int feature_res_num, RandBelow&& rand_below, bool random_select = true) {
        if (!featureSelectionDraws(count, feature_res_num, rand_below, random_select, sel_)) return count;
        check(glio_select_correspondences(ctx_, slot, sel_.empty() ? nullptr : sel_.data(), (int)sel_.size()), "glio_select_correspondences");
        return (int)sel_.size();
    }
    // ... for all W slots of the window at once (the loop of :2198-2248 calls it slot after slot; the draws are made in slot order, exactly as W calls of
    // featureSelection() would make them -- same generator in, same records kept): one upload and two launches instead of W synchronising calls.
    // counts: in = what the searches kept (windowCounts()), out = the residual counts.
    template <typename RandBelow>
    void featureSelectionWindow(std::vector<int32_t>& counts, int feature_res_num, RandBelow&& rand_below, bool random_select = true) {
        std::vector<int32_t> offsets((size_t)W_ + 1, 0), idx;
        std::vector<uint8_t> changed((size_t)W_, 0);
        for (int s = 0; s < W_; ++s) {
            if (featureSelectionDraws(counts[s], feature_res_num, rand_below, random_select, sel_)) { changed[s] = 1; idx.insert(idx.end(), sel_.begin(), sel_.end()); counts[s] = (int32_t)sel_.size(); }
            offsets[s + 1] = (int32_t)idx.size();
        }
        check(glio_select_correspondences_window(ctx_, offsets.data(), idx.empty() ? nullptr : idx.data(), changed.data()), "glio_select_correspondences_window");
    }
    // Estimator.cpp:2182-2192 / 2153-2158 / 2329-2359
    void setImuFactors(const std::vector<glio_preint>& pre) {
        std::vector<int32_t> slots(pre.size());
        for (size_t k = 0; k < pre.size(); ++k) slots[k] = (int32_t)k;
        check(glio_set_imu(ctx_, (int)pre.size(), pre.data(), slots.data()), "glio_set_imu");
    }
    void setMarginalizationPrior(const glio_prior* prior) { check(glio_set_prior(ctx_, prior), "glio_set_prior"); }
    void setGnss(const glio_gnss_frame* frame, const std::vector<glio_dd_psr>& dd, const std::vector<glio_doppler>& dop) {
        check(glio_set_gnss(ctx_, frame, (int)dd.size(), dd.data(), (int)dop.size(), dop.data()), "glio_set_gnss");
    }

    // Estimator.cpp:2424-2433 ceres::Solve + :2439-2457 quaternion sign unification
    glio_summary solve(std::vector<double>* rcv_ddt = nullptr) {
        glio_state st;
        st.trans = tmpTrans.data(); st.quat = tmpQuat.data(); st.speed_bias = tmpSpeedBias.data();
        st.rcv_ddt = rcv_ddt && !rcv_ddt->empty() ? rcv_ddt->data() : nullptr;
        st.n_ddt = rcv_ddt ? (int)rcv_ddt->size() : 0;
        glio_summary sum;
        check(glio_solve(ctx_, &st, &sum), "glio_solve");
        for (int i = 0; i < W_; ++i)
            if (tmpQuat[4 * i] < 0) for (int k = 0; k < 4; ++k) tmpQuat[4 * i + k] = -tmpQuat[4 * i + k];   // unifyQuaternion
        return sum;
    }

    // Estimator.cpp:2462-2607: marginalize the oldest keyframe at the solved state (call after solve()); the
    // result is what setMarginalizationPrior() takes for the next window.
    MarginalizationPrior marginalize() {
        const int n = 6 * (W_ - 1) + 9, nb = 2 * (W_ - 1) + 1;
        MarginalizationPrior m;
        m.linearized_jacobians.assign((size_t)n * n, 0.0); m.linearized_residuals.assign(n, 0.0); m.keep_block_data.assign((size_t)nb * 9, 0.0);
        m.keep_block_slot.assign(nb, 0); m.keep_block_kind.assign(nb, 0); m.keep_block_idx.assign(nb, 0);
        glio_state st;
        st.trans = tmpTrans.data(); st.quat = tmpQuat.data(); st.speed_bias = tmpSpeedBias.data(); st.rcv_ddt = nullptr; st.n_ddt = 0;
        int32_t on = 0, onb = 0;
        check(glio_marginalize(ctx_, &st, m.linearized_jacobians.data(), m.linearized_residuals.data(), m.keep_block_slot.data(),
                               m.keep_block_kind.data(), m.keep_block_idx.data(), m.keep_block_data.data(), &on, &onb), "glio_marginalize");
        m.n = on;
        return m;
    }

    // ---- the same keyframe cycle with everything kept on the device between keyframes (what glio_amd/sliding.py's
    // ResidentSlidingWindow does in Python): only the NEW keyframe's scan crosses PCIe, all W slots are associated in one
    // call, the marginalization result stays on the device as the next window's prior.
    // surf_frames bookkeeping of Estimator.cpp:4240-4300: slot s <- slot s+1 (scans and their correspondences)
    void slideWindow() { check(glio_slide_window(ctx_), "glio_slide_window"); }
    void setScan(int slot, const float* scan_xyzi, int n) { check(glio_set_scan(ctx_, slot, scan_xyzi, n), "glio_set_scan"); }
    void setScan(int slot, const void* points, int n, PointLayout l) { check(glio_set_scan_strided(ctx_, slot, points, n, l.stride_bytes, l.intensity_offset), "glio_set_scan_strided"); }
    // the loop over idx of Estimator.cpp:2216-2222 for the whole window, with the poses held in tmpTrans / tmpQuat
    std::vector<int32_t> findCorrespondingSurfFeaturesWindow() {
        std::vector<double> q2(4 * W_), t2(3 * W_);
        for (int s = 0; s < W_; ++s) lidarPose(s, &q2[4 * s], &t2[3 * s]);
        std::vector<int32_t> counts(W_, 0);
        if (!mapLargeEnough()) {                                    // Estimator.cpp:2221: no LiDAR factors from a map of <= 50 points
            for (int s = 0; s < W_; ++s) check(glio_select_correspondences(ctx_, s, nullptr, 0), "glio_select_correspondences");
            return counts;
        }
        check(glio_associate_window(ctx_, q2.data(), t2.data(), counts.data()), "glio_associate_window");
        return counts;
    }
    // the same without waiting: the searches are enqueued and the call returns -- the caller fills the window's factor tables (setImuFactors, setGnss)
    // while the GPU searches; windowCounts() (or the next solve) waits and returns the per-slot correspondence counts
    void findCorrespondingSurfFeaturesWindowAsync() {
        if (!mapLargeEnough()) {
            for (int s = 0; s < W_; ++s) check(glio_select_correspondences(ctx_, s, nullptr, 0), "glio_select_correspondences");
            return;
        }
        std::vector<double> q2(4 * W_), t2(3 * W_);
        for (int s = 0; s < W_; ++s) lidarPose(s, &q2[4 * s], &t2[3 * s]);
        check(glio_associate_window_async(ctx_, q2.data(), t2.data()), "glio_associate_window_async");
    }
    std::vector<int32_t> windowCounts() {
        std::vector<int32_t> counts(W_, 0);
        if (mapLargeEnough()) check(glio_associate_window_counts(ctx_, counts.data()), "glio_associate_window_counts");
        return counts;
    }
    // the keyframe cloud that setScan() just put into window slot `scan_slot` goes into the local map from device memory (no second upload):
    // body point = scan point - lidar_offset; then the ring map and its search structure are rebuilt
    int pushScanAndBuildLocalMap(int scan_slot, const float lidar_offset[3], const double q[4], const double t[3]) {
        check(glio_localmap_push_scan(ctx_, scan_slot, lidar_offset, q, t), "glio_localmap_push_scan");
        int pts = 0;
        check(glio_localmap_build(ctx_, &pts), "glio_localmap_build");
        map_points_ = pts;
        return pts;
    }
    // Estimator.cpp:2462-2607 with the result kept resident as the prior of the next window (no J0 read-back)
    // (rcv_ddt: the clock-drift slots of the solved state when the window carries Doppler factors -- the state handed over must be the solve's)
    void marginalizeAndKeep(std::vector<double>* rcv_ddt = nullptr) {
        glio_state st;
        st.trans = tmpTrans.data(); st.quat = tmpQuat.data(); st.speed_bias = tmpSpeedBias.data();
        st.rcv_ddt = rcv_ddt && !rcv_ddt->empty() ? rcv_ddt->data() : nullptr; st.n_ddt = rcv_ddt ? (int)rcv_ddt->size() : 0;
        check(glio_marginalize_keep(ctx_, &st), "glio_marginalize_keep");
    }
    // the same in two halves: the marginalization enqueued (the host is free: setScanAhead() of the next keyframe's cloud, say), then waited for
    void marginalizeAndKeepAsync(std::vector<double>* rcv_ddt = nullptr) {
        glio_state st;
        st.trans = tmpTrans.data(); st.quat = tmpQuat.data(); st.speed_bias = tmpSpeedBias.data();
        st.rcv_ddt = rcv_ddt && !rcv_ddt->empty() ? rcv_ddt->data() : nullptr; st.n_ddt = rcv_ddt ? (int)rcv_ddt->size() : 0;
        check(glio_marginalize_keep_async(ctx_, &st), "glio_marginalize_keep_async");
    }
    void marginalizeFinish() { check(glio_marginalize_keep_finish(ctx_), "glio_marginalize_keep_finish"); }
    // the NEXT keyframe's cloud, sent during this keyframe's call (after the solve): the next call's slideWindow() finds it in slot W - 1 and makes no setScan()
    void setScanAhead(const float* scan_xyzi, int n) { check(glio_set_scan_ahead(ctx_, scan_xyzi, n), "glio_set_scan_ahead"); }
    void setScanAhead(const void* points, int n, PointLayout l) { check(glio_set_scan_ahead_strided(ctx_, points, n, l.stride_bytes, l.intensity_offset), "glio_set_scan_ahead_strided"); }
    // ... and the next call's local map behind it (the cloud just sent ahead pushed at the new keyframe's pose, the ring map and its search structure rebuilt): the
    // next call makes no pushScanAndBuildLocalMap()
    int pushScanAheadAndBuildLocalMap(const float lidar_offset[3], const double q[4], const double t[3]) {
        int pts = 0;
        check(glio_localmap_push_scan_ahead_and_build(ctx_, lidar_offset, q, t, &pts), "glio_localmap_push_scan_ahead_and_build");
        map_points_ = pts;
        return pts;
    }
    // buildLocalMapWithLandMark + downSampleCloud (Estimator.cpp:3529-3631) on the device: push the new keyframe's cloud
    // (body frame) with its pose, rebuild the voxel-averaged ring map and its search structure; returns the map size
    void configureLocalMap(int width, float leaf, int max_points_per_keyframe) {
        check(glio_localmap_config(ctx_, width, leaf, max_points_per_keyframe), "glio_localmap_config");
    }
    int pushKeyframeAndBuildLocalMap(const void* points, int n, PointLayout l, const double q[4], const double t[3]) {
        check(glio_localmap_push_strided(ctx_, points, n, l.stride_bytes, l.intensity_offset, q, t), "glio_localmap_push_strided");
        int pts = 0;
        check(glio_localmap_build(ctx_, &pts), "glio_localmap_build");
        map_points_ = pts;
        return pts;
    }
    int pushKeyframeAndBuildLocalMap(const float* cloud_xyzi, int n, const double q[4], const double t[3]) {
        check(glio_localmap_push(ctx_, cloud_xyzi, n, q, t), "glio_localmap_push");
        int pts = 0;
        check(glio_localmap_build(ctx_, &pts), "glio_localmap_build");
        map_points_ = pts;
        return pts;
    }
    // shift the host-side state like the reference's slideWindow(): the newest slot is initialised by the caller
    void slideState(const double new_t[3], const double new_q[4], const double new_sb[9]) {
        for (int i = 0; i + 1 < W_; ++i) {
            for (int k = 0; k < 3; ++k) tmpTrans[3 * i + k] = tmpTrans[3 * (i + 1) + k];
            for (int k = 0; k < 4; ++k) tmpQuat[4 * i + k] = tmpQuat[4 * (i + 1) + k];
            for (int k = 0; k < 9; ++k) tmpSpeedBias[9 * i + k] = tmpSpeedBias[9 * (i + 1) + k];
        }
        for (int k = 0; k < 3; ++k) tmpTrans[3 * (W_ - 1) + k] = new_t[k];
        for (int k = 0; k < 4; ++k) tmpQuat[4 * (W_ - 1) + k] = new_q[k];
        for (int k = 0; k < 9; ++k) tmpSpeedBias[9 * (W_ - 1) + k] = new_sb[k];
    }

    // Estimator.cpp:2611-2726 write-back with the reference's sanity gates (writeBackState below, on this backend's tmp arrays)
    void writeBack(double* Ps, double* Qs, double* Vs, double* para_speed_bias, double* Bas = nullptr, double* Bgs = nullptr,
                   double* abs_poses = nullptr, double* rcv_dt = nullptr, const double* tmp_rcv_dt = nullptr) const {
        writeBackState(W_, tmpTrans.data(), tmpQuat.data(), tmpSpeedBias.data(), tmp_rcv_dt, Ps, Qs, Vs, para_speed_bias, Bas, Bgs, abs_poses, rcv_dt);
    }

    // state, laid out like the reference's double arrays
    std::vector<double> tmpTrans, tmpQuat, tmpSpeedBias;

private:
    void lidarPose(int slot, double q2[4], double t2[3]) const {
        const double* q = &tmpQuat[4 * slot]; const double* t = &tmpTrans[3 * slot];
        const double* l = opts_.q_lb;
        const double n2 = l[0] * l[0] + l[1] * l[1] + l[2] * l[2] + l[3] * l[3];
        const double li[4] = {l[0] / n2, -l[1] / n2, -l[2] / n2, -l[3] / n2};
        q2[0] = q[0] * li[0] - q[1] * li[1] - q[2] * li[2] - q[3] * li[3];
        q2[1] = q[0] * li[1] + q[1] * li[0] + q[2] * li[3] - q[3] * li[2];
        q2[2] = q[0] * li[2] + q[2] * li[0] + q[3] * li[1] - q[1] * li[3];
        q2[3] = q[0] * li[3] + q[3] * li[0] + q[1] * li[2] - q[2] * li[1];
        // T2 = T - Q2 * t_lb  (Eigen q*v)
        const double* v = opts_.t_lb;
        double uv[3] = {q2[2] * v[2] - q2[3] * v[1], q2[3] * v[0] - q2[1] * v[2], q2[1] * v[1] - q2[2] * v[0]};
        for (double& x : uv) x += x;
        const double uuv[3] = {q2[2] * uv[2] - q2[3] * uv[1], q2[3] * uv[0] - q2[1] * uv[2], q2[1] * uv[1] - q2[2] * uv[0]};
        for (int k = 0; k < 3; ++k) t2[k] = t[k] - (v[k] + q2[0] * uv[k] + uuv[k]);
    }
    glio_opts opts_;
    int W_;
    std::vector<int32_t> sel_;
    int map_points_ = 0;
    glio_ctx* ctx_ = nullptr;
};

// ---- (3) the front end -------------------------------------------------------------------------------
// LidarOdometry (GLIO/src/LidarOdometry.cpp): scan-to-map odometry on the same C-ABI with a one-keyframe window.  Per scan, run() (:661-699):
//   poseInitialization (:405-432)  abs_pose <- abs_pose o rel_pose
//   buildLocalMap (:268-292)       the map = the LAST 20 frames' downsampled surf clouds at their solved poses (recent_surf_frames; frame 0 never enters:
//                                  while fewer than 2 poses exist the map is the current scan itself, :271-275) -- here the device-resident ring
//                                  (glio_localmap_config(20, 0.2): one cloud crosses PCIe per scan)
//   downSampleCloud (:306-314)     VoxelGrid 0.2 m over the map (glio_localmap_build); the scan arrives downsampled (surf_last_ds: the caller's filter)
//   updateTransformationWithCeres (:474-581)  match_cnt rounds (8 while fewer than 2 poses exist, else scan_match_cnt, :492-497) of
//                                  [findCorrespondingSurfFeatures at the current abs_pose (:343-404: gates 1.0 / 0.06 / 0.4), one problem of
//                                  LidarPlaneNormIncreFactor under HuberLoss(0.1), Levenberg-Marquardt, max_num_iter, 15 ms budget, unifyQuaternion]
//   savePoses (:316-341), computeRelative (:434-471)  rel_pose <- pose[previous]^-1 o abs_pose
// update() is the solve alone against a map the caller set (kd_tree_surf_last->setInputCloud(surf_from_map_ds), :482).
// The Python twin is glio_amd/odometry.py::ScanToMapOdometry (tests/test_host_cpp.py holds the two to each other bit for bit).
class ScanToMapOdometry {
public:
    struct Round { glio_summary summary; int kept; };
    // glio_opts of the front end: the yaml's `lidar_odometry` block + the constants of LidarOdometry.cpp (odometry.frontend_opts)
    static glio_opts frontendOpts(int max_points, int max_map_points, int max_num_iter = 12) {
        glio_opts o;
        glio_opts_default(&o);
        o.window = 1; o.max_iterations = max_num_iter;                              // config_urban_hk.yaml:19
        o.max_points_per_scan = max_points > 64 ? max_points : 64; o.max_map_points = max_map_points > 64 ? max_map_points : 64; o.max_ddt_epochs = 0;
        o.jacobi_scaling = 1;
        o.kd_max_radius = 1.0; o.surf_dist_thres = 0.06; o.weight_gate = 0.4;       // LidarOdometry.cpp:356,379,392
        o.huber_delta = 0.1; o.doppler_huber_delta = 1.0;                           // :499
        o.q_lb[0] = 1.0; o.q_lb[1] = o.q_lb[2] = o.q_lb[3] = 0.0;
        o.t_lb[0] = o.t_lb[1] = o.t_lb[2] = 0.0;                                    // LidarPlaneNormIncreFactor applies no extrinsic
        o.lidar_const = 7.5;
        o.unit_scores = 1;                                                          // ... and carries no score (LidarKeyframeFactor.h:222-257)
        o.trust_region_strategy = 1;                                                // Ceres default LEVENBERG_MARQUARDT (solverOptions :521-527)
        o.max_solver_time_s = 0.015;                                                // :524
        return o;
    }
    explicit ScanToMapOdometry(const glio_opts& opts, int device = 0, int scan_match_cnt = 1, int local_map_width = 20, float leaf = 0.2f)
        : opts_(opts), scan_match_cnt_(scan_match_cnt) {
        if (opts.window != 1) throw std::invalid_argument("ScanToMapOdometry: the front end is a one-keyframe window");
        check(glio_create(device, &opts_, &ctx_), "glio_create");
        check(glio_localmap_config(ctx_, local_map_width, leaf, opts_.max_points_per_scan), "glio_localmap_config");
        // the window never carries another factor
        check(glio_set_imu(ctx_, 0, nullptr, nullptr), "glio_set_imu");
        check(glio_set_prior(ctx_, nullptr), "glio_set_prior");
        check(glio_set_gnss(ctx_, nullptr, 0, nullptr, 0, nullptr), "glio_set_gnss");
        abs_pose = {{1, 0, 0, 0, 0, 0, 0}}; rel_pose = {{1, 0, 0, 0, 0, 0, 0}};
    }
    ~ScanToMapOdometry() { glio_destroy(ctx_); }
    ScanToMapOdometry(const ScanToMapOdometry&) = delete;
    ScanToMapOdometry& operator=(const ScanToMapOdometry&) = delete;
    glio_ctx* ctx() const { return ctx_; }

    // kd_tree_surf_last->setInputCloud(surf_from_map_ds) (:482) with a map the caller built
    void setMap(const float* xyzi, int n) { check(glio_set_map(ctx_, xyzi, n), "glio_set_map"); map_points_ = n; }
    void setMap(const void* points, int n, PointLayout l) { check(glio_set_map_strided(ctx_, points, n, l.stride_bytes, l.intensity_offset), "glio_set_map_strided"); map_points_ = n; }

    // updateTransformationWithCeres (:474-581) from the given pose (q w,x,y,z then t, the reference's abs_pose[7]); returns the new pose
    std::array<double, 7> update(const float* surf_last_ds, int n, const std::array<double, 7>& pose, int match_cnt, std::vector<Round>* rounds = nullptr) {
        std::array<double, 7> p = pose;
        if (rounds) rounds->clear();
        if (map_points_ < 10) return p;                                             // "Not enough feature points from the map" (:477-480)
        check(glio_set_scan(ctx_, 0, surf_last_ds, n), "glio_set_scan");
        double sb[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int it = 0; it < match_cnt; ++it) {
            int kept = 0;
            check(glio_associate_resident(ctx_, 0, &p[0], &p[4], &kept), "glio_associate_resident");
            glio_state st;
            st.trans = &p[4]; st.quat = &p[0]; st.speed_bias = sb; st.rcv_ddt = nullptr; st.n_ddt = 0;
            glio_summary sum;
            check(glio_solve(ctx_, &st, &sum), "glio_solve");
            if (p[0] < 0) for (int k = 0; k < 4; ++k) p[k] = -p[k];                 // unifyQuaternion (:532-542)
            if (rounds) rounds->push_back({sum, kept});
        }
        return p;
    }

    // LidarOdometry::run() (:661-699) for one scan; surf_last_ds = the scan after the caller's 0.2 m filter (down_size_filter_surf, :312-313).  Returns abs_pose.
    std::array<double, 7> run(const float* surf_last_ds, int n, std::vector<Round>* rounds = nullptr) {
        if (rounds) rounds->clear();
        if (poses_ == 0) {                                                          // !system_initialized: savePoses(), nothing else (:671-675)
            savePoses(surf_last_ds, n);
            return abs_pose;
        }
        // poseInitialization: t0 = q0 * dt + t0, q0 = q0 * dq
        {
            double t[3], q[4];
            rotate(&abs_pose[0], &rel_pose[4], t);
            for (int k = 0; k < 3; ++k) t[k] += abs_pose[4 + k];
            qmul(&abs_pose[0], &rel_pose[0], q);
            for (int k = 0; k < 4; ++k) abs_pose[k] = q[k];
            for (int k = 0; k < 3; ++k) abs_pose[4 + k] = t[k];
        }
        // buildLocalMap + downSampleCloud
        if (poses_ <= 1) setMap(surf_last_ds, n);                                   // the current scan is its own map (:271-275; the two filters see the same cloud)
        else {
            check(glio_localmap_push(ctx_, last_cloud_.data(), (int)(last_cloud_.size() / 4), &last_pose_[0], &last_pose_[4]), "glio_localmap_push");
            int pts = 0;
            check(glio_localmap_build(ctx_, &pts), "glio_localmap_build");
            map_points_ = pts;
        }
        abs_pose = update(surf_last_ds, n, abs_pose, poses_ < 2 ? 8 : scan_match_cnt_, rounds);
        const std::array<double, 7> prev = last_pose_;
        savePoses(surf_last_ds, n);
        // computeRelative: rel = prev^-1 o abs
        {
            const double qi[4] = {prev[0], -prev[1], -prev[2], -prev[3]};           // (unit quaternions: the inverse is the conjugate, as Eigen's inverse() of a normalised one)
            const double n2 = prev[0] * prev[0] + prev[1] * prev[1] + prev[2] * prev[2] + prev[3] * prev[3];
            const double qin[4] = {qi[0] / n2, qi[1] / n2, qi[2] / n2, qi[3] / n2};
            double q[4], d[3] = {abs_pose[4] - prev[4], abs_pose[5] - prev[5], abs_pose[6] - prev[6]}, t[3];
            qmul(qin, &abs_pose[0], q);
            rotate(qin, d, t);
            for (int k = 0; k < 4; ++k) rel_pose[k] = q[k];
            for (int k = 0; k < 3; ++k) rel_pose[4 + k] = t[k];
        }
        return abs_pose;
    }
    int frames() const { return poses_; }
    int mapPoints() const { return map_points_; }

    std::array<double, 7> abs_pose, rel_pose;          // q (w,x,y,z), t -- the reference's abs_pose[7] / rel_pose[7]

private:
    void savePoses(const float* cloud, int n) {
        last_pose_ = abs_pose;
        last_cloud_.assign(cloud, cloud + 4 * (size_t)n);
        ++poses_;
    }
    static void qmul(const double a[4], const double b[4], double o[4]) {
        o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
        o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
        o[2] = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
        o[3] = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
    }
    static void rotate(const double q[4], const double v[3], double o[3]) {        // Eigen's q * v: v + 2 w (u x v) + 2 u x (u x v)
        double uv[3] = {q[2] * v[2] - q[3] * v[1], q[3] * v[0] - q[1] * v[2], q[1] * v[1] - q[2] * v[0]};
        for (double& x : uv) x += x;
        const double uuv[3] = {q[2] * uv[2] - q[3] * uv[1], q[3] * uv[0] - q[1] * uv[2], q[1] * uv[1] - q[2] * uv[0]};
        for (int k = 0; k < 3; ++k) o[k] = v[k] + q[0] * uv[k] + uuv[k];
    }
    glio_opts opts_;
    int scan_match_cnt_;
    glio_ctx* ctx_ = nullptr;
    int poses_ = 0, map_points_ = 0;
    std::array<double, 7> last_pose_{{1, 0, 0, 0, 0, 0, 0}};
    std::vector<float> last_cloud_;
};

}  // namespace glio
