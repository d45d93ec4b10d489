// glio_batch_backend.hpp -- C++14 host mirror of the sharded batch stage (the normal-equation loop of
// Estimator::optimizeBatchWithLandMark, reference GLIO/src/Estimator.cpp:3004-3076 constraints, :3275-3284 solve) on top of
// the C-ABI of libglio_hip.so.  No HIP headers: the reduced buffer is an opaque device pointer, the collective between
// "linearise my shard" and "solve the reduced system" is a hook the caller fills with ncclAllReduce (RCCL over xGMI) on that
// pointer and the batch stream -- see host_demo_batch.cpp for the complete multi-rank program.
//
//   rank r owns the constraints whose source keyframe lies in shardRange(K, r, world)   (same rule as glio_amd/batch.py)
//   damped Gauss-Newton path:  Hg_r = linearize(poses)  ->  allReduce(Hg)  ->  every rank: step(Hg, lambda)
//   trust-region path (solveTrustRegion / solveRounds): everything sharded, five small all-reduces per iteration on stream()
#pragma once
#include <cmath>
#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "glio_hip.h"

namespace glio {

// whole super-blocks of the solver's block cyclic reduction (6 keyframes; 12 for bands > 6): glio_batch_shard_range
inline std::pair<int, int> shardRange(int K, int rank, int world, int band = 6) {
    int32_t lo = 0, hi = K;
    if (glio_batch_shard_range(K, band, rank, world, &lo, &hi) != GLIO_OK) throw std::runtime_error("glio_batch_shard_range");
    return {lo, hi};
}

// The attitude-constraint pairs of optimizeBatch (Estimator.cpp:2831-2891) from the odometry keyframe poses ([K][7] = x y z
// qw qx qy qz): for keyframe i walk backward, then forward, taking (i, j, const_diff = q_i^-1 q_j) whenever the distance to the
// last taken keyframe exceeds 5 / search_range -- an INTEGER division there.  factor_count is reset only when it reaches
// search_range, so a short backward walk leaves its count (and its reference position) to the forward walk.
struct DeltaQPairs { std::vector<int32_t> i, j; std::vector<double> const_diff; };
inline DeltaQPairs deltaQPairs(const std::vector<double>& odo, int K, int search_range, int start_idx = 0) {
    DeltaQPairs out;
    const double thr = (double)(5 / search_range);
    for (int i = start_idx; i < K; ++i) {
        const double* pi = &odo[7 * (size_t)i];
        const double sgn = pi[3] < 0 ? -1.0 : 1.0;                       // unifyQuaternion(qi)
        const double qi[4] = {sgn * pi[3], sgn * pi[4], sgn * pi[5], sgn * pi[6]};
        const double n2 = qi[0] * qi[0] + qi[1] * qi[1] + qi[2] * qi[2] + qi[3] * qi[3];
        const double a[4] = {qi[0] / n2, -qi[1] / n2, -qi[2] / n2, -qi[3] / n2};
        double p_tmp[3] = {pi[0], pi[1], pi[2]};
        int count = 0;
        for (int dir = -1; dir <= 1; dir += 2) {
            for (int j = i; dir < 0 ? j >= start_idx : j < K; j += dir) {
                if (count == search_range) { count = 0; break; }
                if (j == i) continue;
                const double* pj = &odo[7 * (size_t)j];
                const double dx = p_tmp[0] - pj[0], dy = p_tmp[1] - pj[1], dz = p_tmp[2] - pj[2];
                if (std::sqrt(dx * dx + dy * dy + dz * dz) > thr) {
                    p_tmp[0] = pj[0]; p_tmp[1] = pj[1]; p_tmp[2] = pj[2];
                    const double* b = pj + 3;
                    out.i.push_back(i); out.j.push_back(j);
                    out.const_diff.push_back(a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3]);
                    out.const_diff.push_back(a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2]);
                    out.const_diff.push_back(a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3]);
                    out.const_diff.push_back(a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1]);
                    ++count;
                }
            }
        }
    }
    return out;
}

// ceres::Solver::Options of the batch solve (Estimator.cpp:3275-3281; yaml max_num_iter) + Ceres 1.14 defaults
inline glio_batch_tr_opts batchTrOpts(int max_iterations = 100) {
    glio_batch_tr_opts o;
    o.max_iterations = max_iterations; o.use_nonmonotonic_steps = 1; o.max_consecutive_nonmonotonic_steps = 5; o.jacobi_scaling = 1;
    o.initial_trust_region_radius = 1e4; o.max_trust_region_radius = 1e16; o.min_trust_region_radius = 1e-32;
    o.min_relative_decrease = 1e-3; o.function_tolerance = 1e-6; o.gradient_tolerance = 1e-10; o.parameter_tolerance = 1e-8;
    o.dogleg_type = GLIO_DOGLEG_SUBSPACE; o.reserved_ = 0;        // options.dogleg_type = SUBSPACE_DOGLEG, Estimator.cpp:3278
    return o;
}

class BatchBackend {
public:
    // all-reduce (sum, double) of `count` doubles at device pointer `dev` on HIP stream `stream`; empty = single rank
    using AllReduce = std::function<void(double* dev, size_t count, void* stream)>;

    BatchBackend(int K, int band, int64_t max_constraints, int device = 0) : K_(K), band_(band) {
        check(glio_batch_create(device, K, band, max_constraints, &h_), "glio_batch_create");
        check(glio_batch_hg_alloc_dev(h_, &hg_[0]), "glio_batch_hg_alloc_dev");
        check(glio_batch_hg_alloc_dev(h_, &hg_[1]), "glio_batch_hg_alloc_dev");
        check(glio_batch_get_stream(h_, &stream_), "glio_batch_get_stream");
    }
    ~BatchBackend() {
        if (h_) { glio_batch_hg_free_dev(h_, hg_[0]); glio_batch_hg_free_dev(h_, hg_[1]); glio_batch_destroy(h_); }
    }
    BatchBackend(const BatchBackend&) = delete;
    BatchBackend& operator=(const BatchBackend&) = delete;

    void setAllReduce(AllReduce f) { allreduce_ = std::move(f); }
    int64_t hgSize() const { return glio_batch_hg_size(K_, band_); }

    // this rank's constraints, sorted by (ci, cj): BinaryLidarPlaneNormFactor records (point in frame ci, plane normal and
    // centroid in frame cj, score)
    void setConstraints(int64_t n, const int32_t* ci, const int32_t* cj, const float* cp, const double* norm_cent, const double* score) {
        check(glio_batch_set_constraints(h_, n, ci, cj, cp, norm_cent, score), "glio_batch_set_constraints");
    }

    // linearise this rank's shard at `poses` ([K][7] = t, q), reduce over the ranks; returns the (global) cost
    double linearize(const std::vector<double>& poses, int buf) {
        check(glio_batch_linearize_dev(h_, poses.data(), hg_[buf]), "glio_batch_linearize_dev");
        if (allreduce_) { allreduce_(hg_[buf], (size_t)hgSize(), stream_); check(glio_batch_synchronize(h_), "glio_batch_synchronize"); }
        double cost = 0;
        check(glio_batch_read_dev(h_, hg_[buf], hgSize() - 1, 1, &cost), "glio_batch_read_dev");
        return cost;
    }
    // damped Gauss-Newton step from the reduced buffer `buf`
    std::vector<double> step(int buf, double lambda, const std::vector<double>& poses, double* model_decrease = nullptr) {
        std::vector<double> out(poses.size());
        check(glio_batch_step_dev(h_, hg_[buf], lambda, poses.data(), out.data(), model_decrease), "glio_batch_step_dev");
        return out;
    }
    // the loop of glio_amd/batch.py::lm_solve, in C++: accept when the cost drops (lambda / 3), otherwise lambda * 4
    std::vector<double> solve(std::vector<double> poses, int iterations, double lambda, std::vector<double>* history = nullptr) {
        int cur = 0;
        double cost = linearize(poses, cur);
        if (history) history->assign(1, cost);
        for (int it = 0; it < iterations; ++it) {
            const std::vector<double> cand = step(cur, lambda, poses);
            const double c2 = linearize(cand, 1 - cur);
            if (c2 < cost) { poses = cand; cost = c2; cur = 1 - cur; lambda = lambda / 3.0 > 1e-12 ? lambda / 3.0 : 1e-12; }
            else lambda *= 4.0;
            if (history) history->push_back(cost);
        }
        return poses;
    }
    // ---- the full pose problem: small factors replicated on every rank + the trust-region solve inside the library
    void setSmallFactors(const glio_gnss_frame* frame, const DeltaQPairs& dq, std::vector<glio_dd_psr>& dd, double dd_threshold) {
        for (glio_dd_psr& f : dd) f.threshold = dd_threshold;
        check(glio_batch_set_small_factors(h_, frame, (int)dq.i.size(), dq.i.data(), dq.j.data(), dq.const_diff.data(), (int)dd.size(), dd.data()),
              "glio_batch_set_small_factors");
    }
    // the LidarPoseFactorBatchRelativeAutoDiff blocks of sms_fusion_level == 0 (Estimator.cpp:2897-2955; the released default): keyframes (i[f], j[f]),
    // const_[7 f ..] = (tmpQuat w,x,y,z, tmpTrans) of :2901-2921.  Call BEFORE setSmallFactors, which builds the factor table.
    void setRelativePoseFactors(const std::vector<int32_t>& i, const std::vector<int32_t>& j, const std::vector<double>& const_) {
        check(glio_batch_set_relative_pose_factors(h_, (int)i.size(), i.empty() ? nullptr : i.data(), j.empty() ? nullptr : j.data(), const_.empty() ? nullptr : const_.data()),
              "glio_batch_set_relative_pose_factors");
    }
    // this backend is rank `rank` of `world` (call before setSmallFactors): it owns the keyframes of shardRange(); the all-reduce set
    // with setAllReduce is then called five times per trust-region iteration on small device buffers, ordered on stream()
    void setShard(int rank, int world) { check(glio_batch_set_shard(h_, rank, world), "glio_batch_set_shard"); }
    // the ImuFactor chain (Estimator.cpp:2990-3001): K - 1 pre-integrations, edges[k] between keyframes k and k + 1
    void setImu(const std::vector<glio_preint>& edges, double gravity) {
        check(glio_batch_set_imu(h_, (int)edges.size(), edges.empty() ? nullptr : edges.data(), gravity), "glio_batch_set_imu");
        have_imu_ = !edges.empty();
    }
    // ceres::Solve of the batch problem (Estimator.cpp:3275-3284), device resident; speed_bias [K][9] travels with the IMU chain
    glio_summary solveTrustRegion(std::vector<double>& poses, const glio_batch_tr_opts& opts, std::vector<double>* speed_bias = nullptr) {
        glio_summary s;
        if (have_imu_ && (!speed_bias || speed_bias->size() != (size_t)K_ * 9)) throw std::runtime_error("solveTrustRegion: the IMU chain needs speed_bias [K][9]");
        check(glio_batch_solve_tr2(h_, poses.data(), have_imu_ ? speed_bias->data() : nullptr, &opts, allreduce_ ? &BatchBackend::trampoline : nullptr, this, &s),
              "glio_batch_solve_tr2");
        return s;
    }
    // the outer loop of optimizeBatch (Estimator.cpp:2764-3410): iteration_num = 4 rounds with DDpsr_threshold {1e9, 10, 8, 6};
    // `reassociate` (may be empty) re-searches the LiDAR correspondences at the current poses and calls setConstraints
    std::vector<glio_summary> solveRounds(std::vector<double>& poses, const std::vector<double>& odo, int search_range, const glio_gnss_frame* frame,
                                          std::vector<glio_dd_psr>& dd, const glio_batch_tr_opts& opts,
                                          const std::function<void(const std::vector<double>&)>& reassociate = {}, std::vector<double>* speed_bias = nullptr) {
        static const double thresholds[4] = {1000000000, 10, 8, 6};
        const DeltaQPairs dq = deltaQPairs(odo, K_, search_range);
        setSmallFactors(frame, dq, dd, thresholds[0]);
        std::vector<glio_summary> out;
        for (double thr : thresholds) {
            if (reassociate) reassociate(poses);
            check(glio_batch_set_dd_threshold(h_, thr), "glio_batch_set_dd_threshold");
            out.push_back(solveTrustRegion(poses, opts, speed_bias));
        }
        return out;
    }
    glio_batch* handle() { return h_; }
    void* stream() { return stream_; }

private:
    static void trampoline(double* dev, int64_t count, void* stream, void* user) {
        static_cast<BatchBackend*>(user)->allreduce_(dev, (size_t)count, stream);
    }
    static void check(int rc, const char* what) {
        if (rc != GLIO_OK) throw std::runtime_error(std::string(what) + ": " + glio_last_error());
    }
    int K_, band_;
    bool have_imu_ = false;
    glio_batch* h_ = nullptr;
    double* hg_[2] = {nullptr, nullptr};
    void* stream_ = nullptr;
    AllReduce allreduce_;
};

}  // namespace glio
