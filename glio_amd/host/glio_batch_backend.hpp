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
#include <algorithm>
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

// ================================================================================================
// Batch association from C++ (glio_bassoc_* of the C-ABI): findGlobalCorrespondingSurfFeaturesAdd_Batch / _Batch (Estimator.cpp:3808-3892, 3711-3806),
// the search-window rule of optimizeBatchWithLandMark (:3009-3017), batchFeatureAssociation (:3413-3432) and the two selection rules (:4057-4116,
// 3994-4055).  Same rules as glio_amd/batch.py (search_window, pair_list, pair_shard, batch_selection_draws); tests/test_host_cpp.py holds the two
// hosts to the same records.
// ================================================================================================
// first keyframe of the 2 search_range + 1 window searched for keyframe idx: centred in the interior, clamped at the ends of the batch (:3009-3017)
inline int searchWindow(int idx, int size, int search_range, int start_idx = 0) {
    if (idx >= search_range + start_idx && idx < size - 1 - search_range) return idx - search_range;
    if (idx < search_range + start_idx) return start_idx;
    return size - 2 * search_range - 1;
}
struct PairList { std::vector<int32_t> ci, cj; size_t size() const { return ci.size(); } };
// every (idx, search_idx) pair of a batch of K keyframes, ci-major (the loop of :3004-3076)
inline PairList pairList(int K, int search_range) {
    PairList p;
    for (int idx = 0; idx < K; ++idx) {
        const int s0 = searchWindow(idx, K, search_range);
        for (int j = s0; j <= s0 + 2 * search_range; ++j)
            if (j != idx && j >= 0 && j < K) { p.ci.push_back(idx); p.cj.push_back(j); }
    }
    return p;
}
// the pairs a rank owns: those whose SOURCE keyframe lies in its keyframe range (the ownership rule of the constraints)
inline PairList pairShard(const PairList& all, int K, int rank, int world, int band) {
    const std::pair<int, int> rg = shardRange(K, rank, world, band);
    PairList p;
    for (size_t q = 0; q < all.size(); ++q)
        if (all.ci[q] >= rg.first && all.ci[q] < rg.second) { p.ci.push_back(all.ci[q]); p.cj.push_back(all.cj[q]); }
    return p;
}
// globalFeatureSelectionAdd_Batch keeps, of a pair's `count` records, all when count <= res_num, else the first res_num of a shuffle of 0 .. count - 2
// (geneRandArrayNoRepeat(0, count - 1, n): n = high - low values, the LAST record is never drawn; random_generator.hpp:79-93).  `rand_below(n)` returns
// a uniform integer in [0, n): the reference seeds from std::random_device, so the generator is the caller's.  Returns false = keep all.
template <typename RandBelow>
inline bool batchSelectionDraws(int64_t count, int res_num, RandBelow&& rand_below, std::vector<int64_t>& out) {
    out.clear();
    if (count <= res_num) return false;
    // partial Fisher-Yates over 0 .. count - 2 WITHOUT the index array (a pair holds ~60 000 records, 25 are drawn: the array cost 0.2 ms per keyframe):
    // `moved` holds the positions whose content differs from their index
    std::vector<std::pair<int64_t, int64_t>> moved;
    auto at = [&](int64_t k) { for (const auto& m : moved) if (m.first == k) return m.second; return k; };
    auto put = [&](int64_t k, int64_t v) { for (auto& m : moved) if (m.first == k) { m.second = v; return; } moved.push_back({k, v}); };
    for (int64_t i = 0; i < res_num && i < count - 1; ++i) {            // the first res_num of a uniform shuffle
        const int64_t j = i + (int64_t)rand_below((uint64_t)(count - 1 - i));
        const int64_t vi = at(i), vj = at(j);
        put(j, vi);
        out.push_back(vj);
    }
    return true;
}

class BatchAssociationBackend {
public:
    BatchAssociationBackend(int K, int max_points_per_frame, int64_t max_constraints, int device = 0) : K_(K), cap_(max_constraints) {
        check(glio_bassoc_create(device, K, max_points_per_frame, max_constraints, &h_), "glio_bassoc_create");
    }
    ~BatchAssociationBackend() { if (h_) glio_bassoc_destroy(h_); }
    BatchAssociationBackend(const BatchAssociationBackend&) = delete;
    BatchAssociationBackend& operator=(const BatchAssociationBackend&) = delete;

    // surf_frames[k]: PointXYZI as 4 floats, keyframe-local (the batch factor applies no LiDAR-IMU extrinsic, quirk Q10: body-frame clouds)
    void setFrame(int k, const float* xyzi, int n) { check(glio_bassoc_set_frame(h_, k, xyzi, n), "glio_bassoc_set_frame"); }
    // the same straight from a pcl::PointCloud<pcl::PointXYZI>'s points.data(): stride 32, intensity at byte 16
    void setFrame(int k, const void* points, int n, int stride_bytes, int intensity_offset) {
        check(glio_bassoc_set_frame_strided(h_, k, points, n, stride_bytes, intensity_offset), "glio_bassoc_set_frame_strided");
    }
    // the same from the scan resident in window slot `slot` of a sliding-window context (device copy, minus the LiDAR offset)
    void setFrameFromScan(int k, glio_ctx* ctx, int slot, const float lidar_offset[3]) {
        check(glio_bassoc_set_frame_from_scan(h_, k, ctx, slot, lidar_offset), "glio_bassoc_set_frame_from_scan");
    }
    // poses [K][7] = t, q (pose_info_keyframe).  run: the records of `pairs` replace what the object holds; runAppend: they are added behind it
    std::vector<int64_t> run(const std::vector<double>& poses, const PairList& pairs) { return runImpl(poses, pairs, false); }
    std::vector<int64_t> runAppend(const std::vector<double>& poses, const PairList& pairs) { return runImpl(poses, pairs, true); }
    void prepareAsync(const PairList& pairs) {
        check(glio_bassoc_prepare_async(h_, (int)pairs.size(), pairs.ci.data(), pairs.cj.data()), "glio_bassoc_prepare_async");
    }
    void runAppendAsync(const std::vector<double>& poses, const PairList& pairs) {
        needPoses(poses);
        check(glio_bassoc_run_append_async(h_, poses.data(), (int)pairs.size(), pairs.ci.data(), pairs.cj.data()), "glio_bassoc_run_append_async");
        pending_ = (int)pairs.size();
    }
    // the selection of the run in flight on its own stream (raws: res_num numbers per pair, drawn before the counts exist); finish() then returns the counts FOUND
    void selectTailDrawsAsync(int res_num, const std::vector<uint64_t>& raws) {
        check(glio_bassoc_select_tail_draws_async(h_, res_num, raws.data()), "glio_bassoc_select_tail_draws_async");
    }
    std::vector<int64_t> finish() {
        std::vector<int64_t> cnt((size_t)(pending_ > 0 ? pending_ : 1));
        check(glio_bassoc_finish(h_, cnt.data(), &total_), "glio_bassoc_finish");
        cnt.resize((size_t)(pending_ > 0 ? pending_ : 0)); pending_ = 0;
        return cnt;
    }
    void reset() { check(glio_bassoc_reset(h_), "glio_bassoc_reset"); total_ = 0; }
    // forget the records behind `total` (the outer rounds re-search the end keyframes: their regions are rewritten, Estimator.cpp:3018-3030)
    void truncate(int64_t total) { check(glio_bassoc_select_range(h_, total, 0, nullptr, total_), "glio_bassoc_select_range"); total_ = total; }
    // keep src (absolute record indices >= first) of the tail [first, total()): globalFeatureSelection*_Batch
    void selectRange(int64_t first, const std::vector<int64_t>& src) {
        check(glio_bassoc_select_range(h_, first, (int64_t)src.size(), src.empty() ? nullptr : src.data(), total_), "glio_bassoc_select_range");
        total_ = first + (int64_t)src.size();
    }
    int64_t total() const { return total_; }
    int64_t capacity() const { return cap_; }
    void read(int64_t first, int64_t n, float* cp, double* norm_cent, double* score) { check(glio_bassoc_read(h_, first, n, cp, norm_cent, score), "glio_bassoc_read"); }
    // the device-resident records to a batch stage: pair p owns [offset[p], offset[p] + count[p]) (offset empty: the pairs follow each other from 0)
    void feed(BatchBackend& stage, const PairList& pairs, const std::vector<int64_t>& count, const std::vector<int64_t>& offset = {}, const std::vector<uint8_t>& changed = {}) {
        const float* cp; const double* nc; const double* sc;
        check(glio_bassoc_results_dev(h_, &cp, &nc, &sc), "glio_bassoc_results_dev");
        check(glio_batch_update_constraints_pairs_at_dev(stage.handle(), (int)pairs.size(), pairs.ci.data(), pairs.cj.data(), count.data(), offset.empty() ? nullptr : offset.data(),
                                                         cp, nc, sc, changed.empty() ? nullptr : changed.data()), "glio_batch_update_constraints_pairs_at_dev");
    }
    // batchFeatureAssociation() (Estimator.cpp:3413-3432): false when the stream is still too short, else the 2 search_range pairs of keyframe size - search_range - 1
    static bool keyframePairs(int size, int search_range, PairList& out) {
        out.ci.clear(); out.cj.clear();
        const int idx = size - search_range - 1;
        if (size < 2 * search_range || idx < search_range) return false;
        for (int j = idx - search_range; j <= idx + search_range; ++j) if (j != idx) { out.ci.push_back(idx); out.cj.push_back(j); }
        return true;
    }
    glio_bassoc* handle() { return h_; }

private:
    void needPoses(const std::vector<double>& poses) const { if (poses.size() != (size_t)K_ * 7) throw std::runtime_error("batch association: poses must be [K][7]"); }
    std::vector<int64_t> runImpl(const std::vector<double>& poses, const PairList& pairs, bool append) {
        needPoses(poses);
        std::vector<int64_t> cnt(pairs.size() ? pairs.size() : 1);
        check((append ? glio_bassoc_run_append : glio_bassoc_run)(h_, poses.data(), (int)pairs.size(), pairs.ci.data(), pairs.cj.data(), cnt.data(), &total_),
              append ? "glio_bassoc_run_append" : "glio_bassoc_run");
        cnt.resize(pairs.size());
        return cnt;
    }
    static void check(int rc, const char* what) {
        if (rc != GLIO_OK) throw std::runtime_error(std::string(what) + ": " + glio_last_error());
    }
    int K_;
    int64_t cap_, total_ = 0;
    int pending_ = 0;
    glio_bassoc* h_ = nullptr;
};

// The association schedule of optimizeBatchWithLandMark's rounds (Estimator.cpp:3004-3076): the INTERIOR keyframes use the constraints stored when they
// left the sliding window (gl_vec_surf_*: here associated once, at the poses given to start()), the first / last search_range keyframes are re-searched in
// EVERY round at the current poses (findGlobalCorrespondingSurfFeatures_Batch, :3018-3030).  One resident association object: its record arrays are laid
// out [interior | front ends | back ends]; a round truncates them to the interior and appends the two end runs again -- nothing is copied, the stage is
// told every pair's record range and which pairs changed.  operator() is the `reassociate` hook of BatchBackend::solveRounds.  (The random
// globalFeatureSelection_Batch draw is left to the caller, as everywhere.)
class RoundsAssociation {
public:
    // `pairs` = this rank's pairs, ci-major (pairList or pairShard); K = keyframes of the batch
    RoundsAssociation(BatchBackend& stage, BatchAssociationBackend& ba, const PairList& pairs, int K, int search_range) : stage_(stage), ba_(ba) {
        for (size_t q = 0; q < pairs.size(); ++q) {
            PairList& dst = pairs.ci[q] < search_range ? front_ : pairs.ci[q] > K - 1 - search_range ? back_ : inner_;
            dst.ci.push_back(pairs.ci[q]); dst.cj.push_back(pairs.cj[q]);
        }
    }
    void start(const std::vector<double>& poses) {
        cnt_inner_ = ba_.run(poses, inner_);
        inner_total_ = ba_.total();
        ends(poses);
        feed(false);
    }
    void operator()(const std::vector<double>& poses) {
        ba_.truncate(inner_total_);
        ends(poses);
        feed(true);
        ++runs_;
    }
    int64_t constraints() const { return ba_.total(); }
    int rounds() const { return runs_; }

private:
    void ends(const std::vector<double>& poses) {
        cnt_front_ = ba_.runAppend(poses, front_);
        cnt_back_ = ba_.runAppend(poses, back_);
    }
    void feed(bool only_ends) {
        PairList all;
        std::vector<int64_t> cnt, off;
        std::vector<uint8_t> chg;
        int64_t o_front = inner_total_, o_inner = 0, o_back = inner_total_;
        for (int64_t c : cnt_front_) o_back += c;
        const PairList* parts[3] = {&front_, &inner_, &back_};
        const std::vector<int64_t>* counts[3] = {&cnt_front_, &cnt_inner_, &cnt_back_};
        int64_t base[3] = {o_front, o_inner, o_back};
        for (int w = 0; w < 3; ++w)
            for (size_t q = 0; q < parts[w]->size(); ++q) {
                all.ci.push_back(parts[w]->ci[q]); all.cj.push_back(parts[w]->cj[q]);
                cnt.push_back((*counts[w])[q]); off.push_back(base[w]); base[w] += (*counts[w])[q];
                chg.push_back((uint8_t)(!only_ends || w != 1));
            }
        ba_.feed(stage_, all, cnt, off, only_ends ? chg : std::vector<uint8_t>());
    }
    BatchBackend& stage_;
    BatchAssociationBackend& ba_;
    PairList front_, inner_, back_;
    std::vector<int64_t> cnt_front_, cnt_inner_, cnt_back_;
    int64_t inner_total_ = 0;
    int runs_ = 0;
};

// batchFeatureAssociation() per keyframe (Estimator.cpp:3413-3432; it ENDS every optimizeSlidingWindowWithLandMark, :2733): once the stream holds 2 search_range
// keyframes, keyframe idx = size - search_range - 1 is matched against its 2 search_range neighbours at the current poses and the records are ADDED to
// gl_vec_surf_* -- here appended behind what the association object holds; globalFeatureSelectionAdd_Batch (:4057-4116) then keeps batch_feature_res_num
// records per pair.  enqueue() returns at once (own stream: the searches overlap the marginalization of the same keyframe), finish() waits, selects, books.
class KeyframeBatchAssociation {
public:
    explicit KeyframeBatchAssociation(BatchAssociationBackend& ba, int search_range = 6, int feature_res_num = -1) : ba_(ba), sr_(search_range), res_num_(feature_res_num) {}
    // poses [K][7] = t, q of every keyframe slot of the association object (pose_info_keyframe); size = keyframes in the stream so far
    // optional, before the poses exist (i.e. before the solve of the same keyframe call): the pairs of `size` keyframes are formed and their search frames'
    // tables cleared on the device; enqueue(size, poses) then starts with the transform of the clouds
    void prepare(int size) {
        PairList pl;
        if (BatchAssociationBackend::keyframePairs(size, sr_, pl)) ba_.prepareAsync(pl);
    }
    int enqueue(int size, const std::vector<double>& poses) {
        have_ = BatchAssociationBackend::keyframePairs(size, sr_, cur_);
        on_stream_ = false;
        if (!have_) return 0;
        first_ = ba_.total();
        ba_.runAppendAsync(poses, cur_);
        return (int)cur_.size();
    }
    // the same with the selection enqueued behind the searches on the association's stream: rand_u64() returns the caller's raw 64-bit draws (res_num per
    // pair, made now: a draw does not need the counts, only its reduction does); finish() then has nothing to select and nothing to upload
    template <typename RandU64>
    int enqueueWithDraws(int size, const std::vector<double>& poses, RandU64&& rand_u64) {
        const int n = enqueue(size, poses);
        if (n == 0 || res_num_ < 1 || res_num_ > 64) return n;
        std::vector<uint64_t> raws((size_t)n * (size_t)res_num_);
        for (uint64_t& r : raws) r = rand_u64();
        ba_.selectTailDrawsAsync(res_num_, raws);
        on_stream_ = true;
        return n;
    }
    // rand_below(n): uniform integer in [0, n) -- the reference seeds from std::random_device, the generator is the caller's.  Returns the pair counts FOUND
    // (before the selection); `counts` books what was kept.
    template <typename RandBelow>
    std::vector<int64_t> finish(RandBelow&& rand_below) {
        if (!have_) return {};
        have_ = false;
        const std::vector<int64_t> found = ba_.finish();
        std::vector<int64_t> kept = found;
        if (on_stream_) {          // the device applied the rule: all of a pair's records when it has at most res_num, else res_num (never the last record)
            for (size_t p = 0; p < found.size(); ++p) kept[p] = found[p] <= res_num_ ? found[p] : std::min<int64_t>(res_num_, found[p] - 1);
        } else if (res_num_ >= 0) {
            std::vector<int64_t> src, draw;
            int64_t off = first_;
            bool any = false;
            for (size_t p = 0; p < found.size(); ++p) {
                if (batchSelectionDraws(found[p], res_num_, rand_below, draw)) { any = true; for (int64_t d : draw) src.push_back(off + d); kept[p] = (int64_t)draw.size(); }
                else for (int64_t d = 0; d < found[p]; ++d) src.push_back(off + d);
                off += found[p];
            }
            if (any) ba_.selectRange(first_, src);
        }
        for (size_t p = 0; p < cur_.size(); ++p) { pairs.ci.push_back(cur_.ci[p]); pairs.cj.push_back(cur_.cj[p]); counts.push_back(kept[p]); }
        return found;
    }
    std::vector<int64_t> finish() { return finish([](uint64_t) -> uint64_t { return 0; }); }      // (no selection configured: the generator is never called)
    PairList pairs;                    // every pair booked so far, in call order = the order of the records
    std::vector<int64_t> counts;

private:
    BatchAssociationBackend& ba_;
    int sr_, res_num_;
    PairList cur_;
    int64_t first_ = 0;
    bool have_ = false, on_stream_ = false;
};

// ================================================================================================
// Which GNSS epochs enter the batch problem, between which keyframes, and which satellites form a double-difference factor: the host rules of
// optimizeBatchWithLandMark (Estimator.cpp:3086-3272) and prepare{GPS,BDS,GLO,GAL}DDPsrData (:1702-1860).  Pure host arithmetic; the factors
// themselves are glio_dd_psr records evaluated on the device.  glio_amd/batch.py::select_batch_gnss_epochs / dd_group are the Python twins;
// tests/test_host_logic.py holds the two to each other on random streams.
// ================================================================================================
struct GnssEpochSlot { int epoch, left_key, right_key; double ts_ratio; };
// obs_local_ts[i] = time of epoch i minus timeshift_IMUtoGNSS (:3093); keyframe_time[k] as the reference holds it (pose index i is 1-based: its time is
// keyframe_time[i - 1]); first_idx = keyframe_idx[0]; n_poses = pose_info_keyframe_batch->points.size(); trans = gl_tmpTrans [K][3] (0-based keys).
// Per epoch: skipped outside [keyframe_time.front(), keyframe_time.back()] (:3100); lower / upper = the pose strictly before / after the epoch that is
// closest in time, searched over [first_idx, n_poses) with strict "<" on the distance (getGlobalLowerUpperIdx, :1635-1663: the FIRST of equally close
// ones wins; -1 / 10000000 when none); skipped unless both lie in [0, n_poses) (:3104-3105); ts_ratio = (t_upper - t) / (t_upper - t_lower) (:3109);
// keys = idx - 1; skipped when the right keyframe lies less than 1 m from the right keyframe of the last ACCEPTED epoch (:3121-3126; the reference
// point starts at the origin, so an epoch whose right keyframe is within 1 m of (0, 0, 0) is skipped too).
inline std::vector<GnssEpochSlot> selectBatchGnssEpochs(const std::vector<double>& obs_local_ts, const std::vector<double>& keyframe_time, int first_idx, int n_poses,
                                                        const std::vector<double>& trans) {
    std::vector<GnssEpochSlot> out;
    // pose index i reads keyframe_time[i - 1] and trans[i - 1]: first_idx = 0 would index element -1 (the reference's keyframe_idx starts at 1), and the search
    // runs up to pose n_poses - 1.  Refused here rather than read out of bounds (advisor finding of round 5); select_batch_gnss_epochs raises the same way.
    if (first_idx < 1 || n_poses < first_idx || (size_t)(n_poses > 0 ? n_poses - 1 : 0) > keyframe_time.size() || 3 * (size_t)(n_poses > 0 ? n_poses - 1 : 0) > trans.size())
        throw std::invalid_argument("selectBatchGnssEpochs: need first_idx >= 1 and n_poses - 1 <= keyframe_time.size(), trans.size() / 3");
    if (keyframe_time.empty()) return out;
    double padd[3] = {0.0, 0.0, 0.0};
    for (size_t e = 0; e < obs_local_ts.size(); ++e) {
        const double T = obs_local_ts[e];
        if (T > keyframe_time.back() || T < keyframe_time.front()) continue;
        int lower = -1, upper = 10000000;
        double diff = 10000000;
        for (int i = first_idx; i < n_poses; ++i) { const double t = keyframe_time[(size_t)i - 1], d = std::fabs(t - T); if (d < diff && t < T) { lower = i; diff = d; } }
        diff = 10000000;
        for (int i = first_idx; i < n_poses; ++i) { const double t = keyframe_time[(size_t)i - 1], d = std::fabs(t - T); if (d < diff && t > T) { upper = i; diff = d; } }
        if (lower < 0 || lower >= n_poses || upper < 0 || upper >= n_poses) continue;
        const double tl = keyframe_time[(size_t)lower - 1], tu = keyframe_time[(size_t)upper - 1];
        const int lk = lower - 1, rk = upper - 1;
        const double* pj = &trans[3 * (size_t)rk];
        const double dx = pj[0] - padd[0], dy = pj[1] - padd[1], dz = pj[2] - padd[2];
        if (std::sqrt(dx * dx + dy * dy + dz * dz) < 1.0) continue;
        padd[0] = pj[0]; padd[1] = pj[1]; padd[2] = pj[2];
        out.push_back({(int)e, lk, rk, (tu - T) / (tu - tl)});
    }
    return out;
}
// constellation of a PRN as gnss_tools.h:1116-1168 numbers them: 0 GPS (<= 32 or 84), 1 BeiDou (87..121), 2 GLONASS (33..56), 3 Galileo (57..86), -1 none
inline int prnSystem(int prn) {
    if (prn <= 32 || prn == 84) return 0;
    if (prn >= 87 && prn <= 121) return 1;
    if (prn > 32 && prn <= 56) return 2;
    if (prn > 56 && prn < 87) return 3;
    return -1;
}
// prepare<SYS>DDPsrData (:1702-1860): the (rover, station) observation pairs of one constellation -- rover order outside, station order inside, rover
// pseudorange > 1000 -- and the master = the LAST pair whose |elevation| exceeds the running maximum, which the reference updates with the SIGNED
// elevation (maxEle = ele, not fabs(ele): replicated).  master stays -1 for an empty group; the caller adds the factor when the group holds more than 2
// pairs (:3202).  The reference walks the systems in the order GPS, BDS, GLO, GAL (:3198-3271).
struct DdGroup { std::vector<int> user, ref; int master = -1; };
inline DdGroup ddGroup(int system, const std::vector<int>& user_prn, const std::vector<double>& user_psr, const std::vector<double>& user_ele,
                       const std::vector<int>& ref_prn) {
    DdGroup g;
    for (size_t i = 0; i < user_prn.size(); ++i)
        for (size_t j = 0; j < ref_prn.size(); ++j)
            if (user_prn[i] == ref_prn[j] && prnSystem(user_prn[i]) == system && user_psr[i] > 1000) { g.user.push_back((int)i); g.ref.push_back((int)j); }
    double max_ele = 0;
    for (size_t m = 0; m < g.user.size(); ++m) {
        const double ele = user_ele[(size_t)g.user[m]];
        if (std::fabs(ele) > max_ele) { max_ele = ele; g.master = (int)m; }
    }
    return g;
}

}  // namespace glio
