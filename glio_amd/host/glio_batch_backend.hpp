// glio_batch_backend.hpp -- C++14 host mirror of the sharded batch stage (the normal-equation loop of
// Estimator::optimizeBatchWithLandMark, reference GLIO/src/Estimator.cpp:3004-3076 constraints, :3275-3284 solve) on top of
// the C-ABI of libglio_hip.so.  No HIP headers: the reduced buffer is an opaque device pointer, the collective between
// "linearise my shard" and "solve the reduced system" is a hook the caller fills with ncclAllReduce (RCCL over xGMI) on that
// pointer and the batch stream -- see host_demo_batch.cpp for the complete multi-rank program.
//
//   rank r owns the constraints whose source keyframe lies in shardRange(K, r, world)   (same rule as glio_amd/batch.py)
//   iteration:  Hg_r = linearize(poses)  ->  allReduce(Hg)  ->  every rank: step(Hg, lambda)  (identical numbers on every rank)
#pragma once
#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "glio_hip.h"

namespace glio {

inline std::pair<int, int> shardRange(int K, int rank, int world) {
    const int base = K / world, rem = K % world;
    const int lo = rank * base + (rank < rem ? rank : rem);
    return {lo, lo + base + (rank < rem ? 1 : 0)};
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
    glio_batch* handle() { return h_; }
    void* stream() { return stream_; }

private:
    static void check(int rc, const char* what) {
        if (rc != GLIO_OK) throw std::runtime_error(std::string(what) + ": " + glio_last_error());
    }
    int K_, band_;
    glio_batch* h_ = nullptr;
    double* hg_[2] = {nullptr, nullptr};
    void* stream_ = nullptr;
    AllReduce allreduce_;
};

}  // namespace glio
