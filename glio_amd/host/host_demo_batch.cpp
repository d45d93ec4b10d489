// host_demo_batch.cpp -- the sharded batch stage as a C++ program: one process per GPU, every rank linearises the constraints
// of its keyframe range with the HIP kernels behind the C-ABI, ONE ncclAllReduce (RCCL; xGMI between the GPUs of a node) sums
// the block-banded [H | g | cost] buffers on the device, every rank runs the same banded solve.  No MPI: the ncclUniqueId
// travels through a file (rank 0 writes it, the others poll).
//
//   host_demo_batch problem.bin [iterations] [id_file]         RANK / WORLD_SIZE / LOCAL_RANK from the environment (default 0/1/0)
// problem.bin: int32 K, band, iterations_hint, 0 | int64 n | poses [K][7] f64 | ci [n] i32 | cj [n] i32 | cp [n][4] f32 |
//              norm_cent [n][6] f64 | score [n] f64      (all constraints; every rank keeps its own shard)
//              with header word 3 = 1 the file continues: odo [K][7] f64 | int32 search_range, n_dd | glio_gnss_frame | glio_dd_psr [n_dd]
//              and the program runs the full pose problem instead: 4 threshold rounds of the trust-region solve (BatchBackend::solveRounds)
//              with header word 3 = 2 there are NO constraints in the file (n = 0) but the keyframe clouds: after the poses follow
//              odo [K][7] f64 | int32 search_range, n_dd, max_points_per_frame, 0 | glio_gnss_frame | glio_dd_psr [n_dd] | per keyframe: int32 n, xyzi [n][4] f32
//              and the program runs optimizeBatchWithLandMark's LiDAR part end to end: batch association of this rank's keyframe pairs on the device
//              (BatchAssociationBackend), the stored interior constraints + the end keyframes re-searched every round (RoundsAssociation as the
//              `reassociate` hook), 4 threshold rounds of the trust-region solve.
// Build: g++ -std=c++14 -O2 -D__HIP_PLATFORM_AMD__ host_demo_batch.cpp -I../../include -I/opt/rocm/include -L../lib -lglio_hip
//        -L/opt/rocm/lib -lrccl -Wl,-rpath,'$ORIGIN/../lib' -Wl,-rpath,/opt/rocm/lib
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>

#include <rccl/rccl.h>

#include "glio_batch_backend.hpp"

template <typename T> static void rd(FILE* f, T* p, size_t n) { if (n && fread(p, sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } }
static int env_int(const char* k, int dflt) { const char* v = getenv(k); return v ? atoi(v) : dflt; }
#define NCCL_OK(x) do { const ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); return 3; } } while (0)

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: host_demo_batch problem.bin [iterations] [id_file]\n"); return 2; }
    const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), device = env_int("LOCAL_RANK", 0);
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    int32_t hdr[4];
    rd(f, hdr, 4);
    const int K = hdr[0], band = hdr[1];
    const int iterations = argc > 2 ? atoi(argv[2]) : hdr[2];
    int64_t n = 0;
    rd(f, &n, 1);
    std::vector<double> poses((size_t)K * 7), nc((size_t)n * 6), score((size_t)n);
    std::vector<int32_t> ci((size_t)n), cj((size_t)n);
    std::vector<float> cp((size_t)n * 4);
    rd(f, poses.data(), poses.size()); rd(f, ci.data(), ci.size()); rd(f, cj.data(), cj.size());
    rd(f, cp.data(), cp.size()); rd(f, nc.data(), nc.size()); rd(f, score.data(), score.size());
    const bool full = hdr[3] == 1 || hdr[3] == 2, assoc = hdr[3] == 2;
    std::vector<double> odo;
    std::vector<glio_dd_psr> dd;
    glio_gnss_frame frame;
    int32_t sr_ndd[2] = {0, 0};
    int max_pts = 0;
    std::vector<std::vector<float>> clouds;
    memset(&frame, 0, sizeof frame);
    if (full) {
        odo.resize((size_t)K * 7);
        rd(f, odo.data(), odo.size()); rd(f, sr_ndd, 2);
        int32_t more[2] = {0, 0};
        if (assoc) rd(f, more, 2);
        max_pts = more[0];
        rd(f, &frame, 1);
        dd.resize((size_t)sr_ndd[1]);
        rd(f, dd.data(), dd.size());
        if (assoc) {
            clouds.resize((size_t)K);
            for (int k = 0; k < K; ++k) { int32_t m = 0; rd(f, &m, 1); clouds[(size_t)k].resize((size_t)m * 4); rd(f, clouds[(size_t)k].data(), clouds[(size_t)k].size()); }
        }
    }
    fclose(f);
    // this rank's shard: constraints whose source keyframe is in [lo, hi) (they are sorted by (ci, cj))
    const std::pair<int, int> rg = glio::shardRange(K, rank, world, band);
    int64_t a0 = 0, a1 = n;
    while (a0 < n && ci[a0] < rg.first) ++a0;
    a1 = a0;
    while (a1 < n && ci[a1] < rg.second) ++a1;

    // ---- RCCL communicator
    ncclUniqueId id;
    const char* id_file = argc > 3 ? argv[3] : nullptr;
    if (rank == 0) {
        NCCL_OK(ncclGetUniqueId(&id));
        if (world > 1) {
            if (!id_file) { fprintf(stderr, "world > 1 needs an id_file\n"); return 2; }
            std::string tmp = std::string(id_file) + ".tmp";
            FILE* g = fopen(tmp.c_str(), "wb");
            fwrite(&id, sizeof id, 1, g); fclose(g);
            rename(tmp.c_str(), id_file);
        }
    } else {
        FILE* g = nullptr;
        for (int tries = 0; tries < 600 && !(g = fopen(id_file, "rb")); ++tries) std::this_thread::sleep_for(std::chrono::milliseconds(100));
        if (!g) { fprintf(stderr, "rank %d: no id file\n", rank); return 2; }
        rd(g, &id, 1); fclose(g);
    }
    try {
        // association mode: this rank's pairs (source keyframe in its range) and room for every point of every pair
        glio::PairList my_pairs;
        if (assoc) my_pairs = glio::pairShard(glio::pairList(K, sr_ndd[0]), K, rank, world, band);
        const int64_t assoc_cap = assoc ? (int64_t)(my_pairs.size() ? my_pairs.size() : 1) * max_pts : 0;
        glio::BatchBackend be(K, band, assoc ? assoc_cap : (a1 - a0 > 0 ? a1 - a0 : 1), device);     // creates the HIP context on `device` first
        ncclComm_t comm;
        NCCL_OK(ncclCommInitRank(&comm, world, id, rank));
        size_t reduced_bytes = 0; int n_reduce = 0;
        double t_reduce = 0;
        be.setAllReduce([&](double* dev, size_t count, void* stream) {
            const auto t0 = std::chrono::steady_clock::now();
            const ncclResult_t r = ncclAllReduce(dev, dev, count, ncclDouble, ncclSum, comm, (hipStream_t)stream);
            if (r != ncclSuccess) throw std::runtime_error(std::string("ncclAllReduce: ") + ncclGetErrorString(r));
            if (!full && glio_batch_synchronize(be.handle()) != GLIO_OK) throw std::runtime_error("synchronize");   // (timing only; the solve is stream-ordered)
            t_reduce += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            reduced_bytes += count * 8; ++n_reduce;
        });
        if (world > 1) be.setShard(rank, world);
        std::unique_ptr<glio::BatchAssociationBackend> ba;
        std::unique_ptr<glio::RoundsAssociation> ra;
        double assoc_ms = 0;
        if (assoc) {
            ba.reset(new glio::BatchAssociationBackend(K, max_pts, assoc_cap, device));
            for (int k = 0; k < K; ++k) ba->setFrame(k, clouds[(size_t)k].data(), (int)(clouds[(size_t)k].size() / 4));
            ra.reset(new glio::RoundsAssociation(be, *ba, my_pairs, K, sr_ndd[0]));
            const auto ta = std::chrono::steady_clock::now();
            ra->start(poses);                                           // the stored interior constraints + the first search of the ends
            assoc_ms += std::chrono::duration<double>(std::chrono::steady_clock::now() - ta).count() * 1e3;
        } else be.setConstraints(a1 - a0, ci.data() + a0, cj.data() + a0, cp.data() + 4 * a0, nc.data() + 6 * a0, score.data() + a0);
        std::vector<double> hist;
        std::vector<glio_summary> rounds;
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<double> sol;
        if (full) {
            sol = poses;
            std::function<void(const std::vector<double>&)> hook;
            if (assoc) hook = [&](const std::vector<double>& p) {
                const auto ta = std::chrono::steady_clock::now();
                (*ra)(p);
                assoc_ms += std::chrono::duration<double>(std::chrono::steady_clock::now() - ta).count() * 1e3;
            };
            rounds = be.solveRounds(sol, odo, sr_ndd[0], &frame, dd, glio::batchTrOpts(iterations), hook);
        }
        else sol = be.solve(poses, iterations, 1e-4, &hist);
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rank == 0) {
            printf("batch K %d band %d constraints %lld world %d iterations %d wall_ms %.3f allreduces %d allreduce_MB_each %.3f allreduce_ms_mean %.4f\n",
                   K, band, (long long)n, world, iterations, secs * 1e3, n_reduce, n_reduce ? reduced_bytes / 1e6 / n_reduce : 0.0,
                   n_reduce ? t_reduce * 1e3 / n_reduce : 0.0);
            if (assoc) printf("assoc pairs %zu constraints %lld association_ms %.3f\n", my_pairs.size(), (long long)ra->constraints(), assoc_ms);
            for (const glio_summary& r : rounds)
                printf("round iterations %d successful %d termination %d initial_cost %.17g final_cost %.17g\n", r.iterations, r.successful_steps, r.termination,
                       r.initial_cost, r.final_cost);
            printf("cost");
            for (double c : hist) printf(" %.17g", c);
            printf("\n");
            for (int k = 0; k < K; ++k)
                printf("kf %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", k, sol[7 * k], sol[7 * k + 1], sol[7 * k + 2], sol[7 * k + 3], sol[7 * k + 4], sol[7 * k + 5], sol[7 * k + 6]);
        }
        ncclCommDestroy(comm);
    } catch (const std::exception& e) {
        fprintf(stderr, "rank %d error: %s\n", rank, e.what());
        return 1;
    }
    return 0;
}
