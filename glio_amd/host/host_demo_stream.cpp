// host_demo_stream.cpp -- the MOVING-STREAM keyframe cycle of optimizeSlidingWindowWithLandMark() (Estimator.cpp:2046-2736, called per keyframe from
// saveKeyFramesAndFactors, :4269) driven from C++ through glio_backend.hpp, and TIMED: slide the window + take the new keyframe's scan, update the
// 50-keyframe local map on the device, associate all W slots (enqueued), fill the IMU / GNSS factor tables while the GPU searches, solve, marginalize the
// oldest keyframe and keep the result as the next prior, and -- what ends the reference function, :2733 -- batchFeatureAssociation() (:3413-3432): the keyframe
// search_range back is matched against its 2 search_range neighbours at the solved poses (12 hash builds + 12 pair searches on the device, enqueued
// right after the solve on the association's own stream so that they overlap the marginalization), globalFeatureSelectionAdd_Batch keeps 25 per pair.  Input: a flat binary stream file written by glio_amd/host/window_io.py (write_stream); output:
// one JSON line with the per-stage host times -- what bench.py reports as keyframe_pipeline_cpp next to the Python driver's figure.
// Build: g++ -std=c++17 -O2 host_demo_stream.cpp -I../../include -L../lib -lglio_hip -Wl,-rpath,'$ORIGIN/../lib'
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include <random>

#include "glio_backend.hpp"
#include "glio_batch_backend.hpp"

template <typename T> static void rd(FILE* f, T* p, size_t n) { if (n && fread(p, sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } }
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

extern "C" int glio_debug_arrow_stamps(glio_ctx* c, long long* out320);      // debug export of the library (scripts/chain_step_time.py reads the same)
int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: host_demo_stream stream.bin [device] [search_range] [defer] [res=N] [draws=FILE] [timed=N]\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    const int device = argc > 2 ? atoi(argv[2]) : 0;
    glio_opts opts;
    rd(f, &opts, 1);
    int32_t hdr[4];            // n_keyframes (timed; one more runs first as the warm-up), points per scan, local-map width, reserved
    rd(f, hdr, 4);
    float leaf, tlb[3];
    rd(f, &leaf, 1); rd(f, tlb, 3);
    const int W = opts.window, NK = hdr[0], pts = hdr[1], total_kf = W + NK;
    std::vector<std::vector<float>> scans(total_kf, std::vector<float>((size_t)pts * 4));
    std::vector<double> gtq((size_t)total_kf * 4), gtt((size_t)total_kf * 3);
    for (int j = 0; j < total_kf; ++j) rd(f, scans[j].data(), scans[j].size());
    rd(f, gtq.data(), gtq.size()); rd(f, gtt.data(), gtt.size());
    struct Kf { std::vector<double> trans, quat, sb, ddt; std::vector<glio_preint> pre; glio_gnss_frame frame; std::vector<glio_dd_psr> dd; std::vector<glio_doppler> dop; };
    std::vector<Kf> kf(NK + 1);
    for (Kf& k : kf) {
        int32_t n[4];          // n_ddt, n_preint, n_dd, n_dop
        rd(f, n, 4);
        k.trans.resize(3 * W); k.quat.resize(4 * W); k.sb.resize(9 * W); k.ddt.resize(n[0]); k.pre.resize(n[1]); k.dd.resize(n[2]); k.dop.resize(n[3]);
        rd(f, k.trans.data(), k.trans.size()); rd(f, k.quat.data(), k.quat.size()); rd(f, k.sb.data(), k.sb.size()); rd(f, k.ddt.data(), k.ddt.size());
        rd(f, k.pre.data(), k.pre.size()); rd(f, &k.frame, 1); rd(f, k.dd.data(), k.dd.size()); rd(f, k.dop.data(), k.dop.size());
    }
    fclose(f);
    try {
        glio::SlidingWindowBackend be(opts, device);
        be.configureLocalMap(hdr[2], leaf, pts);
        // the map before the first timed keyframe: the window's own earlier keyframes (body-frame clouds); slots 1..W-1 hold their scans
        std::vector<float> body((size_t)pts * 4);
        for (int j = 0; j < W - 1; ++j) {
            body = scans[j];
            for (int i = 0; i < pts; ++i) for (int c = 0; c < 3; ++c) body[4 * (size_t)i + c] -= tlb[c];
            glio::check(glio_localmap_push(be.ctx(), body.data(), pts, &gtq[4 * j], &gtt[3 * j]), "glio_localmap_push");
        }
        for (int s = 0; s < W - 1; ++s) be.setScan(s + 1, scans[s].data(), pts);
        { glio_prior none; memset(&none, 0, sizeof none); be.setMarginalizationPrior(&none); }      // the first window has no prior
        // batchFeatureAssociation: every keyframe of the stream keeps its cloud (body frame) resident; search_range 6, batch_feature_res_num 25 (config_urban_hk.yaml:64,102)
        const int SR = argc > 3 ? atoi(argv[3]) : 6, RES = hdr[3] > 0 ? hdr[3] : 25;
        // defer = 1: the batch association of keyframe j is only ENQUEUED inside call j (own stream) and collected inside call j + 1, right before that call's
        // own enqueue -- its searches then run beside the next keyframe's solve (one CU busy) instead of beside nothing.  The records reach gl_vec_surf_* one
        // keyframe later than in the reference, which only the batch thread could notice; the default (0) returns from the call with them in place.
        const bool defer = argc > 4 && atoi(argv[4]) == 1;
        // 2 = the batch association enqueued AFTER the marginalization (nothing of the call overlaps it) -- A/B of the default order
        const bool after_marg = argc > 4 && atoi(argv[4]) == 2;
        glio::BatchAssociationBackend ba(total_kf, pts, (int64_t)(NK + 2) * 2 * SR * pts, device);
        glio::KeyframeBatchAssociation kba(ba, SR, RES);
        // named arguments behind the positional ones: res=N (feature_res_num of featureSelection, Estimator.cpp:2223; 0 = no selection, the BASELINE workloads),
        // draws=FILE (a table of 64-bit numbers: rand_below(n) = table[k++] mod n -- the generator a test shares with the Python host), timed=N (only the last N
        // keyframes enter the averages: the released configuration fills its 50-keyframe local map first)
        int feature_res = 0, timed_last = NK;
        bool per_slot = false;
        // stream_draws=1 (default): the selection's raw draws travel with the enqueue and globalFeatureSelectionAdd_Batch runs on the association's stream behind the
        // searches (glio_bassoc_select_tail_draws_async).  stream_draws=0: the host waits for the pair counts, draws, uploads the kept indices and does not wait for
        // the gather.  (With twelve hash builds in the association's chain the host form measured faster, 1.37 vs 1.40 ms per keyframe; with the tables in the
        // keyframes' own frames the chain is 165 us shorter, the host round trip at its end counts, and the on-stream form wins: 1.22 vs 1.27 ms, 10 runs each.)
        bool host_draws = false, prepare_early = true;
        // ahead=1: the NEXT keyframe's cloud goes to the device during this keyframe's call, beside the marginalization (glio_set_scan_ahead: the front end has
        // the cloud before the back end is called, Estimator.cpp:5372ff); the next call's slide finds it in slot W - 1.  ahead=0: every call uploads its own.
        bool ahead = false;
        bool have_ahead = false, have_map_ahead = false;
        // map_ahead=1 (with ahead=1): the next call's local map is built during this call's tail too (glio_localmap_push_scan_ahead_and_build)
        bool map_ahead = false;
        // sleep_ms=N: the host sleeps N ms inside every keyframe call (a 10 Hz caller leaves the GPU idle for ~100 ms between calls; the sleep is not part of
        // any stage time).  sleep_at: 0 = between the batch association's preparation and the solve, 1 = before the call's first entry point, 2 = between the solve and
        // the batch association's enqueue
        int sleep_ms = 0, sleep_at = 0;
        std::vector<uint64_t> table;
        size_t table_k = 0;
        for (int a = 5; a < argc; ++a) {
            if (!strncmp(argv[a], "res=", 4)) feature_res = atoi(argv[a] + 4);
            else if (!strncmp(argv[a], "timed=", 6)) timed_last = atoi(argv[a] + 6);
            else if (!strncmp(argv[a], "per_slot=", 9)) per_slot = atoi(argv[a] + 9) != 0;
            else if (!strncmp(argv[a], "stream_draws=", 13)) host_draws = atoi(argv[a] + 13) == 0;
            else if (!strncmp(argv[a], "prepare_early=", 14)) prepare_early = atoi(argv[a] + 14) != 0;
            else if (!strncmp(argv[a], "ahead=", 6)) ahead = atoi(argv[a] + 6) != 0;
            else if (!strncmp(argv[a], "map_ahead=", 10)) map_ahead = atoi(argv[a] + 10) != 0;
            else if (!strncmp(argv[a], "sleep_ms=", 9)) sleep_ms = atoi(argv[a] + 9);
            else if (!strncmp(argv[a], "sleep_at=", 9)) sleep_at = atoi(argv[a] + 9);
            else if (!strncmp(argv[a], "draws=", 6)) {
                FILE* df = fopen(argv[a] + 6, "rb");
                if (!df) { perror("draws"); return 2; }
                fseek(df, 0, SEEK_END); const long bytes = ftell(df); fseek(df, 0, SEEK_SET);
                table.resize((size_t)bytes / 8); rd(df, table.data(), table.size()); fclose(df);
            } else { fprintf(stderr, "unknown argument %s\n", argv[a]); return 2; }
        }
        if (timed_last < 1 || timed_last > NK) timed_last = NK;
        std::mt19937_64 rng(20260925);
        auto rand_u64 = [&]() -> uint64_t { return table.empty() ? (uint64_t)rng() : table[table_k++ % table.size()]; };
        auto rand_below = [&](uint64_t n) -> uint64_t {
            if (!table.empty()) return table[table_k++ % table.size()] % n;
            return std::uniform_int_distribution<uint64_t>(0, n - 1)(rng);
        };
        std::vector<double> kf_poses((size_t)total_kf * 7, 0.0);          // pose_info_keyframe: t, q of every keyframe (ground truth until a window solve moves it)
        for (int j = 0; j < total_kf; ++j) { for (int c = 0; c < 3; ++c) kf_poses[7 * (size_t)j + c] = gtt[3 * (size_t)j + c]; for (int c = 0; c < 4; ++c) kf_poses[7 * (size_t)j + 3 + c] = gtq[4 * (size_t)j + c]; }
        for (int j = 0; j < W - 1; ++j) {                                 // the clouds of the keyframes already in the window
            body = scans[j];
            for (int i = 0; i < pts; ++i) for (int c = 0; c < 3; ++c) body[4 * (size_t)i + c] -= tlb[c];
            ba.setFrame(j, body.data(), pts);
        }
        double stmax[7] = {0, 0, 0, 0, 0, 0, 0};
        double st[7] = {0, 0, 0, 0, 0, 0, 0}, cyc = 0, cmin = 1e9, cmax = 0, fd[3] = {0, 0, 0};
        std::vector<int> iters; std::vector<long> kept; std::vector<long> bfound; std::vector<long> bkept;
        double checksum = 0;
        std::vector<double> last_trans, last_quat;
        int map_pts = 0;
        for (int j = 0; j <= NK; ++j) {
            const Kf& k = kf[j];
            const int nw = j + W - 1;                                   // the keyframe that enters the window
            if (j == 0) { be.tmpTrans = k.trans; be.tmpQuat = k.quat; be.tmpSpeedBias = k.sb; }
            else be.slideState(&k.trans[3 * (W - 1)], &k.quat[4 * (W - 1)], &k.sb[9 * (W - 1)]);      // previous solution shifted + the new keyframe's prediction
            std::vector<double> ddt = k.ddt;
            if (sleep_ms > 0 && sleep_at == 1) std::this_thread::sleep_for(std::chrono::milliseconds(sleep_ms));
            const double t0 = now_s();
            be.slideWindow();
            if (!have_ahead) be.setScan(W - 1, scans[nw].data(), pts);
            have_ahead = false;
            // the keyframe's cloud goes to the batch association's store as soon as it is on the device (body frame: nothing of it depends on the solve); with the
            // deferred variant the previous keyframe's searches are still in flight on that store's stream, and the copy is made once they were collected
            if (!defer) ba.setFrameFromScan(nw, be.ctx(), W - 1, tlb);
            const double t1 = now_s();
            if (!have_map_ahead) map_pts = be.pushScanAndBuildLocalMap(W - 1, tlb, &gtq[4 * nw], &gtt[3 * nw]);
            have_map_ahead = false;
            const double t2 = now_s();
            be.findCorrespondingSurfFeaturesWindowAsync();
            const double t3 = now_s();
            be.setImuFactors(k.pre);
            const double t3a = now_s();
            be.setGnss(&k.frame, k.dd, k.dop);
            // (the batch association's pairs are known: their search frames' tables are cleared NOW, while the GPU searches and the host would only wait --
            //  between the counts and the solve the same call cost 15 us of an idle GPU)
            if (!defer && !after_marg && prepare_early) kba.prepare(nw + 1);
            const double t3b = now_s();
            std::vector<int32_t> counts = be.windowCounts();
            // featureSelection (Estimator.cpp:2223): right behind each slot's search, feature_res_num draws per slot (config_urban_hk.yaml:100: 100)
            // (per_slot=1: W calls of featureSelection() instead of the one window call -- same draws, same records; A/B of the call overhead)
            if (feature_res > 0 && per_slot) for (int s = 0; s < W; ++s) counts[s] = be.featureSelection(s, counts[s], feature_res, rand_below);
            else if (feature_res > 0) be.featureSelectionWindow(counts, feature_res, rand_below);
            if (!defer && !after_marg && !prepare_early) kba.prepare(nw + 1);          // (round 5's place: between the counts and the solve)
            if (sleep_ms > 0 && sleep_at == 0) std::this_thread::sleep_for(std::chrono::milliseconds(sleep_ms));
            const double t4 = now_s();
            const glio_summary sum = be.solve(&ddt);
            const double t5 = now_s();
            // updatePose (:2730): the window's solved poses become pose_info_keyframe; then batchFeatureAssociation, enqueued (it needs only the poses)
            for (int s = 0; s < W; ++s) {
                const int g = j + s;
                for (int c = 0; c < 3; ++c) kf_poses[7 * (size_t)g + c] = be.tmpTrans[3 * s + c];
                for (int c = 0; c < 4; ++c) kf_poses[7 * (size_t)g + 3 + c] = be.tmpQuat[4 * s + c];
            }
            std::vector<int64_t> found;
            double slept2 = 0;
            if (sleep_ms > 0 && sleep_at == 2) { const double ts = now_s(); std::this_thread::sleep_for(std::chrono::milliseconds(sleep_ms)); slept2 = now_s() - ts; }
            if (defer) found = kba.finish(rand_below);                  // the previous keyframe's pairs: they had a whole cycle
            if (defer) ba.setFrameFromScan(nw, be.ctx(), W - 1, tlb);
            if (!after_marg) { if (host_draws) kba.enqueue(nw + 1, kf_poses); else kba.enqueueWithDraws(nw + 1, kf_poses, rand_u64); }
            const double t5b = now_s();
            if (ahead && j < NK) {
                be.marginalizeAndKeepAsync(&ddt);
                be.setScanAhead(scans[nw + 1].data(), pts);             // (the host would only wait for the marginalization here)
                have_ahead = true;
                if (map_ahead) { map_pts = be.pushScanAheadAndBuildLocalMap(tlb, &gtq[4 * (nw + 1)], &gtt[3 * (nw + 1)]); have_map_ahead = true; }
                be.marginalizeFinish();
            } else be.marginalizeAndKeep(&ddt);
            const double t6 = now_s();
            if (after_marg) { if (host_draws) kba.enqueue(nw + 1, kf_poses); else kba.enqueueWithDraws(nw + 1, kf_poses, rand_u64); }
            if (!defer) found = kba.finish(rand_below);
            const double t7 = now_s();
            if (j == 0) continue;                                        // no prior yet, every first-touch cost: warm-up
            { long f = 0; for (int64_t v : found) f += (long)v; bfound.push_back(f); bkept.push_back((long)ba.total()); }
            iters.push_back(sum.iterations);
            { long kk = 0; for (int32_t v : counts) kk += v; kept.push_back(kk); }
            for (double v : be.tmpTrans) checksum += v;
            last_trans = be.tmpTrans; last_quat = be.tmpQuat;
            if (j <= NK - timed_last) continue;                          // (results are booked for every keyframe, times for the last timed_last)
            const double d[7] = {t1 - t0, t2 - t1, t3 - t2, t4 - t3 - (sleep_at == 0 ? sleep_ms * 1e-3 : 0.0), t5 - t4, t6 - t5b, (t5b - t5 - slept2) + (t7 - t6)};
            for (int q = 0; q < 7; ++q) if (d[q] > stmax[q]) stmax[q] = d[q];
            double c = 0;
            for (int q = 0; q < 7; ++q) { st[q] += d[q] / timed_last; c += d[q]; }
            fd[0] += (t3a - t3) / timed_last; fd[1] += (t3b - t3a) / timed_last; fd[2] += (t4 - t3b) / timed_last;
            cyc += c / timed_last; if (c < cmin) cmin = c; if (c > cmax) cmax = c;
        }
        if (defer) kba.finish(rand_below);                               // the last keyframe's pairs
        printf("{\"batch_association_deferred\": %s, \"stages_ms\": {\"slide_and_new_scan\": %.4f, \"local_map\": %.4f, \"associate_enqueue\": %.4f, \"factors_while_the_gpu_searches_then_wait\": %.4f, "
               "\"solve\": %.4f, \"marginalize\": %.4f, \"batch_feature_association_enqueue_and_wait\": %.4f}, \"cycle_ms\": %.4f, \"cycle_ms_min_max\": [%.4f, %.4f], \"keyframes_per_s\": %.1f, \"map_points\": %d, \"iterations\": [",
               defer ? "true" : "false", st[0] * 1e3, st[1] * 1e3, st[2] * 1e3, st[3] * 1e3, st[4] * 1e3, st[5] * 1e3, st[6] * 1e3, cyc * 1e3, cmin * 1e3, cmax * 1e3, 1.0 / cyc, map_pts);
        for (size_t i = 0; i < iters.size(); ++i) printf("%s%d", i ? ", " : "", iters[i]);
        printf("], \"correspondences_kept\": [");
        for (size_t i = 0; i < kept.size(); ++i) printf("%s%ld", i ? ", " : "", kept[i]);
        printf("], \"batch_records_found\": [");
        for (size_t i = 0; i < bfound.size(); ++i) printf("%s%ld", i ? ", " : "", bfound[i]);
        printf("], \"batch_records_held\": [");
        for (size_t i = 0; i < bkept.size(); ++i) printf("%s%ld", i ? ", " : "", bkept[i]);
        printf("], \"factors_stage_ms\": {\"set_imu_host\": %.4f, \"set_gnss_host\": %.4f, \"wait_for_the_searches\": %.4f}", fd[0] * 1e3, fd[1] * 1e3, fd[2] * 1e3);
        printf(", \"stage_max_ms\": [%.4f, %.4f, %.4f, %.4f, %.4f, %.4f, %.4f], \"sleep_ms\": %d, \"sleep_at\": %d", stmax[0] * 1e3, stmax[1] * 1e3, stmax[2] * 1e3, stmax[3] * 1e3, stmax[4] * 1e3,
               stmax[5] * 1e3, stmax[6] * 1e3, sleep_ms, sleep_at);
        printf(", \"batch_feature_res_num\": %d, \"feature_res_num\": %d, \"timed_keyframes\": %d, \"last_trans\": [", RES, feature_res, timed_last);
        for (size_t i = 0; i < last_trans.size(); ++i) printf("%s%.17g", i ? ", " : "", last_trans[i]);
        printf("], \"last_quat\": [");
        for (size_t i = 0; i < last_quat.size(); ++i) printf("%s%.17g", i ? ", " : "", last_quat[i]);
        long long stamps[320] = {0};
        glio_debug_arrow_stamps(be.ctx(), stamps);
        long total_iters = 0; for (int v : iters) total_iters += v;
        // (of the solver steps of the booked keyframes, how many took the helper workgroups' speculative build: slot 300 counts since the context was made,
        //  the warm-up keyframe included)
        printf("], \"steps_with_the_helpers_build\": %lld, \"iterations_booked\": %ld, \"trans_checksum\": %.17g}\n", stamps[300], total_iters, checksum);
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
