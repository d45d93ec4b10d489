// host_demo.cpp -- the C++ call sequence of optimizeSlidingWindowWithLandMark() on the HIP backend, fed from
// a flat binary window file written by tests (glio_amd/host/window_io.py).  Prints the solved state as text.
// Build: g++ -std=c++14 -O2 host_demo.cpp -I../../include -L../lib -lglio_hip -Wl,-rpath,'$ORIGIN/../lib'
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "glio_backend.hpp"

template <typename T> static void rd(FILE* f, T* p, size_t n) { if (fread(p, sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } }

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: host_demo window.bin\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    glio_opts opts;
    rd(f, &opts, 1);
    int32_t hdr[4];   // n_map, n_imu, has_prior(0), reserved
    rd(f, hdr, 4);
    const int W = opts.window;
    try {
        glio::SlidingWindowBackend be(opts);
        std::vector<float> map((size_t)hdr[0] * 4);
        rd(f, map.data(), map.size());
        be.setLocalMap(map.data(), hdr[0]);
        rd(f, be.tmpTrans.data(), 3 * W); rd(f, be.tmpQuat.data(), 4 * W); rd(f, be.tmpSpeedBias.data(), 9 * W);
        std::vector<glio_preint> pre(hdr[1]);
        if (hdr[1]) rd(f, pre.data(), pre.size());
        be.setImuFactors(pre);
        long kept = 0;
        for (int s = 0; s < W; ++s) {
            int32_t n;
            rd(f, &n, 1);
            std::vector<float> scan((size_t)n * 4);
            rd(f, scan.data(), scan.size());
            kept += be.findCorrespondingSurfFeatures(s, scan.data(), n);
        }
        fclose(f);
        std::vector<double> Ps = be.tmpTrans, Qs = be.tmpQuat, Vs(3 * W), psb = be.tmpSpeedBias;
        for (int i = 0; i < W; ++i) for (int k = 0; k < 3; ++k) Vs[3 * i + k] = be.tmpSpeedBias[9 * i + k];
        const glio_summary sum = be.solve();
        be.writeBack(Ps.data(), Qs.data(), Vs.data(), psb.data());
        printf("kept %ld iterations %d termination %d cost %.17g -> %.17g\n", kept, sum.iterations, sum.termination, sum.initial_cost, sum.final_cost);
        for (int i = 0; i < W; ++i)
            printf("kf %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", i, Ps[3 * i], Ps[3 * i + 1], Ps[3 * i + 2], Qs[4 * i], Qs[4 * i + 1], Qs[4 * i + 2], Qs[4 * i + 3]);
        const glio::MarginalizationPrior m = be.marginalize();
        double tr = 0, rr = 0;       // trace(J0^T J0) = |J0|_F^2 and |r0|^2: invariant under the choice of square root
        for (double v : m.linearized_jacobians) tr += v * v;
        for (double v : m.linearized_residuals) rr += v * v;
        printf("prior %d %zu %.17g %.17g\n", m.n, m.keep_block_slot.size(), tr, rr);
        // the device-resident form of the same cycle: the scans are already resident from the per-slot calls above, so one
        // call re-associates the whole window at the solved poses; then marginalize-and-keep, and the kept prior is used by
        // a second solve of the same window
        const std::vector<int32_t> counts = be.findCorrespondingSurfFeaturesWindow();
        long kept_window = 0;
        for (int32_t c : counts) kept_window += c;
        be.marginalizeAndKeep();
        const glio_summary sum2 = be.solve();
        printf("resident %ld %d %.17g\n", kept_window, sum2.iterations, sum2.final_cost);
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
