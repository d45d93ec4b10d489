// host_demo_odometry.cpp -- LidarOdometry::run() (GLIO/src/LidarOdometry.cpp:661-699) per scan of a stream, driven from C++ through
// glio::ScanToMapOdometry (glio_backend.hpp): pose initialisation from the last relative motion, the 20-frame / 0.2 m local map resident on the device,
// match_cnt rounds of [association at the current pose, Levenberg-Marquardt solve], pose bookkeeping.  Input: a flat file written by
// glio_amd/host/window_io.py::write_odometry_stream (opts | n_scans scan_match_cnt 0 0 | per scan: n, points xyzi).  Output: one text line per scan
// (`pose i  q[4] t[3]  rounds  kept  iterations  final_cost  map_points`) and a JSON line with the time per scan -- tests/test_host_cpp.py compares the
// poses with the Python twin (glio_amd/odometry.py) bit for bit; bench.py reports the time as front_end_odometry.cpp_update_ms.
// Build: g++ -std=c++14 -O2 host_demo_odometry.cpp -I../../include -L../lib -lglio_hip -Wl,-rpath,'$ORIGIN/../lib'
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "glio_backend.hpp"

template <typename T> static void rd(FILE* f, T* p, size_t n) { if (n && fread(p, sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } }
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: host_demo_odometry stream.bin [device]\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    const int device = argc > 2 ? atoi(argv[2]) : 0;
    glio_opts opts;
    rd(f, &opts, 1);
    int32_t hdr[4];
    rd(f, hdr, 4);
    const int n_scans = hdr[0], match_cnt = hdr[1] > 0 ? hdr[1] : 1;
    std::vector<std::vector<float>> scans(n_scans);
    for (std::vector<float>& s : scans) { int32_t n; rd(f, &n, 1); s.resize((size_t)n * 4); rd(f, s.data(), s.size()); }
    fclose(f);
    try {
        glio::ScanToMapOdometry odo(opts, device, match_cnt);
        std::vector<glio::ScanToMapOdometry::Round> rounds;
        double t_steady = 0; int n_steady = 0;
        for (int i = 0; i < n_scans; ++i) {
            const double t0 = now_s();
            const std::array<double, 7> p = odo.run(scans[i].data(), (int)(scans[i].size() / 4), &rounds);
            const double dt = now_s() - t0;
            if (i >= 3) { t_steady += dt; ++n_steady; }          // (the first scans initialise: 8 matching rounds, first-touch costs)
            long kept = 0; int iters = 0; double cost = 0;
            for (const auto& r : rounds) { kept += r.kept; iters += r.summary.iterations; cost = r.summary.final_cost; }
            printf("pose %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g %zu %ld %d %.17g %d\n", i, p[0], p[1], p[2], p[3], p[4], p[5], p[6], rounds.size(), kept, iters, cost,
                   odo.mapPoints());
        }
        printf("{\"scans\": %d, \"steady_scans\": %d, \"ms_per_scan\": %.4f}\n", n_scans, n_steady, n_steady ? 1e3 * t_steady / n_steady : 0.0);
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
