// host_writeback_test.cpp -- CPU-only check program of glio::writeBackState (Estimator.cpp:2611-2726): reads W and the arrays
// from stdin as text, prints the arrays after the write-back.  Driven by tests/test_host_logic.py against a Python transcription.
#include <cmath>
#include <cstdio>
#include <vector>

#include "glio_backend.hpp"

static std::vector<double> rd(int n) { std::vector<double> v(n); for (double& x : v) if (scanf("%lf", &x) != 1) return {}; return v; }
static void pr(const char* name, const std::vector<double>& v) { printf("%s", name); for (double x : v) printf(" %.17g", x); printf("\n"); }

int main() {
    int W = 0;
    if (scanf("%d", &W) != 1) return 2;
    std::vector<double> tT = rd(3 * W), tQ = rd(4 * W), tSB = rd(9 * W), tDt = rd(3 * W), Ps = rd(3 * W), Qs = rd(4 * W), Vs = rd(3 * W), psb = rd(9 * W),
                        Bas = rd(3 * W), Bgs = rd(3 * W), ap = rd(7 * W), dt = rd(3 * W);
    glio::writeBackState(W, tT.data(), tQ.data(), tSB.data(), tDt.data(), Ps.data(), Qs.data(), Vs.data(), psb.data(), Bas.data(), Bgs.data(), ap.data(), dt.data());
    pr("Ps", Ps); pr("Qs", Qs); pr("Vs", Vs); pr("psb", psb); pr("Bas", Bas); pr("Bgs", Bgs); pr("abs", ap); pr("dt", dt);
    return 0;
}
