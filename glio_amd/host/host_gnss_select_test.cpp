// host_gnss_select_test.cpp -- CPU-only check program of glio::selectBatchGnssEpochs / glio::ddGroup (Estimator.cpp:3086-3126, :1635-1663, :1702-1860): reads a
// stream description from stdin as text, prints the selected epochs and the double-difference groups.  Driven by tests/test_host_logic.py against the Python twins.
#include <cstdio>
#include <vector>

#include "glio_batch_backend.hpp"

int main() {
    int n_obs = 0, n_kt = 0, first_idx = 0, n_poses = 0, n_tr = 0;
    if (scanf("%d %d %d %d %d", &n_obs, &n_kt, &first_idx, &n_poses, &n_tr) != 5) return 2;
    std::vector<double> obs(n_obs), kt(n_kt), tr(3 * (size_t)n_tr);
    for (double& x : obs) if (scanf("%lf", &x) != 1) return 2;
    for (double& x : kt) if (scanf("%lf", &x) != 1) return 2;
    for (double& x : tr) if (scanf("%lf", &x) != 1) return 2;
    try {
        for (const glio::GnssEpochSlot& s : glio::selectBatchGnssEpochs(obs, kt, first_idx, n_poses, tr)) printf("epoch %d %d %d %.17g\n", s.epoch, s.left_key, s.right_key, s.ts_ratio);
    } catch (const std::invalid_argument& e) { printf("refused %s\n", e.what()); return 0; }
    int nu = 0, nr = 0;
    if (scanf("%d %d", &nu, &nr) != 2) return 0;
    std::vector<int> up(nu), rp(nr);
    std::vector<double> psr(nu), ele(nu);
    for (int& x : up) if (scanf("%d", &x) != 1) return 2;
    for (double& x : psr) if (scanf("%lf", &x) != 1) return 2;
    for (double& x : ele) if (scanf("%lf", &x) != 1) return 2;
    for (int& x : rp) if (scanf("%d", &x) != 1) return 2;
    for (int sys = 0; sys < 4; ++sys) {
        const glio::DdGroup g = glio::ddGroup(sys, up, psr, ele, rp);
        printf("group %d master %d pairs", sys, g.master);
        for (size_t k = 0; k < g.user.size(); ++k) printf(" %d:%d", g.user[k], g.ref[k]);
        printf("\n");
    }
    return 0;
}
