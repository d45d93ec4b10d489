"""Like stress_solve.py, but with the host idle between linearize and solve (as when a CPU oracle runs in between): the
GPU drops to its idle clocks, which changes the relative timing of host enqueue and device execution."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glio_amd import synth, capi
cases = {
    "small_lidar_imu": (synth.make_window(W=4, pts_per_scan=600, with_gnss=True, with_prior=True, seed=synth.SEED_BASE), dict(use_gnss=False, use_prior=False)),
    "c1": (synth.make_window(W=10, pts_per_scan=16384, seed=synth.SEED_BASE + 11), {}),
}
corr = {k: synth.analytic_correspondences(w) for k, (w, _) in cases.items()}
ref, bad = {}, 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 25
idle = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
for it in range(N):
    for name, (w, kw) in cases.items():
        ctx = capi.Context(w.opts); ctx.load_window(w, corr[name], **kw)
        st = w.init.copy()
        if kw.get("use_gnss") is False: st.n_ddt = 0
        ctx.linearize(st)
        time.sleep(idle)
        sol, summ = ctx.solve(st)
        key = (summ.iterations, sol.trans.tobytes(), sol.quat.tobytes())
        if name not in ref: ref[name] = key; print(name, "reference iterations", summ.iterations)
        elif key != ref[name]:
            bad += 1; print("MISMATCH", name, "run", it, "iterations", summ.iterations, "vs", ref[name][0])
        ctx.close()
print("runs", 2 * N, "mismatches", bad)
