"""k_linearize_all on the C2 window (with a prior) under the placements of its small-factor workgroups: merged_linearize 1 = first n_small workgroups, 2 = the same
with the workgroups that would share their CUs idle, 3 = the small-factor workgroups two to a CU."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glio_amd import synth, capi
win = synth.make_window(W=20, pts_per_scan=65536, with_gnss=True, with_prior=True, seed=synth.SEED_BASE + 12)
ctx = capi.Context(win.opts); ctx.load_window(win, synth.analytic_correspondences(win))
for mode in (2, 3, 1, 2, 3):
    capi.load().glio_debug_set_merged_linearize(ctx._h, mode)
    ctx.linearize(win.init, want_H=False)
    la = [ctx.time_kernel(capi.KERNEL_LINEARIZE_ALL, 50) * 1e3 for _ in range(4)]
    sol, summ = ctx.solve(win.init)
    ms, _ = ctx.time_solve(win.init, 20)
    print("placement", mode, "linearize_all us", np.round(la, 2), "solve ms", round(ms, 4), "iterations", summ.iterations, "trans checksum", float(sol.trans.sum()))
print("k3 alone", round(ctx.time_kernel(capi.KERNEL_LIDAR_LINEARIZE, 50) * 1e3, 2))
