"""C5-shaped window (50 keyframes, no prior): dense Cholesky vs arrow factorisation vs the keyframe chain with its blocks in global memory
(k_chain_solve<true>, two and four fronts) -- same iterates to rounding, time per solve."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glio_amd import synth, capi
W = int(os.environ.get("C5_W", "50"))
win = synth.make_window(W=W, pts_per_scan=int(os.environ.get("C5_PTS", "4096")), with_gnss=True, with_prior=False, seed=synth.SEED_BASE + 50, gnss_epoch_dt=0.4)
corr = synth.analytic_correspondences(win)
lib = capi.load()
res = {}
for name, mode, fronts in (("dense", 0, 4), ("arrow", 4, 4), ("chain_global_2", 1, 2), ("chain_global_4", 1, 4)):
    lib.glio_debug_chain_fronts(fronts)
    ctx = capi.Context(win.opts); lib.glio_debug_set_solver(ctx._h, mode); ctx.load_window(win, corr)
    sol, summ = ctx.solve(win.init)
    ms, _ = ctx.time_solve(win.init, 10)
    print(name, "path", lib.glio_debug_solver_path(ctx._h), "fronts", lib.glio_debug_chain_fronts_used(ctx._h), "iterations", summ.iterations, "n", 15 * W + win.init.n_ddt,
          "solve ms", round(ms, 4), "final cost", repr(summ.final_cost), flush=True)
    res[name] = (sol, summ)
    ctx.close()
d = res["dense"][0]
for name in ("arrow", "chain_global_2", "chain_global_4"):
    s = res[name][0]
    print(name, "vs dense: max |d trans|", np.abs(s.trans - d.trans).max(), "quat", np.abs(s.quat - d.quat).max(), "speed_bias", np.abs(s.speed_bias - d.speed_bias).max())
