"""Two fronts vs separator + four fronts in k_chain_step on the steady-state C2 window: same iterates (to rounding), time per step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glio_amd import synth, capi
W = int(os.environ.get("AB_W", "20"))
stream = synth.make_window(W=W + 1, pts_per_scan=int(os.environ.get("AB_PTS", "65536")), with_gnss=True, seed=synth.SEED_BASE + 12)
first = synth.sub_window(stream, 0, W)
c0 = capi.Context(first.opts); c0.load_window(first, synth.analytic_correspondences(first))
s0, _ = c0.solve(first.init); prior = c0.marginalize(s0); c0.close()
win = synth.sub_window(stream, 1, W); win.prior = prior
corr = synth.analytic_correspondences(win)
out = {}
for fronts in (2, 4, 2, 4):
    capi.load().glio_debug_chain_fronts(fronts)
    ctx = capi.Context(win.opts); ctx.load_window(win, corr)
    sol, summ = ctx.solve(win.init)
    ms, _ = ctx.time_solve(win.init, 30)
    us = ctx.time_kernel(2, 60) * 1e3
    print("fronts", fronts, "path", capi.load().glio_debug_solver_path(ctx._h), "iterations", summ.iterations, "final cost", repr(summ.final_cost), "solve ms", round(ms, 4),
          "tr_step us", round(us, 2), flush=True)
    out[fronts] = (sol, summ)
    ctx.close()
(a, sa), (b, sb) = out[2], out[4]
print("max |d trans|", np.abs(a.trans - b.trans).max(), "max |d quat|", np.abs(a.quat - b.quat).max(), "max |d speed_bias|", np.abs(a.speed_bias - b.speed_bias).max(),
      "rel d cost", abs(sa.final_cost - sb.final_cost) / abs(sa.final_cost), "iterations", sa.iterations, sb.iterations)
