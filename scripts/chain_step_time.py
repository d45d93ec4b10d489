"""Phase stamps of k_chain_step on the steady-state C2 window (build with GLIO_DEV_STAMPS=1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
from glio_amd import synth, capi
W = 20
stream = synth.make_window(W=W + 1, pts_per_scan=65536, with_gnss=True, seed=synth.SEED_BASE + 12)
first = synth.sub_window(stream, 0, W)
c0 = capi.Context(first.opts); c0.load_window(first, synth.analytic_correspondences(first))
s0, _ = c0.solve(first.init); prior = c0.marginalize(s0); c0.close()
win = synth.sub_window(stream, 1, W); win.prior = prior
corr = synth.analytic_correspondences(win)
ctx = capi.Context(win.opts); ctx.load_window(win, corr)
sol, summ = ctx.solve(win.init)
ms, _ = ctx.time_solve(win.init, 10)
print("path", capi.load().glio_debug_solver_path(ctx._h), "iterations", summ.iterations, "solve ms", ms, "tr_step us", ctx.time_kernel(2, 20) * 1e3,
      "linearize_all us", ctx.time_kernel(7, 20) * 1e3)
ctx.time_kernel(2, 1)
st = (C.c_longlong * 320)()
capi.load().glio_debug_arrow_stamps(ctx._h, st)
v = list(st)
names = ["tables", "gather diag/g/cost", "state machine", "epoch cols", "block gather+scale", "t = H u", "epoch corrections", "chain", "back subst + z", "factor body", "dogleg"]
print("k_chain_step phases (us):")
for k, nm in enumerate(names):
    print(f"  {nm:22s} {(v[41 + k] - v[40 + k]) / 100.0:7.2f}")
print(f"  total                  {(v[51] - v[40]) / 100.0:7.2f}")
print("chain step phases, totals over the top half-chain of 10 steps (us): loads, 15 pivots, panel store, rank-15 update, correction:", [round(v[60 + k] / 100.0, 2) for k in range(5)])
