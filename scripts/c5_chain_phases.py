"""Phase stamps of k_chain_solve<true> on a C5-shaped window (build with GLIO_DEV_STAMPS: scripts/build_variant.sh stamps -DGLIO_DEV_STAMPS)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
from glio_amd import synth, capi
W = int(os.environ.get("C5_W", "50"))
win = synth.make_window(W=W, pts_per_scan=int(os.environ.get("C5_PTS", "4096")), with_gnss=True, with_prior=False, seed=synth.SEED_BASE + 50, gnss_epoch_dt=0.4)
ctx = capi.Context(win.opts); ctx.load_window(win, synth.analytic_correspondences(win))
sol, summ = ctx.solve(win.init)
ms, _ = ctx.time_solve(win.init, 10)
print("path", capi.load().glio_debug_solver_path(ctx._h), "fronts", capi.load().glio_debug_chain_fronts_used(ctx._h), "iterations", summ.iterations, "solve ms", round(ms, 4),
      "tr_step us", round(ctx.time_kernel(2, 20) * 1e3, 2))
ctx.time_kernel(2, 1)
st = (C.c_longlong * 320)()
capi.load().glio_debug_arrow_stamps(ctx._h, st)
v = list(st)
d = lambda a, b: round((v[a] - v[b]) / 100.0, 2)
print("k_chain_solve<true> phases (us): state machine + tables", d(41, 40) if v[40] else None, "| staging + t = H u", d(42, 41), "| epoch corrections", d(43, 42), "| chain", d(44, 43),
      "| back substitution", d(45, 44), "| z out", d(46, 45))
if v[111]:
    print("four fronts, us after the start of the chain: A", d(100, 43), "B", d(113, 43), "C", d(116, 43), "D", d(114, 43), "| left meeting block", d(111, 43), "right", d(115, 43),
          "| separator factored", d(102, 43), "its back substitution", d(103, 43))
print("chain step phases of front A, totals (us): loads, 15 pivots, panel store, rank-15 update, correction:", [round(v[60 + k] / 100.0, 2) for k in range(5)])
