"""Phase stamps of k_chain_solve on the steady-state C2 window (build with GLIO_DEV_STAMPS=1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
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
print("path", capi.load().glio_debug_solver_path(ctx._h), "iterations", summ.iterations, "solve ms", ms, "tr_step us", ctx.time_kernel(2, 20) * 1e3)
st = (C.c_longlong * 320)()
capi.load().glio_debug_arrow_stamps(ctx._h, st)
v = list(st)
print("chain stamps (us): setup, raw loads, corrections, chain, back, epilogue:", [round((v[k + 1] - v[k]) / 100.0, 2) for k in range(40, 46)])
print("chain step phases, totals over the top half-chain (us): loads, 15 pivots, panel store, rank-15 update, correction:", [round(v[60 + k] / 100.0, 2) for k in range(5)])
