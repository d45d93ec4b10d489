import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glio_amd import synth, capi
win = synth.make_window(W=20, pts_per_scan=65536, with_gnss=True, with_prior=True, seed=synth.SEED_BASE + 12)
corr = synth.analytic_correspondences(win)
for mode in (0, 1):
    ctx = capi.Context(win.opts)
    capi.load().glio_debug_set_solver(ctx._h, mode)
    ctx.load_window(win, corr)
    sol, summ = ctx.solve(win.init)
    ms, s2 = ctx.time_solve(win.init, 10)
    print(f"mode {mode}: iterations {summ.iterations} term {summ.termination} cost {summ.final_cost:.12g} solve {ms:.3f} ms  tr_step {ctx.time_kernel(2, 20)*1e3:.1f} us")
    if mode == 0: ref = sol
    else: print("max dtrans", np.abs(sol.trans - ref.trans).max(), "dsb", np.abs(sol.speed_bias - ref.speed_bias).max())
    ctx.close()
import ctypes as C
ctx = capi.Context(win.opts)
ctx.load_window(win, corr)
ctx.linearize(win.init)
ctx.time_kernel(2, 1)
st = (C.c_longlong * 64)()
capi.load().glio_debug_arrow_stamps(ctx._h, st)
v = list(st)
print("forward stamps (us):", [round((v[k + 1] - v[k]) / 100.0, 2) for k in range(0, 5)])
print("solve stamps (us):", [round((v[k + 1] - v[k]) / 100.0, 2) for k in range(8, 13)])
print("busy wave0/wave1 (us):", v[20] / 100.0, v[21] / 100.0)
print("phase3 detail (us): Ys raw", (v[30]-v[2])/100, "Blk raw", (v[31]-v[30])/100, "sync", (v[32]-v[31])/100, "Ys corr", (v[33]-v[32])/100, "Blk corr+sync", (v[3]-v[33])/100)
