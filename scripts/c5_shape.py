"""BASELINE config C5 shape (W = 50, 256k surf points per keyframe, LiDAR + IMU + prior): K3 at 12.8 M residuals per launch
(512 MB: beyond the 256 MB Infinity Cache, so this is HBM, not cache), full solve, parity vs the oracle on one linearisation."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glio_amd import synth, capi
W, P = 50, 262144
win = synth.make_window(W=W, pts_per_scan=P, with_gnss=False, with_prior=True, seed=synth.SEED_BASE + 15)
corr = synth.analytic_correspondences(win)
nres = sum(len(c[2]) for c in corr)
ctx = capi.Context(win.opts)
ctx.load_window(win, corr, use_gnss=False)
st = win.init.copy(); st.n_ddt = 0
sol, summ = ctx.solve(st)
ms, _ = ctx.time_solve(st, 5)
k3 = min(ctx.time_kernel(0, 20) for _ in range(3))
rd = min(ctx.time_kernel(6, 20) for _ in range(3))
ctx.linearize(st, want_H=False)
print(f"C5: {nres} residuals, n = {15*W}; solve {ms:.3f} ms, {summ.iterations} iterations, term {summ.termination}")
print(f"K3 {k3*1e3:.1f} us -> {nres*40/k3/1e6:.0f} GB/s algorithmic ({nres*40/k3/1e6/8000:.3f} of 8 TB/s); read-only ceiling {nres*40/rd/1e6:.0f} GB/s; tr_step {ctx.time_kernel(2, 10)*1e3:.1f} us, full_linearize {ctx.time_kernel(1, 10)*1e3:.1f} us")
if "--oracle" in sys.argv:
    from oracle import pyoracle as po
    prob = po.Problem(win, corr, use_gnss=False)
    t0 = time.perf_counter(); Ho, go, co = prob.linearize(st); t1 = time.perf_counter() - t0
    Hh, gh, ch = ctx.linearize(st)
    print(f"oracle linearise {t1:.2f} s; rel H {np.linalg.norm(Hh-Ho)/np.linalg.norm(Ho):.2e} g {np.linalg.norm(gh-go)/np.linalg.norm(go):.2e} cost {abs(ch-co)/co:.2e}")
