import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glio_amd import synth, capi
win = synth.make_window(W=20, pts_per_scan=65536, with_gnss=True, with_prior=True, seed=synth.SEED_BASE + 12)
corr = synth.analytic_correspondences(win)
nres = sum(len(c[2]) for c in corr)
ctx = capi.Context(win.opts)
ctx.load_window(win, corr)
H0, g0, c0 = ctx.linearize(win.init)
for unroll in (1, 2, 4, 8):
    for bpk in (12, 25, 38, 51, 64, 102, 128, 204, 256):
        capi.load().glio_debug_set_k3(ctx._h, bpk, unroll)
        H, g, c = ctx.linearize(win.init)
        assert abs(c - c0) < 1e-9 * c0
        ms = ctx.time_kernel(0, 50)
        print(f"unroll {unroll} bpk {bpk:4d} blocks {bpk*20:5d}: {ms*1e3:6.2f} us  {nres*40/ms/1e6:7.1f} GB/s")
# factor subsets
for name, kw in [("all", {}), ("no gnss", dict(use_gnss=False)), ("no prior", dict(use_prior=False)), ("no imu", dict(use_imu=False)),
                 ("lidar only", dict(use_gnss=False, use_prior=False, use_imu=False))]:
    c2 = capi.Context(win.opts)
    c2.load_window(win, corr, **kw)
    st = win.init.copy()
    if kw.get("use_gnss") is False: st.n_ddt = 0
    c2.linearize(st)
    print(f"full_linearize [{name}]: {c2.time_kernel(1, 30)*1e3:.1f} us   tr_step {c2.time_kernel(2, 10)*1e3:.1f} us")
    c2.close()
