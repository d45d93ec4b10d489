import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glio_amd import synth, capi
win = synth.make_window(W=20, pts_per_scan=65536, with_gnss=True, with_prior=True, seed=synth.SEED_BASE + 12)
corr = synth.analytic_correspondences(win)
nres = sum(len(c[2]) for c in corr)
ctx = capi.Context(win.opts)
ctx.load_window(win, corr)
H0, g0, c0 = ctx.linearize(win.init)
for unroll in (22, 24, 32, 33, 34):
    for bpk in (19, 26, 38, 51, 64):
        capi.load().glio_debug_set_k3(ctx._h, bpk, unroll)
        H, g, c = ctx.linearize(win.init)
        if not (abs(c - c0) < 1e-9 * c0 and abs(H - H0).max() <= 1e-9 * abs(H0).max() and abs(g - g0).max() <= 1e-9 * abs(g0).max()):
            print('MISMATCH', unroll, bpk, c, c0, abs(H - H0).max() / abs(H0).max(), abs(g - g0).max() / abs(g0).max()); continue
        ms = min(ctx.time_kernel(0, 50) for _ in range(3))
        print(f"unroll {unroll:2d} bpk {bpk:4d} blocks {bpk*20:5d}: {ms*1e3:6.2f} us  {nres*40/ms/1e6:7.1f} GB/s")
print("stream read us", min(ctx.time_kernel(6, 50) for _ in range(3)) * 1e3)
