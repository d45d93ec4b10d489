"""k_linearize_all against the number of K3 workgroups per keyframe (the merged launch: 63 small-factor workgroups + 20 x bpk K3 workgroups, two
workgroups of 70 KB LDS fit a CU) -- 3 x 50 back-to-back launches each, C2 steady-state window."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glio_amd import synth, capi
W = 20
stream = synth.make_window(W=W + 1, pts_per_scan=65536, with_gnss=True, seed=synth.SEED_BASE + 12)
first = synth.sub_window(stream, 0, W)
c0 = capi.Context(first.opts); c0.load_window(first, synth.analytic_correspondences(first))
s0, _ = c0.solve(first.init); prior = c0.marginalize(s0); c0.close()
win = synth.sub_window(stream, 1, W); win.prior = prior
ctx = capi.Context(win.opts); ctx.load_window(win, synth.analytic_correspondences(win))
ctx.solve(win.init)
for bpk in (24, 12, 16, 18, 19, 20, 21, 22, 23, 24, 25, 26, 28, 32, 40, 48):
    capi.load().glio_debug_set_k3(ctx._h, bpk, 22)
    ctx.solve(win.init)
    t = min(ctx.time_kernel(7, 50) for _ in range(3)) * 1e3
    ms, _ = ctx.time_solve(win.init, 10)
    print(f"bpk {bpk:3d} K3 workgroups {bpk * W:5d}: k_linearize_all {t:6.2f} us   K3 alone {ctx.time_kernel(0, 50) * 1e3:6.2f} us   solve {ms:.4f} ms", flush=True)
