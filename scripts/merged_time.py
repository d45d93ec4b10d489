"""k_linearize_all variants (0: separate launches, 1: merged, 2: merged with idle workgroups behind the small-factor CUs)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glio_amd import synth, capi
W = 20
stream = synth.make_window(W=W + 1, pts_per_scan=65536, with_gnss=True, seed=synth.SEED_BASE + 12)
first = synth.sub_window(stream, 0, W)
c0 = capi.Context(first.opts); c0.load_window(first, synth.analytic_correspondences(first))
s0, _ = c0.solve(first.init); prior = c0.marginalize(s0); c0.close()
win = synth.sub_window(stream, 1, W); win.prior = prior
ctx = capi.Context(win.opts); ctx.load_window(win, synth.analytic_correspondences(win))
for on in (0, 1, 2, 0, 1, 2):
    capi.load().glio_debug_set_merged_linearize(ctx._h, on)
    for _ in range(5): sol, summ = ctx.solve(win.init)
    t0 = time.perf_counter()
    for _ in range(50): sol, summ = ctx.solve(win.init)
    wall = (time.perf_counter() - t0) / 50
    lin = min(ctx.time_kernel(1, 30) for _ in range(3)) * 1e3
    print(f"merged {on}: wall {wall*1e3:.4f} ms/solve  iters {summ.iterations}  full_linearize {lin:.2f} us")
