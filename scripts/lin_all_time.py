"""The fused linearisation launch of the C2 window: k_linearize_all (mean of 3 x 50 back-to-back launches), K3 as its own launch, the solve.
GLIO_HIP_LIB selects a library variant."""
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
for _ in range(5): sol, summ = ctx.solve(win.init)
t0 = time.perf_counter()
for _ in range(100): sol, summ = ctx.solve(win.init)
wall = (time.perf_counter() - t0) / 100
ctx.linearize(win.init, want_H=False)
la = [ctx.time_kernel(capi.KERNEL_LINEARIZE_ALL, 50) * 1e3 for _ in range(3)]
k3 = [ctx.time_kernel(capi.KERNEL_LIDAR_LINEARIZE, 50) * 1e3 for _ in range(3)]
print(os.environ.get("GLIO_HIP_LIB", "product"), f"solve {wall*1e3:.4f} ms iters {summ.iterations} cost {summ.final_cost:.6f}  linearize_all {np.round(la, 2)} us  k3 alone {np.round(k3, 2)} us")
