import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glio_amd import synth, capi
win = synth.make_window(W=20, pts_per_scan=65536, with_gnss=True, with_prior=True, seed=synth.SEED_BASE + 12)
corr = synth.analytic_correspondences(win)
ctx = capi.Context(win.opts)
ctx.load_window(win, corr)
ref = None
for lead in (0, 1, 2, 3):
    capi.load().glio_debug_set_enqueue_lead(ctx._h, lead)
    for _ in range(3): sol, summ = ctx.solve(win.init)
    t0 = time.perf_counter()
    for _ in range(20): sol, summ = ctx.solve(win.init)
    wall = (time.perf_counter() - t0) / 20
    ms, _ = ctx.time_solve(win.init, 10)
    if ref is None: ref = sol
    print(f"lead {lead}: wall {wall*1e3:.3f} ms  events {ms:.3f} ms  iters {summ.iterations} same {np.array_equal(ref.trans, sol.trans)}")
