"""The headline window of bench.py for the seeds ranks 0..N-1 would use (seed = SEED_BASE + 12 + 1000 rank): does every replica's window solve?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glio_amd import synth, capi
W, pts = 20, 65536
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for rank in range(n):
    seed = synth.SEED_BASE + 12 + 1000 * rank
    stream = synth.make_window(W=W + 1, pts_per_scan=pts, with_gnss=True, with_prior=False, seed=seed)
    first = synth.sub_window(stream, 0, W)
    try:
        ctx0 = capi.Context(first.opts); ctx0.load_window(first, synth.analytic_correspondences(first))
        sol0, s0 = ctx0.solve(first.init); prior = ctx0.marginalize(sol0); ctx0.close()
        win = synth.sub_window(stream, 1, W); win.prior = prior
        ctx = capi.Context(win.opts); ctx.load_window(win, synth.analytic_correspondences(win))
        res = []
        for _ in range(3):
            sol, summ = ctx.solve(win.init); res.append((summ.iterations, summ.termination, round(summ.final_cost, 3)))
        print(rank, "first window", (s0.iterations, s0.termination), "timed window", res, "path", capi.load().glio_debug_solver_path(ctx._h))
        ctx.close()
    except Exception as e:
        print(rank, "FAILED:", str(e)[:200])
