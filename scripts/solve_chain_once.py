"""Three solves of the steady-state C2 window (marginalization prior -> keyframe-chain step): the workload of scripts/chain_icache_pmc.sh."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glio_amd import synth, capi
W = 20
stream = synth.make_window(W=W + 1, pts_per_scan=65536, with_gnss=True, seed=synth.SEED_BASE + 12)
first = synth.sub_window(stream, 0, W)
c0 = capi.Context(first.opts); c0.load_window(first, synth.analytic_correspondences(first))
s0, _ = c0.solve(first.init); prior = c0.marginalize(s0); c0.close()
win = synth.sub_window(stream, 1, W); win.prior = prior
ctx = capi.Context(win.opts); ctx.load_window(win, synth.analytic_correspondences(win))
for _ in range(3):
    sol, summ = ctx.solve(win.init)
    time.sleep(0.01)
print("path", capi.load().glio_debug_solver_path(ctx._h), "iterations", summ.iterations)
