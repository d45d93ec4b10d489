"""Three solves of a C5-shaped window (50 keyframes, no prior; few points: the solver's cost does not depend on them) for scripts/gpu_timeline.sh."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glio_amd import synth, capi
W = int(os.environ.get("C5_W", "50"))
win = synth.make_window(W=W, pts_per_scan=int(os.environ.get("C5_PTS", "4096")), with_gnss=True, with_prior=False, seed=synth.SEED_BASE + 50, gnss_epoch_dt=0.4)
ctx = capi.Context(win.opts); ctx.load_window(win, synth.analytic_correspondences(win))
for _ in range(3):
    sol, summ = ctx.solve(win.init)
    time.sleep(0.01)
ms, _ = ctx.time_solve(win.init, 10)
print("path", capi.load().glio_debug_solver_path(ctx._h), "iterations", summ.iterations, "n", 15 * W + win.init.n_ddt, "solve ms", round(ms, 4), "final cost", repr(summ.final_cost))
