"""N processes sharing the GPU, each solving its steady-state C2-sized window R times on one context: every solve must reproduce the process's first
result bit for bit and none may hang (k_chain_step's helper workgroups and its four fronts wait on each other inside one launch while the other
processes' kernels compete for the compute units)."""
import os, subprocess, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
if os.environ.get("STRESS_RANK") is None:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    reps = sys.argv[2] if len(sys.argv) > 2 else "3000"
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), reps], env=dict(os.environ, STRESS_RANK=str(r)), stdout=subprocess.PIPE, text=True) for r in range(n)]
    bad = 0
    for p in procs:
        out, _ = p.communicate(timeout=600)
        line = out.strip().splitlines()[-1] if out.strip() else f"(no output, rc {p.returncode})"
        print(line)
        bad += 0 if line.startswith("OK") else 1
    print("processes", n, "failed", bad, "wall s", round(time.time() - t0, 1))
    sys.exit(1 if bad else 0)
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
from glio_amd import synth, capi
rank = int(os.environ["STRESS_RANK"]); reps = int(sys.argv[1])
W = 20
stream = synth.make_window(W=W + 1, pts_per_scan=int(os.environ.get("STRESS_PTS", "8192")), with_gnss=True, with_prior=False, seed=synth.SEED_BASE + 12 + 1000 * rank)
first = synth.sub_window(stream, 0, W)
c0 = capi.Context(first.opts); c0.load_window(first, synth.analytic_correspondences(first))
s0, _ = c0.solve(first.init); prior = c0.marginalize(s0); c0.close()
win = synth.sub_window(stream, 1, W); win.prior = prior
ctx = capi.Context(win.opts); ctx.load_window(win, synth.analytic_correspondences(win))
ref = None
t0 = time.time()
for k in range(reps):
    sol, sm = ctx.solve(win.init)
    d = (sol.trans.tobytes(), sol.quat.tobytes(), sol.speed_bias.tobytes(), int(sm.iterations), float(sm.final_cost))
    ref = ref or d
    if d != ref:
        print("MISMATCH rank", rank, "solve", k); sys.exit(1)
print("OK rank", rank, "solves", reps, "iterations", ref[3], "path", capi.load().glio_debug_solver_path(ctx._h), "fronts", capi.load().glio_debug_chain_fronts_used(ctx._h),
      "ms per solve under contention", round((time.time() - t0) / reps * 1e3, 3))
