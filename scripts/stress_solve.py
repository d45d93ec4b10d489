"""Repeat-solve stress: alternating contexts (small GNSS window / C1-shaped window), every solve must reproduce its first
result bit for bit (the solver is deterministic) -- flushes out timing-dependent bugs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glio_amd import synth, capi
wa = synth.make_window(W=3, pts_per_scan=384, with_gnss=True, with_prior=True, seed=synth.SEED_BASE + 77)
ca = synth.analytic_correspondences(wa)
wb = synth.make_window(W=10, pts_per_scan=16384, seed=synth.SEED_BASE + 11)
cb = synth.analytic_correspondences(wb)
ref = {}
bad = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for it in range(N):
    for name, (w, c) in (("a", (wa, ca)), ("b", (wb, cb))):
        ctx = capi.Context(w.opts); ctx.load_window(w, c)
        if it % 2: ctx.linearize(w.init)
        sol, summ = ctx.solve(w.init)
        key = (summ.iterations, sol.trans.tobytes(), sol.quat.tobytes())
        if name not in ref: ref[name] = key; print(name, "reference iterations", summ.iterations)
        elif key != ref[name]:
            bad += 1
            print("MISMATCH", name, "run", it, "iterations", summ.iterations, "vs", ref[name][0], "max dt", np.abs(sol.trans - np.frombuffer(ref[name][1]).reshape(sol.trans.shape)).max())
        ctx.close()
print("runs", 2 * N, "mismatches", bad)
