"""N processes sharing one GPU; each repeats (new context, load the first window, solve, marginalize) and compares with its own first good result."""
import os, subprocess, sys, json, time
HERE = os.path.dirname(os.path.abspath(__file__))
if os.environ.get("REPRO_RANK") is None:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    reps = sys.argv[2] if len(sys.argv) > 2 else "20"
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), reps], env=dict(os.environ, REPRO_RANK=str(r)), stdout=subprocess.PIPE, text=True) for r in range(n)]
    for p in procs:
        out, _ = p.communicate()
        print(out.strip().splitlines()[-1] if out.strip() else f"(no output, rc {p.returncode})")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
from glio_amd import synth, capi
rank = int(os.environ["REPRO_RANK"]); reps = int(sys.argv[1])
W, pts = 20, int(os.environ.get("REPRO_PTS", "65536"))
seed = synth.SEED_BASE + 12 + 1000 * rank
stream = synth.make_window(W=W + 1, pts_per_scan=pts, with_gnss=True, with_prior=False, seed=seed)
first = synth.sub_window(stream, 0, W)
corr = synth.analytic_correspondences(first)
events, results, keep = [], {}, []
for it in range(reps):
    try:
        c = capi.Context(first.opts); c.load_window(first, corr)
        sol, sm = c.solve(first.init)
        if os.environ.get("REPRO_TWICE"):
            sol2, sm2 = c.solve(first.init)
            d12 = float(np.abs(sol2.trans - sol.trans).max())
            if d12 > 0 or sm2.iterations != sm.iterations: events.append([it, "first/second solve differ", d12, float(np.abs(sol.trans).sum()), float(np.abs(sol2.trans).sum()), int((sol.trans == 0).all(axis=1).sum())])
            sol, sm = sol2, sm2
        p = c.marginalize(sol)
        key = (sm.iterations, sm.termination, round(float(np.linalg.norm(p["lin_jac"])), 6), round(float(np.abs(sol.trans).sum()), 9))
        results[key] = results.get(key, 0) + 1
        if len(results) > 1 and results[key] == 1: events.append([it, list(key)])
        if os.environ.get("REPRO_DEVSYNC"):
            import torch; torch.cuda.synchronize()
        if os.environ.get("REPRO_NOCLOSE"): keep.append(c)
        else: c.close()
    except Exception as e:
        events.append([it, str(e)[-70:]])
        try:
            if not os.environ.get("REPRO_NOCLOSE"): c.close()
        except Exception: pass
print(json.dumps({"rank": rank, "distinct": len(results), "events": events[:6]}))
