"""N processes sharing ONE GPU, each solving its own replica window (bench.py's seeds) over and over: every solve must reproduce its first
result.  (Found a timing-dependent failure of the window solve under contention in round 3.)   python scripts/contention_repro.py [N] [solves]"""
import os, subprocess, sys, json
HERE = os.path.dirname(os.path.abspath(__file__))
if os.environ.get("REPRO_RANK") is None:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    reps = sys.argv[2] if len(sys.argv) > 2 else "100"
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), reps], env=dict(os.environ, REPRO_RANK=str(r)), stdout=subprocess.PIPE, text=True) for r in range(n)]
    for p in procs:
        out, _ = p.communicate()
        print(out.strip().splitlines()[-1] if out.strip() else f"(no output, rc {p.returncode})")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
from glio_amd import synth, capi
rank = int(os.environ["REPRO_RANK"]); reps = int(sys.argv[1])
W, pts = 20, 65536
seed = synth.SEED_BASE + 12 + 1000 * rank
stream = synth.make_window(W=W + 1, pts_per_scan=pts, with_gnss=True, with_prior=False, seed=seed)
first = synth.sub_window(stream, 0, W)
ctx0 = capi.Context(first.opts); ctx0.load_window(first, synth.analytic_correspondences(first))
try:
    sol0, s0 = ctx0.solve(first.init)
except Exception as e:
    print(json.dumps({"rank": rank, "first_window_failed": str(e)[:120]})); sys.exit(0)
prior = ctx0.marginalize(sol0); ctx0.close()
win = synth.sub_window(stream, 1, W); win.prior = prior
ctx = capi.Context(win.opts); ctx.load_window(win, synth.analytic_correspondences(win))
seen, fails = {}, []
for i in range(reps):
    try:
        sol, summ = ctx.solve(win.init)
        key = (summ.iterations, summ.termination, float(summ.final_cost))
        seen[key] = seen.get(key, 0) + 1
    except Exception as e:
        fails.append((i, str(e)[-60:]))
diag = None
if fails:
    diag = {}
    H, g, cost = ctx.linearize(win.init)
    diag["linearize_cost"] = float(cost); diag["H_finite"] = bool(np.isfinite(H).all()); diag["H_min_diag"] = float(np.diag(H).min())
    from oracle import pyoracle as po
    # which part is off?  the same window linearised WITHOUT the prior / gnss / imu in turn
    for name, kw in (("no_prior", dict(use_prior=False)), ("no_gnss", dict(use_gnss=False)), ("no_imu", dict(use_imu=False))):
        c2 = capi.Context(win.opts); c2.load_window(win, synth.analytic_correspondences(win), **kw)
        try:
            s2, m2 = c2.solve(win.init); diag[name] = [m2.iterations, m2.termination]
        except Exception as e:
            diag[name] = str(e)[-50:]
        c2.close()
    # the same context after re-uploading everything
    ctx.load_window(win, synth.analytic_correspondences(win))
    try:
        s3, m3 = ctx.solve(win.init); diag["after_reupload"] = [m3.iterations, m3.termination, float(m3.final_cost)]
    except Exception as e:
        diag["after_reupload"] = str(e)[-50:]
    # the prior itself: finite?  and the one a fresh marginalization of the first window gives now
    diag["prior_finite"] = bool(np.isfinite(prior["lin_jac"]).all() and np.isfinite(prior["lin_res"]).all())
    diag["prior_norm"] = float(np.linalg.norm(prior["lin_jac"]))
    c4 = capi.Context(first.opts); c4.load_window(first, synth.analytic_correspondences(first))
    s4, m4 = c4.solve(first.init); p4 = c4.marginalize(s4); c4.close()
    diag["prior_again_norm"] = float(np.linalg.norm(p4["lin_jac"])); diag["prior_diff"] = float(np.abs(p4["lin_jac"] - prior["lin_jac"]).max())
    diag["sol0_diff"] = float(np.abs(s4.trans - sol0.trans).max())
print(json.dumps({"rank": rank, "distinct_results": [[list(k), v] for k, v in seen.items()], "failures": len(fails), "first_failures": fails[:3], "diag": diag}))
