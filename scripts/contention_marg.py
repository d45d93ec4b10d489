"""N processes sharing one GPU: first window solve, then glio_marginalize several times: do the results agree?  (round-3 hunt)"""
import os, subprocess, sys, json
HERE = os.path.dirname(os.path.abspath(__file__))
if os.environ.get("REPRO_RANK") is None:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, REPRO_RANK=str(r)), stdout=subprocess.PIPE, text=True) for r in range(n)]
    for p in procs:
        out, _ = p.communicate()
        print(out.strip().splitlines()[-1] if out.strip() else f"(no output, rc {p.returncode})")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
from glio_amd import synth, capi
rank = int(os.environ["REPRO_RANK"])
W, pts = 20, 65536
seed = synth.SEED_BASE + 12 + 1000 * rank
stream = synth.make_window(W=W + 1, pts_per_scan=pts, with_gnss=True, with_prior=False, seed=seed)
first = synth.sub_window(stream, 0, W)
out = {"rank": rank}
ctx0 = capi.Context(first.opts); ctx0.load_window(first, synth.analytic_correspondences(first))
try:
    sol0, s0 = ctx0.solve(first.init)
    out["solve"] = [s0.iterations, s0.termination]
    ps = [ctx0.marginalize(sol0) for _ in range(3)]
    capi.load().glio_synchronize(ctx0._h)
    ps.append(ctx0.marginalize(sol0))
    ref = ps[-1]["lin_jac"]
    out["norms"] = [float(np.linalg.norm(p["lin_jac"])) for p in ps]
    out["diffs_vs_last"] = [float(np.abs(p["lin_jac"] - ref).max()) for p in ps]
    bad = [i for i, p in enumerate(ps) if np.abs(p["lin_jac"] - ref).max() > 1e-6 * np.abs(ref).max()]
    if bad:
        d = np.abs(ps[bad[0]]["lin_jac"] - ref)
        rows = np.where(d.max(axis=1) > 1e-9 * np.abs(ref).max())[0]; cols = np.where(d.max(axis=0) > 1e-9 * np.abs(ref).max())[0]
        out["bad_calls"] = bad; out["rows"] = [int(rows.min()), int(rows.max()), len(rows)]; out["cols"] = [int(cols.min()), int(cols.max()), len(cols)]
        out["res_diff"] = float(np.abs(ps[bad[0]]["lin_res"] - ps[-1]["lin_res"]).max())
except Exception as e:
    out["error"] = str(e)[-100:]
print(json.dumps(out))
