"""Marginalization of the oldest keyframe, the three-launch form against the one-workgroup form (GLIO_MARG_SPLIT=0): prints a hash of the prior it produces
(J0, r0, kept blocks) for a first window (rank-deficient Amm: the eigen-decomposition path) and a steady-state window (fast inverse), and the device time."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glio_amd import synth, capi
W, pts = int(os.environ.get("MS_W", "20")), int(os.environ.get("MS_PTS", "8192"))
stream = synth.make_window(W=W + 1, pts_per_scan=pts, with_gnss=True, seed=synth.SEED_BASE + 12)
out = {"split": os.environ.get("GLIO_MARG_SPLIT", "1"), "W": W}
first = synth.sub_window(stream, 0, W)
c0 = capi.Context(first.opts); c0.load_window(first, synth.analytic_correspondences(first))
s0, _ = c0.solve(first.init); prior = c0.marginalize(s0)
def h(p):
    m = hashlib.sha256()
    for k in ("lin_jac", "lin_res"):
        m.update(np.ascontiguousarray(p[k]).tobytes())
    return m.hexdigest()[:16]
out["first_window"] = h(prior); c0.close()
win = synth.sub_window(stream, 1, W); win.prior = prior
ctx = capi.Context(win.opts); ctx.load_window(win, synth.analytic_correspondences(win))
sol, _ = ctx.solve(win.init)
p2 = ctx.marginalize(sol)
out["steady_window"] = h(p2)
out["marginalize_us"] = round(ctx.time_kernel(capi.KERNEL_MARGINALIZE, 20) * 1e3, 2)
print(json.dumps(out))
