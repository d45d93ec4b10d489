"""Where does the dense path go wrong at n = 414 (W = 22 with GNSS)?  (1) glio_linearize's H, g, cost vs the oracle's; (2) the in-kernel blocked
Cholesky (glio_debug_chol_solve) on the oracle's H + I vs numpy, at n = 376, 400, 414, 430."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from glio_amd import synth, capi
from oracle import pyoracle as po
W = 22
long = synth.make_window(W=W + 1, pts_per_scan=200, with_gnss=True, seed=synth.SEED_BASE + 93)
first = synth.sub_window(long, 0, W)
prob0 = po.Problem(first, synth.analytic_correspondences(first), use_gnss=False, use_prior=False)
st0 = first.init.copy(); st0.n_ddt = 0
sol0, _ = prob0.solve(st0)
win = synth.sub_window(long, 1, W); win.prior = prob0.marginalize(sol0)
corr = synth.analytic_correspondences(win)
st = win.init.copy()
prob = po.Problem(win, corr)
Ho, go, co = prob.linearize(st)
ctx = capi.Context(win.opts); capi.load().glio_debug_set_solver(ctx._h, 0)
ctx.load_window(win, corr)
Hh, gh, ch = ctx.linearize(st)
n = Ho.shape[0]
rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
print("n", n, "linearize: cost rel", abs(ch - co) / abs(co), "g rel", rel(gh, go), "H rel", rel(Hh, Ho), "max |dH|", float(np.abs(Hh - Ho).max()), flush=True)
bad = np.argwhere(np.abs(Hh - Ho) > 1e-6 * np.abs(Ho).max())
print("entries off:", len(bad), bad[:12].tolist(), flush=True)
rng = np.random.default_rng(3)
for m in (376, 400, 414, 430):
    if m > n:
        continue
    A = Ho[:m, :m] + np.eye(m)
    d = 1.0 / np.sqrt(np.diag(A)); A = A * d[:, None] * d[None, :]
    b = rng.normal(0, 1, m)
    L = np.tril(A).copy()
    x = np.zeros(m)
    rc = capi.load().glio_debug_chol_solve(ctx._h, m, L.ctypes.data_as(C.POINTER(C.c_double)), b.ctypes.data_as(C.POINTER(C.c_double)), x.ctypes.data_as(C.POINTER(C.c_double)))
    xs = np.linalg.solve(A, b)
    print("chol_solve n", m, "rc", rc, "rel err vs numpy", rel(x, xs), flush=True)
ctx.close()
