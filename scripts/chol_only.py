import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glio_amd import synth, capi
from glio_amd import ctypes_types as T
print("lib", capi.LIB_PATH)
for n in (8, 15, 16, 17, 33, 60, 150, 376):
    o = synth.default_opts(W=50, pts=64, map_pts=64, n_ddt=0)
    ctx = capi.Context(o)
    rng = np.random.default_rng(n)
    B = rng.normal(size=(n, n)); A = B @ B.T + n * np.eye(n); b = rng.normal(size=n)
    x = np.zeros(n)
    rc = capi.load().glio_debug_chol_solve(ctx._h, n, T.dptr(np.ascontiguousarray(A)), T.dptr(b), T.dptr(x))
    print("chol n", n, "rc", rc, "err", np.abs(x - np.linalg.solve(A, b)).max() / np.abs(x).max())
    ctx.close()
