import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glio_amd import synth, capi
from glio_amd import ctypes_types as T
np.set_printoptions(linewidth=200, precision=4)
print("lib", capi.LIB_PATH)
win = synth.make_window(W=4, pts_per_scan=600)
corr = synth.analytic_correspondences(win)
ctx = capi.Context(win.opts)
ctx.load_window(win, corr, use_gnss=False, use_prior=False, use_imu=False)
st = win.init.copy()
H, g, c = ctx.linearize(st)
n = len(g)
s = st.copy(); cs = s.c(); summ = T.GlioSummary()
rc = capi.load().glio_solve(ctx._h, C.byref(cs), C.byref(summ))
print("rc", rc, summ.as_dict(), capi.load().glio_last_error().decode())
scale = 1.0 / (1.0 + np.sqrt(np.diag(H)))
Hs = scale[:, None] * H * scale[None, :]
d = np.sqrt(np.clip(np.diag(Hs), 1e-6, 1e32))
exp = {0: scale, 1: d, 2: scale * g / d}
for k, nm in [(0, "scale"), (1, "diag"), (2, "grad"), (8, "y_fwd"), (9, "y_back")]:
    v = np.zeros(n); capi.load().glio_debug_read_vec(ctx._h, k, T.dptr(v), n)
    print(nm, "finite", np.isfinite(v).all(), v[:8], "expected", exp.get(k, np.zeros(8))[:8])
