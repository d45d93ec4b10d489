"""Dense (mode 0) vs chain (mode 1) vs oracle iteration counts over window sizes: a probe of the dense fallback at larger n."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glio_amd import synth, capi
from oracle import pyoracle as po
for W, gnss in ((8, True), (12, True), (16, True), (18, True), (20, True), (22, True), (22, False), (24, False)):
    long = synth.make_window(W=W + 1, pts_per_scan=100, with_gnss=True, seed=synth.SEED_BASE + 93)
    first = synth.sub_window(long, 0, W)
    prob0 = po.Problem(first, synth.analytic_correspondences(first), use_gnss=False, use_prior=False)
    st0 = first.init.copy(); st0.n_ddt = 0
    sol0, _ = prob0.solve(st0)
    win = synth.sub_window(long, 1, W); win.prior = prob0.marginalize(sol0)
    corr = synth.analytic_correspondences(win)
    st = win.init.copy()
    if not gnss: st.n_ddt = 0
    so, mo = po.Problem(win, corr, use_gnss=gnss).solve(st.copy())
    out = []
    for mode in (0, 1):
        ctx = capi.Context(win.opts); capi.load().glio_debug_set_solver(ctx._h, mode)
        ctx.load_window(win, corr, use_gnss=gnss)
        s, m = ctx.solve(st.copy())
        out.append((capi.load().glio_debug_solver_path(ctx._h), m.iterations, m.termination, float(np.abs(s.trans - so.trans).max())))
        ctx.close()
    print("W", W, "gnss", gnss, "n", 15 * W + st.n_ddt, "oracle its", mo.iterations, "dense (path, its, term, |dt| vs oracle)", out[0], "structured", out[1], flush=True)
