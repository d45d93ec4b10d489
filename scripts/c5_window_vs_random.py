"""The fp32 / MFMA form of K3 at the C5 shape on the WINDOW's own correspondences (synth.make_window, what bench.py's c5_stress times), launch
geometry sweep interleaved over several rounds: scripts/c5_launch.py --sweep uses random records, and the two disagree on the best geometry."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glio_amd import capi, synth
from glio_amd import ctypes_types as T
W, pts = 50, 262144
win = synth.make_window(W=W, pts_per_scan=pts, with_gnss=True, with_prior=False, seed=synth.SEED_BASE + 50, gnss_epoch_dt=0.4)
corr = synth.analytic_correspondences(win)
ctxs = {}
for prec in (1, 0):
    o = T.GlioOpts.from_buffer_copy(win.opts); o.lidar_precision = prec
    ctxs[prec] = capi.Context(o); ctxs[prec].load_window(win, corr); ctxs[prec].linearize(win.init, want_H=False)
geo = {1: (32, 40, 48, 60, 72, 80, 96, 120, 160, 240), 0: (12, 15, 20, 30)}
res = {}
for rnd in range(4):
    for prec in (1, 0):
        for bpk in geo[prec]:
            capi.load().glio_debug_set_k3(ctxs[prec]._h, bpk, 22 if prec else 24)
            ms = ctxs[prec].time_kernel(capi.KERNEL_LIDAR_LINEARIZE, 20)
            res.setdefault((prec, bpk), []).append(ms * 1e3)
for (prec, bpk), v in sorted(res.items()):
    print("precision", prec, "workgroups per keyframe", bpk, "us per launch:", [round(x, 1) for x in v], "median", round(float(np.median(v)), 1))
