import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from glio_amd import synth, capi
win = synth.make_window(W=20, pts_per_scan=65536, with_gnss=True, with_prior=True, seed=synth.SEED_BASE + 12)
corr = synth.analytic_correspondences(win)
ctx = capi.Context(win.opts)
ctx.load_window(win, corr)
ctx.linearize(win.init)
ctx.linearize(win.init)
st = (C.c_longlong * 320)()
capi.load().glio_debug_arrow_stamps(ctx._h, st)
v = np.array(list(st))[64:] / 100.0
W = 20
nu = 250 - W - 19 - 9
import collections
def stat(name, a):
    a = np.array(a)
    if len(a): print(f"{name:14s} n {len(a):4d} min {a.min():6.1f} mean {a.mean():6.1f} max {a.max():6.1f}")
stat("lidar reduce", v[:W]); stat("imu", v[W:W + 19])
# units: find count from context
import ctypes as C2
n_units = int(sum(1 for _ in win.dd)) + len(set((d.slot_i, d.slot_j, d.epoch) for d in win.dop))
stat("gnss units", v[W + 19:W + 19 + n_units]); stat("prior", v[W + 19 + n_units:W + 19 + n_units + 9])
print("n_units", n_units, "full_linearize", ctx.time_kernel(1, 30) * 1e3)
