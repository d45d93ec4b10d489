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
def stat(name, a):
    a = np.array(a)
    if len(a): print(f"{name:14s} n {len(a):4d} min {a.min():6.1f} mean {a.mean():6.1f} max {a.max():6.1f}")
stat("imu", v[:19]); stat("gnss", v[19:38]); stat("prior", v[38:38 + 25])
print("prior blocks", v[38:47].round(1))
print("full_linearize", ctx.time_kernel(1, 30) * 1e3, " stream_read", ctx.time_kernel(6, 50) * 1e3, " k3", ctx.time_kernel(0, 50) * 1e3)
g = np.array(list(st))[264:272]
print("gnss block 0 stamps (us): dd loads+compute, sync, dd rest, dop compute, sync, dop reduce, scatter:", [round((g[k + 1] - g[k]) / 100.0, 2) for k in range(7)])
im = np.array(list(st))[64 + 190:64 + 196]
print("imu block 0 stamps (us): common + residual, global-Jacobian roles, local parameterisation, whitening, J^T J + stores:", [round((im[k + 1] - im[k]) / 100.0, 2) for k in range(5)])
print("linearize_all", ctx.time_kernel(7, 30) * 1e3)
capi.load().glio_debug_arrow_stamps(ctx._h, st)
raw = list(st)
t0 = raw[64 + 197]
k3 = [(raw[64 + 208 + k] - t0) / 100.0 for k in range(32)]
print("inside k_linearize_all, us after workgroup 0 started: small-factor workgroups done (max of the per-workgroup durations above); K3 workgroups: last done %.2f, slots %s" % (max(k3), [round(x, 1) for x in sorted(k3)[-6:]]))
