"""Host-side cost of the per-keyframe calls (C entry points, pre-marshalled arguments): scripts/README.md"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from glio_amd import capi, synth
from glio_amd.capi import lidar_pose

W = 20
win = synth.make_window(W=W, pts_per_scan=65536, with_gnss=True, with_prior=True)
ctx = capi.Context(win.opts)
ctx.set_map(win.map_pts)
for s in range(W):
    ctx.set_scan(s, win.scans[s])
poses = [lidar_pose(win.opts, win.init.quat[s], win.init.trans[s]) for s in range(W)]
q2s = np.array([p[0] for p in poses]); t2s = np.array([p[1] for p in poses])
ctx.associate_window(q2s, t2s)
m_imu = ctx.marshal_imu(win.preints); m_gnss = ctx.marshal_gnss(win.frame, win.dd, win.dop)
ctx.set_imu_marshalled(m_imu); ctx.set_gnss_marshalled(m_gnss); ctx.set_prior(win.prior)
def t(f, n=20):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
print("set_imu  ms", round(t(lambda: ctx.set_imu_marshalled(m_imu)), 4))
print("set_gnss ms", round(t(lambda: ctx.set_gnss_marshalled(m_gnss)), 4))
print("marshal imu ms", round(t(lambda: ctx.marshal_imu(win.preints)), 4), "gnss", round(t(lambda: ctx.marshal_gnss(win.frame, win.dd, win.dop)), 4))
print("associate_window ms", round(t(lambda: ctx.associate_window(q2s, t2s)), 4))
print("solve ms", round(t(lambda: ctx.solve(win.init)), 4))
sol, summ = ctx.solve(win.init)
print("iterations", summ.iterations)
print("marginalize_keep ms", round(t(lambda: ctx.marginalize_keep(sol), 5), 4))
