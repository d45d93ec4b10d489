import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from glio_amd import capi, synth
W=20
stream = synth.make_window(W=W + 1, pts_per_scan=2048, with_gnss=True, seed=synth.SEED_BASE + 12)
win = synth.sub_window(stream, 1, W)
ctx = capi.Context(win.opts)
m_imu = ctx.marshal_imu(win.preints); m_gnss = ctx.marshal_gnss(win.frame, win.dd, win.dop)
for name, fn in (("imu", lambda: ctx.set_imu_marshalled(m_imu)), ("gnss", lambda: ctx.set_gnss_marshalled(m_gnss))):
    fn()
    t0=time.perf_counter()
    for _ in range(20): fn()
    print(name, (time.perf_counter()-t0)/20*1e3, "ms", len(win.dd), len(win.dop))
k = [f.slot_i * W + f.slot_j for f in win.dd]; print("dd sorted", k == sorted(k))
k = [(f.slot_i * W + f.slot_j, f.epoch) for f in win.dop]; print("dop sorted", k == sorted(k))
for _ in range(3):
    t0=time.perf_counter()
    for _ in range(20): ctx.set_gnss_marshalled(m_gnss)
    print("gnss", (time.perf_counter()-t0)/20*1e3)
