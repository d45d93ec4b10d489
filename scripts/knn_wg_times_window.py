"""Per-workgroup phase stamps of k_knn5_tile (launch row 0) while the one-call WINDOW association (20 x 64k scans) loads the whole chip:
what a wavefront's phases cost under full load, next to scripts/knn_wg_times.py (one scan alone).  Stamped build through GLIO_HIP_LIB."""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
from glio_amd import synth, capi
from glio_amd.capi import lidar_pose
win = synth.make_window(W=20, pts_per_scan=65536, seed=synth.SEED_BASE, with_gnss=False)
ctx = capi.Context(win.opts); ctx.set_map(win.map_pts)
for s in range(win.W):
    ctx.set_scan(s, win.scans[s])
poses = [lidar_pose(win.opts, win.init.quat[s], win.init.trans[s]) for s in range(win.W)]
q2s = np.array([p[0] for p in poses]); t2s = np.array([p[1] for p in poses])
for _ in range(3):
    ctx.associate_window(q2s, t2s)
n = 4096
buf = (C.c_longlong * (12 * n))()
assert capi.load().glio_debug_knn_wg(buf, n) == 0
a = np.array(buf[:], np.int64).reshape(n, 12)
work = a[:, 2] > 0
t0 = a[work, 0].min()
st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0
dur = en - st
ph = a[:, 7:12] / 100.0
print("row 0 of the window launch: workgroups with work", int(work.sum()), "start min/median/max %.1f %.1f %.1f" % (st[work].min(), np.median(st[work]), st[work].max()),
      "end max %.1f" % en[work].max())
print("duration (us): median %.2f p90 %.2f max %.2f" % (np.median(dur[work]), np.percentile(dur[work], 90), dur[work].max()))
print("phases, mean over working workgroups (us): probe %.2f staging %.2f scan %.2f rerank %.2f merge+store %.2f | sum %.2f" % (*ph[work].mean(0), ph[work].mean(0).sum()))
