"""Per-workgroup start / end times of k_knn5_tile on one C2 scan (stamped build through GLIO_HIP_LIB): where the kernel's duration goes --
dispatch ramp, the workgroups' own durations, the tail."""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
from glio_amd import synth, capi
from glio_amd.capi import lidar_pose
win = synth.make_window(W=2, pts_per_scan=65536, seed=synth.SEED_BASE + 12)
ctx = capi.Context(win.opts); ctx.set_map(win.map_pts)
ctx.set_scan(0, win.scans[0])
q, t = lidar_pose(win.opts, win.init.quat[0], win.init.trans[0])
for _ in range(3): ctx.associate_resident(0, q, t)
n = 4096
buf = (C.c_longlong * (12 * n))()
assert capi.load().glio_debug_knn_wg(buf, n) == 0
a = np.array(buf[:], np.int64).reshape(n, 12)
work = a[:, 2] > 0
t0 = a[:, 0].min()
st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0
dur = en - st
print("workgroups", n, "with work", int(work.sum()))
print("start (us): working min/median/max %.2f %.2f %.2f   all max %.2f" % (st[work].min(), np.median(st[work]), st[work].max(), st.max()))
print("duration of working workgroups (us): min/median/p90/max %.2f %.2f %.2f %.2f" % (dur[work].min(), np.median(dur[work]), np.percentile(dur[work], 90), dur[work].max()))
print("end (us): median %.2f  p90 %.2f  max %.2f" % (np.median(en[work]), np.percentile(en[work], 90), en[work].max()))
print("duration of empty workgroups median %.2f" % (np.median(dur[~work]) if (~work).any() else 0))
cand = a[work, 2]
print("candidates of the last unit: median %d max %d; corr(duration, candidates) %.2f" % (np.median(cand), cand.max(), np.corrcoef(dur[work], cand)[0, 1]))
xcc = a[:, 3] >> 32
print("workgroups per XCC (working):", np.bincount(xcc[work].astype(int), minlength=8).tolist())
order = np.argsort(st)
print("start time by dispatch order, every 64th working workgroup:", np.round(st[work][::64], 2).tolist())

ph = a[:, 7:12] / 100.0
slow = work & (dur > np.percentile(dur[work], 95))
typ = work & (dur < np.percentile(dur[work], 60)) & (dur > np.percentile(dur[work], 40))
for name, m in (("typical (p40..p60)", typ), ("slowest 5 %", slow)):
    print(name, "n", int(m.sum()), "dur %.2f" % dur[m].mean(), "| probe %.2f staging %.2f scan %.2f rerank %.2f merge %.2f | cand(wave max) %.0f chunks %.2f unsafe %.2f probe steps(lane 0) %.2f"
          % (*ph[m].mean(0), a[m, 2].mean(), a[m, 4].mean(), a[m, 5].mean(), a[m, 6].mean()))
cu = a[:, 3] & 0xffffffff
print("slowest 10 workgroups:", [(int(i), round(float(dur[i]), 1), int(a[i, 2]), int(a[i, 4]), int(a[i, 5]), [round(float(x), 1) for x in ph[i]]) for i in np.argsort(-dur * work)[:10]])
