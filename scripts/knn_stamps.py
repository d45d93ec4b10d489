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
st = (C.c_longlong * 8)(); capi.load().glio_debug_knn_stamps(st)
v = [x / 100.0 for x in st]
print("k_knn5_tile, workgroup 0 wavefront 0 (us): probe+prefix %.2f, staging %.2f, scan %.2f, exact re-ranking %.2f, merge+store %.2f, units %d" % (v[0], v[1], v[2], v[3], v[4], st[5]))
print("kernel:", ctx.time_kernel(capi.KERNEL_ASSOCIATE, 10) * 1e3, "us (whole association)")
