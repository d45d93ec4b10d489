"""One workload for rocprofv3 --kernel-trace --stats: 10 one-call window associations (20 x 64k scans of C2); GLIO_HIP_LIB selects a library variant."""
import sys
sys.path.insert(0, ".")
import numpy as np
from glio_amd import capi, synth
from glio_amd.capi import lidar_pose
win = synth.make_window(W=20, pts_per_scan=65536, seed=synth.SEED_BASE, with_gnss=False)
ctx = capi.Context(win.opts)
ctx.set_map(win.map_pts)
for s in range(win.W):
    ctx.set_scan(s, win.scans[s])
poses = [lidar_pose(win.opts, win.init.quat[s], win.init.trans[s]) for s in range(win.W)]
q2s = np.array([p[0] for p in poses]); t2s = np.array([p[1] for p in poses])
for _ in range(10):
    ctx.associate_window(q2s, t2s)
ctx.close()
