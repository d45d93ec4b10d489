"""One workload for rocprofv3 --kernel-trace --stats: 30 associations of one C2 scan (64k queries) in the given K2 mode."""
import sys
sys.path.insert(0, ".")
from glio_amd import capi, synth
from glio_amd.capi import lidar_pose
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
capi.load().glio_debug_set_knn_mode(mode)
win = synth.make_window(W=2, pts_per_scan=65536, seed=synth.SEED_BASE)
ctx = capi.Context(win.opts)
ctx.set_map(win.map_pts)
ctx.set_scan(0, win.scans[0])
q, t = lidar_pose(win.opts, win.init.quat[0], win.init.trans[0])
for _ in range(30):
    ctx.associate_resident(0, q, t)
print("us", ctx.time_kernel(capi.KERNEL_ASSOCIATE, 20) * 1e3)
ctx.close()
