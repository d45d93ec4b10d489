"""Front-end scan-to-map odometry: solver path, per-kernel times and the whole update at 64k points."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, time
from glio_amd import capi, odometry, synth
win = synth.make_window(W=1, pts_per_scan=65536, seed=synth.SEED_BASE + 71, perturb=(0.15, 0.8, 0.0), scan_radius=30.0)
scan = win.scans[0].copy(); scan[:, :3] -= np.array(win.opts.t_lb, np.float32)
pose0 = np.r_[win.init.quat[0], win.init.trans[0]]
o = odometry.frontend_opts(len(scan), len(win.map_pts))
ctx = capi.Context(o)
odo = odometry.ScanToMapOdometry(ctx); odo.set_map(win.map_pts)
pose, rounds = odo.update(scan, pose0, match_cnt=2)
print("path", capi.load().glio_debug_solver_path(ctx._h), [r[0].iterations for r in rounds])
print("tr_step us", ctx.time_kernel(capi.KERNEL_TR_STEP, 20) * 1e3, "full_linearize us", ctx.time_kernel(capi.KERNEL_FULL_LINEARIZE, 20) * 1e3, "lin_all", ctx.time_kernel(capi.KERNEL_LINEARIZE_ALL, 20) * 1e3, "assoc us", ctx.time_kernel(capi.KERNEL_ASSOCIATE, 10) * 1e3)
t0 = time.perf_counter()
for _ in range(20): pose, rounds = odo.update(scan, pose0, match_cnt=2)
print("update ms", (time.perf_counter() - t0) / 20 * 1e3)
