"""The grid index of the CPU baseline (orc_set_assoc_grid, bench.py only) returns the brute force's records bit for bit."""
import numpy as np

from glio_amd import synth
from oracle import pyoracle as po


def test_grid_indexed_association_equals_brute_force():
    po.lib().orc_set_assoc_grid.restype = None
    for seed, pts in ((77, 3000), (78, 1500)):
        win = synth.make_window(W=2, pts_per_scan=pts, seed=synth.SEED_BASE + seed, perturb=(0.2, 1.0, 0.0))
        for s in range(2):
            q2, t2 = po.lidar_pose_for_association(win.opts, win.init.quat[s], win.init.trans[s])
            want = po.associate(win.opts, win.map_pts, win.scans[s], q2, t2)
            po.lib().orc_set_assoc_grid(1)
            try:
                got = po.associate(win.opts, win.map_pts, win.scans[s], q2, t2)
            finally:
                po.lib().orc_set_assoc_grid(0)
            assert len(want[2]) > 0.5 * pts
            assert all(np.array_equal(a, b) for a, b in zip(want[:3], got[:3]))
