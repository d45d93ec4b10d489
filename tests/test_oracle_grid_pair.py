"""The grid index of the CPU baseline gives the records of the exhaustive pair association (oracle/orc_assoc.c: orc_set_assoc_grid)."""
import numpy as np

from glio_amd import synth
from oracle import pyoracle as po


def test_pair_association_grid_equals_exhaustive():
    win = synth.make_window(W=2, pts_per_scan=1500, seed=synth.SEED_BASE + 55, perturb=(0.03, 0.2, 0.0), scan_radius=12.0, map_density=1.0)
    tlb = np.array(win.opts.t_lb, np.float32)
    sc = []
    for s in range(2):
        c = win.scans[s].copy(); c[:, :3] -= tlb
        sc.append(np.ascontiguousarray(c))
    poses = np.c_[win.init.trans, win.init.quat]
    po.lib().orc_set_assoc_grid.restype = None
    a = po.associate_pair(sc[0], poses[0], sc[1], poses[1])
    po.lib().orc_set_assoc_grid(1)
    try:
        b = po.associate_pair(sc[0], poses[0], sc[1], poses[1])
    finally:
        po.lib().orc_set_assoc_grid(0)
    assert len(a[2]) > 300
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
