import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def small_window():
    from glio_amd import synth
    return synth.make_window(W=4, pts_per_scan=600, with_gnss=True, with_prior=True, seed=synth.SEED_BASE)


@pytest.fixture(scope="session")
def small_corr(small_window):
    """Correspondences of the small window from the oracle's brute-force association."""
    from oracle import pyoracle as po
    win = small_window
    corr = []
    for s in range(win.W):
        q2, t2 = po.lidar_pose_for_association(win.opts, win.init.quat[s], win.init.trans[s])
        pts, pl, sc, _ = po.associate(win.opts, win.map_pts, win.scans[s], q2, t2)
        corr.append((pts, pl, sc))
    return corr
