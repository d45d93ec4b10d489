"""BASELINE config C3 at its stated size: a 131 072-point scan associated against a map of more than 10^6 points
(findCorrespondingSurfFeatures, GLIO/src/Estimator.cpp:3633-3708; the map is `surf_local_map_ds`, voxel-filtered at 0.4 m,
so a map of that size is one of large extent -- synth.tiled_map).  The whole scan is checked against the oracle's
brute-force association (1.5e11 float distance evaluations, spread over the host cores with OpenMP): kept set, order,
float records and double scores bit for bit, and the neighbour indices of every query that passes the radius gate.  If the
box offers too few cores for the full scan to finish in about a minute, a stratified sample of >= 32 768 queries is checked
instead and the test says so."""
import os
import time

import numpy as np
import pytest

from glio_amd import ctypes_types as T
from glio_amd import synth

pytestmark = pytest.mark.gpu

C3_QUERIES = 131072
C3_TILES = 24


@pytest.fixture(scope="module")
def c3_case():
    win = synth.make_window(W=1, pts_per_scan=C3_QUERIES, seed=synth.SEED_BASE + 7)
    big = synth.tiled_map(win.map_pts, C3_TILES)
    assert len(big) >= 1_000_000
    return win, big


def test_c3_full_scan_bit_exact(c3_case):
    from glio_amd import capi
    from oracle import pyoracle as po
    win, big = c3_case
    o = synth.default_opts(1, pts=C3_QUERIES, map_pts=len(big))
    ctx = capi.Context(o)
    ctx.set_map(big)
    q2, t2 = po.lidar_pose_for_association(o, win.init.quat[0], win.init.trans[0])
    scan = win.scans[0]
    cnt = ctx.associate(0, scan, q2, t2)
    hp, hpl, hsc = ctx.get_correspondences(0)
    hnn = np.zeros((len(scan), 5), np.int32)
    capi.load().glio_debug_last_nn(ctx._h, T.iptr(hnn), len(hnn))
    assert cnt == len(hsc) and cnt > 0.5 * len(scan)

    threads = min(128, len(os.sched_getaffinity(0)))
    est_s = len(scan) * len(big) / (threads * 5e8)
    if est_s <= 90.0:
        sel = np.arange(len(scan))
        what = "full scan"
    else:                                   # stratified: every 4th query, 32 768 of them
        sel = np.arange(0, len(scan), 4)
        what = f"stratified sample of {len(sel)} queries ({threads} host threads: the full scan would take ~{est_s:.0f} s)"
    t0 = time.time()
    pts, pl, sc, src, nn = po.associate(o, big, np.ascontiguousarray(scan[sel]), q2, t2, want_nn=True, threads=threads)
    print(f"C3 oracle: {what}, {len(big)} map points, {threads} threads, {time.time() - t0:.1f} s")
    src = sel[src]                          # indices into the full scan
    # the device's kept set restricted to the checked queries, in order
    view = scan.view(np.uint32).reshape(len(scan), 4)
    if len(sel) == len(scan):
        assert cnt == len(sc)
        assert np.array_equal(hp, pts)
        assert np.array_equal(hpl.view(np.uint32), pl.view(np.uint32))
        assert np.array_equal(hsc, sc)
    else:
        # kept records of the full scan carry the scan point itself: find which of them belong to the sample
        keys = {tuple(r): i for i, r in enumerate(map(tuple, view))}
        hidx = np.array([keys[tuple(r)] for r in hp.view(np.uint32).reshape(len(hp), 4)])
        assert np.all(np.diff(hidx) > 0)                       # scan order preserved
        m = np.isin(hidx, sel)
        assert np.array_equal(hidx[m], src)
        assert np.array_equal(hpl[m].view(np.uint32), pl.view(np.uint32))
        assert np.array_equal(hsc[m], sc)
    gate = hnn[sel][:, 4] >= 0
    assert gate.sum() > 0.5 * len(sel)
    assert np.array_equal(hnn[sel][gate], nn[gate])
    ctx.close()


def test_tiled_map_keeps_the_street(c3_case):
    """The tiles are further apart than the search radius: associating against the tiled map keeps exactly the queries the
    single street keeps (indices differ, records do not)."""
    from glio_amd import capi
    from oracle import pyoracle as po
    win, big = c3_case
    o = synth.default_opts(1, pts=C3_QUERIES, map_pts=len(big))
    q2, t2 = po.lidar_pose_for_association(o, win.init.quat[0], win.init.trans[0])
    scan = np.ascontiguousarray(win.scans[0][:20000])
    a = capi.Context(o); a.set_map(big); na = a.associate(0, scan, q2, t2); ra = a.get_correspondences(0); a.close()
    b = capi.Context(o); b.set_map(win.map_pts); nb = b.associate(0, scan, q2, t2); rb = b.get_correspondences(0); b.close()
    assert na == nb and all(np.array_equal(x, y) for x, y in zip(ra, rb))


def test_c3_dense_candidate_set_bit_exact():
    """The same size of problem on the densest candidate set a voxel-filtered planar map offers (synth.sheets_map: sheets 0.8 m
    apart, inside each other's search radius): 65 536 queries against 1 048 352 map points, the whole scan against the oracle's
    brute-force association, bit for bit, neighbour indices included."""
    from glio_amd import capi
    from oracle import pyoracle as po
    big, z = synth.sheets_map()
    scan = synth.sheets_scan(65536, z)
    o = synth.default_opts(1, pts=len(scan), map_pts=len(big))
    q2, t2 = np.array([1.0, 0, 0, 0]), np.zeros(3)
    ctx = capi.Context(o)
    ctx.set_map(big)
    cnt = ctx.associate(0, scan, q2, t2)
    hp, hpl, hsc = ctx.get_correspondences(0)
    hnn = np.zeros((len(scan), 5), np.int32)
    capi.load().glio_debug_last_nn(ctx._h, T.iptr(hnn), len(hnn))
    threads = min(128, len(os.sched_getaffinity(0)))
    sel = np.arange(len(scan)) if len(scan) * len(big) / (threads * 5e8) <= 90.0 else np.arange(0, len(scan), 8)
    pts, pl, sc, src, nn = po.associate(o, big, np.ascontiguousarray(scan[sel]), q2, t2, want_nn=True, threads=threads)
    assert cnt > 0.8 * len(scan), cnt
    if len(sel) == len(scan):
        assert cnt == len(sc)
        assert np.array_equal(hp, pts) and np.array_equal(hpl.view(np.uint32), pl.view(np.uint32)) and np.array_equal(hsc, sc)
    gate = hnn[sel][:, 4] >= 0
    assert gate.sum() > 0.8 * len(sel)
    assert np.array_equal(hnn[sel][gate], nn[gate])
    # a candidate set several sheets deep: the five neighbours of a query all lie on its own sheet although the sheets above and
    # below are inside the search radius (the exact 5-NN is what keeps the plane fit meaningful here)
    zq = scan[sel][gate][:, 2:3]
    assert np.abs(big[nn[gate]][:, :, 2] - zq).max() < 0.2
    ctx.close()
