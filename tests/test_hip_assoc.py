"""GPU parity of K1/K2 (voxel-hash exact 5-NN + plane fit + gates + compaction) against the oracle's
brute-force restatement of findCorrespondingSurfFeatures.  Bit-exact: same neighbour sets (ties broken by
map index), identical float plane records and double scores, identical order."""
import numpy as np
import pytest

from glio_amd import ctypes_types as T
from glio_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from glio_amd import capi
    assert capi.device_count() >= 1
    return capi


@pytest.fixture(scope="module")
def po():
    from oracle import pyoracle
    return pyoracle


def _check_slot(hip, po, ctx, win, s, scan, q2, t2):
    pts, pl, sc, src, nn = po.associate(win.opts, win.map_pts, scan, q2, t2, want_nn=True)
    cnt = ctx.associate(s, scan, q2, t2)
    hp, hpl, hsc = ctx.get_correspondences(s)
    assert cnt == len(sc)
    assert np.array_equal(hp, pts)
    assert np.array_equal(hpl.view(np.uint32), pl.view(np.uint32))
    assert np.array_equal(hsc, sc)
    hnn = np.zeros((len(scan), 5), np.int32)
    hip.load().glio_debug_last_nn(ctx._h, T.iptr(hnn), len(hnn))
    gate = hnn[:, 4] >= 0
    assert np.array_equal(hnn[gate], nn[gate])
    return cnt


def test_association_bit_exact_small(hip, po):
    win = synth.make_window(W=3, pts_per_scan=4000, seed=synth.SEED_BASE + 5)
    ctx = hip.Context(win.opts)
    ctx.set_map(win.map_pts)
    for s in range(win.W):
        q2, t2 = po.lidar_pose_for_association(win.opts, win.init.quat[s], win.init.trans[s])
        cnt = _check_slot(hip, po, ctx, win, s, win.scans[s], q2, t2)
        assert cnt > 0.9 * len(win.scans[s])
    ctx.close()


def test_association_edge_cases(hip, po):
    """Ragged / degenerate inputs: a single query, queries far outside the map (5th neighbour gate fails),
    a map smaller than five points (nothing can be accepted), re-association with a new pose."""
    win = synth.make_window(W=2, pts_per_scan=777, seed=synth.SEED_BASE + 6)
    ctx = hip.Context(win.opts)
    ctx.set_map(win.map_pts)
    q2, t2 = po.lidar_pose_for_association(win.opts, win.init.quat[0], win.init.trans[0])
    _check_slot(hip, po, ctx, win, 0, win.scans[0][:1].copy(), q2, t2)
    far = win.scans[0].copy()
    far[:, :3] += 500.0
    assert _check_slot(hip, po, ctx, win, 0, far, q2, t2) == 0
    _check_slot(hip, po, ctx, win, 1, win.scans[1], q2, t2 + np.array([0.3, -0.2, 0.05]))   # a different pose
    tiny = win.map_pts[:4].copy()
    ctx2 = hip.Context(win.opts)
    ctx2.set_map(tiny)
    assert ctx2.associate(0, win.scans[0], q2, t2) == 0
    ctx.close(); ctx2.close()


def test_association_full_scan_properties(hip, po):
    """BASELINE config C3 shape: a 131k-point scan against the local map.  Checked against the oracle on
    a 4k-query sample (the brute-force oracle is O(N M)) and through size-independent properties on the
    full scan: order preservation, kept records are a subsequence of the scan, weights inside (0.3, 1]."""
    win = synth.make_window(W=1, pts_per_scan=131072, seed=synth.SEED_BASE + 7)
    ctx = hip.Context(win.opts)
    ctx.set_map(win.map_pts)
    q2, t2 = po.lidar_pose_for_association(win.opts, win.init.quat[0], win.init.trans[0])
    scan = win.scans[0]
    cnt = ctx.associate(0, scan, q2, t2)
    hp, hpl, hsc = ctx.get_correspondences(0)
    assert cnt == len(hsc) and cnt > 0.5 * len(scan)
    w = hsc / win.opts.lidar_const
    assert w.min() > win.opts.weight_gate and w.max() <= 1.0
    # kept points appear in scan order: match them greedily
    pos = 0
    view = scan.view(np.uint32).reshape(len(scan), 4)
    hv = hp.view(np.uint32).reshape(len(hp), 4)
    keys = {tuple(r): i for i, r in enumerate(map(tuple, view))}
    idx = np.array([keys[tuple(r)] for r in hv[:2000]])
    assert np.all(np.diff(idx) > 0)
    sample = np.ascontiguousarray(scan[1000:5000])
    _check_slot(hip, po, ctx, win, 0, sample, q2, t2)
    ctx.close()


def test_solve_from_gpu_association_matches_oracle(hip, po):
    """End-to-end hot path: K1/K2 on the GPU feed K3..K7; the oracle runs on its own brute-force
    association of the same buffers."""
    win = synth.make_window(W=4, pts_per_scan=3000, with_gnss=True, with_prior=True, seed=synth.SEED_BASE + 8)
    ctx = hip.Context(win.opts)
    ctx.set_map(win.map_pts)
    corr = []
    for s in range(win.W):
        q2, t2 = po.lidar_pose_for_association(win.opts, win.init.quat[s], win.init.trans[s])
        ctx.associate(s, win.scans[s], q2, t2)
        pts, pl, sc, _ = po.associate(win.opts, win.map_pts, win.scans[s], q2, t2)
        corr.append((pts, pl, sc))
    ctx.load_window(win, None)
    sh, summ_h = ctx.solve(win.init)
    so, summ_o = po.Problem(win, corr).solve(win.init)
    assert summ_h.iterations == summ_o.iterations
    assert np.linalg.norm(sh.trans - so.trans, axis=1).max() < 1e-9
    ctx.close()


def test_window_association_equals_per_slot(hip, small_window):
    """glio_associate_window (one call, one sync) leaves exactly what W glio_associate_resident calls leave."""
    win = small_window
    ctx = hip.Context(win.opts)
    ctx.set_map(win.map_pts)
    poses = [hip.lidar_pose(win.opts, win.init.quat[s], win.init.trans[s]) for s in range(win.W)]
    per = []
    for s in range(win.W):
        n = ctx.associate(s, win.scans[s], *poses[s])
        per.append((n,) + tuple(a.copy() for a in ctx.get_correspondences(s)))
    cnt = ctx.associate_window(np.array([p[0] for p in poses]), np.array([p[1] for p in poses]))
    for s in range(win.W):
        got = ctx.get_correspondences(s)
        assert cnt[s] == per[s][0] and all(np.array_equal(a, b) for a, b in zip(got, per[s][1:]))
    ctx.close()


def test_feature_selection_gathers_on_device(hip, small_window):
    """featureSelection (Estimator.cpp:3894-3992): seeded draws on the host, gather on the device."""
    from glio_amd import sliding
    win = small_window
    ctx = hip.Context(win.opts)
    ctx.set_map(win.map_pts)
    q2, t2 = hip.lidar_pose(win.opts, win.init.quat[0], win.init.trans[0])
    n = ctx.associate(0, win.scans[0], q2, t2)
    full = [a.copy() for a in ctx.get_correspondences(0)]
    assert sliding.feature_selection(ctx, 0, n, n + 5, np.random.default_rng(1)) == n          # too few candidates: early return, keep all
    rng = np.random.default_rng(7)
    sel = sliding.feature_selection_draws(n, 100, np.random.default_rng(7))
    assert len(sel) == 100 and len(set(sel.tolist())) == 100
    assert sliding.feature_selection(ctx, 0, n, 100, rng) == 100
    got = ctx.get_correspondences(0)
    assert all(np.array_equal(g, f[sel]) for g, f in zip(got, full))
    assert sliding.feature_selection(ctx, 0, 100, 10, rng, random_select=False) == 0            # random_select false empties the set
    assert len(ctx.get_correspondences(0)[2]) == 0
    ctx.close()


def test_tiled_search_exact_ties_and_dense_cells(hip, po):
    """The tiled neighbour search ranks on truncated 32-bit keys and restores the exact (float distance, map index) order
    afterwards.  Two inputs built to defeat a sloppy version of that: (a) a 0.25 m lattice map with queries ON lattice points
    and cell centres -- dozens of exactly equal distances per query, so the selection's safety test fails and the exact rescan
    must run; (b) 60 map points per voxel-hash cell (> 128 candidates per unit: several staging chunks)."""
    win = synth.make_window(W=1, pts_per_scan=512, seed=synth.SEED_BASE + 9)
    rng = np.random.default_rng(5)
    g = np.arange(-16, 17) * 0.25
    lat = np.stack(np.meshgrid(g + 10.0, g, [0.0, 0.25], indexing="ij"), -1).reshape(-1, 3)
    lat_map = np.zeros((len(lat), 4), np.float32); lat_map[:, :3] = lat[rng.permutation(len(lat))]
    qs = np.zeros((1500, 4), np.float32)
    qs[:500, :3] = lat[rng.integers(0, len(lat), 500)]                       # on lattice points
    qs[500:1000, :3] = lat[rng.integers(0, len(lat), 500)] + 0.125          # cell centres: 8 equidistant neighbours
    qs[1000:, :3] = lat[rng.integers(0, len(lat), 500)] + rng.normal(0, 0.3, (500, 3))
    ident_q, ident_t = np.array([1.0, 0, 0, 0]), np.zeros(3)
    dense = np.zeros((40000, 4), np.float32)
    dense[:, :3] = rng.uniform([5, -4, -0.05], [13, 4, 0.05], (40000, 3))   # a 8 x 8 m slab, 625 points per m^2
    dq = np.zeros((3000, 4), np.float32); dq[:, :3] = rng.uniform([6, -3, -0.3], [12, 3, 0.3], (3000, 3))
    # (c) 5003 queries inside ONE voxel-hash cell (a tile of 1024 queries with a single key: 64 units of the same cell)
    cl = np.zeros((5003, 4), np.float32); cl[:, :3] = rng.uniform([8.0, 0.1, -0.2], [8.9, 1.0, 0.2], (5003, 3))
    lib = hip.load()
    for mode in (0, 1, 2, 3):
        lib.glio_debug_set_knn_mode(mode)
        try:
            for m, q in ((lat_map, qs), (dense, dq), (dense, cl)):
                o = synth.default_opts(1, pts=len(q), map_pts=len(m))
                w2 = type("W", (), {"opts": o, "map_pts": m})
                ctx = hip.Context(o)
                ctx.set_map(m)
                _check_slot(hip, po, ctx, w2, 0, q, ident_q, ident_t)
                ctx.close()
        finally:
            lib.glio_debug_set_knn_mode(0)


def test_both_search_modes_give_identical_records(hip):
    """Tiled (default) and one-group-per-query searches on a full 64k scan: identical compacted records."""
    win = synth.make_window(W=2, pts_per_scan=65536, seed=synth.SEED_BASE + 10)
    lib = hip.load()
    out = []
    for mode in (0, 1, 2, 3):
        lib.glio_debug_set_knn_mode(mode)
        try:
            ctx = hip.Context(win.opts)
            ctx.set_map(win.map_pts)
            q2, t2 = hip.lidar_pose(win.opts, win.init.quat[0], win.init.trans[0])
            cnt = ctx.associate(0, win.scans[0], q2, t2)
            hp, hpl, hsc = ctx.get_correspondences(0)
            out.append((cnt, hp.copy(), hpl.copy(), hsc.copy()))
            ctx.close()
        finally:
            lib.glio_debug_set_knn_mode(0)
    assert out[0][0] == out[1][0] > 50000
    for a, b in zip(out[0][1:], out[1][1:]):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))


def test_window_association_ragged_slots_and_slide(hip, small_window):
    """The one-call window association with slots of very different sizes -- an empty slot, a single point, 1025 points (one full
    tile + one point of the query binning), a full scan -- and after glio_slide_window (the presorted copies move with the scans):
    every slot equals its own single-slot association."""
    win = small_window
    W = win.W
    sizes = [0, 1, 1025] + [len(win.scans[s]) for s in range(3, W)]
    scans = [np.ascontiguousarray(win.scans[s][:sizes[s]]) for s in range(W)]
    poses = [hip.lidar_pose(win.opts, win.init.quat[s], win.init.trans[s]) for s in range(W)]
    ref = hip.Context(win.opts); ref.set_map(win.map_pts)
    want = []
    for s in range(W):
        n = ref.associate(0, scans[s], *poses[s])
        want.append((n,) + tuple(a.copy() for a in ref.get_correspondences(0)))
    ref.close()
    ctx = hip.Context(win.opts); ctx.set_map(win.map_pts)
    for s in range(W):
        ctx.set_scan(s, scans[s])
    q2s, t2s = np.array([p[0] for p in poses]), np.array([p[1] for p in poses])
    cnt = ctx.associate_window(q2s, t2s)
    for s in range(W):
        got = ctx.get_correspondences(s)
        assert cnt[s] == want[s][0] and all(np.array_equal(a, b) for a, b in zip(got, want[s][1:])), s
    # slide: slot s takes the scan of slot s + 1; associate with the poses shifted the same way
    ctx.slide_window()
    ctx.set_scan(W - 1, scans[0])
    order = list(range(1, W)) + [0]
    cnt2 = ctx.associate_window(np.array([poses[k][0] for k in order]), np.array([poses[k][1] for k in order]))
    for s, k in enumerate(order):
        got = ctx.get_correspondences(s)
        assert cnt2[s] == want[k][0] and all(np.array_equal(a, b) for a, b in zip(got, want[k][1:])), (s, k)
    ctx.close()


def test_merged_window_grouping_orphan_segments(hip, small_window):
    """The one-call window association groups the queries of all slots by cell in a global table sized from the map.  With the table forced tiny
    (16 slots, kept under 3/4 full) nearly every (tile, cell) segment finds no slot and is laid out by itself from the end of the array: every slot
    must still equal the row-by-row search (mode 3), bit for bit."""
    win = small_window
    W = win.W
    lib = hip.load()
    poses = [hip.lidar_pose(win.opts, win.init.quat[s], win.init.trans[s]) for s in range(W)]
    q2s, t2s = np.array([p[0] for p in poses]), np.array([p[1] for p in poses])
    out = []
    for mode, gcap in ((3, 0), (0, 16), (0, 0)):
        lib.glio_debug_set_knn_mode(mode); lib.glio_debug_set_gbin_cap(gcap)
        try:
            ctx = hip.Context(win.opts); ctx.set_map(win.map_pts)
            for s in range(W):
                ctx.set_scan(s, win.scans[s])
            cnt = ctx.associate_window(q2s, t2s)
            out.append((list(cnt), [tuple(a.copy() for a in ctx.get_correspondences(s)) for s in range(W)]))
            cnt2 = ctx.associate_window(q2s, t2s)                 # the tables are left clean: a second call gives the same
            assert list(cnt2) == list(cnt)
            ctx.close()
        finally:
            lib.glio_debug_set_knn_mode(0); lib.glio_debug_set_gbin_cap(0)
    for k in (1, 2):
        assert out[k][0] == out[0][0] and sum(out[0][0]) > 0
        for s in range(W):
            assert all(np.array_equal(a.view(np.uint8), b.view(np.uint8)) for a, b in zip(out[k][1][s], out[0][1][s])), (k, s)


def test_asynchronous_window_association_gives_the_same_records():
    """glio_associate_window_async + glio_associate_window_counts (the host stages the factor tables in between) against the synchronous call: counts
    and every record identical; a solve right after the asynchronous call waits by itself."""
    from glio_amd import capi
    win = synth.make_window(W=4, pts_per_scan=6000, seed=synth.SEED_BASE + 23, with_gnss=True)
    poses = [capi.lidar_pose(win.opts, win.init.quat[s], win.init.trans[s]) for s in range(win.W)]
    q2s = np.array([p[0] for p in poses]); t2s = np.array([p[1] for p in poses])
    a, b = capi.Context(win.opts), capi.Context(win.opts)
    for c in (a, b):
        c.set_map(win.map_pts)
        for s in range(win.W):
            c.set_scan(s, win.scans[s])
    ca = a.associate_window(q2s, t2s)
    b.associate_window_async(q2s, t2s)
    b.set_imu(win.preints); b.set_gnss(win.frame, win.dd, win.dop)      # host work + uploads behind the searches
    cb = b.associate_window_counts()
    assert np.array_equal(ca, cb) and ca.sum() > 0
    for s in range(win.W):
        ra, rb = a.get_correspondences(s), b.get_correspondences(s)
        assert all(np.array_equal(x, y) for x, y in zip(ra, rb))
    # and without asking for the counts: the solve waits for the searches itself
    a.set_imu(win.preints); a.set_gnss(win.frame, win.dd, win.dop); a.set_prior(None); b.set_prior(None)
    b.associate_window_async(q2s, t2s)
    sa, ma = a.solve(win.init); sb, mb = b.solve(win.init)
    assert ma.iterations == mb.iterations and np.array_equal(sa.trans, sb.trans) and ma.n_lidar_residuals == mb.n_lidar_residuals
    # an asynchronous call followed by a synchronous one at OTHER poses: the later call's counts stand (the earlier one's must not come back)
    t3s = t2s + np.array([0.4, -0.3, 0.0])
    b.associate_window_async(q2s, t2s)
    c_sync = b.associate_window(q2s, t3s)
    assert np.array_equal(c_sync, a.associate_window(q2s, t3s)) and not np.array_equal(c_sync, ca)
    assert np.array_equal(b.associate_window_counts(), c_sync)
    a.close(); b.close()
