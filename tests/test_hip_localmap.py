"""GPU: device-resident local map vs the oracle's restatement of buildLocalMapWithLandMark +
pcl::VoxelGrid (reference GLIO/src/Estimator.cpp:3529-3631).  Same voxel set in the same order; centroids agree to
float-accumulation noise (PCL / the oracle sum in float, the device in exact fixed point); the association run
against the device-built map equals the one against the uploaded oracle map up to those gate-level effects."""
import numpy as np
import pytest

from glio_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def stream():
    win = synth.make_window(W=7, pts_per_scan=5000, seed=synth.SEED_BASE + 81, scan_radius=25.0)
    tlb = np.array(win.opts.t_lb, np.float32)
    clouds = []
    for s in range(win.W):
        c = win.scans[s].copy(); c[:, :3] -= tlb
        clouds.append(np.ascontiguousarray(c))
    return win, clouds


def _oracle_map(po, clouds, poses, leaf):
    glob = [po.transform_cloud(c, q, t) for c, (q, t) in zip(clouds, poses)]
    return po.voxel_grid(np.vstack(glob), leaf)


def test_ring_voxelgrid_matches_oracle(stream):
    from glio_amd import capi
    from oracle import pyoracle as po
    win, clouds = stream
    o = synth.default_opts(1, pts=8192, map_pts=1 << 17)
    ctx = capi.Context(o)
    width, leaf = 4, 0.4
    ctx.localmap_config(width, leaf, 8192)
    poses = [(win.gt.quat[s], win.gt.trans[s]) for s in range(win.W)]
    for s in range(win.W):
        ctx.localmap_push(clouds[s], *poses[s])
        n = ctx.localmap_build()
        lo = max(0, s + 1 - width)                        # the ring keeps the last `width` keyframes
        ref, _ = _oracle_map(po, clouds[lo:s + 1], poses[lo:s + 1], leaf)
        got = ctx.localmap_read()
        assert n == len(ref) == len(got), f"keyframe {s}: {n} voxels vs oracle {len(ref)}"
        assert np.abs(got - ref).max() <= 2e-5, "centroids (ordered by voxel index)"
    # insert/evict is exact: a fresh context that only ever saw the last `width` keyframes holds the same map, bit for bit
    fresh = capi.Context(o)
    fresh.localmap_config(width, leaf, 8192)
    for s2 in range(win.W - width, win.W):
        fresh.localmap_push(clouds[s2], *poses[s2])
    fresh.localmap_build()
    assert np.array_equal(fresh.localmap_read(), ctx.localmap_read())
    fresh.close()
    # rebuilding is reproducible bit for bit (exact fixed-point sums, sorted output)
    a = ctx.localmap_read().copy()
    ctx.localmap_build()
    assert np.array_equal(a, ctx.localmap_read())
    ctx.close()


def test_float_accumulation_mode_equals_the_oracle_bit_for_bit(stream):
    """glio_localmap_set_accumulation(1): the centroids by pcl::VoxelGrid's own arithmetic (float sums in the order of the concatenated cloud) -- the map
    then equals the oracle's restatement BIT FOR BIT at every keyframe of a sliding ring (the default, exact fixed point, differs by <= 2e-5 m)"""
    from glio_amd import capi
    from oracle import pyoracle as po
    win, clouds = stream
    o = synth.default_opts(1, pts=8192, map_pts=1 << 17)
    ctx = capi.Context(o)
    width, leaf = 4, 0.4
    ctx.localmap_config(width, leaf, 8192)
    ctx.localmap_set_accumulation(1)
    poses = [(win.gt.quat[s], win.gt.trans[s]) for s in range(win.W)]
    for s in range(win.W):
        ctx.localmap_push(clouds[s], *poses[s])
        n = ctx.localmap_build()
        lo = max(0, s + 1 - width)
        ref, _ = _oracle_map(po, clouds[lo:s + 1], poses[lo:s + 1], leaf)
        got = ctx.localmap_read()
        assert n == len(ref) and np.array_equal(got, ref), f"keyframe {s}: max diff {np.abs(got - ref).max() if len(got) == len(ref) else None}"
    exact = ctx.localmap_read().copy()
    ctx.localmap_set_accumulation(0)
    ctx.localmap_build()
    d = np.abs(ctx.localmap_read() - exact).max()
    assert 0 < d <= 2e-5                                   # the two arithmetics differ, by float-accumulation noise
    ctx.close()


def test_association_on_device_built_map(stream):
    from glio_amd import capi
    from oracle import pyoracle as po
    win, clouds = stream
    o = synth.default_opts(1, pts=8192, map_pts=1 << 17)
    o.t_lb[:] = [0, 0, 0]
    ctx = capi.Context(o)
    ctx.localmap_config(5, 0.4, 8192)
    poses = [(win.gt.quat[s], win.gt.trans[s]) for s in range(5)]
    for s in range(5):
        ctx.localmap_push(clouds[s], *poses[s])
    ctx.localmap_build()
    dev_map = ctx.localmap_read().copy()
    q, t = win.init.quat[5], win.init.trans[5]
    n_dev = ctx.associate(0, clouds[5], q, t)
    pts_d, pl_d, sc_d = ctx.get_correspondences(0)
    # same association with the SAME map uploaded from the host: identical (K1/K2 do not care where the map came from)
    ctx2 = capi.Context(o)
    ctx2.set_map(dev_map)
    assert ctx2.associate(0, clouds[5], q, t) == n_dev
    p2, l2, s2 = ctx2.get_correspondences(0)
    assert np.array_equal(pts_d, p2) and np.array_equal(pl_d, l2) and np.array_equal(sc_d, s2)
    # and the oracle on that map
    po_pts, po_pl, po_sc, _ = po.associate(o, dev_map, clouds[5], q, t)
    assert len(po_sc) == n_dev and np.array_equal(po_pl, pl_d)
    assert n_dev > 1000
    ctx.close(); ctx2.close()


def test_push_of_the_resident_scan_gives_the_same_map(stream):
    """The newest keyframe's cloud is on the device already (glio_set_scan, window slot W - 1); glio_localmap_push_scan pushes that copy
    (body point = scan point - t_lb in float) instead of uploading it a second time: the ring and the map must be byte for byte what the
    host-cloud push gives."""
    from glio_amd import capi
    win, clouds = stream
    W = 3
    o = synth.default_opts(W, pts=8192, map_pts=1 << 17)
    tlb = np.array(win.opts.t_lb, np.float32)
    a, b = capi.Context(o), capi.Context(o)
    for c in (a, b):
        c.localmap_config(4, 0.4, 8192)
    for s in range(win.W):
        q, t = win.gt.quat[s], win.gt.trans[s]
        a.localmap_push(clouds[s], q, t)
        if s > 0:
            b.slide_window()
        b.set_scan(W - 1, win.scans[s])
        b.localmap_push_scan(W - 1, tlb, q, t)
        na, nb = a.localmap_build(), b.localmap_build()
        assert na == nb and np.array_equal(a.localmap_read(), b.localmap_read()), f"keyframe {s}"
    a.close(); b.close()


def test_ordered_output_by_bitmap_rank_and_by_radix_sort(stream):
    """The ordered voxel list comes from the rank of a voxel's bit in an occupancy bitmap of the bounding box (<= 2^27 cells) or, for larger boxes, from the
    radix sort.  One context goes through: a normal map (bitmap), a map with two far-away outliers that blow the box up to 8e9 cells (radix sort; the bits
    that fitted must be cleared again), the normal map once more (bitmap, must equal the first build bit for bit) -- every build against the oracle's voxel
    grid; and the whole sequence equals a context forced onto the radix sort (GLIO_LM_SORT=1)."""
    import os
    from glio_amd import capi
    from oracle import pyoracle as po
    win, clouds = stream
    o = synth.default_opts(1, pts=8192, map_pts=1 << 17)
    leaf, width = 0.4, 1
    ident = (np.array([1.0, 0, 0, 0]), np.zeros(3))
    far = clouds[1].copy()
    far[0, :3] = [700.0, -650.0, 90.0]; far[1, :3] = [-720.0, 610.0, -95.0]          # (1420 x 1260 x 185 m) / 0.4 m: 8.2e9 cells
    seq = [clouds[0], far, clouds[0], clouds[2]]
    results = []
    for force in ("0", "1"):
        os.environ["GLIO_LM_SORT"] = force
        try:
            ctx = capi.Context(o)
            ctx.localmap_config(width, leaf, 8192)
        finally:
            os.environ.pop("GLIO_LM_SORT", None)
        maps = []
        for cl in seq:
            ctx.localmap_push(cl, *ident)
            n = ctx.localmap_build()
            got = ctx.localmap_read().copy()
            ref, _ = po.voxel_grid(cl, leaf)
            assert n == len(ref) == len(got)
            assert np.abs(got - ref).max() <= 2e-5
            maps.append(got)
        assert np.array_equal(maps[0], maps[2])
        results.append(maps)
        ctx.close()
    for a, b in zip(*results):
        assert np.array_equal(a, b)


def test_map_built_ahead_on_the_upload_stream_is_the_same_map(stream):
    """glio_set_scan_ahead + glio_localmap_push_scan_ahead_and_build (the next keyframe's cloud and the next call's local map during this call's tail, on the upload
    stream; the next glio_slide_window takes both over) against the plain order (slide, glio_set_scan, glio_localmap_push_scan, glio_localmap_build): the same map,
    byte for byte, keyframe after keyframe, and the same association against it."""
    from glio_amd import capi
    win, clouds = stream
    W = 3
    o = synth.default_opts(W, pts=8192, map_pts=1 << 17)
    tlb = np.array(win.opts.t_lb, np.float32)
    a, b = capi.Context(o), capi.Context(o)
    for c in (a, b):
        c.localmap_config(4, 0.4, 8192)
    for s in range(win.W):
        q, t = win.gt.quat[s], win.gt.trans[s]
        if s > 0:
            a.slide_window(); b.slide_window()            # (b: takes the scan and the map sent ahead)
        a.set_scan(W - 1, win.scans[s])
        a.localmap_push_scan(W - 1, tlb, q, t)
        na = a.localmap_build()
        if s == 0:
            b.set_scan(W - 1, win.scans[s]); b.localmap_push_scan(W - 1, tlb, q, t); nb = b.localmap_build()
        assert na == nb and np.array_equal(a.localmap_read(), b.localmap_read()), f"keyframe {s}"
        pose = capi.lidar_pose(win.opts, q, t)
        ca, cb = a.associate_resident(W - 1, *pose), b.associate_resident(W - 1, *pose)
        assert ca == cb and ca > 100
        assert all(np.array_equal(x, y) for x, y in zip(a.get_correspondences(W - 1), b.get_correspondences(W - 1)))
        if s + 1 < win.W:                                  # during "the tail": the next keyframe's scan, then its map
            b.set_scan_ahead(win.scans[s + 1])
            nb = b.localmap_push_scan_ahead_and_build(tlb, win.gt.quat[s + 1], win.gt.trans[s + 1])
    a.close(); b.close()
