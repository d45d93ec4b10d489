"""The ONE-CALL window association (merged grouping + near-block search with its exactness certificate and hand-on lists) on inputs built to
break it, slot against slot with the one-group-per-query search (mode 1, no certificate, no grouping) and, on a sample, with the oracle's brute
force: exact ties by the dozen (lattices), queries and map points ON cell and half-cell faces of the voxel hash (edge 1.25 m: the octant of a
point and the certificate's distance to the cell face are decided there), negative and large coordinates, one cell holding thousands of
queries of several slots, slots of very different density, a map of coincident points.  Bit-exact records, findCorrespondingSurfFeatures
(Estimator.cpp:3633-3708)."""
import numpy as np
import pytest

from glio_amd import ctypes_types as T
from glio_amd import synth

pytestmark = pytest.mark.gpu

CELL = 1.25


def _f4(xyz):
    a = np.zeros((len(xyz), 4), np.float32)
    a[:, :3] = xyz
    return a


def _cases():
    rng = np.random.default_rng(20260926)
    out = {}
    # (a) 0.25 m lattice around the origin (negative and positive cells), three layers; queries on points, centres, faces of cells
    g = np.arange(-20, 21) * 0.25
    lat = np.stack(np.meshgrid(g, g, [-0.25, 0.0, 0.25], indexing="ij"), -1).reshape(-1, 3)
    lat = lat[rng.permutation(len(lat))]
    q_on = lat[rng.integers(0, len(lat), 700)]
    q_ctr = lat[rng.integers(0, len(lat), 700)] + 0.125
    k = rng.integers(-3, 4, (700, 3)).astype(float)
    q_face = k * CELL                                                         # corners of hash cells
    q_half = k * CELL + rng.choice([0.0, 0.625], (700, 3))                    # on half-cell faces
    q_rand = rng.uniform(-5.5, 5.5, (900, 3)) * [1, 1, 0.08]
    out["lattice"] = (_f4(lat), [_f4(q_on), _f4(q_ctr), _f4(q_face), _f4(np.concatenate([q_half, q_rand]))])
    # (b) map points ON cell faces: x, y multiples of 0.625, z jittered; far from the origin on the negative side
    base = np.array([-1875.0, -940.0, 12.5])
    m = np.stack(np.meshgrid(np.arange(0, 12) * 0.625, np.arange(0, 12) * 0.625, [0.0], indexing="ij"), -1).reshape(-1, 3)
    m = np.repeat(m, 6, 0) + np.concatenate([np.zeros((len(m) * 6, 2)), rng.normal(0, 0.05, (len(m) * 6, 1))], 1)
    m = m[rng.permutation(len(m))] + base
    q1 = rng.uniform([0, 0, -0.2], [7, 7, 0.2], (1500, 3)) + base
    q2 = (np.stack([rng.integers(0, 12, 800), rng.integers(0, 12, 800), np.zeros(800)], 1) * 0.625) + base
    out["faces_far_negative"] = (_f4(m), [_f4(q1), _f4(q2), _f4(q1[:3]), _f4(q2[:1])])
    # (c) density contrast: a dense slab (600 per m^2) beside a sparse one (3 per m^2); one slot has 6000 queries in ONE cell, others few
    dense = rng.uniform([5, -4, -0.04], [11, 4, 0.04], (28000, 3))
    sparse = rng.uniform([11, -12, -0.04], [40, 12, 0.04], (2200, 3))
    mp = np.concatenate([dense, sparse]); mp = mp[rng.permutation(len(mp))]
    one_cell = rng.uniform([7.55, 0.05, -0.2], [8.7, 1.2, 0.2], (6000, 3))
    out["density_contrast"] = (_f4(mp), [_f4(one_cell), _f4(rng.uniform([5, -4, -0.3], [40, 12, 0.3], (3000, 3))), _f4(one_cell[:130]),
                                         _f4(rng.uniform([10, -5, -0.1], [13, 5, 0.1], (2500, 3)))])
    # (d) coincident map points (every point eight times) and fewer than five distinct neighbours within reach
    pts = rng.uniform([0, 0, 0], [6, 6, 0.02], (300, 3))
    co = np.repeat(pts, 8, 0); co = co[rng.permutation(len(co))]
    lonely = np.concatenate([pts[:4] + [30, 0, 0], pts[:40] + [0, 0, 1.2]])   # four points far away; a layer 1.2 m above (the gate's edge)
    out["coincident"] = (_f4(np.concatenate([co, lonely])), [_f4(rng.uniform([-1, -1, -0.5], [7, 7, 1.5], (2000, 3))), _f4(pts[:500] + 1e-3),
                                                            _f4(pts[:4] + [30, 0, 0.1]), _f4(np.zeros((0, 3)))])
    return out


@pytest.mark.parametrize("name", ["lattice", "faces_far_negative", "density_contrast", "coincident"])
def test_window_call_on_adversarial_inputs(name):
    from glio_amd import capi
    from oracle import pyoracle as po
    mp, slots = _cases()[name]
    W = len(slots)
    cap = max(16, max(len(s) for s in slots))
    o = synth.default_opts(W, pts=cap, map_pts=len(mp))
    rng = np.random.default_rng(7)
    q2s = np.tile([1.0, 0, 0, 0], (W, 1)); t2s = np.zeros((W, 3))
    if name != "faces_far_negative":                                       # (the face cases keep the identity: the points must stay ON the faces)
        for s in range(1, W):
            ang = rng.normal(0, 0.01, 3)
            q = np.array([1.0, *(0.5 * ang)]); q2s[s] = q / np.linalg.norm(q); t2s[s] = rng.normal(0, 0.05, 3)
    lib = capi.load()
    want = []
    lib.glio_debug_set_knn_mode(1)
    try:
        ref = capi.Context(o); ref.set_map(mp)
        for s in range(W):
            n = ref.associate(0, slots[s], q2s[s], t2s[s]) if len(slots[s]) else 0
            want.append((n,) + (tuple(a.copy() for a in ref.get_correspondences(0)) if len(slots[s]) else ()))
        ref.close()
    finally:
        lib.glio_debug_set_knn_mode(0)
    assert sum(w[0] for w in want) > 0 or name == "coincident"
    for mode in (0, 3):
        lib.glio_debug_set_knn_mode(mode)
        try:
            ctx = capi.Context(o); ctx.set_map(mp)
            for s in range(W):
                ctx.set_scan(s, slots[s] if len(slots[s]) else np.zeros((0, 4), np.float32))
            for rep in range(2):                                           # twice: the grouping tables are left clean
                cnt = ctx.associate_window(q2s, t2s)
                for s in range(W):
                    assert cnt[s] == want[s][0], (name, mode, rep, s, cnt[s], want[s][0])
                    if want[s][0]:
                        got = ctx.get_correspondences(s)
                        assert all(np.array_equal(a.view(np.uint8), b.view(np.uint8)) for a, b in zip(got, want[s][1:])), (name, mode, rep, s)
            ctx.close()
        finally:
            lib.glio_debug_set_knn_mode(0)
    # the oracle's brute force on the first 400 queries of every slot
    w2 = type("Wn", (), {"opts": o, "map_pts": mp})
    for s in range(W):
        if not len(slots[s]):
            continue
        sub = np.ascontiguousarray(slots[s][:400])
        pts, pl, sc, src = po.associate(o, mp, sub, q2s[s], t2s[s])
        ctx = capi.Context(o); ctx.set_map(mp)
        n = ctx.associate(0, sub, q2s[s], t2s[s])
        hp, hpl, hsc = ctx.get_correspondences(0)
        assert n == len(sc) and np.array_equal(hp, pts) and np.array_equal(hpl.view(np.uint32), pl.view(np.uint32)) and np.array_equal(hsc, sc), (name, s)
        ctx.close()
