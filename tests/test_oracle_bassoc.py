"""CPU: the search-window rule and the oracle's restatement of findGlobalCorrespondingSurfFeaturesAdd_Batch
(reference GLIO/src/Estimator.cpp:3009-3017, 3808-3892) against an independent numpy transcription."""
import numpy as np

from glio_amd import batch, synth
from oracle import pyoracle as po


def test_search_window_rule():
    # interior: centred; ends: clamped to a full 2r+1 window (Estimator.cpp:3009-3017)
    assert batch.search_window(10, 40, 6) == 4
    assert batch.search_window(2, 40, 6) == 0
    assert batch.search_window(38, 40, 6) == 40 - 13
    ci, cj = batch.pair_list(40, 6)
    assert len(ci) == 40 * 12 and (ci != cj).all()
    assert np.all(np.diff(ci.astype(np.int64) * 100 + cj) > 0)                 # (ci, cj) sorted
    assert np.abs(ci - cj).max() == 12


def _numpy_pair(scan_a, pose_a, scan_b, pose_b):
    Ra, Rb = synth.q2R(pose_a[3:] / np.linalg.norm(pose_a[3:])), synth.q2R(pose_b[3:] / np.linalg.norm(pose_b[3:]))
    ga = (scan_a[:, :3].astype(np.float64) @ Ra.T + pose_a[:3]).astype(np.float32)
    gb = (scan_b[:, :3].astype(np.float64) @ Rb.T + pose_b[:3]).astype(np.float32)
    out = []
    for i in range(len(ga)):
        d = ((ga[i] - gb) ** 2).sum(1, dtype=np.float32)
        order = np.lexsort((np.arange(len(d)), d))[:5]
        if d[order[4]] >= 1.5:
            continue
        A = gb[order].astype(np.float64)
        Al = scan_b[order, :3].astype(np.float64)
        n = np.linalg.lstsq(A, -np.ones(5), rcond=None)[0]
        nl = np.linalg.lstsq(Al, -np.ones(5), rcond=None)[0]
        ninv = 1.0 / np.linalg.norm(n)
        n = n * ninv
        if (np.abs(A @ n + ninv) > 0.18).any():
            continue
        pd = np.float32(n @ ga[i].astype(np.float64) + ninv)
        w = np.float32(1.0 - 0.9 * float(abs(pd)) / float(np.sqrt(np.sqrt(np.float32((ga[i] ** 2).sum(dtype=np.float32))))))
        if float(w) > 0.3:
            out.append((i, nl / np.linalg.norm(nl), Al.mean(0), 2.5 * float(w)))
    return out


def test_pair_association_matches_numpy():
    win = synth.make_window(W=3, pts_per_scan=1500, seed=synth.SEED_BASE + 52, perturb=(0.03, 0.2, 0.0), scan_radius=10.0, map_density=1.0)
    tlb = np.array(win.opts.t_lb, np.float32)
    scans = []
    for s in range(3):
        sc = win.scans[s].copy(); sc[:, :3] -= tlb
        scans.append(sc)
    poses = np.c_[win.init.trans, win.init.quat]
    total = 0
    for a, b in ((0, 1), (1, 0), (2, 1)):
        cp, nc, sc, src = po.associate_pair(scans[a], poses[a], scans[b], poses[b])
        ref = _numpy_pair(scans[a], poses[a], scans[b], poses[b])
        assert [r[0] for r in ref] == list(src), "kept set / order"
        assert np.array_equal(cp, scans[a][src])
        for k, (_, nl, cen, score) in enumerate(ref):
            assert min(np.linalg.norm(nc[k, :3] - nl), np.linalg.norm(nc[k, :3] + nl)) < 1e-9      # lstsq vs QR: same plane
            assert np.allclose(nc[k, 3:], cen, atol=1e-12) and abs(sc[k] - score) < 1e-6
        total += len(src)
    assert total > 300
