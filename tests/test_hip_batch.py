"""GPU parity of the batch stage (K8 pair kernel + banded assembly + banded solve) against the oracle's
BinaryLidarPlaneNormFactor restatement, and the sharding property the multi-GPU path relies on."""
import numpy as np
import pytest

from glio_amd import batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def po():
    from oracle import pyoracle
    return pyoracle


def _problem(K, band, per_kf, seed=3):
    gt, init = batch.make_poses(K, seed=seed)
    ci, cj, cp, nc, score = batch.make_constraints(gt, 0, K, per_kf, band, seed=seed, device="cuda:0")
    return gt, init, ci, cj, cp, nc, score


@pytest.mark.parametrize("K,band,per_kf", [(14, 3, 301), (40, 6, 500)])
def test_batch_linearize_matches_oracle(po, K, band, per_kf):
    gt, init, ci, cj, cp, nc, score = _problem(K, band, per_kf)
    st = batch.BatchStage(K, band, len(ci))
    st.set_constraints(ci, cj, cp, nc, score)
    Hg = st.new_hg()
    st.linearize(init, Hg)
    Hb, g, cost = batch.unpack_hg(Hg.cpu().numpy(), K, band)
    Ho, go, co = po.batch_linearize(K, band, np.ascontiguousarray(init), ci, cj, cp.cpu().numpy(), nc.cpu().numpy(), score.cpu().numpy())
    assert abs(cost - co) <= 1e-11 * co
    assert np.linalg.norm(g - go) <= 1e-11 * np.linalg.norm(go)
    assert np.linalg.norm(Hb - Ho.reshape(Hb.shape)) <= 1e-11 * np.linalg.norm(Ho)
    st.close()


@pytest.mark.parametrize("K,band", [(30, 4), (60, 9), (64, 12), (50, 16)], ids=["band4", "band9", "band12_two_row_slots", "band16"])
def test_batch_step_matches_dense_solve(K, band):
    gt, init, ci, cj, cp, nc, score = _problem(K, band, 400, seed=5)
    st = batch.BatchStage(K, band, len(ci))
    st.set_constraints(ci, cj, cp, nc, score)
    Hg = st.new_hg()
    st.linearize(init, Hg)
    lam = 1e-3
    new, mdec = st.step(Hg, lam, init)
    Hb, g, cost = batch.unpack_hg(Hg.cpu().numpy(), K, band)
    H = batch.dense_from_band(Hb, K, band)
    Hd = H + np.diag(lam * np.diag(H) + 1e-12)
    d = np.linalg.solve(Hd, -g.ravel())
    assert np.allclose(new[:, :3] - init[:, :3], d.reshape(K, 6)[:, :3], rtol=1e-8, atol=1e-10)
    assert np.isclose(mdec, -(g.ravel() @ d + 0.5 * d @ H @ d), rtol=1e-8)
    assert np.allclose(np.linalg.norm(new[:, 3:], axis=1), 1.0, atol=1e-12)
    st.close()


def test_shard_sum_equals_full_linearisation():
    """What the RCCL all-reduce computes: the ranks' partial [H|g|cost] buffers add up to the full one."""
    K, band, per_kf = 36, 6, 240
    gt, init = batch.make_poses(K, seed=9)
    full = batch.make_constraints(gt, 0, K, per_kf, band, seed=9, device="cuda:0")
    st = batch.BatchStage(K, band, len(full[0]))
    st.set_constraints(*full)
    Hg_full = st.new_hg()
    st.linearize(init, Hg_full)
    # the constraint set split by SOURCE keyframe range (batch.shard_range: the ownership rule of the sharded stage) into
    # 2, 3 and 8 ranks: the ranks' buffers must add up to the full linearisation
    ci, cj, cp, nc, score = full
    f = Hg_full.cpu().numpy()
    for world in (2, 3, 8):
        acc = st.new_hg()
        total = 0
        for r in range(world):
            lo, hi = batch.shard_range(K, r, world)
            a0, a1 = int(np.searchsorted(ci, lo, side="left")), int(np.searchsorted(ci, hi, side="left"))
            total += a1 - a0
            sr = batch.BatchStage(K, band, max(1, a1 - a0))
            sr.set_constraints(ci[a0:a1], cj[a0:a1], cp[a0:a1].contiguous(), nc[a0:a1].contiguous(), score[a0:a1].contiguous())
            Hg = sr.new_hg()
            sr.linearize(init, Hg)
            acc += Hg
            sr.close()
        assert total == len(ci)
        a = acc.cpu().numpy()
        assert np.linalg.norm(a - f) <= 1e-13 * np.linalg.norm(f), f"{world} shards"
        assert abs(a[-1] - f[-1]) <= 1e-13 * abs(f[-1])
    # and the generator's own per-rank sets (what bench.py builds on every rank) cover every source keyframe exactly once
    for world in (2, 3, 8):
        n_tot = 0
        for r in range(world):
            lo, hi = batch.shard_range(K, r, world)
            part = batch.make_constraints(gt, lo, hi, per_kf, band, seed=9, device="cuda:0")
            assert len(part[0]) == (hi - lo) * per_kf and (part[0] >= lo).all() and (part[0] < hi).all()
            n_tot += len(part[0])
        assert n_tot == K * per_kf
    a, f = acc.cpu().numpy(), Hg_full.cpu().numpy()
    assert np.linalg.norm(a - f) <= 1e-13 * np.linalg.norm(f)
    st.close()


def test_batch_lm_reduces_cost_and_recovers_relative_poses():
    K, band = 60, 6
    gt, init, ci, cj, cp, nc, score = _problem(K, band, 600, seed=11)
    st = batch.BatchStage(K, band, len(ci))
    st.set_constraints(ci, cj, cp, nc, score)
    bufs = [st.new_hg(), st.new_hg()]
    flip = [0]

    def lin(p):
        flip[0] ^= 1
        st.linearize(p, bufs[flip[0]])
        return bufs[flip[0]], float(bufs[flip[0]][-1].item())
    poses, hist = batch.lm_solve(lin, st.step, init, iterations=8)
    assert hist[-1] < 5e-2 * hist[0] and all(b <= a for a, b in zip(hist, hist[1:]))      # down to the 2 cm noise floor
    rel = lambda P: np.linalg.norm(np.diff(P[:, :3], axis=0) - np.diff(gt[:, :3], axis=0), axis=1).max()
    assert rel(poses) < 0.2 * rel(init)
    st.close()


@pytest.mark.parametrize("K,band", [(60, 6), (203, 6), (2000, 6), (50, 12), (301, 12), (37, 3)])
def test_block_cyclic_reduction_equals_sequential_banded_solve(K, band):
    """The damped banded solve by block cyclic reduction (batch_solve_kernels.hip: super-blocks of 6 or 12 keyframes, all
    eliminations of a level in parallel) against the one-workgroup sequential banded Cholesky and, at small K, numpy."""
    gt, init, ci, cj, cp, nc, score = _problem(K, band, 40 if K > 500 else 200, seed=21 + K)
    st = batch.BatchStage(K, band, len(ci))
    st.set_constraints(ci, cj, cp, nc, score)
    Hg = st.new_hg()
    st.linearize(init, Hg)
    lam = 1e-4
    st.set_solver(1)
    new1, m1 = st.step(Hg, lam, init)
    new1b, m1b = st.step(Hg, lam, init)
    assert np.array_equal(new1, new1b) and m1 == m1b                       # fixed-order sums: bit-identical runs
    st.set_solver(0)
    new0, m0 = st.step(Hg, lam, init)
    d1, d0 = new1[:, :3] - init[:, :3], new0[:, :3] - init[:, :3]
    assert np.abs(d1 - d0).max() <= 1e-9 * max(1.0, np.abs(d0).max()), np.abs(d1 - d0).max()
    assert np.abs(new1[:, 3:] - new0[:, 3:]).max() <= 1e-10
    assert abs(m1 - m0) <= 1e-9 * abs(m0)
    if K <= 300:
        Hb, g, cost = batch.unpack_hg(Hg.cpu().numpy(), K, band)
        H = batch.dense_from_band(Hb, K, band)
        d = np.linalg.solve(H + np.diag(lam * np.diag(H) + 1e-12), -g.ravel())
        assert np.allclose(d1, d.reshape(K, 6)[:, :3], rtol=1e-8, atol=1e-10)
    st.close()


def test_k8_by_moments_equals_the_streamed_linearisation_and_the_oracle():
    """The plane constraints' residual is linear in (R_b^T R_a, R_b^T (t_a - t_b)): the pairs' moments, taken once at poses P0, give H, g
    and the cost at ANY poses.  At P0 itself, and at poses 0.3 m / 2 degrees away from P0, the moment form must equal the streamed kernel
    (k_batch_pairs) to rounding and the oracle to the tolerance the streamed kernel is held to."""
    from oracle import pyoracle as po
    K, band = 40, 6
    gt, init = batch.make_poses(K, seed=91, perturb=(0.05, 0.003))
    ci, cj, cp, nc, score = batch.make_constraints(gt, 0, K, 300, band, seed=91)
    st = batch.BatchStage(K, band, len(ci))
    cp, nc, score = cp.numpy(), nc.numpy(), score.numpy()
    st.set_constraints(ci, cj, cp, nc, score)
    rng = np.random.default_rng(91)
    moved = init.copy()
    moved[:, :3] += rng.normal(0, 0.3, (K, 3))
    dq = rng.normal(0, 0.02, (K, 3))
    for k in range(K):
        w, x, y, z = moved[k, 3:]
        d = np.concatenate([[1.0], 0.5 * dq[k]]); d /= np.linalg.norm(d)
        moved[k, 3:] = [w * d[0] - x * d[1] - y * d[2] - z * d[3], w * d[1] + x * d[0] + y * d[3] - z * d[2],
                        w * d[2] - x * d[3] + y * d[0] + z * d[1], w * d[3] + x * d[2] - y * d[1] + z * d[0]]
    nH = K * (band + 1) * 36
    for at, centre in ((init, init), (moved, init), (init, moved)):
        streamed, mom = st.new_hg(), st.new_hg()
        st.set_constraints(ci, cj, cp, nc, score)           # (drops the moment records of the previous case: they are a cache keyed by the constraint set)
        st.linearize_mode(at, streamed, 0)
        st.linearize_mode(centre, mom, 1)                   # moments taken at `centre` ...
        st.linearize_mode(at, mom, 2)                       # ... evaluated at `at`
        a, b = streamed.cpu().numpy(), mom.cpu().numpy()
        assert np.abs(a[:nH] - b[:nH]).max() <= 1e-12 * np.abs(a[:nH]).max()
        assert np.abs(a[nH:-1] - b[nH:-1]).max() <= 1e-11 * np.abs(a[nH:-1]).max()
        assert abs(a[-1] - b[-1]) <= 1e-11 * a[-1]
        H, g, cost = po.BatchProblem(K, band, ci, cj, cp, nc, score).linearize(at)
        want = np.concatenate([H.ravel(), g.ravel(), [cost]])
        assert np.abs(b[:nH] - want[:nH]).max() <= 1e-11 * np.abs(want[:nH]).max()
        assert np.abs(b[nH:-1] - want[nH:-1]).max() <= 1e-10 * np.abs(want[nH:-1]).max()
        assert abs(b[-1] - want[-1]) <= 1e-10 * want[-1]
    st.close()


def test_moment_records_of_unchanged_pairs_are_kept_and_replaced_pairs_are_taken_again():
    """A round of optimizeBatch replaces the constraints of the end keyframes only (Estimator.cpp:3018-3030).  The stage is told which pairs
    changed; the next linearisation takes the moments of those pairs again and keeps the others.  Against the streamed kernel on the new set;
    and the control: with the replaced pairs NOT marked the stale records must show."""
    import torch
    K, band, per_kf = 30, 6, 240
    gt, init = batch.make_poses(K, seed=93, perturb=(0.05, 0.003))
    ci, cj, cp, nc, score = batch.make_constraints(gt, 0, K, per_kf, band, seed=93, device="cuda:0")
    ci2, cj2, cp2, nc2, score2 = batch.make_constraints(gt, 0, K, per_kf, band, seed=94, device="cuda:0")       # another draw, same pair structure
    assert np.array_equal(ci, ci2) and np.array_equal(cj, cj2)
    # the pair list: runs of equal (ci, cj)
    key = ci.astype(np.int64) * K + cj
    start = np.r_[0, np.flatnonzero(np.diff(key)) + 1]
    pci, pcj, cnt = ci[start], cj[start], np.diff(np.r_[start, len(ci)])
    ends = (pci < 3) | (pci >= K - 3)                                 # the "end keyframes" of this test
    per_con = np.repeat(ends, cnt)
    mask = torch.as_tensor(per_con, device="cuda:0")
    cpn, ncn, scn = torch.where(mask[:, None], cp2, cp), torch.where(mask[:, None], nc2, nc), torch.where(mask, score2, score)
    st = batch.BatchStage(K, band, len(ci))
    st.set_constraints_pairs(pci, pcj, cnt, cp, nc, score)
    first = st.new_hg(); st.linearize_mode(init, first, 1)             # moments of ALL pairs (first set)
    nH = K * (band + 1) * 36
    moved = init.copy(); moved[:, :3] += np.random.default_rng(93).normal(0, 0.05, (K, 3))

    def close(a, b):
        return (np.abs(a[:nH] - b[:nH]).max() <= 1e-12 * np.abs(a[:nH]).max() and np.abs(a[nH:-1] - b[nH:-1]).max() <= 1e-11 * np.abs(a[nH:-1]).max()
                and abs(a[-1] - b[-1]) <= 1e-11 * a[-1])

    st.set_constraints_pairs(pci, pcj, cnt, cpn, ncn, scn, changed=ends)
    assert capi_counts(st) == (1, int(ends.sum()))
    streamed, mom = st.new_hg(), st.new_hg()
    st.linearize_mode(moved, streamed, 0)
    st.linearize_mode(moved, mom, 1)                                   # takes the marked pairs only
    assert capi_counts(st) == (1, 0)
    assert close(streamed.cpu().numpy(), mom.cpu().numpy())
    # control: back to the first set WITHOUT marking the ends -> their records are the second set's: visibly wrong
    st.set_constraints_pairs(pci, pcj, cnt, cp, nc, score, changed=np.zeros(len(pci), np.uint8))
    st.linearize_mode(moved, streamed, 0)
    st.linearize_mode(moved, mom, 1)
    assert not close(streamed.cpu().numpy(), mom.cpu().numpy())
    # an unmarked call (changed = None) invalidates everything: right again
    st.set_constraints_pairs(pci, pcj, cnt, cp, nc, score)
    assert capi_counts(st)[0] == 0
    st.linearize_mode(moved, mom, 1)
    assert close(streamed.cpu().numpy(), mom.cpu().numpy())
    st.close()


def capi_counts(st):
    import ctypes as C
    from glio_amd import capi
    v = (C.c_int * 2)()
    capi._check(capi.load().glio_debug_batch_moment_state(st._h, v))
    return int(v[0]), int(v[1])
