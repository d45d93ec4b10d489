"""The rest of the batch problem on the device -- delta_q attitude constraints, DD pseudoranges, the trust-region solve behind
glio_batch_solve_tr -- against oracle/orc_batch.c (orc_batch_linearize_full / orc_batch_solve) on the same problem."""
import numpy as np
import pytest

from glio_amd import batch
from glio_amd import ctypes_types as T

pytestmark = pytest.mark.gpu


def _problem(K=60, band=6, per_kf=200, seed=33, search_range=3):
    gt, init = batch.make_poses(K, seed=seed, perturb=(0.08, 0.004))
    ci, cj, cp, nc, score = batch.make_constraints(gt, 0, K, per_kf, band, seed=seed)
    rng = np.random.default_rng(seed)
    odo = gt.copy()
    odo[:, :3] += rng.normal(0, 0.02, (K, 3))
    dq = batch.delta_q_pairs(odo, search_range)
    dd, frame = batch.make_batch_gnss(gt, seed=seed)
    return gt, init, (ci, cj, cp.numpy(), nc.numpy(), score.numpy()), dq, dd, frame


def _oracle(K, band, con, dq, dd, frame):
    from oracle import pyoracle as po
    return po.BatchProblem(K, band, *con, dq=dq, dd=dd, frame=frame)


def test_small_factors_added_after_the_reduce_match_the_oracle():
    K, band = 60, 6
    gt, init, con, dq, dd, frame = _problem(K, band)
    st = batch.BatchStage(K, band, len(con[0]))
    st.set_constraints(*con)
    st.set_small_factors(dq, dd, frame, threshold=10.0)
    Hg = st.new_hg()
    st.linearize(init, Hg)
    lidar_only = Hg.cpu().numpy().copy()
    st.add_small(init, Hg)
    got = Hg.cpu().numpy()
    H, g, cost = _oracle(K, band, con, dq, dd, frame).linearize(init)
    want = np.concatenate([H.ravel(), g.ravel(), [cost]])
    assert np.abs(got - lidar_only).max() > 1.0, "the small factors must contribute"
    nH = K * (band + 1) * 36
    assert np.abs(got[:nH] - want[:nH]).max() <= 1e-11 * np.abs(want[:nH]).max()
    assert np.abs(got[nH:-1] - want[nH:-1]).max() <= 1e-11 * np.abs(want[nH:-1]).max()
    assert abs(got[-1] - want[-1]) <= 1e-12 * want[-1]
    # bit-stable: a second evaluation gives the same bits (no atomics in the add)
    Hg2 = st.new_hg(); st.linearize(init, Hg2); st.add_small(init, Hg2)
    assert np.array_equal(Hg2.cpu().numpy(), got)
    st.close()


def _stage(K, band, con, dq, dd, frame, imu=None, rank=0, world=1, threshold=None):
    lo, hi = batch.shard_range(K, rank, world, band)
    own = (con[0] >= lo) & (con[0] < hi)
    mine = [c[own] for c in con]
    st = batch.BatchStage(K, band, max(1, len(mine[0])))
    if world > 1:
        assert st.set_shard(rank, world) == (lo, hi)
    st.set_constraints(*mine)
    st.set_small_factors(dq, dd, frame, threshold=threshold)
    if imu is not None:
        st.set_imu(imu)
    return st


@pytest.mark.parametrize("with_small", [False, True])
@pytest.mark.parametrize("dogleg", [T.DOGLEG_TRADITIONAL, T.DOGLEG_SUBSPACE])
def test_trust_region_solve_follows_the_oracle(with_small, dogleg):
    K, band = 60, 6
    gt, init, con, dq, dd, frame = _problem(K, band, seed=35)
    if not with_small:
        dq, dd = None, []
    for f in dd:
        f.threshold = 10.0
    st = _stage(K, band, con, dq, dd, frame)
    opts = T.batch_tr_opts(max_iterations=30, dogleg=dogleg)
    poses, summ = st.solve_tr(init, opts)
    want, wsum = _oracle(K, band, con, dq, dd, frame).solve(init, opts)
    assert summ.iterations == wsum.iterations and summ.successful_steps == wsum.successful_steps and summ.termination == wsum.termination, (summ.as_dict(), wsum.as_dict())
    assert np.isclose(summ.initial_cost, wsum.initial_cost, rtol=1e-12)
    assert np.isclose(summ.final_cost, wsum.final_cost, rtol=1e-9)
    assert np.abs(poses - want).max() < 1e-8
    assert summ.final_cost < 0.05 * summ.initial_cost
    if not with_small:
        assert np.abs(poses[:, :3] - gt[:, :3]).max() < 0.05
    assert st.counters()["hook_calls"] == 0          # one rank: no collective at all
    st.close()


def _imu_problem(K=48, band=6, per_kf=150, seed=51, perturb=(0.08, 0.004)):
    gt, init = batch.make_poses(K, seed=seed, perturb=perturb)
    ci, cj, cp, nc, score = batch.make_constraints(gt, 0, K, per_kf, band, seed=seed)
    rng = np.random.default_rng(seed)
    odo = gt.copy(); odo[:, :3] += rng.normal(0, 0.02, (K, 3))
    dq = batch.delta_q_pairs(odo, 3)
    dd, frame = batch.make_batch_gnss(gt, seed=seed)
    for f in dd:
        f.threshold = 10.0
    imu, sb_gt, sb0 = batch.make_batch_imu(K, seed=seed)
    return gt, init, (ci, cj, cp.numpy(), nc.numpy(), score.numpy()), dq, dd, frame, imu, sb0


def test_linearisation_with_the_imu_chain_matches_the_oracle():
    """diag(H), g and the cost of the 15-state problem (plane + delta_q + DD + ImuFactor chain) through the solver's own path"""
    from oracle import pyoracle as po
    K, band = 48, 6
    gt, init, con, dq, dd, frame, imu, sb0 = _imu_problem(K, band)
    st = _stage(K, band, con, dq, dd, frame, imu=imu)
    diag, g, cost = st.linearize_full(init, sb0)
    P = po.BatchProblem(K, band, *con, dq=dq, dd=dd, frame=frame, imu=imu)
    H, gw, cw = P.linearize_dense(init, sb0)
    assert abs(cost - cw) <= 1e-11 * cw
    assert np.abs(diag - np.diag(H)).max() <= 1e-10 * np.diag(H).max()
    assert np.abs(g - gw).max() <= 1e-10 * np.abs(gw).max()
    st.close()


@pytest.mark.parametrize("radius0", [1e4, 0.5])
def test_trust_region_solve_with_the_imu_chain_follows_the_oracle(radius0):
    """the complete batch problem: 15 unknowns per keyframe, SUBSPACE_DOGLEG, non-monotonic steps; the small initial radius makes
    the first iterations boundary-constrained subspace steps (the quartic) and the run accepts cost-raising steps at the end"""
    from oracle import pyoracle as po
    K, band = 48, 6
    gt, init, con, dq, dd, frame, imu, sb0 = _imu_problem(K, band, perturb=(0.3, 0.02) if radius0 < 1 else (0.08, 0.004))
    st = _stage(K, band, con, dq, dd, frame, imu=imu)
    opts = T.batch_tr_opts(max_iterations=16)
    opts.initial_trust_region_radius = radius0
    poses, sb, summ = st.solve_tr(init, opts, speed_bias=sb0)
    P = po.BatchProblem(K, band, *con, dq=dq, dd=dd, frame=frame, imu=imu)
    want, wsb, wsum = P.solve2(init, opts, sb0)
    assert summ.iterations == wsum.iterations and summ.successful_steps == wsum.successful_steps and summ.termination == wsum.termination, (summ.as_dict(), wsum.as_dict())
    assert np.isclose(summ.initial_cost, wsum.initial_cost, rtol=1e-12)
    assert np.isclose(summ.final_cost, wsum.final_cost, rtol=2e-8 if radius0 < 1 else 1e-9)
    assert np.abs(poses - want).max() < 1e-8 and np.abs(sb - wsb).max() < 1e-7
    assert summ.final_cost < 0.01 * summ.initial_cost
    st.close()


@pytest.mark.parametrize("K,world", [(50, 1), (50, 2), (43, 3), (13, 1)])
def test_ragged_last_super_block_with_the_imu_chain(K, world):
    """K not a multiple of 6: the last super-block is padded with identity keyframes, whose inner speed-bias blocks go through the
    pre-elimination like any other (k_bcr_pre / k_bcr_post); also on virtual ranks, where the ragged block is the last rank's."""
    from oracle import pyoracle as po
    import torch
    band = 6
    gt, init, con, dq, dd, frame, imu, sb0 = _imu_problem(K, band, per_kf=100, seed=59)
    opts = T.batch_tr_opts(max_iterations=12)
    P = po.BatchProblem(K, band, *con, dq=dq, dd=dd, frame=frame, imu=imu)
    want, wsb, wsum = P.solve2(init, opts, sb0)
    if world == 1:
        st = _stage(K, band, con, dq, dd, frame, imu=imu)
        poses, sb, summ = st.solve_tr(init, opts, speed_bias=sb0)
        st.close()
    else:
        stages = [_stage(K, band, con, dq, dd, frame, imu=imu, rank=r, world=world) for r in range(world)]
        ranks = batch.ThreadRanks(world, sync=torch.cuda.synchronize)
        res = ranks.run(lambda r, d: stages[r].solve_tr(init, opts, d, speed_bias=sb0))
        poses, sb, summ = res[0]
        for other in res[1:]:
            assert np.array_equal(other[0], poses) and np.array_equal(other[1], sb)
        for s_ in stages:
            s_.close()
    assert summ.iterations == wsum.iterations and summ.termination == wsum.termination, (summ.as_dict(), wsum.as_dict())
    assert np.isclose(summ.final_cost, wsum.final_cost, rtol=1e-8)
    assert np.abs(poses - want).max() < 1e-7 and np.abs(sb - wsb).max() < 1e-6


@pytest.mark.parametrize("world", [2, 3, 4])
@pytest.mark.parametrize("with_imu", [False, True])
def test_sharded_solve_on_virtual_ranks_equals_one_rank(world, with_imu):
    """`world` BatchStage objects in one process (one thread each, all on this GPU), each owning its keyframe range, its
    constraints, its small factors and its IMU edges; the all-reduce hook sums the five small buffers per iteration across the
    threads.  The sharded solve must take the same decisions as the one-rank solve and as the oracle."""
    import torch
    from oracle import pyoracle as po
    K, band = 72, 6
    gt, init, con, dq, dd, frame, imu, sb0 = _imu_problem(K, band, per_kf=90, seed=53)
    if not with_imu:
        imu, sb0 = None, None
    opts = T.batch_tr_opts(max_iterations=12)
    one = _stage(K, band, con, dq, dd, frame, imu=imu)
    r1 = one.solve_tr(init, opts, speed_bias=sb0)
    one.close()
    ranks = batch.ThreadRanks(world, sync=torch.cuda.synchronize)
    stages = [_stage(K, band, con, dq, dd, frame, imu=imu, rank=r, world=world) for r in range(world)]

    def work(r, dist):
        out = stages[r].solve_tr(init, opts, dist, speed_bias=sb0)
        return out, stages[r].allreduce_sizes

    res = ranks.run(work)
    for st in stages:
        st.close()
    summ1 = r1[-1]
    for (out, sizes) in res:
        summ = out[-1]
        assert summ.iterations == summ1.iterations and summ.termination == summ1.termination and summ.successful_steps == summ1.successful_steps
        assert np.isclose(summ.final_cost, summ1.final_cost, rtol=1e-9)
        assert np.abs(out[0] - r1[0]).max() < 1e-9
        if with_imu:
            assert np.abs(out[1] - r1[1]).max() < 1e-8
        assert np.array_equal(out[0], res[0][0][0]), "every rank must hold the same result"
        # five hook calls per trust-region group plus the initial assembly; the largest buffer is the separator system or the assembly
        assert (len(sizes) - 1) % 5 == 0 and len(sizes) >= 1 + 5 * summ.iterations
        B = 15 if with_imu else 6
        assert max(sizes) < 0.5 * batch.hg_size(K, band) + 3 * (world - 1) * (6 * B) ** 2 + 3 * B * K
    P = po.BatchProblem(K, band, *con, dq=dq, dd=dd, frame=frame, imu=imu)
    if with_imu:
        want, wsb, wsum = P.solve2(init, opts, sb0)
    else:
        want, wsum = P.solve(init, opts)
    assert wsum.iterations == summ1.iterations and np.abs(res[0][0][0] - want).max() < 1e-8


def test_threshold_rounds_downweight_the_pseudorange_outliers():
    K, band = 40, 6
    gt, init, con, dq, dd, frame = _problem(K, band, seed=37)
    st = batch.BatchStage(K, band, len(con[0]))
    st.set_constraints(*con)
    odo = gt.copy()
    poses, hist = batch.solve_batch_rounds(st, init, odo, 3, dd, frame, opts=T.batch_tr_opts(max_iterations=30))
    assert len(hist) == 4 and all(h["termination_name"] != "FAILURE" for h in hist)
    # the first round (threshold 1e9) keeps every outlier at full weight; the later rounds start from a much lower cost
    assert hist[1]["initial_cost"] < 0.5 * hist[0]["final_cost"]
    from oracle import pyoracle as po
    ref = init.copy()
    for thr in batch.DDPSR_THRESHOLDS:
        for f in dd:
            f.threshold = thr
        ref, _ = po.BatchProblem(K, band, *con, dq=batch.delta_q_pairs(odo, 3), dd=dd, frame=frame).solve(ref, T.batch_tr_opts(max_iterations=30))
    assert np.abs(poses - ref).max() < 1e-7
    st.close()


def test_one_stage_through_changing_problems_equals_fresh_stages():
    """A BatchStage that is REUSED while its problem changes -- pose-only, then with the IMU chain, then another constraint set, then pose-only again,
    then other small factors -- must return exactly what a fresh stage returns for the same inputs (moment records, elimination workspaces,
    trust-region vectors, device status: nothing of an earlier problem may leak into a later one)."""
    K, band = 42, 6
    A = _imu_problem(K, band, per_kf=120, seed=61)
    B = _imu_problem(K, band, per_kf=90, seed=62)
    opts = T.batch_tr_opts(max_iterations=8)

    def fresh(P, with_imu, with_small=True):
        gt, init, con, dq, dd, frame, imu, sb0 = P
        st = _stage(K, band, con, dq if with_small else None, dd if with_small else [], frame, imu=imu if with_imu else None)
        out = st.solve_tr(init, opts, speed_bias=sb0 if with_imu else None)
        st.close()
        return out

    def same(a, b):
        return all(np.array_equal(x, y) for x, y in zip(a[:-1], b[:-1])) and a[-1].as_dict() == b[-1].as_dict()

    gt, init, con, dq, dd, frame, imu, sb0 = A
    st = _stage(K, band, con, dq, dd, frame)
    steps = []
    steps.append(("A pose-only", st.solve_tr(init, opts), fresh(A, False)))
    st.set_imu(imu)
    steps.append(("A with the IMU chain", st.solve_tr(init, opts, speed_bias=sb0), fresh(A, True)))
    gtB, initB, conB, dqB, ddB, frameB, imuB, sb0B = B
    st.set_constraints(*conB); st.set_small_factors(dqB, ddB, frameB); st.set_imu(imuB)
    steps.append(("B with the IMU chain", st.solve_tr(initB, opts, speed_bias=sb0B), fresh(B, True)))
    st.set_imu([])
    steps.append(("B pose-only", st.solve_tr(initB, opts), fresh(B, False)))
    st.set_small_factors(None, [], frameB)
    steps.append(("B pose-only, plane constraints alone", st.solve_tr(initB, opts), fresh(B, False, with_small=False)))
    st.set_constraints(*con); st.set_small_factors(dq, dd, frame); st.set_imu(imu)
    steps.append(("A with the IMU chain, again", st.solve_tr(init, opts, speed_bias=sb0), fresh(A, True)))
    st.close()
    for name, got, want in steps:
        assert same(got, want), name


_MOMENTS_AB = r"""
import sys, json
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
from glio_amd import batch
from glio_amd import ctypes_types as T
from test_hip_batch_tr import _imu_problem, _stage
K, band = 48, 6
gt, init, con, dq, dd, frame, imu, sb0 = _imu_problem(K, band, per_kf=200, seed=71, perturb=(1.0, 0.15))      # 1 m / ~9 degrees off
st = _stage(K, band, con, dq, dd, frame, imu=imu)
poses, sb, sm = st.solve_tr(init, T.batch_tr_opts(max_iterations=40), speed_bias=sb0)
print("AB", json.dumps({"it": sm.iterations, "ok": sm.successful_steps, "term": sm.termination, "c0": sm.initial_cost, "c1": sm.final_cost, "poses": poses.tolist(), "sb": sb.tolist()}))
"""


def test_moment_form_follows_the_streamed_form_from_a_far_start():
    """The moment records are centred at the solve's first poses; here the solve starts 1 m / 9 degrees away from the solution and the cost falls
    by five orders of magnitude, so later linearisations evaluate the moments far from their centre.  The run with the streamed K8
    (GLIO_BATCH_MOMENTS=0, read when the library loads: child processes) must take the same iterations and end at the same point."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for flag in ("1", "0"):
        env = dict(os.environ, GLIO_BATCH_MOMENTS=flag)
        p = subprocess.run([sys.executable, "-c", _MOMENTS_AB % (root, os.path.join(root, "tests"))], env=env, capture_output=True, text=True, timeout=600, cwd=root)
        assert p.returncode == 0 and "AB " in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]
        outs.append(json.loads(p.stdout.split("AB ", 1)[1]))
    a, b = outs
    assert a["it"] == b["it"] and a["ok"] == b["ok"] and a["term"] == b["term"], (a["it"], b["it"], a["term"], b["term"])
    assert a["c1"] < 1e-4 * a["c0"]
    assert np.isclose(a["c0"], b["c0"], rtol=1e-11) and np.isclose(a["c1"], b["c1"], rtol=1e-8)
    assert np.abs(np.array(a["poses"]) - np.array(b["poses"])).max() < 1e-8 and np.abs(np.array(a["sb"]) - np.array(b["sb"])).max() < 1e-7


@pytest.mark.parametrize("K,world,windows", [(50, 1, True), (61, 1, True), (50, 2, True), (48, 1, False), (73, 3, False)])
def test_imu_chain_under_the_references_end_windows_band_12(K, world, windows):
    """The 15-state problem AS THE REFERENCE BUILDS IT: the first and the last search_range = 6 keyframes search a window of 13 keyframes
    (Estimator.cpp:3009-3017), so the pose band is 12 there (6 in the interior) -- until round 4 glio_batch_set_imu refused band > 6.
    Super-blocks of 12 keyframes, the speed-bias blocks of the ten inner keyframes pre-eliminated (k_bcr_pre12 / k_bcr_post12), 90 x 90 nodes.
    `windows` False: every keyframe couples to +-12 (a full band-12 problem).  Against the banded oracle, on one rank and on virtual ranks."""
    from oracle import pyoracle as po
    import torch
    band, sr = 12, 6
    gt, init = batch.make_poses(K, seed=77 + K, perturb=(0.08, 0.004))
    ci, cj, cp, nc, score = batch.make_constraints(gt, 0, K, 156, band, seed=77 + K, search_range=sr if windows else None)
    if windows:
        assert np.abs(ci - cj).max() == 12 and np.abs(ci - cj)[(ci >= 13) & (ci < K - 13)].max() == 6        # +-12 at the ends only
    con = (ci, cj, cp.numpy(), nc.numpy(), score.numpy())
    rng = np.random.default_rng(77 + K)
    odo = gt.copy(); odo[:, :3] += rng.normal(0, 0.02, (K, 3))
    dq = batch.delta_q_pairs(odo, sr)
    dd, frame = batch.make_batch_gnss(gt, seed=77 + K)
    for f in dd:
        f.threshold = 10.0
    imu, _, sb0 = batch.make_batch_imu(K, seed=77 + K)
    opts = T.batch_tr_opts(max_iterations=12)
    P = po.BatchProblem(K, band, *con, dq=dq, dd=dd, frame=frame, imu=imu)
    want, wsb, wsum = P.solve2(init, opts, sb0)
    if world == 1:
        st = _stage(K, band, con, dq, dd, frame, imu=imu)
        diag, g, cost = st.linearize_full(init, sb0)
        Hb, gw, cw = P.linearize_banded(init, sb0)
        assert abs(cost - cw) <= 1e-11 * cw and np.abs(diag - Hb[:, -1]).max() <= 1e-10 * Hb[:, -1].max() and np.abs(g - gw).max() <= 1e-10 * np.abs(gw).max()
        poses, sb, summ = st.solve_tr(init, opts, speed_bias=sb0)
        st.close()
    else:
        stages = [_stage(K, band, con, dq, dd, frame, imu=imu, rank=r, world=world) for r in range(world)]
        ranks = batch.ThreadRanks(world, sync=torch.cuda.synchronize)
        res = ranks.run(lambda r, d: stages[r].solve_tr(init, opts, d, speed_bias=sb0))
        poses, sb, summ = res[0]
        for other in res[1:]:
            assert np.array_equal(other[0], poses) and np.array_equal(other[1], sb)
        for s_ in stages:
            s_.close()
    assert summ.iterations == wsum.iterations and summ.successful_steps == wsum.successful_steps and summ.termination == wsum.termination, (summ.as_dict(), wsum.as_dict())
    assert np.isclose(summ.final_cost, wsum.final_cost, rtol=1e-8)
    assert np.abs(poses - want).max() < 1e-7 and np.abs(sb - wsb).max() < 1e-6
    assert summ.final_cost < 0.01 * summ.initial_cost


@pytest.mark.parametrize("with_planes", [False, True], ids=["sms_fusion_level_0", "mixed"])
@pytest.mark.parametrize("world", [1, 2])
def test_relative_pose_factors_of_the_released_default(with_planes, world):
    """sms_fusion_level == 0 (config_urban_hk.yaml:63, the SHIPPED default): the scan-to-multiscan constraints are LidarPoseFactorBatchRelativeAutoDiff
    factors between keyframes up to search_range - 1 apart (Estimator.cpp:2897-2955), next to the delta_q and DD factors -- no plane constraints,
    no IMU.  The factor is a role of the small-factor kernel (type 2); its oracle restatement is pinned on the reference's own Jets
    (tests/test_oracle_ref.py::test_relative_pose_factor).  Linearisation and trust-region solve vs the oracle; `mixed` adds plane constraints."""
    from oracle import pyoracle as po
    import torch
    K, band, sr = 60, 6, 6
    gt, init = batch.make_poses(K, seed=88, perturb=(0.08, 0.004))
    rng = np.random.default_rng(88)
    odo = gt.copy(); odo[:, :3] += rng.normal(0, 0.02, (K, 3))
    rp = batch.relative_pose_pairs(odo, sr)
    assert len(rp[0]) == 2 * (K - sr) * (sr - 1) and np.abs(rp[0] - rp[1]).max() == sr - 1
    dq = batch.delta_q_pairs(odo, sr)
    dd, frame = batch.make_batch_gnss(gt, seed=88)
    for f in dd:
        f.threshold = 10.0
    if with_planes:
        ci, cj, cp, nc, score = batch.make_constraints(gt, 0, K, 120, band, seed=88)
        con = (ci, cj, cp.numpy(), nc.numpy(), score.numpy())
    else:
        con = (np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 4), np.float32), np.zeros((0, 6)), np.zeros(0))
    P = po.BatchProblem(K, band, *con, dq=dq, dd=dd, frame=frame, rp=rp)
    opts = T.batch_tr_opts(max_iterations=20)
    want, wsum = P.solve(init, opts)

    def stage(rank):
        lo, hi = batch.shard_range(K, rank, world, band)
        own = (con[0] >= lo) & (con[0] < hi)
        st = batch.BatchStage(K, band, max(1, int(own.sum())))
        if world > 1:
            st.set_shard(rank, world)
        st.set_constraints(*[c[own] for c in con])
        st.set_small_factors(dq, dd, frame, rp=rp)
        return st
    if world == 1:
        st = stage(0)
        Hg = st.new_hg()
        st.linearize(init, Hg); st.add_small(init, Hg)
        got = Hg.cpu().numpy()
        H, g, cost = P.linearize(init)
        nH = K * (band + 1) * 36
        assert np.abs(got[:nH] - H.ravel()).max() <= 1e-11 * np.abs(H).max()
        assert np.abs(got[nH:-1] - g.ravel()).max() <= 1e-11 * np.abs(g).max() and abs(got[-1] - cost) <= 1e-12 * cost
        poses, summ = st.solve_tr(init, opts)
        st.close()
    else:
        stages = [stage(r) for r in range(world)]
        ranks = batch.ThreadRanks(world, sync=torch.cuda.synchronize)
        res = ranks.run(lambda r, d: stages[r].solve_tr(init, opts, d))
        poses, summ = res[0]
        assert np.array_equal(res[1][0], poses)
        for s_ in stages:
            s_.close()
    assert summ.iterations == wsum.iterations and summ.successful_steps == wsum.successful_steps and summ.termination == wsum.termination, (summ.as_dict(), wsum.as_dict())
    assert np.isclose(summ.final_cost, wsum.final_cost, rtol=1e-9) and np.abs(poses - want).max() < 1e-8
    assert summ.final_cost < 0.05 * summ.initial_cost
    rel = lambda X: np.linalg.norm(np.diff(X[:, :3], axis=0) - np.diff(odo[:, :3], axis=0), axis=1).max()
    assert rel(poses) < 0.5 * rel(init)                      # the relative-pose factors pull the chain onto the odometry's increments


@pytest.mark.parametrize("world", [1, 3])
def test_groups_in_flight_do_not_change_the_solve_or_the_collective_count(world):
    """glio_batch_solve_tr2 keeps `lead` trust-region groups in flight (group g is enqueued when group g - lead has decided that the solve goes on; the groups
    behind the deciding one exit at once on the device).  lead 1 (wait for every decision: the round-4 loop), 2 (default) and 4 give the same iterates bit
    for bit, and the number of groups -- hence of collective calls, which must match across ranks -- is a function of the decisions alone: groups(lead) =
    groups(1) + lead - 1 on EVERY rank."""
    import torch
    from glio_amd import capi
    K, band = 60, 6
    gt, init, con, dq, dd, frame, imu, sb0 = _imu_problem(K, band, per_kf=80, seed=57)
    opts = T.batch_tr_opts(max_iterations=10)
    lib = capi.load()
    runs = {}
    for lead in (1, 2, 4):
        stages = [_stage(K, band, con, dq, dd, frame, imu=imu, rank=r, world=world) for r in range(world)]
        for st in stages:
            assert lib.glio_batch_debug_set_enqueue_lead(st._h, lead) == 0
            st.counters()
        if world == 1:
            res = [(stages[0].solve_tr(init, opts, speed_bias=sb0), stages[0].counters())]
        else:
            ranks = batch.ThreadRanks(world, sync=torch.cuda.synchronize)
            res = ranks.run(lambda r, dist: (stages[r].solve_tr(init, opts, dist, speed_bias=sb0), stages[r].counters()))
        for st in stages:
            st.close()
        runs[lead] = res
    base = runs[1]
    g1 = base[0][1]["groups"]
    for lead, res in runs.items():
        for (out, cnt), (out1, cnt1) in zip(res, base):
            assert np.array_equal(out[0], out1[0]) and np.array_equal(out[1], out1[1]) and out[2].iterations == out1[2].iterations and out[2].final_cost == out1[2].final_cost
            assert cnt["groups"] == g1 + lead - 1, (lead, cnt, g1)
            if world > 1:
                assert cnt["hook_calls"] == cnt1["hook_calls"] + 5 * (lead - 1)
