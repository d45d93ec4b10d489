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


@pytest.mark.parametrize("with_small", [False, True])
def test_trust_region_solve_follows_the_oracle(with_small):
    K, band = 60, 6
    gt, init, con, dq, dd, frame = _problem(K, band, seed=35)
    if not with_small:
        dq, dd = None, []
    for f in dd:
        f.threshold = 10.0
    st = batch.BatchStage(K, band, len(con[0]))
    st.set_constraints(*con)
    st.set_small_factors(dq, dd, frame)
    opts = T.batch_tr_opts(max_iterations=30)
    poses, summ = st.solve_tr(init, opts)
    want, wsum = _oracle(K, band, con, dq, dd, frame).solve(init, opts)
    assert summ.iterations == wsum.iterations and summ.successful_steps == wsum.successful_steps and summ.termination == wsum.termination
    assert np.isclose(summ.initial_cost, wsum.initial_cost, rtol=1e-12)
    assert np.isclose(summ.final_cost, wsum.final_cost, rtol=1e-9)
    assert np.abs(poses - want).max() < 1e-8
    assert summ.final_cost < 0.05 * summ.initial_cost
    if not with_small:
        assert np.abs(poses[:, :3] - gt[:, :3]).max() < 0.05
    st.close()


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


def test_allreduce_hook_sees_the_library_buffer_once_per_linearisation():
    """The hook path of solve_tr with a stand-in for torch.distributed (one GPU here): the device buffer handed to the hook is the
    [H | g | cost] of this rank BEFORE the small factors; summing it with itself (a 2-rank job whose ranks hold the same shard)
    equals the solve with every plane constraint given twice."""
    import torch
    K, band = 30, 6
    gt, init, con, dq, dd, frame = _problem(K, band, seed=39)

    class FakeDist:
        class ReduceOp:
            SUM = "sum"

        def __init__(self):
            self.seen = []

        def all_reduce(self, t, op=None):
            assert t.is_cuda and t.dtype == torch.float64 and t.numel() == batch.hg_size(K, band)
            self.seen.append(float(t[-1].item()))
            t.mul_(2.0)

    st = batch.BatchStage(K, band, len(con[0]))
    st.set_constraints(*con)
    st.set_small_factors(dq, dd, frame, threshold=10.0)
    fd = FakeDist()
    opts = T.batch_tr_opts(max_iterations=6)
    poses, summ = st.solve_tr(init, opts, dist=fd)
    assert st.allreduces == len(fd.seen) and len(fd.seen) >= 1 + summ.successful_steps
    st.close()
    twice = [np.concatenate([c, c]) for c in con]
    order = np.lexsort((twice[1], twice[0]))
    twice = [c[order] for c in twice]
    st2 = batch.BatchStage(K, band, len(twice[0]))
    st2.set_constraints(*twice)
    st2.set_small_factors(dq, dd, frame, threshold=10.0)
    want, wsum = st2.solve_tr(init, opts)
    assert summ.iterations == wsum.iterations
    assert np.isclose(summ.final_cost, wsum.final_cost, rtol=1e-10)
    assert np.abs(poses - want).max() < 1e-9
    st2.close()
