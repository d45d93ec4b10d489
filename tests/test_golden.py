"""Committed fixture tests/golden/window_small.npz (made by tests/golden/make_golden.py from the CPU oracle; NOT reference
output -- the reference cannot be run here, parity stays unpinned): the oracle still reproduces it (no GPU needed), and the
HIP path agrees with the frozen numbers (GPU)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as mg      # noqa: E402


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(HERE, "golden", "window_small.npz"))


@pytest.fixture(scope="module")
def case():
    return mg.make_case()


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(1e-300, np.abs(np.asarray(b)).max())


def test_inputs_and_oracle_reproduce_the_fixture(golden, case):
    win, corr, counts, head = case
    assert mg.input_digest(win) == str(golden["input_sha256"]), "the synthetic generator changed: inputs differ from the fixture's"
    assert np.array_equal(counts, golden["assoc_counts"])
    assert np.array_equal(head, golden["assoc_head"])                       # association is float/bit-exact work
    out = mg.oracle_outputs(win, corr)
    assert int(out["iterations"]) == int(golden["iterations"])
    for k in ("H", "g", "cost", "final_cost", "marg_S", "marg_b", "marg_c"):
        assert rel(out[k], golden[k]) <= 1e-11, k
    for k in ("sol_trans", "sol_quat", "sol_speed_bias", "sol_rcv_ddt"):
        assert np.abs(out[k] - golden[k]).max() <= 1e-10, k


@pytest.mark.gpu
def test_hip_path_matches_the_fixture(golden, case):
    from glio_amd import capi
    from glio_amd.capi import lidar_pose
    win, _, _, _ = case
    ctx = capi.Context(win.opts)
    ctx.set_map(win.map_pts)
    ctx.set_imu(win.preints); ctx.set_prior(win.prior); ctx.set_gnss(win.frame, win.dd, win.dop)
    head = []
    for s in range(win.W):
        q2, t2 = lidar_pose(win.opts, win.init.quat[s], win.init.trans[s])
        n = ctx.associate(s, win.scans[s], q2, t2)
        assert n == int(golden["assoc_counts"][s])
        pts, pl, sc = ctx.get_correspondences(s)
        head.append(np.concatenate([pts[:8].ravel(), pl[:8].ravel(), sc[:8]]))
    assert np.array_equal(np.array(head), golden["assoc_head"])             # bit-exact
    H, g, cost = ctx.linearize(win.init)
    assert rel(H, golden["H"]) <= 1e-10 and rel(g, golden["g"]) <= 1e-10 and abs(cost - float(golden["cost"])) <= 1e-10 * abs(cost)
    sol, summ = ctx.solve(win.init)
    assert summ.iterations == int(golden["iterations"])
    assert abs(summ.final_cost - float(golden["final_cost"])) <= 1e-9 * abs(summ.final_cost)
    assert np.abs(sol.trans - golden["sol_trans"]).max() <= 1e-8 and np.abs(sol.quat - golden["sol_quat"]).max() <= 1e-9
    assert np.abs(sol.speed_bias - golden["sol_speed_bias"]).max() <= 1e-7
    m = ctx.marginalize(sol)
    J0 = np.asarray(m["lin_jac"]); r0 = np.asarray(m["lin_res"])
    assert rel(J0.T @ J0, golden["marg_S"]) <= 1e-7 and rel(J0.T @ r0, golden["marg_b"]) <= 1e-7
    ctx.close()


# ---- the rows SURVEY 8f ranks next (f2 batch association, f3 front-end odometry, f4 local map) -----------------------
import make_golden_next as mgn      # noqa: E402


@pytest.fixture(scope="module")
def golden_next():
    return np.load(os.path.join(HERE, "golden", "next_rows_small.npz"))


@pytest.fixture(scope="module")
def case_next():
    return mgn.make_inputs()


def test_next_rows_oracle_reproduces_the_fixture(golden_next, case_next):
    win, body = case_next
    assert mgn.digest(win, body) == str(golden_next["input_sha256"]), "the synthetic generator changed"
    out = mgn.oracle_outputs(win, body)
    assert int(out["pair_count"]) == int(golden_next["pair_count"]) and int(out["map_count"]) == int(golden_next["map_count"])
    for k in ("pair_cp", "pair_nc", "pair_score", "odo_iterations", "odo_kept"):
        assert np.array_equal(out[k], golden_next[k]), k                      # float / index work: exact
    assert np.abs(out["odo_pose"] - golden_next["odo_pose"]).max() <= 1e-11
    assert np.abs(out["map_head"] - golden_next["map_head"]).max() <= 1e-6


@pytest.mark.gpu
def test_next_rows_hip_matches_the_fixture(golden_next, case_next):
    from glio_amd import batch, capi, odometry, synth
    win, body = case_next
    # f2: one keyframe pair, bit-exact records in the same order
    ba = batch.BatchAssociation(2, 2048, 100000)
    ba.set_frame(0, body[0]); ba.set_frame(1, body[1])
    poses = np.c_[win.init.trans, win.init.quat][:2]
    counts, total = ba.run(poses, np.array([0], np.int32), np.array([1], np.int32))
    cp, nc, sc = ba.read()
    assert total == int(golden_next["pair_count"])
    assert np.array_equal(cp[:16], golden_next["pair_cp"]) and np.array_equal(nc[:16], golden_next["pair_nc"]) and np.array_equal(sc[:16], golden_next["pair_score"])
    ba.close()
    # f3: two matching rounds of the front end
    o = odometry.frontend_opts(len(body[0]), len(win.map_pts))
    ctx = capi.Context(o)
    odo = odometry.ScanToMapOdometry(ctx)
    odo.set_map(win.map_pts)
    pose, rounds = odo.update(body[0], np.r_[win.init.quat[0], win.init.trans[0]], match_cnt=2)
    assert [int(r[0].iterations) for r in rounds] == list(golden_next["odo_iterations"])
    assert [int(r[1]) for r in rounds] == list(golden_next["odo_kept"])
    assert np.abs(pose - golden_next["odo_pose"]).max() <= 1e-8
    ctx.close()
    # f4: voxel-grid local map of the three-keyframe ring
    lm = capi.Context(synth.default_opts(1, pts=2048, map_pts=1 << 15))
    lm.localmap_config(3, 0.4, 2048)
    for s in range(3):
        lm.localmap_push(body[s], win.gt.quat[s], win.gt.trans[s])
    assert lm.localmap_build() == int(golden_next["map_count"])
    assert np.abs(lm.localmap_read()[:32] - golden_next["map_head"]).max() <= 2e-5
    lm.close()


# ---- the batch pose problem (plane constraints + delta_q + DD pseudoranges, four threshold rounds) --------------------
import make_golden_batch as mgb      # noqa: E402


@pytest.fixture(scope="module")
def golden_batch():
    return np.load(os.path.join(HERE, "golden", "batch_small.npz"))


@pytest.fixture(scope="module")
def case_batch():
    return mgb.make_inputs()


def test_batch_oracle_reproduces_the_fixture(golden_batch, case_batch):
    assert mgb.digest(case_batch) == str(golden_batch["input_sha256"]), "the synthetic generator changed"
    out = mgb.oracle_outputs(case_batch)
    for k in ("dq_i", "dq_j", "round_iterations", "round_termination"):
        assert np.array_equal(out[k], golden_batch[k]), k
    assert np.abs(out["dq_const"] - golden_batch["dq_const"]).max() <= 1e-15
    assert rel(out["lin_H"], golden_batch["lin_H"]) <= 1e-13 and rel(out["lin_g"], golden_batch["lin_g"]) <= 1e-13
    assert abs(float(out["lin_cost"]) - float(golden_batch["lin_cost"])) <= 1e-13 * float(golden_batch["lin_cost"])
    assert rel(out["round_costs"], golden_batch["round_costs"]) <= 1e-10
    assert np.abs(out["poses"] - golden_batch["poses"]).max() <= 1e-9


@pytest.mark.gpu
def test_batch_hip_matches_the_fixture(golden_batch, case_batch):
    from glio_amd import batch
    from glio_amd import ctypes_types as T
    c = case_batch
    dq = batch.delta_q_pairs(c["odo"], mgb.SEARCH_RANGE)
    assert np.array_equal(dq[0], golden_batch["dq_i"]) and np.array_equal(dq[1], golden_batch["dq_j"])
    st = batch.BatchStage(mgb.K, mgb.BAND, len(c["con"][0]))
    st.set_constraints(*c["con"])
    st.set_small_factors(dq, c["dd"], c["frame"], threshold=batch.DDPSR_THRESHOLDS[0])
    Hg = st.new_hg()
    st.linearize(c["init"], Hg); st.add_small(c["init"], Hg)
    got = Hg.cpu().numpy()
    nH = mgb.K * (mgb.BAND + 1) * 36
    assert rel(got[:nH], golden_batch["lin_H"].ravel()) <= 1e-12 and rel(got[nH:-1], golden_batch["lin_g"].ravel()) <= 1e-12
    assert abs(got[-1] - float(golden_batch["lin_cost"])) <= 1e-12 * got[-1]
    poses, hist = batch.solve_batch_rounds(st, c["init"], c["odo"], mgb.SEARCH_RANGE, c["dd"], c["frame"], opts=T.batch_tr_opts(mgb.MAX_ITER))
    assert [h["iterations"] for h in hist] == list(golden_batch["round_iterations"])
    assert [h["termination"] for h in hist] == list(golden_batch["round_termination"])
    assert rel(np.array([[h["initial_cost"], h["final_cost"]] for h in hist]), golden_batch["round_costs"]) <= 1e-8
    assert np.abs(poses - golden_batch["poses"]).max() <= 1e-7
    st.close()


# ---- the same batch with the ImuFactor chain (15 unknowns per keyframe, SUBSPACE_DOGLEG, minimum-cost iterate) --------------------
@pytest.fixture(scope="module")
def golden_batch_imu():
    return np.load(os.path.join(HERE, "golden", "batch_imu_small.npz"))


def test_batch_imu_oracle_reproduces_the_fixture(golden_batch_imu, case_batch):
    imu, sb0 = mgb.imu_inputs()
    assert mgb.digest(case_batch) == str(golden_batch_imu["input_sha256"]) and mgb.imu_digest(imu, sb0) == str(golden_batch_imu["imu_sha256"]), "the synthetic generator changed"
    out = mgb.oracle_outputs_imu(case_batch, imu, sb0)
    g = golden_batch_imu
    assert int(out["imu_iterations"]) == int(g["imu_iterations"]) and int(out["imu_termination"]) == int(g["imu_termination"]) and int(out["imu_successful"]) == int(g["imu_successful"])
    assert rel(out["imu_diag"], g["imu_diag"]) <= 1e-13 and rel(out["imu_g"], g["imu_g"]) <= 1e-12
    assert rel(out["imu_costs"], g["imu_costs"]) <= 1e-10 and rel(out["imu_history"][:, :2], g["imu_history"][:, :2]) <= 1e-8
    assert np.abs(out["imu_poses"] - g["imu_poses"]).max() <= 1e-9 and np.abs(out["imu_sb"] - g["imu_sb"]).max() <= 1e-8


@pytest.mark.gpu
def test_batch_imu_hip_matches_the_fixture(golden_batch_imu, case_batch):
    from glio_amd import batch
    from glio_amd import ctypes_types as T
    c, g = case_batch, golden_batch_imu
    imu, sb0 = mgb.imu_inputs()
    dq = batch.delta_q_pairs(c["odo"], mgb.SEARCH_RANGE)
    st = batch.BatchStage(mgb.K, mgb.BAND, len(c["con"][0]))
    st.set_constraints(*c["con"])
    st.set_small_factors(dq, c["dd"], c["frame"], threshold=10.0)
    st.set_imu(imu)
    diag, grad, cost = st.linearize_full(c["init"], sb0)
    assert rel(diag, g["imu_diag"]) <= 1e-11 and rel(grad, g["imu_g"]) <= 1e-10 and abs(cost - float(g["imu_cost"])) <= 1e-11 * cost
    opts = T.batch_tr_opts(mgb.MAX_ITER)
    opts.initial_trust_region_radius = 2.0
    poses, sb, summ = st.solve_tr(c["init"], opts, speed_bias=sb0)
    assert summ.iterations == int(g["imu_iterations"]) and summ.termination == int(g["imu_termination"]) and summ.successful_steps == int(g["imu_successful"])
    assert rel(np.array([summ.initial_cost, summ.final_cost]), g["imu_costs"]) <= 1e-8
    assert np.abs(poses - g["imu_poses"]).max() <= 1e-7 and np.abs(sb - g["imu_sb"]).max() <= 1e-6
    st.close()
