"""CPU tests of the host-side mirrors of the reference's call sequences (no GPU, no HIP library): quaternion sign
unification (Estimator.cpp:2439-2457), the featureSelection draw procedure (:3894-3992), the batch search-window rule
(:3009-3017), lidar_pose (:2216-2221), and the streaming driver's sequence against a recording backend."""
import numpy as np

from glio_amd import batch, capi, sliding, synth
from glio_amd import ctypes_types as T


def test_unify_quaternions_flips_only_negative_w():
    st = T.WindowState(3)
    st.quat[:] = [[0.5, 0.5, 0.5, 0.5], [-0.5, 0.5, -0.5, 0.5], [0.0, 1.0, 0.0, 0.0]]
    ref = st.quat.copy()
    sliding.unify_quaternions(st)
    assert np.array_equal(st.quat[0], ref[0]) and np.array_equal(st.quat[1], -ref[1]) and np.array_equal(st.quat[2], ref[2])


def test_feature_selection_draws_follow_the_reference_rules():
    rng = np.random.default_rng(5)
    assert sliding.feature_selection_draws(100, 100, rng) is None              # count - 1 < feature_res_num: untouched (Q9)
    assert sliding.feature_selection_draws(0, 10, rng) is None
    assert len(sliding.feature_selection_draws(101, 100, rng, random_select=False)) == 0     # :3945: the set is emptied
    sel = sliding.feature_selection_draws(500, 120, rng)
    assert len(sel) == 120 and len(set(sel.tolist())) == 120 and sel.min() >= 0 and sel.max() < 500   # no repeats
    a = sliding.feature_selection_draws(500, 120, np.random.default_rng(9))
    b = sliding.feature_selection_draws(500, 120, np.random.default_rng(9))
    assert np.array_equal(a, b)                                                 # the caller's generator decides


def test_search_window_is_centred_and_clamped():
    K, r = 20, 3
    for idx in range(K):
        s0 = batch.search_window(idx, K, r)
        assert 0 <= s0 and s0 + 2 * r + 1 <= K
        assert s0 <= idx <= s0 + 2 * r                                          # the keyframe lies inside its own window
    assert batch.search_window(10, K, r) == 7 and batch.search_window(0, K, r) == 0 and batch.search_window(K - 1, K, r) == K - 2 * r - 1
    ci, cj = batch.pair_list(K, r)
    assert len(ci) == K * 2 * r and np.all(ci != cj) and np.all(np.diff(ci) >= 0)


def test_lidar_pose_is_body_pose_times_inverse_extrinsic():
    o = synth.default_opts(2)
    q_lb = synth.rotvec_q(np.array([0.02, -0.03, 0.5])); t_lb = np.array([0.1, -0.2, 0.28])
    o.q_lb[:] = q_lb; o.t_lb[:] = t_lb
    q = synth.rotvec_q(np.array([0.3, 0.1, -0.7])); t = np.array([4.0, -2.0, 1.0])
    q2, t2 = capi.lidar_pose(o, q, t)
    R, R2, Rlb = synth.q2R(q), synth.q2R(q2), synth.q2R(q_lb)
    assert np.allclose(R2 @ Rlb, R, atol=1e-13)                                  # Q2 = Q q_lb^-1
    assert np.allclose(R2 @ t_lb + t2, t, atol=1e-13)                            # T2 = T - Q2 t_lb


class _Recorder:
    """Backend double: records the call sequence, returns the input state unchanged."""

    def __init__(self): self.calls = []
    def set_map(self, m): self.calls.append("set_map")
    def associate(self, s, scan, q, t): self.calls.append(f"associate{s}"); return 7
    def set_imu(self, p): self.calls.append("set_imu")
    def set_prior(self, p): self.calls.append(("set_prior", p))
    def set_gnss(self, f, a, b): self.calls.append("set_gnss")
    def solve(self, st): self.calls.append("solve"); return st.copy(), "summary"
    def marginalize(self, st): self.calls.append("marginalize"); return {"n": 1}


def test_streaming_driver_call_sequence():
    W = 3
    o = synth.default_opts(W)
    be = _Recorder()
    drv = sliding.SlidingWindowDriver(be, o)
    st = T.WindowState(W); st.quat[:, 0] = 1.0; st.quat[1] = [-1.0, 0, 0, 0]
    drv.start(st)
    the_map = np.zeros((sliding.MIN_MAP_POINTS + 1, 4), np.float32)               # one point more than the guard of Estimator.cpp:2221 asks for
    sol, summ, counts = drv.step(the_map, [None] * W, [])
    assert counts == [7] * W and sol.quat[1, 0] == 1.0                            # unified sign
    names = [c if isinstance(c, str) else c[0] for c in be.calls]
    assert names == ["set_map", "associate0", "associate1", "associate2", "set_imu", "set_prior", "set_gnss", "solve", "marginalize"]
    assert be.calls[5][1] is None                                                 # first window: no prior
    drv.slide(np.ones(3), np.array([1.0, 0, 0, 0]), np.zeros(9))
    assert drv.first == 1 and np.array_equal(drv.state.trans[-1], np.ones(3))
    be.calls.clear()
    drv.step(the_map, [None] * W, [])
    assert [c for c in be.calls if not isinstance(c, str)][0][1] == {"n": 1}      # the marginalization result is the next prior


def test_streaming_driver_skips_the_search_on_a_small_map():
    """`if (surf_local_map_ds->points.size() > 50)` (Estimator.cpp:2221, 2244): with 50 map points or fewer no slot is associated -- the
    window is solved on its IMU / GNSS / prior factors alone; 51 points are enough."""
    W = 3
    o = synth.default_opts(W)

    class _Rec(_Recorder):
        def set_correspondences(self, s, p, pl, sc): self.calls.append(f"empty{s}"); assert len(p) == 0 and len(pl) == 0 and len(sc) == 0

    for n, searched in ((sliding.MIN_MAP_POINTS, False), (sliding.MIN_MAP_POINTS + 1, True)):
        be = _Rec()
        drv = sliding.SlidingWindowDriver(be, o)
        st = T.WindowState(W); st.quat[:, 0] = 1.0
        drv.start(st)
        sol, summ, counts = drv.step(np.zeros((n, 4), np.float32), [None] * W, [])
        names = [c for c in be.calls if isinstance(c, str)]
        if searched:
            assert counts == [7] * W and "associate0" in names and "empty0" not in names
        else:
            assert counts == [0] * W and names[:4] == ["set_map", "empty0", "empty1", "empty2"] and "associate0" not in names
        assert "solve" in names and "marginalize" in names


def test_write_back_gates_cpp_against_transcription(tmp_path):
    """glio::writeBackState (glio_amd/host/glio_backend.hpp, Estimator.cpp:2611-2726) compiled with plain g++ and driven with
    states that trip each gate: |dp| >= 100 keeps Ps and abs_poses t, |dq.vec| >= 10 can never trip for unit quaternions (the
    gate is vacuous, replicated anyway), |dv| >= 100 keeps Vs, and of the six bias components only the FIRST that passes its
    |db| < 22 gate is written (dangling else chain, quirk Q16); rcv_dt is copied unconditionally."""
    import os
    import subprocess
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "glio_amd", "host")
    exe = str(tmp_path / "wb")
    subprocess.check_call(["g++", "-std=c++14", "-O1", os.path.join(here, "host_writeback_test.cpp"), "-I" + os.path.join(here, "..", "..", "include"), "-o", exe])
    rng = np.random.default_rng(5)
    W = 4
    tT, tQ, tSB, tDt = rng.normal(size=(W, 3)), rng.normal(size=(W, 4)), rng.normal(size=(W, 9)), rng.normal(size=(W, 3))
    Ps, Vs, psb = tT + rng.normal(size=(W, 3)), tSB[:, :3] + rng.normal(size=(W, 3)), tSB + rng.normal(size=(W, 9))
    Qs = tQ / np.linalg.norm(tQ, axis=1, keepdims=True)
    Ps[1] += 500.0                  # trips the position gate of keyframe 1
    Vs[2] += 300.0                  # trips the velocity gate of keyframe 2
    psb[3, 3] += 40.0               # ba_x of keyframe 3 fails -> ba_y is the first to pass and the only one written
    psb[0, 3:] += 40.0              # every bias component of keyframe 0 fails: nothing written
    Bas, Bgs, ap, dt = np.full((W, 3), -7.0), np.full((W, 3), -8.0), np.full((W, 7), -9.0), np.zeros((W, 3))
    arrs = [tT, tQ, tSB, tDt, Ps, Qs, Vs, psb, Bas, Bgs, ap, dt]
    txt = str(W) + "\n" + "\n".join(" ".join(repr(float(x)) for x in a.ravel()) for a in arrs)
    out = subprocess.run([exe], input=txt, capture_output=True, text=True, check=True).stdout.splitlines()
    got = {ln.split()[0]: np.array([float(x) for x in ln.split()[1:]]) for ln in out}
    # transcription
    ePs, eQs, eVs, epsb, eBas, eBgs, eap = Ps.copy(), Qs.copy(), Vs.copy(), psb.copy(), Bas.copy(), Bgs.copy(), ap.copy()
    for i in range(W):
        if np.linalg.norm(Ps[i] - tT[i]) < 100:
            ePs[i] = tT[i]; eap[i, 4:] = tT[i]
        eQs[i] = tQ[i] / np.linalg.norm(tQ[i]); eap[i, :4] = tQ[i]        # |dq.vec| <= 1 < 10 always
        if np.linalg.norm(Vs[i] - tSB[i, :3]) < 100:
            eVs[i] = tSB[i, :3]; epsb[i, :3] = tSB[i, :3]
        for k in range(3, 9):
            if abs(psb[i, k] - tSB[i, k]) < 22:
                epsb[i, k] = tSB[i, k]
                (eBas if k < 6 else eBgs)[i, (k - 3) % 3] = tSB[i, k]
                break
    for name, e in (("Ps", ePs), ("Qs", eQs), ("Vs", eVs), ("psb", epsb), ("Bas", eBas), ("Bgs", eBgs), ("abs", eap), ("dt", tDt)):
        assert np.allclose(got[name], e.ravel(), rtol=1e-15, atol=0), name
    assert np.array_equal(got["Ps"].reshape(W, 3)[1], Ps[1]) and np.array_equal(got["Vs"].reshape(W, 3)[2], Vs[2])
    assert got["Bas"].reshape(W, 3)[3, 1] == tSB[3, 4] and got["Bas"].reshape(W, 3)[3, 0] == -7.0 and np.all(got["Bas"].reshape(W, 3)[0] == -7.0)


def test_batch_selection_draw_rules():
    from glio_amd import batch
    rng = np.random.default_rng(1)
    assert batch.batch_selection_draws(25, 25, rng) is None and batch.batch_selection_draws(3, 25, rng) is None
    d = batch.batch_selection_draws(40, 25, rng)
    assert len(d) == 25 and len(set(d.tolist())) == 25 and d.max() <= 38              # the last record (39) is never drawn
    assert batch.batch_selection_draws(49, 25, rng, ends=True) == "return" and batch.batch_selection_draws(25, 25, rng, ends=True) == "return"
    d = batch.batch_selection_draws(60, 25, rng, ends=True)
    assert len(d) == 25 and d.max() <= 58
    assert len(batch.batch_selection_draws(50, 40, rng, ends=True, rand_set_num=400)) == 9       # rand set clamped to count - res_num - 1


def test_delta_q_pairs_follow_the_reference_walk():
    """Estimator.cpp:2831-2891: backward then forward, factor_count shared between the two walks and reset only at search_range,
    the distance gate an integer division (0 for search_range 6, 1 for search_range 3..5), q_i sign-unified."""
    from glio_amd import batch
    K = 8
    odo = np.zeros((K, 7)); odo[:, 0] = np.arange(K) * 2.0; odo[:, 3] = 1.0
    i, j, c = batch.delta_q_pairs(odo, 3)
    pairs = list(zip(i.tolist(), j.tolist()))
    assert [p for p in pairs if p[0] == 0] == [(0, 1), (0, 2), (0, 3)]
    assert [p for p in pairs if p[0] == 1] == [(1, 0), (1, 2), (1, 3)]            # one backward, so only two forward before the count resets
    assert [p for p in pairs if p[0] == 4] == [(4, 3), (4, 2), (4, 1), (4, 5), (4, 6), (4, 7)]
    assert np.allclose(c, [1, 0, 0, 0])
    # gate: spacing 0.5 m with search_range 4 -> threshold 1 m measured from the last TAKEN keyframe
    odo2 = odo.copy(); odo2[:, 0] = np.arange(K) * 0.5
    i2, j2, _ = batch.delta_q_pairs(odo2, 4)
    assert [b for a, b in zip(i2, j2) if a == 0] == [3, 6]
    # search_range 6 -> 5 / 6 == 0: every keyframe at a different position passes
    i3, j3, _ = batch.delta_q_pairs(odo2, 6)
    assert [b for a, b in zip(i3, j3) if a == 0] == [1, 2, 3, 4, 5, 6]
    # sign unification of q_i only
    odo4 = odo.copy(); odo4[2, 3] = -1.0
    i4, j4, c4 = batch.delta_q_pairs(odo4, 3)
    for a, b, d in zip(i4, j4, c4):
        assert np.allclose(d, [-1 if b == 2 else 1, 0, 0, 0]), (a, b, d)
