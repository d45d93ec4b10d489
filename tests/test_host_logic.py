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


def test_batch_gnss_epoch_selection_cpp_equals_python(tmp_path):
    """glio::selectBatchGnssEpochs / glio::ddGroup (the host rules of optimizeBatchWithLandMark, Estimator.cpp:3086-3126 with getGlobalLowerUpperIdx :1635-1663,
    and prepare<SYS>DDPsrData :1702-1860) against their Python twins on random streams, plus the rules themselves on hand-made cases: bracketing by the
    closest keyframe strictly before / after, epochs outside the keyframe span or at a keyframe time without a neighbour dropped, the 1 m spacing gate
    measured from the last ACCEPTED right keyframe (starting at the origin), the master satellite chosen with the signed-elevation quirk."""
    import os
    import subprocess
    from glio_amd import batch
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "glio_amd", "host")
    exe = str(tmp_path / "gsel")
    subprocess.check_call(["g++", "-std=c++14", "-O1", os.path.join(here, "host_gnss_select_test.cpp"), "-I" + os.path.join(here, "..", "..", "include"), "-o", exe])

    def run(obs, kt, first_idx, n_poses, tr, sats=None):
        parts = [f"{len(obs)} {len(kt)} {first_idx} {n_poses} {len(tr)}", " ".join(repr(float(x)) for x in obs), " ".join(repr(float(x)) for x in kt),
                 " ".join(repr(float(x)) for x in np.asarray(tr).ravel())]
        if sats is not None:
            up, psr, ele, rp = sats
            parts += [f"{len(up)} {len(rp)}", " ".join(str(int(x)) for x in up), " ".join(repr(float(x)) for x in psr), " ".join(repr(float(x)) for x in ele),
                      " ".join(str(int(x)) for x in rp)]
        out = subprocess.run([exe], input="\n".join(parts) + "\n", capture_output=True, text=True, check=True).stdout.splitlines()
        ep = [(int(w[1]), int(w[2]), int(w[3]), float(w[4])) for w in (ln.split() for ln in out) if w[0] == "epoch"]
        gr = {int(w[1]): (int(w[3]), [tuple(int(v) for v in t.split(":")) for t in w[5:]]) for w in (ln.split() for ln in out) if w[0] == "group"}
        return ep, gr

    rng = np.random.default_rng(11)
    for trial in range(20):
        K = int(rng.integers(5, 40))
        kt = np.cumsum(rng.uniform(0.2, 0.6, K)) + 100.0
        tr = np.cumsum(rng.normal(0, 1.2, (K, 3)), axis=0) + (0.0 if trial % 3 else 50.0)
        obs = np.sort(rng.uniform(kt[0] - 1.0, kt[-1] + 1.0, int(rng.integers(1, 60))))
        if trial % 4 == 0:
            obs[::5] = kt[rng.integers(0, K, len(obs[::5]))]          # epochs exactly at keyframe times
        first_idx, n_poses = int(rng.integers(1, 3)), K + int(rng.integers(0, 2))
        got, _ = run(obs, kt, first_idx, min(n_poses, K + 1), tr)
        want = batch.select_batch_gnss_epochs(obs, kt, first_idx, min(n_poses, K + 1), tr)
        assert len(got) == len(want) and all(g[:3] == w[:3] and g[3] == w[3] for g, w in zip(got, want)), trial
        for (e, lk, rk, ratio) in want:
            assert lk < rk and kt[lk] < obs[e] < kt[rk] and 0.0 < ratio < 1.0
    # hand-made: keyframes 1 s apart moving 0.6 m per keyframe along x from x = 10: epochs at 0.5 (before the first), 1.25, 1.5, 2.25, 3.0 (at a keyframe), 9.9 (after the last)
    kt = np.arange(1.0, 6.0); tr = np.c_[10.0 + 0.6 * np.arange(5), np.zeros(5), np.zeros(5)]
    got, _ = run([0.5, 1.25, 1.5, 2.25, 3.0, 4.5, 9.9], kt, 1, 5, tr)
    # pose indices are 1-based and the search runs over [1, 5): keyframe_time[0..3]; 1.25 -> (1, 2) right keyframe x = 10.6: accepted (10.6 m from the origin);
    # 1.5 -> the same right keyframe: 0 m from the last accepted: dropped; 2.25 -> right keyframe x = 11.2, 0.6 m: dropped; 3.0 -> (2, 4): strictly before / after,
    # right keyframe x = 11.8, 1.2 m: accepted with ratio (4 - 3) / (4 - 2); 4.5 -> no pose after it inside [1, 5): dropped
    assert got == [(1, 0, 1, 0.75), (4, 1, 3, 0.5)]
    # arguments that would index outside the arrays are refused by both twins: first_idx = 0 (pose indices are 1-based), more poses than keyframe times / positions
    import pytest
    for bad in ((0, 5), (1, 7), (6, 5)):
        exe_out = subprocess.run([exe], input=f"1 5 {bad[0]} {bad[1]} 5\n2.5\n" + " ".join(map(str, kt)) + "\n" + " ".join(repr(float(x)) for x in tr.ravel()) + "\n",
                                 capture_output=True, text=True, check=True).stdout
        assert exe_out.startswith("refused"), (bad, exe_out)
        with pytest.raises(ValueError):
            batch.select_batch_gnss_epochs([2.5], kt, bad[0], bad[1], tr)
    # double-difference groups: GPS 5, 12, 30 (+ 84), BeiDou 90, GLONASS 40, Galileo 60; PRN 12 has no station observation, PRN 30 a bad pseudorange
    up, psr, ele = [5, 12, 30, 84, 90, 40, 60, 7], [2e7, 2e7, 500.0, 2e7, 2e7, 2e7, 2e7, 2e7], [30.0, 80.0, 70.0, -45.0, 10.0, 20.0, 15.0, 40.0]
    rp = [7, 84, 5, 90, 60, 40, 30]
    _, gr = run([1.5], [1.0, 2.0], 1, 2, [[5, 0, 0], [9, 0, 0]], sats=(up, psr, ele, rp))
    for sysid in range(4):
        u, r, m = batch.dd_group(sysid, up, psr, ele, rp)
        assert gr[sysid] == (m, list(zip(u, r)))
    # GPS pairs in rover order: (5 -> station 2), (84 -> 1), (7 -> 0); master: 30 deg sets max 30; |-45| > 30 sets max to -45 (signed!); |40| > -45 -> the LAST pair
    assert gr[0] == (2, [(0, 2), (3, 1), (7, 0)]) and gr[1] == (0, [(4, 3)]) and gr[2] == (0, [(5, 5)]) and gr[3] == (0, [(6, 4)])
