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
    sol, summ, counts = drv.step(None, [None] * W, [])
    assert counts == [7] * W and sol.quat[1, 0] == 1.0                            # unified sign
    names = [c if isinstance(c, str) else c[0] for c in be.calls]
    assert names == ["set_map", "associate0", "associate1", "associate2", "set_imu", "set_prior", "set_gnss", "solve", "marginalize"]
    assert be.calls[5][1] is None                                                 # first window: no prior
    drv.slide(np.ones(3), np.array([1.0, 0, 0, 0]), np.zeros(9))
    assert drv.first == 1 and np.array_equal(drv.state.trans[-1], np.ones(3))
    be.calls.clear()
    drv.step(None, [None] * W, [])
    assert [c for c in be.calls if not isinstance(c, str)][0][1] == {"n": 1}      # the marginalization result is the next prior
