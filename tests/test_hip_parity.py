"""GPU parity tests: the hand-written HIP path (through the C-ABI of libglio_hip.so) against the CPU
oracle on identical buffers.  Tolerances: H, g, cost per linearisation to 1e-10 relative (fp64, only
the summation order differs); poses per solve to 1e-4 m / 1e-5 rad (BASELINE.json north_star)."""
import numpy as np
import pytest

from glio_amd import ctypes_types as T
from glio_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from glio_amd import capi
    assert capi.device_count() >= 1, "no HIP device: the product path has no fallback"
    return capi


@pytest.fixture(scope="module")
def po():
    from oracle import pyoracle
    return pyoracle


def rel_err(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def rot_angle(qa, qb):
    d = synth.qmul(synth.qconj(qa), qb)
    return 2 * np.arctan2(np.linalg.norm(d[1:]), abs(d[0]))


def assert_pose_parity(sa, sb, tol_t=1e-4, tol_r=1e-5):
    dt = np.linalg.norm(sa.trans - sb.trans, axis=1).max()
    dr = max(rot_angle(sa.quat[i], sb.quat[i]) for i in range(sa.W))
    assert dt <= tol_t, f"translation parity {dt:.3e} m"
    assert dr <= tol_r, f"rotation parity {dr:.3e} rad"
    return dt, dr


CASES = [
    dict(name="lidar_only", use_imu=False, use_gnss=False, use_prior=False),
    dict(name="lidar_imu", use_imu=True, use_gnss=False, use_prior=False),
    dict(name="lidar_imu_prior", use_imu=True, use_gnss=False, use_prior=True),
    dict(name="all", use_imu=True, use_gnss=True, use_prior=True),
]


def _state_for(win, use_gnss):
    st = win.init.copy()
    if not use_gnss:
        st.n_ddt = 0
    return st


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_linearize_parity_small(hip, po, small_window, small_corr, case):
    win = small_window
    kw = {k: case[k] for k in ("use_imu", "use_gnss", "use_prior")}
    prob = po.Problem(win, small_corr, **kw)
    ctx = hip.Context(win.opts)
    ctx.load_window(win, small_corr, **kw)
    for st in (_state_for(win, case["use_gnss"]), None):
        if st is None:      # a second, different linearisation point
            st = _state_for(win, case["use_gnss"])
            st.trans += 0.05
            st.speed_bias[:, 3:] += 0.01
            st.rcv_ddt += 0.3
        Ho, go, co = prob.linearize(st)
        Hh, gh, ch = ctx.linearize(st)
        assert abs(ch - co) <= 1e-10 * abs(co)
        assert rel_err(gh, go) <= 1e-10
        assert rel_err(Hh, Ho) <= 1e-10
        assert np.abs(Hh - Hh.T).max() <= 1e-9 * np.abs(Hh).max()
    ctx.close()


@pytest.fixture(scope="module")
def k3_window():
    """1024-point scans (capacity a multiple of 64: the LDS-DMA form of K3 applies) with ragged per-keyframe counts."""
    from glio_amd import synth
    win = synth.make_window(W=3, pts_per_scan=1024, seed=synth.SEED_BASE + 31)
    corr = synth.analytic_correspondences(win)
    corr = [tuple(np.ascontiguousarray(a[: len(c[2]) - 37 * s - 5]) for a in c) for s, c in enumerate(corr)]
    return win, corr


@pytest.mark.parametrize("code", [1, 4, 12, 22, 24, 32, 33, 34])
def test_lidar_kernel_variants(hip, po, k3_window, code):
    """Every K3 code path (register batches, non-temporal, software pipelined, LDS-DMA rings of 2/3/4 chunks) against the
    oracle; the last chunk of a keyframe is partial and the widest geometry leaves wavefronts without work."""
    win, corr = k3_window
    kw = dict(use_imu=False, use_gnss=False, use_prior=False)
    prob = po.Problem(win, corr, **kw)
    ctx = hip.Context(win.opts)
    assert ctx.opts.max_points_per_scan % 64 == 0
    ctx.load_window(win, corr, **kw)
    st = _state_for(win, False)
    Ho, go, co = prob.linearize(st)
    for bpk in (1, 3, 8, 64):
        assert hip.load().glio_debug_set_k3(ctx._h, bpk, code) == 0
        Hh, gh, ch = ctx.linearize(st)
        assert abs(ch - co) <= 1e-10 * abs(co)
        assert rel_err(gh, go) <= 1e-10
        assert rel_err(Hh, Ho) <= 1e-10
    ctx.close()


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_linearize_launch_forms(hip, po, small_window, small_corr, mode):
    """K3 and the small factors as separate launches (0), in one heterogeneous launch (1), and with the idle workgroups
    behind the small-factor CUs (2, the default): the same normal equations, and the same solve, as the oracle's."""
    win = small_window
    prob = po.Problem(win, small_corr)
    ctx = hip.Context(win.opts)
    ctx.load_window(win, small_corr)
    assert hip.load().glio_debug_set_merged_linearize(ctx._h, mode) == 0
    st = _state_for(win, True)
    Ho, go, co = prob.linearize(st)
    Hh, gh, ch = ctx.linearize(st)
    assert abs(ch - co) <= 1e-10 * abs(co) and rel_err(gh, go) <= 1e-10 and rel_err(Hh, Ho) <= 1e-10
    so, summ_o = prob.solve(st)
    sh, summ_h = ctx.solve(st)
    assert summ_h.iterations == summ_o.iterations
    assert np.abs(sh.trans - so.trans).max() <= 1e-8 and np.abs(sh.quat - so.quat).max() <= 1e-9
    ctx.close()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_solve_parity_small(hip, po, small_window, small_corr, case):
    win = small_window
    kw = {k: case[k] for k in ("use_imu", "use_gnss", "use_prior")}
    prob = po.Problem(win, small_corr, **kw)
    ctx = hip.Context(win.opts)
    ctx.load_window(win, small_corr, **kw)
    st = _state_for(win, case["use_gnss"])
    so, summ_o = prob.solve(st)
    sh, summ_h = ctx.solve(st)
    assert summ_h.termination == summ_o.termination
    assert summ_h.iterations == summ_o.iterations and summ_h.successful_steps == summ_o.successful_steps
    assert abs(summ_h.final_cost - summ_o.final_cost) <= 1e-8 * abs(summ_o.final_cost)
    assert_pose_parity(sh, so)
    assert np.abs(sh.speed_bias - so.speed_bias).max() <= 1e-6
    if st.n_ddt:
        assert np.abs(sh.rcv_ddt - so.rcv_ddt).max() <= 1e-6
    assert np.allclose(np.linalg.norm(sh.quat, axis=1), 1.0, atol=1e-12)
    ctx.close()


def test_solve_parity_tight_convergence(hip, po, small_window, small_corr):
    """Both solvers run to tight convergence (function_tolerance 1e-12, 50 iterations): the converged
    optimum itself must agree, independent of where the default schedule happens to stop."""
    import copy
    win = copy.copy(small_window)
    opts = T.GlioOpts.from_buffer_copy(small_window.opts)
    opts.function_tolerance = 1e-12
    opts.parameter_tolerance = 1e-12
    opts.max_iterations = 50
    win.opts = opts
    prob = po.Problem(win, small_corr)
    ctx = hip.Context(opts)
    ctx.load_window(win, small_corr)
    so, _ = prob.solve(win.init)
    sh, _ = ctx.solve(win.init)
    assert_pose_parity(sh, so, 1e-6, 1e-7)
    ctx.close()


def test_evaluators_match_oracle(hip, po, small_window):
    win = small_window
    ctx = hip.Context(win.opts)
    rng = np.random.default_rng(11)
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    t = rng.normal(size=3)
    cp = (rng.normal(size=4) * 5).astype(np.float32)
    pl = np.r_[0.7 * np.array([0.6, 0.0, 0.8]), 1.4].astype(np.float32)
    r1, Jt1, Jq1 = ctx.eval_lidar_plane(cp, pl, 5.25, t, q)
    r0, Jt0, Jq0 = po.eval_lidar_plane(win.opts, cp, pl, 5.25, t, q)
    assert np.isclose(r1, r0, rtol=1e-13) and np.allclose(Jt1, Jt0, rtol=1e-13) and np.allclose(Jq1, Jq0, rtol=1e-12, atol=1e-13)
    pre = win.preints[1]
    ps = T.GlioPreint(); synth.fill_preint(ps, pre)
    st = win.init
    params = [st.trans[1], st.quat[1], st.speed_bias[1] + 0.01, st.trans[2], st.quat[2], st.speed_bias[2]]
    r1, J1 = ctx.eval_imu(pre, params)
    r0, J0 = po.eval_imu(win.opts, ps, params)
    assert np.allclose(r1, r0, rtol=1e-9, atol=1e-9 * np.abs(r0).max())
    for a, b in zip(J1, J0):
        assert np.allclose(a, b, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(b).max()))
    ctx.close()


def test_c1_shape_parity(hip, po):
    """BASELINE config C1: W=10, 16k surf points per keyframe, IMU + LiDAR (analytic correspondences)."""
    win = synth.make_window(W=10, pts_per_scan=16384, seed=synth.SEED_BASE + 11)
    corr = synth.analytic_correspondences(win)
    prob = po.Problem(win, corr)
    ctx = hip.Context(win.opts)
    ctx.load_window(win, corr)
    Ho, go, co = prob.linearize(win.init)
    Hh, gh, ch = ctx.linearize(win.init)
    assert abs(ch - co) <= 1e-10 * co and rel_err(gh, go) <= 1e-10 and rel_err(Hh, Ho) <= 1e-10
    so, summ_o = prob.solve(win.init)
    sh, summ_h = ctx.solve(win.init)
    assert summ_h.iterations == summ_o.iterations
    assert_pose_parity(sh, so)
    ctx.close()


def test_c2_shape_parity_full_size(hip, po):
    """BASELINE config C2 (the bench workload): W=20, 64k pts/keyframe, LiDAR+IMU+GNSS+prior.
    At this size the oracle still linearises in ~0.2 s, so parity is checked directly, plus the
    size-independent properties: H symmetric PSD, cost additive over keyframes, counts preserved."""
    win = synth.make_window(W=20, pts_per_scan=65536, with_gnss=True, with_prior=True, seed=synth.SEED_BASE + 12)
    corr = synth.analytic_correspondences(win)
    prob = po.Problem(win, corr)
    ctx = hip.Context(win.opts)
    ctx.load_window(win, corr)
    Ho, go, co = prob.linearize(win.init)
    Hh, gh, ch = ctx.linearize(win.init)
    assert abs(ch - co) <= 1e-10 * co and rel_err(gh, go) <= 1e-10 and rel_err(Hh, Ho) <= 1e-10
    assert np.abs(Hh - Hh.T).max() <= 1e-9 * np.abs(Hh).max()
    assert np.linalg.eigvalsh(0.5 * (Hh + Hh.T)).min() > -1e-6 * np.abs(Hh).max()
    so, summ_o = prob.solve(win.init)
    sh, summ_h = ctx.solve(win.init)
    assert summ_h.iterations == summ_o.iterations and summ_h.termination == summ_o.termination
    assert summ_h.n_lidar_residuals == sum(len(c[2]) for c in corr)
    assert_pose_parity(sh, so)
    # determinism: the fixed-order reductions make two runs bit-identical
    sh2, _ = ctx.solve(win.init)
    assert np.array_equal(sh.trans, sh2.trans) and np.array_equal(sh.quat, sh2.quat)
    ctx.close()


def test_empty_and_ragged_slots(hip, po, small_window, small_corr):
    """Edge cases: a keyframe with zero correspondences, one with a single one, one ragged."""
    win = small_window
    corr = [tuple(a.copy() for a in c) for c in small_corr]
    corr[1] = (corr[1][0][:0], corr[1][1][:0], corr[1][2][:0])
    corr[2] = (corr[2][0][:1], corr[2][1][:1], corr[2][2][:1])
    corr[3] = (corr[3][0][:257], corr[3][1][:257], corr[3][2][:257])
    prob = po.Problem(win, corr)
    ctx = hip.Context(win.opts)
    ctx.load_window(win, corr)
    Ho, go, co = prob.linearize(win.init)
    Hh, gh, ch = ctx.linearize(win.init)
    assert abs(ch - co) <= 1e-10 * co and rel_err(gh, go) <= 1e-10 and rel_err(Hh, Ho) <= 1e-10
    so, _ = prob.solve(win.init)
    sh, _ = ctx.solve(win.init)
    assert_pose_parity(sh, so)
    ctx.close()


def test_capacity_and_argument_errors(hip, small_window):
    win = small_window
    ctx = hip.Context(win.opts)
    cap = win.opts.max_points_per_scan
    with pytest.raises(hip.GlioError):
        ctx.set_correspondences(0, np.zeros((cap + 1, 4), np.float32), np.zeros((cap + 1, 4), np.float32), np.zeros(cap + 1))
    with pytest.raises(hip.GlioError):
        ctx.solve(win.init)           # no factors yet
    ctx.close()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_structured_solver_equals_dense(hip, small_window, small_corr, case):
    """The arrow factorisation (epochs | speed-bias chain | dense poses) and the dense Cholesky solve the same
    linear systems: identical iteration history, solutions equal to rounding."""
    win = small_window
    kw = {k: case[k] for k in ("use_imu", "use_gnss", "use_prior")}
    res = []
    for mode in (0, 1):
        ctx = hip.Context(win.opts)
        assert hip.load().glio_debug_set_solver(ctx._h, mode) == 0
        ctx.load_window(win, small_corr, **kw)
        res.append(ctx.solve(_state_for(win, case["use_gnss"])))
        ctx.close()
    (sd, md), (sa, ma) = res
    assert ma.iterations == md.iterations and ma.successful_steps == md.successful_steps and ma.termination == md.termination
    assert abs(ma.final_cost - md.final_cost) <= 1e-12 * abs(md.final_cost)
    assert np.abs(sa.trans - sd.trans).max() <= 1e-10 and np.abs(sa.quat - sd.quat).max() <= 1e-11
    assert np.abs(sa.speed_bias - sd.speed_bias).max() <= 1e-9
    if sd.n_ddt:
        assert np.abs(sa.rcv_ddt - sd.rcv_ddt).max() <= 1e-9


def test_gnss_prior_batch_evaluators_match_oracle(hip, po, small_window):
    """Every remaining factor type through the Evaluate() ABI on the GPU vs the oracle's evaluators."""
    win = small_window
    ctx = hip.Context(win.opts)
    st = win.init
    fr = win.frame
    yaw, anc = fr.yaw_enu_local, np.array(fr.anc_ecef)
    # DD pseudorange: ranges ~2e7 m, whitened residuals O(1..10)
    for f in win.dd[:3]:
        Pi, Pj = st.trans[f.slot_i], st.trans[f.slot_j]
        ro, Jo = po.eval_dd_psr(f, Pi, Pj, yaw, anc)
        rh, Jh = ctx.eval_dd_psr(f, Pi, Pj, yaw, anc)
        assert np.abs(rh - ro).max() <= 1e-7 * max(1.0, np.abs(ro).max())
        for a, b in zip(Jh, Jo):
            assert np.abs(a - b).max() <= 1e-9
        assert np.all(rh[f.n_sat - 1:] == 0)
    # Doppler
    ddt = np.linspace(0.1, 0.5, max(st.n_ddt, 1))
    for f in win.dop[:4]:
        args = (st.trans[f.slot_i], st.speed_bias[f.slot_i], st.trans[f.slot_j], st.speed_bias[f.slot_j], ddt, yaw, anc)
        ro, Jo = po.eval_doppler(f, *args)
        rh, Jh = ctx.eval_doppler(f, *args)
        assert abs(rh - ro) <= 1e-8 * max(1.0, abs(ro))
        for a, b in zip(Jh, Jo):
            assert np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(b).max())
    # marginalization prior
    pr = win.prior
    params = []
    for k in range(len(pr["blk_slot"])):
        s, kind = pr["blk_slot"][k], pr["blk_kind"][k]
        params.append([st.trans[s], st.quat[s], st.speed_bias[s]][kind])
    ro, Jo = po.eval_marg(pr, params)
    rh, Jh = ctx.eval_marginalization(pr, params)
    assert rel_err(rh, ro) <= 1e-12
    for a, b in zip(Jh, Jo):
        assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max())
    # binary plane
    rng = np.random.default_rng(3)
    for _ in range(3):
        cp = rng.normal(0, 5, 4).astype(np.float32); pnc = np.r_[synth.rotvec_q(rng.normal(0, 1, 3))[1:], rng.normal(0, 5, 3)]
        pnc[:3] /= np.linalg.norm(pnc[:3])
        t1, t2 = rng.normal(0, 3, 3), rng.normal(0, 3, 3)
        q1, q2 = synth.rotvec_q(rng.normal(0, 0.5, 3)), synth.rotvec_q(rng.normal(0, 0.5, 3))
        ro, Jo = po.eval_binary_plane(cp, pnc, 2.1, t1, q1, t2, q2)
        rh, Jh = ctx.eval_binary_plane(cp, pnc, 2.1, t1, q1, t2, q2)
        assert abs(rh - ro) <= 1e-12 * max(1.0, abs(ro))
        for a, b in zip(Jh, Jo):
            assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max())
    ctx.close()


@pytest.fixture(scope="module")
def steady_window(po):
    """A window whose prior IS a marginalization output (block diagonal by keyframe): keyframes 1..W of a
    W+1 stream, prior from marginalizing keyframe 0 of the window before."""
    W = 5
    long = synth.make_window(W=W + 1, pts_per_scan=600, with_gnss=True, seed=synth.SEED_BASE + 91)
    first = synth.sub_window(long, 0, W)
    corr0 = synth.analytic_correspondences(first)
    prob0 = po.Problem(first, corr0, use_gnss=False, use_prior=False)
    st0 = first.init.copy(); st0.n_ddt = 0
    sol0, _ = prob0.solve(st0)
    prior = prob0.marginalize(sol0)
    win = synth.sub_window(long, 1, W)
    win.prior = prior
    return win, synth.analytic_correspondences(win)


@pytest.mark.parametrize("use_gnss", [False, True], ids=["imu_prior", "imu_gnss_prior"])
def test_keyframe_chain_solver_equals_dense(hip, po, steady_window, use_gnss):
    win, corr = steady_window
    res = []
    for mode in (0, 1):
        ctx = hip.Context(win.opts)
        hip.load().glio_debug_set_solver(ctx._h, mode)
        ctx.load_window(win, corr, use_gnss=use_gnss)
        st = _state_for(win, use_gnss)
        res.append(ctx.solve(st) + (hip.load().glio_debug_solver_path(ctx._h),))
        ctx.close()
    (sd, md, pd_), (sc, mc, pc) = res
    assert pd_ == 0 and pc == 2, "the marginalization prior must select the keyframe-chain factorisation"
    assert mc.iterations == md.iterations and mc.successful_steps == md.successful_steps and mc.termination == md.termination
    assert abs(mc.final_cost - md.final_cost) <= 1e-12 * abs(md.final_cost)
    assert np.abs(sc.trans - sd.trans).max() <= 1e-10 and np.abs(sc.quat - sd.quat).max() <= 1e-11
    assert np.abs(sc.speed_bias - sd.speed_bias).max() <= 1e-9
    so, mo = po.Problem(win, corr, use_gnss=use_gnss).solve(_state_for(win, use_gnss))
    assert mo.iterations == mc.iterations
    assert_pose_parity(sc, so)
    # the chain kernel's breakdown path: every step reports a non-positive pivot, k_tr_finish rebuilds the scaled matrix
    # and t = H u itself and factors densely -- the iterates must be the dense solver's
    ctx = hip.Context(win.opts)
    hip.load().glio_debug_set_solver(ctx._h, 2)
    ctx.load_window(win, corr, use_gnss=use_gnss)
    sf, mf = ctx.solve(_state_for(win, use_gnss))
    assert hip.load().glio_debug_solver_path(ctx._h) == 2
    ctx.close()
    assert mf.iterations == md.iterations and mf.termination == md.termination
    assert np.abs(sf.trans - sd.trans).max() <= 1e-10 and np.abs(sf.quat - sd.quat).max() <= 1e-11


@pytest.mark.parametrize("use_gnss", [True, False], ids=["gnss", "no_gnss"])
def test_chain_step_is_bit_identical_to_the_assembled_sequence(hip, steady_window, use_gnss):
    """k_chain_step gathers H, g and the cost from the factor blocks with the sums k_assemble would have formed, in the same
    order, so the one-launch step must reproduce the legacy sequence [k_assemble, k_chain_solve, k_tr_finish] bit for bit:
    states, iteration history, costs -- here over repeated solves and after a re-linearisation through glio_linearize."""
    win, corr = steady_window
    import copy
    out = []
    lib = hip.load()
    # a second set of options under which most steps are REJECTED (the relative decrease of this nearly quadratic problem sits at 1): the
    # rejected-step branches of the state machine, the stored Gauss-Newton / Cauchy vectors reused with the halved radius
    opts2 = copy.copy(win.opts)
    opts2.min_relative_decrease = 1.05 if use_gnss else 1.0
    opts2.max_iterations = 30
    far = _state_for(win, use_gnss)
    far.trans = far.trans + np.random.default_rng(5).normal(0, 0.4, far.trans.shape)
    far.speed_bias = far.speed_bias + np.random.default_rng(6).normal(0, 0.2, far.speed_bias.shape)
    key = lambda s, m: (s.trans.tobytes(), s.quat.tobytes(), s.speed_bias.tobytes(), s.rcv_ddt[:s.n_ddt].tobytes(), m.iterations, m.successful_steps,
                        m.termination, m.initial_cost, m.final_cost, m.final_radius, m.gradient_max_norm)
    # (solver mode, GLIO_CHAIN_FAST mask): the legacy sequence, then k_chain_step with its generic bodies (0), with the tail (1), the front (2)
    # and both (3, the default) running from the LDS copies instead of the global work vectors
    try:
        for mode, fast in ((3, 3), (1, 0), (1, 1), (1, 2), (1, 3)):
            lib.glio_debug_chain_fast(fast)
            ctx = hip.Context(win.opts)
            lib.glio_debug_set_solver(ctx._h, mode)
            ctx.load_window(win, corr, use_gnss=use_gnss)
            st = _state_for(win, use_gnss)
            runs = []
            for k in range(3):
                runs.append(key(*ctx.solve(st)))
                if k == 0:
                    ctx.linearize(st)            # the dense assembly in between must not disturb the block-fed step
            assert lib.glio_debug_solver_path(ctx._h) == 2
            ctx.close()
            ctx = hip.Context(opts2)
            lib.glio_debug_set_solver(ctx._h, mode)
            ctx.load_window(win, corr, use_gnss=use_gnss)
            s, m = ctx.solve(far)
            assert lib.glio_debug_solver_path(ctx._h) == 2
            assert m.iterations - m.successful_steps >= 10, "this run is meant to consist mostly of rejected steps"
            runs.append(key(s, m))
            ctx.close()
            out.append(runs)
    finally:
        lib.glio_debug_chain_fast(3)
    assert out[0][0] == out[0][1] == out[0][2]
    for k in range(1, len(out)):
        assert out[k] == out[0], k


def _steady(hip, W, pts, seed):
    """keyframes 1..W of a W+1 stream with the prior the DEVICE marginalization of keyframe 0 leaves (block diagonal by keyframe)"""
    stream = synth.make_window(W=W + 1, pts_per_scan=pts, with_gnss=True, seed=synth.SEED_BASE + seed)
    first = synth.sub_window(stream, 0, W)
    c0 = hip.Context(first.opts)
    c0.load_window(first, synth.analytic_correspondences(first))
    s0, _ = c0.solve(first.init)
    prior = c0.marginalize(s0)
    c0.close()
    win = synth.sub_window(stream, 1, W)
    win.prior = prior
    return win, synth.analytic_correspondences(win)


@pytest.mark.parametrize("W,use_gnss", [(12, True), (13, False), (16, True), (17, True), (20, True), (20, False), (21, True), (22, True), (24, False), (28, True), (41, False), (44, True)])
def test_four_front_elimination(hip, po, W, use_gnss):
    """k_chain_step on windows of 12 keyframes and more: the middle keyframe as separator and four elimination fronts (chain_f4_split) instead of
    two.  A different elimination order of the same positive definite system: the iterates must agree with the two-front order and with the dense
    factorisation to rounding, iteration for iteration; the breakdown path (every step reports a bad pivot, the same workgroup rebuilds the system
    densely from the factor blocks) must still find its tables intact behind the four-front panels.  W = 24 without GNSS / W = 21 with its epochs
    are where the panels no longer fit beside the LDS mirrors -- whatever layout the launch picks, the numbers must not care.  From about 25
    keyframes on the blocks themselves no longer fit the LDS: the sequence [band-only k_assemble, k_chain_solve<true>, k_tr_finish] keeps them in
    global memory (same elimination, two or four fronts); its breakdown path reads the WHOLE dense matrix, i.e. relies on the zeros the host put
    outside the band."""
    lib = hip.load()
    win, corr = _steady(hip, W, 300, 300 + W)
    far = _state_for(win, use_gnss)
    far.trans = far.trans + np.random.default_rng(W).normal(0, 0.05, far.trans.shape)
    res = {}
    try:
        for name, mode, fronts in (("dense", 0, 4), ("two", 1, 2), ("four", 1, 4), ("four_breakdown", 2, 4)):
            lib.glio_debug_chain_fronts(fronts)
            ctx = hip.Context(win.opts)
            lib.glio_debug_set_solver(ctx._h, mode)
            ctx.load_window(win, corr, use_gnss=use_gnss)
            res[name] = ctx.solve(far) + (lib.glio_debug_solver_path(ctx._h),)
            if name in ("two", "four"):
                used = lib.glio_debug_chain_fronts_used(ctx._h)
                assert used == (2 if name == "two" else 4) or (name == "four" and (W, use_gnss) in ((21, True), (22, True), (24, False))), (name, used)
            ctx.close()
    finally:
        lib.glio_debug_chain_fronts(4)
    sd, md, pd_ = res["dense"]
    assert pd_ == 0 and res["two"][2] == 2 and res["four"][2] == 2
    for name in ("two", "four", "four_breakdown"):
        s, m, _ = res[name]
        assert m.iterations == md.iterations and m.successful_steps == md.successful_steps and m.termination == md.termination, name
        assert abs(m.final_cost - md.final_cost) <= 1e-11 * abs(md.final_cost), name
        assert np.abs(s.trans - sd.trans).max() <= 1e-9 and np.abs(s.quat - sd.quat).max() <= 1e-10 and np.abs(s.speed_bias - sd.speed_bias).max() <= 1e-8, name
    s2, s4 = res["two"][0], res["four"][0]
    assert np.abs(s4.trans - s2.trans).max() <= 1e-11 and np.abs(s4.quat - s2.quat).max() <= 1e-12 and np.abs(s4.speed_bias - s2.speed_bias).max() <= 1e-10
    if W in (12, 20):
        so, mo = po.Problem(win, corr, use_gnss=use_gnss).solve(_state_for_copy(far))
        assert mo.iterations == res["four"][1].iterations
        assert_pose_parity(s4, so, tol_t=1e-7, tol_r=1e-8)


def _state_for_copy(st):
    return st.copy()


@pytest.mark.parametrize("n", [37, 160, 376, 399, 400, 414, 420])
def test_blocked_cholesky_sizes(hip, n):
    """The in-kernel blocked Cholesky + back substitution of the dense path (chol_left_looking, back_substitute) against numpy on random SPD
    systems, on both sides of n = 400 (where a second round of the panel factorisation sets in)."""
    import ctypes as C
    win = synth.make_window(W=28, pts_per_scan=64, with_gnss=False)
    ctx = hip.Context(win.opts)
    rng = np.random.default_rng(n)
    B = rng.normal(0, 1, (n, n))
    A = B @ B.T / n + np.eye(n)
    b = rng.normal(0, 1, n)
    L = np.tril(A).copy()
    x = np.zeros(n)
    rc = hip.load().glio_debug_chol_solve(ctx._h, n, L.ctypes.data_as(C.POINTER(C.c_double)), b.ctypes.data_as(C.POINTER(C.c_double)), x.ctypes.data_as(C.POINTER(C.c_double)))
    ctx.close()
    assert rc == 0
    xs = np.linalg.solve(A, b)
    assert np.linalg.norm(x - xs) <= 1e-11 * np.linalg.norm(xs)


def _steady_window_22(po):
    W = 22
    long = synth.make_window(W=W + 1, pts_per_scan=200, with_gnss=True, seed=synth.SEED_BASE + 93)
    first = synth.sub_window(long, 0, W)
    prob0 = po.Problem(first, synth.analytic_correspondences(first), use_gnss=False, use_prior=False)
    st0 = first.init.copy(); st0.n_ddt = 0
    sol0, _ = prob0.solve(st0)
    win = synth.sub_window(long, 1, W)
    win.prior = prob0.marginalize(sol0)
    return win, synth.analytic_correspondences(win)


def test_chain_step_on_a_window_too_large_for_its_lds_mirrors(hip, po):
    """W = 22 with GNSS (n = 414): the keyframe blocks leave no room for the LDS copies the fast front / tail of k_chain_step work on, so the
    host launches the kernel with its generic bodies (smaller carve) -- it must still be the chain path and agree with the oracle."""
    win, corr = _steady_window_22(po)
    so, mo = po.Problem(win, corr).solve(win.init.copy())
    ctx = hip.Context(win.opts)
    ctx.load_window(win, corr)
    sc, mc = ctx.solve(win.init.copy())
    assert hip.load().glio_debug_solver_path(ctx._h) == 2
    ctx.close()
    assert mc.iterations == mo.iterations and mc.termination == mo.termination
    assert abs(mc.final_cost - mo.final_cost) <= 1e-9 * abs(mo.final_cost)
    assert_pose_parity(sc, so, tol_t=1e-7, tol_r=1e-8)


def test_dense_fallback_at_n_414(hip, po):
    """The dense fallback (solver mode 0) on a window with more than 384 rows under the first panel of its blocked Cholesky: until the end of
    round 3 the wavefronts' second round re-read a diagonal block that wavefront 0 had already overwritten with its factor (n >= 400)."""
    win, corr = _steady_window_22(po)
    so, mo = po.Problem(win, corr).solve(win.init.copy())
    ctx = hip.Context(win.opts)
    hip.load().glio_debug_set_solver(ctx._h, 0)
    ctx.load_window(win, corr)
    sd, md = ctx.solve(win.init.copy())
    assert hip.load().glio_debug_solver_path(ctx._h) == 0
    ctx.close()
    assert md.iterations == mo.iterations and md.termination == mo.termination
    assert_pose_parity(sd, so)


def test_dense_gnss_pairs_cross_the_chunk_boundaries(hip, po):
    """A keyframe pair with MORE factors than the GNSS role takes side by side: 25 epochs per pair (gnss_epoch_dt 0.016 s) = 50 DD
    factors (7 chunks of 8) and 500 Doppler rows (4 chunks of 128, epochs straddling the chunk boundaries: the carried partial sums
    of the per-epoch lanes).  Linearisation and solve against the oracle; the chain step (no dense H) against the dense path."""
    from glio_amd import synth
    win = synth.make_window(W=4, pts_per_scan=512, with_gnss=True, seed=synth.SEED_BASE + 77, gnss_epoch_dt=0.016)
    corr = synth.analytic_correspondences(win)
    assert len(win.dd) >= 3 * 40 and len(win.dop) >= 3 * 400
    prob = po.Problem(win, corr, use_prior=False)
    ctx = hip.Context(win.opts)
    ctx.load_window(win, corr, use_prior=False)
    st = _state_for(win, True)
    Ho, go, co = prob.linearize(st)
    Hh, gh, ch = ctx.linearize(st)
    assert abs(ch - co) <= 1e-10 * abs(co) and rel_err(gh, go) <= 1e-10 and rel_err(Hh, Ho) <= 1e-10
    so, mo = prob.solve(st)
    sh, mh = ctx.solve(st)
    assert mh.iterations == mo.iterations and mh.termination == mo.termination
    assert np.abs(sh.trans - so.trans).max() < 1e-8 and np.abs(sh.rcv_ddt[:sh.n_ddt] - so.rcv_ddt[:so.n_ddt]).max() < 1e-6
    assert hip.load().glio_debug_solver_path(ctx._h) == 2
    ctx.close()
