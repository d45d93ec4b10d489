"""BASELINE config C5: the LiDAR plane linearisation with fp32 Jacobians and the J^T J / J^T r contraction on the matrix
core (opts.lidar_precision = GLIO_LIDAR_F32_MFMA; LidarPlaneNormFactor, GLIO/include/factors/LidarKeyframeFactor.h:87-103
+ Huber + QuaternionParameterization), against the fp64 oracle.

Stated fp32 tolerance.  The device forms each weighted row a = sqrt(rho') [J, r] in float (2-3 roundings per entry, 6e-8
each), multiplies in float and adds 16 products per accumulator in float before the sum continues in double, so an entry
of H = sum a_i a_j is off by at most ~1.5e-6 sum |a_i a_j| <= 1.5e-6 sqrt(H_ii H_jj), typically a few 1e-7.  The tests
bound  |dH_ij| <= 2e-6 sqrt(H_ii H_jj)  and  |dg_i| <= 2e-6 sqrt(H_ii 2 cost)  (Cauchy-Schwarz scale of the sums) and the
cost, whose terms are formed in double from a float-rounded score, to 1e-6 relative.  Poses after a full solve must agree
with the fp64 oracle within the north-star gate, 1e-4 m / 1e-5 rad."""
import numpy as np
import pytest

from glio_amd import synth

pytestmark = pytest.mark.gpu
F32_TOL = 2e-6


@pytest.fixture(scope="module")
def hip():
    from glio_amd import capi
    assert capi.device_count() >= 1, "no HIP device: the product path has no fallback"
    return capi


@pytest.fixture(scope="module")
def po():
    from oracle import pyoracle
    return pyoracle


def _f32_opts(win):
    from glio_amd import ctypes_types as T
    o = T.GlioOpts.from_buffer_copy(win.opts)
    o.lidar_precision = 1
    return o


def _check_lin(Hh, gh, ch, Ho, go, co, W):
    assert abs(ch - co) <= 1e-6 * abs(co), (ch, co)
    for s in range(W):
        o = 15 * s
        d = np.sqrt(np.diag(Ho)[o:o + 6])
        scale = np.outer(d, d)
        assert np.all(np.abs(Hh[o:o + 6, o:o + 6] - Ho[o:o + 6, o:o + 6]) <= F32_TOL * scale), f"H block of keyframe {s}"
        assert np.all(np.abs(gh[o:o + 6] - go[o:o + 6]) <= F32_TOL * d * np.sqrt(2 * co)), f"g of keyframe {s}"
    assert np.abs(Hh - Hh.T).max() == 0.0 or np.abs(Hh - Hh.T).max() <= 1e-12 * np.abs(Hh).max()


def test_f32_mfma_linearize_ragged(hip, po):
    """LiDAR factors only, ragged counts (partial last chunk, wavefronts without work at the widest geometry)."""
    win = synth.make_window(W=3, pts_per_scan=1024, seed=synth.SEED_BASE + 31)
    corr = synth.analytic_correspondences(win)
    corr = [tuple(np.ascontiguousarray(a[: len(c[2]) - 37 * s - 5]) for a in c) for s, c in enumerate(corr)]
    kw = dict(use_imu=False, use_gnss=False, use_prior=False)
    prob = po.Problem(win, corr, **kw)
    st = win.init.copy()
    Ho, go, co = prob.linearize(st)
    ctx = hip.Context(_f32_opts(win))
    ctx.load_window(win, corr, **kw)
    for bpk in (1, 3, 8, 64):
        assert hip.load().glio_debug_set_k3(ctx._h, bpk, 22) == 0
        Hh, gh, ch = ctx.linearize(st)
        _check_lin(Hh, gh, ch, Ho, go, co, win.W)
    # outliers: a state far enough from the truth that many residuals sit on the linear branch of the Huber loss
    st2 = win.init.copy()
    st2.trans += 0.6
    Ho, go, co = prob.linearize(st2)
    Hh, gh, ch = ctx.linearize(st2)
    _check_lin(Hh, gh, ch, Ho, go, co, win.W)
    # two runs are bit-identical (fixed-order reductions)
    Hh2, gh2, ch2 = ctx.linearize(st2)
    assert np.array_equal(Hh, Hh2) and np.array_equal(gh, gh2) and ch == ch2
    ctx.close()


def test_f32_mfma_in_the_merged_launch_and_after_reassociation(hip, po, small_window, small_corr):
    """All factor types (the f32 K3 workgroups run inside k_linearize_all beside the small factors), then new
    correspondences: the packed 32 B/residual points follow."""
    win = small_window
    prob = po.Problem(win, small_corr)
    ctx = hip.Context(_f32_opts(win))
    ctx.load_window(win, small_corr)
    st = win.init.copy()
    Ho, go, co = prob.linearize(st)
    Hh, gh, ch = ctx.linearize(st)
    assert np.linalg.norm(Hh - Ho) <= 1e-6 * np.linalg.norm(Ho)
    assert np.linalg.norm(gh - go) <= 1e-5 * np.linalg.norm(go) + 1e-6 * np.sqrt(np.diag(Ho).sum() * 2 * co)
    assert abs(ch - co) <= 1e-6 * co
    sh, summ_h = ctx.solve(st)
    so, summ_o = prob.solve(st)
    assert np.linalg.norm(sh.trans - so.trans, axis=1).max() <= 1e-4
    half = [tuple(np.ascontiguousarray(a[: len(c[2]) // 2]) for a in c) for c in small_corr]
    for s in range(win.W):
        ctx.set_correspondences(s, *half[s])
    H2, g2, c2 = ctx.linearize(st)
    Ho2, go2, co2 = po.Problem(win, half).linearize(st)
    assert abs(c2 - co2) <= 1e-6 * co2 and np.linalg.norm(H2 - Ho2) <= 1e-6 * np.linalg.norm(Ho2)
    ctx.close()


@pytest.fixture(scope="module")
def c5_window():
    win = synth.make_window(W=50, pts_per_scan=262144, with_gnss=True, with_prior=False, seed=synth.SEED_BASE + 50, gnss_epoch_dt=0.4)
    corr = synth.analytic_correspondences(win)
    return win, corr


_ORACLE_C5 = {}


def _oracle_c5(po, win, corr):
    """the fp64 oracle's linearisation and solve of the C5 window, computed once for both tests (~20 s of CPU)"""
    if not _ORACLE_C5:
        prob = po.Problem(win, corr)
        _ORACLE_C5["lin"] = prob.linearize(win.init.copy())
        _ORACLE_C5["solve"] = prob.solve(win.init.copy())
    return _ORACLE_C5["lin"], _ORACLE_C5["solve"]


def test_c5_full_size_solve_f32_vs_fp64_oracle(hip, po, c5_window):
    """W = 50 x 262 144 points per keyframe (13.1 M residuals), LiDAR + IMU + GNSS: one linearisation and one full
    solve of the f32/MFMA path against the fp64 oracle (about 2 s per oracle linearisation)."""
    win, corr = c5_window
    n_res = sum(len(c[2]) for c in corr)
    assert n_res > 12_000_000
    (Ho, go, co), (so, summ_o) = _oracle_c5(po, win, corr)
    st = win.init.copy()
    ctx = hip.Context(_f32_opts(win))
    ctx.load_window(win, corr)
    Hh, gh, ch = ctx.linearize(st)
    _check_lin(Hh, gh, ch, Ho, go, co, win.W)
    sh, summ_h = ctx.solve(st)
    dt = np.linalg.norm(sh.trans - so.trans, axis=1).max()
    d = [synth.qmul(synth.qconj(so.quat[i]), sh.quat[i]) for i in range(win.W)]
    dr = max(2 * np.arctan2(np.linalg.norm(q[1:]), abs(q[0])) for q in d)
    print(f"C5 f32 vs fp64 oracle: {summ_h.iterations} / {summ_o.iterations} iterations, cost {summ_h.final_cost:.6f} / {summ_o.final_cost:.6f}, "
          f"max |dt| {dt:.2e} m, max angle {dr:.2e} rad")
    assert dt <= 1e-4 and dr <= 1e-5
    assert abs(summ_h.final_cost - summ_o.final_cost) <= 1e-5 * summ_o.final_cost
    ctx.close()


def test_c5_full_size_solve_fp64_path(hip, po, c5_window):
    """The same window through the default fp64 K3: identical iteration history and poses to 1e-9 m (the C2 bar at C5 size)."""
    win, corr = c5_window
    _, (so, summ_o) = _oracle_c5(po, win, corr)
    st = win.init.copy()
    ctx = hip.Context(win.opts)
    ctx.load_window(win, corr)
    sh, summ_h = ctx.solve(st)
    assert summ_h.iterations == summ_o.iterations and summ_h.termination == summ_o.termination
    assert np.linalg.norm(sh.trans - so.trans, axis=1).max() <= 1e-8
    ctx.close()
