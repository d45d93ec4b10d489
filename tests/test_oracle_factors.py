"""CPU tests that pin the oracle (oracle/) -- the reference has no tests or golden vectors for this
path, so the pins are: central finite differences, an independent numpy
transcription (tests/numpy_factors.py) and closed-form cases."""
import ctypes as C

import numpy as np
import pytest

import numpy_factors as nf
from glio_amd import ctypes_types as T
from glio_amd import synth
from oracle import pyoracle as po


def fd_local(fun, blocks, kinds, eps=1e-6):
    """Central differences of fun(blocks) w.r.t. the Ceres local parameterisation of each block."""
    r0 = np.atleast_1d(fun(blocks))
    out = []
    for b, kind in enumerate(kinds):
        ls = 3 if kind == "q" else len(blocks[b])
        J = np.zeros((len(r0), ls))
        for k in range(ls):
            d = np.zeros(ls)
            d[k] = eps
            plus = [x.copy() for x in blocks]
            minus = [x.copy() for x in blocks]
            if kind == "q":
                plus[b] = po.quat_plus(blocks[b], d)
                minus[b] = po.quat_plus(blocks[b], -d)
            else:
                plus[b] = blocks[b] + d
                minus[b] = blocks[b] - d
            J[:, k] = (np.atleast_1d(fun(plus)) - np.atleast_1d(fun(minus))) / (2 * eps)
        out.append(J)
    return out


def to_local(Jg, q):
    return Jg @ nf.plus_jacobian(q)


def rand_q(rng):
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def test_quat_plus_matches_ceres_definition():
    rng = np.random.default_rng(1)
    q, d = rand_q(rng), rng.normal(size=3) * 0.3
    assert np.allclose(po.quat_plus(q, d), nf.quat_plus(q, d), atol=1e-15)
    assert np.allclose(po.quat_plus(q, np.zeros(3)), q)
    assert abs(np.linalg.norm(po.quat_plus(q, d)) - 1) < 1e-14


def test_lidar_plane_factor_fd_and_closed_form():
    rng = np.random.default_rng(2)
    o = synth.default_opts()
    o.q_lb[:] = list(rand_q(rng))          # exercise a non-identity extrinsic too
    t, q = rng.normal(size=3), rand_q(rng)
    cp = rng.normal(size=4).astype(np.float32) * 5
    n = rng.normal(size=3); n /= np.linalg.norm(n)
    plane = np.r_[0.8 * n, 0.8 * 2.0].astype(np.float32)
    score = 7.5 * 0.8
    r, Jt, Jq = po.eval_lidar_plane(o, cp, plane, score, t, q)
    fun = lambda B: po.eval_lidar_plane(o, cp, plane, score, B[0], B[1], want_J=False)[0]
    Jfd = fd_local(fun, [t, q], ["v", "q"])
    assert np.allclose(Jt, Jfd[0][0], rtol=1e-7, atol=1e-7)
    assert np.allclose(to_local(Jq[None, :], q)[0], Jfd[1][0], rtol=1e-7, atol=1e-7)
    # closed form of the fused local Jacobian the HIP kernel uses: 2 s ((R p_b) x n)
    qlb = np.array(o.q_lb)
    pb = nf.rot(nf.qinv(qlb), cp[:3].astype(float) - np.array(o.t_lb))
    Rpb = nf.rot(q, pb)
    nd = plane[:3].astype(float)
    assert np.allclose(to_local(Jq[None, :], q)[0], 2 * score * np.cross(Rpb, nd), rtol=1e-12, atol=1e-12)
    assert np.isclose(r, score * (nd @ (Rpb + t) + float(plane[3])), rtol=1e-14)


def _preint(rng):
    n = 40
    acc = rng.normal(0, 0.5, (n + 1, 3)) + np.array([0, 0, 9.8])
    gyr = rng.normal(0, 0.2, (n + 1, 3))
    return synth.preintegrate(acc, gyr, np.full(n, 0.01), rng.normal(0, 0.01, 3), rng.normal(0, 0.01, 3))


def test_imu_factor_matches_independent_transcription():
    rng = np.random.default_rng(3)
    o = synth.default_opts()
    pre = _preint(rng)
    ps = T.GlioPreint()
    synth.fill_preint(ps, pre)
    params = [rng.normal(size=3), rand_q(rng) * 1.001, rng.normal(size=9) * 0.1, rng.normal(size=3), rand_q(rng), rng.normal(size=9) * 0.1]
    r, J = po.eval_imu(o, ps, params)
    r2, J2 = nf.imu_factor(pre, params, o.gravity)
    assert np.allclose(r, r2, rtol=1e-9, atol=1e-9 * np.abs(r2).max())
    for a, b in zip(J, J2):
        assert np.allclose(a, b, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(b).max()))


def test_imu_factor_fd_documents_reference_jacobian_quirk():
    """Finite differences agree with the analytic Jacobians everywhere EXCEPT the (P,V) x Qi block,
    where the reference writes the Jacobian of Qi*tmp for a residual that contains Qi^-1*tmp
    (ImuFactor.h:81-95; quirk Q15 in DESIGN.md).  The restatement keeps the reference's block."""
    rng = np.random.default_rng(4)
    o = synth.default_opts()
    pre = _preint(rng)
    ps = T.GlioPreint()
    synth.fill_preint(ps, pre)
    pre_cov = np.asarray(pre["covariance"])
    # near-consistent states so the rotation residual is small (first-order formulas)
    Qi = rand_q(rng)
    dqq = nf.qmul(Qi, np.asarray(pre["delta_q"]))
    Qj = nf.quat_plus(dqq / np.linalg.norm(dqq), rng.normal(size=3) * 1e-3)
    params = [rng.normal(size=3), Qi, np.r_[rng.normal(size=3), pre["linearized_ba"], pre["linearized_bg"]],
              rng.normal(size=3), Qj, np.r_[rng.normal(size=3), pre["linearized_ba"], pre["linearized_bg"]]]
    r, J = po.eval_imu(o, ps, params)
    fun = lambda B: po.eval_imu(o, ps, B, want_J=False)[0]
    Jfd = fd_local(fun, params, ["v", "q", "v", "v", "q", "v"], eps=1e-6)
    S = np.linalg.cholesky(np.linalg.inv(pre_cov)).T
    Sinv = np.linalg.inv(S)
    loc = [J[0], to_local(J[1], params[1]), J[2], J[3], to_local(J[4], params[4]), J[5]]
    for b in range(6):
        A, B = Sinv @ loc[b], Sinv @ Jfd[b]          # un-whitened rows: P,R,V,BA,BG
        scale = max(1.0, np.abs(B).max())
        if b == 1:
            assert np.allclose(A[3:6], B[3:6], atol=2e-3 * scale)          # rotation rows: first-order in the residual
            assert not np.allclose(A[0:3], B[0:3], atol=1e-2 * scale)      # the quirk
            # ... and the as-written block equals the Jacobian of Qi*tmp
            g = np.array([0, 0, -o.gravity]); dt = pre["sum_dt"]
            Qin = params[1] / np.linalg.norm(params[1])
            tmp = -0.5 * g * dt * dt + params[3] - params[0] - params[2][:3] * dt
            assert np.allclose(A[0:3], -2 * nf.skew(nf.rot(Qin, tmp)), atol=1e-9 * scale)
        elif b in (2, 4):
            assert np.allclose(np.delete(A, [3, 4, 5], 0), np.delete(B, [3, 4, 5], 0), atol=1e-5 * scale)
            assert np.allclose(A[3:6], B[3:6], atol=5e-3 * scale)
        else:
            assert np.allclose(A, B, atol=1e-5 * scale)


def test_marg_factor_fd(small_window):
    win = small_window
    pr = win.prior
    rng = np.random.default_rng(5)
    params, kinds = [], []
    for b in range(len(pr["blk_slot"])):
        k = pr["blk_kind"][b]
        x0 = pr["blk_x0"][b]
        if k == T.BLK_QUAT:
            params.append(po.quat_plus(x0[:4], rng.normal(size=3) * 1e-3)); kinds.append("q")
        else:
            sz = 3 if k == T.BLK_TRANS else 9
            params.append(x0[:sz] + rng.normal(size=sz) * 1e-2); kinds.append("v")
    r, J = po.eval_marg(pr, params)
    Jfd = fd_local(lambda B: po.eval_marg(pr, B, want_J=False)[0], params, kinds)
    for b in range(len(params)):
        Jl = to_local(J[b], params[b]) if kinds[b] == "q" else J[b]
        assert np.allclose(Jl, Jfd[b], atol=5e-3 * np.abs(Jfd[b]).max())
    # zero displacement -> residual is r0
    x0s = [pr["blk_x0"][b][:(3 if pr["blk_kind"][b] == 0 else 4 if pr["blk_kind"][b] == 1 else 9)].copy() for b in range(len(params))]
    r0, _ = po.eval_marg(pr, x0s, want_J=False)
    assert np.allclose(r0, pr["lin_res"], atol=1e-12)


def test_dd_psr_factor_fd_and_padding(small_window):
    win = small_window
    f = win.dd[1]
    Pi, Pj = win.init.trans[f.slot_i].copy(), win.init.trans[f.slot_j].copy()
    anc = np.array(win.frame.anc_ecef)
    r, J = po.eval_dd_psr(f, Pi, Pj, 0.0, anc)
    fun = lambda B: po.eval_dd_psr(f, B[0], B[1], 0.0, anc, want_J=False)[0]
    Jfd = fd_local(fun, [Pi, Pj], ["v", "v"], eps=1.0)
    assert np.allclose(J[0], Jfd[0], atol=1e-6) and np.allclose(J[1], Jfd[1], atol=1e-6)
    ns = f.n_sat
    assert np.all(r[ns - 1:] == 0) and np.all(J[0][ns - 1:] == 0)      # rows padded to 19 (quirk Q13)
    assert np.abs(r[:ns - 1]).max() < 20.0                             # DD cancels the clock terms
    # outlier down-weighting: threshold below the residual scales that row's raw residual by 0.05
    import copy
    g = copy.copy(f)
    g.threshold = 0.0
    eye = np.eye(ns - 1)
    f2, g2 = copy.copy(f), copy.copy(g)
    f2.weight[:eye.size] = list(eye.ravel()); g2.weight[:eye.size] = list(eye.ravel())
    ra, _ = po.eval_dd_psr(f2, Pi, Pj, 0.0, anc, want_J=False)
    rb, _ = po.eval_dd_psr(g2, Pi, Pj, 0.0, anc, want_J=False)
    assert np.allclose(rb[:ns - 1], 0.05 * ra[:ns - 1])


def test_doppler_factor_fd(small_window):
    win = small_window
    f = win.dop[7]
    st = win.init
    args = [st.trans[f.slot_i].copy(), st.speed_bias[f.slot_i].copy(), st.trans[f.slot_j].copy(), st.speed_bias[f.slot_j].copy(),
            np.linspace(1, 2, st.n_ddt)]
    anc = np.array(win.frame.anc_ecef)
    r, J = po.eval_doppler(f, *args, 0.0, anc)
    fun = lambda B: po.eval_doppler(f, B[0], B[1], B[2], B[3], B[4], 0.0, anc, want_J=False)[0]
    Jfd = fd_local(fun, args, ["v"] * 5, eps=1.0)
    for b in range(4):
        assert np.allclose(J[b], Jfd[b][0], atol=1e-6 * max(1, np.abs(Jfd[b]).max()))
    ddt_row = Jfd[4][0]
    assert np.isclose(ddt_row[f.epoch], J[4][0]) and np.count_nonzero(np.abs(ddt_row) > 1e-9) == 1
    # at the true state the residual is noise-level (|r| ~ sigma/var)
    gt = win.gt
    rt, _ = po.eval_doppler(f, gt.trans[f.slot_i], gt.speed_bias[f.slot_i], gt.trans[f.slot_j], gt.speed_bias[f.slot_j], gt.rcv_ddt, 0.0, anc, want_J=False)
    assert abs(rt) < 5.0


def test_ecef2rotation_is_orthonormal_and_up_points_out():
    R = np.zeros(9)
    po.lib().orc_ecef2rotation(T.dptr(synth.ANCHOR_ECEF.copy()), T.dptr(R))
    R = R.reshape(3, 3)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-12)
    assert np.allclose(R, synth.ecef2rotation(synth.ANCHOR_ECEF), atol=1e-15)
    up = R[:, 2]
    assert up @ synth.ANCHOR_ECEF / np.linalg.norm(synth.ANCHOR_ECEF) > 0.9999


def test_binary_plane_factor_fd():
    rng = np.random.default_rng(8)
    t1, q1, t2, q2 = rng.normal(size=3), rand_q(rng), rng.normal(size=3), rand_q(rng)
    cp = rng.normal(size=4).astype(np.float32)
    pnc = rng.normal(size=6)
    r, J = po.eval_binary_plane(cp, pnc, 2.5, t1, q1, t2, q2)
    fun = lambda B: po.eval_binary_plane(cp, pnc, 2.5, *B)[0]
    Jfd = fd_local(fun, [t1, q1, t2, q2], ["v", "q", "v", "q"])
    assert np.allclose(J[0], Jfd[0][0], atol=1e-7) and np.allclose(J[2], Jfd[2][0], atol=1e-7)
    assert np.allclose(to_local(J[1][None], q1)[0], Jfd[1][0], atol=1e-7)
    assert np.allclose(to_local(J[3][None], q2)[0], Jfd[3][0], atol=1e-7)


def test_plane_qr_solve_matches_lstsq_and_exact_plane():
    rng = np.random.default_rng(9)
    for _ in range(50):
        A = rng.normal(size=(5, 3)) * rng.uniform(0.1, 50)
        b = -np.ones(5)
        x = po.plane_qr_solve(A, b)
        xr = np.linalg.lstsq(A, b, rcond=None)[0]
        assert np.allclose(x, xr, rtol=1e-9, atol=1e-12)
    n = np.array([0.0, 0.0, 1.0]); d = 2.0            # plane z = -2  ->  n.p + d = 0
    P = np.c_[rng.normal(size=(5, 2)) * 3, np.full(5, -2.0)]
    x = po.plane_qr_solve(P, -np.ones(5))
    assert np.allclose(x / np.linalg.norm(x), n, atol=1e-12) and np.isclose(1 / np.linalg.norm(x), d)
