"""Pins of the oracle that do not come from the oracle itself.  The reference has no tests and
cannot be built here, so "what Ceres would compute" is re-derived independently:

* the autodiff functors' `operator()` transcribed statement by statement in float64 torch and differentiated with
  torch.autograd -- Ceres' Jet evaluation is exact forward-mode AD of the same expression tree, so the GLOBAL Jacobians
  must agree to rounding with the oracle's hand-derived ones (LidarPlaneNormFactor LidarKeyframeFactor.h:87-103,
  BinaryLidarPlaneNormFactor :132-150, tcdopplerFactor dopp_factor.hpp:24-75);
* the marginalization's eigen-decomposition root (MarginalizationFactor.cpp:176-201) reproduced with scipy.linalg.eigh
  from the oracle's own Schur complement: the stored linearized_jacobians / residuals as the reference forms them;
* the converged optimum of a small window cross-checked with scipy.optimize.least_squares on the oracle's residual
  vector (a different solver, same minimum)."""
import numpy as np
import pytest
import torch

from glio_amd import ctypes_types as T
from glio_amd import synth
from oracle import pyoracle as po

torch.set_default_dtype(torch.float64)


# ---- Eigen::Quaternion pieces the functors use, on torch scalars/vectors (w, x, y, z storage order as in the ctor)
def q_inverse(q):                       # Eigen: conjugate / squaredNorm
    n2 = (q * q).sum()
    return torch.stack([q[0], -q[1], -q[2], -q[3]]) / n2


def q_mul(a, b):
    return torch.stack([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                        a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                        a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3],
                        a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1]])


def q_rot(q, v):                        # Eigen QuaternionBase::_transformVector
    u = q[1:]
    uv = torch.linalg.cross(u, v)
    uv = uv + uv
    return v + q[0] * uv + torch.linalg.cross(u, uv)


def rand_q(rng):
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def test_lidar_plane_functor_autograd():
    rng = np.random.default_rng(11)
    o = synth.default_opts()
    o.q_lb[:] = list(rand_q(rng))
    for trial in range(5):
        t, q = rng.normal(size=3) * 5, rand_q(rng) * (1.0 + 0.01 * trial)        # also slightly non-unit: Ceres evaluates at whatever the block holds
        cp = (rng.normal(size=4) * 8).astype(np.float32)
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        plane = np.r_[0.7 * n, 0.7 * 3.0].astype(np.float32)
        score = 7.5 * 0.7
        r, Jt, Jq = po.eval_lidar_plane(o, cp, plane, score, t, q)
        qlb, tlb = torch.tensor(list(o.q_lb)), torch.tensor(list(o.t_lb))
        cpt, nt, d = torch.tensor(cp[:3].astype(np.float64)), torch.tensor(plane[:3].astype(np.float64)), float(plane[3])

        def functor(tt, qq):            # LidarKeyframeFactor.h:89-101
            point_w = q_rot(q_inverse(qlb), cpt - tlb)
            point_w = q_rot(qq, point_w) + tt
            return score * (nt.dot(point_w) + d)
        tt, qq = torch.tensor(t, requires_grad=True), torch.tensor(q, requires_grad=True)
        val = functor(tt, qq)
        gt_, gq_ = torch.autograd.grad(val, (tt, qq))
        assert abs(val.item() - r) <= 1e-12 * max(1.0, abs(r))
        assert np.allclose(Jt, gt_.numpy(), rtol=1e-12, atol=1e-12)
        assert np.allclose(Jq, gq_.numpy(), rtol=1e-12, atol=1e-11)


def test_binary_plane_functor_autograd():
    rng = np.random.default_rng(12)
    for trial in range(5):
        t1, q1, t2, q2 = rng.normal(size=3) * 4, rand_q(rng), rng.normal(size=3) * 4, rand_q(rng)
        cp = (rng.normal(size=4) * 6).astype(np.float32)
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        pnc = np.r_[n, rng.normal(size=3) * 5]
        score = 2.5 * 0.8
        r, J = po.eval_binary_plane(cp, pnc, score, t1, q1, t2, q2)
        cpt, pn = torch.tensor(cp[:3].astype(np.float64)), torch.tensor(pnc)

        def functor(a, b, c, d):        # LidarKeyframeFactor.h:134-148
            point_w = q_rot(b, cpt) + a
            normal_oth = q_rot(d, pn[:3])
            cent_oth = q_rot(d, pn[3:]) + c
            return score * normal_oth.dot(point_w - cent_oth)
        P = [torch.tensor(x, requires_grad=True) for x in (t1, q1, t2, q2)]
        val = functor(*P)
        G = torch.autograd.grad(val, P)
        assert abs(val.item() - r) <= 1e-12 * max(1.0, abs(r))
        for a, b in zip(J, G):
            assert np.allclose(a, b.numpy(), rtol=1e-12, atol=1e-11)


def test_doppler_functor_autograd():
    win = synth.make_window(W=3, pts_per_scan=64, with_gnss=True, seed=synth.SEED_BASE + 2)
    rng = np.random.default_rng(13)
    OMG, C_ = 7.2921151467e-5, 2.99792458e8
    for f in win.dop[:6]:
        Pi, Pj = win.init.trans[f.slot_i].copy(), win.init.trans[f.slot_j].copy()
        SBi, SBj = win.init.speed_bias[f.slot_i].copy(), win.init.speed_bias[f.slot_j].copy()
        ddt = rng.normal(size=win.init.n_ddt) * 3
        r, J = po.eval_doppler(f, Pi, SBi, Pj, SBj, ddt, win.frame.yaw_enu_local, np.array(win.frame.anc_ecef))
        sv_pos, sv_vel = torch.tensor(list(f.sat_pos)), torch.tensor(list(f.sat_vel))
        Rl = torch.tensor(list(f.R_ecef_local)).reshape(3, 3)
        lever, anc = torch.tensor(list(f.lever_arm)), torch.tensor(list(win.frame.anc_ecef))

        def functor(pi, vi, pj, vj, rcv):        # dopp_factor.hpp:27-72 (stateVi = the 9-block, first three used)
            local_pos = f.ratio * pi + (1.0 - f.ratio) * pj + lever
            local_vel = f.ratio * vi[:3] + (1.0 - f.ratio) * vj[:3]
            P_ecef = Rl @ local_pos + anc
            V_ecef = Rl @ local_vel
            r2s = sv_pos - P_ecef
            unit = r2s / r2s.norm()
            sag = OMG / C_ * (sv_vel[0] * P_ecef[1] + sv_pos[0] * V_ecef[1] - sv_vel[1] * P_ecef[0] - sv_pos[1] * V_ecef[0])
            est = (sv_vel - V_ecef).dot(unit) + sag + rcv[f.epoch] - f.sv_ddt
            return (est + f.doppler * f.lamda) / f.var
        P = [torch.tensor(x, requires_grad=True) for x in (Pi, SBi, Pj, SBj, ddt)]
        val = functor(*P)
        G = torch.autograd.grad(val, P)
        assert abs(val.item() - r) <= 1e-9 * max(1.0, abs(r))
        for k in range(4):
            assert np.allclose(J[k], G[k].numpy(), rtol=1e-9, atol=1e-12), k
        assert np.isclose(J[4][0], G[4].numpy()[f.epoch], rtol=1e-12)
        assert np.count_nonzero(G[4].numpy()) == 1


def test_marginalization_eigen_root_with_scipy(small_window, small_corr):
    """MarginalizationInfo::marginalize, second half (MarginalizationFactor.cpp:192-201): S = V diag(lam) V^T with eigenvalues
    below eps = 1e-8 dropped, linearized_jacobians = sqrt(lam) V^T, linearized_residuals = lam^-1/2 V^T b.  Rebuilt here with
    scipy.linalg.eigh from S = J0^T J0, b = J0^T r0 of the oracle's output: the same S, b and |r0|^2 must come back, and the
    oracle's root must be that eigen root up to the sign of each eigenvector (row of J0)."""
    import scipy.linalg
    win = small_window
    prob = po.Problem(win, small_corr)
    sol, _ = prob.solve(win.init)
    m = prob.marginalize(sol)
    J0, r0 = m["lin_jac"], m["lin_res"]
    S, b = J0.T @ J0, J0.T @ r0
    lam, V = scipy.linalg.eigh(S)
    keep = lam > 1e-8
    Sq = np.where(keep, np.sqrt(np.where(keep, lam, 1.0)), 0.0)
    Sqi = np.where(keep, 1.0 / np.where(keep, Sq, 1.0), 0.0)
    J_ref = Sq[:, None] * V.T
    r_ref = Sqi * (V.T @ b)
    assert np.allclose(J_ref.T @ J_ref, S, rtol=1e-10, atol=1e-10 * np.abs(S).max())
    assert np.allclose(J_ref.T @ r_ref, b, rtol=1e-8, atol=1e-8 * np.abs(b).max())
    assert np.isclose(r_ref @ r_ref, r0 @ r0, rtol=1e-8)
    # the oracle forms its root the reference's way: its rows are eigenvectors of S scaled by sqrt(lam)
    rows = J0[np.linalg.norm(J0, axis=1) > 1e-12]
    for row in rows[:: max(1, len(rows) // 12)]:
        v = row / np.linalg.norm(row)
        l = np.linalg.norm(row) ** 2
        assert np.linalg.norm(S @ v - l * v) <= 1e-7 * max(l, lam.max() * 1e-9), "row of linearized_jacobians is not sqrt(lam) * eigenvector"


def test_converged_optimum_matches_scipy_minimiser(small_window, small_corr):
    """A different optimiser on the same objective: plain Gauss-Newton (numpy solve, no trust region, no Jacobi scaling)
    iterated to a fixed point on the oracle's H, g must end where the oracle's Ceres-style dogleg ends when IT runs with tight
    tolerances (the default 1e-6 function tolerance stops ~1e-5 m earlier); scipy's BFGS polish from there must not move
    the cost, and random perturbations must not lower it."""
    import copy
    import scipy.optimize
    from glio_amd import ctypes_types as TT
    win = small_window
    o = TT.GlioOpts.from_buffer_copy(win.opts)
    o.function_tolerance, o.parameter_tolerance, o.gradient_tolerance, o.max_iterations = 1e-15, 1e-14, 1e-14, 60
    wt = copy.copy(win)
    wt.opts = o
    # without the IMU factors: the reference's ImuFactor Jacobian is knowingly inconsistent with its residual (quirk Q15), so
    # J^T r = 0 is not the minimum of the cost there and a cost-monitoring solver stops short of the Gauss-Newton fixed point
    prob = po.Problem(wt, small_corr, use_imu=False)
    sol, summ = prob.solve(win.init)
    n = prob.n(win.init)
    # Gauss-Newton iterations written out with numpy (no trust region, no scaling): x <- x (+) -(H + 1e-9 I)^-1 g at the current point
    st = win.init.copy()
    for it in range(40):
        H, g, c = prob.linearize(st)
        step = -np.linalg.solve(H + 1e-9 * np.eye(n), g)
        nst = st.copy()
        for s in range(win.W):
            nst.trans[s] += step[15 * s:15 * s + 3]
            nst.quat[s] = po.quat_plus(st.quat[s], step[15 * s + 3:15 * s + 6])
            nst.speed_bias[s] += step[15 * s + 6:15 * s + 15]
        nst.rcv_ddt[:nst.n_ddt] += step[15 * win.W:]
        st = nst
        if np.abs(step).max() < 1e-13:
            break
    _, g_end, c_end = prob.linearize(st)
    # a cost-monitoring solver cannot resolve the optimum below ~sqrt(eps) of the cost scale: 2e-6 m, far inside the 1e-4 m gate
    assert np.linalg.norm(sol.trans - st.trans, axis=1).max() <= 2e-6, "dogleg optimum vs plain Gauss-Newton optimum"
    assert abs(summ.final_cost - c_end) <= 1e-9 * c_end
    # scipy on cost(x (+) d) around the fixed point: nothing to gain
    def plus(base, d):
        stp = base.copy()
        for s_ in range(win.W):
            stp.trans[s_] += d[15 * s_:15 * s_ + 3]
            stp.quat[s_] = po.quat_plus(base.quat[s_], d[15 * s_ + 3:15 * s_ + 6])
            stp.speed_bias[s_] += d[15 * s_ + 6:15 * s_ + 15]
        stp.rcv_ddt[:stp.n_ddt] += d[15 * win.W:]
        return stp

    def fun(d):
        _, g_, c_ = prob.linearize(plus(st, d))
        return c_, g_          # g is the gradient w.r.t. the local parameterisation AT that point: exact at d = 0, first order nearby
    res = scipy.optimize.minimize(lambda d: prob.linearize(plus(st, d), want_H=False)[2], np.zeros(n), method="Powell",
                                  options=dict(maxfev=400, xtol=1e-10, ftol=1e-14))
    assert res.fun >= c_end - 1e-9 * c_end
    # and a derivative-free sanity check of the minimum itself: random local perturbations do not lower the cost
    rng = np.random.default_rng(3)
    for _ in range(10):
        d = rng.normal(size=n) * 1e-4
        stp = st.copy()
        for s in range(win.W):
            stp.trans[s] += d[15 * s:15 * s + 3]
            stp.quat[s] = po.quat_plus(st.quat[s], d[15 * s + 3:15 * s + 6])
            stp.speed_bias[s] += d[15 * s + 6:15 * s + 15]
        stp.rcv_ddt[:stp.n_ddt] += d[15 * win.W:]
        assert prob.linearize(stp, want_H=False)[2] >= c_end - 1e-9 * c_end


def test_delta_q_functor_autograd():
    """delta_q_factor_auto (LidarKeyframeFactor.h:283-303): residual 10000 (dq^-1 qi^-1 qj).vec with Eigen's inverse
    (conjugate / squaredNorm); the oracle's hand-derived 3x4 global Jacobians against torch.autograd, also at non-unit qi."""
    rng = np.random.default_rng(21)
    for trial in range(5):
        dq, qi, qj = rand_q(rng), rand_q(rng) * (1.0 + 0.02 * trial), rand_q(rng)
        r, J = po.eval_delta_q(dq, qi, qj)
        dqt = torch.tensor(dq)

        def functor(a, b):
            return 10000.0 * q_mul(q_mul(q_inverse(dqt), q_inverse(a)), b)[1:]
        a, b = torch.tensor(qi, requires_grad=True), torch.tensor(qj, requires_grad=True)
        Ja, Jb = torch.autograd.functional.jacobian(functor, (a, b))
        assert np.allclose(r, functor(a, b).detach().numpy(), rtol=1e-13, atol=1e-9)
        assert np.allclose(J[0], Ja.numpy(), rtol=1e-11, atol=1e-7) and np.allclose(J[1], Jb.numpy(), rtol=1e-11, atol=1e-7)


def _torch_batch_cost(K, poses_t, con, dq, dd, frame):
    """The whole batch pose problem's cost, 0.5 sum r^2, transcribed functor by functor in torch float64: BinaryLidarPlaneNormFactor
    (LidarKeyframeFactor.h:134-148), delta_q_factor_auto (:283-303), dd_psr_factor_20 (dd_psr_factor.hpp:25-171, identity weight as
    the batch stage builds it).  poses_t [K][7] = t, q (w x y z)."""
    from glio_amd import synth
    ci, cj, cp, nc, score = con
    cost = torch.zeros((), dtype=torch.float64)
    for k in range(len(ci)):
        a, b = int(ci[k]), int(cj[k])
        pw = q_rot(poses_t[a, 3:], torch.tensor(cp[k, :3].astype(np.float64))) + poses_t[a, :3]
        n_o = q_rot(poses_t[b, 3:], torch.tensor(nc[k, :3]))
        c_o = q_rot(poses_t[b, 3:], torch.tensor(nc[k, 3:])) + poses_t[b, :3]
        r = float(score[k]) * n_o.dot(pw - c_o)
        cost = cost + 0.5 * r * r
    for k in range(len(dq[0])):
        r = 10000.0 * q_mul(q_mul(q_inverse(torch.tensor(dq[2][k])), q_inverse(poses_t[int(dq[0][k]), 3:])), poses_t[int(dq[1][k]), 3:])[1:]
        cost = cost + 0.5 * r.dot(r)
    Ree = torch.tensor(synth.ecef2rotation(np.array(frame.anc_ecef)))
    anc = torch.tensor(np.array(frame.anc_ecef))
    for f in dd:
        ns, m = f.n_sat, f.master
        lp = f.ratio * poses_t[f.slot_i, :3] + (1.0 - f.ratio) * poses_t[f.slot_j, :3]
        Pe = Ree @ lp + anc                                       # yaw_enu_local = 0 in the generator
        st = torch.tensor(np.array(f.station))
        def rng_(pos, to):
            return torch.linalg.norm(torch.tensor(np.array(pos)) - to)
        for i in range(ns):
            if i == m:
                continue
            est = (rng_(f.user_sat_pos[i], Pe) - rng_(f.ref_sat_pos[i], st)) - (rng_(f.user_sat_pos[m], Pe) - rng_(f.ref_sat_pos[m], st))
            obs = (f.user_psr[i] - f.ref_psr[i]) - (f.user_psr[m] - f.ref_psr[m])
            w = 0.05 if abs(float(est.detach()) - obs) > f.threshold else 1.0
            r = w * (est - obs)
            cost = cost + 0.5 * r * r
    return cost


def test_batch_problem_cost_and_gradient_against_a_torch_transcription():
    """orc_batch_linearize_full as a whole: the cost equals the torch transcription's, the gradient (in the local parameterisation:
    translation additive, quaternion [cos|d|, sin|d| d/|d|] (x) q) equals autograd's through the same Plus, and the solved problem is
    a stationary point whose cost random perturbations do not lower."""
    from glio_amd import batch
    K, band = 8, 4
    gt, init = batch.make_poses(K, seed=51, perturb=(0.05, 0.003))
    ci, cj, cp, nc, score = batch.make_constraints(gt, 0, K, 12, band, seed=51)
    con = (ci, cj, cp.numpy(), nc.numpy(), score.numpy())
    odo = gt.copy()
    dq = batch.delta_q_pairs(odo, 2)
    dd, frame = batch.make_batch_gnss(gt, seed=51, sats_per_sys=6)
    for f in dd:
        f.threshold = 1e9                                          # (no weight switching: the cost is smooth)
    P = po.BatchProblem(K, band, *con, dq=dq, dd=dd, frame=frame)
    H, g, cost = P.linearize(init)
    x0 = torch.tensor(init)

    def plus(d):
        t = x0[:, :3] + d[:, :3]
        nrm = torch.linalg.norm(d[:, 3:], dim=1, keepdim=True)
        dqv = torch.cat([torch.cos(nrm), torch.sin(nrm) / nrm * d[:, 3:]], dim=1)
        q = torch.stack([q_mul(dqv[k], x0[k, 3:]) for k in range(K)])
        return torch.cat([t, q], dim=1)
    d = torch.full((K, 6), 1e-30, dtype=torch.float64, requires_grad=True)      # (|d| -> 0 without the 0/0)
    c_t = _torch_batch_cost(K, plus(d), con, dq, dd, frame)
    (g_t,) = torch.autograd.grad(c_t, d)
    assert abs(c_t.item() - cost) <= 1e-10 * cost
    assert np.abs(g_t.numpy() - g).max() <= 1e-7 * np.abs(g).max()
    # the converged solution: gradient ~ 0 and no nearby point with a lower cost
    from glio_amd import ctypes_types as T
    sol, summ = P.solve(init, T.batch_tr_opts(max_iterations=200))
    Hs, gs, cs = P.linearize(sol)
    assert np.abs(gs).max() <= 1e-5 * np.abs(g).max()
    rng = np.random.default_rng(3)
    for _ in range(50):
        dlt = rng.normal(0, 1e-4, (K, 6))
        pert = sol.copy(); pert[:, :3] += dlt[:, :3]
        for k in range(K):
            pert[k, 3:] = po.quat_plus(sol[k, 3:], dlt[k, 3:])
        assert P.linearize(pert)[2] >= cs * (1 - 1e-12)
