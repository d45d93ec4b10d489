"""CPU tests of the oracle's window-level pieces: brute-force association, linearisation, the dogleg
solver, marginalization and the batch linearisation (no reference golden vectors exist; the pins are
independent numpy computations, finite differences and closed-form cases)."""
import numpy as np
import pytest

import numpy_factors as nf
from glio_amd import ctypes_types as T
from glio_amd import synth
from oracle import pyoracle as po


def test_association_matches_numpy_bruteforce(small_window):
    win = small_window
    q2, t2 = po.lidar_pose_for_association(win.opts, win.init.quat[0], win.init.trans[0])
    scan = win.scans[0][:200]
    pts, pl, sc, src, nn = po.associate(win.opts, win.map_pts, scan, q2, t2, want_nn=True)
    R = synth.q2R(q2 / np.linalg.norm(q2))
    M = win.map_pts[:, :3]
    kept = 0
    for i in range(len(scan)):
        p = (R @ scan[i, :3].astype(np.float64) + t2).astype(np.float32)
        e = p[None, :] - M
        d = (e[:, 0] * e[:, 0] + e[:, 1] * e[:, 1]).astype(np.float32) + e[:, 2] * e[:, 2]
        order = np.lexsort((np.arange(len(d)), d))[:5]
        assert np.array_equal(np.sort(order), np.sort(nn[i])), i
        if d[order[4]] >= win.opts.kd_max_radius:
            continue
        A = M[order].astype(np.float64)
        n = np.linalg.lstsq(A, -np.ones(5), rcond=None)[0]
        ninv = 1 / np.linalg.norm(n)
        n = n * ninv
        if np.any(np.abs(A @ n + ninv) > win.opts.surf_dist_thres):
            continue
        pd = np.float32(n @ p.astype(np.float64) + ninv)
        w = np.float32(1 - 0.9 * abs(float(pd)) / float(np.sqrt(np.sqrt(np.float32(p @ p)))))
        if w > win.opts.weight_gate:
            j = int(np.where(src == i)[0][0])
            assert np.allclose(pl[j, :3], w * n, atol=2e-6) and np.isclose(pl[j, 3], w * ninv, rtol=1e-6)
            assert np.isclose(sc[j], win.opts.lidar_const * float(w), rtol=1e-7)
            kept += 1
    assert kept == len(sc) and kept > 150
    assert np.all(np.diff(src) > 0)                       # order preserved


def test_linearize_gradient_matches_finite_differences(small_window, small_corr):
    """Without the IMU factors (their reference Jacobian is knowingly inconsistent, quirk Q15) g must be
    the gradient of the robustified cost under Ceres' (+), and H must be symmetric PSD."""
    win = small_window
    prob = po.Problem(win, small_corr, use_imu=False)
    st = win.init
    H, g, c = prob.linearize(st)
    assert np.allclose(H, H.T, atol=1e-9 * np.abs(H).max())
    assert np.linalg.eigvalsh(H).min() > -1e-8 * np.abs(H).max()
    n = len(g)
    rng = np.random.default_rng(0)
    for k in rng.choice(n, 12, replace=False):
        d = np.zeros(n); d[k] = 1e-6
        sp, sm = st.copy(), st.copy()
        cs, cp, cm = st.c(), sp.c(), sm.c()
        import ctypes as C
        po.lib().orc_state_plus(C.byref(cs), win.W, T.dptr(d), C.byref(cp))
        po.lib().orc_state_plus(C.byref(cs), win.W, T.dptr(-d), C.byref(cm))
        fd = (prob.linearize(sp, want_H=False)[2] - prob.linearize(sm, want_H=False)[2]) / 2e-6
        assert np.isclose(fd, g[k], rtol=2e-4, atol=1e-4 * np.abs(g).max()), (k, fd, g[k])


def test_solver_recovers_ground_truth_on_noise_free_data():
    """Closed-form case: noise-free scans + exact analytic planes -> the optimum is the ground truth."""
    win = synth.make_window(W=3, pts_per_scan=800, seed=synth.SEED_BASE + 21)
    # rebuild noise-free scans from the ground truth
    t_lb = np.array(win.opts.t_lb)
    rng = np.random.default_rng(5)
    for s in range(win.W):
        pw, pid = synth.sample_scene(win.scene, 800, rng, centre=win.gt.trans[s], radius=40.0)
        Rw = synth.q2R(win.gt.quat[s])
        win.scans[s] = np.ascontiguousarray(np.c_[(pw - win.gt.trans[s]) @ Rw + t_lb, np.zeros(len(pw))].astype(np.float32))
        win.scan_plane_id[s] = pid
    corr = synth.analytic_correspondences(win, win.gt)
    prob = po.Problem(win, corr, use_imu=False)
    sol, summ = prob.solve(win.init)
    assert summ.final_cost < 1e-3 * summ.initial_cost
    assert np.linalg.norm(sol.trans - win.gt.trans, axis=1).max() < 5e-3      # float32 inputs
    Hf, gf, cf = prob.linearize(sol)
    assert cf <= summ.final_cost * (1 + 1e-9)


def test_solver_is_deterministic_and_monotone(small_window, small_corr):
    prob = po.Problem(small_window, small_corr)
    s1, m1 = prob.solve(small_window.init)
    s2, m2 = prob.solve(small_window.init)
    assert np.array_equal(s1.trans, s2.trans) and m1.iterations == m2.iterations
    assert m1.final_cost < m1.initial_cost and 1 <= m1.iterations <= small_window.opts.max_iterations
    assert np.allclose(np.linalg.norm(s1.quat, axis=1), 1.0, atol=1e-12)


def test_marginalization_matches_numpy_schur(small_window, small_corr):
    """J0^T J0 and J0^T r0 of orc_marginalize equal the Schur complement built independently in numpy
    from the per-factor evaluators with the reference's "drop the w column" convention (quirk Q8)."""
    win = small_window
    W = win.W
    prob = po.Problem(win, small_corr)
    sol, _ = prob.solve(win.init)
    out = prob.marginalize(sol)
    m, n = 15, 6 * (W - 1) + 9

    def off(slot, kind):
        if slot == 0:
            return (0, 3, 6)[kind]
        if slot == 1:
            return 15 + (0, 3, 6)[kind]
        return 30 + 6 * (slot - 2) + (0, 3)[kind]
    A = np.zeros((m + n, m + n)); b = np.zeros(m + n)

    def add(r, Js, offs):
        for Ji, oi in zip(Js, offs):
            b[oi:oi + Ji.shape[1]] += Ji.T @ r
            for Jj, oj in zip(Js, offs):
                A[oi:oi + Ji.shape[1], oj:oj + Jj.shape[1]] += Ji.T @ Jj
    # prior
    pr = win.prior
    params = []
    for k in range(len(pr["blk_slot"])):
        s, kind = pr["blk_slot"][k], pr["blk_kind"][k]
        params.append([sol.trans[s], sol.quat[s], sol.speed_bias[s]][kind])
    r, J = po.eval_marg(pr, params)
    add(r, [j[:, -3:] if j.shape[1] == 4 else j for j in J], [off(pr["blk_slot"][k], pr["blk_kind"][k]) for k in range(len(J))])
    # IMU(0,1)
    ps = T.GlioPreint(); synth.fill_preint(ps, win.preints[0])
    r, J = po.eval_imu(win.opts, ps, [sol.trans[0], sol.quat[0], sol.speed_bias[0], sol.trans[1], sol.quat[1], sol.speed_bias[1]])
    add(r, [J[0], J[1][:, 1:], J[2], J[3], J[4][:, 1:], J[5]], [off(0, 0), off(0, 1), off(0, 2), off(1, 0), off(1, 1), off(1, 2)])
    # LiDAR of all frames, Huber
    for s in range(W):
        for cp, pl, sc in zip(*small_corr[s]):
            rr, Jt, Jq = po.eval_lidar_plane(win.opts, cp, pl, sc, sol.trans[s], sol.quat[s])
            w = 1.0 if abs(rr) <= 1.0 else 1.0 / abs(rr)
            sw = np.sqrt(w)
            add(np.array([sw * rr]), [sw * Jt[None, :], sw * Jq[None, 1:]], [off(s, 0), off(s, 1)])
    Amm = 0.5 * (A[:m, :m] + A[:m, :m].T)
    wv, V = np.linalg.eigh(Amm)
    Ainv = V @ np.diag(np.where(wv > 1e-8, 1 / wv, 0)) @ V.T
    S = A[m:, m:] - A[m:, :m] @ Ainv @ A[:m, m:]
    bs = b[m:] - A[m:, :m] @ Ainv @ b[:m]
    J0, r0 = out["lin_jac"], out["lin_res"]
    assert np.allclose(J0.T @ J0, S, rtol=1e-7, atol=1e-7 * np.abs(S).max())
    assert np.allclose(J0.T @ r0, bs, rtol=1e-6, atol=1e-6 * np.abs(bs).max())
    # kept blocks are shifted by one slot and linearised at the solution
    assert out["blk_slot"].min() == 0 and out["blk_slot"].max() == W - 2
    assert np.allclose(out["blk_x0"][0][:3], sol.trans[1])


def test_batch_linearize_matches_dense_assembly():
    rng = np.random.default_rng(3)
    K, band, n = 12, 3, 400
    poses = np.zeros((K, 7))
    poses[:, :3] = rng.normal(size=(K, 3)) * 5
    q = rng.normal(size=(K, 4)); poses[:, 3:] = q / np.linalg.norm(q, axis=1, keepdims=True)
    ci = rng.integers(0, K, n).astype(np.int32)
    cj = np.clip(ci + rng.choice([-3, -2, -1, 1, 2, 3], n), 0, K - 1).astype(np.int32)
    keep = ci != cj
    ci, cj = np.ascontiguousarray(ci[keep]), np.ascontiguousarray(cj[keep])
    n = len(ci)
    cp = rng.normal(size=(n, 4)).astype(np.float32)
    pnc = rng.normal(size=(n, 6)); pnc[:, :3] /= np.linalg.norm(pnc[:, :3], axis=1, keepdims=True)
    score = rng.uniform(0.5, 2.5, n)
    Hb, g, cost = po.batch_linearize(K, band, np.ascontiguousarray(poses), ci, cj, cp, np.ascontiguousarray(pnc), score)
    Hd = np.zeros((6 * K, 6 * K)); gd = np.zeros(6 * K); cd = 0.0
    for c in range(n):
        a, b = ci[c], cj[c]
        r, J = po.eval_binary_plane(cp[c], pnc[c], score[c], poses[a, :3], poses[a, 3:], poses[b, :3], poses[b, 3:])
        Ja = np.r_[J[0], J[1] @ nf.plus_jacobian(poses[a, 3:])]
        Jb = np.r_[J[2], J[3] @ nf.plus_jacobian(poses[b, 3:])]
        row = np.zeros(6 * K); row[6 * a:6 * a + 6] = Ja; row[6 * b:6 * b + 6] = Jb
        Hd += np.outer(row, row); gd += row * r; cd += 0.5 * r * r
    assert np.isclose(cost, cd, rtol=1e-12) and np.allclose(g.ravel(), gd, rtol=1e-10, atol=1e-10)
    for k in range(K):
        for d in range(band + 1):
            if k + d < K:
                assert np.allclose(Hb[k, d].reshape(6, 6), Hd[6 * k:6 * k + 6, 6 * (k + d):6 * (k + d) + 6], rtol=1e-10, atol=1e-9)
