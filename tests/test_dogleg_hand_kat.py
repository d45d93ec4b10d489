"""DOGLEG known answers derived BY HAND (nothing here comes from this repository's solver code).

The reference solves with trust_region_strategy_type = DOGLEG (Estimator.cpp:2425).  The bundled Ceres documentation prints progress tables for the
Levenberg-Marquardt default only (tests/test_ceres_docs_kat.py holds the loop to those); for the dogleg STEP itself the textbook construction (Powell's
dogleg, Nocedal & Wright "Numerical Optimization" (4.16); GraphGNSSLibV1.1/docs/source/nnls_solving.rst:196-222 describes the same three cases) has a
closed form on a linear least-squares problem, and a linear problem can be handed to every implementation in this tree -- the numpy restatement, the C
oracle and the HIP solver -- as a window whose ONLY factor is a marginalization prior on one translation block (MarginalizationFactor::Evaluate,
MarginalizationFactor.cpp:242-257: r = r0 + J0 (T - T0), linear in T).

The problem.  J0 = [[1, .6, 0], [0, .8, 0], [0, 0, 1]], r0 = (-2, 0, 0), start at T = T0:
    H = J0^T J0 = [[1, .6, 0], [.6, 1, 0], [0, 0, 1]]   (unit diagonal: Ceres' Jacobi scaling 1 / (1 + sqrt(H_ii)) = 1/2 and the dogleg's own diagonal
        sqrt(H_ii / 4) = 1/2 cancel, so the trust region is the ball |dT| <= radius in metres),
    g = J0^T r0 = (-2, -1.2, 0),   cost(0) = |r0|^2 / 2 = 2.
  Gauss-Newton point   p_gn = -H^-1 g:  H^-1 = (1 / .64) [[1, -.6], [-.6, 1]] on the xy block, so p_gn = (2, 0, 0), |p_gn| = 2
      (Ceres solves (H + mu D^2) p = -g with mu = 1e-8 for the first factorisation: p_gn (1 - O(1e-8))).
  Cauchy point         p_c = -alpha g, alpha = g.g / g.H g = 5.44 / 8.32 = 17/26:   p_c = (17/13, 51/65, 0) = (1.3076923.., 0.7846153.., 0),
      |p_c| = (17/26) sqrt(5.44) = 1.5250180..
  The three cases:
    radius 3.0 >= |p_gn|           -> the Gauss-Newton step (2, 0, 0); cost afterwards ~ 0
    radius 1.0 <= |p_c|            -> the gradient direction scaled to the ball: -(1 / |g|) g = (2, 1.2, 0) / sqrt(5.44) = (0.857493.., 0.514496.., 0)
    |p_c| < radius 1.8 < |p_gn|    -> p_c + beta (p_gn - p_c) with |.| = 1.8: with d = p_gn - p_c = (9/13, -51/65, 0),
                                      beta = (-p_c.d + sqrt((p_c.d)^2 + |d|^2 (1.8^2 - |p_c|^2))) / |d|^2
  After an accepted step with model-exact decrease (a linear problem: ratio 1 > 0.75) the radius becomes max(radius, 3 |step|).
  SUBSPACE_DOGLEG (the batch solve, Estimator.cpp:3278) minimises the model over span{g, p_gn} inside the ball: here that plane is the xy plane and z is
  decoupled with g_z = 0, so its step is the EXACT trust-region step (H + lambda I) p = -g, |p| = radius (secular equation, solved below by bisection)."""
import numpy as np
import pytest

import np_ceres as nc
from glio_amd import synth
from glio_amd import ctypes_types as T

J0 = np.array([[1.0, 0.6, 0.0], [0.0, 0.8, 0.0], [0.0, 0.0, 1.0]])
R0 = np.array([-2.0, 0.0, 0.0])
H = J0.T @ J0
G = J0.T @ R0
P_GN = np.array([2.0, 0.0, 0.0])
P_C = np.array([17.0 / 13.0, 51.0 / 65.0, 0.0])


def hand_step(radius):
    if radius >= 2.0:
        return P_GN.copy()
    if radius <= np.linalg.norm(P_C):
        return np.array([2.0, 1.2, 0.0]) / np.sqrt(5.44) * radius
    d = P_GN - P_C
    pd, dd = P_C @ d, d @ d
    beta = (-pd + np.sqrt(pd * pd + dd * (radius * radius - P_C @ P_C))) / dd
    return P_C + beta * d


def exact_tr_step(radius):
    """(H + lambda I) p = -g with |p| = radius, lambda >= 0 by bisection on the secular equation (|p| decreases monotonically in lambda)"""
    if radius >= 2.0:
        return P_GN.copy()
    lo, hi = 0.0, 100.0
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        p = -np.linalg.solve(H + mid * np.eye(3), G)
        lo, hi = (mid, hi) if np.linalg.norm(p) > radius else (lo, mid)
    return -np.linalg.solve(H + 0.5 * (lo + hi) * np.eye(3), G)


def cost_at(p):
    r = R0 + J0 @ p
    return 0.5 * float(r @ r)


def test_the_hand_numbers_themselves():
    assert np.allclose(H, [[1, .6, 0], [.6, 1, 0], [0, 0, 1]]) and np.allclose(G, [-2, -1.2, 0])
    assert np.allclose(-np.linalg.solve(H, G), P_GN) and np.isclose(G @ G / (G @ H @ G), 17.0 / 26.0) and np.allclose(-(17.0 / 26.0) * G, P_C)
    assert np.isclose(np.linalg.norm(P_C), 1.5250180, atol=1e-7)
    assert np.allclose(hand_step(1.0), [0.8574929, 0.5144958, 0.0], atol=1e-7)
    s = hand_step(1.8)
    assert np.isclose(np.linalg.norm(s), 1.8) and cost_at(P_C) > cost_at(s) > cost_at(P_GN)           # on the dogleg path, between its corner and its end
    # the exact trust-region step is at least as good as the dogleg step of the same length, and different from it inside the bend
    e = exact_tr_step(1.8)
    assert np.isclose(np.linalg.norm(e), 1.8) and cost_at(e) < cost_at(s) - 1e-4


def _np_ceres_first_step(radius, dogleg):
    def evaluate(x):
        r = R0 + J0 @ x
        return 0.5 * float(r @ r), H.copy(), J0.T @ r
    tr = []
    x, summ, hist = nc.minimize(np.zeros(3), evaluate, lambda x, d: x + d, lambda x: x, nc.Options(strategy="dogleg", dogleg=dogleg, initial_radius=radius, max_iterations=1), trace=tr)
    return x, summ, tr


@pytest.mark.parametrize("radius", [1.0, 1.8, 3.0])
def test_numpy_restatement_takes_the_hand_steps(radius):
    x, summ, tr = _np_ceres_first_step(radius, "traditional")
    want = hand_step(radius)
    assert np.abs(x - want).max() < 5e-8 and abs(summ["final_cost"] - cost_at(want)) < 1e-7          # (5e-8: the mu = 1e-8 regularisation of the Gauss-Newton solve)
    assert np.isclose(tr[0]["ratio"], 1.0, atol=1e-6) and np.isclose(summ["final_radius"], max(radius, 3.0 * np.linalg.norm(want)), rtol=1e-7)
    xs, ss, _ = _np_ceres_first_step(radius, "subspace")
    assert np.abs(xs - exact_tr_step(radius)).max() < 5e-7 and abs(ss["final_cost"] - cost_at(exact_tr_step(radius))) < 1e-7


def _prior_only_window(radius, max_iterations=1):
    """a W = 2 window whose only factor is the prior r0 + J0 (T_0 - x0) on the translation of slot 0"""
    win = synth.make_window(W=2, pts_per_scan=64, seed=synth.SEED_BASE + 77)
    win.prior = dict(n=3, lin_jac=np.ascontiguousarray(J0), lin_res=np.ascontiguousarray(R0), blk_slot=np.array([0], np.int32), blk_kind=np.array([T.BLK_TRANS], np.int32),
                     blk_idx=np.array([0], np.int32), blk_x0=np.ascontiguousarray(np.r_[win.init.trans[0], np.zeros(6)][None, :]))
    win.opts.initial_trust_region_radius = radius
    win.opts.max_iterations = max_iterations
    empty = [(np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32), np.zeros(0)) for _ in range(2)]
    st = win.init.copy(); st.n_ddt = 0
    return win, empty, st


@pytest.mark.parametrize("radius", [1.0, 1.8, 3.0])
def test_c_oracle_takes_the_hand_steps(radius):
    from oracle import pyoracle as po
    win, empty, st = _prior_only_window(radius)
    prob = po.Problem(win, empty, use_gnss=False, use_imu=False)
    Hh, gh, c0 = prob.linearize(st)
    assert np.allclose(Hh[:3, :3], H) and np.allclose(gh[:3], G) and np.isclose(c0, 2.0) and not Hh[3:, 3:].any()
    sol, summ = prob.solve(st)
    want = hand_step(radius)
    assert summ.iterations == 1 and np.abs(sol.trans[0] - st.trans[0] - want).max() < 5e-8 and abs(summ.final_cost - cost_at(want)) < 1e-7
    assert np.array_equal(sol.trans[1], st.trans[1]) and np.array_equal(sol.quat, st.quat) and np.array_equal(sol.speed_bias, st.speed_bias)      # (blocks without residuals stay, quirk Q6)
    # two iterations from the smallest radius: the radius became 3 |step| = 3, the remaining 1.25 m are one Gauss-Newton step
    win2, _, st2 = _prior_only_window(1.0, max_iterations=2)
    s2, m2 = po.Problem(win2, empty, use_gnss=False, use_imu=False).solve(st2)
    assert m2.iterations == 2 and np.abs(s2.trans[0] - st2.trans[0] - P_GN).max() < 1e-7 and m2.final_cost < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("radius", [1.0, 1.8, 3.0])
def test_hip_solver_takes_the_hand_steps(radius):
    from glio_amd import capi
    win, empty, st = _prior_only_window(radius)
    ctx = capi.Context(win.opts)
    ctx.load_window(win, empty, use_gnss=False, use_imu=False)
    sol, summ = ctx.solve(st)
    ctx.close()
    want = hand_step(radius)
    assert summ.iterations == 1 and np.abs(sol.trans[0] - st.trans[0] - want).max() < 5e-8 and abs(summ.final_cost - cost_at(want)) < 1e-7
    assert np.array_equal(sol.trans[1], st.trans[1]) and np.array_equal(sol.quat, st.quat)
