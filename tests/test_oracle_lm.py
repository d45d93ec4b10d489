"""CPU: the oracle's Levenberg-Marquardt strategy (Ceres 1.14 LevenbergMarquardtStrategy + TrustRegionMinimizer,
used by the front end, reference GLIO/src/LidarOdometry.cpp:521-530)."""
import numpy as np

from glio_amd import synth
from oracle import pyoracle as po


def _solve(win, corr, strategy, **kw):
    win.opts.trust_region_strategy = strategy
    try:
        prob = po.Problem(win, corr, **kw)
        st = win.init.copy(); st.n_ddt = 0
        return prob.solve(st)
    finally:
        win.opts.trust_region_strategy = 0


def test_lm_and_dogleg_reach_the_same_minimum(small_window, small_corr):
    kw = dict(use_gnss=False, use_prior=False)
    sd, md = _solve(small_window, small_corr, 0, **kw)
    sl, ml = _solve(small_window, small_corr, 1, **kw)
    assert ml.termination in (1, 2, 3) and md.termination in (1, 2, 3)          # a tolerance, not failure / iteration cap
    assert abs(ml.final_cost - md.final_cost) <= 1e-5 * md.final_cost
    assert np.linalg.norm(sl.trans - sd.trans, axis=1).max() < 2e-3
    assert ml.final_cost < 0.2 * ml.initial_cost


def test_lm_radius_grows_on_good_steps(small_window, small_corr):
    # every accepted step with quality ~1 multiplies the radius by 3 (1 / max(1/3, 1 - (2 rho - 1)^3))
    s, m = _solve(small_window, small_corr, 1, use_gnss=False, use_prior=False)
    assert m.successful_steps >= 2
    assert m.final_radius > 1e4 * 3 ** (m.successful_steps - 1) * 0.3
