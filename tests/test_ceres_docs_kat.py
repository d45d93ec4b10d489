"""Known answers for the trust-region loop from the AUTHORS of the third-party solver the reference delegates to.

The reference solves its window with ceres::Solve (GLIO/src/Estimator.cpp:2424-2433); Ceres itself is not in /root/reference (version
1.14.0 by the bundled documentation), so the oracle's loop is a restatement.  The bundled documentation does print the solver's own progress tables
for three example programs (GraphGNSSLibV1.1/docs/source/nnls_tutorial.rst:139-143 helloworld, :378-394 + :411-432 Powell's function,
:508-523 curve fitting) -- iteration by iteration: cost (7 digits), cost change, gradient max norm, step norm, tr_ratio, trust-region radius.  These
are outputs of the real library with its default options (TRUST_REGION, LEVENBERG_MARQUARDT, Jacobi scaling, DENSE_QR).  tests/np_ceres.py -- the
numpy restatement that tests/test_oracle_tr_pins.py holds oracle/orc_solver.c and oracle/orc_batch2.c against, iteration by iteration -- must
reproduce them: that pins the loop's semantics (Jacobi scaling, the LM diagonal and its clamp, the step acceptance ratio, both radius rules, the
gradient-tolerance exit) on numbers nobody in this repository produced.  The curve-fitting program's data file is not in the documentation, but its
table is five REJECTED steps followed by accepted ones: the radius column alone is a known answer for the rejection rule (radius / 2, / 4, / 8, ...)
and for the acceptance rule radius / max(1/3, 1 - (2 rho - 1)^3) with the printed rho."""
import numpy as np

import np_ceres as nc


def _run(residuals, x0, **kw):
    def evaluate(x):
        r, J = residuals(x)
        return 0.5 * float(r @ r), J.T @ J, J.T @ r
    trace = []
    x, summ, _ = nc.minimize(np.array(x0, float), evaluate, lambda x, d: x + d, lambda x: x, nc.Options(strategy="lm", **kw), trace=trace)
    return x, summ, trace


def _sig(value, want, digits):
    """`value` printed with `digits` significant digits is `want` (a table entry), allowing one unit in the last printed place"""
    if want == 0.0:
        return abs(value) < 1e-300
    return abs(value - want) <= 1.0 * 10.0 ** (np.floor(np.log10(abs(want))) - (digits - 1)) * 1.0 + 1e-300


def test_helloworld_table():
    """nnls_tutorial.rst:139-143: f(x) = 10 - x from x = 0.5 (the printed run: "x : 0.5 -> 10", initial cost 4.512500e+01)"""
    x, summ, tr = _run(lambda x: (np.array([10.0 - x[0]]), np.array([[-1.0]])), [0.5])
    table = [  # cost, cost_change, |gradient|, |step|, tr_ratio, tr_radius
        (4.511598e-07, 4.51e+01, 9.50e-04, 9.50e+00, 1.00e+00, 3.00e+04),
        (5.012552e-16, 4.51e-07, 3.17e-08, 9.50e-04, 1.00e+00, 9.00e+04)]
    assert abs(summ["initial_cost"] - 4.512500e+01) < 5e-5
    assert len(tr) >= 2
    for row, want in zip(tr, table):
        assert _sig(row["cost"], want[0], 7), (row, want)
        assert _sig(row["cost_change"], want[1], 3) and _sig(row["gradient"], want[2], 3) and _sig(row["step"], want[3], 3), (row, want)
        assert _sig(row["ratio"], want[4], 3) and _sig(row["radius"], want[5], 3), (row, want)
    assert abs(x[0] - 10.0) < 1e-6


def _powell(x):
    x1, x2, x3, x4 = x
    r = np.array([x1 + 10.0 * x2, np.sqrt(5.0) * (x3 - x4), (x2 - 2.0 * x3) ** 2, np.sqrt(10.0) * (x1 - x4) ** 2])
    J = np.array([[1.0, 10.0, 0.0, 0.0],
                  [0.0, 0.0, np.sqrt(5.0), -np.sqrt(5.0)],
                  [0.0, 2.0 * (x2 - 2.0 * x3), -4.0 * (x2 - 2.0 * x3), 0.0],
                  [2.0 * np.sqrt(10.0) * (x1 - x4), 0.0, 0.0, -2.0 * np.sqrt(10.0) * (x1 - x4)]])
    return r, J


def test_powell_table():
    """nnls_tutorial.rst:378-394 (table), :411-432 (report): Powell's function from (3, -1, 0, 1); 14 iterations, all successful, termination by the
    gradient tolerance at max norm 3.642190e-11, final cost 1.791438e-14, final x = (0.000292189, -2.92189e-05, 4.79511e-05, 4.79511e-05)"""
    x, summ, tr = _run(_powell, [3.0, -1.0, 0.0, 1.0])
    table = [
        (5.036190e+00, 1.02e+02, 2.00e+01, 2.16e+00, 9.53e-01, 3.00e+04),
        (3.148168e-01, 4.72e+00, 2.50e+00, 6.23e-01, 9.37e-01, 9.00e+04),
        (1.967760e-02, 2.95e-01, 3.13e-01, 3.08e-01, 9.37e-01, 2.70e+05),
        (1.229900e-03, 1.84e-02, 3.91e-02, 1.54e-01, 9.37e-01, 8.10e+05),
        (7.687123e-05, 1.15e-03, 4.89e-03, 7.69e-02, 9.37e-01, 2.43e+06),
        (4.804625e-06, 7.21e-05, 6.11e-04, 3.85e-02, 9.37e-01, 7.29e+06),
        (3.003028e-07, 4.50e-06, 7.64e-05, 1.92e-02, 9.37e-01, 2.19e+07),
        (1.877006e-08, 2.82e-07, 9.54e-06, 9.62e-03, 9.37e-01, 6.56e+07),
        (1.173223e-09, 1.76e-08, 1.19e-06, 4.81e-03, 9.37e-01, 1.97e+08),
        (7.333425e-11, 1.10e-09, 1.49e-07, 2.40e-03, 9.37e-01, 5.90e+08),
        (4.584044e-12, 6.88e-11, 1.86e-08, 1.20e-03, 9.37e-01, 1.77e+09),
        (2.865573e-13, 4.30e-12, 2.33e-09, 6.02e-04, 9.37e-01, 5.31e+09),
        (1.791438e-14, 2.69e-13, 2.91e-10, 3.01e-04, 9.37e-01, 1.59e+10)]
    assert abs(summ["initial_cost"] - 1.075000e+02) < 5e-5
    assert len(tr) == 14 and summ["iterations"] == 14 and summ["successful_steps"] == 14
    for k, (row, want) in enumerate(zip(tr, table)):
        assert _sig(row["cost"], want[0], 7), (k, row, want)
        assert _sig(row["cost_change"], want[1], 3) and _sig(row["gradient"], want[2], 3) and _sig(row["step"], want[3], 3), (k, row, want)
        assert _sig(row["ratio"], want[4], 3) and _sig(row["radius"], want[5], 3), (k, row, want)
    assert summ["termination"] == nc.GRADIENT_TOL
    assert _sig(tr[13]["gradient"], 3.642190e-11, 6)
    # The report under the table prints "Final" cost 1.791438e-14 and x = (0.000292189, -2.92189e-05, 4.79511e-05, 4.79511e-05): the state of the
    # table's LAST PRINTED row (13 steps), while its termination line quotes the gradient after one more step (3.64e-11 = row 13's 2.91e-10 / 8: the
    # iterate halves per step and the gradient is cubic in it).  Both are reproduced: the iterate after 13 steps and the gradient after 14.
    for v, want in zip(tr[12]["x"], (0.000292189, -2.92189e-05, 4.79511e-05, 4.79511e-05)):
        assert _sig(v, want, 5), (tr[12]["x"],)


def test_curve_fitting_radius_column():
    """nnls_tutorial.rst:508-523: the radius column of a run that starts with five rejected steps.  Rejection: radius /= decrease factor, which doubles
    (2, 4, 8, 16, 32); acceptance with ratio rho: radius /= max(1/3, 1 - (2 rho - 1)^3), decrease factor back to 2."""
    ratios = [-1.87e+01, -1.86e+01, -1.85e+01, -1.70e+01, -6.32e+00, 1.37e+00, 1.10e+00, 1.03e+00, 9.94e-01, 9.89e-01, 9.97e-01, 1.00e+00, 1.00e+00]
    radii = [5.00e+03, 1.25e+03, 1.56e+02, 9.77e+00, 3.05e-01, 9.16e-01, 2.75e+00, 8.24e+00, 2.47e+01, 7.42e+01, 2.22e+02, 6.67e+02, 2.00e+03]
    lm = nc.LevenbergMarquardt(nc.Options(strategy="lm"))
    assert lm.radius == 1e4
    for rho, want in zip(ratios, radii):
        if rho > 1e-3:
            lm.accepted(rho)
        else:
            lm.rejected()
        assert _sig(lm.radius, want, 3), (rho, lm.radius, want)
