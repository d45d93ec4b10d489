"""An INDEPENDENT numpy restatement of the Ceres 1.14 trust-region loop, used only to pin the oracle's solver semantics
(tests/test_oracle_tr_pins.py): oracle/orc_solver.c and oracle/orc_batch2.c were written as C loops over raw arrays, this file
is a second reading of the same algorithm in numpy's vocabulary (np.linalg.cholesky / qr / roots), from
  * the bundled docs GraphGNSSLibV1.1/docs/source/nnls_solving.rst:83-260 (trust-region loop: rho = actual / model decrease,
    radius update; Levenberg-Marquardt: (J^T J + D^T D / mu) step; dogleg: Gauss-Newton + Cauchy point, TRADITIONAL_DOGLEG and
    SUBSPACE_DOGLEG; non-monotonic steps: 5 consecutive, reference iterate) and :1056-1188 (option defaults), and
  * the published Ceres 1.14 sources for what the docs leave open (trust_region_minimizer.cc, dogleg_strategy.cc,
    levenberg_marquardt_strategy.cc, trust_region_step_evaluator.cc, polynomial.cc): Jacobi scaling 1 / (1 + sqrt(diag)), the
    diagonal clamp [1e-6, 1e32], mu in [1e-8, 1] x 10 on a failed factorisation, radius x 0.5 / max(radius, 3 |step|) at
    0.25 / 0.75, LM radius / max(1/3, 1 - (2 rho - 1)^3) and the doubling decrease factor, five invalid steps = failure, the
    tolerance tests before the acceptance test, the user's parameters following the minimum-cost iterate only.
The problem is handed over as callbacks, so the same class runs the sliding-window problem and the batch problem:
  evaluate(x) -> (cost, H, g)   dense J^T J and J^T r in the local parameterisation, or None when the evaluation fails
  plus(x, delta) -> x'          the local parameterisation
  flat(x) -> 1-D array          all parameter blocks as Ceres sees them (norms of x and of x - x')
"""
import numpy as np

NO_CONVERGENCE, FUNCTION_TOL, PARAMETER_TOL, GRADIENT_TOL, MIN_RADIUS, FAILURE = range(6)


class Options:
    def __init__(self, **kw):
        self.max_iterations = 50
        self.strategy = "dogleg"                 # "dogleg" | "lm"
        self.dogleg = "traditional"              # "traditional" | "subspace"
        self.nonmonotonic = False
        self.max_consecutive_nonmonotonic_steps = 5
        self.jacobi_scaling = True
        self.initial_radius, self.max_radius, self.min_radius = 1e4, 1e16, 1e-32
        self.min_relative_decrease = 1e-3
        self.function_tolerance, self.gradient_tolerance, self.parameter_tolerance = 1e-6, 1e-10, 1e-8
        self.__dict__.update(kw)


class StepEvaluator:
    """trust_region_step_evaluator.cc"""

    def __init__(self, initial_cost, max_nonmonotonic):
        self.max_n = max_nonmonotonic
        self.minimum = self.current = self.reference = self.candidate = initial_cost
        self.acc_reference = self.acc_candidate = 0.0
        self.n = 0

    def quality(self, cost, model_change):
        rel = (self.current - cost) / model_change
        hist = (self.reference - cost) / (self.acc_reference + model_change)
        return max(rel, hist) if self.max_n > 0 else rel

    def accepted(self, cost, model_change):
        self.current = cost
        self.acc_candidate += model_change
        self.acc_reference += model_change
        if self.current < self.minimum:
            self.minimum, self.n, self.candidate, self.acc_candidate = self.current, 0, self.current, 0.0
        else:
            self.n += 1
            if self.current > self.candidate:
                self.candidate, self.acc_candidate = self.current, 0.0
        if self.n == self.max_n:
            self.reference, self.acc_reference = self.candidate, self.acc_candidate


def _chol_solve(A, b):
    try:
        L = np.linalg.cholesky(A)
    except np.linalg.LinAlgError:
        return None
    y = np.linalg.solve(L, b)
    x = np.linalg.solve(L.T, y)
    return x if np.all(np.isfinite(x)) else None


class Dogleg:
    def __init__(self, o):
        self.o = o
        self.radius, self.mu, self.reuse = o.initial_radius, 1e-8, False
        self.step_norm = 0.0

    def compute(self, Hs, gs):
        """returns the step in the (Jacobi-)scaled variables or None (linear solver failure)"""
        o = self.o
        if not self.reuse:
            self.D = np.sqrt(np.clip(np.diag(Hs), 1e-6, 1e32))
            self.grad = gs / self.D
            u = self.grad / self.D
            self.alpha = (self.grad @ self.grad) / (u @ (Hs @ u))
            y = None
            while self.mu < 1.0:
                y = _chol_solve(Hs + np.diag(self.mu * self.D * self.D), gs)
                if y is not None:
                    break
                self.mu *= 10.0
            if y is None:
                return None
            self.gn = -self.D * y
            if o.dogleg == "subspace" and not self._subspace_model(Hs):
                return None
            self.reuse = True
        step = self._subspace_step() if o.dogleg == "subspace" else self._traditional_step()
        return step / self.D

    def _traditional_step(self):
        g, gn, r, a = self.grad, self.gn, self.radius, self.alpha
        if np.linalg.norm(gn) <= r:
            self.step_norm = np.linalg.norm(gn)
            return gn.copy()
        if np.linalg.norm(g) * a >= r:
            self.step_norm = r
            return -(r / np.linalg.norm(g)) * g
        b_dot_a = -a * (g @ gn)
        a2 = a * a * (g @ g)
        bma2 = gn @ gn - 2 * b_dot_a + a2
        c = b_dot_a - a2
        d = np.sqrt(c * c + bma2 * (r * r - a2))
        beta = (d - c) / bma2 if c <= 0 else (r * r - a2) / (d + c)
        s = (-a * (1 - beta)) * g + beta * gn
        self.step_norm = np.linalg.norm(s)
        return s

    def _subspace_model(self, Hs):
        A = np.stack([self.grad, self.gn], 1)
        if np.linalg.norm(A[:, 1]) > np.linalg.norm(A[:, 0]):           # column pivoting
            A = A[:, ::-1]
        Q, R = np.linalg.qr(A)
        piv = np.abs(np.diag(R))
        rank = int((piv > piv.max() * 2 * np.finfo(float).eps).sum())
        if rank == 0:
            return False
        self.one_dim = rank == 1
        if self.one_dim:
            return True
        self.U = Q
        self.sg = Q.T @ self.grad
        V = Q / self.D[:, None]
        self.sB = V.T @ (Hs @ V)
        return True

    def _subspace_step(self):
        r = self.radius
        if np.linalg.norm(self.gn) <= r:
            self.step_norm = np.linalg.norm(self.gn)
            return self.gn.copy()
        if self.one_dim:
            self.step_norm = r
            return -(r / np.linalg.norm(self.grad)) * self.grad
        B, g = self.sB, self.sg
        detB, trB, r2 = np.linalg.det(B), np.trace(B), r * r
        adj = np.array([[B[1, 1], -B[0, 1]], [-B[1, 0], B[0, 0]]])
        poly = [r2, 2 * r2 * trB, r2 * (trB * trB + 2 * detB) - g @ g, -2 * (g @ adj @ g - r2 * detB * trB), r2 * detB * detB - (adj @ g) @ (adj @ g)]
        roots = np.roots(poly)
        best, xbest = np.inf, None
        for y in np.real(roots):
            x = -np.linalg.solve(B + y * np.eye(2), g)
            nx = np.linalg.norm(x)
            if nx > 0:
                xs = r / nx * x
                f = 0.5 * xs @ B @ xs + g @ xs
                if f < best:
                    best, xbest = f, x
        if xbest is None:
            return self._traditional_step()
        self.step_norm = r
        return self.U @ xbest

    def accepted(self, q):
        if q < 0.25:
            self.radius *= 0.5
        if q > 0.75:
            self.radius = max(self.radius, 3.0 * self.step_norm)
        self.mu = max(1e-8, 2.0 * self.mu / 10.0)
        self.reuse = False

    def rejected(self):
        self.radius *= 0.5
        self.reuse = True

    def invalid(self):
        self.mu *= 10.0
        self.reuse = False


class LevenbergMarquardt:
    def __init__(self, o):
        self.o = o
        self.radius, self.decrease = o.initial_radius, 2.0

    def compute(self, Hs, gs):
        D2 = np.clip(np.diag(Hs), 1e-6, 1e32) / self.radius
        y = _chol_solve(Hs + np.diag(D2), gs)
        return None if y is None else -y

    def accepted(self, q):
        self.radius = min(self.o.max_radius, self.radius / max(1.0 / 3.0, 1.0 - (2.0 * q - 1.0) ** 3))
        self.decrease = 2.0

    def rejected(self):
        self.radius /= self.decrease
        self.decrease *= 2.0

    invalid = rejected


def minimize(x0, evaluate, plus, flat, o, trace=None):
    """Returns (x_user, summary dict, history): history rows (candidate cost, radius at the step, |x - candidate|) per iteration.
    `trace` (a list): one dict per iteration with the columns of Ceres' progress table -- cost, cost_change, gradient max norm, step norm, tr_ratio
    and the radius AFTER the iteration (tests/test_ceres_docs_kat.py holds them against the tables printed in the bundled Ceres documentation)."""
    x = x0
    ev = evaluate(x)
    if ev is None:
        return x0, dict(termination=FAILURE, iterations=0), []
    cost, H, g = ev
    scale = 1.0 / (1.0 + np.sqrt(np.diag(H))) if o.jacobi_scaling else np.ones(len(g))
    strat = LevenbergMarquardt(o) if o.strategy == "lm" else Dogleg(o)
    evalr = StepEvaluator(cost, o.max_consecutive_nonmonotonic_steps if o.nonmonotonic else 0)
    x_user, user_min = x, cost
    it, invalid, ok_steps = 0, 0, 0
    term = NO_CONVERGENCE
    hist = []
    initial = cost
    while True:
        gmax = np.abs(flat(x) - flat(plus(x, -g))).max()
        if it >= o.max_iterations:
            term = NO_CONVERGENCE
            break
        if gmax <= o.gradient_tolerance:
            term = GRADIENT_TOL
            break
        if strat.radius <= o.min_radius:
            term = MIN_RADIUS
            break
        it += 1
        Hs, gs = scale[:, None] * H * scale[None, :], scale * g
        radius_used = strat.radius
        step = strat.compute(Hs, gs)
        mcc = -(gs @ step + 0.5 * step @ (Hs @ step)) if step is not None else -1.0
        if step is None or not mcc > 0.0:
            invalid += 1
            if invalid >= 5:
                term = FAILURE
                break
            strat.invalid()
            hist.append((evalr.current, radius_used, 0.0))
            continue
        invalid = 0
        cand = plus(x, step * scale)
        ev = evaluate(cand)
        if ev is None:
            strat.rejected()
            continue
        ccost, Hc, gc = ev
        dx = np.linalg.norm(flat(x) - flat(cand))
        hist.append((ccost, radius_used, dx))
        if dx <= o.parameter_tolerance * (np.linalg.norm(flat(x)) + o.parameter_tolerance):
            term = PARAMETER_TOL
            break
        if abs(evalr.current - ccost) <= o.function_tolerance * evalr.current:
            term = FUNCTION_TOL
            break
        q = evalr.quality(ccost, mcc)
        cost_before = evalr.current
        if q > o.min_relative_decrease:
            x, H, g = cand, Hc, gc
            ok_steps += 1
            strat.accepted(q)
            evalr.accepted(ccost, mcc)
            if ccost < user_min:
                user_min, x_user = ccost, x
            if trace is not None:
                trace.append(dict(cost=ccost, cost_change=cost_before - ccost, gradient=float(np.abs(flat(x) - flat(plus(x, -g))).max()), step=dx, ratio=q, radius=strat.radius,
                                  x=np.array(flat(x), float).copy()))
        else:
            strat.rejected()
            if trace is not None:
                trace.append(dict(cost=cost_before, cost_change=cost_before - ccost, gradient=0.0, step=dx, ratio=q, radius=strat.radius))
    return x_user, dict(termination=term, iterations=it, successful_steps=ok_steps, initial_cost=initial, final_cost=user_min, final_radius=strat.radius), hist
