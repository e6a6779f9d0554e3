"""Independent SOLVER for the reference's NLP: SciPy's SLSQP (a sequential-quadratic-programming code that shares neither
algorithm nor code with the interior-point solver of this repository) on the NLP as oracle/kkt_check.py restates it from the
reference's text, derivatives by complex step.  TEST INFRASTRUCTURE ONLY.

Started from the same cold start as the solver (X rolled out from x0 with zero controls), it has to arrive at the same local
minimiser.  This is not the reference's own CasADi+IPOPT (not installable here: parity stays UNPINNED), but it is an
off-the-shelf NLP solver that nobody here wrote."""
import numpy as np
from scipy.optimize import minimize


def cold_start(nlp, T, rhs_zero_controls):
    """z0 of the solvers with cfg.init_rollout = 1: U = 0, X rolled out from x0 by explicit Euler steps."""
    N = nlp.N
    nx = (nlp.nz - 2 * N) // (N + 1)
    X = np.zeros((N + 1, nx)); X[0] = nlp.x0
    for k in range(N):
        X[k + 1] = X[k] + T * rhs_zero_controls(X[k])
    z0 = np.zeros(nlp.nz); z0[2 * N:] = X.reshape(-1)
    return z0


def solve_slsqp(nlp, z0, scale=1e-4, maxiter=1000):
    """minimise scale * f(z) s.t. lbg <= g(z) <= ubg, lbx <= z <= ubx with SLSQP.  `scale` only conditions SLSQP's line search
    (f is ~1e8 with the reference's weights); the minimiser does not depend on it.  Returns (z, f(z), scipy result)."""
    eq = nlp.lbg == nlp.ubg
    lo = ~eq & np.isfinite(nlp.lbg); hi = ~eq & np.isfinite(nlp.ubg)

    def ineq(z):
        g = nlp.g(z)
        return np.concatenate([(g - nlp.lbg)[lo], (nlp.ubg - g)[hi]])

    def ineq_jac(z):
        J = nlp.jac_g(z)
        return np.concatenate([J[lo], -J[hi]])

    cons = [dict(type="eq", fun=lambda z: (nlp.g(z) - nlp.lbg)[eq], jac=lambda z: nlp.jac_g(z)[eq]), dict(type="ineq", fun=ineq, jac=ineq_jac)]
    lb = np.where(np.isfinite(nlp.lbx), nlp.lbx, -1e20); ub = np.where(np.isfinite(nlp.ubx), nlp.ubx, 1e20)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        s = minimize(lambda z: scale * nlp.f(z), z0, jac=lambda z: scale * nlp.grad_f(z), bounds=list(zip(lb, ub)), constraints=cons,
                     method="SLSQP", options=dict(maxiter=maxiter, ftol=1e-16))
    return s.x, float(nlp.f(s.x)), s


def min_violation_slsqp(nlp, z0, maxiter=400):
    """Independent audit of an "infeasible" verdict: SLSQP on the pure feasibility problem of `nlp`
        minimise sum(s)   s.t.  equality rows and two-sided rows of g as they are, one-sided rows  g_i(z) + s_i >= lbg_i,  s >= 0,  lbx <= z <= ubx
    (the one-sided rows are the obstacle rows: the only reverse-convex ones).  Returns (sum(s) at the solution, largest violation
    of ANY row or box at the z found, z).  A value <= 1e-8 means a feasible point exists and the verdict was wrong."""
    eq = nlp.lbg == nlp.ubg
    two = ~eq & np.isfinite(nlp.lbg) & np.isfinite(nlp.ubg)
    one = ~eq & np.isfinite(nlp.lbg) & ~np.isfinite(nlp.ubg)
    assert not (~eq & ~np.isfinite(nlp.lbg)).any()
    nz, ns = nlp.nz, int(one.sum())
    g0 = nlp.g(np.asarray(z0, float))
    s0 = np.maximum(0.0, (nlp.lbg - g0)[one]) + 1e-3
    y0 = np.concatenate([np.asarray(z0, float), s0])

    def parts(y):
        return y[:nz], y[nz:]

    def c_eq(y):
        z, _ = parts(y); return (nlp.g(z) - nlp.lbg)[eq]

    def j_eq(y):
        z, _ = parts(y); return np.hstack([nlp.jac_g(z)[eq], np.zeros((int(eq.sum()), ns))])

    def c_in(y):
        z, s = parts(y); g = nlp.g(z)
        return np.concatenate([(g - nlp.lbg)[two], (nlp.ubg - g)[two], (g - nlp.lbg)[one] + s])

    def j_in(y):
        z, _ = parts(y); J = nlp.jac_g(z)
        return np.vstack([np.hstack([J[two], np.zeros((int(two.sum()), ns))]), np.hstack([-J[two], np.zeros((int(two.sum()), ns))]),
                          np.hstack([J[one], np.eye(ns)])])

    lb = np.concatenate([np.where(np.isfinite(nlp.lbx), nlp.lbx, -1e20), np.zeros(ns)])
    ub = np.concatenate([np.where(np.isfinite(nlp.ubx), nlp.ubx, 1e20), np.full(ns, 1e20)])
    grad = np.concatenate([np.zeros(nz), np.ones(ns)])
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = minimize(lambda y: y[nz:].sum(), y0, jac=lambda y: grad, bounds=list(zip(lb, ub)),
                     constraints=[dict(type="eq", fun=c_eq, jac=j_eq), dict(type="ineq", fun=c_in, jac=j_in)], method="SLSQP",
                     options=dict(maxiter=maxiter, ftol=1e-14))
    z, s = parts(r.x)
    g = nlp.g(z)
    viol = max(np.maximum(0, np.maximum(nlp.lbg - g, g - nlp.ubg)).max(), np.maximum(0, np.maximum(nlp.lbx - z, z - nlp.ubx)).max())
    return float(s.sum()), float(viol), z
