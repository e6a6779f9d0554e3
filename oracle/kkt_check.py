"""Independent KKT certificate for the reference's NLP.  TEST INFRASTRUCTURE ONLY (numpy).

The NLP is written here a second time, straight from the reference's text and in its flat ordering
(z = [vec(U); vec(X)], g rows in the order they are appended), with NO hand-written derivative: gradients and the
constraint Jacobian come from complex-step differentiation of these functions.  A solver result (z, lam_g, lam_x)
is then judged by the first-order optimality conditions in IPOPT's sign convention
    grad f(z) + J_g(z)^T lam_g + lam_x = 0,   lbg <= g(z) <= ubg,   lbx <= z <= ubx,
    lam > 0 only at an upper bound, lam < 0 only at a lower bound.
This depends on neither the oracle's nor the kernel's algebra, so it certifies both.

References (CMOM = CasaDi_MPC_Optimize_Multishoot/):
  model           CMOM/MPC_CBF_optimize_kin.py:153-156
  objective       CMOM/MPC_CBF_optimize_kin.py:168-205
  rows            CMOM/MPC_CBF_optimize_kin.py:190-191,207-216,236-248 ; _kin_pre.py:236-253
  bounds          CMOM/MPC_CBF_optimize_kin.py:84-134
"""
import numpy as np


class KinNlp:
    """The kinematic NLP for one instance.  obs: (n_obs,6) static or (n_obs,N+1,6) predicted; rows [x,y,th,v,l,w]."""

    def __init__(self, N, T, x0, xs, obs=None, Q=(1e1, 1e5, 3e5, 1e4), R=(1e4, 1e4), DR=(1e5, 1e2), veh_l=2.6,
                 veh_L=4.8, veh_W=1.8, safe_disl=1.0, safe_disw=0.5, df_lim=35 * np.pi / 180, a_lim=3.0, y_lim=(-1.0, 5.0),
                 v_lim=(0.0, 40.0), ddf_lim=5 * np.pi / 180, obs_mode="keepout", gamma=1.0, u_last=(0.0, 0.0), integrator="euler"):
        self.N, self.T = N, T
        self.integrator = integrator          # "euler": X+ = X + T f (kin.py:207); "rk4": classical Runge-Kutta step, control held (MPCB_INT_RK4)
        self.x0 = np.asarray(x0, float).reshape(4); self.xs = np.asarray(xs, float).reshape(4)
        self.Q, self.R, self.DR = np.asarray(Q, float), np.asarray(R, float), np.asarray(DR, float)
        self.veh_l = veh_l
        self.u_last = np.asarray(u_last, float)
        self.obs_mode, self.gamma = obs_mode, gamma
        if obs is None or np.size(obs) == 0:
            self.obs = np.zeros((0, N + 1, 6))
        else:
            obs = np.asarray(obs, float)
            if obs.ndim == 2:
                obs = np.repeat(obs[:, None, :], N + 1, axis=1)
            self.obs = obs
        self.n_obs = self.obs.shape[0]
        self.sx = veh_L / 2 + self.obs[:, :, 4] / 2 + safe_disl      # kin.py:242
        self.sy = veh_W / 2 + self.obs[:, :, 5] / 2 + safe_disw      # kin.py:243
        self.nz = 2 * N + 4 * (N + 1)
        self.ng = 4 * (N + 1) + (N - 1) + N * self.n_obs
        # bounds (kin.py:84-134)
        self.lbx = np.concatenate([np.tile([-df_lim, -a_lim], N), np.tile([-np.inf, y_lim[0], -np.inf, v_lim[0]], N + 1)])
        self.ubx = np.concatenate([np.tile([df_lim, a_lim], N), np.tile([np.inf, y_lim[1], np.inf, v_lim[1]], N + 1)])
        self.lbg = np.concatenate([np.zeros(4 * (N + 1)), np.full(N - 1, -ddf_lim * T), np.zeros(N * self.n_obs)])
        self.ubg = np.concatenate([np.zeros(4 * (N + 1)), np.full(N - 1, ddf_lim * T), np.full(N * self.n_obs, np.inf)])

    def split(self, z):
        N = self.N
        U = z[:2 * N].reshape(N, 2)          # U[i] = [df_i, ax_i]
        X = z[2 * N:].reshape(N + 1, 4)      # X[k] = [x, y, phi, vx]
        return U, X

    def rhs(self, X, U):
        return np.stack([X[:, 3] * np.cos(X[:, 2]), X[:, 3] * np.sin(X[:, 2]), X[:, 3] * np.tan(U[:, 0]) / self.veh_l, U[:, 1]], axis=1)

    def step(self, X, U):
        """One shooting step for every stage at once: X [N,4], U [N,2] -> X+ [N,4]."""
        T = self.T
        if self.integrator == "euler":
            return X + T * self.rhs(X, U)
        k1 = self.rhs(X, U); k2 = self.rhs(X + 0.5 * T * k1, U); k3 = self.rhs(X + 0.5 * T * k2, U); k4 = self.rhs(X + T * k3, U)
        return X + (T / 6.0) * (k1 + 2 * k2 + 2 * k3 + k4)

    def f(self, z):
        U, X = self.split(z)
        e = X[:-1] - self.xs
        Up = np.vstack([self.u_last.astype(z.dtype)[None, :], U[:-1]])
        return (e * e * self.Q).sum() + (U * U * self.R).sum() + ((U - Up) ** 2 * self.DR).sum()

    def h(self, X, k, step):
        o = self.obs[:, step, :]
        return (X[k, 0] - o[:, 0]) ** 2 / self.sx[:, step] ** 2 + (X[k, 1] - o[:, 1]) ** 2 / self.sy[:, step] ** 2 - 1.0

    def g(self, z):
        U, X = self.split(z)
        N = self.N
        rows = [X[0] - self.x0]
        nxt = self.step(X[:-1], U)
        rows.append((X[1:] - nxt).reshape(-1))
        rows.append(U[1:, 0] - U[:-1, 0])
        for i in range(N):
            if self.n_obs == 0:
                break
            hi = self.h(X, i, i)
            if self.obs_mode == "keepout":
                rows.append(hi)                                           # kin.py:247
            else:
                rows.append(self.gamma * hi + (self.h(X, i + 1, i) - hi))   # kin.py:245-248
        return np.concatenate([np.asarray(r).reshape(-1) for r in rows])

    # complex-step derivatives ------------------------------------------------------------------------------------
    def grad_f(self, z, h=1e-30):
        g = np.empty(self.nz)
        zc = z.astype(complex)
        for i in range(self.nz):
            zc[i] += 1j * h
            g[i] = self.f(zc).imag / h
            zc[i] = z[i]
        return g

    def jac_g_dense(self, z, h=1e-30):
        """One complex step per variable: the plain definition (nz evaluations of g)."""
        J = np.empty((self.ng, self.nz))
        zc = z.astype(complex)
        for i in range(self.nz):
            zc[i] += 1j * h
            J[:, i] = self.g(zc).imag / h
            zc[i] = z[i]
        return J

    _patterns = {}

    def _pattern_key(self):
        return (type(self).__name__, self.N, self.n_obs, getattr(self, "obs_mode", ""), getattr(self, "integrator", "euler"))

    def jac_g(self, z, h=1e-30):
        """The same Jacobian from a handful of evaluations: columns that share no row (variables three or more stages apart) are
        perturbed together (Curtis-Powell-Reid colouring of the structural pattern, which is found once per NLP structure from two
        dense Jacobians at random points and cached).  tests/test_oracle.py checks it against jac_g_dense."""
        key = self._pattern_key()
        if key not in KinNlp._patterns:
            rng = np.random.default_rng(0)
            P = np.zeros((self.ng, self.nz), bool)
            for _ in range(2):
                zr = z + rng.uniform(0.05, 0.5, self.nz) * rng.choice([-1.0, 1.0], self.nz)
                P |= self.jac_g_dense(zr) != 0.0
            groups = []                                  # greedy: a column joins the first group none of whose rows it touches
            rows_of = []
            for j in range(self.nz):
                rj = P[:, j]
                for gi, used in enumerate(rows_of):
                    if not (used & rj).any():
                        groups[gi].append(j); used |= rj
                        break
                else:
                    groups.append([j]); rows_of.append(rj.copy())
            KinNlp._patterns[key] = (P, [np.array(g_) for g_ in groups])
        P, groups = KinNlp._patterns[key]
        J = np.zeros((self.ng, self.nz))
        zc = z.astype(complex)
        for cols in groups:
            zc[cols] += 1j * h
            d = self.g(zc).imag / h
            zc[cols] = z[cols]
            for j in cols:
                r = P[:, j]
                J[r, j] = d[r]
        return J


def certificate(nlp, z, lam_g, lam_x, act_tol=1e-6):
    """Returns dict of unscaled KKT residuals."""
    z = np.asarray(z, float); lam_g = np.asarray(lam_g, float); lam_x = np.asarray(lam_x, float)
    gv = nlp.g(z)
    stat = nlp.grad_f(z) + nlp.jac_g(z).T @ lam_g + lam_x
    viol_g = np.maximum(0, np.maximum(nlp.lbg - gv, gv - nlp.ubg)).max()
    viol_x = np.maximum(0, np.maximum(nlp.lbx - z, z - nlp.ubx)).max()

    def compl(lam, v, lo, hi):
        eq = lo == hi
        dist_hi = np.where(np.isfinite(hi), hi - v, np.inf)
        dist_lo = np.where(np.isfinite(lo), v - lo, np.inf)
        with np.errstate(invalid="ignore"):
            c = np.where(lam > 0, lam * np.maximum(dist_hi, 0), -lam * np.maximum(dist_lo, 0))
        c = np.where(eq, 0.0, c)
        wrong_sign = np.where(~eq & (lam > 0) & ~np.isfinite(hi), lam, 0.0) + np.where(~eq & (lam < 0) & ~np.isfinite(lo), -lam, 0.0)
        return np.nan_to_num(c, posinf=0.0).max(), wrong_sign.max()

    cg, sg = compl(lam_g, gv, nlp.lbg, nlp.ubg)
    cx, sx = compl(lam_x, z, nlp.lbx, nlp.ubx)
    return dict(stationarity=np.abs(stat).max(), feas_g=viol_g, feas_x=viol_x, compl=max(cg, cx), sign=max(sg, sx),
                f=float(nlp.f(z)), lam_scale=max(1.0, np.abs(lam_g).max(), np.abs(lam_x).max()))


class DynNlp:
    """The dynamic-bicycle NLP (CMOM/MPC_CBF_optimize_dyn.py) for one instance, g rows and bounds ALIGNED (the
    reference's own lbg/ubg are interleaved one stage off its g).  obs: (n_obs, >=2) static centres.
    The obstacle row is kept in the reference's form sqrt(h) >= 1 so that this certificate also shows that the
    solver's h >= 1 formulation reaches a KKT point of the reference's row (multipliers scale by 2 sqrt(h))."""

    def __init__(self, N, T, x0, xs, obs, Q=(10, 1e5, 1e3, 1e3, 1, 1), R=(1e3, 1e3), DR=(5e3, 5e2), m=1575.0, lf=1.2, lr=1.6, Iz=2875.0,
                 aopt_f=0.3490658503988659, aopt_r=0.19198621771937624, Cf0=-50000.0, Cr0=-50000.0, sx=4.0, sy=1.0):
        self.N, self.T = N, T
        self.x0 = np.asarray(x0, float).reshape(6); self.xs = np.asarray(xs, float).reshape(6)
        self.Q, self.R, self.DR = np.asarray(Q, float), np.asarray(R, float), np.asarray(DR, float)
        self.m, self.lf, self.lr, self.Iz, self.af, self.ar = m, lf, lr, Iz, aopt_f, aopt_r
        self.Fyf, self.Fyr = Cf0 * aopt_f / 2, Cr0 * aopt_r / 2                       # dyn.py:55-56
        self.obs = np.asarray(obs, float).reshape(-1, np.shape(obs)[-1])[:, :2]
        self.n_obs, self.sx, self.sy = len(self.obs), sx, sy
        self.nz = 2 * N + 6 * (N + 1)
        deg = np.pi / 180
        self.lbx = np.concatenate([np.tile([-35 * deg, -3.0], N), np.tile([-np.inf, -1.0, -np.inf, 0.0, -5.0, -np.inf], N + 1)])
        self.ubx = np.concatenate([np.tile([35 * deg, 3.0], N), np.tile([np.inf, 5.0, np.inf, 40.0, 5.0, np.inf], N + 1)])
        lbg, ubg = [0.0] * 6, [0.0] * 6
        for i in range(N):
            lbg += [0.0] * 6; ubg += [0.0] * 6
            if i > 0:
                lbg += [-5 * deg * T, -3.0 * T]; ubg += [5 * deg * T, 1.5 * T]
        lbg += [1.0] * ((N + 1) * self.n_obs); ubg += [np.inf] * ((N + 1) * self.n_obs)
        self.lbg, self.ubg = np.array(lbg), np.array(ubg)
        self.ng = len(lbg)

    def split(self, z):
        N = self.N
        return z[:2 * N].reshape(N, 2), z[2 * N:].reshape(N + 1, 6)

    def rhs(self, X, U):                                                               # dyn.py:156-170
        phi, vx, vy, r, df, ax = X[:, 2], X[:, 3], X[:, 4], X[:, 5], U[:, 0], U[:, 1]
        alf = df - (vy + self.lf * r) / vx
        alr = -(vy - self.lr * r) / vx
        Cf = self.Fyf * 2 * self.af / (self.af ** 2 + alf ** 2)
        Cr = self.Fyr * 2 * self.ar / (self.ar ** 2 + alr ** 2)
        Fcf, Fcr = -Cf * alf, -Cr * alr
        return np.stack([vx * np.cos(phi) - vy * np.sin(phi), vx * np.sin(phi) + vy * np.cos(phi), r, ax + r * vy,
                         -r * vx + 2 / self.m * (Fcf * np.cos(df) + Fcr), 2 / self.Iz * (self.lf * Fcf - self.lr * Fcr)], axis=1)

    def f(self, z):                                                                    # dyn.py:212-225
        U, X = self.split(z)
        e = X[:-1] - self.xs
        dU = U[1:] - U[:-1]
        return (e * e * self.Q).sum() + (U * U * self.R).sum() + (dU * dU * self.DR).sum()

    def g(self, z):                                                                    # dyn.py:215-243
        U, X = self.split(z)
        rows = [X[0] - self.x0]
        nxt = X[:-1] + self.T * self.rhs(X[:-1], U)
        for i in range(self.N):
            rows.append(X[i + 1] - nxt[i])
            if i > 0:
                rows.append(U[i] - U[i - 1])
        for k in range(self.N + 1):
            for j in range(self.n_obs):
                rows.append(np.sqrt((X[k, 0] - self.obs[j, 0]) ** 2 / self.sx ** 2 + (X[k, 1] - self.obs[j, 1]) ** 2 / self.sy ** 2 - 1)[None])
        return np.concatenate([np.asarray(r).reshape(-1) for r in rows])

    grad_f = KinNlp.grad_f
    jac_g_dense = KinNlp.jac_g_dense
    _pattern_key = KinNlp._pattern_key
    jac_g = KinNlp.jac_g

    def convert_obstacle_multipliers(self, z, lam_g):
        """The solver's row is h >= 1 with multiplier lam_h; the reference's is sqrt(h) >= 1: lam_sqrt = 2 sqrt(h) lam_h."""
        out = np.array(lam_g, float)
        gv = self.g(np.asarray(z, float))
        n0 = self.ng - (self.N + 1) * self.n_obs
        out[n0:] = out[n0:] * 2 * gv[n0:]
        return out
