"""Shared implementation of the three `MPC_optimize` drop-in classes.

The reference's classes (CasaDi_MPC_Optimize_Multishoot/MPC_CBF_optimize_{kin,kin_pre,dyn}.py) describe the NLP
symbolically with CasADi and hand it to IPOPT.  Here the same public surface — constructor reading
mpc_parameters.yaml, `initialize_constraints`, `optimize_problem` returning a callable `solver`, the model
function `f`, `num_states`, `num_controls`, `N_p`, `T_S` — sits on the HIP library through ctypes.  Nothing in
this module computes a solution on the CPU.
"""
import numpy as np

from . import _abi
from .dm import DM
from .helpers import load_config, find_params_file, horizon_steps, PARAMS_FILE
from .solver import BatchSolver, default_config, model_rhs

_RETURN_STATUS = {
    _abi.ST_SOLVED: "Solve_Succeeded",
    _abi.ST_MAXITER: "Maximum_Iterations_Exceeded",
    _abi.ST_LINESEARCH: "Line_Search_Failed_Restoration_Disabled",   # cfg.restoration = 0: no restoration was attempted
    _abi.ST_INFEASIBLE_X0: "Infeasible_Problem_Detected",              # x0 violates a row of node 0: no z can satisfy it
    _abi.ST_NUMERIC: "Error_In_Step_Computation",
    _abi.ST_INFEASIBLE: "Infeasible_Problem_Detected",                 # restoration converged to a point of LOCAL infeasibility
    _abi.ST_RESTO_FAILED: "Restoration_Failed",
}


class ModelFunction:
    """`mpc_solver.f(x, u)` -> DM (kin.py:159; called by shift_movement, main_cbf_kin_c_sim.py:17)."""

    def __init__(self, cfg):
        self._cfg = cfg

    def __call__(self, x, u):
        return DM(model_rhs(self._cfg, np.asarray(x, dtype=np.float64).reshape(-1), np.asarray(u, dtype=np.float64).reshape(-1)))


def nlp_constraints(cfg, z, x0, obs, obs_kind):
    """g(z) in the reference's row order (kin.py:190-247 / dyn.py:215-243), evaluated on the host for the result dict.
    Obstacle rows are reported as h (the dyn reference's row is sqrt(h))."""
    N, nx = cfg.N, cfg.nx()
    z = np.asarray(z, dtype=np.float64).reshape(-1)
    U = z[:2 * N].reshape(N, 2); X = z[2 * N:].reshape(N + 1, nx)
    f = np.stack([model_rhs(cfg, X[i], U[i]) for i in range(N)])
    dyn_rows = X[1:] - (X[:-1] + cfg.T * f)
    rate_cols = [c for c in range(2) if np.isfinite(cfg.du_lo[c]) or np.isfinite(cfg.du_hi[c])]
    rows = [X[0] - np.asarray(x0, dtype=np.float64).reshape(-1)]
    if cfg.rate_interleaved:
        for i in range(N):
            rows.append(dyn_rows[i])
            if i > 0:
                rows.append(np.array([U[i, c] - U[i - 1, c] for c in rate_cols]))
    else:
        rows.append(dyn_rows.reshape(-1))
        for c in rate_cols:
            rows.append(U[1:, c] - U[:-1, c])
    if cfg.n_obs:
        o = np.asarray(obs, dtype=np.float64).reshape(-1)
        last = N if cfg.obs_terminal else N - 1
        for i in range(last + 1):
            for j in range(cfg.n_obs):
                q = o[(j * (N + 1) + i) * 6:(j * (N + 1) + i) * 6 + 6] if obs_kind == _abi.OBSIN_PREDICTED else o[j * 6:j * 6 + 6]
                sx = cfg.obs_sx_fixed if cfg.obs_sx_fixed > 0 else cfg.ego_hl + q[4] / 2 + cfg.safe_disl
                sy = cfg.obs_sy_fixed if cfg.obs_sy_fixed > 0 else cfg.ego_hw + q[5] / 2 + cfg.safe_disw
                h = lambda node: (X[node, 0] - q[0]) ** 2 / sx ** 2 + (X[node, 1] - q[1]) ** 2 / sy ** 2 - 1.0   # noqa: E731
                if cfg.obs_mode == _abi.OBS_KEEPOUT:
                    rows.append(np.array([h(i)]))                                        # kin.py:247
                else:                                                                     # kin.py:245-248: gamma h_i + (h_next - h_i)
                    rows.append(np.array([h(i + 1) - (1.0 - cfg.gamma) * h(i)]))
    return np.concatenate([np.asarray(r, dtype=np.float64).reshape(-1) for r in rows])


class NlpSolver:
    """What `optimize_problem` returns: callable like the object `ca.nlpsol(...)` gives (kin.py:254)."""

    def __init__(self, owner, cfg, obs, obs_kind):
        self._owner = owner
        self._cfg = cfg
        self._obs = obs
        self._obs_kind = obs_kind
        self._stats = {}

    def __call__(self, x0=None, p=None, lbg=None, lbx=None, ubg=None, ubx=None, **_ignored):
        cfg = self._cfg
        nx = cfg.nx()
        bs = self._owner._batch_solver(cfg)
        bs.set_time_grid(self._owner.stage_lengths() if self._owner.time_grid_in_nlp else None)
        if lbx is not None and ubx is not None and lbg is not None and ubg is not None:
            bs.set_bounds(np.asarray(lbx, dtype=np.float64).reshape(-1), np.asarray(ubx, dtype=np.float64).reshape(-1),
                          np.asarray(lbg, dtype=np.float64).reshape(-1), np.asarray(ubg, dtype=np.float64).reshape(-1))
        pv = np.asarray(p, dtype=np.float64).reshape(-1)
        if pv.size != 2 * nx:
            raise ValueError("p must hold [x0; xs] (%d values)" % (2 * nx))
        z0 = None if x0 is None else np.asarray(x0, dtype=np.float64).reshape(1, -1)
        r = bs.solve_batch(pv[:nx].reshape(1, nx), pv[nx:].reshape(1, nx), self._obs, z0, multipliers=True)
        st = int(r["status"][0])
        self._stats = {"success": st == _abi.ST_SOLVED, "return_status": _RETURN_STATUS.get(st, "Unknown"),
                       "iter_count": int(r["iters"][0]), "status_code": st, "kkt": r["kkt"][0].copy()}
        self._owner.last_stats = self._stats
        return {"x": DM(r["z"][0]), "f": DM(r["obj"][0]), "lam_g": DM(r["lam_g"][0]), "lam_x": DM(r["lam_x"][0]),
                "lam_p": DM(np.zeros(2 * nx)), "g": DM(nlp_constraints(bs.cfg, r["z"][0], pv[:nx], self._obs, self._obs_kind))}

    def stats(self):
        return dict(self._stats)


class MpcBase:
    MODEL = _abi.MODEL_KIN

    def __init__(self, params_file=PARAMS_FILE):
        self.config = load_config(find_params_file(params_file))
        mp = self.config["mpc_params"]
        self.T_horizon = mp["horizon"]
        self.T_S = mp["T_S"]
        self.pre_time = mp["pre_time"]
        self.T_L = mp["T_L"]
        self.t_ratio = mp["t_ratio"]
        self.is_variable_time = mp["is_variable_time"]
        if self.is_variable_time == True:  # noqa: E712  (the reference compares with == True; 'Flase' is a str -> False)
            t1 = np.arange(0, self.T_horizon * self.t_ratio, self.T_S, dtype=float)
            t2 = np.arange(t1[-1] + self.T_L, t1[-1] + self.T_L + self.T_horizon * (1 - self.t_ratio), self.T_L)
            self.N_p = len(t1) + len(t2)
            self.t_vector = np.concatenate((t1, t2))
        else:
            self.t_vector = np.arange(0, self.T_horizon + self.T_S, self.T_S, dtype=float)
            self.N_p = horizon_steps(self.T_horizon, self.T_S)
        vp = self.config["vehicle_params"]
        for key in ("Veh_l", "Veh_L", "Veh_m", "Veh_lf", "Veh_lr", "Veh_Iz"):
            setattr(self, key, vp[key])
        self.Veh_W = vp["Veh_W"] if "Veh_W" in vp else vp["Veh_w"]
        tp = self.config["tire_params"]
        self.aopt_f, self.aopt_r, self.Cf_0, self.Cr_0 = tp["aopt_f"], tp["aopt_r"], tp["Cf_0"], tp["Cr_0"]
        self.Fymax_f = self.Cf_0 * self.aopt_f / 2
        self.Fymax_r = self.Cr_0 * self.aopt_r / 2
        dc = self.config["dynamics_constraints"]
        self.vy_max, self.vy_min = dc["vy_max"], dc["vy_min"]
        self.jerk_min, self.jerk_max = dc["jerk_min"], dc["jerk_max"]
        self.df_dot_min = dc["df_dot_min"] * np.pi / 180
        self.df_dot_max = dc["df_dot_max"] * np.pi / 180
        kc = self.config["kinematics_constraints"]
        self.vx_max, self.vx_min = kc["vx_max"], kc["vx_min"]
        self.ax_max, self.ax_min = kc["ax_max"], kc["ax_min"]
        self.df_max = kc["df_max"] * np.pi / 180
        self.df_min = kc["df_min"] * np.pi / 180
        self.Y_max, self.Y_min = kc["Y_max"], kc["Y_min"]
        self.model_type = self.config["model_type"]
        self.num_states = 6 if self.MODEL == _abi.MODEL_DYN else 4
        self.num_controls = 2
        self.last_stats = {}
        self._solvers = {}
        # The reference ships the keep-out rows `g.append(h_func)` and keeps the CBF rows `gamma*h_func + h_dot` commented
        # next to them with `gamma = 1.00` (kin.py:235,247-248).  Same switch here: cbf_rows = True selects the commented form.
        self.cbf_rows = False
        self.gamma = 1.0
        # The reference builds the two-rate grid t_vector when is_variable_time is true (kin.py:19-25) but its NLP keeps T_S in
        # every shooting row and rate bound (kin.py:207,116-121).  Same default here; time_grid_in_nlp = True makes the grid
        # effective: stage i integrates over stage_lengths()[i] (mpcb_set_time_grid).
        self.time_grid_in_nlp = False
        # Start of the solve.  "rollout" (default): U of the caller's start is kept, X is rolled out from x0 with the model and the
        # barrier starts at mu = 10; an instance that fails from there — its restoration phase included — is solved once more from the
        # reference's own first-step start z = 0 (cfg.second_start = 2).  "ipopt": the start is taken exactly as given and mu_init = 0.1,
        # IPOPT's documented behaviour, one attempt.  Why the roll-out stays the default although IPOPT does the other thing: with this
        # library's restatement of IPOPT's algorithm the "ipopt" start does NOT solve the reference's own scene at the shipped horizon
        # (N_p = 50: restoration failed after 42 iterations) and fails on ~25 % of random scenes, the roll-out start with its fallback on
        # < 1 %; on the instances both solve, 92-93 % end at the same trajectory (tests/test_parity_evidence.py).
        self.start = "rollout"
        self.integrator = "euler"                      # "rk4": MPCB_INT_RK4 (kinematic model), NLP rows and plant step alike
        self.f = ModelFunction(self._make_cfg(0))

    # ----- configuration of the HIP library from the YAML values ------------------------------------------
    def _make_cfg(self, n_obs):
        c = default_config(self.MODEL, int(self.N_p), float(self.T_S), int(n_obs))
        c.u_lo[0], c.u_hi[0] = self.df_min, self.df_max
        c.u_lo[1], c.u_hi[1] = self.ax_min, self.ax_max
        c.x_lo[1], c.x_hi[1] = self.Y_min, self.Y_max
        c.x_lo[3], c.x_hi[3] = self.vx_min, self.vx_max
        c.du_lo[0], c.du_hi[0] = self.df_dot_min * self.T_S, self.df_dot_max * self.T_S
        c.veh_l = self.Veh_l
        c.ego_hl, c.ego_hw = self.Veh_L / 2, self.Veh_W / 2
        c.veh_m, c.veh_lf, c.veh_lr, c.veh_Iz = self.Veh_m, self.Veh_lf, self.Veh_lr, self.Veh_Iz
        c.Fymax_f, c.Fymax_r, c.aopt_f, c.aopt_r = self.Fymax_f, self.Fymax_r, self.aopt_f, self.aopt_r
        if getattr(self, "cbf_rows", False) and self.MODEL == _abi.MODEL_KIN:
            c.obs_mode, c.gamma = _abi.OBS_DCBF, float(self.gamma)
        if self.MODEL == _abi.MODEL_DYN:
            c.x_lo[4], c.x_hi[4] = self.vy_min, self.vy_max
            c.du_lo[1], c.du_hi[1] = self.jerk_min * self.T_S, self.jerk_max * self.T_S
        if getattr(self, "start", "rollout") == "ipopt":
            c.init_rollout, c.mu_init, c.second_start = 0, 0.1, 0
        else:
            c.second_start = 2                         # after the first attempt's restoration phase: nothing one attempt solves is lost
        c.integrator = _abi.INT_RK4 if getattr(self, "integrator", "euler") == "rk4" else _abi.INT_EULER
        return c

    def _batch_solver(self, cfg):
        key = (cfg.model, cfg.N, cfg.n_obs, cfg.obs_mode, cfg.gamma, cfg.init_rollout, cfg.second_start, cfg.integrator)
        bs = self._solvers.get(key)
        if bs is None:
            bs = BatchSolver(cfg)
            self._solvers[key] = bs
        return bs

    def stage_lengths(self):
        """Step length of every stage from t_vector: its differences, the last stage as long as the one before it (the variable
        grid of kin.py:19-25 has N_p points for N_p stages); T_S everywhere for the fixed grid."""
        if self.is_variable_time == True:  # noqa: E712
            d = np.diff(np.asarray(self.t_vector, dtype=np.float64))
            return np.append(d, d[-1])[: self.N_p]
        return np.full(int(self.N_p), float(self.T_S))

    def generate_ref_path(self, x0, xs):
        """Quintic lane-change reference (kin.py:258-308; never called by the reference's drivers, kept for the surface):
        over the first 3 s a fifth-order polynomial in x and in y from (x0, vx0, 0 acceleration) to (x0 + vxs*3, y_s, vx_s), then a
        straight extension at the set-point speed up to T_horizon; sampled every 0.1 s.  Returns rows [x, y, heading in DEGREES, v]."""
        x0 = np.asarray(x0, dtype=np.float64).reshape(-1); xs = np.asarray(xs, dtype=np.float64).reshape(-1)
        t_blend, dt = 3.0, 0.1
        n1, n2 = int(t_blend / dt), int((self.T_horizon - t_blend) / dt)
        ta = np.linspace(0.0, t_blend, n1); tb = np.linspace(t_blend, self.T_horizon, n2 + 1)
        pw = lambda t, d: np.array([0.0 if k < d else np.prod(np.arange(k, k - d, -1.0)) * t ** (k - d) for k in range(6)])  # noqa: E731
        A = np.array([pw(0.0, 0), pw(0.0, 1), pw(0.0, 2), pw(t_blend, 0), pw(t_blend, 1), pw(t_blend, 2)])
        cx = np.linalg.solve(A, np.array([x0[0], x0[3], 0.0, xs[3] * t_blend + x0[0], xs[3], 0.0]))
        cy = np.linalg.solve(A, np.array([x0[1], 0.0, 0.0, xs[1], 0.0, 0.0]))
        P = np.stack([ta ** k for k in range(6)], axis=1)
        dP = np.stack([np.zeros_like(ta)] + [k * ta ** (k - 1) for k in range(1, 6)], axis=1)
        xa, ya = P @ cx, P @ cy
        heading = np.degrees(np.arctan2(dP @ cy, dP @ cx))
        xb = xa[-1] + xs[3] * (tb - ta[-1])
        return np.column_stack((np.concatenate((xa, xb)), np.concatenate((ya, np.full_like(tb, ya[-1]))),
                                np.concatenate((heading, np.full_like(tb, heading[-1]))), np.full(n1 + n2 + 1, xs[3])))

    # ----- bounds in the reference's z / g order ---------------------------------------------------------------
    def _box_lists(self):
        lbx, ubx = [], []
        for _ in range(self.N_p):
            lbx += [self.df_min, self.ax_min]
            ubx += [self.df_max, self.ax_max]
        lo = [-np.inf, self.Y_min, -np.inf, self.vx_min]
        hi = [np.inf, self.Y_max, np.inf, self.vx_max]
        if self.MODEL == _abi.MODEL_DYN:
            lo += [self.vy_min, -np.inf]
            hi += [self.vy_max, np.inf]
        for _ in range(self.N_p + 1):
            lbx += lo
            ubx += hi
        return lbx, ubx
