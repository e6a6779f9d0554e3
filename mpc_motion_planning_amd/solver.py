"""BatchSolver — thin Python front of the C ABI (include/mpcbatch.h) for batches of MPC instances.

This is the batched extension of the reference's per-step  `solver = ca.nlpsol(...)` / `res = solver(x0=, p=, ...)`
pair (CasaDi_MPC_Optimize_Multishoot/MPC_CBF_optimize_kin.py:251-254, main_cbf_kin_c_sim.py:100): one handle per
NLP structure, one call per batch of parameter vectors and obstacle sets.  numpy + ctypes only.
"""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import MpcbConfig, dptr, iptr
from ._lib import lib, check


def default_config(model=_abi.MODEL_KIN, N=30, T=0.1, n_obs=0):
    cfg = MpcbConfig()
    check(lib().mpcb_default_config(C.byref(cfg), model, N, T))
    cfg.n_obs = n_obs
    return cfg


def dims(cfg):
    nx, nz, ng = C.c_int32(), C.c_int32(), C.c_int32()
    check(lib().mpcb_dims(C.byref(cfg), C.byref(nx), C.byref(nz), C.byref(ng)))
    return nx.value, nz.value, ng.value


def device_count():
    return lib().mpcb_device_count()


def shard_bounds(B, world, rank):
    """Contiguous slice [lo, hi) of `rank` among `world` (mpcb_shard_bounds: the rule the library itself shards with)."""
    lo, hi = C.c_int64(), C.c_int64()
    check(lib().mpcb_shard_bounds(int(B), int(world), int(rank), C.byref(lo), C.byref(hi)))
    return lo.value, hi.value


def comm_unique_id():
    """128 bytes that identify a new RCCL group (ncclGetUniqueId); rank 0 makes them, every rank passes them to comm_init."""
    buf = C.create_string_buffer(_abi.UNIQUE_ID_BYTES)
    check(lib().mpcb_comm_unique_id(C.cast(buf, C.c_void_p)))
    return buf.raw


def model_rhs(cfg, x, u):
    """f(x,u) of the configured model (the reference's `mpc_solver.f`, kin.py:159)."""
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1))
    u = np.ascontiguousarray(np.asarray(u, dtype=np.float64).reshape(-1))
    out = np.zeros(cfg.nx())
    check(lib().mpcb_model_rhs(C.byref(cfg), dptr(x), dptr(u), dptr(out)))
    return out


class DeviceArray:
    """A caller-owned device buffer (row-major, float64 or int32)."""

    def __init__(self, solver, shape, dtype=np.float64):
        self.solver = solver
        self.shape = tuple(int(s) for s in np.atleast_1d(shape))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        check(lib().mpcb_dev_alloc(solver._h, self.nbytes, C.byref(p)), solver._h)
        self.ptr = p

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.size * a.itemsize == self.nbytes, "size mismatch"
        check(lib().mpcb_dev_upload(self.solver._h, self.ptr, a.ctypes.data_as(C.c_void_p), self.nbytes), self.solver._h)
        return self

    def download(self):
        out = np.empty(self.shape, dtype=self.dtype)
        check(lib().mpcb_dev_download(self.solver._h, out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes), self.solver._h)
        return out

    def free(self):
        if self.ptr is not None and self.solver._h:
            lib().mpcb_dev_free(self.solver._h, self.ptr)
        self.ptr = None


class BatchSolver:
    def __init__(self, cfg, device=0, inflight=1):
        """inflight: launch lanes of the handle (mpcb_set_inflight).  With k > 1 consecutive asynchronous `solve_device` calls
        overlap on the GPU and one large call is cut into chunks that do; results do not depend on it."""
        self.cfg = cfg.copy()
        self._h = C.c_void_p()
        rc = lib().mpcb_create(C.byref(self.cfg), device, C.byref(self._h))
        if rc != 0:
            check(rc, None)
        self.nx, self.nz, self.ng = dims(self.cfg)
        self.N = self.cfg.N
        self.device = device
        self.inflight = 1
        if inflight != 1:
            self.set_inflight(inflight)

    def set_inflight(self, k):
        check(lib().mpcb_set_inflight(self._h, int(k)), self._h)
        self.inflight = int(k)

    def close(self):
        if self._h:
            lib().mpcb_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------------
    def set_bounds(self, lbx, ubx, lbg, ubg):
        """Adopt (and validate) the four lists `initialize_constraints` returns (kin.py:84-134)."""
        lbx = np.ascontiguousarray(lbx, dtype=np.float64); ubx = np.ascontiguousarray(ubx, dtype=np.float64)
        lbg = np.ascontiguousarray(lbg, dtype=np.float64); ubg = np.ascontiguousarray(ubg, dtype=np.float64)
        if lbx.shape != ubx.shape or lbg.shape != ubg.shape:
            raise ValueError("lbx/ubx or lbg/ubg lengths differ")
        check(lib().mpcb_set_bounds(self._h, dptr(lbx), dptr(ubx), lbx.size, dptr(lbg), dptr(ubg), lbg.size), self._h)

    def set_time_grid(self, T_i=None):
        """Step length of every stage (N values), or None for cfg.T everywhere (the reference's behaviour)."""
        if T_i is None:
            check(lib().mpcb_set_time_grid(self._h, None, 0), self._h)
        else:
            t = np.ascontiguousarray(T_i, dtype=np.float64).reshape(-1)
            check(lib().mpcb_set_time_grid(self._h, dptr(t), t.size), self._h)

    def _obs(self, obs, B):
        if self.cfg.n_obs == 0:
            return None, _abi.OBSIN_STATIC
        obs = np.ascontiguousarray(obs, dtype=np.float64)
        n = self.cfg.n_obs
        if obs.size == B * n * 6:
            return obs.reshape(B, n, 6), _abi.OBSIN_STATIC
        if obs.size == B * n * (self.N + 1) * 6:
            return obs.reshape(B, n, self.N + 1, 6), _abi.OBSIN_PREDICTED
        raise ValueError("obs has %d values; expected [B,%d,6] or [B,%d,%d,6]" % (obs.size, n, n, self.N + 1))

    def solve_batch(self, x0, xs, obs=None, z0=None, multipliers=False):
        """x0, xs [B,nx]; obs [B,n_obs,6] | [B,n_obs,N+1,6]; z0 [B,nz] | None  ->  dict(z, obj, status, iters, kkt[, lam_g, lam_x])"""
        x0 = np.ascontiguousarray(np.atleast_2d(np.asarray(x0, dtype=np.float64)))
        xs = np.ascontiguousarray(np.atleast_2d(np.asarray(xs, dtype=np.float64)))
        B = x0.shape[0]
        if x0.shape != (B, self.nx) or xs.shape != (B, self.nx):
            raise ValueError("x0 and xs must be [B,%d]" % self.nx)
        obs, kind = self._obs(obs, B)
        if z0 is not None:
            z0 = np.ascontiguousarray(np.asarray(z0, dtype=np.float64).reshape(B, self.nz))
        z = np.empty((B, self.nz)); obj = np.empty(B); st = np.empty(B, np.int32); it = np.empty(B, np.int32)
        kkt = np.empty((B, 4))
        lam_g = np.empty((B, self.ng)) if multipliers else None
        lam_x = np.empty((B, self.nz)) if multipliers else None
        check(lib().mpcb_solve(self._h, B, dptr(x0), dptr(xs), dptr(obs), kind, dptr(z0), dptr(z), dptr(obj), iptr(st),
                               iptr(it), dptr(kkt), dptr(lam_g), dptr(lam_x)), self._h)
        out = dict(z=z, obj=obj, status=st, iters=it, kkt=kkt)
        if multipliers:
            out["lam_g"] = lam_g; out["lam_x"] = lam_x
        return out

    def solve_trace(self, x0, xs, obs=None, z0=None):
        """One instance with its iteration log: dict(z, status, iters, trace[iters+1, 8]) with trace columns
        mu, scaled error, theta, f, alpha_primal_max, alpha, alpha_dual, delta_w."""
        x0 = np.ascontiguousarray(np.asarray(x0, dtype=np.float64).reshape(1, self.nx)); xs = np.ascontiguousarray(np.asarray(xs, dtype=np.float64).reshape(1, self.nx))
        obs, kind = self._obs(obs, 1)
        if z0 is not None:
            z0 = np.ascontiguousarray(np.asarray(z0, dtype=np.float64).reshape(1, self.nz))
        z = np.empty((1, self.nz)); st = np.empty(1, np.int32); it = np.empty(1, np.int32); tr = np.zeros((self.cfg.max_iter + 1, 8))
        check(lib().mpcb_solve_trace(self._h, dptr(x0), dptr(xs), dptr(obs), kind, dptr(z0), dptr(z), iptr(st), iptr(it), dptr(tr)), self._h)
        return dict(z=z[0], status=int(st[0]), iters=int(it[0]), trace=tr[: int(it[0]) + 1])

    def closed_loop(self, x0, xs, obs_state=None, steps=80, obs_motion=_abi.OBSMOVE_STATIC, hold_on_failure=False,
                    advance_first_only=False):
        """Receding-horizon loop on the device (main_cbf_kin_c_sim.py:87-123).  obs_motion: OBSMOVE_STATIC (obstacles fixed,
        main_cbf_kin_c_sim.py), OBSMOVE_PREDICTED (constant-velocity obstacles predicted per solve and advanced per step,
        main_cbf_kin_c_sim_pre.py), OBSMOVE_CURRENT (advanced, no prediction).
        hold_on_failure: a step whose solve fails applies the previous plan (hold-and-shift) instead of the failed iterate.
        advance_first_only: only obstacle 0 moves between steps (main_cbf_kin_c_sim_pre.py:106).
        Returns dict(x_hist, u_hist, status, iters, obs_state)."""
        x0 = np.ascontiguousarray(np.atleast_2d(np.asarray(x0, dtype=np.float64)))
        xs = np.ascontiguousarray(np.atleast_2d(np.asarray(xs, dtype=np.float64)))
        B = x0.shape[0]
        ob = None
        if self.cfg.n_obs:
            ob = np.array(obs_state, dtype=np.float64).reshape(B, self.cfg.n_obs, 6).copy()
        xh = np.empty((B, steps + 1, self.nx)); uh = np.empty((B, steps, 2))
        st = np.empty((B, steps), np.int32); it = np.empty((B, steps), np.int32)
        flags = (_abi.CL_HOLD_ON_FAILURE if hold_on_failure else 0) | (_abi.CL_ADVANCE_FIRST_ONLY if advance_first_only else 0)
        check(lib().mpcb_closed_loop(self._h, B, steps, dptr(x0), dptr(xs), dptr(ob), int(obs_motion), flags, dptr(xh), dptr(uh),
                                     iptr(st), iptr(it)), self._h)
        return dict(x_hist=xh, u_hist=uh, status=st, iters=it, obs_state=ob)

    # ----- scene generation on the device (include/mpcbatch.h, "scene generation") ----------------------------------
    def sample_scenes(self, kind, B, seed, first_index=0):
        """B scenes of distribution `kind` (_abi.SCENES_C2 / C3 / C4) drawn on the device with the counter-based generator;
        scene i of a population is the same whichever call or GPU draws it (global index first_index + i).  Downloads
        (x0 [B,nx], xs [B,nx], obs [B,n_obs,6])."""
        no = self.cfg.n_obs
        d_x0 = self.device_array((B, self.nx)); d_xs = self.device_array((B, self.nx)); d_ob = self.device_array((B, max(no, 1), 6))
        check(lib().mpcb_sample_scenes(self._h, int(kind), int(B), int(seed), int(first_index), d_x0.ptr, d_xs.ptr, d_ob.ptr), self._h)
        self.sync()
        out = d_x0.download(), d_xs.download(), d_ob.download()[:, :no]
        for d in (d_x0, d_xs, d_ob):
            d.free()
        return out

    def closed_loop_sampled(self, kind, B, seed, first_index=0, steps=80, obs_motion=_abi.OBSMOVE_STATIC, hold_on_failure=False,
                            advance_first_only=False):
        """closed_loop on scenes drawn on the device; also returns the scenes (x0, obs0)."""
        no = self.cfg.n_obs
        x0 = np.empty((B, self.nx)); ob = np.empty((B, no, 6)) if no else None
        xh = np.empty((B, steps + 1, self.nx)); uh = np.empty((B, steps, 2))
        st = np.empty((B, steps), np.int32); it = np.empty((B, steps), np.int32)
        flags = (_abi.CL_HOLD_ON_FAILURE if hold_on_failure else 0) | (_abi.CL_ADVANCE_FIRST_ONLY if advance_first_only else 0)
        check(lib().mpcb_closed_loop_sampled(self._h, int(kind), int(B), int(seed), int(first_index), int(steps), int(obs_motion), flags,
                                             dptr(x0), dptr(ob), dptr(xh), dptr(uh), iptr(st), iptr(it)), self._h)
        return dict(x0=x0, obs0=ob, x_hist=xh, u_hist=uh, status=st, iters=it)

    def predict_obstacles(self, obs, dt, N):
        """Device twin of Obs_prediction.obs_prediction: obs [n,6] -> [n,N+1,6]."""
        obs = np.ascontiguousarray(np.asarray(obs, dtype=np.float64).reshape(-1, 6))
        out = np.empty((len(obs), N + 1, 6))
        check(lib().mpcb_predict_obstacles(self._h, len(obs), int(N), float(dt), dptr(obs), dptr(out)), self._h)
        return out

    def ref_path_window(self, x_start, x0, xs, T_horizon, dt, last_idx):
        """Device twin of RefPathGenerator.define_ref_path + find_ref_traj for B instances: (window [B,N_p+1,4], min_idx [B])."""
        x0 = np.ascontiguousarray(np.asarray(x0, dtype=np.float64).reshape(-1, 4)); xs = np.ascontiguousarray(np.asarray(xs, dtype=np.float64).reshape(-1, 4))
        li = np.ascontiguousarray(np.asarray(last_idx, dtype=np.int32).reshape(-1))
        win = np.empty((len(x0), int(T_horizon / dt) + 1, 4))
        check(lib().mpcb_ref_path_window(self._h, len(x0), float(x_start), dptr(x0), dptr(xs), float(T_horizon), float(dt), iptr(li), dptr(win)), self._h)
        return win, li

    # ----- device-resident path (bench.py, multi-GPU plumbing) -------------------------------------------
    def device_array(self, shape, dtype=np.float64):
        return DeviceArray(self, shape, dtype)

    def solve_device(self, B, d_x0, d_xs, d_obs, obs_kind, d_z0, d_z, d_obj=None, d_status=None, d_iters=None, d_kkt=None,
                     d_lam_g=None, d_lam_x=None, sync=False):
        """Raw device pointers (ints / c_void_p / DeviceArray).  Asynchronous on the handle's stream unless sync."""
        def p(v):
            if v is None:
                return None
            if isinstance(v, DeviceArray):
                return v.ptr
            return C.c_void_p(int(v)) if not isinstance(v, C.c_void_p) else v
        check(lib().mpcb_solve_device(self._h, B, p(d_x0), p(d_xs), p(d_obs), obs_kind, p(d_z0), p(d_z), p(d_obj), p(d_status),
                                      p(d_iters), p(d_kkt), p(d_lam_g), p(d_lam_x), 1 if sync else 0), self._h)

    # ----- multi-GPU (include/mpcbatch.h, "multi-GPU") ------------------------------------------------------
    def set_devices(self, ids):
        """One process driving several devices: solve_batch then shards the batch over them and all-gathers z (RCCL)."""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        check(lib().mpcb_set_devices(self._h, iptr(ids), len(ids)), self._h)

    def comm_init(self, unique_id, rank, world):
        """One process per GPU: join the RCCL group identified by `unique_id` (comm_unique_id() of rank 0)."""
        buf = C.create_string_buffer(bytes(unique_id), _abi.UNIQUE_ID_BYTES)
        check(lib().mpcb_comm_init_rank(self._h, C.cast(buf, C.c_void_p), int(rank), int(world)), self._h)

    def comm_info(self):
        w, r = C.c_int32(), C.c_int32()
        check(lib().mpcb_comm_info(self._h, C.byref(w), C.byref(r)), self._h)
        return w.value, r.value

    def allgather(self, d_send, d_recv, count):
        p = lambda v: v.ptr if isinstance(v, DeviceArray) else C.c_void_p(int(v))   # noqa: E731
        check(lib().mpcb_allgather(self._h, p(d_send), p(d_recv), int(count)), self._h)

    def allreduce(self, values, op="sum"):
        """In-place over the ranks on a small float64 array; also a barrier."""
        v = np.ascontiguousarray(values, dtype=np.float64)
        check(lib().mpcb_allreduce(self._h, dptr(v), v.size, 0 if op == "sum" else 1), self._h)
        return v

    def gathered_z(self, index, B):
        """(device group) the [B, nz] trajectories of the last solve_batch as device `index` of the group holds them after the
        all-gather, downloaded for inspection."""
        p = C.c_void_p()
        check(lib().mpcb_gathered_z(self._h, int(index), C.byref(p)), self._h)
        G = self.comm_info()[0]
        longest = max(hi - lo for lo, hi in (shard_bounds(B, G, g) for g in range(G)))
        raw = np.empty((G * longest, self.nz))
        check(lib().mpcb_dev_download(self._h, raw.ctypes.data_as(C.c_void_p), p, raw.nbytes), self._h)
        return np.concatenate([raw[g * longest: g * longest + (hi - lo)] for g, (lo, hi) in enumerate(shard_bounds(B, G, g) for g in range(G))])

    def sync(self):
        check(lib().mpcb_sync(self._h), self._h)

    def wait_for(self, other):
        """This handle's stream waits (on the device) for everything queued so far on `other`'s stream."""
        check(lib().mpcb_stream_wait(self._h, other._h), self._h)

    def record(self, slot):
        """Mark "everything queued so far on this handle" in event slot `slot` (mpcb_event_record)."""
        check(lib().mpcb_event_record(self._h, int(slot)), self._h)

    def wait_mark(self, other, slot):
        """This handle's later work waits for `other`'s mark `slot` (mpcb_event_wait); no-op if never recorded."""
        check(lib().mpcb_event_wait(self._h, other._h, int(slot)), self._h)

    def timing(self, reset=False):
        n = C.c_int32(); tot = C.c_double(); last = C.c_double()
        check(lib().mpcb_timing(self._h, 1 if reset else 0, C.byref(n), C.byref(tot), C.byref(last)), self._h)
        return dict(launches=n.value, total_ms=tot.value, last_ms=last.value)
