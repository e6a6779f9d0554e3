"""Seeded synthetic scene samplers for the BASELINE configs.  numpy only.

C2: kinematic bicycle, N=30, one static obstacle at the shipped position (main_cbf_kin_c_sim.py:55),
    set-point xs = [400, 3.5, 0, 30] (main_cbf_kin_c_sim.py:49), random feasible x0.
C3: three moving obstacles predicted at constant velocity (Obs_prediction.py:27-30).
"""
import numpy as np

SHIPPED_X0 = np.array([0.0, 3.0, 0.0, 15.0])          # main_cbf_kin_c_sim.py:45
SHIPPED_XS = np.array([400.0, 3.5, 0.0, 30.0])        # main_cbf_kin_c_sim.py:49
SHIPPED_OBS = np.array([[50.0, 3.5, 0.0, 8.0, 4.8, 1.8]])   # main_cbf_kin_c_sim.py:55


def ellipse_h(xy, obs, ego_hl=2.4, ego_hw=0.9, safe_l=1.0, safe_w=0.5):
    """h_j = (x-ox)^2/sX^2 + (y-oy)^2/sY^2 - 1 for points xy[...,2] and obstacles obs[...,6] (kin.py:238-244)."""
    sx = ego_hl + obs[..., 4] / 2 + safe_l
    sy = ego_hw + obs[..., 5] / 2 + safe_w
    return (xy[..., 0] - obs[..., 0]) ** 2 / sx ** 2 + (xy[..., 1] - obs[..., 1]) ** 2 / sy ** 2 - 1.0


def sample_c2(B, seed=0, margin=0.05):
    """x0 [B,4], xs [B,4], obs [B,1,6]."""
    rng = np.random.default_rng(seed)
    x0 = np.empty((B, 4))
    n = 0
    while n < B:
        m = 2 * (B - n) + 16
        c = np.stack([rng.uniform(0, 30, m), rng.uniform(-0.5, 4.5, m), rng.uniform(-0.1, 0.1, m),
                      rng.uniform(5, 25, m)], axis=1)
        ok = ellipse_h(c[:, :2], SHIPPED_OBS[0]) >= margin
        c = c[ok][: B - n]
        x0[n:n + len(c)] = c
        n += len(c)
    xs = np.tile(SHIPPED_XS, (B, 1))
    obs = np.tile(SHIPPED_OBS, (B, 1, 1))
    return x0, xs, obs


def predict_obstacles(obs, dt, N):
    """Constant-velocity, constant-heading roll-out: obs [..., 6] -> [..., N+1, 6]  (Obs_prediction.py:19-34)."""
    obs = np.asarray(obs, dtype=np.float64)
    k = np.arange(N + 1, dtype=np.float64)
    out = np.repeat(obs[..., None, :], N + 1, axis=-2).copy()
    # the reference accumulates x += v cos(theta) dt step by step; do the same for bit-equal sums
    x = obs[..., 0].copy(); y = obs[..., 1].copy()
    vx = obs[..., 3] * np.cos(obs[..., 2]) * dt
    vy = obs[..., 3] * np.sin(obs[..., 2]) * dt
    for i in range(N + 1):
        out[..., i, 0] = x; out[..., i, 1] = y
        x = x + vx; y = y + vy
    del k
    return out


def sample_c3(B, N=30, dt=0.1, seed=0, n_obs=3, margin=0.05):
    """x0 [B,4], xs [B,4], obs0 [B,n_obs,6], obs_traj [B,n_obs,N+1,6]."""
    rng = np.random.default_rng(seed)
    x0 = np.empty((B, 4)); obs0 = np.empty((B, n_obs, 6))
    n = 0
    while n < B:
        m = 2 * (B - n) + 16
        c = np.stack([rng.uniform(0, 30, m), rng.uniform(-0.5, 4.5, m), rng.uniform(-0.1, 0.1, m),
                      rng.uniform(5, 25, m)], axis=1)
        o = np.empty((m, n_obs, 6))
        o[..., 0] = rng.uniform(30, 120, (m, n_obs))
        o[..., 1] = rng.choice([0.0, 3.5], (m, n_obs)) + rng.uniform(-0.3, 0.3, (m, n_obs))
        o[..., 2] = 0.0
        o[..., 3] = rng.uniform(5, 15, (m, n_obs))
        o[..., 4] = 4.8; o[..., 5] = 1.8
        ok = np.all(ellipse_h(c[:, None, :2], o) >= margin, axis=1)
        # obstacles must not overlap each other: centres at least one car length / width apart
        for a in range(n_obs):
            for b in range(a + 1, n_obs):
                ok &= (np.abs(o[:, a, 0] - o[:, b, 0]) > 12.0) | (np.abs(o[:, a, 1] - o[:, b, 1]) > 2.5)
        c = c[ok][: B - n]; o = o[ok][: B - n]
        x0[n:n + len(c)] = c; obs0[n:n + len(c)] = o
        n += len(c)
    xs = np.tile(SHIPPED_XS, (B, 1))
    return x0, xs, obs0, predict_obstacles(obs0, dt, N)


DYN_X0 = np.array([0.0, 0.0, 0.0, 10.0, 0.0, 0.0])            # main_cbf_dyn_c_sim.py:44
DYN_XS = np.array([600.0, 3.5, 0.0, 15.0, 0.0, 0.0])          # main_cbf_dyn_c_sim.py:48
DYN_OBS = np.array([[100.0, -3.5, 0.0, 0.0, 0.0, 0.0]])       # main_cbf_dyn_c_sim.py:51 (only x, y are used, dyn.py:238-239)


def dyn_h(xy, obs, sx=4.0, sy=1.0):
    """dyn.py:240-243: (x-ox)^2/4^2 + (y-oy)^2/1^2 - 1  (the row is sqrt(h) >= 1, i.e. h >= 1)."""
    return (xy[..., 0] - obs[..., 0]) ** 2 / sx ** 2 + (xy[..., 1] - obs[..., 1]) ** 2 / sy ** 2 - 1.0


def sample_c4(B, seed=0, n_obs=3):
    """Dynamic-bicycle scenes: x0 = [x, y, phi, vx~U(8,20), 0, 0], xs = [600, 3.5, 0, 15, 0, 0],
    n_obs static obstacles with the fixed 4 x 1 semi-axes of dyn.py:240-241, none within h < 1.5 of the start."""
    rng = np.random.default_rng(seed)
    x0 = np.zeros((B, 6)); obs = np.zeros((B, n_obs, 6))
    n = 0
    while n < B:
        m = 2 * (B - n) + 16
        c = np.zeros((m, 6))
        c[:, 0] = rng.uniform(0, 30, m); c[:, 1] = rng.uniform(-0.5, 4.5, m); c[:, 2] = rng.uniform(-0.05, 0.05, m); c[:, 3] = rng.uniform(8, 20, m)
        o = np.zeros((m, n_obs, 6))
        o[..., 0] = rng.uniform(40, 200, (m, n_obs)); o[..., 1] = rng.choice([-3.5, 0.0, 3.5, 7.0], (m, n_obs)) + rng.uniform(-0.3, 0.3, (m, n_obs))
        ok = np.all(dyn_h(c[:, None, :2], o) >= 1.5, axis=1)
        c = c[ok][: B - n]; o = o[ok][: B - n]
        x0[n:n + len(c)] = c; obs[n:n + len(c)] = o
        n += len(c)
    return x0, np.tile(DYN_XS, (B, 1)), obs
