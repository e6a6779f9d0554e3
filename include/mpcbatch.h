/*
 * mpcbatch.h — C ABI of libmpcbatch: batched multiple-shooting MPC solves on AMD MI355X (gfx950).
 *
 * This library takes the place of ONE call in the reference (ZhuorenLi/MPC_motion_planning):
 *
 *     solver = ca.nlpsol('solver', 'ipopt', nlp_prob, opts)      CMOM/MPC_CBF_optimize_kin.py:251-254
 *     res    = solver(x0=, p=, lbg=, lbx=, ubg=, ubx=)            CMOM/main_cbf_kin_c_sim.py:100
 *
 * (CMOM = CasaDi_MPC_Optimize_Multishoot/).  The reference has no FFI of its own — its solve lives in
 * the third-party casadi wheel (IPOPT + MUMPS) — so the entry points below are what a ctypes binding
 * placed at that seam needs: "describe the NLP once" (mpcb_create / mpcb_set_bounds) and "solve it for a
 * batch of parameter vectors P = [x0; xs] and obstacle sets" (mpcb_solve).  INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - plain C, opaque handle, caller-owned buffers, every function returns an int code
 *     (MPCB_OK = 0, < 0 = API error, text via mpcb_last_error).  Nothing aborts.
 *   - all floating point is IEEE double, as in the reference (CasADi DM / numpy float64).
 *   - batch arrays are row-major [B, ...] both on the host and on the device: one problem instance
 *     is one contiguous row, which is the coalesced layout for "one wavefront per instance".
 *   - decision vector order is the reference's:  z = [vec(U); vec(X)], z[2i+c] = U[c,i],
 *     z[2N + nx*k + s] = X[s,k]                                  CMOM/MPC_CBF_optimize_kin.py:160-161,250
 *   - constraint row order (g, lbg, ubg) is the reference's       CMOM/MPC_CBF_optimize_kin.py:107-132,190-247
 *   - per-instance solver outcome goes to status[] (MPCB_ST_*), never to the return code.
 *   - one handle = one device + one stream (+ optional launch lanes, mpcb_set_inflight); calls on one handle must be
 *     serialised by the caller (one host thread at a time).  A launch ends with its slowest instance, so throughput needs
 *     several launches in flight: mpcb_set_inflight(h, k) lets consecutive asynchronous mpcb_solve_device calls of ONE handle
 *     overlap (bench.py's default: one handle, sixteen lanes); distinct handles are independent as well.
 *   - buffers of solves that are in flight at the same time (lanes, or several handles) must be distinct: in particular the
 *     status array, through which the two passes of a solve communicate.
 */
#ifndef MPCBATCH_H
#define MPCBATCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPCB_ABI_VERSION 3

/* return codes */
#define MPCB_OK              0
#define MPCB_E_INVALID      -1   /* bad argument / config */
#define MPCB_E_BOUNDS       -2   /* lbx/ubx/lbg/ubg do not match the row pattern of the configured NLP */
#define MPCB_E_DEVICE       -3   /* HIP runtime error (no GPU, out of memory, launch failure) */
#define MPCB_E_UNSUPPORTED  -4   /* valid request that this build does not implement */

/* per-instance status */
#define MPCB_ST_SOLVED        0  /* scaled NLP error <= tol (IPOPT "Optimal Solution Found") */
#define MPCB_ST_MAXITER       1  /* max_iter reached; last iterate returned (the reference ignores status too) */
#define MPCB_ST_LINESEARCH    2  /* no acceptable step and cfg.restoration == 0 (where IPOPT would enter restoration) */
#define MPCB_ST_INFEASIBLE_X0 3  /* x0 violates a state box or lies inside an obstacle row at node 0 */
#define MPCB_ST_NUMERIC       4  /* regularisation exhausted or non-finite number */
#define MPCB_ST_INFEASIBLE    5  /* the restoration phase converged to a stationary point of the constraint violation with
                                    violation > tol: LOCAL infeasibility (IPOPT "Converged to a point of local infeasibility.
                                    Problem may be infeasible", return_status Infeasible_Problem_Detected) */
#define MPCB_ST_RESTO_FAILED  6  /* the restoration phase itself found no acceptable step / ran into max_iter without reducing
                                    the violation (IPOPT "Restoration_Failed") */
/*      7 is used internally between the two passes of a solve and never returned */
#define MPCB_ST_ACCEPTABLE    8  /* acceptable_iter iterations in a row met the acceptable_* tolerances (IPOPT "Solved To Acceptable
                                    Level"): the only two options the reference sets belong to this test, kin.py:252-253 */

/* models                                                     reference */
#define MPCB_MODEL_KIN 0      /* 4-state kinematic bicycle     CMOM/MPC_CBF_optimize_kin.py:153-156 */
#define MPCB_MODEL_DYN 1      /* 6-state dynamic bicycle       CMOM/MPC_CBF_optimize_dyn.py:156-170 */

/* obstacle rows */
#define MPCB_OBS_KEEPOUT 0    /* h_j(X_i) >= 0, the shipped form          CMOM/MPC_CBF_optimize_kin.py:247 */
#define MPCB_OBS_DCBF    1    /* gamma*h_i(X_i) + h_i(X_{i+1}) - h_i(X_i) >= 0, the commented form   kin.py:245-248 */

/* barrier-parameter strategy: IPOPT's default (monotone Fiacco-McCormick, kappa_mu = 0.2, theta_mu = 1.5) is the only one */
#define MPCB_MU_MONOTONE 0

/* integrator of the shooting rows */
#define MPCB_INT_EULER 0      /* X_{i+1} = X_i + T f(X_i,U_i): what the reference's NLP and plant use   kin.py:207, main_cbf_kin_c_sim.py:17-18 */
#define MPCB_INT_RK4   1      /* classical fourth-order Runge-Kutta step with the control held over the interval (BASELINE.json's north_star; the
                                 reference itself has no RK4 NLP): kinematic model, up to 3 obstacles, keep-out / gamma = 1 rows; the plant step
                                 of mpcb_closed_loop follows the same integrator */

/* obstacle input kinds for mpcb_solve */
#define MPCB_OBSIN_STATIC    0 /* [B, n_obs, 6]       rows [x,y,theta,v,l,w]   CMOM/main_cbf_kin_c_sim.py:55 */
#define MPCB_OBSIN_PREDICTED 1 /* [B, n_obs, N+1, 6]  CMOM/Obs_prediction.py:33-38 (rows i<N are read) */

#define MPCB_NX_MAX   6
#define MPCB_NU       2
#define MPCB_NOBS_MAX 8
#define MPCB_N_MAX    63      /* N+1 nodes map onto the 64 lanes of one wavefront */

typedef struct mpcb_config {
  uint32_t struct_size;       /* = sizeof(mpcb_config); checked */
  int32_t  model;             /* MPCB_MODEL_* */
  int32_t  N;                 /* N_p, shooting intervals                  kin.py:32-33 */
  int32_t  n_obs;             /* obstacle rows per node, 0..MPCB_NOBS_MAX */
  int32_t  obs_mode;          /* MPCB_OBS_* */
  int32_t  obs_terminal;      /* 0: rows at nodes 0..N-1 (kin.py:236); 1: nodes 0..N (dyn.py:242) */
  int32_t  du0_cost;          /* 1: (U_0-Ulast)' DR (U_0-Ulast) in the cost (kin.py:203-204); 0: none (dyn.py:223-224) */
  int32_t  rate_interleaved;  /* order of the rate rows inside g: 0 = one block after all dynamics rows (kin.py:211-216),
                                 1 = after each stage's dynamics rows (dyn.py:226-231).  Affects lam_g / bounds order only. */
  int32_t  max_iter;          /* ipopt.max_iter = 100                     kin.py:252 */
  int32_t  mu_strategy;       /* MPCB_MU_MONOTONE */
  int32_t  init_rollout;      /* 0: take the X part of the start as given (what IPOPT receives);
                                 1: keep U of the start, roll X out from x0 with the model (multiple-shooting warm start) */
  int32_t  integrator;        /* MPCB_INT_EULER (the reference, default) or MPCB_INT_RK4 */
  int32_t  restoration;       /* 1 (default): a failed line search enters the feasibility-restoration phase (IPOPT's default
                                 behaviour); 0: it ends the solve with MPCB_ST_LINESEARCH (abi 1 behaviour) */
  double   T;                 /* T_S */
  double   gamma;             /* DCBF gamma in (0, 1] (kin.py:235 sets 1.0); < 1: kinematic model only, rows i = 0..N-1 */
  double   Q[MPCB_NX_MAX];    /* diag of Q                                kin.py:168-172 */
  double   R[MPCB_NU];        /* diag of R                                kin.py:179-181 */
  double   DR[MPCB_NU];       /* diag of DR                               kin.py:182-184 */
  double   u_last[MPCB_NU];   /* Ulast = [0,0]                            kin.py:193 */
  double   u_lo[MPCB_NU], u_hi[MPCB_NU];       /* boxes on U (lbx)        kin.py:90-95 */
  double   x_lo[MPCB_NX_MAX], x_hi[MPCB_NX_MAX]; /* boxes on X, +-inf = none  kin.py:97-105 */
  double   du_lo[MPCB_NU], du_hi[MPCB_NU];     /* rows U[c,i]-U[c,i-1], i=1..N-1; +-inf = no row  kin.py:116-121,211-216 */
  double   obs_hmin;          /* row is h >= obs_hmin: 0 (kin.py:129-132); 1 for the dyn sqrt(h)>=1 form (dyn.py:131-133,243) */
  double   ego_hl, ego_hw;    /* Veh_L/2, Veh_W/2                         kin.py:220-221 */
  double   safe_disl, safe_disw; /* 1.0, 0.5                              kin.py:224-225 */
  double   obs_sx_fixed, obs_sy_fixed; /* > 0: fixed semi-axes (dyn.py:240-241: 4, 1); else from l,w */
  double   veh_l;             /* wheelbase (kin)                          kin.py:155 */
  double   veh_m, veh_lf, veh_lr, veh_Iz;      /* dyn                     dyn.py:165-170 */
  double   Fymax_f, Fymax_r, aopt_f, aopt_r;   /* dyn tyre                dyn.py:158-159 */
  /* interior-point options: IPOPT's documented defaults unless the reference sets them (kin.py:252-253) */
  double   tol;               /* 1e-8 */
  double   mu_init;           /* IPOPT: 0.1; mpcb_default_config sets 10 because its start (init_rollout = 1) is dynamics-feasible */
  double   bound_push;        /* 0.01 (kappa_1) */
  double   bound_frac;        /* 0.01 (kappa_2) */
  double   bound_relax;       /* 1e-8 (bound_relax_factor) */
  double   max_gradient;      /* 100 (nlp_scaling_max_gradient) */
  /* termination as IPOPT's OptimalityErrorConvergenceCheck does it: "optimal" needs the scaled error <= tol AND three UNSCALED
     gates (the objective scaling of these NLPs is ~1e-4, so the complementarity gate can bind); "acceptable" after acceptable_iter
     iterations in a row within the acceptable_* tolerances whose objective changed by less than acceptable_obj_change_tol */
  double   dual_inf_tol;                 /* 1      max-norm of the unscaled dual infeasibility */
  double   constr_viol_tol;              /* 1e-4   max-norm of the unscaled constraint violation */
  double   compl_inf_tol;                /* 1e-4   max-norm of the unscaled complementarity */
  double   acceptable_tol;               /* IPOPT 1e-6; the reference sets 1e-8               kin.py:252 */
  double   acceptable_obj_change_tol;    /* IPOPT 1e20; the reference sets 1e-6               kin.py:253 */
  double   acceptable_constr_viol_tol;   /* 1e-2 */
  double   acceptable_dual_inf_tol;      /* 1e10 */
  double   acceptable_compl_inf_tol;     /* 1e-2 */
  int32_t  acceptable_iter;              /* 15; 0 switches the acceptable test off */
  int32_t  second_start;                 /* An instance whose solve from a roll-out start (init_rollout = 1) fails (no acceptable step, a run of
                                            tiny steps, max_iter, numerics) is solved once more from the reference's own first-step start z = 0
                                            (main_cbf_kin_c_sim.py:47-50; dynamic model: 0 except vx = x0's, the tyre model divides by vx).
                                              1: INSTEAD of the first attempt's restoration phase — only the second attempt enters the
                                                restoration phase; three launches per solve.  Right for cold starts: a first attempt that
                                                stalls in front of an obstacle is better restarted than restored;
                                              2: AFTER the first attempt's restoration phase — every instance one attempt solves stays solved,
                                                bit for bit; four launches per solve.  Right for warm starts (a closed loop): restoring from
                                                a good start vector beats starting over;
                                              3 (mpcb_default_config): 1 for a solve without a start vector (z0 = NULL), 2 for a solve with one
                                                (every step of mpcb_closed_loop, the drop-in classes);
                                              0: one attempt, as IPOPT.
                                            `iters` counts both attempts, each has max_iter of its own.  Without a roll-out (init_rollout = 0:
                                            the start taken as given, IPOPT's behaviour) there is one attempt, whatever this field says. */
  double   start_steer;                  /* Cold start only (z0 = NULL, init_rollout = 1, n_obs > 0).  The roll-out from zero controls is a straight
                                            line; when it passes an obstacle row closer than h - obs_hmin < 1 at some node, it is replaced by the
                                            roll-out with the constant steering angle +-start_steer [rad] (acceleration 0): away from the centre of
                                            that obstacle, or to its other side when the y box (x_lo[1], x_hi[1]) leaves no room for the row's
                                            ellipse on that side.  A straight path that runs head-on into an obstacle is a stationary point of the
                                            violation at which an interior-point iteration stalls; the slight turn breaks the tie.
                                            mpcb_default_config: 0.03; 0 = off (the start of rounds 1-2). */
} mpcb_config;

typedef struct mpcb_handle mpcb_handle;

/* Fill cfg with the shipped kinematic problem (weights kin.py:168-184, limits from mpc_parameters.yaml,
 * IPOPT defaults) for horizon N and step T; model = MPCB_MODEL_KIN or MPCB_MODEL_DYN. */
int mpcb_default_config(mpcb_config* cfg, int32_t model, int32_t N, double T);

/* sizes implied by a config: nx, nz = 2N + nx(N+1), ng = rows of g in the reference's order */
int mpcb_dims(const mpcb_config* cfg, int32_t* nx, int32_t* nz, int32_t* ng);

int mpcb_device_count(void);

/* Create a solver for one NLP structure on HIP device `device` (replaces ca.nlpsol(...), kin.py:254). */
int mpcb_create(const mpcb_config* cfg, int32_t device, mpcb_handle** out);
int mpcb_destroy(mpcb_handle* h);
const char* mpcb_last_error(const mpcb_handle* h);   /* h may be NULL: last creation error */

/* Check caller-supplied bounds (the lists initialize_constraints returns, kin.py:84-134) against the
 * configured row pattern and adopt the numeric limits.  A length or pattern mismatch — e.g. the
 * interleaving defect of MPC_CBF_optimize_dyn.py:112-129 vs :215-231 — is MPCB_E_BOUNDS. */
int mpcb_set_bounds(mpcb_handle* h, const double* lbx, const double* ubx, int32_t nz,
                    const double* lbg, const double* ubg, int32_t ng);

/* Per-stage step lengths T_0..T_{N-1} (host pointer, n = N; T_i = NULL or n = 0 returns to cfg.T everywhere).  The reference
 * computes a two-rate grid t_vector when `is_variable_time` is true (kin.py:19-25: steps of T_S, then of T_L) but never uses it:
 * its shooting rows and rate bounds keep T_S (kin.py:207,116-121).  With a grid set, stage i integrates X_{i+1} = X_i + T_i f,
 * rate row i is bounded by rate * T_{i-1}, obstacle predictions inside mpcb_closed_loop are taken at the grid's node times and
 * its plant step uses T_0.  Without one (the default, and what the reference does) everything uses cfg.T. */
int mpcb_set_time_grid(mpcb_handle* h, const double* T_i, int32_t n);

/* Solve B independent instances (replaces `res = solver(x0=, p=, ...)`, main_cbf_kin_c_sim.py:100).
 *   x0   [B, nx]   initial state  (P[0:nx])
 *   xs   [B, nx]   set-point      (P[nx:2nx])
 *   obs  [B, n_obs, 6] or [B, n_obs, N+1, 6] per obs_kind; may be NULL when n_obs == 0
 *   z0   [B, nz]   primal start (solver(x0=...)); NULL = zeros (main_cbf_kin_c_sim.py:47-50)
 * outputs (each may be NULL except z):
 *   z [B, nz] = res['x'];  obj [B] = res['f'];  status [B];  iters [B];
 *   kkt [B, 4] = {scaled NLP error, max |constraint violation|, max |dual infeasibility| (unscaled), final mu}
 *   lam_g [B, ng] = res['lam_g'];  lam_x [B, nz] = res['lam_x']   (IPOPT sign convention)
 * Host pointers; copies in and out go over the handle's stream. */
int mpcb_solve(mpcb_handle* h, int32_t B,
               const double* x0, const double* xs,
               const double* obs, int32_t obs_kind,
               const double* z0,
               double* z, double* obj, int32_t* status, int32_t* iters, double* kkt,
               double* lam_g, double* lam_x);

/* Same, with every pointer a DEVICE pointer on the handle's device (buffers owned by the caller, e.g.
 * allocated with mpcb_dev_alloc or by any other HIP allocator).  Asynchronous on the handle's stream
 * unless `sync` != 0.  This is the entry bench.py times: inputs already resident in HBM. */
int mpcb_solve_device(mpcb_handle* h, int32_t B,
                      const double* d_x0, const double* d_xs,
                      const double* d_obs, int32_t obs_kind,
                      const double* d_z0,
                      double* d_z, double* d_obj, int32_t* d_status, int32_t* d_iters, double* d_kkt,
                      double* d_lam_g, double* d_lam_x, int32_t sync);

/* Launch lanes.  k = 1 (the default): every call is queued on the handle's one stream, in order.  k > 1: the handle owns k - 1
 * further streams, and asynchronous mpcb_solve_device calls (sync = 0) go to the lanes in turn, call j on lane j mod k: up to k
 * consecutive solves are in flight together, so the SIMDs that the slowest instances of one launch leave idle are filled by the
 * next launches — inside one handle.  Calls j and j + k share a lane and are ordered; calls closer than k apart may run
 * concurrently: they must not share output buffers (rotate k sets) nor consume each other's output (a warm-start chain needs
 * mpcb_sync, mpcb_event_*, or k = 1).  Work queued earlier on the handle (uploads, scene sampling) happens before a lane's
 * launch; every other entry point (mpcb_sync, mpcb_dev_download / _upload, mpcb_allgather, mpcb_stream_wait, mpcb_solve,
 * mpcb_closed_loop ...) waits for all lanes first and runs on the handle's own stream.  Results do not depend on k (bit-identical). */
#define MPCB_INFLIGHT_MAX 16
int mpcb_set_inflight(mpcb_handle* h, int32_t k);

/* Closed loop on the device: `steps` receding-horizon iterations of  solve -> apply U_0 with the plant
 * x0 <- x0 + T f(x0,U_0) -> shift warm start [-> advance obstacles]   (main_cbf_kin_c_sim.py:87-123,16-26;
 * main_cbf_kin_c_sim_pre.py:98-106; with model = MPCB_MODEL_DYN the loop of main_cbf_dyn_c_sim.py:75-108, plant = the dyn
 * model's own right-hand side).  Host pointers.
 *   obs_state [B, n_obs, 6] in/out
 *   obs_motion  MPCB_OBSMOVE_STATIC    obstacles never move, rows use obs_state as is      (main_cbf_kin_c_sim.py:55,99)
 *               MPCB_OBSMOVE_PREDICTED constant-velocity obstacles (Obs_prediction.py:27-30): predicted over the horizon
 *                                      for every solve, advanced one step per MPC step       (main_cbf_kin_c_sim_pre.py:98-106)
 *               MPCB_OBSMOVE_CURRENT   advanced one step per MPC step, rows use the current position at every node
 *   x_hist [B, steps+1, nx], u_hist [B, steps, 2] (may be NULL), status_hist [B, steps], iters_hist [B, steps] */
#define MPCB_OBSMOVE_STATIC    0
#define MPCB_OBSMOVE_PREDICTED 1
#define MPCB_OBSMOVE_CURRENT   2
/* flags (or-ed):
 *   MPCB_CL_HOLD_ON_FAILURE   an instance whose solve does not end with MPCB_ST_SOLVED applies the first control of its PREVIOUS
 *                             plan and keeps that plan, shifted, as the next warm start (hold-and-shift; prior art in the
 *                             reference tree: `reference code/MPC-D-CBF.py:341-353`).  Without it the failed iterate is applied,
 *                             which is what the reference's drivers do (they never read IPOPT's status, main_cbf_kin_c_sim.py:100-104).
 *   MPCB_CL_ADVANCE_FIRST_ONLY only obstacle 0 moves between MPC steps: main_cbf_kin_c_sim_pre.py:106 rebuilds its obstacle list
 *                             from the first predicted trajectory alone; without the flag every obstacle moves. */
#define MPCB_CL_HOLD_ON_FAILURE    1
#define MPCB_CL_ADVANCE_FIRST_ONLY 2
int mpcb_closed_loop(mpcb_handle* h, int32_t B, int32_t steps,
                     const double* x0, const double* xs, double* obs_state, int32_t obs_motion, int32_t flags,
                     double* x_hist, double* u_hist, int32_t* status_hist, int32_t* iters_hist);

/* ---- scene generation on the device -----------------------------------------------------------------------
 * Counter-based random scenes (Philox4x32-10 keyed by `seed`, counter = global scene index): scene i is the same whichever GPU,
 * batch or chunk it is generated in, so 8 GPUs draw disjoint slices of one Monte-Carlo population from (seed, first_index).
 * Distributions of SURVEY.md 8(d) (the samplers of mpc_motion_planning_amd/scenes.py, rejection included):
 *   MPCB_SCENES_C2  x0 = [U(0,30), U(-0.5,4.5), U(-0.1,0.1), U(5,25)], rejected inside the shipped obstacle's ellipse (h < 0.05);
 *                   xs = [400,3.5,0,30], obs = [50,3.5,0,8,4.8,1.8]                       (main_cbf_kin_c_sim.py:45-55)
 *   MPCB_SCENES_C3  the same x0; n_obs moving obstacles x U(30,120), y in {0,3.5} +- 0.3, theta 0, v U(5,15), 4.8 x 1.8 m,
 *                   rejected when they overlap each other or the ego (also the scenes of C5)
 *   MPCB_SCENES_C4  dynamic bicycle: x0 = [U(0,30), U(-0.5,4.5), U(-0.05,0.05), U(8,20), 0, 0], xs = [600,3.5,0,15,0,0],
 *                   n_obs static obstacles x U(40,200), y in {-3.5,0,3.5,7} +- 0.3 (main_cbf_dyn_c_sim.py:44-51)
 * Device pointers x0 [B,nx], xs [B,nx], obs [B,n_obs,6]; asynchronous on the handle's stream. */
#define MPCB_SCENES_C2 2
#define MPCB_SCENES_C3 3
#define MPCB_SCENES_C4 4
int mpcb_sample_scenes(mpcb_handle* h, int32_t kind, int32_t B, uint64_t seed, uint64_t first_index,
                       double* d_x0, double* d_xs, double* d_obs);
/* mpcb_closed_loop on scenes drawn on the device (nothing but the histories crosses PCIe).  x0_out [B,nx] and obs0_out
 * [B,n_obs,6] (may be NULL) return the scenes that were drawn. */
int mpcb_closed_loop_sampled(mpcb_handle* h, int32_t kind, int32_t B, uint64_t seed, uint64_t first_index, int32_t steps,
                             int32_t obs_motion, int32_t flags, double* x0_out, double* obs0_out,
                             double* x_hist, double* u_hist, int32_t* status_hist, int32_t* iters_hist);
/* Device twins of the reference's scene helpers (host pointers in and out; the work is done by the device kernels the closed
 * loop uses):
 *   mpcb_predict_obstacles  constant-velocity roll-out [n,6] -> [n,N+1,6]            Obs_prediction.py:3-40
 *   mpcb_ref_path_window    straight 1 m-spaced global path from x_start to xs[0] at (xs[1], xs[2], xs[3]); per instance the
 *                           nearest path point searched from last_idx - 5 (first local minimum of the distance) and the
 *                           (N_p+1)-point preview window resampled from there; last_idx is updated   RefPathGenerator.py:9-59 */
int mpcb_predict_obstacles(mpcb_handle* h, int32_t n, int32_t N, double dt, const double* obs, double* traj);
int mpcb_ref_path_window(mpcb_handle* h, int32_t B, double x_start, const double* x0, const double* xs, double T_horizon, double dt,
                         int32_t* last_idx, double* window);

/* ---- multi-GPU ---------------------------------------------------------------------------------------------
 * Instances are independent NLPs (every `solver(...)` call of main_cbf_kin_c_sim.py:100 stands alone), so a batch is cut into
 * contiguous shards, one per GPU, with NO data-path collective; one RCCL all-gather of the converged trajectories over xGMI
 * afterwards gives every GPU all of them.  RCCL is called from inside this library (librccl is loaded on first use); no
 * torch.distributed, no MPI.  Two ways to form the group:
 *   (a) one process per GPU (the launch bench.py is given: RANK / WORLD_SIZE / MASTER_* in the environment):
 *       rank 0 calls mpcb_comm_unique_id, the 128 bytes travel to the other ranks over any host channel, every rank calls
 *       mpcb_comm_init_rank on its handle (ncclCommInitRank).  mpcb_allgather / mpcb_allreduce then work on that handle.
 *   (b) one process driving n devices: mpcb_set_devices(h, ids, n) (ncclCommInitAll).  mpcb_solve on that handle cuts the
 *       host batch into n contiguous shards, solves them concurrently, all-gathers z so that every device holds all of it
 *       (mpcb_gathered_z), and returns the whole batch to the caller. */
#define MPCB_UNIQUE_ID_BYTES 128
/* contiguous shard [lo, hi) of `rank` among `world`: sizes differ by at most one, earlier ranks take the remainder (host only) */
int mpcb_shard_bounds(int64_t B, int32_t world, int32_t rank, int64_t* lo, int64_t* hi);
int mpcb_comm_unique_id(void* id128);
int mpcb_comm_init_rank(mpcb_handle* h, const void* id128, int32_t rank, int32_t world);
int mpcb_set_devices(mpcb_handle* h, const int32_t* device_ids, int32_t n);
int mpcb_comm_info(const mpcb_handle* h, int32_t* world, int32_t* rank);       /* world = 1 before any of the two calls above */
/* d_recv [world, count] <- every rank's d_send [count] (doubles, device pointers), asynchronous on the handle's stream */
int mpcb_allgather(mpcb_handle* h, const double* d_send, double* d_recv, uint64_t count);
/* in-place reduction of n host doubles over the ranks, op 0 = sum, 1 = max; synchronises the handle's stream (a barrier) */
int mpcb_allreduce(mpcb_handle* h, double* values, int32_t n, int32_t op);
/* (b) only: device pointer of the [B, nz] trajectories of the last mpcb_solve as gathered on device `index` of the group */
int mpcb_gathered_z(mpcb_handle* h, int32_t index, const double** d_z);

/* device memory helpers so that Python (ctypes, no torch) can keep batches resident */
int mpcb_dev_alloc(mpcb_handle* h, uint64_t bytes, void** dptr);
int mpcb_dev_free(mpcb_handle* h, void* dptr);
int mpcb_dev_upload(mpcb_handle* h, void* dptr, const void* src, uint64_t bytes);
int mpcb_dev_download(mpcb_handle* h, void* dst, const void* dptr, uint64_t bytes);
int mpcb_sync(mpcb_handle* h);
/* h's stream waits (on the device, the host does not block) for everything queued so far on other's stream: lets work of
 * several handles be chained, e.g. an all-gather on a communication handle behind the solve of a solver handle */
int mpcb_stream_wait(mpcb_handle* h, mpcb_handle* other);

/* Fine-grained ordering between handles of one device (e.g. solves on a solver handle, all-gathers on a communication handle,
 * a ring of z buffers): mpcb_event_record(h, slot) marks "everything queued so far on h, its lanes included" WITHOUT joining the
 * lanes, so later launches of h keep running ahead; mpcb_event_wait(h, other, slot) makes h's stream — and with it every later
 * lane launch of h — wait for other's mark `slot` (no-op if that slot was never recorded).  mpcb_stream_wait above is the coarse
 * form: everything queued so far on other. */
#define MPCB_EVENT_SLOTS 32
int mpcb_event_record(mpcb_handle* h, int32_t slot);
int mpcb_event_wait(mpcb_handle* h, const mpcb_handle* other, int32_t slot);

/* HIP-event timing of the solve kernel on the handle's stream since the last reset:
 * number of launches, total and last kernel milliseconds. */
int mpcb_timing(mpcb_handle* h, int32_t reset, int32_t* launches, double* total_ms, double* last_ms);

/* Diagnostic: solve ONE instance and return the per-iteration log  trace[(max_iter+1) * 8] =
 * {mu, scaled NLP error, theta, f, alpha_primal_max, alpha accepted, alpha_dual, delta_w} per iteration (host pointers).
 * Used by the parity tests to compare the iteration history with the oracle's. */
int mpcb_solve_trace(mpcb_handle* h, const double* x0, const double* xs, const double* obs, int32_t obs_kind, const double* z0,
                     double* z, int32_t* status, int32_t* iters, double* trace);

/* Model right-hand side f(x,u) on the host (the `mpc_solver.f` the drivers call for the plant step,
 * main_cbf_kin_c_sim.py:17).  Tiny, scalar, no device involved. */
int mpcb_model_rhs(const mpcb_config* cfg, const double* x, const double* u, double* xdot);

const char* mpcb_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MPCBATCH_H */
