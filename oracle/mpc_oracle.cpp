// oracle/mpc_oracle.cpp — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's library
// (oracle/_build/libmpcoracle.so).  The product path (mpc_motion_planning_amd/, libmpcbatch.so) never
// links, imports or calls it and has no CPU fallback.
//
// PARITY STATUS: "parity unpinned" versus the reference's own solver.  The arithmetic of the reference's hot
// path lives in the third-party `casadi` wheel (bundled IPOPT + MUMPS; no version pinned anywhere in the
// reference: no requirements.txt / setup.py / lockfile), which is absent from /root/reference, is not installed
// in this image and cannot be fetched (no network).  The reference holds no tests, golden vectors or result
// files for this path either.  What this file restates, scalar and dependency-free:
//   (1) the NLP exactly as the reference builds it — model, cost, rows, bounds and orderings —
//         CasaDi_MPC_Optimize_Multishoot/MPC_CBF_optimize_kin.py:84-134 (bounds), :136-255 (model/cost/rows)
//         CasaDi_MPC_Optimize_Multishoot/MPC_CBF_optimize_kin_pre.py:236-253 (predicted obstacles)
//         CasaDi_MPC_Optimize_Multishoot/MPC_CBF_optimize_dyn.py:85-135,137-250 (dynamic model)
//   (2) IPOPT's published algorithm (Waechter & Biegler, Math. Program. 106 (2006) 25-57: primal-dual
//       barrier method, monotone mu, fraction-to-boundary rule, filter line search, inertia correction,
//       gradient-based objective scaling, bound push / bound relaxation) with the option values the reference
//       passes (kin.py:252-253: max_iter 100, acceptable_tol 1e-8) and IPOPT's documented defaults otherwise.
// Results are pinned instead by an independent numpy KKT certificate (oracle/kkt_check.py: the NLP written a second time
// in the reference's flat z / g ordering, complex-step derivatives) and by the committed golden vectors.
//
// Deliberate differences from IPOPT (none changes the KKT point that is reached in a given basin):
//   - X_0 is eliminated (it is pinned by the rows X_0 - P[0:nx] = 0, kin.py:191); rows that only involve X_0
//     are constants and are reported with zero multipliers.
//   - the KKT system is solved by a Riccati recursion over the stages (state augmented with the previous
//     control because of the (U_i - U_{i-1}) cost and rate rows, kin.py:201-204,216) instead of MUMPS.
//   - restoration phase (cfg.restoration = 1): IPOPT's idea — minimise the l1 norm of the constraint violation plus a
//     proximity term with the same interior-point method, return to the main problem as soon as a point acceptable to the
//     (augmented) filter with 10 % less violation is found — in a form that keeps the stage structure: the shooting rows
//     stay hard (a Riccati sweep always satisfies their linearisation), the general inequality rows (rate, obstacle)
//     become elastic  c(w) - s - p + n = 0, p, n >= 0  with cost rho (p + n).  See Solver::restoration().
//     cfg.restoration = 0: a failed line search ends with MPCB_ST_LINESEARCH.
//   - multipliers of the slack equalities are eliminated (y_d = v_L - v_U); no constraint-row scaling
//     (every row gradient of this NLP is < max_gradient at sane scenes).
//
// Build: see oracle/Makefile  (g++ -O2 -fopenmp -shared -fPIC).

#include "../include/mpcbatch.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

extern "C" int mpco_dims(const mpcb_config* c, int32_t* nx, int32_t* nz, int32_t* ng);

namespace {

constexpr int NXM = MPCB_NX_MAX;      // 6
constexpr int NU = MPCB_NU;           // 2
constexpr int NVM = NXM + NU;         // model variables [X, U]
constexpr int NAM = NXM + NU;         // augmented state [X, Uprev]
constexpr int NWM = NAM + NU;         // stage variables [X, Uprev, U]
constexpr int NODES = MPCB_N_MAX + 1; // 64
constexpr int NOBM = MPCB_NOBS_MAX;
constexpr double INF = std::numeric_limits<double>::infinity();

// ---------------------------------------------------------------------------------------------------------
// Second-order forward-mode AD over NV independent variables (value, gradient, Hessian).  Used to
// differentiate the model right-hand sides exactly as the reference writes them, and to cross-check the
// hand-written kinematic derivatives.
// ---------------------------------------------------------------------------------------------------------
template <int NV>
struct D2 {
  double v;
  double g[NV];
  double h[NV][NV];
  D2() : v(0) { std::memset(g, 0, sizeof g); std::memset(h, 0, sizeof h); }
  D2(double c) : v(c) { std::memset(g, 0, sizeof g); std::memset(h, 0, sizeof h); }
  static D2 var(double x, int i) { D2 r(x); r.g[i] = 1.0; return r; }
  // r = phi(a) given phi, phi', phi''
  static D2 chain(const D2& a, double p, double p1, double p2) {
    D2 r; r.v = p;
    for (int i = 0; i < NV; ++i) r.g[i] = p1 * a.g[i];
    for (int i = 0; i < NV; ++i) for (int j = 0; j < NV; ++j) r.h[i][j] = p1 * a.h[i][j] + p2 * a.g[i] * a.g[j];
    return r;
  }
};
template <int NV> D2<NV> operator+(const D2<NV>& a, const D2<NV>& b) {
  D2<NV> r; r.v = a.v + b.v;
  for (int i = 0; i < NV; ++i) r.g[i] = a.g[i] + b.g[i];
  for (int i = 0; i < NV; ++i) for (int j = 0; j < NV; ++j) r.h[i][j] = a.h[i][j] + b.h[i][j];
  return r;
}
template <int NV> D2<NV> operator-(const D2<NV>& a, const D2<NV>& b) {
  D2<NV> r; r.v = a.v - b.v;
  for (int i = 0; i < NV; ++i) r.g[i] = a.g[i] - b.g[i];
  for (int i = 0; i < NV; ++i) for (int j = 0; j < NV; ++j) r.h[i][j] = a.h[i][j] - b.h[i][j];
  return r;
}
template <int NV> D2<NV> operator-(const D2<NV>& a) { return D2<NV>(0.0) - a; }
template <int NV> D2<NV> operator*(const D2<NV>& a, const D2<NV>& b) {
  D2<NV> r; r.v = a.v * b.v;
  for (int i = 0; i < NV; ++i) r.g[i] = a.g[i] * b.v + a.v * b.g[i];
  for (int i = 0; i < NV; ++i) for (int j = 0; j < NV; ++j)
    r.h[i][j] = a.h[i][j] * b.v + a.v * b.h[i][j] + a.g[i] * b.g[j] + a.g[j] * b.g[i];
  return r;
}
template <int NV> D2<NV> operator/(const D2<NV>& a, const D2<NV>& b) {
  double ib = 1.0 / b.v;
  D2<NV> rb = D2<NV>::chain(b, ib, -ib * ib, 2.0 * ib * ib * ib);
  return a * rb;
}
template <int NV> D2<NV> operator*(double c, const D2<NV>& a) { return D2<NV>(c) * a; }
template <int NV> D2<NV> operator*(const D2<NV>& a, double c) { return a * D2<NV>(c); }
template <int NV> D2<NV> operator/(const D2<NV>& a, double c) { return a * D2<NV>(1.0 / c); }
template <int NV> D2<NV> operator/(double c, const D2<NV>& a) { return D2<NV>(c) / a; }
template <int NV> D2<NV> operator+(const D2<NV>& a, double c) { return a + D2<NV>(c); }
template <int NV> D2<NV> operator+(double c, const D2<NV>& a) { return a + D2<NV>(c); }
template <int NV> D2<NV> operator-(const D2<NV>& a, double c) { return a - D2<NV>(c); }
template <int NV> D2<NV> operator-(double c, const D2<NV>& a) { return D2<NV>(c) - a; }
template <int NV> D2<NV> sin(const D2<NV>& a) { return D2<NV>::chain(a, std::sin(a.v), std::cos(a.v), -std::sin(a.v)); }
template <int NV> D2<NV> cos(const D2<NV>& a) { return D2<NV>::chain(a, std::cos(a.v), -std::sin(a.v), -std::cos(a.v)); }
template <int NV> D2<NV> tan(const D2<NV>& a) {
  double t = std::tan(a.v), s2 = 1.0 + t * t;
  return D2<NV>::chain(a, t, s2, 2.0 * t * s2);
}
using std::cos; using std::sin; using std::tan;

// ---------------------------------------------------------------------------------------------------------
// Model right-hand sides, written once for double and for D2 (the reference's CasADi expressions).
// ---------------------------------------------------------------------------------------------------------
// CMOM/MPC_CBF_optimize_kin.py:153-156   rhs = [vx cos(phi), vx sin(phi), vx tan(df)/Veh_l, ax]
template <class S>
void rhs_kin(const mpcb_config& c, const S* x, const S* u, S* o) {
  o[0] = x[3] * cos(x[2]);
  o[1] = x[3] * sin(x[2]);
  o[2] = x[3] * tan(u[0]) / c.veh_l;
  o[3] = u[1];
}
// CMOM/MPC_CBF_optimize_dyn.py:156-170
template <class S>
void rhs_dyn(const mpcb_config& c, const S* x, const S* u, S* o) {
  const S &phi = x[2], &vx = x[3], &vy = x[4], &r = x[5], &df = u[0], &ax = u[1];
  S alpha_f = df - (vy + c.veh_lf * r) / vx;
  S alpha_r = -(vy - c.veh_lr * r) / vx;
  S Cf = (c.Fymax_f * 2.0 * c.aopt_f) / (c.aopt_f * c.aopt_f + alpha_f * alpha_f);
  S Cr = (c.Fymax_r * 2.0 * c.aopt_r) / (c.aopt_r * c.aopt_r + alpha_r * alpha_r);
  S Fcf = -(Cf * alpha_f);
  S Fcr = -(Cr * alpha_r);
  o[0] = vx * cos(phi) - vy * sin(phi);
  o[1] = vx * sin(phi) + vy * cos(phi);
  o[2] = r;
  o[3] = ax + r * vy;
  o[4] = -(r * vx) + (2.0 / c.veh_m) * (Fcf * cos(df) + Fcr);
  o[5] = (2.0 / c.veh_Iz) * (c.veh_lf * Fcf - c.veh_lr * Fcr);
}

inline int nx_of(const mpcb_config& c) { return c.model == MPCB_MODEL_DYN ? 6 : 4; }

void rhs_any(const mpcb_config& c, const double* x, const double* u, double* o) {
  if (c.model == MPCB_MODEL_DYN) rhs_dyn<double>(c, x, u, o); else rhs_kin<double>(c, x, u, o);
}

// One shooting step X+ = Phi(X, U; T): explicit Euler (what the reference's NLP and plant use, kin.py:207 / dyn.py:227), or the
// classical fourth-order Runge-Kutta step with the control held over the interval (cfg.integrator = MPCB_INT_RK4: BASELINE's
// north_star names it; the only Runge-Kutta code in the reference tree is the scratch Reference/MPC/sim_test.py:35-37).  Written once
// for double and for the AD type, so that Jacobians and Hessians of the RK4 step come from four differentiated rhs evaluations.
template <class S>
void step_model(const mpcb_config& c, double T, const S* x, const S* u, S* out) {
  const int nx = c.model == MPCB_MODEL_DYN ? 6 : 4;
  auto rhs = [&](const S* xx, S* o) { if (c.model == MPCB_MODEL_DYN) rhs_dyn<S>(c, xx, u, o); else rhs_kin<S>(c, xx, u, o); };
  S k1[NXM];
  rhs(x, k1);
  if (c.integrator != MPCB_INT_RK4) { for (int i = 0; i < nx; ++i) out[i] = x[i] + T * k1[i]; return; }
  S k2[NXM], k3[NXM], k4[NXM], xt[NXM];
  for (int i = 0; i < nx; ++i) xt[i] = x[i] + (0.5 * T) * k1[i];
  rhs(xt, k2);
  for (int i = 0; i < nx; ++i) xt[i] = x[i] + (0.5 * T) * k2[i];
  rhs(xt, k3);
  for (int i = 0; i < nx; ++i) xt[i] = x[i] + T * k3[i];
  rhs(xt, k4);
  for (int i = 0; i < nx; ++i) out[i] = x[i] + (T / 6.0) * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
}
void step_any(const mpcb_config& c, double T, const double* x, const double* u, double* out) { step_model<double>(c, T, x, u, out); }

struct ModelEval {
  double F[NXM];          // X+ = Phi(X, U; T): X + T f(X,U) (explicit Euler, kin.py:207 / dyn.py:227) or the RK4 step
  double A[NXM][NXM];     // dF/dX
  double B[NXM][NU];      // dF/dU
};

// F, A, B and (optionally) Hc = sum_a lam[a] * d2F_a/d[X,U]^2 by AD
void model_eval_ad(const mpcb_config& c, double T, const double* X, const double* U, const double* lam, ModelEval& me,
                   double Hc[NVM][NVM]) {
  const int nx = nx_of(c);
  typedef D2<NVM> S;
  S x[NXM], u[NU], o[NXM];
  for (int i = 0; i < nx; ++i) x[i] = S::var(X[i], i);
  for (int i = 0; i < NU; ++i) u[i] = S::var(U[i], nx + i);
  step_model<S>(c, T, x, u, o);                      // o = Phi(x, u) with first and second derivatives
  for (int a = 0; a < nx; ++a) {
    me.F[a] = o[a].v;
    for (int j = 0; j < nx; ++j) me.A[a][j] = o[a].g[j];
    for (int j = 0; j < NU; ++j) me.B[a][j] = o[a].g[nx + j];
  }
  if (Hc) {
    for (int i = 0; i < NVM; ++i) for (int j = 0; j < NVM; ++j) Hc[i][j] = 0.0;
    if (lam)
      for (int a = 0; a < nx; ++a)
        for (int i = 0; i < nx + NU; ++i) for (int j = 0; j < nx + NU; ++j) Hc[i][j] += lam[a] * o[a].h[i][j];
  }
}

// hand-written kinematic derivatives (what the HIP kernel also codes); checked against AD in tests
void model_eval_kin(const mpcb_config& c, double T, const double* X, const double* U, const double* lam, ModelEval& me,
                    double Hc[NVM][NVM]) {
  const double il = 1.0 / c.veh_l;
  const double phi = X[2], v = X[3], df = U[0], a = U[1];
  const double sp = std::sin(phi), cp = std::cos(phi), td = std::tan(df), sec2 = 1.0 + td * td;
  me.F[0] = X[0] + T * v * cp;
  me.F[1] = X[1] + T * v * sp;
  me.F[2] = X[2] + T * v * td * il;
  me.F[3] = X[3] + T * a;
  for (int i = 0; i < NXM; ++i) for (int j = 0; j < NXM; ++j) me.A[i][j] = (i == j) ? 1.0 : 0.0;
  for (int i = 0; i < NXM; ++i) for (int j = 0; j < NU; ++j) me.B[i][j] = 0.0;
  me.A[0][2] = -T * v * sp; me.A[0][3] = T * cp;
  me.A[1][2] = T * v * cp;  me.A[1][3] = T * sp;
  me.A[2][3] = T * td * il;
  me.B[2][0] = T * v * sec2 * il;
  me.B[3][1] = T;
  if (Hc) {
    for (int i = 0; i < NVM; ++i) for (int j = 0; j < NVM; ++j) Hc[i][j] = 0.0;
    if (lam) {
      // variable order [x, y, phi, v, df, a]
      Hc[2][2] = T * (-lam[0] * v * cp - lam[1] * v * sp);
      Hc[2][3] = Hc[3][2] = T * (-lam[0] * sp + lam[1] * cp);
      Hc[3][4] = Hc[4][3] = T * lam[2] * sec2 * il;
      Hc[4][4] = T * lam[2] * v * 2.0 * td * sec2 * il;
    }
  }
}

// T = step length of the stage: cfg.T, or the stage's entry of the time grid (mpcb_set_time_grid)
void model_eval(const mpcb_config& c, double T, const double* X, const double* U, const double* lam, ModelEval& me,
                double Hc[NVM][NVM], bool force_ad = false) {
  if (c.model == MPCB_MODEL_KIN && !force_ad && c.integrator != MPCB_INT_RK4) model_eval_kin(c, T, X, U, lam, me, Hc);
  else model_eval_ad(c, T, X, U, lam, me, Hc);
}

// ---------------------------------------------------------------------------------------------------------
// Inequality item: value c(w) with  L <= c <= U,  slack s (for a variable box s IS the variable),
// duals vL, vU of s - L >= 0 and U - s >= 0.
// ---------------------------------------------------------------------------------------------------------
struct Ineq {
  bool on = false, hasL = false, hasU = false;
  double L = -INF, U = INF;
  double s = 0, vL = 0, vU = 0;
  double r = 0;                  // c(w) - s   (0 for boxes)
  double ds = 0, dvL = 0, dvU = 0;
  double tL = 0, tU = 0;         // complementarity targets of the current solve
  // gradient wrt the stage vector [X, Uprev, U]: at most two nonzeros (i0,i1) — four (x, y, phi, v) for a general-gamma
  // discrete-CBF row, which also carries its own 4x4 second derivative hc
  int i0 = -1, i1 = -1, i2 = -1, i3 = -1; double g0 = 0, g1 = 0, g2 = 0, g3 = 0;
  double hc[4][4] = {{0}};
  // restoration phase only: elastic variables of the row  c(w) - s - p + n = 0  and their duals / steps
  bool el = false;
  double p = 0, n = 0, vp = 0, vn = 0, dp = 0, dn = 0, dvp = 0, dvn = 0;
  int idx(int a) const { return a == 0 ? i0 : a == 1 ? i1 : a == 2 ? i2 : i3; }
  double gv(int a) const { return a == 0 ? g0 : a == 1 ? g1 : a == 2 ? g2 : g3; }
  double y() const { return (hasL ? vL : 0.0) - (hasU ? vU : 0.0); }
};

struct ObsP { double ox, oy, ix2, iy2; };   // centre and 1/sX^2, 1/sY^2

struct Opt {   // line-search / barrier constants of IPOPT (Waechter-Biegler 2006, and IPOPT's option defaults)
  double kappa_eps = 10.0, kappa_mu = 0.2, theta_mu = 1.5, tau_min = 0.99;
  double gamma_theta = 1e-5, gamma_phi = 1e-8, delta = 1.0, s_theta = 1.1, s_phi = 2.3, eta_phi = 1e-8;
  double gamma_alpha = 0.05, kappa_sigma = 1e10, s_max = 100.0;
  double dw_first = 1e-4, dw_min = 1e-20, dw_max = 1e40, kw_minus = 1.0 / 3.0, kw_plus = 8.0, kw_plus_first = 100.0;
  // restoration phase
  double resto_rho = 1000.0;       // IPOPT resto_penalty_parameter
  double resto_kappa = 0.5;        // leave restoration when the original violation is <= resto_kappa * violation at entry (IPOPT: 0.9)
  double gap_safety = 10.0;        // safety factor on the barrier duality gap n_v * mu in the local-infeasibility certificate
  int resto_max_calls = 3, resto_max_iters = 40;   // per instance: entries into the phase and iterations spent inside it; beyond -> Restoration_Failed
  int trig_k = 5; double trig_alpha = 0.05, trig_theta = 0.8;   // early entry: trig_k consecutive accepted steps < trig_alpha with
                                                                //   theta reduced by less than the factor trig_theta over them
};

struct Solver {
  const mpcb_config& c;
  Opt o;
  int N, nx, na, nw, nobs;
  const double *x0, *xs;
  ObsP obs[NODES][NOBM];
  bool obs_node[NODES];
  double Tk[NODES];                // step length of stage k: cfg.T, or the time grid (the variable-time grid of kin.py:19-25 made effective)

  // iterate
  double X[NODES][NXM], U[NODES][NU], lam[NODES][NXM];
  Ineq bU[NODES][NU], bX[NODES][NXM], rR[NODES][NU], rO[NODES][NOBM];
  double os = 1.0, mu = 0.1, tau = 0.99;

  // the objective of the current phase:  sum_k sum_i Qc[k][i] (X_k,i - Xr[k][i])^2 + Rc[k][i] (U_k,i - Ur[k][i])^2 + DRc[i] (dU)^2,
  // scaled by osc.  Main phase: Q, xs, R, 0, DR of the reference (kin.py:168-205), osc = os.  Restoration phase: the proximity
  // term zeta/2 ||D_R (w - w_R)||^2 of IPOPT's restoration problem, osc = 1, plus rho (p + n) over the elastic rows.
  double Qc[NODES][NXM], Xr[NODES][NXM], Rc[NODES][NU], Ur[NODES][NU], DRc[NU], osc = 1.0;
  bool resto = false;
  double rho = 1000.0;                 // IPOPT resto_penalty_parameter
  double XR[NODES][NXM], UR[NODES][NU];   // reference point of the proximity term (the iterate at which restoration started)
  int n_resto_calls = 0, n_resto_iters = 0;

  // evaluation at the iterate
  ModelEval me[NODES];
  double dfc[NODES][NXM];          // defect F_k - X_{k+1}
  double fval = 0, theta = 0;

  // step
  double dX[NODES][NXM], dU[NODES][NU], lamF[NODES][NXM];
  double Kg[NODES][NU][NAM], kf[NODES][NU], P[NODES][NAM][NAM], p[NODES][NAM], Mi[NODES][3];
  double H[NODES][NWM][NWM], g[NODES][NWM];
  double dw_last = 0.0, dw_used = 0.0;
  bool debug = std::getenv("MPCO_DEBUG") != nullptr;

  std::vector<std::pair<double, double>> filter;
  double theta_max = 0, theta_min = 0;

  int iters = 0, status = MPCB_ST_MAXITER;
  double err0 = 0;
  int n_dyn_eval = 0, n_factor = 0, n_trial = 0;

  Solver(const mpcb_config& cfg) : c(cfg) {
    if (std::getenv("MPCO_KSIG")) o.kappa_sigma = std::atof(std::getenv("MPCO_KSIG"));
    if (std::getenv("MPCO_KMU")) o.kappa_mu = std::atof(std::getenv("MPCO_KMU"));
    if (std::getenv("MPCO_THMU")) o.theta_mu = std::atof(std::getenv("MPCO_THMU"));
    if (std::getenv("MPCO_KEPS")) o.kappa_eps = std::atof(std::getenv("MPCO_KEPS"));
    if (std::getenv("MPCO_TAUMIN")) o.tau_min = std::atof(std::getenv("MPCO_TAUMIN"));
    if (std::getenv("MPCO_KRESTO")) o.resto_kappa = std::atof(std::getenv("MPCO_KRESTO"));
    if (std::getenv("MPCO_GAP")) o.gap_safety = std::atof(std::getenv("MPCO_GAP"));
    if (std::getenv("MPCO_TRIG_K")) o.trig_k = std::atoi(std::getenv("MPCO_TRIG_K"));
    if (std::getenv("MPCO_TRIG_A")) o.trig_alpha = std::atof(std::getenv("MPCO_TRIG_A"));
    if (std::getenv("MPCO_TRIG_TH")) o.trig_theta = std::atof(std::getenv("MPCO_TRIG_TH"));
    rho = o.resto_rho;
    N = c.N; nx = nx_of(c); na = nx + NU; nw = na + NU; nobs = c.n_obs;
  }

  // ----- item enumeration --------------------------------------------------------------------------------
  template <class F> void each_item(int k, F&& f) {
    if (k < N) for (int i = 0; i < NU; ++i) if (bU[k][i].on) f(bU[k][i]);
    if (k >= 1) for (int i = 0; i < nx; ++i) if (bX[k][i].on) f(bX[k][i]);
    if (k >= 1 && k < N) for (int i = 0; i < NU; ++i) if (rR[k][i].on) f(rR[k][i]);
    if (k >= 1 && obs_node[k]) for (int j = 0; j < nobs; ++j) if (rO[k][j].on) f(rO[k][j]);
  }

  double hval(int k, int j, const double* Xk) const {
    const ObsP& q = obs[k][j];
    double dx = Xk[0] - q.ox, dy = Xk[1] - q.oy;
    return dx * dx * q.ix2 + dy * dy * q.iy2 - 1.0;
  }
  // General discrete-CBF row (kin.py:248, commented form, 0 < gamma < 1):  gamma h(X_i) + h(X_{i+1}) - h(X_i) >= 0 with the
  // stage-i obstacle in both terms.  On the feasible set X_{i+1} = F(X_i,U_i), and the position part of the Euler step depends
  // on X_i only, so the row is the STATE constraint  c_i(X_i) = h(X_i + T f(X_i)) - (1 - gamma) h(X_i) >= gamma hmin  of node i:
  // same feasible set, same minimisers; the multipliers of the reference form follow from these (see write_outputs).
  bool gen() const { return c.model == MPCB_MODEL_KIN && c.obs_mode == MPCB_OBS_DCBF && c.gamma < 1.0 - 1e-12; }
  template <class S> S crow(int k, int j, const S* x) const {   // x = (x, y, phi, v)
    const ObsP& q = obs[k][j];
    S qx = x[0] + Tk[k] * (x[3] * cos(x[2])), qy = x[1] + Tk[k] * (x[3] * sin(x[2]));
    S a = qx - q.ox, b = qy - q.oy, d = x[0] - q.ox, e = x[1] - q.oy;
    S hq = a * a * q.ix2 + b * b * q.iy2 - 1.0, hp = d * d * q.ix2 + e * e * q.iy2 - 1.0;
    return hq - (1.0 - c.gamma) * hp;
  }
  double rowval(int k, int j, const double* Xk) const { return gen() ? crow<double>(k, j, Xk) : hval(k, j, Xk); }
  void row_derivs(int k, int j, const double* Xk, Ineq& it) const {
    const ObsP& q = obs[k][j];
    if (!gen()) { it.g0 = 2 * (Xk[0] - q.ox) * q.ix2; it.g1 = 2 * (Xk[1] - q.oy) * q.iy2; return; }
    D2<4> x[4]; for (int i = 0; i < 4; ++i) x[i] = D2<4>::var(Xk[i], i);
    D2<4> v = crow<D2<4>>(k, j, x);
    it.g0 = v.g[0]; it.g1 = v.g[1]; it.g2 = v.g[2]; it.g3 = v.g[3];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) it.hc[a][b] = v.h[a][b];
  }

  static void relax(const mpcb_config& c, Ineq& it) {
    if (it.hasL) it.L -= c.bound_relax * std::max(1.0, std::fabs(it.L));
    if (it.hasU) it.U += c.bound_relax * std::max(1.0, std::fabs(it.U));
  }
  double push(const Ineq& it, double v) const {
    const double k1 = c.bound_push, k2 = c.bound_frac;
    if (it.hasL && it.hasU) {
      double pl = std::min(k1 * std::max(1.0, std::fabs(it.L)), k2 * (it.U - it.L));
      double pu = std::min(k1 * std::max(1.0, std::fabs(it.U)), k2 * (it.U - it.L));
      v = std::max(v, it.L + pl); v = std::min(v, it.U - pu);
    } else if (it.hasL) v = std::max(v, it.L + k1 * std::max(1.0, std::fabs(it.L)));
    else if (it.hasU) v = std::min(v, it.U - k1 * std::max(1.0, std::fabs(it.U)));
    return v;
  }
  static void setup(Ineq& it, double L, double U) {
    it = Ineq();
    it.hasL = std::isfinite(L); it.hasU = std::isfinite(U);
    it.on = it.hasL || it.hasU; it.L = L; it.U = U;
  }

  // ----- problem set-up ----------------------------------------------------------------------------------
  // obstacles: static [nobs][6] or predicted [nobs][N+1][6]; rows [x,y,theta,v,l,w]
  // zeros_start: the second attempt of cfg.second_start (only after a roll-out start) — z = 0 whatever z0 says (the reference's own
  // first-step start, main_cbf_kin_c_sim.py:47-50; dynamic model: 0 except vx = x0's, the tyre model divides by vx), no roll-out
  bool init(const double* x0_, const double* xs_, const double* ob, int obs_kind, const double* z0, const double* tgrid, bool zeros_start = false) {
    const bool rollout = c.init_rollout && !zeros_start;
    if (zeros_start) z0 = nullptr;
    x0 = x0_; xs = xs_;
    for (int k = 0; k < NODES; ++k) Tk[k] = tgrid ? tgrid[k < N ? k : N - 1] : c.T;
    const int last_row = c.obs_terminal ? N : N - 1;       // reference row index range 0..last_row
    for (int k = 0; k <= N; ++k) {
      obs_node[k] = false;
      int step;                                             // obstacle sample used by the row at node k
      if (c.obs_mode == MPCB_OBS_KEEPOUT) { obs_node[k] = (k <= last_row); step = k; }
      else if (gen()) { obs_node[k] = (k <= N - 1); step = k; }             // row i is c_i(X_i), i = 0..N-1
      else { obs_node[k] = (k >= 1 && k - 1 <= last_row); step = k - 1; }   // gamma = 1: row i is h_i(X_{i+1})
      if (!obs_node[k] || nobs == 0) continue;
      for (int j = 0; j < nobs; ++j) {
        const double* q = (obs_kind == MPCB_OBSIN_PREDICTED) ? ob + ((size_t)j * (N + 1) + step) * 6 : ob + (size_t)j * 6;
        double sx = c.obs_sx_fixed > 0 ? c.obs_sx_fixed : c.ego_hl + q[4] / 2 + c.safe_disl;   // kin.py:242
        double sy = c.obs_sy_fixed > 0 ? c.obs_sy_fixed : c.ego_hw + q[5] / 2 + c.safe_disw;   // kin.py:243
        obs[k][j] = ObsP{q[0], q[1], 1.0 / (sx * sx), 1.0 / (sy * sy)};
      }
    }
    // iterate from z0 (reference order, kin.py:250), X_0 pinned
    for (int k = 0; k < N; ++k) for (int i = 0; i < NU; ++i) U[k][i] = z0 ? z0[NU * k + i] : 0.0;
    for (int k = 0; k <= N; ++k) for (int i = 0; i < nx; ++i) X[k][i] = z0 ? z0[NU * N + nx * k + i] : 0.0;
    if (zeros_start && c.model == MPCB_MODEL_DYN) for (int k = 0; k <= N; ++k) X[k][3] = x0[3];
    double X0guess[NXM]; for (int i = 0; i < nx; ++i) X0guess[i] = X[0][i];
    for (int i = 0; i < nx; ++i) X[0][i] = x0[i];
    for (int i = 0; i < NU; ++i) U[N][i] = 0.0;
    std::memset(lam, 0, sizeof lam);

    // objective scaling at the user's start (IPOPT gradient-based scaling, nlp_scaling_max_gradient)
    {
      double gmax = 0;
      for (int k = 0; k < N; ++k) {
        const double* Xk = (k == 0) ? X0guess : X[k];
        for (int i = 0; i < nx; ++i) gmax = std::max(gmax, std::fabs(2 * c.Q[i] * (Xk[i] - xs[i])));
        for (int i = 0; i < NU; ++i) {
          double gu = 2 * c.R[i] * U[k][i];
          const double* Up = (k == 0) ? c.u_last : U[k - 1];
          if (k > 0 || c.du0_cost) gu += 2 * c.DR[i] * (U[k][i] - Up[i]);
          if (k + 1 < N) gu -= 2 * c.DR[i] * (U[k + 1][i] - U[k][i]);
          gmax = std::max(gmax, std::fabs(gu));
        }
      }
      os = (gmax > c.max_gradient) ? c.max_gradient / gmax : 1.0;
    }

    // optional roll-out of X from x0 with the guessed (clipped) controls
    auto roll_out = [&]() {
      for (int k = 0; k < N; ++k) {
        double Uc[NU]; for (int i = 0; i < NU; ++i) Uc[i] = std::min(std::max(U[k][i], c.u_lo[i]), c.u_hi[i]);
        step_any(c, Tk[k], X[k], Uc, X[k + 1]);
      }
    };
    if (rollout) {
      roll_out();
      // cfg.start_steer (include/mpcbatch.h; the kernels' set-up does the same): a cold start whose straight roll-out passes an obstacle
      // row closer than h - obs_hmin < 1 is rolled out with a slight constant turn instead — away from the centre of that obstacle
      // (the first minimum of h over the nodes, then over the obstacles of a node), or to its other side when the y box has no room
      // for the row's ellipse on that side
      if (!z0 && c.start_steer > 0.0 && nobs > 0) {
        double hm = 1e300; int kb = -1, jb = -1;
        for (int k = 1; k <= N; ++k) if (obs_node[k]) {
          double hk = 1e300; int jk = 0;
          for (int j = 0; j < nobs; ++j) { const double hj = hval(k, j, X[k]) - c.obs_hmin; if (hj < hk) { hk = hj; jk = j; } }
          if (hk < hm) { hm = hk; kb = k; jb = jk; }
        }
        if (kb >= 0 && hm < 1.0) {
          const ObsP& q = obs[kb][jb];
          const double sy = 1.0 / std::sqrt(q.iy2);
          double sgn = X[kb][1] >= q.oy ? 1.0 : -1.0;
          const bool up = q.oy + sy <= c.x_hi[1], dn = q.oy - sy >= c.x_lo[1];
          if (sgn > 0 && !up && dn) sgn = -1.0; else if (sgn < 0 && !dn && up) sgn = 1.0;
          for (int k = 0; k < N; ++k) U[k][0] = sgn * c.start_steer;
          roll_out();
        }
      }
    }

    set_main_cost();
    // feasibility of the pinned node 0 (rows that IPOPT could never satisfy)
    for (int i = 0; i < nx; ++i)
      if (x0[i] < c.x_lo[i] - 1e-8 || x0[i] > c.x_hi[i] + 1e-8) { status = MPCB_ST_INFEASIBLE_X0; return false; }
    if (obs_node[0]) for (int j = 0; j < nobs; ++j)
      if (rowval(0, j, x0) < (gen() ? c.gamma : 1.0) * c.obs_hmin - 1e-8) { status = MPCB_ST_INFEASIBLE_X0; return false; }

    // boxes: relax (bound_relax_factor), push the start inside (bound_push / bound_frac), duals = 1
    for (int k = 0; k <= N; ++k) {
      for (int i = 0; i < NU; ++i) {
        setup(bU[k][i], c.u_lo[i], c.u_hi[i]);
        Ineq& it = bU[k][i]; it.on = it.on && k < N; if (!it.on) continue;
        relax(c, it); U[k][i] = push(it, U[k][i]); it.s = U[k][i]; it.vL = it.vU = 1.0; it.i0 = na + i; it.g0 = 1.0;
      }
      for (int i = 0; i < nx; ++i) {
        setup(bX[k][i], c.x_lo[i], c.x_hi[i]);
        Ineq& it = bX[k][i]; it.on = it.on && k >= 1; if (!it.on) continue;
        relax(c, it); X[k][i] = push(it, X[k][i]); it.s = X[k][i]; it.vL = it.vU = 1.0; it.i0 = i; it.g0 = 1.0;
      }
    }
    // general rows: slack = row value at the pushed start, pushed inside its own bounds
    for (int k = 0; k <= N; ++k) {
      for (int i = 0; i < NU; ++i) {
        const double sc = (k >= 1) ? Tk[k - 1] / c.T : 1.0;        // cfg.du_* are rate * cfg.T (kin.py:116-121); with a time grid: rate * T_{k-1}
        setup(rR[k][i], c.du_lo[i] * sc, c.du_hi[i] * sc);
        Ineq& it = rR[k][i]; it.on = it.on && k >= 1 && k < N; if (!it.on) continue;
        relax(c, it); it.s = push(it, U[k][i] - U[k - 1][i]); it.vL = it.vU = 1.0;
        it.i0 = na + i; it.g0 = 1.0; it.i1 = nx + i; it.g1 = -1.0;
      }
      for (int j = 0; j < NOBM; ++j) {
        Ineq& it = rO[k][j]; it = Ineq();
        if (!(j < nobs && k >= 1 && obs_node[k])) continue;
        setup(it, (gen() ? c.gamma : 1.0) * c.obs_hmin, INF);
        relax(c, it); it.s = push(it, rowval(k, j, X[k])); it.vL = 1.0; it.i0 = 0; it.i1 = 1;
        if (gen()) { it.i2 = 2; it.i3 = 3; }
      }
    }
    mu = c.mu_init; tau = std::max(o.tau_min, 1.0 - mu);
    set_main_cost();
    return true;
  }

  // ----- evaluation at the current iterate ------------------------------------------------------------------
  void set_main_cost() {                 // kin.py:168-205 / dyn.py:189-225
    for (int k = 0; k <= N; ++k) {
      for (int i = 0; i < NXM; ++i) { Qc[k][i] = (k < N && i < nx) ? c.Q[i] : 0.0; Xr[k][i] = (i < nx) ? xs[i] : 0.0; }   // no terminal cost
      for (int i = 0; i < NU; ++i) { Rc[k][i] = (k < N) ? c.R[i] : 0.0; Ur[k][i] = 0.0; }
    }
    for (int i = 0; i < NU; ++i) DRc[i] = c.DR[i];
    osc = os;
  }
  void set_resto_cost(double zeta) {     // zeta/2 * sum D^2 (w - w_R)^2,  D = 1 / max(1, |w_R|)
    for (int k = 0; k <= N; ++k) {
      for (int i = 0; i < NXM; ++i) {
        const double d = 1.0 / std::max(1.0, std::fabs(XR[k][i]));
        Qc[k][i] = (k >= 1 && i < nx) ? 0.5 * zeta * d * d : 0.0; Xr[k][i] = XR[k][i];
      }
      for (int i = 0; i < NU; ++i) {
        const double d = 1.0 / std::max(1.0, std::fabs(UR[k][i]));
        Rc[k][i] = (k < N) ? 0.5 * zeta * d * d : 0.0; Ur[k][i] = UR[k][i];
      }
    }
    for (int i = 0; i < NU; ++i) DRc[i] = 0.0;
    osc = 1.0;
  }
  bool du_cost(int k) const { return k > 0 || c.du0_cost; }

  // objective of the current phase without the elastic part (unscaled)
  double objective(const double Xa[][NXM], const double Ua[][NU]) const {
    double f = 0;
    for (int k = 0; k <= N; ++k) {
      for (int i = 0; i < nx; ++i) { double e = Xa[k][i] - Xr[k][i]; f += Qc[k][i] * e * e; }
      if (k < N) for (int i = 0; i < NU; ++i) {
        double e = Ua[k][i] - Ur[k][i];
        f += Rc[k][i] * e * e;
        if (du_cost(k)) { double d = Ua[k][i] - (k ? Ua[k - 1][i] : c.u_last[i]); f += DRc[i] * d * d; }
      }
    }
    return f;
  }
  double elastic_cost() const {
    double f = 0;
    if (resto) for (int k = 0; k <= N; ++k) const_cast<Solver*>(this)->each_item(k, [&](Ineq& it) { if (it.el) f += rho * (it.p + it.n); });
    return f;
  }

  void eval_point() {
    theta = 0;
    for (int k = 0; k < N; ++k) {
      model_eval(c, Tk[k], X[k], U[k], nullptr, me[k], nullptr); ++n_dyn_eval;
      for (int i = 0; i < nx; ++i) { dfc[k][i] = me[k].F[i] - X[k + 1][i]; theta += std::fabs(dfc[k][i]); }
    }
    for (int k = 0; k <= N; ++k) {
      for (int i = 0; i < NU; ++i) if (bU[k][i].on) { bU[k][i].s = U[k][i]; bU[k][i].r = 0; }
      for (int i = 0; i < nx; ++i) if (bX[k][i].on) { bX[k][i].s = X[k][i]; bX[k][i].r = 0; }
      for (int i = 0; i < NU; ++i) if (rR[k][i].on) { Ineq& it = rR[k][i]; it.r = (U[k][i] - U[k - 1][i]) - it.s - (it.el ? it.p - it.n : 0.0); theta += std::fabs(it.r); }
      for (int j = 0; j < nobs; ++j) if (rO[k][j].on) {
        Ineq& it = rO[k][j]; const ObsP& q = obs[k][j];
        it.r = rowval(k, j, X[k]) - it.s - (it.el ? it.p - it.n : 0.0); theta += std::fabs(it.r);
        (void)q; row_derivs(k, j, X[k], it);
      }
    }
    fval = objective(X, U) + elastic_cost();
  }

  double barrier_phi(double f_unscaled, double mu_) {
    double phi = osc * f_unscaled;
    for (int k = 0; k <= N; ++k) each_item(k, [&](Ineq& it) {
      if (it.hasL) phi -= mu_ * std::log(it.s - it.L);
      if (it.hasU) phi -= mu_ * std::log(it.U - it.s);
      if (it.el) phi -= mu_ * (std::log(it.p) + std::log(it.n));
    });
    return phi;
  }

  // dual residual of the scaled problem, per variable;  returns E_mu pieces
  struct Err { double dual = 0, prim = 0, comp = 0, sd = 1, sc = 1; };
  Err kkt_error(double mu_, double* dual_unscaled = nullptr) {
    Err e;
    double sum_lam = 0, sum_v = 0; int n_lam = 0, n_v = 0;
    for (int k = 0; k <= N; ++k) {
      double rX[NXM] = {0}, rU[NU] = {0};
      if (k >= 1) {
        for (int i = 0; i < nx; ++i) rX[i] += osc * 2 * Qc[k][i] * (X[k][i] - Xr[k][i]);
        for (int i = 0; i < nx; ++i) { rX[i] -= lam[k][i]; sum_lam += std::fabs(lam[k][i]); ++n_lam; }
        if (k < N) for (int i = 0; i < nx; ++i) for (int a = 0; a < nx; ++a) rX[i] += me[k].A[a][i] * lam[k + 1][a];
      }
      if (k < N) {
        for (int i = 0; i < NU; ++i) {
          rU[i] += osc * 2 * Rc[k][i] * (U[k][i] - Ur[k][i]);
          if (du_cost(k)) rU[i] += osc * 2 * DRc[i] * (U[k][i] - (k ? U[k - 1][i] : c.u_last[i]));
          if (k + 1 < N) rU[i] -= osc * 2 * DRc[i] * (U[k + 1][i] - U[k][i]);
          for (int a = 0; a < nx; ++a) rU[i] += me[k].B[a][i] * lam[k + 1][a];
          if (k + 1 < N && rR[k + 1][i].on) rU[i] += rR[k + 1][i].y();      // d(row k+1)/dU_k = -1
        }
      }
      each_item(k, [&](Ineq& it) {
        double y = it.y();
        auto add = [&](int idx, double gg) {
          if (idx < 0) return;
          if (idx < nx) rX[idx] -= y * gg;
          else if (idx >= na) rU[idx - na] -= y * gg;      // Uprev part is accounted at stage k-1 above
        };
        for (int a = 0; a < 4; ++a) add(it.idx(a), it.gv(a));
        if (it.hasL) { e.comp = std::max(e.comp, std::fabs((it.s - it.L) * it.vL - mu_)); sum_v += it.vL; ++n_v; }
        if (it.hasU) { e.comp = std::max(e.comp, std::fabs((it.U - it.s) * it.vU - mu_)); sum_v += it.vU; ++n_v; }
        if (it.el) {   // stationarity in p and n: rho + y - vp = 0, rho - y - vn = 0
          e.dual = std::max(e.dual, std::max(std::fabs(rho + y - it.vp), std::fabs(rho - y - it.vn)));
          e.comp = std::max(e.comp, std::max(std::fabs(it.p * it.vp - mu_), std::fabs(it.n * it.vn - mu_)));
          sum_v += it.vp + it.vn; n_v += 2;
        }
        e.prim = std::max(e.prim, std::fabs(it.r));
      });
      if (k >= 1) for (int i = 0; i < nx; ++i) e.dual = std::max(e.dual, std::fabs(rX[i]));
      if (k < N) for (int i = 0; i < NU; ++i) e.dual = std::max(e.dual, std::fabs(rU[i]));
      if (k < N) for (int i = 0; i < nx; ++i) e.prim = std::max(e.prim, std::fabs(dfc[k][i]));
    }
    e.sd = std::max(o.s_max, (sum_lam + sum_v) / std::max(1, n_lam + n_v)) / o.s_max;
    e.sc = std::max(o.s_max, sum_v / std::max(1, n_v)) / o.s_max;
    if (dual_unscaled) *dual_unscaled = e.dual / osc;
    return e;
  }
  double Emu(const Err& e) const { return std::max(e.dual / e.sd, std::max(e.prim, e.comp / e.sc)); }

  // ----- condensed stage QP --------------------------------------------------------------------------------
  // Per item the complementarity targets tL, tU: the plain barrier step uses tL = tU = mu; the affine
  // (predictor) step 0; Mehrotra's corrector mu_t -/+ ds_aff * dv_aff.
  // Hessian of the condensed stage QP (independent of mu and of the targets)
  void build_stage_matrix(int k, double dw) {
    double (*Hk)[NWM] = H[k];
    for (int i = 0; i < NWM; ++i) for (int j = 0; j < NWM; ++j) Hk[i][j] = 0;
    if (k >= 1) for (int i = 0; i < nx; ++i) Hk[i][i] += osc * 2 * Qc[k][i];
    if (k < N) {
      for (int i = 0; i < NU; ++i) {
        int iu = na + i, ip = nx + i;
        Hk[iu][iu] += osc * 2 * Rc[k][i];
        if (du_cost(k)) { double w2 = osc * 2 * DRc[i]; Hk[iu][iu] += w2; Hk[ip][ip] += w2; Hk[iu][ip] -= w2; Hk[ip][iu] -= w2; }
      }
      double Hc[NVM][NVM];
      ModelEval tmp;
      model_eval(c, Tk[k], X[k], U[k], lam[k + 1], tmp, Hc);
      auto map = [&](int i) { return i < nx ? i : na + (i - nx); };
      for (int i = 0; i < nx + NU; ++i) for (int j = 0; j < nx + NU; ++j) Hk[map(i)][map(j)] += Hc[i][j];
    }
    each_item(k, [&](Ineq& it) {
      double sig = 0;
      if (it.hasL) sig += it.vL / (it.s - it.L);
      if (it.hasU) sig += it.vU / (it.U - it.s);
      if (it.el) sig = sig / (1.0 + sig * it.p / it.vp + sig * it.n / it.vn);     // series combination with Sigma_p, Sigma_n
      for (int a = 0; a < 4; ++a) if (it.idx(a) >= 0)
        for (int b = 0; b < 4; ++b) if (it.idx(b) >= 0) Hk[it.idx(a)][it.idx(b)] += sig * it.gv(a) * it.gv(b);
    });
    if (k >= 1 && obs_node[k]) for (int j = 0; j < nobs; ++j) if (rO[k][j].on) {   // - y * d2h
      double y = rO[k][j].y();
      if (gen()) { for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) Hk[a][b] -= y * rO[k][j].hc[a][b]; }
      else { Hk[0][0] -= y * 2 * obs[k][j].ix2; Hk[1][1] -= y * 2 * obs[k][j].iy2; }
    }
    if (k >= 1) for (int i = 0; i < nx; ++i) Hk[i][i] += dw;
    if (k < N) for (int i = 0; i < NU; ++i) Hk[na + i][na + i] += dw;
  }

  // gradient of the condensed stage QP for the current targets
  void build_stage_grad(int k) {
    double* gk = g[k];
    for (int i = 0; i < NWM; ++i) gk[i] = 0;
    if (k >= 1) for (int i = 0; i < nx; ++i) gk[i] += osc * 2 * Qc[k][i] * (X[k][i] - Xr[k][i]);
    if (k < N) for (int i = 0; i < NU; ++i) {
      int iu = na + i, ip = nx + i;
      gk[iu] += osc * 2 * Rc[k][i] * (U[k][i] - Ur[k][i]);
      if (du_cost(k)) {
        double d = U[k][i] - (k ? U[k - 1][i] : c.u_last[i]), w2 = osc * 2 * DRc[i];
        gk[iu] += w2 * d; gk[ip] -= w2 * d;
      }
    }
    each_item(k, [&](Ineq& it) {
      double sig = 0, gb = 0;
      if (it.hasL) { sig += it.vL / (it.s - it.L); gb += it.tL / (it.s - it.L); }
      if (it.hasU) { sig += it.vU / (it.U - it.s); gb -= it.tU / (it.U - it.s); }
      if (!it.el) gb -= sig * it.r;
      else {         // elastic row: Sigma_e = kappa Sigma_s, residual r~ (see restoration())
        const double sp = it.vp / it.p, sn = it.vn / it.n, kap = 1.0 / (1.0 + sig / sp + sig / sn);
        const double rt = it.r + (rho + gb - it.tL / it.p) / sp + (gb - rho + it.tL / it.n) / sn;
        gb -= sig * kap * rt;
      }
      for (int a = 0; a < 4; ++a) if (it.idx(a) >= 0) gk[it.idx(a)] -= gb * it.gv(a);
    });
  }

  void stage_AB(int k, double AB[NAM][NWM]) const {
    for (int a = 0; a < na; ++a) for (int j = 0; j < nw; ++j) AB[a][j] = 0;
    for (int a = 0; a < nx; ++a) {
      for (int j = 0; j < nx; ++j) AB[a][j] = me[k].A[a][j];
      for (int j = 0; j < NU; ++j) AB[a][na + j] = me[k].B[a][j];
    }
    for (int j = 0; j < NU; ++j) AB[nx + j][na + j] = 1.0;
  }

  // matrix part of the backward Riccati sweep: P_k, K_k, inv(Muu_k).
  // false when some Muu is not positive definite (wrong inertia of the KKT matrix)
  bool riccati_matrix() {
    for (int i = 0; i < na; ++i) for (int j = 0; j < na; ++j) P[N][i][j] = H[N][i][j];
    for (int k = N - 1; k >= 0; --k) {
      double AB[NAM][NWM]; stage_AB(k, AB);
      double W[NAM][NWM];
      for (int a = 0; a < na; ++a)
        for (int j = 0; j < nw; ++j) { double s = 0; for (int b = 0; b < na; ++b) s += P[k + 1][a][b] * AB[b][j]; W[a][j] = s; }
      double M[NWM][NWM];
      for (int i = 0; i < nw; ++i)
        for (int j = 0; j < nw; ++j) { double s = H[k][i][j]; for (int a = 0; a < na; ++a) s += AB[a][i] * W[a][j]; M[i][j] = s; }
      double m11 = M[na][na], m12 = 0.5 * (M[na][na + 1] + M[na + 1][na]), m22 = M[na + 1][na + 1];
      double det = m11 * m22 - m12 * m12;
      if (!(m11 > 0) || !(det > 1e-14 * m11 * m22) || !std::isfinite(det)) return false;
      Mi[k][0] = m22 / det; Mi[k][1] = -m12 / det; Mi[k][2] = m11 / det;
      for (int j = 0; j < na; ++j) {
        Kg[k][0][j] = -(Mi[k][0] * M[na][j] + Mi[k][1] * M[na + 1][j]);
        Kg[k][1][j] = -(Mi[k][1] * M[na][j] + Mi[k][2] * M[na + 1][j]);
      }
      for (int i = 0; i < na; ++i)
        for (int j = 0; j < na; ++j) P[k][i][j] = M[i][j] + M[i][na] * Kg[k][0][j] + M[i][na + 1] * Kg[k][1][j];
      for (int i = 0; i < na; ++i) for (int j = i + 1; j < na; ++j) { double s = 0.5 * (P[k][i][j] + P[k][j][i]); P[k][i][j] = P[k][j][i] = s; }
    }
    return true;
  }

  // vector part: backward (p_k, kff_k) for the current gradient g, then the forward roll-out of the step
  void riccati_vector() {
    for (int i = 0; i < na; ++i) p[N][i] = g[N][i];
    for (int k = N - 1; k >= 0; --k) {
      double AB[NAM][NWM]; stage_AB(k, AB);
      double q[NAM], m[NWM];
      for (int a = 0; a < na; ++a) { double s = p[k + 1][a]; for (int b = 0; b < nx; ++b) s += P[k + 1][a][b] * dfc[k][b]; q[a] = s; }
      for (int i = 0; i < nw; ++i) { double s = g[k][i]; for (int a = 0; a < na; ++a) s += AB[a][i] * q[a]; m[i] = s; }
      kf[k][0] = -(Mi[k][0] * m[na] + Mi[k][1] * m[na + 1]);
      kf[k][1] = -(Mi[k][1] * m[na] + Mi[k][2] * m[na + 1]);
      for (int i = 0; i < na; ++i) p[k][i] = m[i] + Kg[k][0][i] * m[na] + Kg[k][1][i] * m[na + 1];
    }
    double dxa[NAM] = {0};
    for (int i = 0; i < nx; ++i) dX[0][i] = 0;
    for (int k = 0; k < N; ++k) {
      for (int i = 0; i < NU; ++i) { double s = kf[k][i]; for (int j = 0; j < na; ++j) s += Kg[k][i][j] * dxa[j]; dU[k][i] = s; }
      double nxt[NAM] = {0};
      for (int a = 0; a < nx; ++a) {
        double s = dfc[k][a];
        for (int j = 0; j < nx; ++j) s += me[k].A[a][j] * dxa[j];
        for (int j = 0; j < NU; ++j) s += me[k].B[a][j] * dU[k][j];
        nxt[a] = s;
      }
      for (int j = 0; j < NU; ++j) nxt[nx + j] = dU[k][j];
      for (int a = 0; a < na; ++a) dxa[a] = nxt[a];
      for (int a = 0; a < nx; ++a) {
        dX[k + 1][a] = dxa[a];
        double s = p[k + 1][a]; for (int b = 0; b < na; ++b) s += P[k + 1][a][b] * dxa[b]; lamF[k + 1][a] = s;
      }
    }
    // slack and dual steps
    for (int k = 0; k <= N; ++k) each_item(k, [&](Ineq& it) {
      auto comp = [&](int idx) -> double {
        if (idx < 0) return 0.0;
        if (idx < nx) return dX[k][idx];
        if (idx < na) return k ? dU[k - 1][idx - nx] : 0.0;
        return dU[k][idx - na];
      };
      it.ds = it.g0 * comp(it.i0) + it.g1 * comp(it.i1) + it.g2 * comp(it.i2) + it.g3 * comp(it.i3) + it.r;
      if (it.el) {
        double sig = 0, bs = 0;
        if (it.hasL) { sig += it.vL / (it.s - it.L); bs += it.tL / (it.s - it.L); }
        if (it.hasU) { sig += it.vU / (it.U - it.s); bs -= it.tU / (it.U - it.s); }
        const double sp = it.vp / it.p, sn = it.vn / it.n, kap = 1.0 / (1.0 + sig / sp + sig / sn);
        const double cp = (rho + bs - it.tL / it.p) / sp, cn = (bs - rho + it.tL / it.n) / sn;
        it.ds = kap * (it.ds + cp + cn);
        it.dp = sig * it.ds / sp - cp;
        it.dn = -sig * it.ds / sn + cn;
        it.dvp = it.tL / it.p - it.vp - sp * it.dp;
        it.dvn = it.tL / it.n - it.vn - sn * it.dn;
      }
      if (it.hasL) { double d = it.s - it.L; it.dvL = it.tL / d - it.vL - it.vL / d * it.ds; }
      if (it.hasU) { double d = it.U - it.s; it.dvU = it.tU / d - it.vU + it.vU / d * it.ds; }
    });
  }

  void set_targets(double t) {
    for (int k = 0; k <= N; ++k) each_item(k, [&](Ineq& it) { it.tL = t; it.tU = t; });
  }
  void solve_direction() { for (int k = 0; k <= N; ++k) build_stage_grad(k); riccati_vector(); }

  // KKT matrix factorisation with IPOPT's inertia-correction schedule for delta_w
  bool factorize() {
    double dw = 0.0; bool first_try = true;
    for (int tries = 0; tries < 60; ++tries) {
      for (int k = 0; k <= N; ++k) build_stage_matrix(k, dw);
      ++n_factor;
      if (riccati_matrix()) { if (dw > 0) dw_last = dw; dw_used = dw; return true; }
      if (first_try) { dw = (dw_last == 0.0) ? o.dw_first : std::max(o.dw_min, o.kw_minus * dw_last); first_try = false; }
      else dw *= (dw_last == 0.0) ? o.kw_plus_first : o.kw_plus;
      if (dw > o.dw_max) return false;
    }
    return false;
  }

  // ----- line search ------------------------------------------------------------------------------------------
  struct Trial { double theta, phi, f; bool ok; };
  Trial eval_trial(double alpha, double mu_) {
    static thread_local double Xt[NODES][NXM], Ut[NODES][NU];
    Trial t{0, 0, 0, true};
    for (int k = 0; k <= N; ++k) {
      for (int i = 0; i < nx; ++i) Xt[k][i] = X[k][i] + alpha * dX[k][i];
      for (int i = 0; i < NU; ++i) Ut[k][i] = (k < N) ? U[k][i] + alpha * dU[k][i] : 0.0;
    }
    for (int k = 0; k < N; ++k) {
      double F[NXM]; step_any(c, Tk[k], Xt[k], Ut[k], F); ++n_dyn_eval;
      for (int i = 0; i < nx; ++i) t.theta += std::fabs(F[i] - Xt[k + 1][i]);
    }
    t.f = objective(Xt, Ut);
    double phi = osc * t.f;
    for (int k = 0; k <= N; ++k) {
      auto bar = [&](const Ineq& it, double s) {
        if (it.hasL) { double d = s - it.L; if (!(d > 0)) t.ok = false; else phi -= mu_ * std::log(d); }
        if (it.hasU) { double d = it.U - s; if (!(d > 0)) t.ok = false; else phi -= mu_ * std::log(d); }
      };
      if (k < N) for (int i = 0; i < NU; ++i) if (bU[k][i].on) bar(bU[k][i], Ut[k][i]);
      if (k >= 1) for (int i = 0; i < nx; ++i) if (bX[k][i].on) bar(bX[k][i], Xt[k][i]);
      auto elastic = [&](const Ineq& it) -> double {      // p - n of the trial point, barrier and cost terms added to phi
        if (!it.el) return 0.0;
        const double pt = it.p + alpha * it.dp, nt = it.n + alpha * it.dn;
        if (!(pt > 0) || !(nt > 0)) { t.ok = false; return 0.0; }
        phi += rho * (pt + nt) - mu_ * (std::log(pt) + std::log(nt));
        t.f += rho * (pt + nt);
        return pt - nt;
      };
      if (k >= 1 && k < N) for (int i = 0; i < NU; ++i) if (rR[k][i].on) {
        double s = rR[k][i].s + alpha * rR[k][i].ds; bar(rR[k][i], s);
        t.theta += std::fabs((Ut[k][i] - Ut[k - 1][i]) - s - elastic(rR[k][i]));
      }
      if (k >= 1 && obs_node[k]) for (int j = 0; j < nobs; ++j) if (rO[k][j].on) {
        double s = rO[k][j].s + alpha * rO[k][j].ds; bar(rO[k][j], s);
        t.theta += std::fabs(rowval(k, j, Xt[k]) - s - elastic(rO[k][j]));
      }
    }
    t.phi = phi;
    if (!std::isfinite(t.theta) || !std::isfinite(t.phi)) t.ok = false;
    return t;
  }

  bool filter_ok(double th, double ph) const {
    for (auto& e : filter) if (th >= e.first && ph >= e.second) return false;
    return true;
  }

  void step_lengths(double tau_, double& a_pr, double& a_du) {
    a_pr = 1.0; a_du = 1.0;
    for (int k = 0; k <= N; ++k) each_item(k, [&](Ineq& it) {
      if (it.hasL) {
        if (it.ds < 0) a_pr = std::min(a_pr, -tau_ * (it.s - it.L) / it.ds);
        if (it.dvL < 0) a_du = std::min(a_du, -tau_ * it.vL / it.dvL);
      }
      if (it.hasU) {
        if (it.ds > 0) a_pr = std::min(a_pr, tau_ * (it.U - it.s) / it.ds);
        if (it.dvU < 0) a_du = std::min(a_du, -tau_ * it.vU / it.dvU);
      }
      if (it.el) {
        if (it.dp < 0) a_pr = std::min(a_pr, -tau_ * it.p / it.dp);
        if (it.dn < 0) a_pr = std::min(a_pr, -tau_ * it.n / it.dn);
        if (it.dvp < 0) a_du = std::min(a_du, -tau_ * it.vp / it.dvp);
        if (it.dvn < 0) a_du = std::min(a_du, -tau_ * it.vn / it.dvn);
      }
    });
  }

  double dir_deriv(double mu_) {   // directional derivative of the barrier function along (dX, dU, ds)
    double dphi = 0;
    for (int k = 0; k <= N; ++k) {
      if (k >= 1) for (int i = 0; i < nx; ++i) dphi += osc * 2 * Qc[k][i] * (X[k][i] - Xr[k][i]) * dX[k][i];
      if (k < N) for (int i = 0; i < NU; ++i) {
        dphi += osc * 2 * Rc[k][i] * (U[k][i] - Ur[k][i]) * dU[k][i];
        if (du_cost(k)) {
          double d = U[k][i] - (k ? U[k - 1][i] : c.u_last[i]);
          dphi += osc * 2 * DRc[i] * d * (dU[k][i] - (k ? dU[k - 1][i] : 0.0));
        }
      }
      each_item(k, [&](Ineq& it) {
        if (it.hasL) dphi -= mu_ * it.ds / (it.s - it.L);
        if (it.hasU) dphi += mu_ * it.ds / (it.U - it.s);
        if (it.el) dphi += (rho - mu_ / it.p) * it.dp + (rho - mu_ / it.n) * it.dn;
      });
    }
    return dphi;
  }

  // residuals of the constraints at the trial point  w + alpha d  (d = the direction arrays as they stand): shooting defects and
  // c(w) - s - (p - n) of the general rows; boxes have none.  Used by the second-order correction only.
  void trial_residuals(double alpha, double dfc_t[NODES][NXM], double rR_t[NODES][NU], double rO_t[NODES][NOBM]) {
    static thread_local double Xt[NODES][NXM], Ut[NODES][NU];
    for (int k = 0; k <= N; ++k) {
      for (int i = 0; i < nx; ++i) Xt[k][i] = X[k][i] + alpha * dX[k][i];
      for (int i = 0; i < NU; ++i) Ut[k][i] = (k < N) ? U[k][i] + alpha * dU[k][i] : 0.0;
    }
    for (int k = 0; k < N; ++k) {
      double F[NXM]; step_any(c, Tk[k], Xt[k], Ut[k], F);
      for (int i = 0; i < nx; ++i) dfc_t[k][i] = F[i] - Xt[k + 1][i];
    }
    for (int k = 0; k <= N; ++k) {
      auto el = [&](const Ineq& it) { return it.el ? (it.p + alpha * it.dp) - (it.n + alpha * it.dn) : 0.0; };
      for (int i = 0; i < NU; ++i) rR_t[k][i] = (k >= 1 && k < N && rR[k][i].on) ? (Ut[k][i] - Ut[k - 1][i]) - (rR[k][i].s + alpha * rR[k][i].ds) - el(rR[k][i]) : 0.0;
      for (int j = 0; j < nobs; ++j) rO_t[k][j] = (k >= 1 && obs_node[k] && rO[k][j].on) ? rowval(k, j, Xt[k]) - (rO[k][j].s + alpha * rO[k][j].ds) - el(rO[k][j]) : 0.0;
    }
  }

  // acceptance of a trial point against the filter and the current iterate (Waechter-Biegler A-5.3 / A-5.4); `alpha` is the step
  // size that enters the switching condition and the Armijo test
  bool trial_acceptable(const Trial& t, double alpha, double dphi, double th0, double phi0, bool& armijo_type) {
    armijo_type = false;
    if (!(t.ok && t.theta <= theta_max && filter_ok(t.theta, t.phi))) return false;
    bool sw = dphi < 0 && alpha * std::pow(-dphi, o.s_phi) > o.delta * std::pow(th0, o.s_theta);
    if (th0 <= theta_min && sw) {
      // IPOPT's Compare_le(lhs, rhs, base): lhs - rhs <= 10 eps |base| (IpUtils.cpp), used by ArmijoHolds and
      // IsAcceptableToCurrentIterate of IpFilterLSAcceptor.cpp: round-off slack on both acceptance tests
      if ((t.phi - phi0) - o.eta_phi * alpha * dphi <= 10 * 2.220446049250313e-16 * std::fabs(phi0)) { armijo_type = true; return true; }
      return false;
    }
    return t.theta - (1 - o.gamma_theta) * th0 <= 10 * 2.220446049250313e-16 * std::fabs(th0) ||
           (t.phi - phi0) + o.gamma_phi * th0 <= 10 * 2.220446049250313e-16 * std::fabs(phi0);
  }

  // Second-order correction (Waechter-Biegler A-5.5 .. A-5.9; IPOPT IpFilterLSAcceptor::TrySecondOrderCorrection): when the FIRST trial
  // point of an iteration is rejected and does not reduce the violation, the step is corrected with the constraint values at the trial
  // point,  c_soc = alpha c_soc + c(trial)  in place of c(w_k), same matrix; up to max_soc = 4 corrections while theta shrinks by
  // kappa_soc = 0.99.  NOT part of the shipped algorithm: the HIP kernels have no such step; the oracle
  // carries it behind MPCO_SOC=1 to measure what it would change.  Returns the accepted step size (direction arrays then hold the
  // corrected step, a_du its dual step size), or 0 with the original direction restored.
  int n_soc_try = 0, n_soc_ok = 0;
  double try_soc(const Trial& first, double a_max, double mu_, double dphi, double th0, double phi0, bool& armijo_type, double& a_du) {
    ++n_soc_try;
    static thread_local double dX0[NODES][NXM], dU0[NODES][NU], lamF0[NODES][NXM], dfc0[NODES][NXM], dfc_s[NODES][NXM], dfc_t[NODES][NXM];
    static thread_local double rR_s[NODES][NU], rO_s[NODES][NOBM], rR_t[NODES][NU], rO_t[NODES][NOBM];
    static thread_local Ineq bU0[NODES][NU], bX0[NODES][NXM], rR0[NODES][NU], rO0[NODES][NOBM];
    std::memcpy(dX0, dX, sizeof(dX0)); std::memcpy(dU0, dU, sizeof(dU0)); std::memcpy(lamF0, lamF, sizeof(lamF0)); std::memcpy(dfc0, dfc, sizeof(dfc0));
    for (int k = 0; k <= N; ++k) {
      for (int i = 0; i < NU; ++i) { bU0[k][i] = bU[k][i]; rR0[k][i] = rR[k][i]; rR_s[k][i] = rR[k][i].r; }
      for (int i = 0; i < nx; ++i) bX0[k][i] = bX[k][i];
      for (int j = 0; j < nobs; ++j) { rO0[k][j] = rO[k][j]; rO_s[k][j] = rO[k][j].r; }
    }
    std::memcpy(dfc_s, dfc, sizeof(dfc_s));
    double alpha_soc = a_max, theta_trial = first.theta, theta_old = 0;
    for (int count = 0; count < 4 && (count == 0 || theta_trial <= 0.99 * theta_old); ++count) {
      theta_old = theta_trial;
      trial_residuals(alpha_soc, dfc_t, rR_t, rO_t);            // at w + alpha_soc * (direction as it stands)
      for (int k = 0; k <= N; ++k) {
        for (int i = 0; i < nx; ++i) dfc_s[k][i] = alpha_soc * dfc_s[k][i] + dfc_t[k][i];
        for (int i = 0; i < NU; ++i) rR_s[k][i] = alpha_soc * rR_s[k][i] + rR_t[k][i];
        for (int j = 0; j < nobs; ++j) rO_s[k][j] = alpha_soc * rO_s[k][j] + rO_t[k][j];
      }
      // the corrected step: same matrix, constraint values replaced
      for (int k = 0; k <= N; ++k) {
        for (int i = 0; i < nx; ++i) dfc[k][i] = dfc_s[k][i];
        for (int i = 0; i < NU; ++i) if (rR[k][i].on) rR[k][i].r = rR_s[k][i];
        for (int j = 0; j < nobs; ++j) if (rO[k][j].on) rO[k][j].r = rO_s[k][j];
      }
      solve_direction();
      for (int k = 0; k <= N; ++k) {                             // the iterate's own residuals back (kkt_error, the next trial_residuals)
        for (int i = 0; i < nx; ++i) dfc[k][i] = dfc0[k][i];
        for (int i = 0; i < NU; ++i) if (rR[k][i].on) rR[k][i].r = rR0[k][i].r;
        for (int j = 0; j < nobs; ++j) if (rO[k][j].on) rO[k][j].r = rO0[k][j].r;
      }
      double a_soc, a_du_soc; step_lengths(tau, a_soc, a_du_soc);
      Trial t = eval_trial(a_soc, mu_); ++n_trial;
      if (trial_acceptable(t, a_max, dphi, th0, phi0, armijo_type)) { ++n_soc_ok; a_du = a_du_soc; return a_soc; }
      alpha_soc = a_soc; theta_trial = t.theta;
    }
    std::memcpy(dX, dX0, sizeof(dX0)); std::memcpy(dU, dU0, sizeof(dU0)); std::memcpy(lamF, lamF0, sizeof(lamF0));
    for (int k = 0; k <= N; ++k) {
      for (int i = 0; i < NU; ++i) { bU[k][i] = bU0[k][i]; rR[k][i] = rR0[k][i]; }
      for (int i = 0; i < nx; ++i) bX[k][i] = bX0[k][i];
      for (int j = 0; j < nobs; ++j) rO[k][j] = rO0[k][j];
    }
    armijo_type = false;
    return 0.0;
  }
  static bool soc_enabled() { static const bool on = [] { const char* e = std::getenv("MPCO_SOC"); return e && e[0] == '1'; }(); return on; }

  // IPOPT's filter line search (Waechter-Biegler Alg. A, steps A-5); the second-order correction only with MPCO_SOC=1 (see try_soc).
  // Returns the accepted alpha, or 0 when alpha fell under alpha_min.  a_du: replaced when a corrected step is accepted.
  double line_search(double a_max, double mu_, double dphi, bool& armijo_type, double& a_du) {
    const double phi0 = barrier_phi(fval, mu_), th0 = theta;
    double a_min;
    if (dphi < 0) {
      a_min = std::min(o.gamma_theta, o.gamma_phi * th0 / (-dphi));
      if (th0 <= theta_min) a_min = std::min(a_min, o.delta * std::pow(th0, o.s_theta) / std::pow(-dphi, o.s_phi));
    } else a_min = o.gamma_theta;
    a_min *= o.gamma_alpha;
    double alpha = a_max; armijo_type = false;
    while (true) {
      Trial t = eval_trial(alpha, mu_);
      ++n_trial;
      if (trial_acceptable(t, alpha, dphi, th0, phi0, armijo_type)) return alpha;
      if (alpha == a_max) {
        ++n_first_rejected; if (t.theta >= th0) ++n_first_rejected_theta_up;
        if (soc_enabled() && t.ok && t.theta >= th0) {
          const double a = try_soc(t, a_max, mu_, dphi, th0, phi0, armijo_type, a_du);
          if (a > 0) return a;
        }
      }
      alpha *= 0.5;
      if (alpha < a_min || alpha < 1e-16) return 0.0;
    }
  }
  int n_first_rejected = 0, n_first_rejected_theta_up = 0;

  void apply_step(double alpha, double alpha_du, double mu_) {
    for (int k = 0; k <= N; ++k) {
      if (k >= 1) for (int i = 0; i < nx; ++i) { X[k][i] += alpha * dX[k][i]; lam[k][i] += alpha * (lamF[k][i] - lam[k][i]); }
      if (k < N) for (int i = 0; i < NU; ++i) U[k][i] += alpha * dU[k][i];
      each_item(k, [&](Ineq& it) {
        it.s += alpha * it.ds;
        if (it.hasL) it.vL += alpha_du * it.dvL;
        if (it.hasU) it.vU += alpha_du * it.dvU;
        if (it.el) { it.p += alpha * it.dp; it.n += alpha * it.dn; it.vp += alpha_du * it.dvp; it.vn += alpha_du * it.dvn; }
      });
    }
    // boxes: slack is the variable itself
    for (int k = 0; k <= N; ++k) {
      for (int i = 0; i < NU; ++i) if (bU[k][i].on) bU[k][i].s = U[k][i];
      for (int i = 0; i < nx; ++i) if (bX[k][i].on) bX[k][i].s = X[k][i];
      each_item(k, [&](Ineq& it) {   // IPOPT eq. (16): keep Sigma within kappa_Sigma of mu / slack^2
        if (it.hasL) { double d = it.s - it.L; it.vL = std::max(std::min(it.vL, o.kappa_sigma * mu_ / d), mu_ / (o.kappa_sigma * d)); }
        if (it.hasU) { double d = it.U - it.s; it.vU = std::max(std::min(it.vU, o.kappa_sigma * mu_ / d), mu_ / (o.kappa_sigma * d)); }
        if (it.el) {
          it.vp = std::max(std::min(it.vp, o.kappa_sigma * mu_ / it.p), mu_ / (o.kappa_sigma * it.p));
          it.vn = std::max(std::min(it.vn, o.kappa_sigma * mu_ / it.n), mu_ / (o.kappa_sigma * it.n));
        }
      });
    }
  }

  // ----- one interior-point iteration of the current phase (main or restoration) --------------------------------
  // false: no acceptable step (status_fail says why)
  bool ip_iteration(double mu_floor, int& status_fail) {
    double a_pr, a_du, alpha = 0, dphi = 0; bool armijo_type = false;
    // barrier parameter update (monotone, Fiacco-McCormick)
    const double mu_before = mu;
    for (;;) {
      Err em = kkt_error(mu);
      if (Emu(em) <= o.kappa_eps * mu && mu > mu_floor) {
        mu = std::max(mu_floor, std::min(o.kappa_mu * mu, std::pow(mu, o.theta_mu)));
        tau = std::max(o.tau_min, 1.0 - mu);
        filter.clear();
      } else break;
    }
    // the proximity weight of the restoration problem follows mu: zeta = sqrt(mu)
    if (resto && mu != mu_before) { set_resto_cost(std::sqrt(mu)); fval = objective(X, U) + elastic_cost(); }
    if (!factorize()) { status_fail = MPCB_ST_NUMERIC; return false; }
    set_targets(mu);
    solve_direction();
    step_lengths(tau, a_pr, a_du);
    dphi = dir_deriv(mu);
    alpha = line_search(a_pr, mu, dphi, armijo_type, a_du);
    if (debug) std::fprintf(stderr, "%s %3d mu %.2e E0 %.3e th %.3e f %.8e a_pr %.3e a %.3e a_du %.3e dw %.1e dphi %.2e |F|=%zu\n", resto ? "R " : "it",
                            iters, mu, err0, theta, fval, a_pr, alpha, a_du, dw_used, dphi, filter.size());
    if (debug) {   // the item that limits the primal step
      double best = 2; const Ineq* bi = nullptr; int bk = -1; const char* what = "";
      for (int k = 0; k <= N; ++k) each_item(k, [&](Ineq& it) {
        auto upd = [&](double a, const char* w) { if (a < best) { best = a; bi = &it; bk = k; what = w; } };
        if (it.hasL && it.ds < 0) upd(-(it.s - it.L) / it.ds, "sL");
        if (it.hasU && it.ds > 0) upd((it.U - it.s) / it.ds, "sU");
        if (it.el && it.dp < 0) upd(-it.p / it.dp, "p");
        if (it.el && it.dn < 0) upd(-it.n / it.dn, "n");
      });
      if (bi) std::fprintf(stderr, "      block %s: node %d idx(%d,%d) s-L %.3e U-s %.3e ds %.3e vL %.3e vU %.3e r %.3e p %.2e n %.2e dp %.2e dn %.2e\n", what, bk, bi->i0, bi->i1,
                           bi->s - bi->L, bi->U - bi->s, bi->ds, bi->vL, bi->vU, bi->r, bi->p, bi->n, bi->dp, bi->dn);
    }
    last_alpha = alpha; last_apr = a_pr;
    if (!(alpha > 0)) { status_fail = MPCB_ST_LINESEARCH; return false; }
    if (!armijo_type) filter.emplace_back((1 - o.gamma_theta) * theta, barrier_phi(fval, mu) - o.gamma_phi * theta);
    apply_step(alpha, a_du, mu);
    eval_point();
    if (!std::isfinite(theta) || !std::isfinite(fval)) { status_fail = MPCB_ST_NUMERIC; return false; }
    return true;
  }
  double last_alpha = 0, last_apr = 0, slow_theta0 = 0;
  int slow_run = 0, trips = 0;
  int acc_cnt = 0; double f_last = 1e300;       // acceptable-point counter and the objective of the previous convergence check
  int iters_prev = 0;                           // iterations of the failed first attempt (reported in the sum)
  static constexpr int ST_HANDED_OVER = 7;      // internal (= the kernels' MPCB_ST_NEEDS_RESTO): the attempt stopped where restoration would begin
  bool defer_restoration = false;               // first attempt under cfg.second_start = 1

  // violation of the ORIGINAL constraints at the iterate: shooting defects and  c(w) - s  of the general rows (l1 and max norm)
  void original_violation(double& th1, double& thinf) {
    th1 = 0; thinf = 0;
    for (int k = 0; k < N; ++k) for (int i = 0; i < nx; ++i) { th1 += std::fabs(dfc[k][i]); thinf = std::max(thinf, std::fabs(dfc[k][i])); }
    for (int k = 0; k <= N; ++k) each_item(k, [&](Ineq& it) {
      const double r0 = it.r + (it.el ? it.p - it.n : 0.0);
      th1 += std::fabs(r0); thinf = std::max(thinf, std::fabs(r0));
    });
  }

  // ----- restoration phase (Waechter & Biegler 2006, section 3.3, on the stage-structured problem) ----------------------
  //   min  rho sum_i (p_i + n_i) + zeta/2 ||D_R (w - w_R)||^2      zeta = sqrt(mu), D_R = diag(1 / max(1, |w_R|)), rho = 1000
  //   s.t. X_{k+1} = F(X_k, U_k)                                    shooting rows stay hard
  //        c_i(w) - s_i - p_i + n_i = 0,  L <= s <= U,  p, n >= 0   general rows (rate, obstacle) elastic
  //        boxes on U and X as they are
  // solved by the same interior-point iteration (own filter, own mu, starting at max(mu, ||violation||_inf); p, n initialised
  // from the residuals as in IPOPT eq. (33); bound multipliers capped at rho; equality multipliers 0).  Leaves
  //   0  with an iterate acceptable to the main filter (augmented with the entry point) whose original violation is <= 0.9 of the
  //      violation at entry: the main phase continues from it (mu of the main phase, lam = 0, bound duals kept if <= 1000 else 1);
  //   MPCB_ST_INFEASIBLE   the restoration problem itself is solved to tol and the original violation is still > tol: a
  //      stationary point of the l1 violation = LOCAL infeasibility;
  //   MPCB_ST_RESTO_FAILED / MPCB_ST_MAXITER / MPCB_ST_NUMERIC otherwise.
  int restoration() {
    if (n_resto_calls >= o.resto_max_calls) return MPCB_ST_RESTO_FAILED;     // the phase has been tried often enough on this instance
    ++n_resto_calls;
    acc_cnt = 0; f_last = 1e300;           // the acceptable-point counter starts afresh after a restoration (the HIP kernel's second pass knows nothing of the first's)
    const double mu_main = mu;
    // Entry.  The slacks of the general rows (rate, obstacle) are auxiliary variables; after a stalled main phase they lag behind
    // the row values or sit pinned at a bound.  They are re-initialised from w exactly as at a fresh start (row value pushed
    // inside its bounds), and the entry pair (theta, phi) of the main filter is evaluated there.  The main filter restarts from
    // that pair alone.  (So the whole state the phase starts from is w, mu and the filter bounds: that is what the HIP kernel
    // hands from its first pass to its restoration pass.)
    for (int k = 0; k <= N; ++k) {
      auto fresh = [&](Ineq& it) { if (it.on) { const double cval = it.r + it.s; it.s = push(it, cval); } };
      if (k >= 1 && k < N) for (int i = 0; i < NU; ++i) fresh(rR[k][i]);
      if (k >= 1 && obs_node[k]) for (int j = 0; j < nobs; ++j) fresh(rO[k][j]);
    }
    eval_point();
    double th_entry, thinf_entry; original_violation(th_entry, thinf_entry);
    const double phi_entry = barrier_phi(fval, mu_main);
    std::vector<std::pair<double, double>> filter_main;
    filter_main.emplace_back((1 - o.gamma_theta) * th_entry, phi_entry - o.gamma_phi * th_entry);
    const double theta_max_main = theta_max, theta_min_main = theta_min;
    // the restoration problem: obstacle rows elastic (the reverse-convex rows are what can make an instance infeasible; rate rows
    // and boxes are linear and jointly satisfiable, they stay as they are), p, n from the residuals (IPOPT eq. (33)), every
    // bound dual centred at mu / distance and capped at rho
    resto = true;
    std::memcpy(XR, X, sizeof XR); std::memcpy(UR, U, sizeof UR);
    mu = std::max(mu_main, thinf_entry);
    tau = std::max(o.tau_min, 1.0 - mu);
    set_resto_cost(std::sqrt(mu));
    for (int k = 0; k <= N; ++k) {
      if (k >= 1 && obs_node[k]) for (int j = 0; j < nobs; ++j) {
        Ineq& it = rO[k][j];
        if (!it.on) continue;
        const double r0 = it.r, a = (mu - rho * r0) / (2 * rho);
        it.el = true;
        it.n = a + std::sqrt(a * a + mu * r0 / (2 * rho)); it.p = r0 + it.n;
        it.vp = mu / it.p; it.vn = mu / it.n;
      }
      each_item(k, [&](Ineq& it) {
        if (it.hasL) it.vL = std::min(rho, mu / (it.s - it.L));
        if (it.hasU) it.vU = std::min(rho, mu / (it.U - it.s));
      });
    }
    std::memset(lam, 0, sizeof lam);
    filter.clear();
    eval_point();
    theta_max = 1e4 * std::max(1.0, theta); theta_min = 1e-4 * std::max(1.0, theta);
    const double mu_floor = c.tol / (o.kappa_eps + 1.0);
    int rc = -1;
    for (int rit = 0;; ++rit, ++iters, ++n_resto_iters) {
      if (++trips > 3 * c.max_iter + 50) { rc = MPCB_ST_RESTO_FAILED; break; }
      // did the restoration do its job?  (original violation and original barrier function at this iterate)
      double th1, thinf; original_violation(th1, thinf);
      if (rit >= 1 && th1 <= o.resto_kappa * th_entry && th1 <= theta_max_main) {
        const bool was = resto; resto = false; set_main_cost();
        double phi_o = os * objective(X, U);
        for (int k = 0; k <= N; ++k) each_item(k, [&](Ineq& it) {
          if (it.hasL) phi_o -= mu_main * std::log(it.s - it.L);
          if (it.hasU) phi_o -= mu_main * std::log(it.U - it.s);
        });
        resto = was; set_resto_cost(std::sqrt(mu));
        bool okf = true;
        for (auto& e : filter_main) if (th1 >= e.first && phi_o >= e.second) okf = false;
        if (okf) { rc = 0; break; }
      }
      Err e0 = kkt_error(0.0);
      err0 = Emu(e0);
      // the restoration problem itself is solved: violated rows -> locally infeasible; a feasible point that the main phase could
      // not leave (no restoration step was taken) -> the restoration has nothing to offer (IPOPT: Restoration_Failed)
      if (err0 <= c.tol) { rc = (thinf > c.tol) ? MPCB_ST_INFEASIBLE : (rit == 0 ? MPCB_ST_RESTO_FAILED : 0); break; }
      // Local-infeasibility certificate without driving mu to 1e-9.  Once the barrier subproblem of the current mu is solved
      // (the test that lets mu decrease) and the hard rows (shooting, rate) are satisfied to a small fraction of it, the elastic
      // violation V = sum (p + n) at this iterate exceeds the locally minimal one by at most the barrier duality gap n_v mu / rho
      // plus what the proximity term can buy, (n_w / 2) sqrt(mu) / rho.  If V is larger than that (with a safety factor), no
      // nearby point satisfies the obstacle rows.
      {
        Err em = kkt_error(mu);
        int n_v = 0, n_w = nx * N + NU * N;
        double V = 0;
        for (int k = 0; k <= N; ++k) each_item(k, [&](Ineq& it) { n_v += (it.hasL ? 1 : 0) + (it.hasU ? 1 : 0) + (it.el ? 2 : 0); if (it.el) V += it.p + it.n; });
        const double gap = (o.gap_safety * n_v * mu + 0.5 * n_w * std::sqrt(mu)) / rho;
        if (Emu(em) <= o.kappa_eps * mu && V > gap + 1e-6 && theta <= 0.01 * V) { rc = MPCB_ST_INFEASIBLE; break; }
      }
      if (iters >= c.max_iter) { rc = MPCB_ST_MAXITER; break; }
      if (n_resto_iters >= o.resto_max_iters) { rc = MPCB_ST_RESTO_FAILED; break; }   // effort bound of the phase (all its entries together)
      int why = MPCB_ST_RESTO_FAILED;
      if (!ip_iteration(mu_floor, why)) { rc = (why == MPCB_ST_LINESEARCH) ? MPCB_ST_RESTO_FAILED : why; break; }
    }
    // back to the main problem (also on failure: the outputs are those of the original NLP)
    resto = false; set_main_cost();
    double vmax = 0;
    for (int k = 0; k <= N; ++k) each_item(k, [&](Ineq& it) {
      it.el = false; it.p = it.n = it.vp = it.vn = it.dp = it.dn = 0;
      if (it.hasL) vmax = std::max(vmax, it.vL);
      if (it.hasU) vmax = std::max(vmax, it.vU);
    });
    if (vmax > 1000.0) for (int k = 0; k <= N; ++k) each_item(k, [&](Ineq& it) { if (it.hasL) it.vL = 1.0; if (it.hasU) it.vU = 1.0; });
    std::memset(lam, 0, sizeof lam);
    mu = mu_main; tau = std::max(o.tau_min, 1.0 - mu);
    filter = filter_main; theta_max = theta_max_main; theta_min = theta_min_main;
    eval_point();
    return rc;
  }

  // ----- main loop ------------------------------------------------------------------------------------------
  void solve() {
    eval_point();
    theta_max = 1e4 * std::max(1.0, theta); theta_min = 1e-4 * std::max(1.0, theta);
    const double mu_floor = c.tol / (o.kappa_eps + 1.0);
    status = MPCB_ST_MAXITER;
    for (iters = 0;; ++iters) {
      if (++trips > 3 * c.max_iter + 50) { status = MPCB_ST_RESTO_FAILED; break; }   // phase changes are not iterations: bound them too
      Err e0 = kkt_error(0.0);
      err0 = Emu(e0);
      // IPOPT's OptimalityErrorConvergenceCheck (IpOptErrorConvCheck.cpp): "optimal" = scaled error <= tol AND the unscaled gates
      // dual_inf_tol / constr_viol_tol / compl_inf_tol (dual infeasibility and complementarity of the scaled problem divided by the
      // objective scaling; no constraint scaling here); then the acceptable-point counter: acceptable_iter iterations in a row
      // within the acceptable_* tolerances whose (scaled) objective changed by <= acceptable_obj_change_tol relative
      if (err0 <= c.tol && e0.dual <= c.dual_inf_tol * os && e0.prim <= c.constr_viol_tol && e0.comp <= c.compl_inf_tol * os) { status = MPCB_ST_SOLVED; break; }
      {
        const double fcur = os * fval;
        const bool acc = c.acceptable_iter > 0 && err0 <= c.acceptable_tol && e0.dual <= c.acceptable_dual_inf_tol * os &&
                         e0.prim <= c.acceptable_constr_viol_tol && e0.comp <= c.acceptable_compl_inf_tol * os &&
                         std::fabs(fcur - f_last) <= c.acceptable_obj_change_tol * std::max(1.0, std::fabs(fcur));
        f_last = fcur; acc_cnt = acc ? acc_cnt + 1 : 0;
        if (acc && acc_cnt >= c.acceptable_iter) { status = MPCB_ST_ACCEPTABLE; break; }
      }
      if (iters >= c.max_iter) { status = MPCB_ST_MAXITER; break; }
      int why = MPCB_ST_LINESEARCH;
      const double th_before = theta;
      if (ip_iteration(mu_floor, why)) {
        // early entry into restoration: trig_k accepted steps in a row shorter than trig_alpha that together reduced the violation
        // by less than the factor trig_theta (a slack pinned at its bound with the row still violated: the pattern of an
        // infeasible instance; IPOPT itself waits for the line search to fail, dozens of such steps later)
        if (last_alpha < o.trig_alpha && theta > 1e-6) { if (slow_run == 0) slow_theta0 = th_before; ++slow_run; } else slow_run = 0;
        if (!(c.restoration && o.trig_k > 0 && slow_run >= o.trig_k && theta > o.trig_theta * slow_theta0)) continue;
        ++iters;                         // this iteration was completed
      }
      slow_run = 0;
      if (why != MPCB_ST_LINESEARCH || !c.restoration) { status = why; break; }
      // a line search that fails at an (almost) feasible point is round-off in the end game, not infeasibility: restoration
      // has nothing to restore there and would throw the nearly converged multipliers away (IPOPT: "Restoration phase is
      // called at point that is almost feasible" -> Restoration_Failed).  The iterate is returned as it is.
      if (why == MPCB_ST_LINESEARCH && e0.prim <= c.tol) { status = MPCB_ST_RESTO_FAILED; break; }
      if (defer_restoration) { status = ST_HANDED_OVER; break; }   // cfg.second_start = 1: the second start takes over instead
      const int rc = restoration();      // counts its iterations in `iters`
      if (rc != 0) { status = rc; break; }
      --iters;                           // the for-increment belongs to an iteration; the hand-over itself is none
    }
  }

  // ----- outputs in the reference's ordering ----------------------------------------------------------------
  void write(double* z, double* obj, int32_t* st, int32_t* it, double* kkt, double* lam_g, double* lam_x) {
    const int nz = NU * N + nx * (N + 1);
    if (z) {
      for (int k = 0; k < N; ++k) for (int i = 0; i < NU; ++i) z[NU * k + i] = U[k][i];
      for (int k = 0; k <= N; ++k) for (int i = 0; i < nx; ++i) z[NU * N + nx * k + i] = X[k][i];
    }
    if (obj) { set_main_cost(); *obj = objective(X, U); }
    if (st) *st = status;
    if (std::getenv("MPCO_STATS")) std::fprintf(stderr, "STATS %d %d %d %d %d %d %d %d %d %d\n", status, iters, n_factor, n_trial, n_resto_calls, n_resto_iters,
                                                 n_first_rejected, n_first_rejected_theta_up, n_soc_try, n_soc_ok);
    if (it) *it = iters + iters_prev;
    if (kkt) {
      double du = 0; Err e = kkt_error(0.0, &du);
      kkt[0] = Emu(e); kkt[1] = e.prim; kkt[2] = du; kkt[3] = mu;
    }
    if (lam_x) {   // IPOPT: lam_x = z_U - z_L, unscaled
      for (int i = 0; i < nz; ++i) lam_x[i] = 0;
      for (int k = 0; k < N; ++k) for (int i = 0; i < NU; ++i) if (bU[k][i].on) lam_x[NU * k + i] = -bU[k][i].y() / os;
      for (int k = 1; k <= N; ++k) for (int i = 0; i < nx; ++i) if (bX[k][i].on) lam_x[NU * N + nx * k + i] = -bX[k][i].y() / os;
    }
    if (lam_g) {
      int ng = 0; mpco_dims(&c, nullptr, nullptr, &ng);
      for (int i = 0; i < ng; ++i) lam_g[i] = 0;
      // multiplier of row (X_k - F_{k-1}) is -lam_k (Riccati sign) / os
      int nrate = 0; for (int i = 0; i < NU; ++i) if (std::isfinite(c.du_lo[i]) || std::isfinite(c.du_hi[i])) ++nrate;
      std::vector<int> dyn_row(N + 1), rate_row(N + 1, -1);
      int r = nx;
      if (!c.rate_interleaved) {
        for (int k = 1; k <= N; ++k) { dyn_row[k] = r; r += nx; }
        for (int k = 1; k < N; ++k) { rate_row[k] = r; r += nrate; }
      } else {
        for (int i = 0; i < N; ++i) { dyn_row[i + 1] = r; r += nx; if (i > 0) { rate_row[i] = r; r += nrate; } }
      }
      for (int k = 1; k <= N; ++k) for (int i = 0; i < nx; ++i) lam_g[dyn_row[k] + i] = -lam[k][i] / os;
      // initial-condition rows: stationarity wrt X_0
      for (int i = 0; i < nx; ++i) {
        double s = -2 * c.Q[i] * (X[0][i] - xs[i]);
        for (int a = 0; a < nx; ++a) s -= me[0].A[a][i] * lam[1][a] / os;
        lam_g[i] = s;
      }
      for (int k = 1; k < N; ++k) { int q = 0; for (int i = 0; i < NU; ++i) if (rR[k][i].on) lam_g[rate_row[k] + q++] = -rR[k][i].y() / os; }
      const int last_row = c.obs_terminal ? N : N - 1;
      for (int i = 0; i <= last_row; ++i) {
        int k = (c.obs_mode == MPCB_OBS_KEEPOUT || gen()) ? i : i + 1;
        for (int j = 0; j < nobs; ++j) if (k >= 1 && k <= N && rO[k][j].on) lam_g[r + i * nobs + j] = -rO[k][j].y() / os;
      }
      // general-gamma rows were solved as c_i(X_i) = h(F(X_i)) - (1-gamma) h(X_i); in the reference's form the same row is
      // h(X_{i+1}) - (1-gamma) h(X_i) and its dependence on X_{i+1} moves  lam_c,i * grad h(X_{i+1})  into the multiplier of the
      // dynamics row that defines X_{i+1}
      if (gen()) for (int k = 1; k <= N - 1; ++k) for (int j = 0; j < nobs; ++j) if (rO[k][j].on) {
        const ObsP& q = obs[k][j];
        const double lc = -rO[k][j].y() / os;
        lam_g[dyn_row[k + 1] + 0] -= lc * 2 * (X[k + 1][0] - q.ox) * q.ix2;
        lam_g[dyn_row[k + 1] + 1] -= lc * 2 * (X[k + 1][1] - q.oy) * q.iy2;
      }
    }
  }
};

int check_cfg(const mpcb_config* c) {
  if (!c || c->struct_size != sizeof(mpcb_config)) return MPCB_E_INVALID;
  if (c->model != MPCB_MODEL_KIN && c->model != MPCB_MODEL_DYN) return MPCB_E_INVALID;
  if (c->N < 1 || c->N > MPCB_N_MAX || c->n_obs < 0 || c->n_obs > MPCB_NOBS_MAX) return MPCB_E_INVALID;
  if (!(c->T > 0) || !(c->tol > 0) || c->max_iter < 0) return MPCB_E_INVALID;
  if (c->obs_mode == MPCB_OBS_DCBF && !(c->gamma > 0.0 && c->gamma <= 1.0 + 1e-12)) return MPCB_E_INVALID;
  if (c->obs_mode == MPCB_OBS_DCBF && c->gamma < 1.0 - 1e-12 && (c->model != MPCB_MODEL_KIN || c->obs_terminal)) return MPCB_E_UNSUPPORTED;
  if (c->obs_mode == MPCB_OBS_DCBF && c->obs_terminal) return MPCB_E_UNSUPPORTED;
  if (c->integrator != MPCB_INT_EULER && c->integrator != MPCB_INT_RK4) return MPCB_E_INVALID;
  return MPCB_OK;
}

}  // namespace

// =============================================================================================================
// C entry points (loaded with ctypes by tests / smoke / bench cpu_baseline).  Same argument meaning as
// mpcb_solve in include/mpcbatch.h.
// =============================================================================================================
extern "C" {

int mpco_default_config(mpcb_config* cfg, int32_t model, int32_t N, double T) {
  if (!cfg) return MPCB_E_INVALID;
  std::memset(cfg, 0, sizeof *cfg);
  mpcb_config& c = *cfg;
  c.struct_size = sizeof(mpcb_config); c.model = model; c.N = N; c.T = T;
  c.n_obs = 0; c.obs_mode = MPCB_OBS_KEEPOUT; c.gamma = 1.0; c.max_iter = 100;
  c.mu_strategy = MPCB_MU_MONOTONE; c.init_rollout = 0; c.integrator = MPCB_INT_EULER; c.restoration = 1;
  const double deg = M_PI / 180.0;
  for (int i = 0; i < NXM; ++i) { c.x_lo[i] = -INF; c.x_hi[i] = INF; }
  c.u_lo[0] = -35 * deg; c.u_hi[0] = 35 * deg; c.u_lo[1] = -3.0; c.u_hi[1] = 3.0;   // mpc_parameters.yaml:34-37
  c.x_lo[1] = -1.0; c.x_hi[1] = 5.0; c.x_lo[3] = 0.0; c.x_hi[3] = 40.0;             // mpc_parameters.yaml:32-33,38-39
  c.du_lo[0] = -5 * deg * T; c.du_hi[0] = 5 * deg * T;                               // mpc_parameters.yaml:48-49
  c.du_lo[1] = -INF; c.du_hi[1] = INF;
  c.ego_hl = 2.4; c.ego_hw = 0.9; c.safe_disl = 1.0; c.safe_disw = 0.5; c.veh_l = 2.6;
  c.veh_m = 1575; c.veh_lf = 1.2; c.veh_lr = 1.6; c.veh_Iz = 2875;
  c.aopt_f = 0.3490658503988659; c.aopt_r = 0.19198621771937624;
  c.Fymax_f = -50000 * c.aopt_f / 2; c.Fymax_r = -50000 * c.aopt_r / 2;
  if (model == MPCB_MODEL_KIN) {
    double Q[4] = {1e1, 1e5, 3e5, 1e4}; std::memcpy(c.Q, Q, sizeof Q);               // kin.py:168-172
    c.R[0] = 1e4; c.R[1] = 1e4; c.DR[0] = 1e5; c.DR[1] = 1e2; c.du0_cost = 1;        // kin.py:179-184
    c.obs_terminal = 0; c.obs_hmin = 0.0; c.rate_interleaved = 0;
  } else {
    double Q[6] = {10, 1e5, 1e3, 1e3, 1, 1}; std::memcpy(c.Q, Q, sizeof Q);          // dyn.py:189-195
    c.R[0] = 1e3; c.R[1] = 1e3; c.DR[0] = 5e3; c.DR[1] = 5e2; c.du0_cost = 0;        // dyn.py:204-209
    c.x_lo[4] = -5.0; c.x_hi[4] = 5.0;                                                // mpc_parameters.yaml:44-45
    c.du_lo[1] = -3.0 * T; c.du_hi[1] = 1.5 * T;                                      // mpc_parameters.yaml:46-47
    c.obs_terminal = 1; c.obs_hmin = 1.0; c.obs_sx_fixed = 4.0; c.obs_sy_fixed = 1.0; c.rate_interleaved = 1;
  }
  c.tol = 1e-8; c.mu_init = 0.1; c.bound_push = 0.01; c.bound_frac = 0.01; c.bound_relax = 1e-8; c.max_gradient = 100.0;
  c.dual_inf_tol = 1.0; c.constr_viol_tol = 1e-4; c.compl_inf_tol = 1e-4;                       // IPOPT defaults
  c.acceptable_tol = 1e-8; c.acceptable_obj_change_tol = 1e-6; c.acceptable_iter = 15;         // kin.py:252-253, IPOPT acceptable_iter
  c.acceptable_constr_viol_tol = 1e-2; c.acceptable_dual_inf_tol = 1e10; c.acceptable_compl_inf_tol = 1e-2;
  c.second_start = 0;                          // (the oracle's own defaults are IPOPT's: one attempt from the given start)
  c.start_steer = 0.0;
  return MPCB_OK;
}

int mpco_dims(const mpcb_config* c, int32_t* nx, int32_t* nz, int32_t* ng) {
  if (!c) return MPCB_E_INVALID;
  int n = nx_of(*c), nrate = 0;
  for (int i = 0; i < NU; ++i) if (std::isfinite(c->du_lo[i]) || std::isfinite(c->du_hi[i])) ++nrate;
  if (nx) *nx = n;
  if (nz) *nz = NU * c->N + n * (c->N + 1);
  if (ng) *ng = n * (c->N + 1) + nrate * (c->N - 1) + c->n_obs * (c->obs_terminal ? c->N + 1 : c->N);
  return MPCB_OK;
}

int mpco_model_rhs(const mpcb_config* cfg, const double* x, const double* u, double* xdot) {
  if (!cfg || !x || !u || !xdot) return MPCB_E_INVALID;
  rhs_any(*cfg, x, u, xdot);
  return MPCB_OK;
}

// Oracle batch solve: same arrays as mpcb_solve.  threads <= 0: all OpenMP threads.
int mpco_solve(const mpcb_config* cfg, int32_t B, const double* x0, const double* xs, const double* obs,
               int32_t obs_kind, const double* z0, double* z, double* obj, int32_t* status, int32_t* iters,
               double* kkt, double* lam_g, double* lam_x, int32_t threads, const double* tgrid) {
  int rc = check_cfg(cfg);
  if (rc != MPCB_OK) return rc;
  if (B < 0 || !x0 || !xs || !z || (cfg->n_obs > 0 && !obs)) return MPCB_E_INVALID;
  int nx, nz, ng; mpco_dims(cfg, &nx, &nz, &ng);
  const size_t obs_stride = (size_t)cfg->n_obs * 6 * (obs_kind == MPCB_OBSIN_PREDICTED ? cfg->N + 1 : 1);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : omp_get_max_threads())
#endif
  for (int b = 0; b < B; ++b) {
    // one arena per thread, re-used for every instance (a Solver is ~1 MB: allocating it per instance means an mmap / page-fault
    // storm per solve)
    static thread_local void* arena = nullptr;
    if (!arena) arena = ::operator new(sizeof(Solver));
    Solver* s = new (arena) Solver(*cfg);
    // cfg.second_start = 3: a cold start (z0 = NULL) behaves as 1, a solve with a start vector as 2
    const int ss = cfg->second_start == 3 ? (z0 ? 2 : 1) : cfg->second_start;
    s->defer_restoration = ss == 1 && cfg->init_rollout != 0;
    bool ok = s->init(x0 + (size_t)b * nx, xs + (size_t)b * nx, obs ? obs + b * obs_stride : nullptr, obs_kind,
                      z0 ? z0 + (size_t)b * nz : nullptr, tgrid);
    if (ok) s->solve(); else s->eval_point();
    // cfg.second_start (after a roll-out start): an attempt that did not succeed — restoration phase included — is followed by a
    // second attempt from z = 0 with a fresh solver state, as the HIP library's second-start passes
    bool fin = true;                                  // non-finite inputs: the first attempt's verdict (at iteration 0) stands
    for (int i = 0; i < nx; ++i) fin = fin && std::isfinite(x0[(size_t)b * nx + i]) && std::isfinite(xs[(size_t)b * nx + i]);
    if (ok && fin && cfg->second_start && cfg->init_rollout && s->status != MPCB_ST_SOLVED && s->status != MPCB_ST_ACCEPTABLE) {
      const int it0 = s->iters;
      s->~Solver();
      s = new (arena) Solver(*cfg);
      ok = s->init(x0 + (size_t)b * nx, xs + (size_t)b * nx, obs ? obs + b * obs_stride : nullptr, obs_kind, nullptr, tgrid, true);
      if (ok) s->solve(); else s->eval_point();
      s->iters_prev = it0;
    }
    s->write(z + (size_t)b * nz, obj ? obj + b : nullptr, status ? status + b : nullptr, iters ? iters + b : nullptr,
             kkt ? kkt + (size_t)b * 4 : nullptr, lam_g ? lam_g + (size_t)b * ng : nullptr,
             lam_x ? lam_x + (size_t)b * nz : nullptr);
    s->~Solver();
  }
  return MPCB_OK;
}

// model derivatives for tests: F[nx], A[nx*nx], Bm[nx*2], Hc[(nx+2)^2] (= sum lam_a T d2f_a), ad = 1 -> AD path
int mpco_model_eval(const mpcb_config* cfg, const double* X, const double* U, const double* lam, double* F, double* A,
                    double* Bm, double* Hc, int32_t ad) {
  if (check_cfg(cfg) != MPCB_OK) return MPCB_E_INVALID;
  const int nx = nx_of(*cfg);
  ModelEval me; double H[NVM][NVM];
  model_eval(*cfg, cfg->T, X, U, lam, me, H, ad != 0);
  for (int i = 0; i < nx; ++i) {
    F[i] = me.F[i];
    for (int j = 0; j < nx; ++j) A[i * nx + j] = me.A[i][j];
    for (int j = 0; j < NU; ++j) Bm[i * NU + j] = me.B[i][j];
  }
  for (int i = 0; i < nx + NU; ++i) for (int j = 0; j < nx + NU; ++j) Hc[i * (nx + NU) + j] = H[i][j];
  return MPCB_OK;
}

}  // extern "C"
