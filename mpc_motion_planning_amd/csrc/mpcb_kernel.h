// mpcb_kernel.h — the batched MPC solve: one wavefront (64 lanes) per problem instance.
//
// What it replaces: the IPOPT solve of the NLP that CasaDi_MPC_Optimize_Multishoot/MPC_CBF_optimize_kin.py
// builds (:136-255) and main_cbf_kin_c_sim.py:100 calls.  The NLP (model :153-156, cost :195-205, dynamics rows
// :207-208, steering-rate rows :211-216, obstacle rows :236-247, boxes :84-105) is restated here in closed form
// with hand-written first and second derivatives.
//
// Algorithm (IPOPT's published primal-dual barrier method, monotone mu, filter line search, inertia
// correction) in wave-parallel form:
//   * lane k owns shooting node k (k = 0..N <= 63): X_k, U_k, the costate of the dynamics row that defines X_k,
//     and the slacks/duals of every inequality row attached to node k.  Model evaluation, row residuals,
//     condensing of the inequality rows into the stage Hessian/gradient, step-length rules and all
//     line-search trial evaluations are lane-parallel; scalars are combined with wave reductions.
//   * the KKT system is solved by a Riccati sweep over the stages with the state augmented by the previous
//     control: stage block (nx+2+2)^2 = 8x8 = 64 entries = ONE ENTRY PER LANE for the kinematic model.
//     Stage data are staged in LDS as [entry][node] (node fastest, odd leading dimension) so that the
//     node-lanes write and the entry-lanes read without bank conflicts.
//   * HBM traffic is the compulsory I/O only: one coalesced row of z0 in, one row of z out (+ multipliers).
//
// The source is written against mpcb_wave.h so that tests/emu can step exactly this code on the CPU.
#pragma once

#include "../../include/mpcbatch.h"
#include "mpcb_wave.h"

#ifdef MPCB_WAVE_EMU
#include <cmath>
using std::cos; using std::sin; using std::tan; using std::log; using std::fabs; using std::pow; using std::fmin; using std::fmax;
using std::isfinite;
#endif

struct MpcbKArgs {
  mpcb_config cfg;
  int32_t B, nz, ng, obs_kind, want_mult, trace_instance;
  const double *x0, *xs, *obs, *z0;
  double *z, *obj, *kkt, *lam_g, *lam_x;
  int32_t *status, *iters;
  double* trace;   // optional: [max_iter + 1][8] log of instance trace_instance (debug / parity tests)
};

// ---------------------------------------------------------------------------------------------------------
// LDS layout (doubles).  ld = (N+1)|1: odd leading dimension of the [entry][node] tables.
// ---------------------------------------------------------------------------------------------------------
namespace mpcbk {

constexpr int NU = 2;
constexpr int FILTER_MAX = 64;

// compact stage entries of the kinematic model, variable order of a stage: [x, y, phi, v, dprev, aprev, d, a]
enum KinEnt {
  E_ZERO = 0, E_ONE, E_T,
  E_A02, E_A03, E_A12, E_A13, E_A23, E_B20,           // nontrivial entries of [A B]
  E_D0, E_D1, E_D2, E_D3,                               // dynamics defect F_k - X_{k+1}
  E_G0, E_G1, E_G2, E_G3, E_G4, E_G5, E_G6, E_G7,       // condensed gradient
  E_HXX, E_HXY, E_HYY, E_HPP, E_HPV, E_HVV, E_HVD, E_HDD, E_HAA, E_H44, E_H55, E_H46, E_H57,
  KIN_NENT
};

struct Layout {
  int ld, ent, Pst, pst, Kst, kff, W, q, M, m, dX, dU, filt, zbuf, total;
};
MPCB_HD Layout layout_kin(int N, int nz) {
  Layout L;
  const int N1 = N + 1, NA = 6, NW = 8, NX = 4;
  L.ld = N1 | 1;
  int o = 0;
  L.ent = o; o += KIN_NENT * L.ld;
  L.Pst = o; o += N1 * NA * NA;
  L.pst = o; o += N1 * NA;
  L.Kst = o; o += N1 * 2 * NA;
  L.kff = o; o += N1 * 2;
  L.W = o; o += NW * NA;
  L.q = o; o += NW;
  L.M = o; o += NW * NW;
  L.m = o; o += NW;
  L.dX = o; o += N1 * NX;
  L.dU = o; o += N1 * NU;
  L.filt = o; o += 2 * FILTER_MAX;
  L.zbuf = L.Pst;                 // staging of z rows aliases the Riccati storage (used before / after the loop)
  (void)nz;
  L.total = o;
  return L;
}

struct Bnd { double L, U; bool hasL, hasU, on; };
MPCB_DEV Bnd mk_bnd(double L, double U, double relax) {
  Bnd q;
  q.hasL = L > -1e300; q.hasU = U < 1e300; q.on = q.hasL || q.hasU;
  q.L = q.hasL ? L - relax * fmax(1.0, fabs(L)) : L;
  q.U = q.hasU ? U + relax * fmax(1.0, fabs(U)) : U;
  return q;
}
MPCB_DEV double push_in(const Bnd& q, double v, double k1, double k2) {
  if (q.hasL && q.hasU) {
    double pl = fmin(k1 * fmax(1.0, fabs(q.L)), k2 * (q.U - q.L));
    double pu = fmin(k1 * fmax(1.0, fabs(q.U)), k2 * (q.U - q.L));
    v = fmax(v, q.L + pl); v = fmin(v, q.U - pu);
  } else if (q.hasL) v = fmax(v, q.L + k1 * fmax(1.0, fabs(q.L)));
  else if (q.hasU) v = fmin(v, q.U - k1 * fmax(1.0, fabs(q.U)));
  return v;
}
// Sigma = vL/(s-L) + vU/(U-s) and the barrier gradient coefficient gb = mu/(s-L) - mu/(U-s) - Sigma r
MPCB_DEV void sig_gb(const Bnd& q, double s, double vL, double vU, double r, double mu, double& sig, double& gb) {
  sig = 0; gb = 0;
  if (q.hasL) { double d = s - q.L; sig += vL / d; gb += mu / d; }
  if (q.hasU) { double d = q.U - s; sig += vU / d; gb -= mu / d; }
  gb -= sig * r;
}
MPCB_DEV double dual_y(const Bnd& q, double vL, double vU) { return (q.hasL ? vL : 0.0) - (q.hasU ? vU : 0.0); }

// IPOPT constants (Waechter & Biegler 2006 / IPOPT option defaults)
constexpr double K_EPS = 10.0, K_MU = 0.2, TH_MU = 1.5, TAU_MIN = 0.99;
constexpr double G_THETA = 1e-5, G_PHI = 1e-8, DELTA = 1.0, S_THETA = 1.1, S_PHI = 2.3, ETA_PHI = 1e-8, G_ALPHA = 0.05;
constexpr double K_SIGMA = 1e10, S_MAX = 100.0;
constexpr double DW_FIRST = 1e-4, DW_MIN = 1e-20, DW_MAX = 1e40, KW_MINUS = 1.0 / 3.0, KW_PLUS = 8.0, KW_PLUS_FIRST = 100.0;

}  // namespace mpcbk

// =============================================================================================================
// Kinematic bicycle, NOBS = compile-time capacity of obstacle rows per node (cfg.n_obs <= NOBS at run time).
// =============================================================================================================
template <int NOBS>
MPCB_DEVFN void mpcb_solve_kin(const MpcbKArgs& a, const int b, double* lds) {
  using namespace mpcbk;
  constexpr int NX = 4, NA = 6, NW = 8;
  const mpcb_config& c = a.cfg;
  const int N = c.N, lane = wv::lane(), k = lane;
  const int nz = a.nz, nobs = c.n_obs;
  const Layout L = layout_kin(N, nz);
  const int ld = L.ld;
  double* ent = lds + L.ent;
  const double T = c.T, il = 1.0 / c.veh_l;

  const bool isnode = k <= N, hasu = k < N, xnode = k >= 1 && k <= N, xcost = k >= 1 && k < N;
  const double* gx0 = a.x0 + (size_t)b * NX;
  const double* gxs = a.xs + (size_t)b * NX;
  double xs[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) xs[i] = gxs[i];

  // ----- obstacles of this node -----------------------------------------------------------------------------
  const int last_row = c.obs_terminal ? N : N - 1;
  bool obs_node; int ostep;
  if (c.obs_mode == MPCB_OBS_KEEPOUT) { obs_node = k <= last_row; ostep = k; }
  else { obs_node = k >= 1 && k - 1 <= last_row; ostep = k - 1; }
  obs_node = obs_node && isnode;
  double ox[NOBS > 0 ? NOBS : 1], oy[NOBS > 0 ? NOBS : 1], ix2[NOBS > 0 ? NOBS : 1], iy2[NOBS > 0 ? NOBS : 1];
#pragma unroll
  for (int j = 0; j < NOBS; ++j) {
    ox[j] = 0; oy[j] = 0; ix2[j] = 0; iy2[j] = 0;
    if (j < nobs && obs_node) {
      const double* q = (a.obs_kind == MPCB_OBSIN_PREDICTED)
                            ? a.obs + (((size_t)b * nobs + j) * (N + 1) + ostep) * 6
                            : a.obs + ((size_t)b * nobs + j) * 6;
      double sx = c.obs_sx_fixed > 0 ? c.obs_sx_fixed : c.ego_hl + q[4] / 2 + c.safe_disl;
      double sy = c.obs_sy_fixed > 0 ? c.obs_sy_fixed : c.ego_hw + q[5] / 2 + c.safe_disw;
      ox[j] = q[0]; oy[j] = q[1]; ix2[j] = 1.0 / (sx * sx); iy2[j] = 1.0 / (sy * sy);
    }
  }
  auto hval = [&](int j, double px, double py) {
    double dx = px - ox[j], dy = py - oy[j];
    return dx * dx * ix2[j] + dy * dy * iy2[j] - 1.0;
  };

  // ----- start point: z0 row (coalesced) -> LDS -> node lanes ---------------------------------------------------
  double* zbuf = lds + L.zbuf;
  for (int i = lane; i < nz; i += 64) zbuf[i] = a.z0 ? a.z0[(size_t)b * nz + i] : 0.0;
  wv::sync();
  double X[NX], U[NU], lam[NX];
#pragma unroll
  for (int i = 0; i < NU; ++i) U[i] = hasu ? zbuf[NU * k + i] : 0.0;
#pragma unroll
  for (int i = 0; i < NX; ++i) { X[i] = isnode ? zbuf[NU * N + NX * k + i] : 0.0; lam[i] = 0.0; }
  wv::sync();

  // objective scaling at the user's start (gradient-based, nlp_scaling_max_gradient)
  double os;
  {
    double g = 0;
    double Un[NU], Up[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) { Un[i] = wv::shfl(U[i], k + 1); Up[i] = wv::shfl(U[i], k - 1); }
    if (hasu) {
#pragma unroll
      for (int i = 0; i < NX; ++i) g = fmax(g, fabs(2 * c.Q[i] * (X[i] - xs[i])));
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        double gu = 2 * c.R[i] * U[i];
        double up = (k == 0) ? c.u_last[i] : Up[i];
        if (k > 0 || c.du0_cost) gu += 2 * c.DR[i] * (U[i] - up);
        if (k + 1 < N) gu -= 2 * c.DR[i] * (Un[i] - U[i]);
        g = fmax(g, fabs(gu));
      }
    }
    g = wv::max(g);
    os = (g > c.max_gradient) ? c.max_gradient / g : 1.0;
  }
  // pin node 0
  if (k == 0) {
#pragma unroll
    for (int i = 0; i < NX; ++i) X[i] = gx0[i];
  }

  int status = MPCB_ST_MAXITER, iters = 0;
  // feasibility of the pinned node
  {
    bool bad = false;
    if (k == 0) {
#pragma unroll
      for (int i = 0; i < NX; ++i) bad = bad || (X[i] < c.x_lo[i] - 1e-8) || (X[i] > c.x_hi[i] + 1e-8);
      if (obs_node) {
#pragma unroll
        for (int j = 0; j < NOBS; ++j) if (j < nobs) bad = bad || (hval(j, X[0], X[1]) < c.obs_hmin - 1e-8);
      }
    }
    if (wv::any(bad)) status = MPCB_ST_INFEASIBLE_X0;
  }

  // bounds (uniform)
  const Bnd qU0 = mk_bnd(c.u_lo[0], c.u_hi[0], c.bound_relax), qU1 = mk_bnd(c.u_lo[1], c.u_hi[1], c.bound_relax);
  const Bnd qY = mk_bnd(c.x_lo[1], c.x_hi[1], c.bound_relax), qV = mk_bnd(c.x_lo[3], c.x_hi[3], c.bound_relax);
  const Bnd qR = mk_bnd(c.du_lo[0], c.du_hi[0], c.bound_relax);
  const Bnd qO = mk_bnd(c.obs_hmin, 1e308, c.bound_relax);
  const bool bu0_on = hasu && qU0.on, bu1_on = hasu && qU1.on, by_on = xnode && qY.on, bv_on = xnode && qV.on;
  const bool rr_on = xcost && qR.on;
  const bool ro_node = xnode && obs_node;

  // model evaluation at (X, U): F = X + T f, nontrivial Jacobian entries
  double F[NX], a02, a03, a12, a13, a23, b20, sp, cp, td, sec2;
  auto model = [&]() {
    sp = sin(X[2]); cp = cos(X[2]); td = tan(U[0]); sec2 = 1.0 + td * td;
    const double v = X[3];
    F[0] = X[0] + T * v * cp; F[1] = X[1] + T * v * sp; F[2] = X[2] + T * v * td * il; F[3] = X[3] + T * U[1];
    a02 = -T * v * sp; a03 = T * cp; a12 = T * v * cp; a13 = T * sp; a23 = T * td * il; b20 = T * v * sec2 * il;
  };

  // optional roll-out of X from x0 with the guessed controls
  if (c.init_rollout) {
    U[0] = hasu ? fmin(fmax(U[0], c.u_lo[0]), c.u_hi[0]) : 0.0;
    U[1] = hasu ? fmin(fmax(U[1], c.u_lo[1]), c.u_hi[1]) : 0.0;
    for (int s = 0; s < N; ++s) {
      model();
#pragma unroll
      for (int i = 0; i < NX; ++i) { double nx_ = wv::shfl(F[i], s); if (k == s + 1) X[i] = nx_; }
    }
  }

  // push the start inside the (relaxed) boxes; duals = 1
  double zU0L = 1, zU0U = 1, zU1L = 1, zU1U = 1, zYL = 1, zYU = 1, zVL = 1, zVU = 1;
  if (bu0_on) U[0] = push_in(qU0, U[0], c.bound_push, c.bound_frac);
  if (bu1_on) U[1] = push_in(qU1, U[1], c.bound_push, c.bound_frac);
  if (by_on) X[1] = push_in(qY, X[1], c.bound_push, c.bound_frac);
  if (bv_on) X[3] = push_in(qV, X[3], c.bound_push, c.bound_frac);
  // general rows: slack = row value at the pushed start, pushed inside its own bounds
  double Up0 = wv::shfl(U[0], k - 1), Up1 = wv::shfl(U[1], k - 1);   // U_{k-1}
  double sR = 0, vRL = 1, vRU = 1, rR = 0;
  if (rr_on) sR = push_in(qR, U[0] - Up0, c.bound_push, c.bound_frac);
  double sO[NOBS > 0 ? NOBS : 1], vO[NOBS > 0 ? NOBS : 1], rO[NOBS > 0 ? NOBS : 1], gO0[NOBS > 0 ? NOBS : 1], gO1[NOBS > 0 ? NOBS : 1];
  bool ro_on[NOBS > 0 ? NOBS : 1];
#pragma unroll
  for (int j = 0; j < NOBS; ++j) {
    ro_on[j] = ro_node && j < nobs;
    sO[j] = ro_on[j] ? push_in(qO, hval(j, X[0], X[1]), c.bound_push, c.bound_frac) : 1.0;
    vO[j] = 1.0; rO[j] = 0; gO0[j] = 0; gO1[j] = 0;
  }

  double mu = c.mu_init, tau = fmax(TAU_MIN, 1.0 - mu);
  double dfc[NX] = {0, 0, 0, 0};
  double theta = 0, fval = 0;

  // unscaled objective of a trajectory given in node lanes (kin.py:195-205)
  auto objective = [&](const double* Xa, const double* Ua, double up0, double up1) {
    double f = 0;
    if (hasu) {
#pragma unroll
      for (int i = 0; i < NX; ++i) { double e = Xa[i] - xs[i]; f += c.Q[i] * e * e; }
      f += c.R[0] * Ua[0] * Ua[0] + c.R[1] * Ua[1] * Ua[1];
      if (k > 0 || c.du0_cost) {
        double d0 = Ua[0] - (k ? up0 : c.u_last[0]), d1 = Ua[1] - (k ? up1 : c.u_last[1]);
        f += c.DR[0] * d0 * d0 + c.DR[1] * d1 * d1;
      }
    }
    return wv::sum(f);
  };

  // evaluation at the iterate: model, defects, row residuals, theta, f
  auto eval_point = [&]() {
    model();
    double th = 0;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      double xn = wv::shfl(X[i], k + 1);
      dfc[i] = hasu ? F[i] - xn : 0.0;
      th += fabs(dfc[i]);
    }
    Up0 = wv::shfl(U[0], k - 1); Up1 = wv::shfl(U[1], k - 1);
    if (rr_on) { rR = (U[0] - Up0) - sR; th += fabs(rR); }
#pragma unroll
    for (int j = 0; j < NOBS; ++j) if (ro_on[j]) {
      rO[j] = hval(j, X[0], X[1]) - sO[j]; th += fabs(rO[j]);
      gO0[j] = 2 * (X[0] - ox[j]) * ix2[j]; gO1[j] = 2 * (X[1] - oy[j]) * iy2[j];
    }
    theta = wv::sum(th);
    fval = objective(X, U, Up0, Up1);
  };

  // barrier function of the current point
  auto barrier_phi = [&](double mu_) {
    double s = 0;
    auto bar = [&](const Bnd& q, double v) {
      if (q.hasL) s -= log(v - q.L);
      if (q.hasU) s -= log(q.U - v);
    };
    if (bu0_on) bar(qU0, U[0]);
    if (bu1_on) bar(qU1, U[1]);
    if (by_on) bar(qY, X[1]);
    if (bv_on) bar(qV, X[3]);
    if (rr_on) bar(qR, sR);
#pragma unroll
    for (int j = 0; j < NOBS; ++j) if (ro_on[j]) bar(qO, sO[j]);
    return os * fval + mu_ * wv::sum(s);
  };

  // number of multipliers (constant): equality multipliers NX*N, bound multipliers
  double n_lam, n_v;
  {
    double cnt = 0;
    auto two = [&](const Bnd& q) { return (q.hasL ? 1.0 : 0.0) + (q.hasU ? 1.0 : 0.0); };
    if (bu0_on) cnt += two(qU0);
    if (bu1_on) cnt += two(qU1);
    if (by_on) cnt += two(qY);
    if (bv_on) cnt += two(qV);
    if (rr_on) cnt += two(qR);
#pragma unroll
    for (int j = 0; j < NOBS; ++j) if (ro_on[j]) cnt += 1.0;
    n_v = wv::sum(cnt);
    n_lam = (double)(NX * N);
  }

  // scaled KKT error pieces at barrier parameter mu_
  double e_dual, e_prim, e_comp, e_sd, e_sc;
  auto kkt_error = [&](double mu_) {
    double rX[NX] = {0, 0, 0, 0}, rU[NU] = {0, 0};
    double ln[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) ln[i] = wv::shfl(lam[i], k + 1);          // lam_{k+1}
    const double Un0 = wv::shfl(U[0], k + 1), Un1 = wv::shfl(U[1], k + 1);
    const double yR = rr_on ? dual_y(qR, vRL, vRU) : 0.0;
    const double yRn = wv::shfl(yR, k + 1);
    double sum_lam = 0, sum_v = 0, comp = 0, prim = 0;
    if (xnode) {
      if (k < N) {
#pragma unroll
        for (int i = 0; i < NX; ++i) rX[i] += os * 2 * c.Q[i] * (X[i] - xs[i]);
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) { rX[i] -= lam[i]; sum_lam += fabs(lam[i]); }
      if (k < N) {   // A^T lam_{k+1}
        rX[0] += ln[0]; rX[1] += ln[1];
        rX[2] += a02 * ln[0] + a12 * ln[1] + ln[2];
        rX[3] += a03 * ln[0] + a13 * ln[1] + a23 * ln[2] + ln[3];
      }
    }
    if (hasu) {
      rU[0] += os * 2 * c.R[0] * U[0]; rU[1] += os * 2 * c.R[1] * U[1];
      if (k > 0 || c.du0_cost) {
        rU[0] += os * 2 * c.DR[0] * (U[0] - (k ? Up0 : c.u_last[0]));
        rU[1] += os * 2 * c.DR[1] * (U[1] - (k ? Up1 : c.u_last[1]));
      }
      if (k + 1 < N) { rU[0] -= os * 2 * c.DR[0] * (Un0 - U[0]); rU[1] -= os * 2 * c.DR[1] * (Un1 - U[1]); }
      rU[0] += b20 * ln[2]; rU[1] += T * ln[3];
      if (k + 1 < N) rU[0] += yRn;                                          // d(row k+1)/dU_k = -1
    }
    auto item = [&](const Bnd& q, double s, double vL, double vU) {
      if (q.hasL) { comp = fmax(comp, fabs((s - q.L) * vL - mu_)); sum_v += vL; }
      if (q.hasU) { comp = fmax(comp, fabs((q.U - s) * vU - mu_)); sum_v += vU; }
    };
    if (bu0_on) { rU[0] -= dual_y(qU0, zU0L, zU0U); item(qU0, U[0], zU0L, zU0U); }
    if (bu1_on) { rU[1] -= dual_y(qU1, zU1L, zU1U); item(qU1, U[1], zU1L, zU1U); }
    if (by_on) { rX[1] -= dual_y(qY, zYL, zYU); item(qY, X[1], zYL, zYU); }
    if (bv_on) { rX[3] -= dual_y(qV, zVL, zVU); item(qV, X[3], zVL, zVU); }
    if (rr_on) { rU[0] -= yR; item(qR, sR, vRL, vRU); prim = fmax(prim, fabs(rR)); }
#pragma unroll
    for (int j = 0; j < NOBS; ++j) if (ro_on[j]) {
      rX[0] -= vO[j] * gO0[j]; rX[1] -= vO[j] * gO1[j];
      item(qO, sO[j], vO[j], 0.0); prim = fmax(prim, fabs(rO[j]));
    }
    double dual = 0;
    if (xnode) {
#pragma unroll
      for (int i = 0; i < NX; ++i) dual = fmax(dual, fabs(rX[i]));
    }
    if (hasu) {
      dual = fmax(dual, fmax(fabs(rU[0]), fabs(rU[1])));
#pragma unroll
      for (int i = 0; i < NX; ++i) prim = fmax(prim, fabs(dfc[i]));
    }
    e_dual = wv::max(dual); e_prim = wv::max(prim); e_comp = wv::max(comp);
    sum_lam = wv::sum(sum_lam); sum_v = wv::sum(sum_v);
    e_sd = fmax(S_MAX, (sum_lam + sum_v) / fmax(1.0, n_lam + n_v)) / S_MAX;
    e_sc = fmax(S_MAX, sum_v / fmax(1.0, n_v)) / S_MAX;
    return fmax(e_dual / e_sd, fmax(e_prim, e_comp / e_sc));
  };

  // ----- per-lane constants of the Riccati sweep: lane = entry (i, j) of the 8x8 stage block -------------------
  const int ei = lane >> 3, ej = lane & 7;
  auto slotAB = [&](int r, int col) -> int {   // entry [A B | aug](r, col), r < NA, col < NW
    if (r < NX) {
      if (col < NX) {
        if (r == col) return E_ONE;
        if (r == 0 && col == 2) return E_A02;
        if (r == 0 && col == 3) return E_A03;
        if (r == 1 && col == 2) return E_A12;
        if (r == 1 && col == 3) return E_A13;
        if (r == 2 && col == 3) return E_A23;
        return E_ZERO;
      }
      if (r == 2 && col == 6) return E_B20;
      if (r == 3 && col == 7) return E_T;
      return E_ZERO;
    }
    return (col == r + 2) ? E_ONE : E_ZERO;    // Uprev_{k+1} = U_k
  };
  auto slotH = [&](int r, int col) -> int {
    int lo = r < col ? r : col, hi = r < col ? col : r;
    if (lo == 0 && hi == 0) return E_HXX;
    if (lo == 0 && hi == 1) return E_HXY;
    if (lo == 1 && hi == 1) return E_HYY;
    if (lo == 2 && hi == 2) return E_HPP;
    if (lo == 2 && hi == 3) return E_HPV;
    if (lo == 3 && hi == 3) return E_HVV;
    if (lo == 3 && hi == 6) return E_HVD;
    if (lo == 6 && hi == 6) return E_HDD;
    if (lo == 7 && hi == 7) return E_HAA;
    if (lo == 4 && hi == 4) return E_H44;
    if (lo == 5 && hi == 5) return E_H55;
    if (lo == 4 && hi == 6) return E_H46;
    if (lo == 5 && hi == 7) return E_H57;
    return E_ZERO;
  };
  int sABj[NA], sABi[NA];
#pragma unroll
  for (int r = 0; r < NA; ++r) { sABj[r] = slotAB(r, ej) * ld; sABi[r] = slotAB(r, ei) * ld; }
  const int sHij = slotH(ei, ej) * ld, sGi = (E_G0 + ei) * ld;

  // constant rows of the entry table
  if (isnode) { ent[E_ZERO * ld + k] = 0.0; ent[E_ONE * ld + k] = 1.0; ent[E_T * ld + k] = T; }

  double* Pst = lds + L.Pst; double* pst = lds + L.pst; double* Kst = lds + L.Kst; double* kffs = lds + L.kff;
  double* Wl = lds + L.W; double* ql = lds + L.q; double* Ml = lds + L.M; double* ml = lds + L.m;
  double* dXs = lds + L.dX; double* dUs = lds + L.dU; double* filt = lds + L.filt;
  int nfilt = 0;
  double theta_max = 0, theta_min = 0;
  double dw_last = 0.0;
  const double mu_floor = c.tol / (K_EPS + 1.0);
  double err0 = 0;

  // step (per node lane)
  double dX[NX] = {0, 0, 0, 0}, dU[NU] = {0, 0}, lamF[NX] = {0, 0, 0, 0};
  double dzU0L = 0, dzU0U = 0, dzU1L = 0, dzU1U = 0, dzYL = 0, dzYU = 0, dzVL = 0, dzVU = 0;
  double dsR = 0, dvRL = 0, dvRU = 0;
  double dsO[NOBS > 0 ? NOBS : 1], dvO[NOBS > 0 ? NOBS : 1];

  if (status != MPCB_ST_INFEASIBLE_X0) {
    eval_point();
    theta_max = 1e4 * fmax(1.0, theta); theta_min = 1e-4 * fmax(1.0, theta);

    for (iters = 0;; ++iters) {
      err0 = kkt_error(0.0);
      if (a.trace && b == a.trace_instance && lane == 0) {
        double* t = a.trace + (size_t)iters * 8;
        t[0] = mu; t[1] = err0; t[2] = theta; t[3] = fval;
      }
      if (err0 <= c.tol) { status = MPCB_ST_SOLVED; break; }
      if (iters >= c.max_iter) { status = MPCB_ST_MAXITER; break; }

      // barrier parameter update (monotone, Fiacco-McCormick)
      for (;;) {
        double em = kkt_error(mu);
        if (em <= K_EPS * mu && mu > mu_floor) {
          mu = fmax(mu_floor, fmin(K_MU * mu, pow(mu, TH_MU)));
          tau = fmax(TAU_MIN, 1.0 - mu);
          nfilt = 0;
        } else break;
      }

      // ----- condensed stage QP: lane k writes the compact entries of stage k ------------------------------------
      double ln[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) ln[i] = wv::shfl(lam[i], k + 1);
      double hxx = 0, hxy = 0, hyy = 0, hpp = 0, hpv = 0, hvv = 0, hvd = 0, hdd = 0, haa = 0, h44 = 0, h55 = 0, h46 = 0, h57 = 0;
      double g[NW] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (xcost) {
        hxx += os * 2 * c.Q[0]; hyy += os * 2 * c.Q[1]; hpp += os * 2 * c.Q[2]; hvv += os * 2 * c.Q[3];
#pragma unroll
        for (int i = 0; i < NX; ++i) g[i] += os * 2 * c.Q[i] * (X[i] - xs[i]);
      }
      if (hasu) {
        hdd += os * 2 * c.R[0]; haa += os * 2 * c.R[1];
        g[6] += os * 2 * c.R[0] * U[0]; g[7] += os * 2 * c.R[1] * U[1];
        if (k > 0 || c.du0_cost) {
          const double w0 = os * 2 * c.DR[0], w1 = os * 2 * c.DR[1];
          const double d0 = U[0] - (k ? Up0 : c.u_last[0]), d1 = U[1] - (k ? Up1 : c.u_last[1]);
          hdd += w0; h44 += w0; h46 -= w0; haa += w1; h55 += w1; h57 -= w1;
          g[6] += w0 * d0; g[4] -= w0 * d0; g[7] += w1 * d1; g[5] -= w1 * d1;
        }
        // sum_a lam_{k+1,a} T d2f_a
        const double v = X[3];
        hpp += T * (-ln[0] * v * cp - ln[1] * v * sp);
        hpv += T * (-ln[0] * sp + ln[1] * cp);
        hvd += T * ln[2] * sec2 * il;
        hdd += T * ln[2] * v * 2.0 * td * sec2 * il;
      }
      double sig, gb;
      if (bu0_on) { sig_gb(qU0, U[0], zU0L, zU0U, 0.0, mu, sig, gb); hdd += sig; g[6] -= gb; }
      if (bu1_on) { sig_gb(qU1, U[1], zU1L, zU1U, 0.0, mu, sig, gb); haa += sig; g[7] -= gb; }
      if (by_on) { sig_gb(qY, X[1], zYL, zYU, 0.0, mu, sig, gb); hyy += sig; g[1] -= gb; }
      if (bv_on) { sig_gb(qV, X[3], zVL, zVU, 0.0, mu, sig, gb); hvv += sig; g[3] -= gb; }
      if (rr_on) { sig_gb(qR, sR, vRL, vRU, rR, mu, sig, gb); hdd += sig; h44 += sig; h46 -= sig; g[6] -= gb; g[4] += gb; }
#pragma unroll
      for (int j = 0; j < NOBS; ++j) if (ro_on[j]) {
        sig_gb(qO, sO[j], vO[j], 0.0, rO[j], mu, sig, gb);
        hxx += sig * gO0[j] * gO0[j] - vO[j] * 2 * ix2[j];
        hxy += sig * gO0[j] * gO1[j];
        hyy += sig * gO1[j] * gO1[j] - vO[j] * 2 * iy2[j];
        g[0] -= gb * gO0[j]; g[1] -= gb * gO1[j];
      }
      if (isnode) {
        ent[E_A02 * ld + k] = hasu ? a02 : 0.0; ent[E_A03 * ld + k] = hasu ? a03 : 0.0;
        ent[E_A12 * ld + k] = hasu ? a12 : 0.0; ent[E_A13 * ld + k] = hasu ? a13 : 0.0;
        ent[E_A23 * ld + k] = hasu ? a23 : 0.0; ent[E_B20 * ld + k] = hasu ? b20 : 0.0;
#pragma unroll
        for (int i = 0; i < NX; ++i) ent[(E_D0 + i) * ld + k] = dfc[i];
#pragma unroll
        for (int i = 0; i < NW; ++i) ent[(E_G0 + i) * ld + k] = g[i];
        ent[E_HXY * ld + k] = hxy; ent[E_HPV * ld + k] = hpv; ent[E_HVD * ld + k] = hvd;
        ent[E_H44 * ld + k] = h44; ent[E_H55 * ld + k] = h55; ent[E_H46 * ld + k] = h46; ent[E_H57 * ld + k] = h57;
      }

      // ----- factorisation with inertia correction: backward Riccati sweep, lanes = entries ----------------------
      double dw = 0.0; bool first_try = true, fact_ok = false;
      for (int tries = 0; tries < 60; ++tries) {
        if (isnode) {
          ent[E_HXX * ld + k] = hxx + (xnode ? dw : 0.0); ent[E_HYY * ld + k] = hyy + (xnode ? dw : 0.0);
          ent[E_HPP * ld + k] = hpp + (xnode ? dw : 0.0); ent[E_HVV * ld + k] = hvv + (xnode ? dw : 0.0);
          ent[E_HDD * ld + k] = hdd + (hasu ? dw : 0.0); ent[E_HAA * ld + k] = haa + (hasu ? dw : 0.0);
        }
        wv::sync();
        // terminal: P_N = H_N (state block), p_N = g_N
        if (ei < NA && ej < NA) Pst[N * NA * NA + ei * NA + ej] = ent[sHij + N];
        if (ei < NA && ej == 0) pst[N * NA + ei] = ent[sGi + N];
        wv::sync();
        bool pd = true;
        for (int s = N - 1; s >= 0; --s) {
          double abj[NA], abi[NA];
#pragma unroll
          for (int r = 0; r < NA; ++r) { abj[r] = ent[sABj[r] + s]; abi[r] = ent[sABi[r] + s]; }
          const double hij = ent[sHij + s], gi = ent[sGi + s];
          const double* Pn = Pst + (s + 1) * NA * NA;
          const double* pn = pst + (s + 1) * NA;
          if (ei < NA) {
            double w = 0, q = pn[ei];
#pragma unroll
            for (int r = 0; r < NA; ++r) w += Pn[ei * NA + r] * abj[r];
#pragma unroll
            for (int r = 0; r < NX; ++r) q += Pn[ei * NA + r] * ent[(E_D0 + r) * ld + s];
            Wl[ej * NA + ei] = w;
            if (ej == 0) ql[ei] = q;
          }
          wv::sync();
          {
            double Mij = hij, mi = gi;
#pragma unroll
            for (int r = 0; r < NA; ++r) { Mij += abi[r] * Wl[ej * NA + r]; mi += abi[r] * ql[r]; }
            Ml[ei * NW + ej] = Mij;
            if (ej == 0) ml[ei] = mi;
          }
          wv::sync();
          const double m11 = Ml[NA * NW + NA], m12 = 0.5 * (Ml[NA * NW + NA + 1] + Ml[(NA + 1) * NW + NA]), m22 = Ml[(NA + 1) * NW + NA + 1];
          const double det = m11 * m22 - m12 * m12;
          if (!(m11 > 0) || !(det > 1e-14 * m11 * m22) || !isfinite(det)) { pd = false; break; }
          const double i11 = m22 / det, i12 = -m12 / det, i22 = m11 / det;
          const double mu6 = ml[NA], mu7 = ml[NA + 1];
          const double kf0 = -(i11 * mu6 + i12 * mu7), kf1 = -(i12 * mu6 + i22 * mu7);
          if (ei < NA && ej < NA) {
            const double M6j = Ml[NA * NW + ej], M7j = Ml[(NA + 1) * NW + ej];
            const double Mi6 = Ml[ei * NW + NA], Mi7 = Ml[ei * NW + NA + 1];
            const double K0j = -(i11 * M6j + i12 * M7j), K1j = -(i12 * M6j + i22 * M7j);
            Pst[s * NA * NA + ei * NA + ej] = Ml[ei * NW + ej] + Mi6 * K0j + Mi7 * K1j;
            if (ei == 0) { Kst[s * 2 * NA + ej] = K0j; Kst[s * 2 * NA + NA + ej] = K1j; }
            if (ej == 0) pst[s * NA + ei] = ml[ei] + Mi6 * kf0 + Mi7 * kf1;
          }
          if (lane == 0) { kffs[s * 2] = kf0; kffs[s * 2 + 1] = kf1; }
          wv::sync();
          // symmetrise P_s (keeps the recursion symmetric to rounding)
          if (ei < NA && ej < NA && ei < ej) {
            double sym = 0.5 * (Pst[s * NA * NA + ei * NA + ej] + Pst[s * NA * NA + ej * NA + ei]);
            Pst[s * NA * NA + ei * NA + ej] = sym; Pst[s * NA * NA + ej * NA + ei] = sym;
          }
          wv::sync();
        }
        if (pd) { fact_ok = true; if (dw > 0) dw_last = dw; break; }
        if (first_try) { dw = (dw_last == 0.0) ? DW_FIRST : fmax(DW_MIN, KW_MINUS * dw_last); first_try = false; }
        else dw *= (dw_last == 0.0) ? KW_PLUS_FIRST : KW_PLUS;
        if (dw > DW_MAX) break;
      }
      if (!fact_ok) { status = MPCB_ST_NUMERIC; break; }

      // ----- forward roll-out of the step (every lane carries the same 6-vector) ---------------------------------
      {
        double dx[NA] = {0, 0, 0, 0, 0, 0};
        if (lane == 0) { dXs[0] = 0; dXs[1] = 0; dXs[2] = 0; dXs[3] = 0; }
        for (int s = 0; s < N; ++s) {
          const double* Ks = Kst + s * 2 * NA;
          double du0 = kffs[s * 2], du1 = kffs[s * 2 + 1];
#pragma unroll
          for (int r = 0; r < NA; ++r) { du0 += Ks[r] * dx[r]; du1 += Ks[NA + r] * dx[r]; }
          const double A02 = ent[E_A02 * ld + s], A03 = ent[E_A03 * ld + s], A12 = ent[E_A12 * ld + s], A13 = ent[E_A13 * ld + s];
          const double A23 = ent[E_A23 * ld + s], B20 = ent[E_B20 * ld + s];
          const double n0 = dx[0] + A02 * dx[2] + A03 * dx[3] + ent[E_D0 * ld + s];
          const double n1 = dx[1] + A12 * dx[2] + A13 * dx[3] + ent[E_D1 * ld + s];
          const double n2 = dx[2] + A23 * dx[3] + B20 * du0 + ent[E_D2 * ld + s];
          const double n3 = dx[3] + T * du1 + ent[E_D3 * ld + s];
          dx[0] = n0; dx[1] = n1; dx[2] = n2; dx[3] = n3; dx[4] = du0; dx[5] = du1;
          if (lane == 0) {
            dUs[s * 2] = du0; dUs[s * 2 + 1] = du1;
            dXs[(s + 1) * NX + 0] = n0; dXs[(s + 1) * NX + 1] = n1; dXs[(s + 1) * NX + 2] = n2; dXs[(s + 1) * NX + 3] = n3;
          }
        }
        wv::sync();
#pragma unroll
        for (int i = 0; i < NX; ++i) dX[i] = isnode ? dXs[k * NX + i] : 0.0;
        dU[0] = hasu ? dUs[k * 2] : 0.0; dU[1] = hasu ? dUs[k * 2 + 1] : 0.0;
      }
      const double dUp0 = wv::shfl(dU[0], k - 1), dUp1 = wv::shfl(dU[1], k - 1);   // dU_{k-1}
      // costate of the full step: lamF_k = P_k [dX_k; dU_{k-1}] + p_k  (node-parallel)
      if (xnode) {
        const double* Pk = Pst + k * NA * NA;
        const double dxa[NA] = {dX[0], dX[1], dX[2], dX[3], dUp0, dUp1};
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          double s = pst[k * NA + i];
#pragma unroll
          for (int r = 0; r < NA; ++r) s += Pk[i * NA + r] * dxa[r];
          lamF[i] = s;
        }
      }
      // slack and dual steps
      auto dstep = [&](const Bnd& q, double s, double ds, double vL, double vU, double& dvL, double& dvU) {
        dvL = 0; dvU = 0;
        if (q.hasL) { double d = s - q.L; dvL = mu / d - vL - vL / d * ds; }
        if (q.hasU) { double d = q.U - s; dvU = mu / d - vU + vU / d * ds; }
      };
      if (bu0_on) dstep(qU0, U[0], dU[0], zU0L, zU0U, dzU0L, dzU0U);
      if (bu1_on) dstep(qU1, U[1], dU[1], zU1L, zU1U, dzU1L, dzU1U);
      if (by_on) dstep(qY, X[1], dX[1], zYL, zYU, dzYL, dzYU);
      if (bv_on) dstep(qV, X[3], dX[3], zVL, zVU, dzVL, dzVU);
      if (rr_on) { dsR = (dU[0] - dUp0) + rR; dstep(qR, sR, dsR, vRL, vRU, dvRL, dvRU); }
#pragma unroll
      for (int j = 0; j < NOBS; ++j) {
        dsO[j] = 0; dvO[j] = 0;
        if (ro_on[j]) { double dummy; dsO[j] = gO0[j] * dX[0] + gO1[j] * dX[1] + rO[j]; dstep(qO, sO[j], dsO[j], vO[j], 0.0, dvO[j], dummy); }
      }

      // ----- fraction to the boundary -----------------------------------------------------------------------------
      double a_pr = 1.0, a_du = 1.0;
      auto ftb = [&](const Bnd& q, double s, double ds, double vL, double vU, double dvL, double dvU) {
        if (q.hasL) {
          if (ds < 0) a_pr = fmin(a_pr, -tau * (s - q.L) / ds);
          if (dvL < 0) a_du = fmin(a_du, -tau * vL / dvL);
        }
        if (q.hasU) {
          if (ds > 0) a_pr = fmin(a_pr, tau * (q.U - s) / ds);
          if (dvU < 0) a_du = fmin(a_du, -tau * vU / dvU);
        }
      };
      if (bu0_on) ftb(qU0, U[0], dU[0], zU0L, zU0U, dzU0L, dzU0U);
      if (bu1_on) ftb(qU1, U[1], dU[1], zU1L, zU1U, dzU1L, dzU1U);
      if (by_on) ftb(qY, X[1], dX[1], zYL, zYU, dzYL, dzYU);
      if (bv_on) ftb(qV, X[3], dX[3], zVL, zVU, dzVL, dzVU);
      if (rr_on) ftb(qR, sR, dsR, vRL, vRU, dvRL, dvRU);
#pragma unroll
      for (int j = 0; j < NOBS; ++j) if (ro_on[j]) ftb(qO, sO[j], dsO[j], vO[j], 0.0, dvO[j], 0.0);
      a_pr = wv::min(a_pr); a_du = wv::min(a_du);

      // ----- directional derivative of the barrier function -----------------------------------------------------
      double dphi;
      {
        double d = 0;
        if (xcost) {
#pragma unroll
          for (int i = 0; i < NX; ++i) d += os * 2 * c.Q[i] * (X[i] - xs[i]) * dX[i];
        }
        if (hasu) {
          d += os * 2 * c.R[0] * U[0] * dU[0] + os * 2 * c.R[1] * U[1] * dU[1];
          if (k > 0 || c.du0_cost) {
            d += os * 2 * c.DR[0] * (U[0] - (k ? Up0 : c.u_last[0])) * (dU[0] - (k ? dUp0 : 0.0));
            d += os * 2 * c.DR[1] * (U[1] - (k ? Up1 : c.u_last[1])) * (dU[1] - (k ? dUp1 : 0.0));
          }
        }
        auto bd = [&](const Bnd& q, double s, double ds) {
          if (q.hasL) d -= mu * ds / (s - q.L);
          if (q.hasU) d += mu * ds / (q.U - s);
        };
        if (bu0_on) bd(qU0, U[0], dU[0]);
        if (bu1_on) bd(qU1, U[1], dU[1]);
        if (by_on) bd(qY, X[1], dX[1]);
        if (bv_on) bd(qV, X[3], dX[3]);
        if (rr_on) bd(qR, sR, dsR);
#pragma unroll
        for (int j = 0; j < NOBS; ++j) if (ro_on[j]) bd(qO, sO[j], dsO[j]);
        dphi = wv::sum(d);
      }
      const double phi0 = barrier_phi(mu), th0 = theta;
      double a_min;
      if (dphi < 0) {
        a_min = fmin(G_THETA, G_PHI * th0 / (-dphi));
        if (th0 <= theta_min) a_min = fmin(a_min, DELTA * pow(th0, S_THETA) / pow(-dphi, S_PHI));
      } else a_min = G_THETA;
      a_min *= G_ALPHA;

      // ----- filter line search: trial evaluations are lane-parallel ----------------------------------------------
      double alpha = a_pr; bool accepted = false, armijo_type = false;
      for (;;) {
        double Xt[NX], Ut[NU];
#pragma unroll
        for (int i = 0; i < NX; ++i) Xt[i] = X[i] + alpha * dX[i];
        Ut[0] = U[0] + alpha * dU[0]; Ut[1] = U[1] + alpha * dU[1];
        double th = 0, bs = 0; bool ok = true;
        {
          const double s_ = sin(Xt[2]), c_ = cos(Xt[2]), t_ = tan(Ut[0]), v = Xt[3];
          const double Ft[NX] = {Xt[0] + T * (v * c_), Xt[1] + T * (v * s_), Xt[2] + T * (v * t_ * il), Xt[3] + T * Ut[1]};
#pragma unroll
          for (int i = 0; i < NX; ++i) { double xn = wv::shfl(Xt[i], k + 1); if (hasu) th += fabs(Ft[i] - xn); }
        }
        const double Utp0 = wv::shfl(Ut[0], k - 1), Utp1 = wv::shfl(Ut[1], k - 1);
        auto bar = [&](const Bnd& q, double s) {
          if (q.hasL) { double d = s - q.L; if (!(d > 0)) ok = false; else bs -= log(d); }
          if (q.hasU) { double d = q.U - s; if (!(d > 0)) ok = false; else bs -= log(d); }
        };
        if (bu0_on) bar(qU0, Ut[0]);
        if (bu1_on) bar(qU1, Ut[1]);
        if (by_on) bar(qY, Xt[1]);
        if (bv_on) bar(qV, Xt[3]);
        if (rr_on) { double s = sR + alpha * dsR; bar(qR, s); th += fabs((Ut[0] - Utp0) - s); }
#pragma unroll
        for (int j = 0; j < NOBS; ++j) if (ro_on[j]) { double s = sO[j] + alpha * dsO[j]; bar(qO, s); th += fabs(hval(j, Xt[0], Xt[1]) - s); }
        const double tht = wv::sum(th);
        const double ft = objective(Xt, Ut, Utp0, Utp1);
        const double phit = os * ft + mu * wv::sum(bs);
        ok = wv::all(ok) && isfinite(tht) && isfinite(phit);
        if (ok && tht <= theta_max) {
          bool fok = true;
          for (int e = lane; e < nfilt; e += 64) if (tht >= filt[2 * e] && phit >= filt[2 * e + 1]) fok = false;
          if (wv::all(fok)) {
            const bool sw = dphi < 0 && alpha * pow(-dphi, S_PHI) > DELTA * pow(th0, S_THETA);
            if (th0 <= theta_min && sw) {
              if (phit <= phi0 + ETA_PHI * alpha * dphi || phit - phi0 <= 10 * 2.2e-16 * fabs(phi0)) { accepted = true; armijo_type = true; }
            } else if (tht <= (1 - G_THETA) * th0 || phit <= phi0 - G_PHI * th0) accepted = true;
          }
        }
        if (accepted) break;
        alpha *= 0.5;
        if (alpha < a_min || alpha < 1e-16) break;
      }
      if (a.trace && b == a.trace_instance && lane == 0) {
        double* t = a.trace + (size_t)iters * 8;
        t[4] = a_pr; t[5] = accepted ? alpha : 0.0; t[6] = a_du; t[7] = dw;
      }
      if (!accepted) { status = MPCB_ST_LINESEARCH; break; }
      if (!armijo_type) {
        if (nfilt < FILTER_MAX) {
          if (lane == 0) { filt[2 * nfilt] = (1 - G_THETA) * th0; filt[2 * nfilt + 1] = phi0 - G_PHI * th0; }
          ++nfilt;
        }
        wv::sync();
      }

      // ----- apply the step ---------------------------------------------------------------------------------------
      if (xnode) {
#pragma unroll
        for (int i = 0; i < NX; ++i) { X[i] += alpha * dX[i]; lam[i] += alpha * (lamF[i] - lam[i]); }
      }
      if (hasu) { U[0] += alpha * dU[0]; U[1] += alpha * dU[1]; }
      auto upd = [&](const Bnd& q, double s, double& vL, double& vU, double dvL, double dvU) {
        if (q.hasL) { vL += a_du * dvL; double d = s - q.L; vL = fmax(fmin(vL, K_SIGMA * mu / d), mu / (K_SIGMA * d)); }
        if (q.hasU) { vU += a_du * dvU; double d = q.U - s; vU = fmax(fmin(vU, K_SIGMA * mu / d), mu / (K_SIGMA * d)); }
      };
      if (bu0_on) upd(qU0, U[0], zU0L, zU0U, dzU0L, dzU0U);
      if (bu1_on) upd(qU1, U[1], zU1L, zU1U, dzU1L, dzU1U);
      if (by_on) upd(qY, X[1], zYL, zYU, dzYL, dzYU);
      if (bv_on) upd(qV, X[3], zVL, zVU, dzVL, dzVU);
      if (rr_on) { sR += alpha * dsR; upd(qR, sR, vRL, vRU, dvRL, dvRU); }
#pragma unroll
      for (int j = 0; j < NOBS; ++j) if (ro_on[j]) { double dummy = 0; sO[j] += alpha * dsO[j]; upd(qO, sO[j], vO[j], dummy, dvO[j], 0.0); }
      eval_point();
      if (!isfinite(theta) || !isfinite(fval)) { status = MPCB_ST_NUMERIC; break; }
    }
  } else {
    eval_point();
  }

  // ----- outputs (reference ordering), staged through LDS for coalesced stores ----------------------------------
  wv::sync();
  if (hasu) { zbuf[NU * k] = U[0]; zbuf[NU * k + 1] = U[1]; }
  if (isnode) {
#pragma unroll
    for (int i = 0; i < NX; ++i) zbuf[NU * N + NX * k + i] = X[i];
  }
  wv::sync();
  for (int i = lane; i < nz; i += 64) a.z[(size_t)b * nz + i] = zbuf[i];
  const double e_final = kkt_error(0.0);
  if (lane == 0) {
    if (a.obj) a.obj[b] = fval;
    if (a.status) a.status[b] = status;
    if (a.iters) a.iters[b] = iters;
    if (a.kkt) { double* q = a.kkt + (size_t)b * 4; q[0] = e_final; q[1] = e_prim; q[2] = e_dual / os; q[3] = mu; }
  }
  if (a.want_mult && a.lam_x) {
    wv::sync();
    for (int i = lane; i < nz; i += 64) zbuf[i] = 0.0;
    wv::sync();
    if (bu0_on) zbuf[NU * k] = -dual_y(qU0, zU0L, zU0U) / os;
    if (bu1_on) zbuf[NU * k + 1] = -dual_y(qU1, zU1L, zU1U) / os;
    if (by_on) zbuf[NU * N + NX * k + 1] = -dual_y(qY, zYL, zYU) / os;
    if (bv_on) zbuf[NU * N + NX * k + 3] = -dual_y(qV, zVL, zVU) / os;
    wv::sync();
    for (int i = lane; i < nz; i += 64) a.lam_x[(size_t)b * nz + i] = zbuf[i];
  }
  if (a.want_mult && a.lam_g) {
    // rows: [X_0 - P](NX), dynamics (NX*N), rate (N-1 if on), obstacles (rows * nobs)   kin.py:190-247
    double* out = a.lam_g + (size_t)b * a.ng;
    const int r_dyn = NX, r_rate = NX + NX * N, n_rate = qR.on ? N - 1 : 0, r_obs = r_rate + n_rate;
    double ln[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) ln[i] = wv::shfl(lam[i], k + 1);
    if (xnode) {
#pragma unroll
      for (int i = 0; i < NX; ++i) out[r_dyn + NX * (k - 1) + i] = -lam[i] / os;
    }
    if (k == 0) {   // stationarity wrt the pinned X_0
      const double At[NX] = {ln[0], ln[1], a02 * ln[0] + a12 * ln[1] + ln[2], a03 * ln[0] + a13 * ln[1] + a23 * ln[2] + ln[3]};
#pragma unroll
      for (int i = 0; i < NX; ++i) out[i] = -2 * c.Q[i] * (X[i] - xs[i]) - At[i] / os;
    }
    if (rr_on) out[r_rate + (k - 1)] = -dual_y(qR, vRL, vRU) / os;
    if (isnode) {
      const int row = (c.obs_mode == MPCB_OBS_KEEPOUT) ? k : k - 1;
#pragma unroll
      for (int j = 0; j < NOBS; ++j) if (j < nobs && row >= 0 && row <= last_row) out[r_obs + row * nobs + j] = ro_on[j] ? -vO[j] / os : 0.0;
    }
  }
}
