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
//     line-search trial evaluations are lane-parallel; scalars are combined with fused wave reductions.
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
using std::log; using std::fabs; using std::sqrt; using std::pow; using std::fmin; using std::fmax; using std::rint; using std::fma;
using std::isfinite;
#endif

struct MpcbKArgs {
  mpcb_config cfg;
  int32_t B, nz, ng, obs_kind, want_mult, trace_instance;
  int32_t st_stride;   // status / iters of instance b go to index b * st_stride (the closed loop writes its [B, steps] histories directly)
  int32_t pass;        // MPCB_PASS_FIRST: main phase from the caller's start; an instance that needs the restoration phase ends with
                       //   MPCB_ST_NEEDS_RESTO and a hand-over record in `work`;
                       // MPCB_PASS_RESTO: restoration pass (RESTO kernel instantiation), only the MPCB_ST_NEEDS_RESTO instances run;
                       // MPCB_PASS_SECOND: the lean kernel once more from the second start (cfg.second_start), only the instances whose first
                       //   attempt did not succeed run; mpcb_api.hip follows it with another MPCB_PASS_RESTO launch
  const double* tgrid; // [N] step length of every stage (mpcb_set_time_grid), NULL = cfg.T everywhere
  double* work;        // [B][WK_SIZE] hand-over records between the passes (device scratch of the handle), NULL when cfg.restoration == 0
  const double *x0, *xs, *obs, *z0;
  double *z, *obj, *kkt, *lam_g, *lam_x;
  int32_t *status, *iters;
  double* trace;   // optional: [max_iter + 1][8] log of instance trace_instance (debug / parity tests)
};

// MPCB_STAMPS (diagnostic build only, never shipped): overwrite trace columns 4..7 with cycle counts of the phases of
// an iteration (condense+KKT, Riccati, forward+ratios, line search) instead of the step lengths.
#if defined(MPCB_STAMPS) && !defined(MPCB_WAVE_EMU)
#define MPCB_STAMP(var) unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define MPCB_STAMP(var) unsigned long long var = 0
#endif
#ifndef MPCB_WAVE_EMU
#define MPCB_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define MPCB_SCHED_FENCE() ((void)0)
#endif

// MPCB_LATE_TRACE (diagnostic builds only, tools/sync_ab.sh): the trace pointer late-loaded as well — together with the late-loaded output
// pointers this build of kin<8, GEN> returned NONDETERMINISTIC wrong statuses on the GPU
#ifdef MPCB_LATE_TRACE
#define MPCB_TRACE_ARGS(a) wv::late_args(a)
#else
#define MPCB_TRACE_ARGS(a) (&(a))
#endif
// internal statuses between the passes of a solve; never returned to the caller
#define MPCB_ST_NEEDS_RESTO 7
#define MPCB_PASS_FIRST 0
#define MPCB_PASS_SECOND 1
#define MPCB_PASS_RESTO 2

namespace mpcbk {

constexpr int NU = 2;
constexpr int FILTER_MAX = 64;
// restoration phase (oracle: Solver::restoration()): penalty rho, required reduction of the violation, safety
// factor of the duality-gap certificate; early entry after TRIG_K accepted steps in a row shorter than TRIG_ALPHA that together
// reduced theta by less than the factor TRIG_THETA
constexpr double RS_RHO = 1000.0, RS_KAPPA = 0.5, RS_GAP = 10.0, TRIG_ALPHA = 0.05, TRIG_THETA = 0.8;
constexpr int TRIG_K = 5;
// effort bounds of the restoration phase per instance: entries into it, iterations inside it (all entries together); beyond them the
// solve ends with MPCB_ST_RESTO_FAILED (a launch ends with its slowest instance: the phase must not run to max_iter)
constexpr int RS_MAX_CALLS = 3, RS_MAX_ITERS = 40;
// hand-over record of an instance that needs the restoration pass
constexpr int WK_MU = 0, WK_THMAX = 1, WK_THMIN = 2, WK_ITERS = 3, WK_DW = 4, WK_ITPREV = 5, WK_START = 6, WK_SIZE = 8;   // WK_ITPREV: iterations of the failed first attempt; WK_START: 1 = the attempt handed over started from z = 0
// per-node cost table of the RESTO instantiations, [row][N+2] in LDS (lane = node, lanes beyond the last node share the dummy column N+1): scaled weights osc*2*Qc, reference point,
// osc*2*Rc, control reference, the unscaled weights Qc, Rc, and the rate-cost weights osc*2*DRc, DRc
enum CostRow { CT_WQ = 0, CT_XR = 4, CT_WR = 8, CT_UR = 10, CT_QQ = 12, CT_RR = 16, CT_WDR = 18, CT_DRR = 20, CT_ROWS = 22 };

// compact stage entries of the kinematic model, variable order of a stage: [x, y, phi, v, dprev, aprev, d, a]
enum KinEnt {
  E_ZERO = 0, E_ONE,                                    // ([A B], the defect and the step length live in the fw rows only)
  E_G0, E_G1, E_G2, E_G3, E_G4, E_G5, E_G6, E_G7,       // condensed gradient
  E_HXX, E_HXY, E_HYY, E_HPP, E_HPV, E_HVV, E_HVD, E_HDD, E_HAA, E_H44, E_H55, E_H46, E_H57,
  E_HXP, E_HXV, E_HYP, E_HYV,                           // only with general-gamma CBF rows: the GEN kernels' tables have these four rows more
  KIN_NENT,
  // RK4 instantiations (never together with GEN) use the same four rows for the pairs (phi,delta) (phi,a) (v,a) (delta,a) of the step's Hessian
  E_HPD = E_HXP, E_HPA = E_HXV, E_HVA = E_HYP, E_HDA = E_HYV
};

// LDS layout (doubles).  ld = (N+1)|1: odd leading dimension of the [entry][node] tables.
//   Pst [N+1][PST]   P_k (6x6, row-major, slots 0..35), p_k (slots 36..41), a permanent 0.0 (slot 42: unit term of lanes
//                    without one), pad slot 43 (p stores of non-affine lanes), pad slots 44..47 (P stores of lanes without a
//                    P entry)
//   fw  [N+1][FWS]   the stage's [A | d | B], step length and gains as six records of FWR doubles, one per component of
//                    [dX_{k+1}; dU_k]: coefficients of [dX_k; dU_{k-1}] (6), constant, coefficient of the stage's control (b20 on
//                    dU_k[0] in record 2, T_k on dU_k[1] in record 3).  Records 0..3 are the rows of [A | d | B] (unit and zero
//                    entries written once per solve, a02 a03 a12 a13 a23 b20 d0..d3 T_k per iteration), records 4, 5 the gain rows
//                    K and kff (written by the sweep).  Read by the forward roll-out (lane i its record) AND by the sweep (each
//                    lane its own slots of [A B | d]): a single copy.  Two pad slots take the K stores of lanes without a K entry.
constexpr int PST = 50, PS_P = 36, PS_ZERO = 42, PS_PADP = 43, PS_PAD = 44;
constexpr int FWR = 8, FWS = 50, FW_C0 = 6, FW_BX = 7, FW_PAD = 48, FW_ZERO = 1, FW_ONE = 0, WSZ = 136, W_ZERO = 64, W_STAGE = 72;
// constant block: uniform numbers of the instance that the node-parallel phases read from LDS (one ds_read, short live range)
// instead of holding ~25 SGPR pairs through the whole solve
constexpr int CS_WQ = 0, CS_WR = 4, CS_WDR = 6, CS_Q = 8, CS_R = 12, CS_DR = 14, CS_UL = 16, CS_XS = 18, CS_ACC = 24, CSZ = 26;   // CS_ACC: state of the acceptable-point test (objective at the previous check, iterations in a row)
struct Layout {
  int ld, ent, Pst, fw, W, cst, filt, zbuf, ct, obl, total;
};
// obstacle-row capacity of the kernel instantiation that serves n obstacles, and how many of its obstacle constants (centre, 1/sX^2,
// 1/sY^2: four doubles per obstacle and lane) live in LDS instead of registers: none up to 3, all of them above (the 5- and
// 8-obstacle instantiations are far beyond the register file otherwise)
MPCB_HD int obs_capacity_kin(int n, bool gen = false) { return n <= 0 ? (gen ? 1 : 0) : n == 1 ? 1 : n <= 3 ? 3 : (n <= 5 && !gen) ? 5 : 8; }   // (no GEN<5> instantiation)
MPCB_HD int obs_in_lds(int capacity) { return capacity > 3 ? capacity : 0; }
MPCB_HD Layout layout_kin(int N, int nz, bool resto = false, int nobl = 0, bool gen = true) {
  Layout L;
  const int N1 = N + 1;
  L.ld = N1 | 1;
  int o = 0;
  L.Pst = o; o += N1 * PST;          // first: 16-byte aligned rows for wide LDS reads
  L.fw = o; o += N1 * FWS; o += o & 1;
  L.W = o; o += WSZ;                 // W^T (8 columns x 6) + 16 pad slots + a permanent 0.0 (W_ZERO) + the staging block of the sweep's tail (64)
  L.cst = o; o += CSZ;
  L.filt = o; o += 2 * FILTER_MAX;
  L.ent = o; o += (gen ? KIN_NENT : KIN_NENT - 4) * L.ld;
  L.zbuf = L.Pst;                    // staging of z rows aliases the Riccati storage (used before / after the loop)
  (void)nz;
  L.ct = o; if (resto) o += CT_ROWS * (N + 2);
  L.obl = o; o += 4 * nobl * (N + 2);
  L.total = o;
  return L;
}

struct Bnd { double L, U; bool hasL, hasU, on; };
MPCB_DEV Bnd mk_bnd(double L, double U, double relax) {
  Bnd q;
  q.hasL = L > -1e300; q.hasU = U < 1e300; q.on = q.hasL || q.hasU;
  q.L = wv::uni(q.hasL ? L - relax * fmax(1.0, fabs(L)) : L);
  q.U = wv::uni(q.hasU ? U + relax * fmax(1.0, fabs(U)) : U);
  return q;
}
// the same for bounds that differ from lane to lane (rate rows under a time grid)
MPCB_DEV Bnd mk_bnd_lane(double L, double U, double relax) {
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

// One inequality item in registers: duals and the cached reciprocals of the two distances to the bounds.
// For a variable box the slack IS the variable; for a general row `s` is its own iterate.
struct Item { double vL, vU, iL, iU; };
MPCB_DEV void item_recip(const Bnd& q, double s, Item& it) {
  it.iL = q.hasL ? wv::rcp(s - q.L) : 0.0;
  it.iU = q.hasU ? wv::rcp(q.U - s) : 0.0;
}
MPCB_DEV double item_y(const Bnd& q, const Item& it) { return (q.hasL ? it.vL : 0.0) - (q.hasU ? it.vU : 0.0); }
// Sigma = vL/(s-L) + vU/(U-s);  gb = mu/(s-L) - mu/(U-s) - Sigma r
MPCB_DEV void item_sig_gb(const Item& it, double r, double mu, double& sig, double& gb) {
  sig = it.vL * it.iL + it.vU * it.iU;
  gb = mu * (it.iL - it.iU) - sig * r;
}
MPCB_DEV void item_dv(const Bnd& q, const Item& it, double ds, double mu, double& dvL, double& dvU) {
  dvL = q.hasL ? mu * it.iL - it.vL - it.vL * it.iL * ds : 0.0;
  dvU = q.hasU ? mu * it.iU - it.vU + it.vU * it.iU * ds : 0.0;
}

// sin and cos for |x| up to ~1e5 (headings and steering angles): Cody-Waite reduction by pi/2 in two pieces and the
// fdlibm kernel polynomials on [-pi/4, pi/4]; < 1 ulp there.  Much smaller than the library's Payne-Hanek path.
MPCB_DEV void sincos_b(double x, double& s, double& c) {
  const double kq = rint(x * 6.36619772367581382433e-01);
  double r = fma(-kq, 1.57079632673412561417e+00, x);
  r = fma(-kq, 6.07710050650619224932e-11, r);
  const double z = r * r;
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                                        -1.98412698298579493134e-04), 8.33333333332248946124e-03), -1.66666666666666324348e-01);
  const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                                        2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
  const double sk = fma(r * z, ps, r);
  const double ck = fma(z * z, pc, fma(-0.5, z, 1.0));
  const int q = ((int)kq) & 3;
  s = (q == 0) ? sk : (q == 1) ? ck : (q == 2) ? -sk : -ck;
  c = (q == 0) ? ck : (q == 1) ? -sk : (q == 2) ? -ck : sk;
}

// ---- classical fourth-order Runge-Kutta step of the kinematic bicycle with the control held over the interval (cfg.integrator =
// MPCB_INT_RK4).  With p = (phi, v, delta, a), c = tan(delta) / L and h = T / 2 the four stage states are
//     phi_s = phi + alpha_s v c + beta_s a c,   v_s = v + gamma_s a,   (alpha, beta, gamma)_s = (0,0,0), (h,0,h), (h,h^2,h), (T,Th,T)
// (the speed and the heading are polynomials in time, which RK4 integrates exactly), and
//     x+ = x + T/6 sum_s w_s v_s cos phi_s,  y+ = y + T/6 sum_s w_s v_s sin phi_s,  phi+ = phi + T (v + T a / 2) c,  v+ = v + T a,  w = (1,2,2,1).
struct KinRkJac { double a02, a03, a12, a13, a23, b00, b01, b10, b11, b20, b21; };   // dF/d(phi, v) and dF/d(delta, a); b31 = T, the rest is the identity / zero
struct KinRkHess { double pp, pv, pd, pa, vv, vd, va, dd, da, aa; };                 // sum_r lam_r d2F_r over the pairs of (phi, v, delta, a)
MPCB_DEV void kin_rk4_stage(int s, double T, double& al, double& be, double& ga, double& w) {
  const double h = 0.5 * T;
  al = s == 0 ? 0.0 : s == 3 ? T : h; be = s == 2 ? h * h : s == 3 ? T * h : 0.0; ga = s == 0 ? 0.0 : s == 3 ? T : h; w = (s == 1 || s == 2) ? 2.0 : 1.0;
}
// F = the step from (X, U); sp, cp = sin, cos of the heading, td = tan(delta)
MPCB_DEV void kin_rk4_step(const double* X, const double* U, double T, double il, double sp, double cp, double td, double* F) {
  const double v = X[3], a = U[1], c = td * il;
  double sx = v * cp, sy = v * sp;
#pragma unroll
  for (int s = 1; s < 4; ++s) {
    double al, be, ga, w; kin_rk4_stage(s, T, al, be, ga, w);
    double ss, cs; sincos_b(X[2] + (al * v + be * a) * c, ss, cs);
    const double vs = v + ga * a;
    sx += w * (vs * cs); sy += w * (vs * ss);
  }
  F[0] = X[0] + (T / 6.0) * sx; F[1] = X[1] + (T / 6.0) * sy;
  F[2] = X[2] + T * ((v + 0.5 * T * a) * c); F[3] = v + T * a;
}
// Jacobian entries and  sum_r lam_r d2F_r  (lam = costate of the step's four rows); sec2 = 1 / cos^2(delta)
MPCB_DEV void kin_rk4_derivs(const double* X, const double* U, double T, double il, double sp, double cp, double td, double sec2,
                             const double* lam, KinRkJac& J, KinRkHess& H) {
  const double v = X[3], a = U[1], c = td * il, c1 = sec2 * il, c2 = 2.0 * td * sec2 * il;
  double gx[4] = {0, 0, 0, 0}, gy[4] = {0, 0, 0, 0}, hh[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    double al, be, ga, w; kin_rk4_stage(s, T, al, be, ga, w);
    double ss = sp, cs = cp;
    if (s > 0) sincos_b(X[2] + (al * v + be * a) * c, ss, cs);
    const double vs = v + ga * a;
    const double fp[4] = {1.0, al * c, (al * v + be * a) * c1, be * c};        // d phi_s / d(phi, v, delta, a)
    const double vp[4] = {0.0, 1.0, 0.0, ga};                                   // d v_s / d(...)
    // second derivatives of phi_s: (v,delta) = al c', (delta,delta) = (al v + be a) c'', (delta,a) = be c'
    int q = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      gx[i] += w * (vp[i] * cs - vs * ss * fp[i]);
      gy[i] += w * (vp[i] * ss + vs * cs * fp[i]);
#pragma unroll
      for (int j = i; j < 4; ++j, ++q) {
        const double fij = (i == 1 && j == 2) ? al * c1 : (i == 2 && j == 2) ? (al * v + be * a) * c2 : (i == 2 && j == 3) ? be * c1 : 0.0;
        const double cross = vp[i] * fp[j] + vp[j] * fp[i], quad = vs * fp[i] * fp[j];
        const double gxx = -cross * ss - quad * cs - vs * ss * fij;
        const double gyy = cross * cs - quad * ss + vs * cs * fij;
        hh[q] += w * (lam[0] * gxx + lam[1] * gyy);
      }
    }
  }
  const double t6 = T / 6.0;
  J.a02 = t6 * gx[0]; J.a03 = t6 * gx[1]; J.b00 = t6 * gx[2]; J.b01 = t6 * gx[3];
  J.a12 = t6 * gy[0]; J.a13 = t6 * gy[1]; J.b10 = t6 * gy[2]; J.b11 = t6 * gy[3];
  J.a23 = T * c; J.b20 = T * (v + 0.5 * T * a) * c1; J.b21 = 0.5 * T * T * c;
  // pairs in the order (p,p) (p,v) (p,d) (p,a) (v,v) (v,d) (v,a) (d,d) (d,a) (a,a); the heading row adds (v,d), (d,d), (d,a)
  H.pp = t6 * hh[0]; H.pv = t6 * hh[1]; H.pd = t6 * hh[2]; H.pa = t6 * hh[3]; H.vv = t6 * hh[4];
  H.vd = t6 * hh[5] + lam[2] * T * c1; H.va = t6 * hh[6];
  H.dd = t6 * hh[7] + lam[2] * T * (v + 0.5 * T * a) * c2; H.da = t6 * hh[8] + lam[2] * 0.5 * T * T * c1; H.aa = t6 * hh[9];
}

// IPOPT constants (Waechter & Biegler 2006 / IPOPT option defaults)
constexpr double K_EPS = 10.0, K_MU = 0.2, TAU_MIN = 0.99;
constexpr double G_THETA = 1e-5, G_PHI = 1e-8, DELTA = 1.0, S_THETA = 1.1, S_PHI = 2.3, ETA_PHI = 1e-8, G_ALPHA = 0.05;
constexpr double K_SIGMA = 1e10, S_MAX = 100.0;
constexpr double DW_FIRST = 1e-4, DW_MIN = 1e-20, DW_MAX = 1e40, KW_MINUS = 1.0 / 3.0, KW_PLUS = 8.0, KW_PLUS_FIRST = 100.0;

}  // namespace mpcbk

// =============================================================================================================
// Kinematic bicycle, NOBS = compile-time capacity of obstacle rows per node (cfg.n_obs <= NOBS at run time).
// =============================================================================================================
// GEN = general-gamma discrete-CBF rows (kin.py:245-248 with 0 < gamma < 1): row i is solved as the STATE constraint of node i
//   c_i(X_i) = h(X_i + T f(X_i)) - (1 - gamma) h(X_i) >= gamma hmin      (stage-i obstacle in both terms, as the reference writes it)
// which equals the reference's  gamma h(X_i) + h(X_{i+1}) - h(X_i)  on the feasible set because the position part of the Euler
// step depends on X_i only.  Its gradient has four entries (x, y, phi, v) and its Hessian fills the state block.
// RESTO = the instantiation that runs the restoration pass (a.pass == 1): main phase + restoration phase with a run-time phase
// flag, per-node cost table, elastic obstacle rows.  The RESTO = false instantiation is the lean main phase of the first pass.
// RK4 = the shooting rows use the Runge-Kutta step (cfg.integrator = MPCB_INT_RK4; keep-out / gamma = 1 rows only, never with GEN): the
//   stage's B block becomes dense (the control enters x+, y+ and phi+ through both columns), four more Hessian pairs exist.
template <int NOBS, bool GEN = false, bool RESTO = false, bool RK4 = false>
MPCB_DEVFN void mpcb_solve_kin(const MpcbKArgs& a_in, const int b, double* lds, const int pass) {   // pass: MPCB_PASS_* (see MpcbKArgs::pass; a parameter of its own because one launch can run two passes of an instance)
  // every kernel argument is read through a pointer the optimiser cannot see through (wv::late_args): the compiler then loads a field
  // where the code needs it instead of preloading the whole 800-byte argument block into scalar registers at entry, most of which it
  // has to spill into VGPR lanes again (kin<3>: 806 -> 582 v_readlane of SGPR reloads)
  const MpcbKArgs& a = *wv::late_args(a_in);
  static_assert(!(GEN && RK4), "general-gamma CBF rows are written for the Euler step");
  using namespace mpcbk;
  constexpr int NX = 4, NA = 6, NW = 8, NOB = NOBS > 0 ? NOBS : 1, NEL = RESTO ? NOB : 1;
  const mpcb_config& c = a.cfg;
  // (the lane index opaque: nothing derived from it is hoisted out of the attempt loop of the fusing kernels and kept alive across attempts)
  const int N = c.N, lane = wv::opaque(wv::lane()), k = lane;
  const int nz = a.nz, nobs = c.n_obs;
  if (RESTO && a.status[(size_t)b * a.st_stride] != MPCB_ST_NEEDS_RESTO) return;     // wave-uniform: this instance is done
  if (!RESTO && pass == MPCB_PASS_SECOND) {          // second start: only instances whose first attempt (restoration included) did not succeed
    const int st1 = a.status[(size_t)b * a.st_stride];
    if (st1 == MPCB_ST_SOLVED || st1 == MPCB_ST_ACCEPTABLE || st1 == MPCB_ST_INFEASIBLE_X0) return;
    bool fin = true;                                    // non-finite inputs: the first attempt's verdict (at iteration 0) stands
    for (int i = 0; i < 4; ++i) fin = fin && isfinite(a.x0[(size_t)b * 4 + i]) && isfinite(a.xs[(size_t)b * 4 + i]);
    if (!fin) return;
  }
  // Which start does this solve run from?  First attempt: the caller's z0, with X rolled out from x0 (cfg.init_rollout).  Second
  // attempt (cfg.second_start, only after a roll-out start; mpcb_api.hip launches its passes after the first attempt's): the
  // reference's own first-step start z = 0 (main_cbf_kin_c_sim.py:47-50), no roll-out.  A restoration pass continues whichever
  // attempt handed over (WK_START).
  const bool zeros_start = RESTO ? (a.work && a.work[(size_t)b * mpcbk::WK_SIZE + mpcbk::WK_START] != 0.0) : pass == MPCB_PASS_SECOND;
  const bool rollout = c.init_rollout && !zeros_start;
  constexpr bool OBL = NOBS > 3;                  // obstacle constants in LDS ([4 * j + q][lane]) instead of registers
  const Layout L = layout_kin(N, nz, RESTO, obs_in_lds(NOBS), GEN || RK4);
  const int ld = L.ld;
  double* ent = lds + L.ent;
  // step length of this lane's stage: cfg.T, or the stage's entry of the time grid (lanes past the last stage take its value)
  double T_ = wv::uni(c.T);              // (a VALUE before the branch: from a ?: of two loads the compiler selects between a global and a constant-space ADDRESS and loads through FLAT)
  if (a.tgrid) T_ = a.tgrid[k < N ? k : N - 1];
  const double T = T_, il = 1.0 / c.veh_l;

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
  else if (GEN) { obs_node = k <= N - 1; ostep = k; }
  else { obs_node = k >= 1 && k - 1 <= last_row; ostep = k - 1; }
  obs_node = obs_node && isnode;
  double ox_[OBL ? 1 : NOB], oy_[OBL ? 1 : NOB], ix2_[OBL ? 1 : NOB], iy2_[OBL ? 1 : NOB];
  double* obl = lds + L.obl;
  auto OC = [&](int j, int q) -> double& {          // constant q (0 centre x, 1 centre y, 2 1/sX^2, 3 1/sY^2) of obstacle slot j
    if constexpr (OBL) return obl[(4 * j + q) * (N + 2) + (lane <= N ? lane : N + 1)];   // rows of N+2: lanes beyond the last node share a dummy column
    else return q == 0 ? ox_[j] : q == 1 ? oy_[j] : q == 2 ? ix2_[j] : iy2_[j];
  };
#define ox(j) OC(j, 0)
#define oy(j) OC(j, 1)
#define ix2(j) OC(j, 2)
#define iy2(j) OC(j, 3)
#pragma unroll
  for (int j = 0; j < NOBS; ++j) {
    ox(j) = 0; oy(j) = 0; ix2(j) = 0; iy2(j) = 0;
    if (j < nobs && obs_node) {
      const double* q = (a.obs_kind == MPCB_OBSIN_PREDICTED)
                            ? a.obs + (((size_t)b * nobs + j) * (N + 1) + ostep) * 6
                            : a.obs + ((size_t)b * nobs + j) * 6;
      double sx = c.obs_sx_fixed > 0 ? c.obs_sx_fixed : c.ego_hl + q[4] / 2 + c.safe_disl;
      double sy = c.obs_sy_fixed > 0 ? c.obs_sy_fixed : c.ego_hw + q[5] / 2 + c.safe_disw;
      ox(j) = q[0]; oy(j) = q[1]; ix2(j) = 1.0 / (sx * sx); iy2(j) = 1.0 / (sy * sy);
    }
  }
  auto hval = [&](int j, double px, double py) {
    double dx = px - ox(j), dy = py - oy(j);
    return dx * dx * ix2(j) + dy * dy * iy2(j) - 1.0;
  };
  const double omg = GEN ? 1.0 - c.gamma : 0.0;                     // (1 - gamma) of the general CBF row
  // value of obstacle row j at a node with position (px,py), heading sin/cos (s_,c_) and speed v
  auto rowval = [&](int j, double px, double py, double s_, double c_, double v) {
    if (!GEN) return hval(j, px, py);
    return hval(j, px + T * (v * c_), py + T * (v * s_)) - omg * hval(j, px, py);
  };

  // ----- start point: z0 row (coalesced) -> LDS -> node lanes ---------------------------------------------------
  double* zbuf = lds + L.zbuf;
  for (int i = lane; i < nz; i += 64) zbuf[i] = (a.z0 && !zeros_start) ? a.z0[(size_t)b * nz + i] : 0.0;
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
    g = wv::uni(wv::max(g));
    os = wv::uni((g > c.max_gradient) ? c.max_gradient / g : 1.0);
  }
  if (RESTO) {     // restoration pass: the iterate is what the first pass left in z (the scaling above is that of the user's start)
    wv::sync();
    for (int i = lane; i < nz; i += 64) zbuf[i] = a.z[(size_t)b * nz + i];
    wv::sync();
#pragma unroll
    for (int i = 0; i < NU; ++i) U[i] = hasu ? zbuf[NU * k + i] : 0.0;
#pragma unroll
    for (int i = 0; i < NX; ++i) X[i] = isnode ? zbuf[NU * N + NX * k + i] : 0.0;
    wv::sync();
  }
  double* cst = lds + L.cst;
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NX; ++i) { cst[CS_WQ + i] = os * 2 * c.Q[i]; cst[CS_Q + i] = c.Q[i]; cst[CS_XS + i] = xs[i]; }
#pragma unroll
    for (int i = 0; i < NU; ++i) { cst[CS_WR + i] = os * 2 * c.R[i]; cst[CS_WDR + i] = os * 2 * c.DR[i]; cst[CS_R + i] = c.R[i]; cst[CS_DR + i] = c.DR[i]; cst[CS_UL + i] = c.u_last[i]; }
    cst[CS_ACC] = 1e300; cst[CS_ACC + 1] = 0.0;
  }
  wv::sync();
  // The objective of the running phase.  First-pass instantiation: the uniform constants of the block above.  RESTO instantiation:
  // a per-node table (lane = node, every lane reads and writes only its own column), because the restoration phase replaces
  // the cost by the proximity term zeta/2 ||D_R (w - w_R)||^2 with per-node weights and reference (oracle: set_resto_cost).
  double* ct = lds + L.ct;
  const int cts = N + 2, ctl = lane <= N ? lane : N + 1;     // row length of the cost table; lanes beyond the last node share a dummy column
  double osc = os;                       // scale of the running phase's objective: os, or 1 in the restoration phase
  bool rs = false;                       // restoration phase active
  auto cWQ = [&](int i) { return RESTO ? ct[(CT_WQ + i) * cts + ctl] : cst[CS_WQ + i]; };
  auto cXS = [&](int i) { return RESTO ? ct[(CT_XR + i) * cts + ctl] : cst[CS_XS + i]; };
  auto cQQ = [&](int i) { return RESTO ? ct[(CT_QQ + i) * cts + ctl] : cst[CS_Q + i]; };
  auto cWR = [&](int i) { return RESTO ? ct[(CT_WR + i) * cts + ctl] : cst[CS_WR + i]; };
  auto cRR = [&](int i) { return RESTO ? ct[(CT_RR + i) * cts + ctl] : cst[CS_R + i]; };
  auto cUR = [&](int i) { return RESTO ? ct[(CT_UR + i) * cts + ctl] : 0.0; };
  auto cWDR = [&](int i) { return RESTO ? ct[(CT_WDR + i) * cts + ctl] : cst[CS_WDR + i]; };
  auto cDRR = [&](int i) { return RESTO ? ct[(CT_DRR + i) * cts + ctl] : cst[CS_DR + i]; };
  auto write_main_cost = [&]() {         // kin.py:168-205: Q on nodes 0..N-1 (no terminal cost), R, DR; set-point xs, U reference 0
    if (RESTO) {
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        ct[(CT_WQ + i) * cts + ctl] = hasu ? os * 2 * c.Q[i] : 0.0; ct[(CT_QQ + i) * cts + ctl] = hasu ? c.Q[i] : 0.0;
        ct[(CT_XR + i) * cts + ctl] = xs[i];
      }
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        ct[(CT_WR + i) * cts + ctl] = hasu ? os * 2 * c.R[i] : 0.0; ct[(CT_RR + i) * cts + ctl] = hasu ? c.R[i] : 0.0;
        ct[(CT_UR + i) * cts + ctl] = 0.0;
        ct[(CT_WDR + i) * cts + ctl] = os * 2 * c.DR[i]; ct[(CT_DRR + i) * cts + ctl] = c.DR[i];
      }
    }
  };
  // zeta/2 * D^2 (w - w_R)^2 on X_1..X_N and U_0..U_{N-1}, D = 1 / max(1, |w_R|); `fresh` also latches w_R = current (X, U)
  auto write_resto_cost = [&](double zeta, bool fresh, const double* Xc, const double* Uc) {
    if (RESTO) {
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        if (fresh) ct[(CT_XR + i) * cts + ctl] = Xc[i];
        const double d = 1.0 / fmax(1.0, fabs(ct[(CT_XR + i) * cts + ctl]));
        const double q = xnode ? 0.5 * zeta * d * d : 0.0;
        ct[(CT_QQ + i) * cts + ctl] = q; ct[(CT_WQ + i) * cts + ctl] = 2 * q;
      }
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        if (fresh) ct[(CT_UR + i) * cts + ctl] = Uc[i];
        const double d = 1.0 / fmax(1.0, fabs(ct[(CT_UR + i) * cts + ctl]));
        const double q = hasu ? 0.5 * zeta * d * d : 0.0;
        ct[(CT_RR + i) * cts + ctl] = q; ct[(CT_WR + i) * cts + ctl] = 2 * q;
        ct[(CT_WDR + i) * cts + ctl] = 0.0; ct[(CT_DRR + i) * cts + ctl] = 0.0;
      }
    }
  };
  write_main_cost();
  // masks of the cost terms: the table of the RESTO instantiation holds zeros where a term does not exist in the running phase
  const bool xq = RESTO ? xnode : xcost;                 // stage cost on X_k in the Hessian / gradient
  const bool xobj = RESTO ? isnode : hasu;               // ... in the objective value (node 0: a constant)
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
        for (int j = 0; j < NOBS; ++j) if (j < nobs) {
          double s0 = 0, c0 = 1;
          if (GEN) sincos_b(X[2], s0, c0);
          bad = bad || (rowval(j, X[0], X[1], s0, c0, X[3]) < (GEN ? c.gamma : 1.0) * c.obs_hmin - 1e-8);
        }
      }
    }
    if (wv::any(bad)) status = MPCB_ST_INFEASIBLE_X0;
  }

  // bounds (uniform)
  const Bnd qU0 = mk_bnd(c.u_lo[0], c.u_hi[0], c.bound_relax), qU1 = mk_bnd(c.u_lo[1], c.u_hi[1], c.bound_relax);
  const Bnd qY = mk_bnd(c.x_lo[1], c.x_hi[1], c.bound_relax), qV = mk_bnd(c.x_lo[3], c.x_hi[3], c.bound_relax);
  // rate row k compares U_k - U_{k-1} with rate * (time between the two controls): cfg.du_* are rate * cfg.T (kin.py:116-121),
  // with a time grid the bounds of lane k scale by T_{k-1} / cfg.T (1 exactly without a grid)
  const double rsc = a.tgrid ? wv::shfl(T, k - 1) / c.T : 1.0;
  const Bnd qR = mk_bnd_lane(c.du_lo[0] * rsc, c.du_hi[0] * rsc, c.bound_relax);
  const Bnd qO = mk_bnd((GEN ? c.gamma : 1.0) * c.obs_hmin, 1e308, c.bound_relax);
  const bool bu0_on = hasu && qU0.on, bu1_on = hasu && qU1.on, by_on = xnode && qY.on, bv_on = xnode && qV.on;
  const bool rr_on = xcost && qR.on;
  const bool ro_node = xnode && obs_node;

  // model quantities at (X, U): sin/cos of the heading, tan and sec^2 of the steering angle
  double sp, cp, td, sec2;
  auto trig = [&](double phi, double df, double& s_, double& c_, double& t_, double& e_) {
    sincos_b(phi, s_, c_);
    double sd, cd; sincos_b(df, sd, cd);
    const double icd = wv::rcp(cd);
    t_ = sd * icd; e_ = icd * icd;
  };

  // optional roll-out of X from x0 with the guessed (clipped) controls
  if (!RESTO && rollout) {
    auto roll_out = [&]() {
      U[0] = hasu ? fmin(fmax(U[0], c.u_lo[0]), c.u_hi[0]) : 0.0;
      U[1] = hasu ? fmin(fmax(U[1], c.u_lo[1]), c.u_hi[1]) : 0.0;
      double sd, cd; sincos_b(U[0], sd, cd);
      const double tdr = sd * wv::rcp(cd);
#pragma clang loop unroll(disable)
      for (int s = 0; s < N; ++s) {
        double sps, cps; sincos_b(X[2], sps, cps);
        const double v = X[3];
        double F0 = X[0] + T * (v * cps), F1 = X[1] + T * (v * sps), F2 = X[2] + T * (v * tdr * il), F3 = X[3] + T * U[1];
        if (RK4) { double Fr[NX]; kin_rk4_step(X, U, T, il, sps, cps, tdr, Fr); F0 = Fr[0]; F1 = Fr[1]; F2 = Fr[2]; F3 = Fr[3]; }
        const double n0 = wv::bcast(F0, s), n1 = wv::bcast(F1, s), n2 = wv::bcast(F2, s), n3 = wv::bcast(F3, s);
        if (k == s + 1) { X[0] = n0; X[1] = n1; X[2] = n2; X[3] = n3; }
      }
    };
    roll_out();
    // cfg.start_steer (include/mpcbatch.h; oracle: Solver::init): a cold start whose straight roll-out passes an obstacle row closer than
    // h - obs_hmin < 1 is rolled out with a slight constant turn instead — away from the centre of that obstacle (the first minimum of
    // h over the nodes, then over the obstacles of a node), or to its other side when the y box has no room for the row's ellipse there
    if (NOBS > 0 && !a.z0 && nobs > 0 && wv::late_args(a)->cfg.start_steer > 0.0) {       // wave-uniform
      double hm = 1e300, oyk = 0.0, iyk = 1.0;
#pragma unroll
      for (int j = 0; j < NOBS; ++j) if (j < nobs) {
        const double hj = hval(j, X[0], X[1]) - c.obs_hmin;
        if (hj < hm) { hm = hj; oyk = oy(j); iyk = iy2(j); }
      }
      if (!(obs_node && k >= 1)) hm = 1e300;
      const double hw = wv::uni(wv::min(hm));
      if (hw < 1.0) {
        const int kb = (int)wv::uni(wv::min(hm == hw ? (double)k : 1e9));      // the first node at which the minimum is taken
        const double py = wv::shfl(X[1], kb), qy = wv::shfl(oyk, kb), sy = 1.0 / sqrt(wv::shfl(iyk, kb));
        double sgn = py >= qy ? 1.0 : -1.0;
        const bool up = qy + sy <= c.x_hi[1], dn = qy - sy >= c.x_lo[1];
        if (sgn > 0 && !up && dn) sgn = -1.0; else if (sgn < 0 && !dn && up) sgn = 1.0;
        U[0] = hasu ? sgn * wv::late_args(a)->cfg.start_steer : 0.0;
        roll_out();
      }
    }
  }

  // push the start inside the (relaxed) boxes; duals = 1
  Item iU0{1, 1, 0, 0}, iU1{1, 1, 0, 0}, iY{1, 1, 0, 0}, iV{1, 1, 0, 0}, iR{1, 1, 0, 0};
  if (!RESTO) {    // (the restoration pass starts from an interior iterate of the first pass)
    if (bu0_on) U[0] = push_in(qU0, U[0], c.bound_push, c.bound_frac);
    if (bu1_on) U[1] = push_in(qU1, U[1], c.bound_push, c.bound_frac);
    if (by_on) X[1] = push_in(qY, X[1], c.bound_push, c.bound_frac);
    if (bv_on) X[3] = push_in(qV, X[3], c.bound_push, c.bound_frac);
  }
  // general rows: slack = row value at the pushed start, pushed inside its own bounds
  double Up0 = wv::shfl(U[0], k - 1), Up1 = wv::shfl(U[1], k - 1);   // U_{k-1}
  double sR = 0, rR = 0;
  if (rr_on) sR = push_in(qR, U[0] - Up0, c.bound_push, c.bound_frac);
  double sO[NOB], vO[NOB], iO[NOB], rO[NOB], gO0[NOB], gO1[NOB], gO2[GEN ? NOB : 1], gO3[GEN ? NOB : 1];
  bool ro_on[NOB];
  // restoration phase: elastic variables of the obstacle rows  c(w) - s - p + n = 0  and their duals (zero outside it)
  double eP[NEL], eN[NEL], vP[NEL], vN[NEL];
#pragma unroll
  for (int j = 0; j < NEL; ++j) { eP[j] = 0; eN[j] = 0; vP[j] = 0; vN[j] = 0; }
  {
    double s0 = 0, c0 = 1;
    if (GEN) sincos_b(X[2], s0, c0);
#pragma unroll
    for (int j = 0; j < NOBS; ++j) {
      ro_on[j] = ro_node && j < nobs;
      sO[j] = ro_on[j] ? push_in(qO, rowval(j, X[0], X[1], s0, c0, X[3]), c.bound_push, c.bound_frac) : 1.0;
      vO[j] = 1.0; rO[j] = 0; gO0[j] = 0; gO1[j] = 0; iO[j] = 0;
      if (GEN) { gO2[j] = 0; gO3[j] = 0; }
    }
  }
  // gradient of the obstacle rows at the iterate (X, sp, cp)
  auto row_grads = [&](double sp_, double cp_) {
#pragma unroll
    for (int j = 0; j < NOBS; ++j) if (ro_on[j]) {
      const double dpx = 2 * (X[0] - ox(j)) * ix2(j), dpy = 2 * (X[1] - oy(j)) * iy2(j);
      if (!GEN) { gO0[j] = dpx; gO1[j] = dpy; }
      else {
        const double v = X[3];
        const double dqx = 2 * (X[0] + T * (v * cp_) - ox(j)) * ix2(j), dqy = 2 * (X[1] + T * (v * sp_) - oy(j)) * iy2(j);
        gO0[j] = dqx - omg * dpx; gO1[j] = dqy - omg * dpy;
        gO2[j] = T * v * (dqy * cp_ - dqx * sp_); gO3[j] = T * (dqx * cp_ + dqy * sp_);
      }
    }
  };
  auto recips = [&]() {
    if (bu0_on) item_recip(qU0, U[0], iU0);
    if (bu1_on) item_recip(qU1, U[1], iU1);
    if (by_on) item_recip(qY, X[1], iY);
    if (bv_on) item_recip(qV, X[3], iV);
    if (rr_on) item_recip(qR, sR, iR);
#pragma unroll
    for (int j = 0; j < NOBS; ++j) if (ro_on[j]) iO[j] = wv::rcp(sO[j] - qO.L);
  };

  double mu = c.mu_init, tau = fmax(TAU_MIN, 1.0 - mu);
  double dfc[NX] = {0, 0, 0, 0};
  double theta = 0, fval = 0, logsum = 0;   // sum |constraint residual|, unscaled objective, sum of log(distance to bound)

  // per-lane pieces of an evaluation at (Xa, Ua): defects, row residuals, theta/f/log partial sums.
  // Returns false in a lane whose slack or box distance is not positive.
  auto eval_lane = [&](const double* Xa, const double* Ua, double sRa, const double* sOa, const double* pa, const double* na,
                       double s_, double c_, double t_,
                       double* dfa, double& rRa, double* rOa, double& up0, double& up1, double& th, double& fl, double& prod) {
    bool ok = true;
    const double v = Xa[3];
    double Ft[NX] = {Xa[0] + T * (v * c_), Xa[1] + T * (v * s_), Xa[2] + T * (v * t_ * il), Xa[3] + T * Ua[1]};
    if (RK4) kin_rk4_step(Xa, Ua, T, il, s_, c_, t_, Ft);
    th = 0; fl = 0; prod = 1.0;
#pragma unroll
    for (int i = 0; i < NX; ++i) { const double xn = wv::shfl(Xa[i], k + 1); dfa[i] = hasu ? Ft[i] - xn : 0.0; th += fabs(dfa[i]); }
    up0 = wv::shfl(Ua[0], k - 1); up1 = wv::shfl(Ua[1], k - 1);
    auto bar = [&](const Bnd& q, double s) {
      if (q.hasL) { const double d = s - q.L; ok = ok && (d > 0); prod *= d; }
      if (q.hasU) { const double d = q.U - s; ok = ok && (d > 0); prod *= d; }
    };
    if (bu0_on) bar(qU0, Ua[0]);
    if (bu1_on) bar(qU1, Ua[1]);
    if (by_on) bar(qY, Xa[1]);
    if (bv_on) bar(qV, Xa[3]);
    rRa = 0;
    if (rr_on) { bar(qR, sRa); rRa = (Ua[0] - up0) - sRa; th += fabs(rRa); }
#pragma unroll
    for (int j = 0; j < NOBS; ++j) {
      rOa[j] = 0;
      if (ro_on[j]) {
        bar(qO, sOa[j]); rOa[j] = rowval(j, Xa[0], Xa[1], s_, c_, v) - sOa[j];
        if (RESTO && rs) {        // elastic row: residual of c - s - p + n, cost rho (p + n), barrier on p and n
          const double pj = pa[j], nj = na[j];
          ok = ok && (pj > 0) && (nj > 0); prod *= pj * nj;
          rOa[j] -= pj - nj; fl += RS_RHO * (pj + nj);
        }
        th += fabs(rOa[j]);
      }
    }
    if (xobj) {   // objective terms of node k (kin.py:195-205; restoration phase: proximity term)
#pragma unroll
      for (int i = 0; i < NX; ++i) { const double e = Xa[i] - cXS(i); fl += cQQ(i) * e * e; }
    }
    if (hasu) {
      const double e0 = Ua[0] - cUR(0), e1 = Ua[1] - cUR(1);
      fl += cRR(0) * e0 * e0 + cRR(1) * e1 * e1;
      if (k > 0 || c.du0_cost) {
        const double d0 = Ua[0] - (k ? up0 : cst[CS_UL]), d1 = Ua[1] - (k ? up1 : cst[CS_UL + 1]);
        fl += cDRR(0) * d0 * d0 + cDRR(1) * d1 * d1;
      }
    }
    return ok;
  };

  // number of multipliers (constants of the problem)
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
    n_v = wv::uni(wv::sum(cnt));
    n_lam = (double)(NX * N);
  }

  // ----- per-lane constants of the Riccati sweep: lane = entry (i, j) of the 8x8 stage block -------------------
#ifdef MPCB_EXP_OPAQUE_SWEEP     // diagnostic build (tools/sync_ab.sh): the round-2 incident "asm barrier on the lane index in the sweep preamble"
  const int lz = wv::opaque(lane);
  const int ei = lz >> 3, ej = lz & 7;
#else
  const int ei = lane >> 3, ej = lane & 7;
#endif
  auto slotAB = [&](int r, int col) -> int {   // slot of entry [A B | aug](r, col), r < NA, col < NW, inside the stage's fw row
    if (r < NX) {
      if (col < NX) {
        if (r == col) return FW_ONE;
        if (r == 0 && col == 2) return 2;
        if (r == 0 && col == 3) return 3;
        if (r == 1 && col == 2) return FWR + 2;
        if (r == 1 && col == 3) return FWR + 3;
        if (r == 2 && col == 3) return 2 * FWR + 3;
        return FW_ZERO;
      }
      if (RK4) {       // dense B: the two control coefficients of a state row sit in slots 4, 5 of its record (its U_prev coefficients, which are zero, are not stored)
        if (col == 6) return r == 3 ? FW_ZERO : r * FWR + 4;
        if (col == 7) return r * FWR + 5;
        return FW_ZERO;
      }
      if (r == 2 && col == 6) return 2 * FWR + FW_BX;
      if (r == 3 && col == 7) return 3 * FWR + FW_BX;
      return FW_ZERO;
    }
    return (col == r + 2) ? FW_ONE : FW_ZERO;    // Uprev_{k+1} = U_k
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
    if (RK4) {
      if (lo == 2 && hi == 6) return E_HPD;
      if (lo == 2 && hi == 7) return E_HPA;
      if (lo == 3 && hi == 7) return E_HVA;
      if (lo == 6 && hi == 7) return E_HDA;
    }
    if (GEN) {
      if (lo == 0 && hi == 2) return E_HXP;
      if (lo == 0 && hi == 3) return E_HXV;
      if (lo == 1 && hi == 2) return E_HYP;
      if (lo == 1 && hi == 3) return E_HYV;
    }
    return E_ZERO;
  };
  // Column 4 of the stage block belongs to U_prev, whose column of [A B] is zero: W(:,4) = 0 and M(:,4) = H(:,4) need no
  // arithmetic.  The lanes of that column therefore carry the AFFINE part of the recursion through the same instruction
  // stream: their "[A B] column" is the defect d, their accumulation starts from p+ (resp. g), so W(:,4) := q and their
  // M accumulator := m.  No lane computes q or m redundantly.
  // [A B] = [I + sparse | 0 | sparse; 0 | 0 | I]: rows 4,5 (U_prev+ = U) hold a single 1, so every column / row of the
  // products needs the 4 coefficients of the state rows plus at most ONE unit term read straight from P+ (resp. W):
  //   W(i,j) = sum_{r<4} P+(i,r) AB(r,j) + [j=6] P+(i,4) + [j=7] P+(i,5)          (+ p+_i in the affine column)
  //   M(i,j) = H(i,j) + sum_{r<4} AB(r,i) W(r,j) + [i=6] W(4,j) + [i=7] W(5,j)
  const bool aff = (ej == 4);
  int sCW[NX], sCM[NX];
#pragma unroll
  for (int r = 0; r < NX; ++r) {
    sCW[r] = aff ? r * FWR + FW_C0 : slotAB(r, ej);      // (slots of the fw row, not of the entry table)
    sCM[r] = slotAB(r, ei);
  }
  const int sHij = slotH(ei, ej) * ld, sGi = (E_G0 + ei) * ld;
  const int sStart = aff ? sGi : sHij;         // start value of the M accumulation: g_i in the affine lanes, H_ij elsewhere
  const int eiA = ei < NA ? ei : 0;            // clamped row for lanes of the control rows (their W is unused)
  // unit term of W inside the Pst row of stage s+1: p+_i (affine lanes), P+(i,4) / P+(i,5) (control columns), else the 0.0 slot
  const int uWOff = ei >= NA ? PS_ZERO : aff ? PS_P + ei : ej == 6 ? ei * NA + 4 : ej == 7 ? ei * NA + 5 : PS_ZERO;
  // unit term of M inside W^T: W(4,j) / W(5,j) in the control rows, else the 0.0 slot
  const int uMOff = ei == 6 ? ej * NA + 4 : ei == 7 ? ej * NA + 5 : W_ZERO;
  // store targets; lanes without an entry write into pad slots, so that the sweep has no divergent branches
  const int wOff = ei < NA ? ej * NA + ei : NW * NA + (lane & 15);
  const bool upper = ei < NA && ej < NA && ei <= ej;
  const int pOff = upper ? ei * NA + ej : PS_PAD + (lane & 1), pOffT = upper ? ej * NA + ei : PS_PAD + 2 + (lane & 1);
  const int psOff = (aff && ei < NA) ? PS_P + ei : PS_PADP;
  // gains: lanes (0,j) store K0j, lanes (1,j) store K1j (j < 6); lanes 62 / 63 store the feed-forward terms
  const int kOff = (ej < NA && ei == 0) ? NX * FWR + ej : (ej < NA && ei == 1) ? (NX + 1) * FWR + ej : FW_PAD + (lane & 1);
  const int kfOff = (lane == 62) ? NX * FWR + FW_C0 : (lane == 63) ? (NX + 1) * FWR + FW_C0 : FW_PAD + (lane & 1);
  const bool kRow1 = (ei == 1), kfLane1 = (lane == 63);

  // constant rows of the entry table
  if (isnode) { ent[E_ZERO * ld + k] = 0.0; ent[E_ONE * ld + k] = 1.0; }
  // (Pst slot PS_ZERO = 0.0 is written after the z0 staging below has finished with the aliased region)

  double* Pst = lds + L.Pst; double* fw = lds + L.fw;
  double* Wl = (double*)__builtin_assume_aligned(lds + L.W, 16); double* filt = lds + L.filt;
  double* Sl = (double*)__builtin_assume_aligned(Wl + W_STAGE, 16);
  if (lane == 0) Wl[W_ZERO] = 0.0;
  if (isnode) {            // unit and zero entries of the records (never written again)
    double* fk = fw + k * FWS;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
#pragma unroll
      for (int r = 0; r < FWR; ++r) {
        const bool var = i >= NX ? r <= FW_C0 : (r == FW_C0 || (i == 0 && (r == 2 || r == 3)) || (i == 1 && (r == 2 || r == 3)) || (i == 2 && (r == 3 || r == FW_BX)) || (i == 3 && r == FW_BX) ||
                                                 (RK4 && (r == 4 || r == 5)));
        if (!var) fk[i * FWR + r] = (i < NX && r == i) ? 1.0 : 0.0;
      }
    }
  }
  int nfilt = 0;
  double theta_max = 0, theta_min = 0;
  double dw_last = 0.0;
  const double mu_floor = c.tol / (K_EPS + 1.0);
  double err0 = 0, e_dual = 0, e_prim = 0;

  // restoration-phase state (RESTO instantiation): the main phase's mu and filter bounds while the restoration runs, the entry
  // pair of the main filter, the violation at entry, iterations inside the phase; `enter` asks the loop top to start the phase
  double mu_main = 0, tmax_main = 0, tmin_main = 0, fm_theta = 0, fm_phi = 0, th_entry = 0;
  int rit = 0, slow_run = 0, n_rcalls = 0, n_riters = 0;
  double slow_theta0 = 0;
  bool enter = false;
  double n_el = 0;                       // number of elastic rows (constant of the instance)
  if (RESTO) {
    double cnt = 0;
#pragma unroll
    for (int j = 0; j < NOBS; ++j) if (ro_on[j]) cnt += 1.0;
    n_el = wv::uni(wv::sum(cnt));
  }

  if (status != MPCB_ST_INFEASIBLE_X0) {
    if (!RESTO) {
      // evaluation at the start point
      trig(X[2], U[0], sp, cp, td, sec2);
      double th, fl, prod;
      eval_lane(X, U, sR, sO, eP, eN, sp, cp, td, dfc, rR, rO, Up0, Up1, th, fl, prod);
      double sv[3] = {th, fl, log(prod)};
      wv::reduce<3, 0>(sv, nullptr);
      theta = wv::uni(sv[0]); fval = wv::uni(sv[1]); logsum = wv::uni(sv[2]);
      recips();
      row_grads(sp, cp);
      theta_max = wv::uni(1e4 * fmax(1.0, theta)); theta_min = wv::uni(1e-4 * fmax(1.0, theta));
    } else {
      // restoration pass: the iterate is what the first pass left in z, the scalars come from its hand-over record
      const double* wk = a.work + (size_t)b * WK_SIZE;
      mu = wv::uni(wk[WK_MU]); tau = fmax(TAU_MIN, 1.0 - mu);
      theta_max = wv::uni(wk[WK_THMAX]); theta_min = wv::uni(wk[WK_THMIN]);
      iters = (int)wk[WK_ITERS]; dw_last = wv::uni(wk[WK_DW]);
      enter = true;
    }

    int trips = 0;                       // passes through the loop top, counted or not: the exit every wave reaches
#pragma clang loop unroll(disable)
    for (;;) {
      if (++trips > 3 * c.max_iter + 50) { status = MPCB_ST_RESTO_FAILED; break; }   // (phase changes do not count as iterations)
      if (RESTO && enter && n_rcalls >= RS_MAX_CALLS) { status = MPCB_ST_RESTO_FAILED; break; }
      if (RESTO && enter) {
        ++n_rcalls;
        // ----- entry into the restoration phase (oracle: Solver::restoration) ------------------------------------------------
        enter = false;
        if (lane == 0) { cst[CS_ACC] = 1e300; cst[CS_ACC + 1] = 0.0; }      // (the acceptable-point counter starts afresh after a restoration, in both passes alike)
        wv::sync();
        mu_main = mu; tmax_main = theta_max; tmin_main = theta_min;
        rs = false; osc = os;
        // slacks of the general rows re-initialised from w as at a fresh start; the entry pair of the main filter is taken there
        trig(X[2], U[0], sp, cp, td, sec2);
        Up0 = wv::shfl(U[0], k - 1); Up1 = wv::shfl(U[1], k - 1);
        if (rr_on) sR = push_in(qR, U[0] - Up0, c.bound_push, c.bound_frac);
#pragma unroll
        for (int j = 0; j < NOBS; ++j) if (ro_on[j]) sO[j] = push_in(qO, rowval(j, X[0], X[1], sp, cp, X[3]), c.bound_push, c.bound_frac);
        double th, fl, prod;
        eval_lane(X, U, sR, sO, eP, eN, sp, cp, td, dfc, rR, rO, Up0, Up1, th, fl, prod);
        double vi = fabs(rR);
#pragma unroll
        for (int j = 0; j < NOBS; ++j) vi = fmax(vi, fabs(rO[j]));
        if (hasu) {
#pragma unroll
          for (int i = 0; i < NX; ++i) vi = fmax(vi, fabs(dfc[i]));
        }
        double sv[3] = {th, fl, log(prod)}, mv[1] = {vi};
        wv::reduce<3, 1>(sv, mv);
        th_entry = wv::uni(sv[0]);
        const double phi_entry = os * wv::uni(sv[1]) - mu_main * wv::uni(sv[2]);
        fm_theta = (1 - G_THETA) * th_entry; fm_phi = phi_entry - G_PHI * th_entry;
        // the restoration problem: proximity cost around the entry point, elastic obstacle rows, centred duals capped at rho
        rs = true; osc = 1.0;
        mu = wv::uni(fmax(mu_main, wv::uni(mv[0]))); tau = fmax(TAU_MIN, 1.0 - mu);
        write_resto_cost(sqrt(mu), true, X, U);
#pragma unroll
        for (int j = 0; j < NOBS; ++j) if (ro_on[j]) {
          const double r0 = rO[j], aa = (mu - RS_RHO * r0) / (2 * RS_RHO);
          eN[j] = aa + sqrt(aa * aa + mu * r0 / (2 * RS_RHO)); eP[j] = r0 + eN[j];
          vP[j] = mu / eP[j]; vN[j] = mu / eN[j];
        }
        recips();
        auto centre = [&](const Bnd& q, Item& it) {
          if (q.hasL) it.vL = fmin(RS_RHO, mu * it.iL);
          if (q.hasU) it.vU = fmin(RS_RHO, mu * it.iU);
        };
        if (bu0_on) centre(qU0, iU0);
        if (bu1_on) centre(qU1, iU1);
        if (by_on) centre(qY, iY);
        if (bv_on) centre(qV, iV);
        if (rr_on) centre(qR, iR);
#pragma unroll
        for (int j = 0; j < NOBS; ++j) if (ro_on[j]) vO[j] = fmin(RS_RHO, mu * iO[j]);
#pragma unroll
        for (int i = 0; i < NX; ++i) lam[i] = 0.0;
        nfilt = 0;
        eval_lane(X, U, sR, sO, eP, eN, sp, cp, td, dfc, rR, rO, Up0, Up1, th, fl, prod);
        double sw[3] = {th, fl, log(prod)};
        wv::reduce<3, 0>(sw, nullptr);
        theta = wv::uni(sw[0]); fval = wv::uni(sw[1]); logsum = wv::uni(sw[2]);
        theta_max = wv::uni(1e4 * fmax(1.0, theta)); theta_min = wv::uni(1e-4 * fmax(1.0, theta));
        row_grads(sp, cp);
        rit = 0; slow_run = 0;
      }
      MPCB_STAMP(t_a);
      // ----- KKT residuals of the scaled problem at the iterate (one pass, fused reductions) ---------------------
      double a02 = -T * X[3] * sp, a03 = T * cp, a12 = T * X[3] * cp, a13 = T * sp, a23 = T * td * il, b20 = T * X[3] * sec2 * il;
      double ln[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) ln[i] = wv::shfl(lam[i], k + 1);          // lam_{k+1}
      KinRkJac Jr{}; KinRkHess Hr{};
      if (RK4) {
        kin_rk4_derivs(X, U, T, il, sp, cp, td, sec2, ln, Jr, Hr);
        a02 = Jr.a02; a03 = Jr.a03; a12 = Jr.a12; a13 = Jr.a13; a23 = Jr.a23; b20 = Jr.b20;
      }
      {
        double rX[NX] = {0, 0, 0, 0}, rU[NU] = {0, 0};
        const double Un0 = wv::shfl(U[0], k + 1), Un1 = wv::shfl(U[1], k + 1);
        const double yR = rr_on ? item_y(qR, iR) : 0.0;
        const double yRn = wv::shfl(yR, k + 1);
        double sum_lam = 0, sum_v = 0, svmax = 0, svmin = 1e300, prim = 0, edual_el = 0, Vel = 0;
        if (xnode) {
          if (RESTO || k < N) {
#pragma unroll
            for (int i = 0; i < NX; ++i) rX[i] += cWQ(i) * (X[i] - cXS(i));
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
          rU[0] += cWR(0) * (U[0] - cUR(0)); rU[1] += cWR(1) * (U[1] - cUR(1));
          if (k > 0 || c.du0_cost) {
            rU[0] += cWDR(0) * (U[0] - (k ? Up0 : cst[CS_UL]));
            rU[1] += cWDR(1) * (U[1] - (k ? Up1 : cst[CS_UL + 1]));
          }
          if (k + 1 < N) { rU[0] -= cWDR(0) * (Un0 - U[0]); rU[1] -= cWDR(1) * (Un1 - U[1]); }
          rU[0] += b20 * ln[2]; rU[1] += T * ln[3];
          if (RK4) { rU[0] += Jr.b00 * ln[0] + Jr.b10 * ln[1]; rU[1] += Jr.b01 * ln[0] + Jr.b11 * ln[1] + Jr.b21 * ln[2]; }
          if (k + 1 < N) rU[0] += yRn;                                          // d(row k+1)/dU_k = -1
        }
        auto item = [&](const Bnd& q, double s, const Item& it) {
          if (q.hasL) { const double p = (s - q.L) * it.vL; svmax = fmax(svmax, p); svmin = fmin(svmin, p); sum_v += it.vL; }
          if (q.hasU) { const double p = (q.U - s) * it.vU; svmax = fmax(svmax, p); svmin = fmin(svmin, p); sum_v += it.vU; }
        };
        if (bu0_on) { rU[0] -= item_y(qU0, iU0); item(qU0, U[0], iU0); }
        if (bu1_on) { rU[1] -= item_y(qU1, iU1); item(qU1, U[1], iU1); }
        if (by_on) { rX[1] -= item_y(qY, iY); item(qY, X[1], iY); }
        if (bv_on) { rX[3] -= item_y(qV, iV); item(qV, X[3], iV); }
        if (rr_on) { rU[0] -= yR; item(qR, sR, iR); prim = fmax(prim, fabs(rR)); }
#pragma unroll
        for (int j = 0; j < NOBS; ++j) if (ro_on[j]) {
          rX[0] -= vO[j] * gO0[j]; rX[1] -= vO[j] * gO1[j];
          if (GEN) { rX[2] -= vO[j] * gO2[j]; rX[3] -= vO[j] * gO3[j]; }
          const double p = (sO[j] - qO.L) * vO[j]; svmax = fmax(svmax, p); svmin = fmin(svmin, p); sum_v += vO[j];
          prim = fmax(prim, fabs(rO[j]));
          if (RESTO && rs) {     // stationarity in p and n (rho + y - vp = 0, rho - y - vn = 0) and their complementarity
            edual_el = fmax(edual_el, fmax(fabs(RS_RHO + vO[j] - vP[j]), fabs(RS_RHO - vO[j] - vN[j])));
            const double cp_ = eP[j] * vP[j], cn_ = eN[j] * vN[j];
            svmax = fmax(svmax, fmax(cp_, cn_)); svmin = fmin(svmin, fmin(cp_, cn_)); sum_v += vP[j] + vN[j];
            Vel += eP[j] + eN[j];
          }
        }
        double dual = edual_el;
        if (xnode) {
#pragma unroll
          for (int i = 0; i < NX; ++i) dual = fmax(dual, fabs(rX[i]));
        }
        if (hasu) {
          dual = fmax(dual, fmax(fabs(rU[0]), fabs(rU[1])));
#pragma unroll
          for (int i = 0; i < NX; ++i) prim = fmax(prim, fabs(dfc[i]));
        }
        double ss[3] = {sum_lam, sum_v, Vel}, mm[4] = {dual, prim, svmax, -svmin};
        wv::reduce<RESTO ? 3 : 2, 4>(ss, mm);
        e_dual = wv::uni(mm[0]); e_prim = wv::uni(mm[1]);
        const double sv_hi = wv::uni(mm[2]), sv_lo = -wv::uni(mm[3]);
        const double n_vr = (RESTO && rs) ? n_v + 2 * n_el : n_v;               // bound multipliers of the running phase
        const double e_sd = fmax(S_MAX, (wv::uni(ss[0]) + wv::uni(ss[1])) / fmax(1.0, n_lam + n_vr)) / S_MAX;
        const double e_sc = fmax(S_MAX, wv::uni(ss[1]) / fmax(1.0, n_vr)) / S_MAX;
        const double base = fmax(e_dual / e_sd, e_prim);
        err0 = fmax(base, (n_vr > 0 ? sv_hi : 0.0) / e_sc);                      // complementarity error at mu = 0
        if (MPCB_TRACE_ARGS(a)->trace && b == MPCB_TRACE_ARGS(a)->trace_instance && lane == 0 && iters <= c.max_iter) {
          double* t = MPCB_TRACE_ARGS(a)->trace + (size_t)iters * 8;
          t[0] = mu; t[1] = err0; t[2] = theta; t[3] = fval;
        }
        if (!(RESTO && rs)) {
          // IPOPT's OptimalityErrorConvergenceCheck: "optimal" = scaled error AND the three unscaled gates (dual infeasibility and
          // complementarity of the scaled problem divided by the objective scaling; there is no constraint scaling); then the
          // acceptable-point counter with the reference's two options (kin.py:252-253)
          const double compl0 = n_vr > 0 ? sv_hi : 0.0;
          const auto* lc = &wv::late_args(a)->cfg;       // the nine tolerances are loaded here, once per iteration, and are dead again after the test
          if (err0 <= lc->tol && e_dual <= lc->dual_inf_tol * os && e_prim <= lc->constr_viol_tol && compl0 <= lc->compl_inf_tol * os) { status = MPCB_ST_SOLVED; break; }
          const double fcur = os * fval;
          const bool acc = lc->acceptable_iter > 0 && err0 <= lc->acceptable_tol && e_dual <= lc->acceptable_dual_inf_tol * os &&
                           e_prim <= lc->acceptable_constr_viol_tol && compl0 <= lc->acceptable_compl_inf_tol * os &&
                           fabs(fcur - cst[CS_ACC]) <= lc->acceptable_obj_change_tol * fmax(1.0, fabs(fcur));
          const double acc_cnt = acc ? cst[CS_ACC + 1] + 1.0 : 0.0;
          wv::sync();
          if (lane == 0) { cst[CS_ACC] = fcur; cst[CS_ACC + 1] = acc_cnt; }
          wv::sync();
          if (acc && acc_cnt >= (double)lc->acceptable_iter) { status = MPCB_ST_ACCEPTABLE; break; }
          if (iters >= c.max_iter) { status = MPCB_ST_MAXITER; break; }
        } else {
          // ----- restoration phase: has it done its job?  Violation of the ORIGINAL rows and the original barrier function here
          double t1 = fabs(rR), fm = 0, pl = 1.0, ti = fabs(rR);
          if (hasu) {
#pragma unroll
            for (int i = 0; i < NX; ++i) { t1 += fabs(dfc[i]); ti = fmax(ti, fabs(dfc[i])); }
          }
#pragma unroll
          for (int j = 0; j < NOBS; ++j) if (ro_on[j]) { const double r0 = rO[j] + (eP[j] - eN[j]); t1 += fabs(r0); ti = fmax(ti, fabs(r0)); pl *= sO[j] - qO.L; }
          auto dist = [&](const Bnd& q, double s_) { if (q.hasL) pl *= s_ - q.L; if (q.hasU) pl *= q.U - s_; };
          if (bu0_on) dist(qU0, U[0]);
          if (bu1_on) dist(qU1, U[1]);
          if (by_on) dist(qY, X[1]);
          if (bv_on) dist(qV, X[3]);
          if (rr_on) dist(qR, sR);
          if (hasu) {      // kin.py:195-205 with the main phase's constants
#pragma unroll
            for (int i = 0; i < NX; ++i) { const double e = X[i] - cst[CS_XS + i]; fm += cst[CS_Q + i] * e * e; }
            fm += cst[CS_R] * U[0] * U[0] + cst[CS_R + 1] * U[1] * U[1];
            if (k > 0 || c.du0_cost) {
              const double d0 = U[0] - (k ? Up0 : cst[CS_UL]), d1 = U[1] - (k ? Up1 : cst[CS_UL + 1]);
              fm += cst[CS_DR] * d0 * d0 + cst[CS_DR + 1] * d1 * d1;
            }
          }
          double so[3] = {t1, fm, log(pl)}, mo[1] = {ti};
          wv::reduce<3, 1>(so, mo);
          const double th1 = wv::uni(so[0]), thinf = wv::uni(mo[0]);
          bool leave = false;
          if (rit >= 1 && th1 <= RS_KAPPA * th_entry && th1 <= tmax_main) {
            const double phi_o = os * wv::uni(so[1]) - mu_main * wv::uni(so[2]);
            leave = !(th1 >= fm_theta && phi_o >= fm_phi);                          // acceptable to the main filter (= the entry pair)
          }
          if (!leave && err0 <= c.tol) {
            // the restoration problem itself is solved: violated rows -> locally infeasible; a feasible point that the main phase
            // could not leave (no restoration step was taken) -> the restoration has nothing to offer (IPOPT: Restoration_Failed)
            if (thinf > c.tol) { status = MPCB_ST_INFEASIBLE; break; }
            if (rit == 0) { status = MPCB_ST_RESTO_FAILED; break; }
            leave = true;
          }
          if (!leave) {
            // local-infeasibility certificate: barrier subproblem solved, hard rows satisfied, elastic violation above the gap bound
            const double V = wv::uni(ss[2]);
            const double em = fmax(base, fmax(fabs(sv_hi - mu), fabs(sv_lo - mu)) / e_sc);
            const double gap = (RS_GAP * n_vr * mu + 0.5 * (double)((NX + NU) * N) * sqrt(mu)) / RS_RHO;
            if (em <= K_EPS * mu && V > gap + 1e-6 && theta <= 0.01 * V) { status = MPCB_ST_INFEASIBLE; break; }
            if (iters >= c.max_iter) { status = MPCB_ST_MAXITER; break; }
            if (n_riters >= RS_MAX_ITERS) { status = MPCB_ST_RESTO_FAILED; break; }
          } else {
            // ----- back to the main phase: original cost, mu and filter bounds; lam = 0; bound duals kept unless one exceeds 1000
            rs = false; osc = os; write_main_cost();
            double vm = 0;
            auto vmx = [&](const Bnd& q, const Item& it) { if (q.hasL) vm = fmax(vm, it.vL); if (q.hasU) vm = fmax(vm, it.vU); };
            if (bu0_on) vmx(qU0, iU0);
            if (bu1_on) vmx(qU1, iU1);
            if (by_on) vmx(qY, iY);
            if (bv_on) vmx(qV, iV);
            if (rr_on) vmx(qR, iR);
#pragma unroll
            for (int j = 0; j < NOBS; ++j) { if (ro_on[j]) vm = fmax(vm, vO[j]); }
#pragma unroll
            for (int j = 0; j < NEL; ++j) { eP[j] = 0; eN[j] = 0; vP[j] = 0; vN[j] = 0; }
            if (wv::uni(wv::max(vm)) > 1000.0) {
              iU0.vL = iU0.vU = iU1.vL = iU1.vU = iY.vL = iY.vU = iV.vL = iV.vU = iR.vL = iR.vU = 1.0;
#pragma unroll
              for (int j = 0; j < NOBS; ++j) vO[j] = 1.0;
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) lam[i] = 0.0;
            mu = mu_main; tau = fmax(TAU_MIN, 1.0 - mu);
            theta_max = tmax_main; theta_min = tmin_main;
            if (lane == 0) { filt[0] = fm_theta; filt[1] = fm_phi; }
            nfilt = 1;
            wv::sync();
            double th, fl, prod;
            eval_lane(X, U, sR, sO, eP, eN, sp, cp, td, dfc, rR, rO, Up0, Up1, th, fl, prod);
            double sw[3] = {th, fl, log(prod)};
            wv::reduce<3, 0>(sw, nullptr);
            theta = wv::uni(sw[0]); fval = wv::uni(sw[1]); logsum = wv::uni(sw[2]);
            slow_run = 0;
            continue;
          }
        }
        // barrier parameter update (monotone, Fiacco-McCormick): max_i |s_i v_i - mu| from the two extremes
        const double mu_before = mu;
        for (;;) {
          const double comp = (n_vr > 0) ? fmax(fabs(sv_hi - mu), fabs(sv_lo - mu)) : 0.0;
          const double em = fmax(base, comp / e_sc);
          if (em <= K_EPS * mu && mu > mu_floor) {
            mu = wv::uni(fmax(mu_floor, fmin(K_MU * mu, mu * sqrt(mu))));
            tau = wv::uni(fmax(TAU_MIN, 1.0 - mu));
            nfilt = 0;
          } else break;
        }
        if (RESTO && rs && mu != mu_before) {      // the proximity weight follows mu: zeta = sqrt(mu); objective value of the new cost
          write_resto_cost(sqrt(mu), false, X, U);
          double fl = 0;
          if (xobj) {
#pragma unroll
            for (int i = 0; i < NX; ++i) { const double e = X[i] - cXS(i); fl += cQQ(i) * e * e; }
          }
          if (hasu) { const double e0 = U[0] - cUR(0), e1 = U[1] - cUR(1); fl += cRR(0) * e0 * e0 + cRR(1) * e1 * e1; }
#pragma unroll
          for (int j = 0; j < NOBS; ++j) if (ro_on[j]) fl += RS_RHO * (eP[j] + eN[j]);
          fval = wv::uni(wv::sum(fl));
        }
      }

      // ----- condensed stage QP: lane k writes the compact entries of stage k ------------------------------------
      double hxx = 0, hxy = 0, hyy = 0, hpp = 0, hpv = 0, hvv = 0, hvd = 0, hdd = 0, haa = 0, h44 = 0, h55 = 0, h46 = 0, h57 = 0;
      double hxp = 0, hxv = 0, hyp = 0, hyv = 0;      // GEN only
      {
        double g[NW] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (xq) {
          hxx += cWQ(0); hyy += cWQ(1); hpp += cWQ(2); hvv += cWQ(3);
#pragma unroll
          for (int i = 0; i < NX; ++i) g[i] += cWQ(i) * (X[i] - cXS(i));
        }
        if (hasu) {
          hdd += cWR(0); haa += cWR(1);
          g[6] += cWR(0) * (U[0] - cUR(0)); g[7] += cWR(1) * (U[1] - cUR(1));
          if (k > 0 || c.du0_cost) {
            const double w0 = cWDR(0), w1 = cWDR(1);
            const double d0 = U[0] - (k ? Up0 : cst[CS_UL]), d1 = U[1] - (k ? Up1 : cst[CS_UL + 1]);
            hdd += w0; h44 += w0; h46 -= w0; haa += w1; h55 += w1; h57 -= w1;
            g[6] += w0 * d0; g[4] -= w0 * d0; g[7] += w1 * d1; g[5] -= w1 * d1;
          }
          // sum_a lam_{k+1,a} T d2f_a
          const double v = X[3];
          if (!RK4) {
            hpp += T * (-ln[0] * v * cp - ln[1] * v * sp);
            hpv += T * (-ln[0] * sp + ln[1] * cp);
            hvd += T * ln[2] * sec2 * il;
            hdd += T * ln[2] * v * 2.0 * td * sec2 * il;
          } else {       // sum_a lam_{k+1,a} d2 Phi_a of the Runge-Kutta step (kin_rk4_derivs); (phi,delta) (phi,a) (v,a) (delta,a) use the rows GEN uses otherwise
            hpp += Hr.pp; hpv += Hr.pv; hvv += Hr.vv; hvd += Hr.vd; hdd += Hr.dd; haa += Hr.aa;
            hxp = Hr.pd; hxv = Hr.pa; hyp = Hr.va; hyv = Hr.da;
          }
        }
        double sig, gb;
        if (bu0_on) { item_sig_gb(iU0, 0.0, mu, sig, gb); hdd += sig; g[6] -= gb; }
        if (bu1_on) { item_sig_gb(iU1, 0.0, mu, sig, gb); haa += sig; g[7] -= gb; }
        if (by_on) { item_sig_gb(iY, 0.0, mu, sig, gb); hyy += sig; g[1] -= gb; }
        if (bv_on) { item_sig_gb(iV, 0.0, mu, sig, gb); hvv += sig; g[3] -= gb; }
        if (rr_on) { item_sig_gb(iR, rR, mu, sig, gb); hdd += sig; h44 += sig; h46 -= sig; g[6] -= gb; g[4] += gb; }
#pragma unroll
        for (int j = 0; j < NOBS; ++j) if (ro_on[j]) {
          sig = vO[j] * iO[j]; gb = mu * iO[j];
          if (RESTO && rs) {   // elastic row: Sigma_e = kappa Sigma_s and the residual r~ (oracle: build_stage_grad)
            const double isp = eP[j] / vP[j], isn = eN[j] / vN[j];              // 1 / Sigma_p, 1 / Sigma_n
            const double kap = 1.0 / (1.0 + sig * isp + sig * isn);
            const double rt = rO[j] + (RS_RHO + gb - mu / eP[j]) * isp + (gb - RS_RHO + mu / eN[j]) * isn;
            sig *= kap; gb -= sig * rt;
          } else gb -= sig * rO[j];
          if (!GEN) {
            hxx += sig * gO0[j] * gO0[j] - vO[j] * 2 * ix2(j);
            hxy += sig * gO0[j] * gO1[j];
            hyy += sig * gO1[j] * gO1[j] - vO[j] * 2 * iy2(j);
            g[0] -= gb * gO0[j]; g[1] -= gb * gO1[j];
          } else {
            // sig g g^T - v d2c, c = h(q) - (1-gamma) h(p), q = p + T v (cos phi, sin phi)
            const double v = X[3], y = vO[j], Tv = T * v;
            const double dqx = 2 * (X[0] + Tv * cp - ox(j)) * ix2(j), dqy = 2 * (X[1] + Tv * sp - oy(j)) * iy2(j);
            const double g0 = gO0[j], g1 = gO1[j], g2 = gO2[j], g3 = gO3[j];
            hxx += sig * g0 * g0 - y * (2 * c.gamma * ix2(j));
            hxy += sig * g0 * g1;
            hyy += sig * g1 * g1 - y * (2 * c.gamma * iy2(j));
            hxp += sig * g0 * g2 - y * (-2 * ix2(j) * Tv * sp);
            hxv += sig * g0 * g3 - y * (2 * ix2(j) * T * cp);
            hyp += sig * g1 * g2 - y * (2 * iy2(j) * Tv * cp);
            hyv += sig * g1 * g3 - y * (2 * iy2(j) * T * sp);
            hpp += sig * g2 * g2 - y * (2 * Tv * Tv * (ix2(j) * sp * sp + iy2(j) * cp * cp) - Tv * (dqx * cp + dqy * sp));
            hpv += sig * g2 * g3 - y * (T * (dqy * cp - dqx * sp) + 2 * T * Tv * sp * cp * (iy2(j) - ix2(j)));
            hvv += sig * g3 * g3 - y * (2 * T * T * (ix2(j) * cp * cp + iy2(j) * sp * sp));
            g[0] -= gb * g0; g[1] -= gb * g1; g[2] -= gb * g2; g[3] -= gb * g3;
          }
        }
        if (isnode) {
#pragma unroll
          for (int i = 0; i < NW; ++i) ent[(E_G0 + i) * ld + k] = g[i];
          double* fk = fw + k * FWS;                     // [A | d | B] and T of the stage: the variable slots of records 0..3
          fk[2] = hasu ? a02 : 0.0; fk[3] = hasu ? a03 : 0.0; fk[FWR + 2] = hasu ? a12 : 0.0; fk[FWR + 3] = hasu ? a13 : 0.0;
          fk[2 * FWR + 3] = hasu ? a23 : 0.0;
          if (!RK4) { fk[2 * FWR + FW_BX] = hasu ? b20 : 0.0; fk[3 * FWR + FW_BX] = T; }
          else {         // dense B: (b_r0, b_r1) in slots 4, 5 of record r
            fk[4] = hasu ? Jr.b00 : 0.0; fk[5] = hasu ? Jr.b01 : 0.0; fk[FWR + 4] = hasu ? Jr.b10 : 0.0; fk[FWR + 5] = hasu ? Jr.b11 : 0.0;
            fk[2 * FWR + 4] = hasu ? b20 : 0.0; fk[2 * FWR + 5] = hasu ? Jr.b21 : 0.0; fk[3 * FWR + 4] = 0.0; fk[3 * FWR + 5] = T;
          }
#pragma unroll
          for (int i = 0; i < NX; ++i) fk[i * FWR + FW_C0] = dfc[i];
          Pst[k * PST + PS_ZERO] = 0.0;
          ent[E_HXY * ld + k] = hxy; ent[E_HPV * ld + k] = hpv; ent[E_HVD * ld + k] = hvd;
          ent[E_H44 * ld + k] = h44; ent[E_H55 * ld + k] = h55; ent[E_H46 * ld + k] = h46; ent[E_H57 * ld + k] = h57;
          if (GEN || RK4) { ent[E_HXP * ld + k] = hxp; ent[E_HXV * ld + k] = hxv; ent[E_HYP * ld + k] = hyp; ent[E_HYV * ld + k] = hyv; }
        }
      }

      MPCB_STAMP(t_b);
      // ----- factorisation with inertia correction: backward Riccati sweep, lanes = entries ----------------------
      double dw = 0.0; bool first_try = true, fact_ok = false;
#pragma clang loop unroll(disable)
      for (int tries = 0; tries < 60; ++tries) {
        if (isnode) {
          ent[E_HXX * ld + k] = hxx + (xnode ? dw : 0.0); ent[E_HYY * ld + k] = hyy + (xnode ? dw : 0.0);
          ent[E_HPP * ld + k] = hpp + (xnode ? dw : 0.0); ent[E_HVV * ld + k] = hvv + (xnode ? dw : 0.0);
          ent[E_HDD * ld + k] = hdd + (hasu ? dw : 0.0); ent[E_HAA * ld + k] = haa + (hasu ? dw : 0.0);
        }
        wv::sync();
        // terminal: P_N = H_N (state block), p_N = g_N
        if (ei < NA && ej < NA) Pst[N * PST + ei * NA + ej] = ent[sHij + N];
        if (ei < NA && ej == 0) Pst[N * PST + PS_P + ei] = ent[sGi + N];
        wv::sync();
        // Stage entries (10 table values per lane) do not depend on the recursion: they are prefetched one stage ahead,
        // into the SAME registers as soon as the running stage has spent them (no second set), so that only the P+ row and
        // the W exchange sit on the critical path.  MPCB_SCHED_FENCE keeps the compiler from sinking the batched loads next
        // to their uses.
        // staging block of the sweep's tail: rows 6, 7 of M interleaved at 0..15 (M6j at 2j, M7j at 2j+1), their accumulators likewise at
        // 32..47, the other lanes store into pads
        const int sOff = lane >= NA * 8 ? 2 * ej + (ei - NA) : 16 + (lane & 15);
        struct StageEnt { double cw[NX], cm[NX], start, hmat; };
        auto load_ent = [&](int s, StageEnt& e) {
#pragma unroll
          for (int r = 0; r < NX; ++r) { e.cw[r] = fw[s * FWS + sCW[r]]; e.cm[r] = fw[s * FWS + sCM[r]]; }
          e.start = ent[sStart + s]; e.hmat = ent[sHij + s];
        };
        // gains of the previous stage, stored one stage late: the selects and the two DS stores then issue while this stage
        // waits for its P+ row instead of sitting between the P store and the next P+ read
        double pK0 = 0, pK1 = 0, pkf0 = 0, pkf1 = 0;
        auto stage = [&](int s, StageEnt& e) -> bool {
          // W = P+ [A B | d] (+ p+ in the affine column)      lane (i,j): row i of P+, column j
          const double* Pn = Pst + (s + 1) * PST;
          double Pr[NX];
#pragma unroll
          for (int r = 0; r < NX; ++r) Pr[r] = Pn[eiA * NA + r];
          const double w0 = Pn[uWOff];
          MPCB_SCHED_FENCE();
          fw[(s + 1) * FWS + kOff] = kRow1 ? pK1 : pK0;           // (the first stage of a sweep writes zeros into the unused row N)
          fw[(s + 1) * FWS + kfOff] = kfLane1 ? pkf1 : pkf0;
          MPCB_SCHED_FENCE();
          // one FMA chain: a dependent v_fma_f64 issues after 5.6 ticks against 4.5 for an independent one (tools/ubench/
          // fma_latency.hip), so splitting the chain only adds the instruction that joins the halves
          const double w = fma(Pr[3], e.cw[3], fma(Pr[2], e.cw[2], fma(Pr[1], e.cw[1], fma(Pr[0], e.cw[0], w0))));
          Wl[wOff] = w;
          wv::sync();
          double Wc[NX];
#pragma unroll
          for (int r = 0; r < NX; ++r) Wc[r] = Wl[ej * NA + r];
          const double m0 = Wl[uMOff];
          MPCB_SCHED_FENCE();
          // M = H + [A B]^T W; in the affine lanes the same sum is m = g + [A B]^T q
          const double acc = fma(e.cm[3], Wc[3], fma(e.cm[2], Wc[2], fma(e.cm[1], Wc[1], fma(e.cm[0], Wc[0], e.start)))) + m0;
          const double Mx = aff ? e.hmat : acc;                   // matrix value of this lane (column 4: H itself)
          // rows 6, 7 of M and the two m_u (accumulator of the affine lanes (6,4), (7,4)) go through a 64-double staging block,
          // rows interleaved ({M6j, M7j} adjacent): two stores and five 16-byte reads (three of them wave-uniform) instead of ten
          // v_readlane and eight ds_bpermute
          Sl[sOff] = Mx; Sl[sOff + 32] = acc;
          wv::sync();
          // control block Muu and m_u
          const double m11 = Sl[2 * NA], m12 = Sl[2 * (NA + 1)], m22 = Sl[2 * (NA + 1) + 1];
          const double mu6 = Sl[32 + 2 * 4], mu7 = Sl[32 + 2 * 4 + 1];
          // rows 6,7 of M at column j and at column i (M is symmetric up to rounding)
          const double M6j = Sl[2 * ej], M7j = Sl[2 * ej + 1];
          const double M6i = Sl[2 * ei], M7i = Sl[2 * ei + 1];
          // e is spent.  Prefetch for the next stage behind the exchanges of this one: the DS queue is in order, so these 10
          // reads must not sit in front of the W / M traffic of the recursion (index -1 after stage 0 reads the tail of the
          // filter region: in bounds, never used)
          MPCB_SCHED_FENCE();
          load_ent(s - 1, e);
          MPCB_SCHED_FENCE();
          const double det = m11 * m22 - m12 * m12, dmar = det - 1e-14 * m11 * m22;
          const bool okpd = (m11 > 0) & (dmar > 0) & (dmar < 1e300);                    // wave-uniform; false for NaN / inf
          const double idet = wv::rcp(det);
          const double i11 = m22 * idet, i12 = -m12 * idet, i22 = m11 * idet;
          const double kf0 = -(i11 * mu6 + i12 * mu7), kf1 = -(i12 * mu6 + i22 * mu7);
          const double K0j = -(i11 * M6j + i12 * M7j), K1j = -(i12 * M6j + i22 * M7j);
          // P_s: the lanes of the upper triangle (i <= j) store their entry to (i,j) AND (j,i), the lanes below the diagonal
          // store into pad slots.  P(i,j) and P(j,i) are equal in exact arithmetic, but their cancellation errors (entries of
          // 1e12 when bounds are active at mu -> 1e-9) differ, and without this mirroring the asymmetry compounds through the
          // sweep: the positive-definiteness test turns noisy near convergence and delta_w escalates to 1e2 where 1e-2 does.
          const double Pij = Mx + M6i * K0j + M7i * K1j;
          Pst[s * PST + pOff] = Pij;
          Pst[s * PST + pOffT] = Pij;
          Pst[s * PST + psOff] = acc + M6i * kf0 + M7i * kf1;               // p_s (affine lanes), pad slot elsewhere
          pK0 = K0j; pK1 = K1j; pkf0 = kf0; pkf1 = kf1;
          wv::sync();
          return okpd;                                   // a failed stage leaves garbage behind; the sweep is repeated with a larger delta_w
        };
        bool pd = true;
        {
          StageEnt eA;
          load_ent(N - 1, eA);
          int s = N - 1;
          // four stages per trip: the per-lane table addresses (10 registers) advance once per trip, the stages in between
          // use immediate offsets
#pragma clang loop unroll(disable)
          for (; s >= 3 && pd; s -= 4) {
            const bool p1 = stage(s, eA);
            const bool p2 = stage(s - 1, eA);
            const bool p3 = stage(s - 2, eA);
            const bool p4 = stage(s - 3, eA);
            pd = p1 & p2 & p3 & p4;
          }
          if (pd && s >= 0) {
            pd = stage(s, eA);
            if (pd && s >= 1) {
              pd = stage(s - 1, eA);
              if (pd && s >= 2) pd = stage(s - 2, eA);
            }
          }
          fw[kOff] = kRow1 ? pK1 : pK0;                           // gains of stage 0
          fw[kfOff] = kfLane1 ? pkf1 : pkf0;
          wv::sync();
        }
        if (pd) { fact_ok = true; if (dw > 0) dw_last = dw; break; }
        if (first_try) { dw = (dw_last == 0.0) ? DW_FIRST : fmax(DW_MIN, KW_MINUS * dw_last); first_try = false; }
        else dw *= (dw_last == 0.0) ? KW_PLUS_FIRST : KW_PLUS;
        if (dw > DW_MAX) break;
      }
      if (!fact_ok) { status = MPCB_ST_NUMERIC; break; }

      MPCB_STAMP(t_c);
      // ----- forward roll-out of the step: lane i < 6 advances component i of [dX_s; dU_{s-1}] ---------------------
      // One stage is  t_i = c_i0 + sum_r C_ir v_r  (rows 0..3: the A part of the state update plus the defect, rows 4, 5: the gain
      // rows, t = dU_s), then  n_i = t_i + b_i0 dU_s[0] + b_i1 dU_s[1]  (the B part: b20 in row 2, T_s in row 3).  The six numbers
      // of v travel through scalar registers (v_readlane); a lane reads the nine numbers of its own record of the stage's fw row
      // instead of every lane reading the whole row, and the steps are written to the rows of the (spent) condensed gradient, [component][node], where lane k picks up its node's.
      double dX[NX] = {0, 0, 0, 0}, dU[NU] = {0, 0};
      {
        const int lq = wv::opaque(lane);                 // (addresses re-formed per iteration instead of staying live — and spilled — through the solve)
        const int li = lq < NA ? lq : 0;
        // rows 0..3 hold dX_{s+1}, rows 4, 5 hold dU_s; the other lanes store into the two remaining (equally spent) gradient rows,
        // so that the store needs no EXEC change
        double* hist = ent + (E_G0 + (lq < NA ? lq : NA + (lq & 1))) * ld + (lq < NX ? 1 : 0);
        if (lane < NX) hist[-1] = 0.0;                                   // dX_0 = 0 (X_0 is pinned)
        struct FwRec { double c[NA], c0, bx; };
        auto load_fw = [&](int s, FwRec& f) {
          const double* q = fw + s * FWS + li * FWR;      // one address, four 16-byte reads
#pragma unroll
          for (int r = 0; r < NA; ++r) f.c[r] = q[r];
          f.c0 = q[FW_C0]; f.bx = q[FW_BX];
        };
        double v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0;           // wave-uniform
        auto fstage = [&](int s, const FwRec& f, FwRec& nxt) {
          load_fw(s + 1, nxt);                             // (row N exists and holds finite numbers; its record is never used)
          MPCB_SCHED_FENCE();
          double t, du0, du1, n;
          if (!RK4) {
            t = fma(f.c[5], v5, fma(f.c[4], v4, fma(f.c[3], v3, fma(f.c[2], v2, fma(f.c[1], v1, fma(f.c[0], v0, f.c0))))));
            du0 = wv::bcast(t, NX); du1 = wv::bcast(t, NX + 1);
            n = fma(f.bx, lq == 3 ? du1 : du0, t);      // the control enters row 2 (b20 dU[0]) and row 3 (T dU[1])
          } else {
            // slots 4, 5 of a state row's record hold its two control coefficients (dense B), of a gain row's the U_prev coefficients
            const double tx = fma(f.c[3], v3, fma(f.c[2], v2, fma(f.c[1], v1, fma(f.c[0], v0, f.c0))));
            t = lq >= NX ? fma(f.c[5], v5, fma(f.c[4], v4, tx)) : tx;
            du0 = wv::bcast(t, NX); du1 = wv::bcast(t, NX + 1);
            n = lq < NX ? fma(f.c[5], du1, fma(f.c[4], du0, t)) : t;
          }
          hist[s] = n;
          v0 = wv::bcast(n, 0); v1 = wv::bcast(n, 1); v2 = wv::bcast(n, 2); v3 = wv::bcast(n, 3); v4 = du0; v5 = du1;
        };
        {
          FwRec fA, fB;
          load_fw(0, fA);
          int s = 0;
#pragma clang loop unroll(disable)
          for (; s + 1 < N; s += 2) { fstage(s, fA, fB); fstage(s + 1, fB, fA); }
          if (s < N) fstage(s, fA, fB);
        }
        wv::sync();
        if (isnode) {
#pragma unroll
          for (int i = 0; i < NX; ++i) dX[i] = ent[(E_G0 + i) * ld + k];
        }
        if (hasu) { dU[0] = ent[(E_G0 + NX) * ld + k]; dU[1] = ent[(E_G0 + NX + 1) * ld + k]; }
      }
      const double dUp0 = wv::shfl(dU[0], k - 1), dUp1 = wv::shfl(dU[1], k - 1);   // dU_{k-1}
      // costate of the full step: lamF_k = P_k [dX_k; dU_{k-1}] + p_k  (node-parallel)
      double lamF[NX] = {0, 0, 0, 0};
      if (xnode) {
        const double* Pk = Pst + k * PST;
        const double dxa[NA] = {dX[0], dX[1], dX[2], dX[3], dUp0, dUp1};
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          double s = Pk[PS_P + i];
#pragma unroll
          for (int r = 0; r < NA; ++r) s += Pk[i * NA + r] * dxa[r];
          lamF[i] = s;
        }
      }
      // slack steps of the general rows
      const double dsR = rr_on ? (dU[0] - dUp0) + rR : 0.0;
      double dsO[NOB], dP[NEL], dN[NEL], dvP[NEL], dvN[NEL];
#pragma unroll
      for (int j = 0; j < NEL; ++j) { dP[j] = 0; dN[j] = 0; dvP[j] = 0; dvN[j] = 0; }
#pragma unroll
      for (int j = 0; j < NOBS; ++j) {
        dsO[j] = ro_on[j] ? gO0[j] * dX[0] + gO1[j] * dX[1] + rO[j] : 0.0;
        if (GEN && ro_on[j]) dsO[j] += gO2[j] * dX[2] + gO3[j] * dX[3];
        if (RESTO && rs && ro_on[j]) {        // elastic row (oracle: riccati_vector)
          const double sig = vO[j] * iO[j], bs = mu * iO[j];
          const double sp_ = vP[j] / eP[j], sn_ = vN[j] / eN[j], kap = 1.0 / (1.0 + sig / sp_ + sig / sn_);
          const double cp_ = (RS_RHO + bs - mu / eP[j]) / sp_, cn_ = (bs - RS_RHO + mu / eN[j]) / sn_;
          dsO[j] = kap * (dsO[j] + cp_ + cn_);
          dP[j] = sig * dsO[j] / sp_ - cp_;
          dN[j] = -sig * dsO[j] / sn_ + cn_;
          dvP[j] = mu / eP[j] - vP[j] - sp_ * dP[j];
          dvN[j] = mu / eN[j] - vN[j] - sn_ * dN[j];
        }
      }

      // ----- fraction to the boundary (as the largest step ratios) and d(barrier function) along the step --------
      double a_pr, a_du, dphi;
      {
        double rpr = 0, rdu = 0, d = 0;
        auto ftb = [&](const Bnd& q, const Item& it, double ds) {
          double dvL, dvU; item_dv(q, it, ds, mu, dvL, dvU);
          if (q.hasL) { rpr = fmax(rpr, -ds * it.iL); rdu = fmax(rdu, -dvL * wv::rcp(it.vL)); d -= mu * ds * it.iL; }
          if (q.hasU) { rpr = fmax(rpr, ds * it.iU); rdu = fmax(rdu, -dvU * wv::rcp(it.vU)); d += mu * ds * it.iU; }
        };
        if (bu0_on) ftb(qU0, iU0, dU[0]);
        if (bu1_on) ftb(qU1, iU1, dU[1]);
        if (by_on) ftb(qY, iY, dX[1]);
        if (bv_on) ftb(qV, iV, dX[3]);
        if (rr_on) ftb(qR, iR, dsR);
#pragma unroll
        for (int j = 0; j < NOBS; ++j) if (ro_on[j]) {
          const double dv = mu * iO[j] - vO[j] - vO[j] * iO[j] * dsO[j];
          rpr = fmax(rpr, -dsO[j] * iO[j]); rdu = fmax(rdu, -dv * wv::rcp(vO[j])); d -= mu * dsO[j] * iO[j];
          if (RESTO && rs) {
            rpr = fmax(rpr, fmax(-dP[j] / eP[j], -dN[j] / eN[j])); rdu = fmax(rdu, fmax(-dvP[j] / vP[j], -dvN[j] / vN[j]));
            d += (RS_RHO - mu / eP[j]) * dP[j] + (RS_RHO - mu / eN[j]) * dN[j];
          }
        }
        if (xq) {
#pragma unroll
          for (int i = 0; i < NX; ++i) d += cWQ(i) * (X[i] - cXS(i)) * dX[i];
        }
        if (hasu) {
          d += cWR(0) * (U[0] - cUR(0)) * dU[0] + cWR(1) * (U[1] - cUR(1)) * dU[1];
          if (k > 0 || c.du0_cost) {
            d += cWDR(0) * (U[0] - (k ? Up0 : cst[CS_UL])) * (dU[0] - (k ? dUp0 : 0.0));
            d += cWDR(1) * (U[1] - (k ? Up1 : cst[CS_UL + 1])) * (dU[1] - (k ? dUp1 : 0.0));
          }
        }
        double ss[1] = {d}, mm[2] = {rpr, rdu};
        wv::reduce<1, 2>(ss, mm);
        dphi = wv::uni(ss[0]);
        const double r1 = wv::uni(mm[0]), r2 = wv::uni(mm[1]);
        a_pr = wv::uni((r1 > tau) ? tau / r1 : 1.0);          // min(1, tau / max ratio)
        a_du = wv::uni((r2 > tau) ? tau / r2 : 1.0);
      }
      const double phi0 = wv::uni(osc * fval - mu * logsum), th0 = theta;
      // theta^s_theta and (-dphi)^s_phi of the switching condition: once per iteration (a_min and every trial of the line search use them)
      const bool sw_on = th0 <= theta_min && dphi < 0;
      double pw_th = 0.0, pw_dp = 1.0;
      if (sw_on) { pw_th = pow(th0, S_THETA); pw_dp = pow(-dphi, S_PHI); }
      double a_min;
      if (dphi < 0) {
        a_min = fmin(G_THETA, G_PHI * th0 / (-dphi));
        if (sw_on) a_min = fmin(a_min, DELTA * pw_th / pw_dp);
      } else a_min = G_THETA;
      a_min = wv::uni(a_min * G_ALPHA);

      MPCB_STAMP(t_d);
      // ----- filter line search: trial evaluations are lane-parallel ----------------------------------------------
      double alpha = a_pr; bool accepted = false, armijo_type = false;
      double Xt[NX], Ut[NU], dft[NX], sRt, rRt, sOt[NOB], rOt[NOB], pt[NEL], nt[NEL], upt0, upt1, st_, ct_, tt_, et_, tht = 0, ft = 0, lst = 0;
#pragma unroll
      for (int j = 0; j < NEL; ++j) { pt[j] = 0; nt[j] = 0; }
#pragma clang loop unroll(disable)
      for (;;) {
#pragma unroll
        for (int i = 0; i < NX; ++i) Xt[i] = X[i] + alpha * dX[i];
        Ut[0] = U[0] + alpha * dU[0]; Ut[1] = U[1] + alpha * dU[1];
        sRt = sR + alpha * dsR;
#pragma unroll
        for (int j = 0; j < NOBS; ++j) sOt[j] = sO[j] + alpha * dsO[j];
        if (RESTO && rs) {
#pragma unroll
          for (int j = 0; j < NOBS; ++j) { pt[j] = eP[j] + alpha * dP[j]; nt[j] = eN[j] + alpha * dN[j]; }
        }
        trig(Xt[2], Ut[0], st_, ct_, tt_, et_);
        double th, fl, prod;
        const bool okl = eval_lane(Xt, Ut, sRt, sOt, pt, nt, st_, ct_, tt_, dft, rRt, rOt, upt0, upt1, th, fl, prod);
        double sv[4] = {th, fl, okl ? log(prod) : 0.0, okl ? 0.0 : 1.0};
        wv::reduce<4, 0>(sv, nullptr);
        tht = wv::uni(sv[0]); ft = wv::uni(sv[1]); lst = wv::uni(sv[2]);
        const double phit = osc * ft - mu * lst;
        const bool ok = (wv::uni(sv[3]) == 0.0) && isfinite(tht) && isfinite(phit);
        if (ok && tht <= theta_max) {
          bool fok = true;
          for (int e = lane; e < nfilt; e += 64) if (tht >= filt[2 * e] && phit >= filt[2 * e + 1]) fok = false;
          if (wv::all(fok)) {
            bool sw = false;
            if (sw_on) sw = alpha * pw_dp > DELTA * pw_th;
            if (th0 <= theta_min && sw) {
              // IPOPT's Compare_le(lhs, rhs, base): lhs - rhs <= 10 eps |base| — round-off slack on both acceptance tests
              // (ArmijoHolds / IsAcceptableToCurrentIterate in IpFilterLSAcceptor.cpp)
              if ((phit - phi0) - ETA_PHI * alpha * dphi <= 10 * 2.220446049250313e-16 * fabs(phi0)) { accepted = true; armijo_type = true; }
            } else if (tht - (1 - G_THETA) * th0 <= 10 * 2.220446049250313e-16 * fabs(th0) ||
                       (phit - phi0) + G_PHI * th0 <= 10 * 2.220446049250313e-16 * fabs(phi0)) accepted = true;
          }
        }
        if (accepted) break;
        alpha = wv::uni(alpha * 0.5);
        if (alpha < a_min || alpha < 1e-16) break;
      }
      if (MPCB_TRACE_ARGS(a)->trace && b == MPCB_TRACE_ARGS(a)->trace_instance && lane == 0) {
        double* t = MPCB_TRACE_ARGS(a)->trace + (size_t)iters * 8;
        t[4] = a_pr; t[5] = accepted ? alpha : 0.0; t[6] = a_du; t[7] = dw;
#if defined(MPCB_STAMPS) && !defined(MPCB_WAVE_EMU)
        MPCB_STAMP(t_e);
        t[4] = (double)(t_b - t_a); t[5] = (double)(t_c - t_b); t[6] = (double)(t_d - t_c); t[7] = (double)(t_e - t_d);
#endif
      }
      // hand-over of an instance that needs the restoration phase from the first pass to the restoration pass
      auto hand_over = [&](int it_done) {
        status = MPCB_ST_NEEDS_RESTO;
        if (lane == 0 && a.work) {
          double* wk = a.work + (size_t)b * WK_SIZE;
          wk[WK_MU] = mu; wk[WK_THMAX] = theta_max; wk[WK_THMIN] = theta_min; wk[WK_ITERS] = (double)it_done; wk[WK_DW] = dw_last;
          wk[WK_START] = pass == MPCB_PASS_SECOND ? 1.0 : 0.0;
        }
      };
      if (!accepted) {
        if (RESTO && rs) { status = MPCB_ST_RESTO_FAILED; break; }                    // the restoration's own line search failed
        if (!c.restoration) { status = MPCB_ST_LINESEARCH; break; }
        // failure at an (almost) feasible point = round-off in the end game: nothing to restore (IPOPT: "Restoration phase is called
        // at point that is almost feasible" -> Restoration_Failed); the nearly converged iterate is returned as it is
        if (e_prim <= c.tol) { status = MPCB_ST_RESTO_FAILED; break; }
        if (!RESTO) { hand_over(iters); break; }                                      // where IPOPT enters restoration
        enter = true; continue;
      }
      if (!armijo_type) {
        if (nfilt < FILTER_MAX) {
          if (lane == 0) { filt[2 * nfilt] = (1 - G_THETA) * th0; filt[2 * nfilt + 1] = phi0 - G_PHI * th0; }
          ++nfilt;
        }
        wv::sync();
      }

      // ----- accept the trial point: duals first (they use the reciprocals of the old point) ----------------------
      // safeguard of the duals (IPOPT kappa_sigma): v within [mu / (kappa s), kappa mu / s]; the two scalars once, a multiplication per item
      const double mu_hi = K_SIGMA * mu, mu_lo = mu * (1.0 / K_SIGMA);
      auto upd = [&](const Bnd& q, Item& it, double ds, double snew) {
        double dvL, dvU; item_dv(q, it, ds, mu, dvL, dvU);
        it.vL += a_du * dvL; it.vU += a_du * dvU;
        item_recip(q, snew, it);
        if (q.hasL) it.vL = fmax(fmin(it.vL, mu_hi * it.iL), mu_lo * it.iL);
        if (q.hasU) it.vU = fmax(fmin(it.vU, mu_hi * it.iU), mu_lo * it.iU);
      };
      if (bu0_on) upd(qU0, iU0, dU[0], Ut[0]);
      if (bu1_on) upd(qU1, iU1, dU[1], Ut[1]);
      if (by_on) upd(qY, iY, dX[1], Xt[1]);
      if (bv_on) upd(qV, iV, dX[3], Xt[3]);
      if (rr_on) upd(qR, iR, dsR, sRt);
#pragma unroll
      for (int j = 0; j < NOBS; ++j) if (ro_on[j]) {
        const double dv = mu * iO[j] - vO[j] - vO[j] * iO[j] * dsO[j];
        vO[j] += a_du * dv;
        iO[j] = wv::rcp(sOt[j] - qO.L);
        vO[j] = fmax(fmin(vO[j], mu_hi * iO[j]), mu_lo * iO[j]);
        if (RESTO && rs) {
          eP[j] = pt[j]; eN[j] = nt[j];
          vP[j] += a_du * dvP[j]; vN[j] += a_du * dvN[j];
          vP[j] = fmax(fmin(vP[j], K_SIGMA * mu / eP[j]), mu / (K_SIGMA * eP[j]));
          vN[j] = fmax(fmin(vN[j], K_SIGMA * mu / eN[j]), mu / (K_SIGMA * eN[j]));
        }
      }
      if (xnode) {
#pragma unroll
        for (int i = 0; i < NX; ++i) { X[i] = Xt[i]; lam[i] += alpha * (lamF[i] - lam[i]); }
      }
      if (hasu) { U[0] = Ut[0]; U[1] = Ut[1]; }
      sR = sRt; rR = rRt; Up0 = upt0; Up1 = upt1;
      sp = st_; cp = ct_; td = tt_; sec2 = et_;
#pragma unroll
      for (int i = 0; i < NX; ++i) dfc[i] = dft[i];
#pragma unroll
      for (int j = 0; j < NOBS; ++j) { sO[j] = sOt[j]; rO[j] = rOt[j]; }
      row_grads(sp, cp);
      theta = tht; fval = ft; logsum = lst;
      if (!isfinite(theta) || !isfinite(fval)) { status = MPCB_ST_NUMERIC; break; }
      ++iters;
      if (RESTO && rs) { ++rit; ++n_riters; }
      else if (c.restoration) {
        // early entry into restoration (or hand-over to the second start): TRIG_K accepted steps in a row shorter than TRIG_ALPHA that together reduced theta by less
        // than the factor TRIG_THETA (a slack pinned at its bound with the row still violated: the pattern of an infeasible
        // instance; IPOPT itself waits for the line search to fail, dozens of such steps later)
        if (alpha < TRIG_ALPHA && theta > 1e-6) { if (slow_run == 0) slow_theta0 = th0; ++slow_run; } else slow_run = 0;
        if (slow_run >= TRIG_K && theta > TRIG_THETA * slow_theta0) {
          slow_run = 0;
          if (!RESTO) { hand_over(iters); break; }
          enter = true;
        }
      }
    }
  } else {
    trig(X[2], U[0], sp, cp, td, sec2);
    double th, fl, prod;
    eval_lane(X, U, sR, sO, eP, eN, sp, cp, td, dfc, rR, rO, Up0, Up1, th, fl, prod);
    fval = wv::sum(fl);
  }

  if (RESTO && rs) {    // ended inside the restoration phase: report the objective of the original problem (kin.py:195-205)
    rs = false; osc = os; write_main_cost();
    double th, fl, prod;
    eval_lane(X, U, sR, sO, eP, eN, sp, cp, td, dfc, rR, rO, Up0, Up1, th, fl, prod);
    fval = wv::sum(fl);
  }
  // ----- outputs (reference ordering), staged through LDS for coalesced stores ----------------------------------
  // the output pointers and sizes are read from the kernel arguments HERE (wv::late_args) instead of being held in scalar registers
  // through the whole solve: the kernels spill ~190 SGPRs into VGPR lanes, every reload is a v_readlane in the iteration loop
#ifdef MPCB_NO_LATE_OUT
  const MpcbKArgs& ao = a;
#else
  const MpcbKArgs& ao = *wv::late_args(a);
#endif
  // `ko` = k behind an optimisation barrier: the LDS addresses of the z staging are re-formed here instead of being kept
  // live (and spilled) from the identical expressions at kernel start — hipcc 7.2 mis-reloaded such a spilled address in the
  // dyn<3> build (lanes >= 1 wrote their X rows to zbuf[0..5]).
  const int ko = wv::opaque(k);
  const int lo = wv::opaque(lane);      // same for the lane index of the coalesced copy loops
  wv::sync();
  if (hasu) { zbuf[NU * ko] = U[0]; zbuf[NU * ko + 1] = U[1]; }
  if (isnode) {
#pragma unroll
    for (int i = 0; i < NX; ++i) zbuf[NU * N + NX * ko + i] = X[i];
  }
  wv::sync();
  for (int i = lo; i < nz; i += 64) ao.z[(size_t)b * nz + i] = zbuf[i];
  if (lo == 0) {
    if (ao.obj) ao.obj[b] = fval;
    if (ao.status) ao.status[(size_t)b * ao.st_stride] = status;
    // iterations of both attempts are counted (cfg.second_start): the first attempt leaves its total in the hand-over record, the
    // passes of the second attempt add it (read here, not kept live through the solve)
    int it_prev = 0;
    if (ao.work) {
      double* wk = ao.work + (size_t)b * mpcbk::WK_SIZE;
      const bool second_attempt = RESTO ? wk[mpcbk::WK_START] != 0.0 : pass == MPCB_PASS_SECOND;
      if (second_attempt) it_prev = (int)wk[mpcbk::WK_ITPREV];
      else wk[mpcbk::WK_ITPREV] = (double)iters;       // (also at a hand-over: with cfg.second_start = 1 no restoration pass of the first attempt follows)
    }
    if (ao.iters) ao.iters[(size_t)b * ao.st_stride] = iters + it_prev;
    if (ao.kkt) { double* q = ao.kkt + (size_t)b * 4; q[0] = err0; q[1] = e_prim; q[2] = e_dual / os; q[3] = mu; }
  }
  if (ao.want_mult && ao.lam_x) {
    wv::sync();
    for (int i = lo; i < nz; i += 64) zbuf[i] = 0.0;
    wv::sync();
    if (bu0_on) zbuf[NU * ko] = -item_y(qU0, iU0) / os;
    if (bu1_on) zbuf[NU * ko + 1] = -item_y(qU1, iU1) / os;
    if (by_on) zbuf[NU * N + NX * ko + 1] = -item_y(qY, iY) / os;
    if (bv_on) zbuf[NU * N + NX * ko + 3] = -item_y(qV, iV) / os;
    wv::sync();
    for (int i = lo; i < nz; i += 64) ao.lam_x[(size_t)b * nz + i] = zbuf[i];
  }
  if (ao.want_mult && ao.lam_g) {
    // rows: [X_0 - P](NX), dynamics (NX*N), rate (N-1 if on), obstacles (rows * nobs)   kin.py:190-247
    double* out = ao.lam_g + (size_t)b * ao.ng;
    const int r_dyn = NX, r_rate = NX + NX * N, n_rate = qR.on ? N - 1 : 0, r_obs = r_rate + n_rate;
    double ln[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) ln[i] = wv::shfl(lam[i], k + 1);
    double cx = 0, cy = 0;
    if (GEN) {
      // the general-gamma row of node k-1 reads h(X_k) in the reference's form: its multiplier times grad h(X_k) belongs to
      // the dynamics row that defines X_k (lam_c = -v / os, so the correction is + v / os * grad h)
#pragma unroll
      for (int j = 0; j < NOBS; ++j) {
        const double vp = wv::shfl(ro_on[j] ? vO[j] : 0.0, k - 1);
        const double oxp = wv::shfl(ox(j), k - 1), oyp = wv::shfl(oy(j), k - 1), ixp = wv::shfl(ix2(j), k - 1), iyp = wv::shfl(iy2(j), k - 1);
        if (k >= 2) { cx += vp * 2 * (X[0] - oxp) * ixp; cy += vp * 2 * (X[1] - oyp) * iyp; }
      }
    }
    if (xnode) {
#pragma unroll
      for (int i = 0; i < NX; ++i) out[r_dyn + NX * (k - 1) + i] = (-lam[i] + (i == 0 ? cx : i == 1 ? cy : 0.0)) / os;
    }
    if (k == 0) {   // stationarity wrt the pinned X_0
      double a02 = -T * X[3] * sp, a03 = T * cp, a12 = T * X[3] * cp, a13 = T * sp, a23 = T * td * il;
      if (RK4) {
        KinRkJac Jo{}; KinRkHess Ho{}; kin_rk4_derivs(X, U, T, il, sp, cp, td, sec2, ln, Jo, Ho);
        a02 = Jo.a02; a03 = Jo.a03; a12 = Jo.a12; a13 = Jo.a13; a23 = Jo.a23;
      }
      const double At[NX] = {ln[0], ln[1], a02 * ln[0] + a12 * ln[1] + ln[2], a03 * ln[0] + a13 * ln[1] + a23 * ln[2] + ln[3]};
#pragma unroll
      for (int i = 0; i < NX; ++i) out[i] = -2 * c.Q[i] * (X[i] - xs[i]) - At[i] / os;
    }
    if (rr_on) out[r_rate + (k - 1)] = -item_y(qR, iR) / os;
    if (isnode) {
      const int row = (c.obs_mode == MPCB_OBS_KEEPOUT || GEN) ? k : k - 1;
#pragma unroll
      for (int j = 0; j < NOBS; ++j) if (j < nobs && row >= 0 && row <= last_row) out[r_obs + row * nobs + j] = ro_on[j] ? -vO[j] / os : 0.0;
    }
  }
}
#undef ox
#undef oy
#undef ix2
#undef iy2

