// mpcb_kernel_dyn.h — the batched MPC solve for the 6-state dynamic bicycle with the nonlinear tyre model.
//
// What it replaces: the IPOPT solve of the NLP of CasaDi_MPC_Optimize_Multishoot/MPC_CBF_optimize_dyn.py
// (model :156-170, cost :189-225, dynamics rows :226-228, rate rows on both controls :229-231, obstacle rows at
// all N+1 nodes :238-243, boxes :85-110).  Same algorithm and wave mapping as mpcb_kernel.h (lane k = node k for
// everything lane-parallel); the differences are sizes and the model:
//   * nx = 6, augmented state 8: the 8x8 STATE block of the stage is one entry per lane; the two control rows/columns
//     of the 10x10 stage block have the fixed patterns (0,0,0,0,b4,b5,1,0) and (0,0,0,T,0,0,0,1) in [A B] and are
//     computed with explicit formulas by the lanes of rows 6,7 (whose own rows of [A B] are zero); the affine part of
//     the recursion rides in column 6.  Same 2 LDS round trips per stage as the kinematic kernel;
//   * closed-form first and second derivatives of the tyre model (checked against the oracle's forward-mode AD);
//   * boxes on y, vx, vy; rate rows on steering AND acceleration; obstacle rows of the form h >= obs_hmin.
// The reference's own bounds lists for this model are mis-aligned with its rows; the rows here are
// the aligned ones.  The obstacle row is written h >= hmin (hmin = 1) instead of the reference's sqrt(h) >= 1: same
// feasible set and same KKT points, no NaN inside the ellipse.
#pragma once

#include "mpcb_kernel.h"

namespace mpcbk {

enum DynEnt {
  DE_ZERO = 0, DE_ONE,                 // (the entries of [A B], the defect and the step length live in the fw rows only)
  DE_G0, DE_G1, DE_G2, DE_G3, DE_G4, DE_G5, DE_G6, DE_G7, DE_G8, DE_G9,
  DE_H00, DE_H01, DE_H11, DE_H22, DE_H23, DE_H24, DE_H33, DE_H34, DE_H35, DE_H44, DE_H45, DE_H55,
  DE_H38, DE_H48, DE_H58, DE_H88, DE_H99, DE_H66, DE_H77, DE_H68, DE_H79,
  DYN_NENT
};

// LDS layout (doubles):
//   Pst [N+1][66]  P_k (8x8) + 2 pad   pst [N+1][10]  p_k (8), slot 8 = permanent 0.0, slot 9 pad
//   fw  [N+1][DFWS] per-stage numbers, [node][slot]: K (2x8), kff (2), the 14 A entries, b4 b5, d (6), 2 pad slots, 1.0, 0.0, T_k, a spare.  Read by the forward roll-out (whole row) AND by the sweep (each lane its own slots of [A B | d]): the
//                   single copy keeps the N = 40 instance at 53 KB, three workgroups per CU instead of two
constexpr int DPST = 66 /* P_k 8x8 + 2 pad slots for the stores of the lanes below the diagonal */, DPSS = 10, DFWS = 46, DFW_KFF = 16, DFW_A = 18, DFW_B = 32, DFW_D = 34, DFW_PAD = 40, DFW_ONE = 42, DFW_ZERO = 43, DFW_T = 44, DWSZ = 64, DWU = 16 + 16;
// constant block (as in the kin kernel): cost weights, reference state and last control, read from LDS by the node-parallel phases
constexpr int DCS_WQ = 0, DCS_WR = 6, DCS_WDR = 8, DCS_Q = 10, DCS_R = 16, DCS_DR = 18, DCS_UL = 20, DCS_XS = 22, DCS_ACC = 28, DCSZ = 30;   // DCS_ACC: state of the acceptable-point test (objective at the previous check, iterations in a row)
// per-node cost table of the RESTO instantiation, [row][N+2] (see CostRow in mpcb_kernel.h)
enum DynCostRow { DCT_WQ = 0, DCT_XR = 6, DCT_WR = 12, DCT_UR = 14, DCT_QQ = 16, DCT_RR = 22, DCT_WDR = 24, DCT_DRR = 26, DCT_ROWS = 28 };
struct LayoutDyn { int ld, ent, Pst, pst, fw, W, Wu, cst, filt, zbuf, ct, obl, total; };
MPCB_HD int obs_capacity_dyn(int n) { return n <= 1 ? 1 : n <= 3 ? 3 : n <= 5 ? 5 : 8; }
MPCB_HD LayoutDyn layout_dyn(int N, bool resto = false, int nobl = 0) {
  LayoutDyn L;
  const int N1 = N + 1, NA = 8;
  L.ld = N1 | 1;
  int o = 0;
  L.Pst = o; o += N1 * DPST;
  L.pst = o; o += N1 * DPSS;
  L.fw = o; o += N1 * DFWS;
  L.W = o; o += DWSZ;
  L.Wu = o; o += DWU;
  L.cst = o; o += DCSZ;
  L.filt = o; o += 2 * FILTER_MAX;
  L.ent = o; o += DYN_NENT * L.ld;
  L.zbuf = L.Pst;
  L.ct = o; if (resto) o += DCT_ROWS * (N + 2);
  L.obl = o; o += 4 * nobl * (N + 2);
  L.total = o;
  return L;
}

// everything the model needs at one node: trig of heading and steering, slip angles, tyre forces and their derivatives
struct DynEval { double sp, cp, sd, cd, ivx, pf, pr, Ff, Ff1, Ff2, Fr, Fr1, Fr2; };
MPCB_DEV void dyn_eval(const mpcb_config& c, const double* X, const double* U, DynEval& e) {
  sincos_b(X[2], e.sp, e.cp);
  sincos_b(U[0], e.sd, e.cd);
  e.ivx = wv::rcp(X[3]);
  e.pf = X[4] + c.veh_lf * X[5];
  e.pr = X[4] - c.veh_lr * X[5];
  const double af = U[0] - e.pf * e.ivx, ar = -e.pr * e.ivx;
  const double cF = c.Fymax_f * 2.0 * c.aopt_f, cR = c.Fymax_r * 2.0 * c.aopt_r;
  const double a2f = c.aopt_f * c.aopt_f, a2r = c.aopt_r * c.aopt_r;
  const double qf = wv::rcp(a2f + af * af), qr = wv::rcp(a2r + ar * ar);
  e.Ff = -cF * af * qf;                                   // F_cf = -C_f alpha_f, C_f = cF / (aopt^2 + alpha^2)   dyn.py:158-161
  e.Ff1 = -cF * (a2f - af * af) * qf * qf;
  e.Ff2 = 2.0 * cF * af * (3.0 * a2f - af * af) * qf * qf * qf;
  e.Fr = -cR * ar * qr;
  e.Fr1 = -cR * (a2r - ar * ar) * qr * qr;
  e.Fr2 = 2.0 * cR * ar * (3.0 * a2r - ar * ar) * qr * qr * qr;
}
// F = X + T f(X, U)                                                                          dyn.py:165-170,227
MPCB_DEV void dyn_F(const mpcb_config& c, const double T, const double* X, const double* U, const DynEval& e, double* F) {
  const double vx = X[3], vy = X[4], r = X[5];
  F[0] = X[0] + T * (vx * e.cp - vy * e.sp);
  F[1] = X[1] + T * (vx * e.sp + vy * e.cp);
  F[2] = X[2] + T * r;
  F[3] = X[3] + T * (U[1] + r * vy);
  F[4] = X[4] + T * (-r * vx + (2.0 / c.veh_m) * (e.Ff * e.cd + e.Fr));
  F[5] = X[5] + T * ((2.0 / c.veh_Iz) * (c.veh_lf * e.Ff - c.veh_lr * e.Fr));
}
struct DynJac { double a02, a03, a04, a12, a13, a14, a34, a35, a43, a44, a45, a53, a54, a55, b4, b5; };
MPCB_DEV void dyn_jac(const mpcb_config& c, const double T, const double* X, const DynEval& e, DynJac& J) {
  const double vx = X[3], vy = X[4], r = X[5], k4 = 2.0 / c.veh_m, k5 = 2.0 / c.veh_Iz, lf = c.veh_lf, lr = c.veh_lr;
  const double i2 = e.ivx * e.ivx;
  const double afx = e.pf * i2, afy = -e.ivx, afr = -lf * e.ivx, arx = e.pr * i2, ary = -e.ivx, arr = lr * e.ivx;
  J.a02 = T * (-vx * e.sp - vy * e.cp); J.a03 = T * e.cp; J.a04 = -T * e.sp;
  J.a12 = T * (vx * e.cp - vy * e.sp); J.a13 = T * e.sp; J.a14 = T * e.cp;
  J.a34 = T * r; J.a35 = T * vy;
  J.a43 = T * (-r + k4 * (e.Ff1 * afx * e.cd + e.Fr1 * arx));
  J.a44 = 1.0 + T * k4 * (e.Ff1 * afy * e.cd + e.Fr1 * ary);
  J.a45 = T * (-vx + k4 * (e.Ff1 * afr * e.cd + e.Fr1 * arr));
  J.b4 = T * k4 * (e.Ff1 * e.cd - e.Ff * e.sd);
  J.a53 = T * k5 * (lf * e.Ff1 * afx - lr * e.Fr1 * arx);
  J.a54 = T * k5 * (lf * e.Ff1 * afy - lr * e.Fr1 * ary);
  J.a55 = 1.0 + T * k5 * (lf * e.Ff1 * afr - lr * e.Fr1 * arr);
  J.b5 = T * k5 * lf * e.Ff1;
}
// sum_a lam_a T d2 f_a: entries over (phi=2, vx=3, vy=4, r=5, delta=8 in stage numbering)
struct DynHess { double h22, h23, h24, h33, h34, h35, h44, h45, h55, h38, h48, h58, h88; };
MPCB_DEV void dyn_hess(const mpcb_config& c, const double T, const double* X, const DynEval& e, const double* l, DynHess& H) {
  const double vx = X[3], vy = X[4], k4 = 2.0 / c.veh_m, k5 = 2.0 / c.veh_Iz, lf = c.veh_lf, lr = c.veh_lr;
  const double i2 = e.ivx * e.ivx, i3 = i2 * e.ivx;
  const double af[3] = {e.pf * i2, -e.ivx, -lf * e.ivx}, ar[3] = {e.pr * i2, -e.ivx, lr * e.ivx};   // d alpha / d(vx, vy, r)
  // second derivatives of alpha: (vx,vx), (vx,vy), (vx,r); all others zero
  const double afxx = -2.0 * e.pf * i3, afxy = i2, afxr = lf * i2, arxx = -2.0 * e.pr * i3, arxy = i2, arxr = -lr * i2;
  auto G = [&](int z, int w, double azw_f, double azw_r, double extra4) {
    const double gf = e.Ff2 * af[z] * af[w] + e.Ff1 * azw_f, gr = e.Fr2 * ar[z] * ar[w] + e.Fr1 * azw_r;
    return T * (l[4] * (k4 * (e.cd * gf + gr) + extra4) + l[5] * k5 * (lf * gf - lr * gr));
  };
  H.h22 = T * (l[0] * (-vx * e.cp + vy * e.sp) + l[1] * (-vx * e.sp - vy * e.cp));
  H.h23 = T * (-l[0] * e.sp + l[1] * e.cp);
  H.h24 = T * (-l[0] * e.cp - l[1] * e.sp);
  H.h33 = G(0, 0, afxx, arxx, 0.0);
  H.h34 = G(0, 1, afxy, arxy, 0.0);
  H.h35 = G(0, 2, afxr, arxr, -1.0);
  H.h44 = G(1, 1, 0.0, 0.0, 0.0);
  H.h45 = G(1, 2, 0.0, 0.0, 0.0) + T * l[3];
  H.h55 = G(2, 2, 0.0, 0.0, 0.0);
  const double cz = T * (l[4] * k4 * (-e.sd * e.Ff1 + e.cd * e.Ff2) + l[5] * k5 * lf * e.Ff2);
  H.h38 = cz * af[0]; H.h48 = cz * af[1]; H.h58 = cz * af[2];
  H.h88 = T * (l[4] * k4 * (e.Ff2 * e.cd - 2.0 * e.Ff1 * e.sd - e.Ff * e.cd) + l[5] * k5 * lf * e.Ff2);
}

}  // namespace mpcbk

// RESTO: the instantiation of the restoration pass (see mpcb_solve_kin)
template <int NOBS, bool RESTO = false>
MPCB_DEVFN void mpcb_solve_dyn(const MpcbKArgs& a_in, const int b, double* lds, const int pass) {   // pass: MPCB_PASS_* (see MpcbKArgs::pass; a parameter of its own because one launch can run two passes of an instance)
  // every kernel argument is read through a pointer the optimiser cannot see through (wv::late_args): the compiler then loads a field
  // where the code needs it instead of preloading the whole 800-byte argument block into scalar registers at entry, most of which it
  // has to spill into VGPR lanes again (kin<3>: 806 -> 582 v_readlane of SGPR reloads)
  const MpcbKArgs& a = *wv::late_args(a_in);
  using namespace mpcbk;
  constexpr int NX = 6, NA = 8, NW = 10, NOB = NOBS > 0 ? NOBS : 1, NEL = RESTO ? NOB : 1;
  const mpcb_config& c = a.cfg;
  // the eight vehicle / tyre constants of the model are loaded from the kernel arguments at every model evaluation instead of living in
  // 16+ scalar registers for the whole solve (wv::late_args): the dyn kernels spill SGPRs into VGPR lanes by the hundred
  auto MC = [&]() -> const mpcb_config& { return wv::late_args(a)->cfg; };
  const int N = c.N, lane = wv::opaque(wv::lane()), k = lane;     // (opaque: see mpcb_kernel.h)
  const int nz = a.nz, nobs = c.n_obs;
  if (RESTO && a.status[(size_t)b * a.st_stride] != MPCB_ST_NEEDS_RESTO) return;     // wave-uniform: this instance is done
  if (!RESTO && pass == MPCB_PASS_SECOND) {          // second start: only instances whose first attempt (restoration included) did not succeed
    const int st1 = a.status[(size_t)b * a.st_stride];
    if (st1 == MPCB_ST_SOLVED || st1 == MPCB_ST_ACCEPTABLE || st1 == MPCB_ST_INFEASIBLE_X0) return;
    bool fin = true;                                    // non-finite inputs: the first attempt's verdict (at iteration 0) stands
    for (int i = 0; i < 6; ++i) fin = fin && isfinite(a.x0[(size_t)b * 6 + i]) && isfinite(a.xs[(size_t)b * 6 + i]);
    if (!fin) return;
  }
  // Which start does this solve run from?  First attempt: the caller's z0, with X rolled out from x0 (cfg.init_rollout).  Second
  // attempt (cfg.second_start, only after a roll-out start; mpcb_api.hip launches its passes after the first attempt's): the
  // reference's own first-step start z = 0 (main_cbf_kin_c_sim.py:47-50), no roll-out.  A restoration pass continues whichever
  // attempt handed over (WK_START).
  const bool zeros_start = RESTO ? (a.work && a.work[(size_t)b * mpcbk::WK_SIZE + mpcbk::WK_START] != 0.0) : pass == MPCB_PASS_SECOND;
  const bool rollout = c.init_rollout && !zeros_start;
  constexpr bool OBL = NOBS > 3;                  // obstacle constants in LDS instead of registers (see mpcb_solve_kin)
  const LayoutDyn L = layout_dyn(N, RESTO, obs_in_lds(NOBS));
  const int ld = L.ld;
  double* ent = lds + L.ent;
  double T_ = wv::uni(c.T);                                        // step length of this lane's stage (time grid or cfg.T); see mpcb_solve_kin
  if (a.tgrid) T_ = a.tgrid[k < N ? k : N - 1];
  const double T = T_;

  const bool isnode = k <= N, hasu = k < N, xnode = k >= 1 && k <= N, xcost = k >= 1 && k < N;
  const double* gx0 = a.x0 + (size_t)b * NX;
  const double* gxs = a.xs + (size_t)b * NX;
  double xs[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) xs[i] = gxs[i];

  const int last_row = c.obs_terminal ? N : N - 1;
  bool obs_node; int ostep;
  if (c.obs_mode == MPCB_OBS_KEEPOUT) { obs_node = k <= last_row; ostep = k; }
  else { obs_node = k >= 1 && k - 1 <= last_row; ostep = k - 1; }
  obs_node = obs_node && isnode;
  double ox_[OBL ? 1 : NOB], oy_[OBL ? 1 : NOB], ix2_[OBL ? 1 : NOB], iy2_[OBL ? 1 : NOB];
  double* obl = lds + L.obl;
  auto OC = [&](int j, int q) -> double& {
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

  double* zbuf = lds + L.zbuf;
  for (int i = lane; i < nz; i += 64) zbuf[i] = (a.z0 && !zeros_start) ? a.z0[(size_t)b * nz + i] : 0.0;
  wv::sync();
  double X[NX], U[NU], lam[NX];
#pragma unroll
  for (int i = 0; i < NU; ++i) U[i] = hasu ? zbuf[NU * k + i] : 0.0;
#pragma unroll
  for (int i = 0; i < NX; ++i) { X[i] = isnode ? zbuf[NU * N + NX * k + i] : 0.0; lam[i] = 0.0; }
  if (zeros_start && isnode) X[3] = gx0[3];       // second start: z = 0 except the longitudinal speed (the tyre model divides by vx, dyn.py:156-157)
  wv::sync();

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
        if (k + 1 < N && (k + 1 > 0 || c.du0_cost)) gu -= 2 * c.DR[i] * (Un[i] - U[i]);
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
    for (int i = 0; i < NX; ++i) { cst[DCS_WQ + i] = os * 2 * c.Q[i]; cst[DCS_Q + i] = c.Q[i]; cst[DCS_XS + i] = xs[i]; }
#pragma unroll
    for (int i = 0; i < NU; ++i) { cst[DCS_WR + i] = os * 2 * c.R[i]; cst[DCS_WDR + i] = os * 2 * c.DR[i]; cst[DCS_R + i] = c.R[i]; cst[DCS_DR + i] = c.DR[i]; cst[DCS_UL + i] = c.u_last[i]; }
    cst[DCS_ACC] = 1e300; cst[DCS_ACC + 1] = 0.0;
  }
  wv::sync();
  // objective of the running phase: uniform constants (first pass) or the per-node table of the RESTO instantiation
  double* ct = lds + L.ct;
  const int cts = N + 2, ctl = lane <= N ? lane : N + 1;     // row length of the cost table; lanes beyond the last node share a dummy column
  double osc = os;
  bool rs = false;
  auto cWQ = [&](int i) { return RESTO ? ct[(DCT_WQ + i) * cts + ctl] : cst[DCS_WQ + i]; };
  auto cXS = [&](int i) { return RESTO ? ct[(DCT_XR + i) * cts + ctl] : cst[DCS_XS + i]; };
  auto cQQ = [&](int i) { return RESTO ? ct[(DCT_QQ + i) * cts + ctl] : cst[DCS_Q + i]; };
  auto cWR = [&](int i) { return RESTO ? ct[(DCT_WR + i) * cts + ctl] : cst[DCS_WR + i]; };
  auto cRR = [&](int i) { return RESTO ? ct[(DCT_RR + i) * cts + ctl] : cst[DCS_R + i]; };
  auto cUR = [&](int i) { return RESTO ? ct[(DCT_UR + i) * cts + ctl] : 0.0; };
  auto cWDR = [&](int i) { return RESTO ? ct[(DCT_WDR + i) * cts + ctl] : cst[DCS_WDR + i]; };
  auto cDRR = [&](int i) { return RESTO ? ct[(DCT_DRR + i) * cts + ctl] : cst[DCS_DR + i]; };
  auto write_main_cost = [&]() {         // dyn.py:189-225
    if (RESTO) {
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        ct[(DCT_WQ + i) * cts + ctl] = hasu ? os * 2 * c.Q[i] : 0.0; ct[(DCT_QQ + i) * cts + ctl] = hasu ? c.Q[i] : 0.0;
        ct[(DCT_XR + i) * cts + ctl] = xs[i];
      }
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        ct[(DCT_WR + i) * cts + ctl] = hasu ? os * 2 * c.R[i] : 0.0; ct[(DCT_RR + i) * cts + ctl] = hasu ? c.R[i] : 0.0;
        ct[(DCT_UR + i) * cts + ctl] = 0.0;
        ct[(DCT_WDR + i) * cts + ctl] = os * 2 * c.DR[i]; ct[(DCT_DRR + i) * cts + ctl] = c.DR[i];
      }
    }
  };
  auto write_resto_cost = [&](double zeta, bool fresh, const double* Xc, const double* Uc) {
    if (RESTO) {
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        if (fresh) ct[(DCT_XR + i) * cts + ctl] = Xc[i];
        const double d = 1.0 / fmax(1.0, fabs(ct[(DCT_XR + i) * cts + ctl]));
        const double q = xnode ? 0.5 * zeta * d * d : 0.0;
        ct[(DCT_QQ + i) * cts + ctl] = q; ct[(DCT_WQ + i) * cts + ctl] = 2 * q;
      }
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        if (fresh) ct[(DCT_UR + i) * cts + ctl] = Uc[i];
        const double d = 1.0 / fmax(1.0, fabs(ct[(DCT_UR + i) * cts + ctl]));
        const double q = hasu ? 0.5 * zeta * d * d : 0.0;
        ct[(DCT_RR + i) * cts + ctl] = q; ct[(DCT_WR + i) * cts + ctl] = 2 * q;
        ct[(DCT_WDR + i) * cts + ctl] = 0.0; ct[(DCT_DRR + i) * cts + ctl] = 0.0;
      }
    }
  };
  write_main_cost();
  const bool xq = RESTO ? xnode : xcost;
  const bool xobj = RESTO ? isnode : hasu;
  if (k == 0) {
#pragma unroll
    for (int i = 0; i < NX; ++i) X[i] = gx0[i];
  }

  int status = MPCB_ST_MAXITER, iters = 0;
  {
    bool bad = false;
    if (k == 0) {
#pragma unroll
      for (int i = 0; i < NX; ++i) bad = bad || (X[i] < c.x_lo[i] - 1e-8) || (X[i] > c.x_hi[i] + 1e-8);
      if (obs_node) {
#pragma unroll
        for (int j = 0; j < NOBS; ++j) if (j < nobs) bad = bad || (hval(j, X[0], X[1]) < c.obs_hmin - 1e-8);
      }
      bad = bad || !(X[3] > 0);                         // the tyre model divides by vx
    }
    if (wv::any(bad)) status = MPCB_ST_INFEASIBLE_X0;
  }

  const Bnd qU0 = mk_bnd(c.u_lo[0], c.u_hi[0], c.bound_relax), qU1 = mk_bnd(c.u_lo[1], c.u_hi[1], c.bound_relax);
  const Bnd qY = mk_bnd(c.x_lo[1], c.x_hi[1], c.bound_relax), qVx = mk_bnd(c.x_lo[3], c.x_hi[3], c.bound_relax);
  const Bnd qVy = mk_bnd(c.x_lo[4], c.x_hi[4], c.bound_relax);
  const double rsc = a.tgrid ? wv::shfl(T, k - 1) / c.T : 1.0;    // rate rows: bounds scale with the time between the two controls
  const Bnd qR0 = mk_bnd_lane(c.du_lo[0] * rsc, c.du_hi[0] * rsc, c.bound_relax), qR1 = mk_bnd_lane(c.du_lo[1] * rsc, c.du_hi[1] * rsc, c.bound_relax);
  const Bnd qO = mk_bnd(c.obs_hmin, 1e308, c.bound_relax);
  const bool bu0_on = hasu && qU0.on, bu1_on = hasu && qU1.on;
  const bool by_on = xnode && qY.on, bvx_on = xnode && qVx.on, bvy_on = xnode && qVy.on;
  const bool r0_on = xcost && qR0.on, r1_on = xcost && qR1.on;
  const bool ro_node = xnode && obs_node;
  const bool ducost = hasu && (k > 0 || c.du0_cost);     // (U_k - U_{k-1})' DR (.) present in stage k     dyn.py:221-224

  if (!RESTO && rollout) {
    auto roll_out = [&]() {
      U[0] = hasu ? fmin(fmax(U[0], c.u_lo[0]), c.u_hi[0]) : 0.0;
      U[1] = hasu ? fmin(fmax(U[1], c.u_lo[1]), c.u_hi[1]) : 0.0;
#pragma clang loop unroll(disable)
      for (int s = 0; s < N; ++s) {
        double Xs[NX]; bool okx = X[3] > 1e-3;
#pragma unroll
        for (int i = 0; i < NX; ++i) Xs[i] = X[i];
        if (!okx) Xs[3] = 1e-3;
        DynEval e; dyn_eval(MC(), Xs, U, e);
        double F[NX]; dyn_F(MC(), T, Xs, U, e, F);
#pragma unroll
        for (int i = 0; i < NX; ++i) { const double n = wv::bcast(F[i], s); if (k == s + 1) X[i] = n; }
      }
    };
    roll_out();
    // cfg.start_steer: a cold start whose straight roll-out passes an obstacle row closer than h - obs_hmin < 1 is rolled out with a
    // slight constant turn instead (see mpcb_kernel.h; oracle: Solver::init)
    if (NOBS > 0 && !a.z0 && nobs > 0 && MC().start_steer > 0.0) {       // wave-uniform
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
        U[0] = hasu ? sgn * MC().start_steer : 0.0;
        roll_out();
      }
    }
  }

  Item iU0{1, 1, 0, 0}, iU1{1, 1, 0, 0}, iY{1, 1, 0, 0}, iVx{1, 1, 0, 0}, iVy{1, 1, 0, 0}, iR0{1, 1, 0, 0}, iR1{1, 1, 0, 0};
  if (!RESTO) {    // (the restoration pass starts from an interior iterate of the first pass)
    if (bu0_on) U[0] = push_in(qU0, U[0], c.bound_push, c.bound_frac);
    if (bu1_on) U[1] = push_in(qU1, U[1], c.bound_push, c.bound_frac);
    if (by_on) X[1] = push_in(qY, X[1], c.bound_push, c.bound_frac);
    if (bvx_on) X[3] = push_in(qVx, X[3], c.bound_push, c.bound_frac);
    if (bvy_on) X[4] = push_in(qVy, X[4], c.bound_push, c.bound_frac);
  }
  double Up0 = wv::shfl(U[0], k - 1), Up1 = wv::shfl(U[1], k - 1);
  double sR0 = 0, sR1 = 0, rR0 = 0, rR1 = 0;
  if (r0_on) sR0 = push_in(qR0, U[0] - Up0, c.bound_push, c.bound_frac);
  if (r1_on) sR1 = push_in(qR1, U[1] - Up1, c.bound_push, c.bound_frac);
  double sO[NOB], vO[NOB], iO[NOB], rO[NOB], gO0[NOB], gO1[NOB];
  bool ro_on[NOB];
  double eP[NEL], eN[NEL], vP[NEL], vN[NEL];      // restoration phase: elastic variables of the obstacle rows and their duals
#pragma unroll
  for (int j = 0; j < NEL; ++j) { eP[j] = 0; eN[j] = 0; vP[j] = 0; vN[j] = 0; }
#pragma unroll
  for (int j = 0; j < NOBS; ++j) {
    ro_on[j] = ro_node && j < nobs;
    sO[j] = ro_on[j] ? push_in(qO, hval(j, X[0], X[1]), c.bound_push, c.bound_frac) : 1.0 + qO.L;
    vO[j] = 1.0; rO[j] = 0; gO0[j] = 0; gO1[j] = 0; iO[j] = 0;
  }
  auto recips = [&]() {
    if (bu0_on) item_recip(qU0, U[0], iU0);
    if (bu1_on) item_recip(qU1, U[1], iU1);
    if (by_on) item_recip(qY, X[1], iY);
    if (bvx_on) item_recip(qVx, X[3], iVx);
    if (bvy_on) item_recip(qVy, X[4], iVy);
    if (r0_on) item_recip(qR0, sR0, iR0);
    if (r1_on) item_recip(qR1, sR1, iR1);
#pragma unroll
    for (int j = 0; j < NOBS; ++j) if (ro_on[j]) iO[j] = wv::rcp(sO[j] - qO.L);
  };

  double mu = c.mu_init, tau = fmax(TAU_MIN, 1.0 - mu);
  double dfc[NX] = {0, 0, 0, 0, 0, 0};
  double theta = 0, fval = 0, logsum = 0;

  auto eval_lane = [&](const double* Xa, const double* Ua, double sR0a, double sR1a, const double* sOa, const double* pa, const double* na, const DynEval& e,
                       double* dfa, double& rR0a, double& rR1a, double* rOa, double& up0, double& up1, double& th, double& fl, double& prod) {
    bool ok = true;
    double Ft[NX]; dyn_F(MC(), T, Xa, Ua, e, Ft);
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
    if (bvx_on) bar(qVx, Xa[3]);
    if (bvy_on) bar(qVy, Xa[4]);
    rR0a = 0; rR1a = 0;
    if (r0_on) { bar(qR0, sR0a); rR0a = (Ua[0] - up0) - sR0a; th += fabs(rR0a); }
    if (r1_on) { bar(qR1, sR1a); rR1a = (Ua[1] - up1) - sR1a; th += fabs(rR1a); }
#pragma unroll
    for (int j = 0; j < NOBS; ++j) {
      rOa[j] = 0;
      if (ro_on[j]) {
        bar(qO, sOa[j]); rOa[j] = hval(j, Xa[0], Xa[1]) - sOa[j];
        if (RESTO && rs) {        // elastic row: residual of c - s - p + n, cost rho (p + n), barrier on p and n
          const double pj = pa[j], nj = na[j];
          ok = ok && (pj > 0) && (nj > 0); prod *= pj * nj;
          rOa[j] -= pj - nj; fl += RS_RHO * (pj + nj);
        }
        th += fabs(rOa[j]);
      }
    }
    if (isnode && !(Xa[3] > 0)) ok = false;              // vx must stay positive where the model is evaluated
    if (xobj) {
#pragma unroll
      for (int i = 0; i < NX; ++i) { const double d = Xa[i] - cXS(i); fl += cQQ(i) * d * d; }
    }
    if (hasu) {
      const double e0 = Ua[0] - cUR(0), e1 = Ua[1] - cUR(1);
      fl += cRR(0) * e0 * e0 + cRR(1) * e1 * e1;
      if (ducost) {
        const double d0 = Ua[0] - (k ? up0 : cst[DCS_UL]), d1 = Ua[1] - (k ? up1 : cst[DCS_UL + 1]);
        fl += cDRR(0) * d0 * d0 + cDRR(1) * d1 * d1;
      }
    }
    return ok;
  };

  double n_lam, n_v;
  {
    double cnt = 0;
    auto two = [&](const Bnd& q) { return (q.hasL ? 1.0 : 0.0) + (q.hasU ? 1.0 : 0.0); };
    if (bu0_on) cnt += two(qU0);
    if (bu1_on) cnt += two(qU1);
    if (by_on) cnt += two(qY);
    if (bvx_on) cnt += two(qVx);
    if (bvy_on) cnt += two(qVy);
    if (r0_on) cnt += two(qR0);
    if (r1_on) cnt += two(qR1);
#pragma unroll
    for (int j = 0; j < NOBS; ++j) if (ro_on[j]) cnt += 1.0;
    n_v = wv::uni(wv::sum(cnt));
    n_lam = (double)(NX * N);
  }

  // ----- Riccati lane constants: lane = entry (i,j) of the 8x8 state block [X, U_prev]; the control rows ride along ---------------------------------
  // slot of entry (r, col) of [A B] inside the stage's fw row; a[] = {a02,a03,a04, a12,a13,a14, a34,a35, a43,a44,a45, a53,a54,a55}
  auto slotAB = [&](int r, int col) -> int {
    if (r < NX) {
      if (col < NX) {
        if (r == 0) return col == 0 ? DFW_ONE : col == 2 ? DFW_A + 0 : col == 3 ? DFW_A + 1 : col == 4 ? DFW_A + 2 : DFW_ZERO;
        if (r == 1) return col == 1 ? DFW_ONE : col == 2 ? DFW_A + 3 : col == 3 ? DFW_A + 4 : col == 4 ? DFW_A + 5 : DFW_ZERO;
        if (r == 2) return col == 2 ? DFW_ONE : col == 5 ? DFW_T : DFW_ZERO;
        if (r == 3) return col == 3 ? DFW_ONE : col == 4 ? DFW_A + 6 : col == 5 ? DFW_A + 7 : DFW_ZERO;
        if (r == 4) return col == 3 ? DFW_A + 8 : col == 4 ? DFW_A + 9 : col == 5 ? DFW_A + 10 : DFW_ZERO;
        return col == 3 ? DFW_A + 11 : col == 4 ? DFW_A + 12 : col == 5 ? DFW_A + 13 : DFW_ZERO;
      }
      if (col == 8) return r == 4 ? DFW_B : r == 5 ? DFW_B + 1 : DFW_ZERO;
      if (col == 9) return r == 3 ? DFW_T : DFW_ZERO;
      return DFW_ZERO;
    }
    return (col == r + 2) ? DFW_ONE : DFW_ZERO;
  };
  auto slotH = [&](int r, int col) -> int {
    const int lo = r < col ? r : col, hi = r < col ? col : r;
    const int key = lo * 10 + hi;
    switch (key) {
      case 0: return DE_H00; case 1: return DE_H01; case 11: return DE_H11;
      case 22: return DE_H22; case 23: return DE_H23; case 24: return DE_H24;
      case 33: return DE_H33; case 34: return DE_H34; case 35: return DE_H35;
      case 44: return DE_H44; case 45: return DE_H45; case 55: return DE_H55;
      case 38: return DE_H38; case 48: return DE_H48; case 58: return DE_H58;
      case 88: return DE_H88; case 99: return DE_H99; case 66: return DE_H66; case 77: return DE_H77;
      case 68: return DE_H68; case 79: return DE_H79;
      default: return DE_ZERO;
    }
  };
  // Block form of the stage:  state block (8x8, one entry per lane) + control rows/columns (2x8 + 2x2) computed by the
  // lanes of rows 6,7 (U_prev rows, whose own [A B] rows are zero) with explicit formulas, because the control columns of
  // [A B] are (0,0,0,0,b4,b5,1,0) and (0,0,0,T,0,0,0,1).  Column 6 (U_prev, zero column of [A B]) carries the affine part.
  const int ei = lane >> 3, ej = lane & 7;
  const bool aff = (ej == 6);
  const int cU = ei & 1;                         // control index of the rows 6,7 lanes (and of the odd / even W_u pattern)
  int sABj[NX], sABi[NX];
#pragma unroll
  for (int r = 0; r < NX; ++r) {
    sABj[r] = aff ? DFW_D + r : slotAB(r, ej);
    sABi[r] = slotAB(r, ei);
  }
  const int sHij = slotH(ei, ej) * ld;
  const int sStart = (aff ? DE_G0 + ei : slotH(ei, ej)) * ld;
  const int sStartU = (ei >= 6 ? (aff ? DE_G8 + cU : slotH(8 + cU, ej)) : DE_ZERO) * ld;     // start of M_ux(c, j) / m_u(c)
  const int sHU = (ei >= 6 ? slotH(8 + cU, ej) : DE_ZERO) * ld;                               // matrix value H(8+c, j)
  const int sStartUU = ((ei >= 6 && ej < 2) ? slotH(8 + cU, 8 + ej) : DE_ZERO) * ld;          // H(8+c, 8+c')
  const int pvOff = aff ? ei : 8;
  const int wuOff = ej < 2 ? ej * NA + ei : 16 + (lane & 15);
  const int psOff = aff ? ei : 9;
  // gains: lanes (0,j) store K0j, lanes (1,j) store K1j; lanes 62 / 63 store the feed-forward terms; pad slots elsewhere
  const int kOff = (ei == 0) ? ej : (ei == 1) ? NA + ej : DFW_PAD + (lane & 1);
  const int kfOff = (lane == 62) ? DFW_KFF : (lane == 63) ? DFW_KFF + 1 : DFW_PAD + (lane & 1);
  const bool kRow1 = (ei == 1), kfLane1 = (lane == 63);
  const int sOff = lane >= 48 ? 2 * ej + cU : 16 + (lane & 15);      // staging of the sweep's tail: lanes of rows 6,7 at 2j + c, pads elsewhere
  const int pOff = ei <= ej ? ei * NA + ej : NA * NA, pOffT = ei <= ej ? ej * NA + ei : NA * NA + 1;
  if (isnode) { ent[DE_ZERO * ld + k] = 0.0; ent[DE_ONE * ld + k] = 1.0; }

  double* Pst = lds + L.Pst; double* pst = lds + L.pst; double* fw = lds + L.fw;
  double* Wl = lds + L.W; double* WuL = lds + L.Wu; double* filt = lds + L.filt;
  int nfilt = 0;
  double theta_max = 0, theta_min = 0, dw_last = 0.0;
  const double mu_floor = c.tol / (K_EPS + 1.0);
  double err0 = 0, e_dual = 0, e_prim = 0;

  // restoration-phase state (see mpcb_solve_kin)
  double mu_main = 0, tmax_main = 0, tmin_main = 0, fm_theta = 0, fm_phi = 0, th_entry = 0;
  int rit = 0, slow_run = 0, n_rcalls = 0, n_riters = 0;
  double slow_theta0 = 0;
  bool enter = false;
  double n_el = 0;
  if (RESTO) {
    double cnt = 0;
#pragma unroll
    for (int j = 0; j < NOBS; ++j) if (ro_on[j]) cnt += 1.0;
    n_el = wv::uni(wv::sum(cnt));
  }

  if (status != MPCB_ST_INFEASIBLE_X0) {
#pragma unroll
    for (int j = 0; j < NOBS; ++j) if (ro_on[j]) { gO0[j] = 2 * (X[0] - ox(j)) * ix2(j); gO1[j] = 2 * (X[1] - oy(j)) * iy2(j); }
    if (!RESTO) {
      DynEval ev; dyn_eval(MC(), X, U, ev);
      double th, fl, prod;
      eval_lane(X, U, sR0, sR1, sO, eP, eN, ev, dfc, rR0, rR1, rO, Up0, Up1, th, fl, prod);
      double sv[3] = {th, fl, log(prod)};
      wv::reduce<3, 0>(sv, nullptr);
      theta = wv::uni(sv[0]); fval = wv::uni(sv[1]); logsum = wv::uni(sv[2]);
      recips();
      theta_max = wv::uni(1e4 * fmax(1.0, theta)); theta_min = wv::uni(1e-4 * fmax(1.0, theta));
    } else {
      const double* wk = a.work + (size_t)b * WK_SIZE;
      mu = wv::uni(wk[WK_MU]); tau = fmax(TAU_MIN, 1.0 - mu);
      theta_max = wv::uni(wk[WK_THMAX]); theta_min = wv::uni(wk[WK_THMIN]);
      iters = (int)wk[WK_ITERS]; dw_last = wv::uni(wk[WK_DW]);
      enter = true;
    }

    int trips = 0;
#pragma clang loop unroll(disable)
    for (;;) {
      if (++trips > 3 * c.max_iter + 50) { status = MPCB_ST_RESTO_FAILED; break; }
      if (RESTO && enter && n_rcalls >= RS_MAX_CALLS) { status = MPCB_ST_RESTO_FAILED; break; }
      if (RESTO && enter) {
        ++n_rcalls;
        // ----- entry into the restoration phase (oracle: Solver::restoration; comments in mpcb_solve_kin) ----------------------
        enter = false;
        if (lane == 0) { cst[DCS_ACC] = 1e300; cst[DCS_ACC + 1] = 0.0; }      // (the acceptable-point counter starts afresh after a restoration, in both passes alike)
        wv::sync();
        mu_main = mu; tmax_main = theta_max; tmin_main = theta_min;
        rs = false; osc = os;
        DynEval ev; dyn_eval(MC(), X, U, ev);
        Up0 = wv::shfl(U[0], k - 1); Up1 = wv::shfl(U[1], k - 1);
        if (r0_on) sR0 = push_in(qR0, U[0] - Up0, c.bound_push, c.bound_frac);
        if (r1_on) sR1 = push_in(qR1, U[1] - Up1, c.bound_push, c.bound_frac);
#pragma unroll
        for (int j = 0; j < NOBS; ++j) if (ro_on[j]) sO[j] = push_in(qO, hval(j, X[0], X[1]), c.bound_push, c.bound_frac);
        double th, fl, prod;
        eval_lane(X, U, sR0, sR1, sO, eP, eN, ev, dfc, rR0, rR1, rO, Up0, Up1, th, fl, prod);
        double vi = fmax(fabs(rR0), fabs(rR1));
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
        if (bvx_on) centre(qVx, iVx);
        if (bvy_on) centre(qVy, iVy);
        if (r0_on) centre(qR0, iR0);
        if (r1_on) centre(qR1, iR1);
#pragma unroll
        for (int j = 0; j < NOBS; ++j) if (ro_on[j]) vO[j] = fmin(RS_RHO, mu * iO[j]);
#pragma unroll
        for (int i = 0; i < NX; ++i) lam[i] = 0.0;
        nfilt = 0;
        eval_lane(X, U, sR0, sR1, sO, eP, eN, ev, dfc, rR0, rR1, rO, Up0, Up1, th, fl, prod);
        double sw[3] = {th, fl, log(prod)};
        wv::reduce<3, 0>(sw, nullptr);
        theta = wv::uni(sw[0]); fval = wv::uni(sw[1]); logsum = wv::uni(sw[2]);
        theta_max = wv::uni(1e4 * fmax(1.0, theta)); theta_min = wv::uni(1e-4 * fmax(1.0, theta));
        rit = 0; slow_run = 0;
      }
      MPCB_STAMP(t_a);
      // tyre forces and trig at the iterate are recomputed here rather than kept across the line search (13 doubles/lane)
      DynEval ev; dyn_eval(MC(), X, U, ev);
      DynJac J; dyn_jac(MC(), T, X, ev, J);
      recips();
      double ln[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) ln[i] = wv::shfl(lam[i], k + 1);
      {
        double rX[NX] = {0, 0, 0, 0, 0, 0}, rU[NU] = {0, 0};
        const double Un0 = wv::shfl(U[0], k + 1), Un1 = wv::shfl(U[1], k + 1);
        const double yR0 = r0_on ? item_y(qR0, iR0) : 0.0, yR1 = r1_on ? item_y(qR1, iR1) : 0.0;
        const double yR0n = wv::shfl(yR0, k + 1), yR1n = wv::shfl(yR1, k + 1);
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
            rX[2] += J.a02 * ln[0] + J.a12 * ln[1] + ln[2];
            rX[3] += J.a03 * ln[0] + J.a13 * ln[1] + ln[3] + J.a43 * ln[4] + J.a53 * ln[5];
            rX[4] += J.a04 * ln[0] + J.a14 * ln[1] + J.a34 * ln[3] + J.a44 * ln[4] + J.a54 * ln[5];
            rX[5] += T * ln[2] + J.a35 * ln[3] + J.a45 * ln[4] + J.a55 * ln[5];
          }
        }
        if (hasu) {
          rU[0] += cWR(0) * (U[0] - cUR(0)); rU[1] += cWR(1) * (U[1] - cUR(1));
          if (ducost) {
            rU[0] += cWDR(0) * (U[0] - (k ? Up0 : cst[DCS_UL]));
            rU[1] += cWDR(1) * (U[1] - (k ? Up1 : cst[DCS_UL + 1]));
          }
          if (k + 1 < N) { rU[0] -= cWDR(0) * (Un0 - U[0]); rU[1] -= cWDR(1) * (Un1 - U[1]); }
          rU[0] += J.b4 * ln[4] + J.b5 * ln[5]; rU[1] += T * ln[3];
          if (k + 1 < N) { rU[0] += yR0n; rU[1] += yR1n; }
        }
        auto item = [&](const Bnd& q, double s, const Item& it) {
          if (q.hasL) { const double p = (s - q.L) * it.vL; svmax = fmax(svmax, p); svmin = fmin(svmin, p); sum_v += it.vL; }
          if (q.hasU) { const double p = (q.U - s) * it.vU; svmax = fmax(svmax, p); svmin = fmin(svmin, p); sum_v += it.vU; }
        };
        if (bu0_on) { rU[0] -= item_y(qU0, iU0); item(qU0, U[0], iU0); }
        if (bu1_on) { rU[1] -= item_y(qU1, iU1); item(qU1, U[1], iU1); }
        if (by_on) { rX[1] -= item_y(qY, iY); item(qY, X[1], iY); }
        if (bvx_on) { rX[3] -= item_y(qVx, iVx); item(qVx, X[3], iVx); }
        if (bvy_on) { rX[4] -= item_y(qVy, iVy); item(qVy, X[4], iVy); }
        if (r0_on) { rU[0] -= yR0; item(qR0, sR0, iR0); prim = fmax(prim, fabs(rR0)); }
        if (r1_on) { rU[1] -= yR1; item(qR1, sR1, iR1); prim = fmax(prim, fabs(rR1)); }
#pragma unroll
        for (int j = 0; j < NOBS; ++j) if (ro_on[j]) {
          rX[0] -= vO[j] * gO0[j]; rX[1] -= vO[j] * gO1[j];
          const double p = (sO[j] - qO.L) * vO[j]; svmax = fmax(svmax, p); svmin = fmin(svmin, p); sum_v += vO[j];
          prim = fmax(prim, fabs(rO[j]));
          if (RESTO && rs) {
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
        const double n_vr = (RESTO && rs) ? n_v + 2 * n_el : n_v;
        const double e_sd = fmax(S_MAX, (wv::uni(ss[0]) + wv::uni(ss[1])) / fmax(1.0, n_lam + n_vr)) / S_MAX;
        const double e_sc = fmax(S_MAX, wv::uni(ss[1]) / fmax(1.0, n_vr)) / S_MAX;
        const double base = fmax(e_dual / e_sd, e_prim);
        err0 = fmax(base, (n_vr > 0 ? sv_hi : 0.0) / e_sc);
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
                           fabs(fcur - cst[DCS_ACC]) <= lc->acceptable_obj_change_tol * fmax(1.0, fabs(fcur));
          const double acc_cnt = acc ? cst[DCS_ACC + 1] + 1.0 : 0.0;
          wv::sync();
          if (lane == 0) { cst[DCS_ACC] = fcur; cst[DCS_ACC + 1] = acc_cnt; }
          wv::sync();
          if (acc && acc_cnt >= (double)lc->acceptable_iter) { status = MPCB_ST_ACCEPTABLE; break; }
          if (iters >= c.max_iter) { status = MPCB_ST_MAXITER; break; }
        } else {
          // ----- restoration phase: violation of the ORIGINAL rows and the original barrier function at this iterate
          double t1 = fabs(rR0) + fabs(rR1), fm = 0, pl = 1.0, ti = fmax(fabs(rR0), fabs(rR1));
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
          if (bvx_on) dist(qVx, X[3]);
          if (bvy_on) dist(qVy, X[4]);
          if (r0_on) dist(qR0, sR0);
          if (r1_on) dist(qR1, sR1);
          if (hasu) {      // dyn.py:189-225 with the main phase's constants
#pragma unroll
            for (int i = 0; i < NX; ++i) { const double e = X[i] - cst[DCS_XS + i]; fm += cst[DCS_Q + i] * e * e; }
            fm += cst[DCS_R] * U[0] * U[0] + cst[DCS_R + 1] * U[1] * U[1];
            if (ducost) {
              const double d0 = U[0] - (k ? Up0 : cst[DCS_UL]), d1 = U[1] - (k ? Up1 : cst[DCS_UL + 1]);
              fm += cst[DCS_DR] * d0 * d0 + cst[DCS_DR + 1] * d1 * d1;
            }
          }
          double so[3] = {t1, fm, log(pl)}, mo[1] = {ti};
          wv::reduce<3, 1>(so, mo);
          const double th1 = wv::uni(so[0]), thinf = wv::uni(mo[0]);
          bool leave = false;
          if (rit >= 1 && th1 <= RS_KAPPA * th_entry && th1 <= tmax_main) {
            const double phi_o = os * wv::uni(so[1]) - mu_main * wv::uni(so[2]);
            leave = !(th1 >= fm_theta && phi_o >= fm_phi);
          }
          if (!leave && err0 <= c.tol) {
            if (thinf > c.tol) { status = MPCB_ST_INFEASIBLE; break; }
            if (rit == 0) { status = MPCB_ST_RESTO_FAILED; break; }
            leave = true;
          }
          if (!leave) {
            const double V = wv::uni(ss[2]);
            const double em = fmax(base, fmax(fabs(sv_hi - mu), fabs(sv_lo - mu)) / e_sc);
            const double gap = (RS_GAP * n_vr * mu + 0.5 * (double)((NX + NU) * N) * sqrt(mu)) / RS_RHO;
            if (em <= K_EPS * mu && V > gap + 1e-6 && theta <= 0.01 * V) { status = MPCB_ST_INFEASIBLE; break; }
            if (iters >= c.max_iter) { status = MPCB_ST_MAXITER; break; }
            if (n_riters >= RS_MAX_ITERS) { status = MPCB_ST_RESTO_FAILED; break; }
          } else {
            // ----- back to the main phase
            rs = false; osc = os; write_main_cost();
            double vm = 0;
            auto vmx = [&](const Bnd& q, const Item& it) { if (q.hasL) vm = fmax(vm, it.vL); if (q.hasU) vm = fmax(vm, it.vU); };
            if (bu0_on) vmx(qU0, iU0);
            if (bu1_on) vmx(qU1, iU1);
            if (by_on) vmx(qY, iY);
            if (bvx_on) vmx(qVx, iVx);
            if (bvy_on) vmx(qVy, iVy);
            if (r0_on) vmx(qR0, iR0);
            if (r1_on) vmx(qR1, iR1);
#pragma unroll
            for (int j = 0; j < NOBS; ++j) { if (ro_on[j]) vm = fmax(vm, vO[j]); }
#pragma unroll
            for (int j = 0; j < NEL; ++j) { eP[j] = 0; eN[j] = 0; vP[j] = 0; vN[j] = 0; }
            if (wv::uni(wv::max(vm)) > 1000.0) {
              iU0.vL = iU0.vU = iU1.vL = iU1.vU = iY.vL = iY.vU = iVx.vL = iVx.vU = iVy.vL = iVy.vU = iR0.vL = iR0.vU = iR1.vL = iR1.vU = 1.0;
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
            eval_lane(X, U, sR0, sR1, sO, eP, eN, ev, dfc, rR0, rR1, rO, Up0, Up1, th, fl, prod);
            double sw[3] = {th, fl, log(prod)};
            wv::reduce<3, 0>(sw, nullptr);
            theta = wv::uni(sw[0]); fval = wv::uni(sw[1]); logsum = wv::uni(sw[2]);
            slow_run = 0;
            continue;
          }
        }
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
        if (RESTO && rs && mu != mu_before) {
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

      // ----- condensed stage QP entries of node k -----------------------------------------------------------------------
      double hd[NW] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};    // diagonal of the stage Hessian (dw is added to X and U parts)
      {
        double g[NW] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        double h01 = 0, h68 = 0, h79 = 0;
        DynHess Hh = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (xq) {
#pragma unroll
          for (int i = 0; i < NX; ++i) { hd[i] += cWQ(i); g[i] += cWQ(i) * (X[i] - cXS(i)); }
        }
        if (hasu) {
          hd[8] += cWR(0); hd[9] += cWR(1);
          g[8] += cWR(0) * (U[0] - cUR(0)); g[9] += cWR(1) * (U[1] - cUR(1));
          if (ducost) {
            const double w0 = cWDR(0), w1 = cWDR(1);
            const double d0 = U[0] - (k ? Up0 : cst[DCS_UL]), d1 = U[1] - (k ? Up1 : cst[DCS_UL + 1]);
            hd[8] += w0; hd[6] += w0; h68 -= w0; hd[9] += w1; hd[7] += w1; h79 -= w1;
            g[8] += w0 * d0; g[6] -= w0 * d0; g[9] += w1 * d1; g[7] -= w1 * d1;
          }
          dyn_hess(MC(), T, X, ev, ln, Hh);
          hd[2] += Hh.h22; hd[3] += Hh.h33; hd[4] += Hh.h44; hd[5] += Hh.h55; hd[8] += Hh.h88;
        }
        double sig, gb;
        if (bu0_on) { item_sig_gb(iU0, 0.0, mu, sig, gb); hd[8] += sig; g[8] -= gb; }
        if (bu1_on) { item_sig_gb(iU1, 0.0, mu, sig, gb); hd[9] += sig; g[9] -= gb; }
        if (by_on) { item_sig_gb(iY, 0.0, mu, sig, gb); hd[1] += sig; g[1] -= gb; }
        if (bvx_on) { item_sig_gb(iVx, 0.0, mu, sig, gb); hd[3] += sig; g[3] -= gb; }
        if (bvy_on) { item_sig_gb(iVy, 0.0, mu, sig, gb); hd[4] += sig; g[4] -= gb; }
        if (r0_on) { item_sig_gb(iR0, rR0, mu, sig, gb); hd[8] += sig; hd[6] += sig; h68 -= sig; g[8] -= gb; g[6] += gb; }
        if (r1_on) { item_sig_gb(iR1, rR1, mu, sig, gb); hd[9] += sig; hd[7] += sig; h79 -= sig; g[9] -= gb; g[7] += gb; }
#pragma unroll
        for (int j = 0; j < NOBS; ++j) if (ro_on[j]) {
          sig = vO[j] * iO[j]; gb = mu * iO[j];
          if (RESTO && rs) {
            const double isp = eP[j] / vP[j], isn = eN[j] / vN[j];
            const double kap = 1.0 / (1.0 + sig * isp + sig * isn);
            const double rt = rO[j] + (RS_RHO + gb - mu / eP[j]) * isp + (gb - RS_RHO + mu / eN[j]) * isn;
            sig *= kap; gb -= sig * rt;
          } else gb -= sig * rO[j];
          hd[0] += sig * gO0[j] * gO0[j] - vO[j] * 2 * ix2(j);
          h01 += sig * gO0[j] * gO1[j];
          hd[1] += sig * gO1[j] * gO1[j] - vO[j] * 2 * iy2(j);
          g[0] -= gb * gO0[j]; g[1] -= gb * gO1[j];
        }
        if (isnode) {
          const double z = hasu ? 1.0 : 0.0;
#pragma unroll
          for (int i = 0; i < NW; ++i) ent[(DE_G0 + i) * ld + k] = g[i];
          double* fk = fw + k * DFWS;
          const double ja[14] = {J.a02, J.a03, J.a04, J.a12, J.a13, J.a14, J.a34, J.a35, J.a43, J.a44, J.a45, J.a53, J.a54, J.a55};
#pragma unroll
          for (int i = 0; i < 14; ++i) fk[DFW_A + i] = z * ja[i];
          fk[DFW_B] = z * J.b4; fk[DFW_B + 1] = z * J.b5;
#pragma unroll
          for (int i = 0; i < NX; ++i) fk[DFW_D + i] = dfc[i];
          fk[DFW_T] = T; fk[DFW_ZERO] = 0.0; fk[DFW_ONE] = 1.0;
          pst[k * DPSS + 8] = 0.0;
          ent[DE_H01 * ld + k] = h01; ent[DE_H23 * ld + k] = Hh.h23; ent[DE_H24 * ld + k] = Hh.h24;
          ent[DE_H34 * ld + k] = Hh.h34; ent[DE_H35 * ld + k] = Hh.h35; ent[DE_H45 * ld + k] = Hh.h45;
          ent[DE_H38 * ld + k] = Hh.h38; ent[DE_H48 * ld + k] = Hh.h48; ent[DE_H58 * ld + k] = Hh.h58;
          ent[DE_H66 * ld + k] = hd[6]; ent[DE_H77 * ld + k] = hd[7]; ent[DE_H68 * ld + k] = h68; ent[DE_H79 * ld + k] = h79;
        }
      }

      MPCB_STAMP(t_b);
      // ----- factorisation with inertia correction ---------------------------------------------------------------------
      double dw = 0.0; bool first_try = true, fact_ok = false;
#pragma clang loop unroll(disable)
      for (int tries = 0; tries < 60; ++tries) {
        if (isnode) {
          const double dx_ = xnode ? dw : 0.0, du_ = hasu ? dw : 0.0;
          ent[DE_H00 * ld + k] = hd[0] + dx_; ent[DE_H11 * ld + k] = hd[1] + dx_; ent[DE_H22 * ld + k] = hd[2] + dx_;
          ent[DE_H33 * ld + k] = hd[3] + dx_; ent[DE_H44 * ld + k] = hd[4] + dx_; ent[DE_H55 * ld + k] = hd[5] + dx_;
          ent[DE_H88 * ld + k] = hd[8] + du_; ent[DE_H99 * ld + k] = hd[9] + du_;
        }
        wv::sync();
        // terminal: P_N = H_N (state block), p_N = g_N
        Pst[N * DPST + ei * NA + ej] = ent[sHij + N];
        if (ej == 0) pst[N * DPSS + ei] = ent[(DE_G0 + ei) * ld + N];
        wv::sync();
        struct StageEnt { double abj[NX], abi[NX], start, hmat, startU, hU, startUU, b4, b5, Ts; };
        auto load_ent = [&](int s, StageEnt& e) {
          const double* fs = fw + s * DFWS;              // [A B | d], T: the stage's fw row, each lane its own slots
#pragma unroll
          for (int r = 0; r < NX; ++r) { e.abj[r] = fs[sABj[r]]; e.abi[r] = fs[sABi[r]]; }
          e.start = ent[sStart + s]; e.hmat = ent[sHij + s];
          e.startU = ent[sStartU + s]; e.hU = ent[sHU + s]; e.startUU = ent[sStartUU + s];
          e.b4 = fs[DFW_B]; e.b5 = fs[DFW_B + 1]; e.Ts = fs[DFW_T];
        };
        auto stage = [&](int s) -> bool {
          StageEnt e; load_ent(s, e);
          const double* Pn = Pst + (s + 1) * DPST + ei * NA;
          double Pr[NA];
#pragma unroll
          for (int r = 0; r < NA; ++r) Pr[r] = Pn[r];
          const double w0 = pst[(s + 1) * DPSS + pvOff];
          MPCB_SCHED_FENCE();
          // W_x = P+ [A | d] over the state columns (rows 6,7 of [A B] are zero), W_u = P+ B over the two control columns
          const double w = fma(Pr[4], e.abj[4], fma(Pr[2], e.abj[2], fma(Pr[0], e.abj[0], w0))) +
                           fma(Pr[5], e.abj[5], fma(Pr[3], e.abj[3], Pr[1] * e.abj[1]));
          const double wu = (ej & 1) ? fma(e.Ts, Pr[3], Pr[7]) : fma(e.b4, Pr[4], fma(e.b5, Pr[5], Pr[6]));
          Wl[ej * NA + ei] = w;
          WuL[wuOff] = wu;
          wv::sync();
          double Wc[NA], Wuc[NA];
#pragma unroll
          for (int r = 0; r < NA; ++r) Wc[r] = Wl[ej * NA + r];
#pragma unroll
          for (int r = 3; r < NA; ++r) Wuc[r] = WuL[(ej & 1) * NA + r];
          MPCB_SCHED_FENCE();
          // M_xx = H_xx + A^T W_x  (affine lanes: m_x);  M_ux(c,:) = H_ux + B_c^T W_x (affine lanes: m_u);  M_uu = H_uu + B^T W_u
          const double acc = fma(e.abi[4], Wc[4], fma(e.abi[2], Wc[2], fma(e.abi[0], Wc[0], e.start))) +
                             fma(e.abi[5], Wc[5], fma(e.abi[3], Wc[3], e.abi[1] * Wc[1]));
          const double mux = e.startU + (cU ? fma(e.Ts, Wc[3], Wc[7]) : fma(e.b4, Wc[4], fma(e.b5, Wc[5], Wc[6])));
          const double muu = e.startUU + (cU ? fma(e.Ts, Wuc[3], Wuc[7]) : fma(e.b4, Wuc[4], fma(e.b5, Wuc[5], Wuc[6])));
          const double Mx = aff ? e.hmat : acc;
          const double MxU = aff ? e.hU : mux;
          // the control rows of M, M_uu and m_u go to every lane through a staging block that aliases W^T / W_u^T (both spent):
          // three stores and five 16-byte reads instead of ten v_readlane and eight ds_bpermute.  Rows interleaved: (c, j) at 2j + c
          wv::sync();                                    // every lane has read its W column
          Wl[sOff] = MxU; Wl[sOff + 32] = muu; Wl[sOff + 64] = mux;
          wv::sync();
          const double m11 = Wl[32], m12 = Wl[32 + 2], m22 = Wl[32 + 3];
          const double mu8 = Wl[64 + 12], mu9 = Wl[64 + 13];
          const double M8j = Wl[2 * ej], M9j = Wl[2 * ej + 1];
          const double M8i = Wl[2 * ei], M9i = Wl[2 * ei + 1];
          const double det = m11 * m22 - m12 * m12, dmar = det - 1e-14 * m11 * m22;
          const bool okpd = (m11 > 0) & (dmar > 0) & (dmar < 1e300);                    // wave-uniform; false for NaN / inf
          const double idet = wv::rcp(det);
          const double i11 = m22 * idet, i12 = -m12 * idet, i22 = m11 * idet;
          const double kf0 = -(i11 * mu8 + i12 * mu9), kf1 = -(i12 * mu8 + i22 * mu9);
          const double K0j = -(i11 * M8j + i12 * M9j), K1j = -(i12 * M8j + i22 * M9j);
          // upper-triangle lanes store to (i,j) and (j,i), the others into pad slots: P stays symmetric through the sweep (see
          // the kinematic kernel: without it the asymmetry of the cancellation errors compounds and delta_w escalates)
          const double Pij = Mx + M8i * K0j + M9i * K1j;
          Pst[s * DPST + pOff] = Pij;
          Pst[s * DPST + pOffT] = Pij;
          pst[s * DPSS + psOff] = acc + M8i * kf0 + M9i * kf1;
          fw[s * DFWS + kOff] = kRow1 ? K1j : K0j;
          fw[s * DFWS + kfOff] = kfLane1 ? kf1 : kf0;
          wv::sync();
          return okpd;
        };
        bool pd = true;
#pragma clang loop unroll(disable)
        for (int s = N - 1; s >= 0 && pd; --s) pd = stage(s);       // one register set for the stage tables, loaded at the stage top: refilling
                                                                    // it inside the stage (prefetch without a second set) still costs 64 B more scratch and 3 %
        if (pd) { fact_ok = true; if (dw > 0) dw_last = dw; break; }
        if (first_try) { dw = (dw_last == 0.0) ? DW_FIRST : fmax(DW_MIN, KW_MINUS * dw_last); first_try = false; }
        else dw *= (dw_last == 0.0) ? KW_PLUS_FIRST : KW_PLUS;
        if (dw > DW_MAX) break;
      }
      if (!fact_ok) { status = MPCB_ST_NUMERIC; break; }

      MPCB_STAMP(t_c);
      // ----- forward roll-out of the step: lane i < 8 advances component i of [dX_s; dU_{s-1}] (see the kinematic kernel) ---------
      // t_i = c_i + sum_r C_ir v_r (rows 0..5: the A part and the defect, rows 6, 7: the gain rows, t = dU_s), n_i = t_i + b_i0 dU_s[0] +
      // b_i1 dU_s[1].  v travels through scalar registers; a lane reads its eleven numbers at its own slots of the stage's fw row;
      // the steps go to the rows of the spent condensed gradient, [component][node].
      double dX[NX] = {0, 0, 0, 0, 0, 0}, dU[NU] = {0, 0};
      {
        const int lq = wv::opaque(lane);               // (re-formed per iteration: the offsets must not stay live through the solve)
        const int li = lq < NA ? lq : 0;
        int fo[NA + 3];
#pragma unroll
        for (int r = 0; r < NA; ++r) fo[r] = li < NX ? slotAB(li, r) : (li - NX) * NA + r;
        fo[NA] = li < NX ? DFW_D + li : DFW_KFF + (li - NX);
        fo[NA + 1] = li < NX ? slotAB(li, 8) : DFW_ZERO;
        fo[NA + 2] = li < NX ? slotAB(li, 9) : DFW_ZERO;
        // rows 0..5 hold dX_{s+1}, rows 6, 7 hold dU_s; the other lanes store into the two remaining gradient rows
        double* hist = ent + (DE_G0 + (lq < NA ? lq : NA + (lq & 1))) * ld + (lq < NX ? 1 : 0);
        if (lane < NX) hist[-1] = 0.0;                 // dX_0 = 0 (X_0 is pinned)
        struct FwRec { double c[NA], c0, b0, b1; };
        auto load_fw = [&](int s, FwRec& f) {
          const double* q = fw + s * DFWS;
#pragma unroll
          for (int r = 0; r < NA; ++r) f.c[r] = q[fo[r]];
          f.c0 = q[fo[NA]]; f.b0 = q[fo[NA + 1]]; f.b1 = q[fo[NA + 2]];
        };
        double v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0, v6 = 0, v7 = 0;      // wave-uniform
        auto fstage = [&](int s, const FwRec& f, FwRec& nxt) {
          load_fw(s + 1, nxt);                         // (row N exists and holds finite numbers; its values are never used)
          MPCB_SCHED_FENCE();
          const double t = fma(f.c[7], v7, fma(f.c[6], v6, fma(f.c[5], v5, fma(f.c[4], v4, fma(f.c[3], v3, fma(f.c[2], v2, fma(f.c[1], v1, fma(f.c[0], v0, f.c0))))))));
          const double du0 = wv::bcast(t, NX), du1 = wv::bcast(t, NX + 1);
          const double n = fma(f.b1, du1, fma(f.b0, du0, t));
          hist[s] = n;
          v0 = wv::bcast(n, 0); v1 = wv::bcast(n, 1); v2 = wv::bcast(n, 2); v3 = wv::bcast(n, 3); v4 = wv::bcast(n, 4); v5 = wv::bcast(n, 5);
          v6 = du0; v7 = du1;
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
          for (int i = 0; i < NX; ++i) dX[i] = ent[(DE_G0 + i) * ld + k];
        }
        if (hasu) { dU[0] = ent[(DE_G0 + NX) * ld + k]; dU[1] = ent[(DE_G0 + NX + 1) * ld + k]; }
      }
      const double dUp0 = wv::shfl(dU[0], k - 1), dUp1 = wv::shfl(dU[1], k - 1);
      double lamF[NX] = {0, 0, 0, 0, 0, 0};
      if (xnode) {
        const double* Pk = Pst + k * DPST;
        const double dxa[NA] = {dX[0], dX[1], dX[2], dX[3], dX[4], dX[5], dUp0, dUp1};
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          double s = pst[k * DPSS + i];
#pragma unroll
          for (int r = 0; r < NA; ++r) s += Pk[i * NA + r] * dxa[r];
          lamF[i] = s;
        }
      }
      const double dsR0 = r0_on ? (dU[0] - dUp0) + rR0 : 0.0, dsR1 = r1_on ? (dU[1] - dUp1) + rR1 : 0.0;
      recips();
      double dsO[NOB], dP[NEL], dN[NEL], dvP[NEL], dvN[NEL];
#pragma unroll
      for (int j = 0; j < NEL; ++j) { dP[j] = 0; dN[j] = 0; dvP[j] = 0; dvN[j] = 0; }
#pragma unroll
      for (int j = 0; j < NOBS; ++j) {
        dsO[j] = ro_on[j] ? gO0[j] * dX[0] + gO1[j] * dX[1] + rO[j] : 0.0;
        if (RESTO && rs && ro_on[j]) {
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

      double a_pr, a_du, dphi;     // (the reciprocals were recomputed above instead of being kept live across the sweeps: register pressure)
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
        if (bvx_on) ftb(qVx, iVx, dX[3]);
        if (bvy_on) ftb(qVy, iVy, dX[4]);
        if (r0_on) ftb(qR0, iR0, dsR0);
        if (r1_on) ftb(qR1, iR1, dsR1);
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
          if (ducost) {
            d += cWDR(0) * (U[0] - (k ? Up0 : cst[DCS_UL])) * (dU[0] - (k ? dUp0 : 0.0));
            d += cWDR(1) * (U[1] - (k ? Up1 : cst[DCS_UL + 1])) * (dU[1] - (k ? dUp1 : 0.0));
          }
        }
        double ss[1] = {d}, mm[2] = {rpr, rdu};
        wv::reduce<1, 2>(ss, mm);
        dphi = wv::uni(ss[0]);
        const double r1 = wv::uni(mm[0]), r2 = wv::uni(mm[1]);
        a_pr = wv::uni((r1 > tau) ? tau / r1 : 1.0);
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
      double alpha = a_pr; bool accepted = false, armijo_type = false;
      double Xt[NX], Ut[NU], dft[NX], sR0t, sR1t, rR0t, rR1t, sOt[NOB], rOt[NOB], pt[NEL], nt[NEL], upt0, upt1, tht = 0, ft = 0, lst = 0;
#pragma unroll
      for (int j = 0; j < NEL; ++j) { pt[j] = 0; nt[j] = 0; }
      DynEval et;
#pragma clang loop unroll(disable)
      for (;;) {
#pragma unroll
        for (int i = 0; i < NX; ++i) Xt[i] = X[i] + alpha * dX[i];
        Ut[0] = U[0] + alpha * dU[0]; Ut[1] = U[1] + alpha * dU[1];
        sR0t = sR0 + alpha * dsR0; sR1t = sR1 + alpha * dsR1;
#pragma unroll
        for (int j = 0; j < NOBS; ++j) sOt[j] = sO[j] + alpha * dsO[j];
        if (RESTO && rs) {
#pragma unroll
          for (int j = 0; j < NOBS; ++j) { pt[j] = eP[j] + alpha * dP[j]; nt[j] = eN[j] + alpha * dN[j]; }
        }
        dyn_eval(MC(), Xt, Ut, et);
        double th, fl, prod;
        const bool okl = eval_lane(Xt, Ut, sR0t, sR1t, sOt, pt, nt, et, dft, rR0t, rR1t, rOt, upt0, upt1, th, fl, prod);
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
      auto hand_over = [&](int it_done) {
        status = MPCB_ST_NEEDS_RESTO;
        if (lane == 0 && a.work) {
          double* wk = a.work + (size_t)b * WK_SIZE;
          wk[WK_MU] = mu; wk[WK_THMAX] = theta_max; wk[WK_THMIN] = theta_min; wk[WK_ITERS] = (double)it_done; wk[WK_DW] = dw_last;
          wk[WK_START] = pass == MPCB_PASS_SECOND ? 1.0 : 0.0;
        }
      };
      if (!accepted) {
        if (RESTO && rs) { status = MPCB_ST_RESTO_FAILED; break; }
        if (!c.restoration) { status = MPCB_ST_LINESEARCH; break; }
        // failure at an (almost) feasible point = round-off in the end game: nothing to restore (IPOPT: "Restoration phase is called
        // at point that is almost feasible" -> Restoration_Failed); the nearly converged iterate is returned as it is
        if (e_prim <= c.tol) { status = MPCB_ST_RESTO_FAILED; break; }
        if (!RESTO) { hand_over(iters); break; }
        enter = true; continue;
      }
      if (!armijo_type) {
        if (nfilt < FILTER_MAX) {
          if (lane == 0) { filt[2 * nfilt] = (1 - G_THETA) * th0; filt[2 * nfilt + 1] = phi0 - G_PHI * th0; }
          ++nfilt;
        }
        wv::sync();
      }

      recips();
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
      if (bvx_on) upd(qVx, iVx, dX[3], Xt[3]);
      if (bvy_on) upd(qVy, iVy, dX[4], Xt[4]);
      if (r0_on) upd(qR0, iR0, dsR0, sR0t);
      if (r1_on) upd(qR1, iR1, dsR1, sR1t);
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
      sR0 = sR0t; sR1 = sR1t; rR0 = rR0t; rR1 = rR1t; Up0 = upt0; Up1 = upt1;
#pragma unroll
      for (int i = 0; i < NX; ++i) dfc[i] = dft[i];
#pragma unroll
      for (int j = 0; j < NOBS; ++j) {
        sO[j] = sOt[j]; rO[j] = rOt[j];
        if (ro_on[j]) { gO0[j] = 2 * (X[0] - ox(j)) * ix2(j); gO1[j] = 2 * (X[1] - oy(j)) * iy2(j); }
      }
      theta = tht; fval = ft; logsum = lst;
      if (!isfinite(theta) || !isfinite(fval)) { status = MPCB_ST_NUMERIC; break; }
      ++iters;
      if (RESTO && rs) { ++rit; ++n_riters; }
      else if (c.restoration) {       // early entry into restoration / hand-over to the second start (see mpcb_solve_kin)
        if (alpha < TRIG_ALPHA && theta > 1e-6) { if (slow_run == 0) slow_theta0 = th0; ++slow_run; } else slow_run = 0;
        if (slow_run >= TRIG_K && theta > TRIG_THETA * slow_theta0) {
          slow_run = 0;
          if (!RESTO) { hand_over(iters); break; }
          enter = true;
        }
      }
    }
  } else {
    double Xs[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) Xs[i] = X[i];
    if (!(Xs[3] > 1e-3)) Xs[3] = 1e-3;
    DynEval ev; dyn_eval(MC(), Xs, U, ev);
    double th, fl, prod;
    eval_lane(Xs, U, sR0, sR1, sO, eP, eN, ev, dfc, rR0, rR1, rO, Up0, Up1, th, fl, prod);
    fval = wv::sum(fl);
  }
  if (RESTO && rs) {    // ended inside the restoration phase: report the objective of the original problem
    rs = false; osc = os; write_main_cost();
    DynEval ev; dyn_eval(MC(), X, U, ev);
    double th, fl, prod;
    eval_lane(X, U, sR0, sR1, sO, eP, eN, ev, dfc, rR0, rR1, rO, Up0, Up1, th, fl, prod);
    fval = wv::sum(fl);
  }

  // ----- outputs -----------------------------------------------------------------------------------------------------------
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
    if (bvx_on) zbuf[NU * N + NX * ko + 3] = -item_y(qVx, iVx) / os;
    if (bvy_on) zbuf[NU * N + NX * ko + 4] = -item_y(qVy, iVy) / os;
    wv::sync();
    for (int i = lo; i < nz; i += 64) ao.lam_x[(size_t)b * nz + i] = zbuf[i];
  }
  if (ao.want_mult && ao.lam_g) {
    // g order: [X_0 - P](6); then per stage i: dynamics(6) and, for i > 0, the rate rows (interleaved, dyn.py:226-231)
    // or all dynamics rows followed by the rate block; then the obstacle rows
    double* out = ao.lam_g + (size_t)b * ao.ng;
    const int nr = (qR0.on ? 1 : 0) + (qR1.on ? 1 : 0);
    double ln[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) ln[i] = wv::shfl(lam[i], k + 1);
    auto dyn_row = [&](int node) { return c.rate_interleaved ? NX + (node - 1) * NX + (node >= 2 ? (node - 2) * nr : 0) : NX + (node - 1) * NX; };
    auto rate_row = [&](int node) { return c.rate_interleaved ? NX + (node + 1) * NX + (node - 1) * nr : NX + NX * N + (node - 1) * nr; };
    const int r_obs = NX * (N + 1) + nr * (N - 1);
    if (xnode) {
#pragma unroll
      for (int i = 0; i < NX; ++i) out[dyn_row(k) + i] = -lam[i] / os;
    }
    if (k == 0) {
      double Xs[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) Xs[i] = X[i];
      if (!(Xs[3] > 1e-3)) Xs[3] = 1e-3;
      DynEval ev; dyn_eval(MC(), Xs, U, ev);
      DynJac J; dyn_jac(MC(), T, Xs, ev, J);
      const double At[NX] = {ln[0], ln[1], J.a02 * ln[0] + J.a12 * ln[1] + ln[2],
                             J.a03 * ln[0] + J.a13 * ln[1] + ln[3] + J.a43 * ln[4] + J.a53 * ln[5],
                             J.a04 * ln[0] + J.a14 * ln[1] + J.a34 * ln[3] + J.a44 * ln[4] + J.a54 * ln[5],
                             T * ln[2] + J.a35 * ln[3] + J.a45 * ln[4] + J.a55 * ln[5]};
#pragma unroll
      for (int i = 0; i < NX; ++i) out[i] = -2 * c.Q[i] * (X[i] - cst[DCS_XS + i]) - At[i] / os;
    }
    if (xcost) {
      int q = 0;
      if (qR0.on) out[rate_row(k) + q++] = r0_on ? -item_y(qR0, iR0) / os : 0.0;
      if (qR1.on) out[rate_row(k) + q++] = r1_on ? -item_y(qR1, iR1) / os : 0.0;
    }
    if (isnode) {
      const int row = (c.obs_mode == MPCB_OBS_KEEPOUT) ? k : k - 1;
#pragma unroll
      for (int j = 0; j < NOBS; ++j) if (j < nobs && row >= 0 && row <= last_row) out[r_obs + row * nobs + j] = ro_on[j] ? -vO[j] / os : 0.0;
    }
  }
}
#undef ox
#undef oy
#undef ix2
#undef iy2
