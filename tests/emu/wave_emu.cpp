// tests/emu/wave_emu.cpp — TEST INFRASTRUCTURE ONLY.
//
// Steps the device source mpc_motion_planning_amd/csrc/mpcb_kernel.h on the CPU: every lane of the wavefront is a
// host thread, cross-lane primitives (mpcb_wave.h, MPCB_WAVE_EMU branch) go through a barrier.  It exists so that
// the kernel's logic can be checked against the oracle in the `-m "not gpu"` suite and while developing without
// a GPU.  It is NOT part of libmpcbatch.so, is never loaded by the mpc_motion_planning_amd package and is far
// too slow to be a fallback (64 OS threads per instance).
#define MPCB_WAVE_EMU 1
#include "../../mpc_motion_planning_amd/csrc/mpcb_kernel_dyn.h"

#include <thread>
#include <limits>
#include <cstdlib>
#include <vector>

namespace wv {
thread_local int t_lane = 0;
thread_local Emu* t_emu = nullptr;
}

template <int NOBS>
static void run_instance(const MpcbKArgs& a, int b) {
  using namespace mpcbk;
  const bool dyn = a.cfg.model == MPCB_MODEL_DYN;
  const bool rp = a.pass == MPCB_PASS_RESTO;                      // restoration pass: the RESTO instantiations
  const bool rk4 = !dyn && a.cfg.integrator == MPCB_INT_RK4;
  const int total = dyn ? layout_dyn(a.cfg.N, rp, obs_in_lds(NOBS)).total : layout_kin(a.cfg.N, a.nz, rp, obs_in_lds(NOBS)).total;
  // LDS starts as garbage on the device: poison it here (MPCB_EMU_LDS_FILL, default NaN), so that a read of a never-written slot shows
  const char* fill_env = std::getenv("MPCB_EMU_LDS_FILL");
  std::vector<double> lds(total + 64, fill_env ? std::atof(fill_env) : std::numeric_limits<double>::quiet_NaN());
  std::barrier<> bar(64);
  wv::Emu emu; emu.bar = &bar;
  std::vector<std::thread> th;
  for (int l = 0; l < 64; ++l)
    th.emplace_back([&, l]() {
      wv::t_lane = l; wv::t_emu = &emu;
      const bool gen = !dyn && a.cfg.obs_mode == MPCB_OBS_DCBF && a.cfg.gamma < 1.0 - 1e-12 && NOBS > 0;   // as mpcb_api.hip dispatches
      if (dyn && rp) mpcb_solve_dyn<NOBS, true>(a, b, lds.data(), a.pass);
      else if (dyn) mpcb_solve_dyn<NOBS>(a, b, lds.data(), a.pass);
      else if (gen && rp) mpcb_solve_kin<(NOBS > 0 ? NOBS : 1), true, true>(a, b, lds.data(), a.pass);
      else if (gen) mpcb_solve_kin<(NOBS > 0 ? NOBS : 1), true>(a, b, lds.data(), a.pass);
      else if (rk4 && rp) mpcb_solve_kin<(NOBS <= 3 ? NOBS : 3), false, true, true>(a, b, lds.data(), a.pass);
      else if (rk4) mpcb_solve_kin<(NOBS <= 3 ? NOBS : 3), false, false, true>(a, b, lds.data(), a.pass);
      else if (rp) mpcb_solve_kin<NOBS, false, true>(a, b, lds.data(), a.pass);
      else mpcb_solve_kin<NOBS>(a, b, lds.data(), a.pass);
    });
  for (auto& t : th) t.join();
}

extern "C" int mpcb_emu_solve(const mpcb_config* cfg, int32_t B, const double* x0, const double* xs, const double* obs,
                              int32_t obs_kind, const double* z0, double* z, double* obj, int32_t* status, int32_t* iters,
                              double* kkt, double* lam_g, double* lam_x, double* trace, int32_t trace_instance, const double* tgrid) {
  if (!cfg) return MPCB_E_INVALID;
  const int nx = cfg->model == MPCB_MODEL_DYN ? 6 : 4;
  int nrate = 0;
  for (int i = 0; i < 2; ++i) if (cfg->du_lo[i] > -1e300 || cfg->du_hi[i] < 1e300) ++nrate;
  MpcbKArgs a{};
  a.st_stride = 1;
  a.cfg = *cfg; a.B = B; a.obs_kind = obs_kind; a.want_mult = (lam_g || lam_x) ? 1 : 0; a.trace_instance = trace_instance;
  a.nz = 2 * cfg->N + nx * (cfg->N + 1);
  a.ng = nx * (cfg->N + 1) + nrate * (cfg->N - 1) + cfg->n_obs * (cfg->obs_terminal ? cfg->N + 1 : cfg->N);
  a.x0 = x0; a.xs = xs; a.obs = obs; a.z0 = z0; a.z = z; a.obj = obj; a.kkt = kkt; a.lam_g = lam_g; a.lam_x = lam_x;
  a.status = status; a.iters = iters; a.trace = trace; a.tgrid = tgrid;
  std::vector<double> work((size_t)B * mpcbk::WK_SIZE, 0.0);
  // as mpcb_api.hip: first pass, second-start pass (cfg.second_start), restoration pass (cfg.restoration)
  const bool second = cfg->second_start && cfg->init_rollout;
  a.work = (cfg->restoration || second) ? work.data() : nullptr;
  // launch order of mpcb_api.hip: first attempt, its restoration pass, second attempt, its restoration pass
  const int order[4] = {MPCB_PASS_FIRST, MPCB_PASS_RESTO, MPCB_PASS_SECOND, MPCB_PASS_RESTO};
  for (int q = 0; q < 4; ++q) {
    const int pass = order[q];
    if ((q >= 2 && !second) || (pass == MPCB_PASS_RESTO && !cfg->restoration)) continue;
    const int ss = cfg->second_start == 3 ? (z0 ? 2 : 1) : cfg->second_start;   // 3: by the kind of start, as mpcb_api.hip
    if (q == 1 && second && ss == 1) continue;      // second start instead of the first attempt's restoration
    a.pass = pass;
    for (int b = 0; b < B; ++b) {
      if (pass == MPCB_PASS_SECOND && (status[b] == MPCB_ST_SOLVED || status[b] == MPCB_ST_ACCEPTABLE || status[b] == MPCB_ST_INFEASIBLE_X0)) continue;
      if (pass == MPCB_PASS_RESTO && status[b] != MPCB_ST_NEEDS_RESTO) continue;
      if (cfg->n_obs == 0) run_instance<0>(a, b);
      else if (cfg->n_obs == 1) run_instance<1>(a, b);
      else if (cfg->n_obs <= 3) run_instance<3>(a, b);
      else if (cfg->n_obs <= 8) run_instance<8>(a, b);
      else return MPCB_E_UNSUPPORTED;
    }
  }
  return MPCB_OK;
}

// closed-form dyn model derivatives of the kernel source (host-compiled) for comparison with the oracle's AD
// LDS bytes of one instance (= one workgroup) as the kernels lay it out: `pass` 0 first pass, 1 restoration pass; as mpcb_api.hip computes it
extern "C" int64_t mpcb_emu_lds_bytes(const mpcb_config* cfg, int32_t pass) {
  using namespace mpcbk;
  const int n = cfg->n_obs;
  if (cfg->model == MPCB_MODEL_DYN) return (int64_t)layout_dyn(cfg->N, pass == 1, obs_in_lds(obs_capacity_dyn(n))).total * 8;
  const bool gen = cfg->obs_mode == MPCB_OBS_DCBF && cfg->gamma < 1.0 - 1e-12 && n > 0;
  const int nz = 2 * cfg->N + 4 * (cfg->N + 1);
  return (int64_t)layout_kin(cfg->N, nz, pass == 1, obs_in_lds(obs_capacity_kin(n, gen)), gen).total * 8;
}

extern "C" int mpcb_emu_dyn_model(const mpcb_config* cfg, const double* X, const double* U, const double* lam, double* F, double* jac16,
                                  double* hess13) {
  using namespace mpcbk;
  DynEval e; dyn_eval(*cfg, X, U, e);
  dyn_F(*cfg, cfg->T, X, U, e, F);
  DynJac J; dyn_jac(*cfg, cfg->T, X, e, J);
  const double j[16] = {J.a02, J.a03, J.a04, J.a12, J.a13, J.a14, J.a34, J.a35, J.a43, J.a44, J.a45, J.a53, J.a54, J.a55, J.b4, J.b5};
  for (int i = 0; i < 16; ++i) jac16[i] = j[i];
  DynHess H; dyn_hess(*cfg, cfg->T, X, e, lam, H);
  const double h[13] = {H.h22, H.h23, H.h24, H.h33, H.h34, H.h35, H.h44, H.h45, H.h55, H.h38, H.h48, H.h58, H.h88};
  for (int i = 0; i < 13; ++i) hess13[i] = h[i];
  return 0;
}
