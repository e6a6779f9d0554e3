// mpcb_api.hip — C ABI of libmpcbatch.so (include/mpcbatch.h) and the kernel launches.  gfx950 only.
//
// Host side: plain HIP runtime, one stream per handle, caller-owned buffers.  No torch, no oracle, no CPU
// fallback: without a HIP device mpcb_create fails with MPCB_E_DEVICE.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>     // types and prototypes only: librccl is loaded on first use (dlopen), see Rccl below
#include <dlfcn.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "mpcb_kernel_dyn.h"

#ifndef MPCB_WAVES_PER_SIMD
#define MPCB_WAVES_PER_SIMD 1
#endif

namespace {

constexpr double INF = std::numeric_limits<double>::infinity();
thread_local std::string g_create_error;

// Second attempt inside the first launch (second start of kind 1, the cold-start batches): the wave whose first attempt failed starts
// over from z = 0 at once instead of in a second launch that can only begin when the slowest first attempt of the batch has finished.
// Only the instantiations with registers to spare do this (kin<0>, kin<1>: +17 AGPRs; measured C2 +2.5 % with six lanes, +11 % with one
// launch at a time): the loop around the inlined solve keeps loop-invariant per-lane values alive across both attempts, which costs
// kin<3> 200 -> 256 AGPRs + 124 B of scratch (C3 -9 %) and dyn<3> 236 -> 256 + 452 B (C4 -7 %).  Those launch the second attempt as a
// pass of its own (launch_solve).  -DMPCB_NO_FUSED_SECOND: no instantiation fuses (A/B builds).
#ifdef MPCB_NO_FUSED_SECOND
template <int NOBS, bool GEN, bool RK4> constexpr bool mpcb_kin_fuses = false;
template <int NOBS> constexpr bool mpcb_dyn_fuses = false;
#else
template <int NOBS, bool GEN, bool RK4> constexpr bool mpcb_kin_fuses = NOBS <= 3 && !GEN && !RK4;
template <int NOBS> constexpr bool mpcb_dyn_fuses = NOBS <= 3;
#endif
__host__ __device__ inline bool mpcb_second_kind1(const mpcb_config& c, const void* z0) {   // cfg.second_start = 3: by the kind of start
  return c.init_rollout && (c.second_start == 1 || (c.second_start == 3 && !z0));
}
__device__ __forceinline__ bool mpcb_second_attempt_here(const MpcbKArgs& a, int b, int pass) {
  if (pass != MPCB_PASS_FIRST || !mpcb_second_kind1(a.cfg, a.z0) || !a.status) return false;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");          // the status lane 0 has just stored
  const int st = __builtin_nontemporal_load(a.status + (size_t)b * a.st_stride);
  return st != MPCB_ST_SOLVED && st != MPCB_ST_ACCEPTABLE && st != MPCB_ST_INFEASIBLE_X0;
}

template <int NOBS, bool GEN = false, bool RK4 = false>
__global__ __launch_bounds__(64, MPCB_WAVES_PER_SIMD) void mpcb_kernel_kin(const MpcbKArgs a) {
  extern __shared__ __attribute__((aligned(16))) double mpcb_lds[];
  if constexpr (mpcb_kin_fuses<NOBS, GEN, RK4>) {
    int pass = a.pass;
#pragma clang loop unroll(disable)
    for (;;) {
      mpcb_solve_kin<NOBS, GEN, false, RK4>(a, (int)blockIdx.x, mpcb_lds, pass);
      if (!mpcb_second_attempt_here(a, (int)blockIdx.x, pass)) break;
      pass = MPCB_PASS_SECOND;
    }
  } else {
    mpcb_solve_kin<NOBS, GEN, false, RK4>(a, (int)blockIdx.x, mpcb_lds, a.pass);
  }
}

// restoration pass: main phase + restoration phase; a workgroup whose instance does not need it returns at once
template <int NOBS, bool GEN = false, bool RK4 = false>
__global__ __launch_bounds__(64, 1) void mpcb_kernel_kin_resto(const MpcbKArgs a) {
  extern __shared__ __attribute__((aligned(16))) double mpcb_lds[];
  mpcb_solve_kin<NOBS, GEN, true, RK4>(a, (int)blockIdx.x, mpcb_lds, MPCB_PASS_RESTO);
}

template <int NOBS>
__global__ __launch_bounds__(64, MPCB_WAVES_PER_SIMD) void mpcb_kernel_dyn(const MpcbKArgs a) {
  extern __shared__ __attribute__((aligned(16))) double mpcb_lds[];
  if constexpr (mpcb_dyn_fuses<NOBS>) {
    int pass = a.pass;
#pragma clang loop unroll(disable)
    for (;;) {
      mpcb_solve_dyn<NOBS>(a, (int)blockIdx.x, mpcb_lds, pass);
      if (!mpcb_second_attempt_here(a, (int)blockIdx.x, pass)) break;
      pass = MPCB_PASS_SECOND;
    }
  } else {
    mpcb_solve_dyn<NOBS>(a, (int)blockIdx.x, mpcb_lds, a.pass);
  }
}

template <int NOBS>
__global__ __launch_bounds__(64, 1) void mpcb_kernel_dyn_resto(const MpcbKArgs a) {
  extern __shared__ __attribute__((aligned(16))) double mpcb_lds[];
  mpcb_solve_dyn<NOBS, true>(a, (int)blockIdx.x, mpcb_lds, MPCB_PASS_RESTO);
}

// f(x,u) of the configured model: the reference's `mpc_solver.f` (kin.py:153-159, dyn.py:156-177); host and device
__host__ __device__ inline void model_rhs(const mpcb_config& c, const double* x, const double* u, double* xdot) {
#pragma clang fp contract(off)   // every product and sum rounded on its own, as numpy / CasADi evaluate the reference's expressions
  if (c.model == MPCB_MODEL_KIN) {                       // kin.py:153-156
    xdot[0] = x[3] * cos(x[2]);
    xdot[1] = x[3] * sin(x[2]);
    xdot[2] = x[3] * tan(u[0]) / c.veh_l;
    xdot[3] = u[1];
  } else {                                               // dyn.py:156-170
    const double phi = x[2], vx = x[3], vy = x[4], r = x[5], df = u[0], ax = u[1];
    const double af = df - (vy + c.veh_lf * r) / vx, ar = -(vy - c.veh_lr * r) / vx;
    const double Cf = c.Fymax_f * 2 * c.aopt_f / (c.aopt_f * c.aopt_f + af * af);
    const double Cr = c.Fymax_r * 2 * c.aopt_r / (c.aopt_r * c.aopt_r + ar * ar);
    const double Fcf = -Cf * af, Fcr = -Cr * ar;
    xdot[0] = vx * cos(phi) - vy * sin(phi);
    xdot[1] = vx * sin(phi) + vy * cos(phi);
    xdot[2] = r;
    xdot[3] = ax + r * vy;
    xdot[4] = -r * vx + 2.0 / c.veh_m * (Fcf * cos(df) + Fcr);
    xdot[5] = 2.0 / c.veh_Iz * (c.veh_lf * Fcf - c.veh_lr * Fcr);
  }
}

// closed-loop helper: plant step with the first control, warm-start shift, obstacle advance; either model.
// one thread per instance (tiny, HBM-bound, runs between two solves of the closed loop)
//   main_cbf_kin_c_sim.py:16-26 / main_cbf_dyn_c_sim.py:15-25 (shift_movement), main_cbf_kin_c_sim_pre.py:106 (obstacle advance)
// NX is a template parameter so that x[] / f[] live in registers (with a run-time nx the compiler parked them in LDS).
//   move_obs: 0 none, 1 every obstacle one constant-velocity step, 2 only the first one (what main_cbf_kin_c_sim_pre.py:106 does)
//   hold: 1 = an instance whose solve did not end with MPCB_ST_SOLVED applies its previous plan instead (the shifted warm start
//         z0 IS that plan, already moved one step): hold-and-shift, prior art `reference code/MPC-D-CBF.py:341-353`
template <int NX>
__global__ __launch_bounds__(128) void mpcb_advance(const mpcb_config c, int B, int nz, const double* __restrict__ z,
                                                    double* __restrict__ x0, double* __restrict__ z0, double* __restrict__ obs,
                                                    double* __restrict__ x_hist, double* __restrict__ u_hist,
                                                    const int32_t* __restrict__ status, int st_stride, int step, int steps,
                                                    int move_obs, int hold, double T) {
#pragma clang fp contract(off)   // st = x0 + T*f and x += v*cos(theta)*dt with numpy's roundings (no FMA): bit-equal to the reference's helpers
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int N = c.N, n_obs = c.n_obs;     // T: length of the executed step (cfg.T, or T_0 of the time grid)
  double* w = z0 + (size_t)b * nz;
  // the plan that is executed: this step's solution, or (hold) the previous plan kept in z0
  const int st_b = status ? status[(size_t)b * st_stride] : MPCB_ST_SOLVED;
  const bool keep = hold && st_b != MPCB_ST_SOLVED && st_b != MPCB_ST_ACCEPTABLE;
  const double* zb = keep ? w : z + (size_t)b * nz;
  double* xb = x0 + (size_t)b * NX;
  const double u[2] = {zb[0], zb[1]};
  double x[MPCB_NX_MAX] = {0, 0, 0, 0, 0, 0}, f[MPCB_NX_MAX];   // constant indices after unrolling: registers
#pragma unroll
  for (int q = 0; q < NX; ++q) x[q] = xb[q];
  model_rhs(c, x, u, f);
  if (c.integrator == MPCB_INT_RK4) {                                             // the plant follows the NLP's integrator: classical RK4 step, control held
    double k2[MPCB_NX_MAX], k3[MPCB_NX_MAX], k4[MPCB_NX_MAX], xt[MPCB_NX_MAX] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < NX; ++q) xt[q] = x[q] + 0.5 * T * f[q];
    model_rhs(c, xt, u, k2);
#pragma unroll
    for (int q = 0; q < NX; ++q) xt[q] = x[q] + 0.5 * T * k2[q];
    model_rhs(c, xt, u, k3);
#pragma unroll
    for (int q = 0; q < NX; ++q) xt[q] = x[q] + T * k3[q];
    model_rhs(c, xt, u, k4);
#pragma unroll
    for (int q = 0; q < NX; ++q) f[q] = (f[q] + 2.0 * k2[q] + 2.0 * k3[q] + k4[q]) / 6.0;
  }
#pragma unroll
  for (int q = 0; q < NX; ++q) xb[q] = x[q] + T * f[q];                          // st = x0 + T f(x0, u[0])  (RK4: T times the weighted slope)
  if (u_hist) { u_hist[((size_t)b * steps + step) * 2] = u[0]; u_hist[((size_t)b * steps + step) * 2 + 1] = u[1]; }
  if (x_hist) {
    double* h = x_hist + ((size_t)b * (steps + 1) + step + 1) * NX;
#pragma unroll
    for (int q = 0; q < NX; ++q) h[q] = xb[q];
  }
  // u <- [u[1:]; u[-1]],  x_f <- [x_f[1:]; x_f[-1]]   (in place when zb == w: ascending order reads ahead of the writes)
  for (int i = 0; i < N; ++i) { int s = (i + 1 < N) ? i + 1 : N - 1; const double a0 = zb[2 * s], a1 = zb[2 * s + 1]; w[2 * i] = a0; w[2 * i + 1] = a1; }
  for (int i = 0; i <= N; ++i) {
    int s = (i + 1 <= N) ? i + 1 : N;
    double t[NX];
#pragma unroll
    for (int q = 0; q < NX; ++q) t[q] = zb[2 * N + NX * s + q];
#pragma unroll
    for (int q = 0; q < NX; ++q) w[2 * N + NX * i + q] = t[q];
  }
  // obstacles move one step with constant velocity and heading (Obs_prediction.py:27-30)
  const int nmove = move_obs == 1 ? n_obs : move_obs == 2 ? (n_obs < 1 ? n_obs : 1) : 0;
  for (int j = 0; j < nmove; ++j) {
    double* o = obs + ((size_t)b * n_obs + j) * 6;
    o[0] += o[3] * cos(o[2]) * T; o[1] += o[3] * sin(o[2]) * T;
  }
}

// constant-velocity prediction of every obstacle over the horizon: [B, n_obs, 6] -> [B, n_obs, N+1, 6]
__global__ void mpcb_predict_obs(int total, int N, double T, const double* __restrict__ tgrid, const double* __restrict__ obs, double* __restrict__ traj) {
#pragma clang fp contract(off)   // next_x = x + v*cos(theta)*dt, two roundings per step as in Obs_prediction.py:27-28
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const double* o = obs + (size_t)i * 6;
  double x = o[0], y = o[1];
  const double vx = o[3] * cos(o[2]), vy = o[3] * sin(o[2]);
  for (int k = 0; k <= N; ++k) {
    double* t = traj + ((size_t)i * (N + 1) + k) * 6;
    t[0] = x; t[1] = y; t[2] = o[2]; t[3] = o[3]; t[4] = o[4]; t[5] = o[5];
    const double Tk = tgrid ? tgrid[k < N ? k : N - 1] : T;                      // node times of the time grid, or k * T
    x += vx * Tk; y += vy * Tk;
  }
}

// ---- scene generation: Philox4x32-10 (Salmon et al., SC'11), key = seed, counter = (scene index, draw block) --------------
struct SceneRng {
  uint32_t k0, k1, i0, i1, block, buf[4]; int have;
  __device__ SceneRng(uint64_t seed, uint64_t index) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)), i0((uint32_t)index), i1((uint32_t)(index >> 32)), block(0), have(0) {}
  __device__ void refill() {
    uint32_t c0 = i0, c1 = i1, c2 = block++, c3 = 0x4d504342u /* "MPCB" */, a = k0, b = k1;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0, h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
      const uint32_t n0 = h1 ^ c1 ^ a, n2 = h0 ^ c3 ^ b;
      c0 = n0; c1 = l1; c2 = n2; c3 = l0;
      a += 0x9E3779B9u; b += 0xBB67AE85u;
    }
    buf[0] = c0; buf[1] = c1; buf[2] = c2; buf[3] = c3; have = 4;
  }
  __device__ uint32_t next() { if (!have) refill(); return buf[--have]; }
  __device__ double uniform(double lo, double hi) {            // 53 random bits -> [0, 1)
    const uint64_t hi32 = next(), lo32 = next();
    const double u = (double)(((hi32 << 32) | lo32) >> 11) * (1.0 / 9007199254740992.0);
    return lo + (hi - lo) * u;
  }
  __device__ int choice(int n) { return (int)(next() % (uint32_t)n); }
};

// one thread per scene; whole-scene rejection sampling as mpc_motion_planning_amd/scenes.py (sample_c2 / sample_c3 / sample_c4): the same
// population as the host samplers draw.  The loop is bounded (the exit every thread reaches), but far beyond what any accepted
// configuration needs: the worst case, eight mutually separated C3 obstacles, accepts ~0.9 % of the draws, so 2^16 attempts
// fail with probability < 1e-250 (the 256 of abi 2 left ~10 % of such scenes with overlapping obstacles, unflagged)
__global__ __launch_bounds__(128) void mpcb_sample_kernel(const mpcb_config c, int kind, int B, uint64_t seed, uint64_t first,
                                                         double* __restrict__ x0, double* __restrict__ xs, double* __restrict__ obs) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  SceneRng g(seed, first + (uint64_t)b);
  const int no = c.n_obs, nx = c.model == MPCB_MODEL_DYN ? 6 : 4;
  double e[6] = {0, 0, 0, 0, 0, 0}, o[MPCB_NOBS_MAX][6];
  for (int attempt = 0; attempt < (1 << 16); ++attempt) {
    bool ok = true;
    if (kind == MPCB_SCENES_C4) {
      e[0] = g.uniform(0, 30); e[1] = g.uniform(-0.5, 4.5); e[2] = g.uniform(-0.05, 0.05); e[3] = g.uniform(8, 20); e[4] = 0; e[5] = 0;
      const double lanes[4] = {-3.5, 0.0, 3.5, 7.0};
      for (int j = 0; j < no; ++j) {
        o[j][0] = g.uniform(40, 200); o[j][1] = lanes[g.choice(4)] + g.uniform(-0.3, 0.3); o[j][2] = 0; o[j][3] = 0; o[j][4] = 0; o[j][5] = 0;
        const double dx = (e[0] - o[j][0]) / 4.0, dy = (e[1] - o[j][1]) / 1.0;              // dyn.py:240-243, fixed 4 x 1 semi-axes
        ok = ok && (dx * dx + dy * dy - 1.0 >= 1.5);
      }
    } else {
      e[0] = g.uniform(0, 30); e[1] = g.uniform(-0.5, 4.5); e[2] = g.uniform(-0.1, 0.1); e[3] = g.uniform(5, 25);
      for (int j = 0; j < no; ++j) {
        if (kind == MPCB_SCENES_C2) { o[j][0] = 50.0; o[j][1] = 3.5; o[j][2] = 0.0; o[j][3] = 8.0; }            // main_cbf_kin_c_sim.py:55
        else { o[j][0] = g.uniform(30, 120); o[j][1] = (g.choice(2) ? 3.5 : 0.0) + g.uniform(-0.3, 0.3); o[j][2] = 0.0; o[j][3] = g.uniform(5, 15); }
        o[j][4] = 4.8; o[j][5] = 1.8;
        const double sx = c.ego_hl + o[j][4] / 2 + c.safe_disl, sy = c.ego_hw + o[j][5] / 2 + c.safe_disw;     // kin.py:242-243
        const double dx = (e[0] - o[j][0]) / sx, dy = (e[1] - o[j][1]) / sy;
        ok = ok && (dx * dx + dy * dy - 1.0 >= 0.05);
        for (int i = 0; i < j; ++i) ok = ok && (fabs(o[i][0] - o[j][0]) > 12.0 || fabs(o[i][1] - o[j][1]) > 2.5);   // obstacles apart
      }
    }
    if (ok) break;
  }
  const double xs_kin[4] = {400.0, 3.5, 0.0, 30.0}, xs_dyn[6] = {600.0, 3.5, 0.0, 15.0, 0.0, 0.0};           // main_cbf_kin_c_sim.py:49, main_cbf_dyn_c_sim.py:48
  for (int q = 0; q < nx; ++q) { x0[(size_t)b * nx + q] = e[q]; xs[(size_t)b * nx + q] = nx == 6 ? xs_dyn[q] : xs_kin[q]; }
  for (int j = 0; j < no; ++j) for (int q = 0; q < 6; ++q) obs[((size_t)b * no + j) * 6 + q] = o[j][q];
}

// straight global path + preview window per instance (RefPathGenerator.py:9-59).  Path point i = (x_start + i * step, xs[1], xs[2], xs[3]),
// step = +-1 m towards xs[0]; M = number of points of np.arange(x_start, xs[0] + step, step).
__global__ __launch_bounds__(128) void mpcb_ref_window_kernel(int B, double x_start, const double* __restrict__ x0, const double* __restrict__ xs,
                                                             double T_horizon, double dt, int N_p, int32_t* __restrict__ last_idx, double* __restrict__ win) {
#pragma clang fp contract(off)
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double* e = x0 + (size_t)b * 4; const double* t = xs + (size_t)b * 4;
  const double step = t[0] > x_start ? 1.0 : -1.0;
  const int M = (int)ceil(((t[0] + step) - x_start) / step);                          // len(np.arange(start, stop, step))
  const double pv = 0.5 * e[3] + 0.5 * t[3];                                          // preview speed        :29
  const int pidx = (int)(pv * T_horizon / 1.0);                                       // int(preview_v * T_horizon / step_x), step_x = 1
  const int last = last_idx[b];
  const int lo = last - 5 > 0 ? last - 5 : 0, hi = last + pidx < M ? last + pidx : M;
  int mi = lo; double best = INFINITY;
  for (int i = lo; i < hi; ++i) {                                                     // first local minimum of the distance    :36-45
    const double dx = (x_start + i * step) - e[0], dy = t[1] - e[1];
    const double d = sqrt(dx * dx + dy * dy);
    if (d < best) { best = d; mi = i; } else break;
  }
  last_idx[b] = mi;
  const double lstep = ((double)(mi + pidx) - (double)mi) / (double)N_p;              // np.linspace(mi, mi + pidx, N_p + 1)
  for (int j = 0; j <= N_p; ++j) {
    double v = j == N_p ? (double)(mi + pidx) : (double)j * lstep + (double)mi;
    v = v < 0 ? 0 : v > (double)(M - 1) ? (double)(M - 1) : v;                        // np.clip(., 0, ref_len - 1).astype(int)
    const int idx = (int)v;
    double* w = win + ((size_t)b * (N_p + 1) + j) * 4;
    w[0] = x_start + idx * step; w[1] = t[1]; w[2] = t[2]; w[3] = t[3];
  }
}

}  // namespace

struct mpcb_handle {
  mpcb_config cfg;
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  int nx = 4, nz = 0, ng = 0;
  // scratch for the host-pointer entries
  void* d_buf = nullptr; size_t d_cap = 0;
  // timing: a ring of event pairs created once; a pair is harvested (synchronised, accumulated) before it is reused
  static constexpr int EV_RING = 256;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;   // EV_RING pairs, created on first use
  int ev_head = 0, ev_pending = 0;                     // next pair to record into; pairs recorded and not yet harvested
  int launches = 0; double total_ms = 0, last_ms = 0;
  // Launch lanes (mpcb_set_inflight): lane 0 IS the handle's stream; lanes 1..K-1 own a stream each.  An asynchronous
  // mpcb_solve_device goes to the next lane(s) in turn, so that consecutive solves of one handle overlap on the GPU (the next
  // launch fills the SIMDs the tail of the running one leaves idle) without the caller juggling handles.  Every lane has its own
  // two-pass scratch: hand-over records [B][WK_SIZE] and, when the caller passes no status array, the status column the passes
  // communicate through.  `done` marks the end of the lane's last launch; join_lanes() makes the handle's stream wait for it.
  struct Lane { hipStream_t stream = nullptr; double* d_work = nullptr; int32_t* d_st_own = nullptr; int work_cap = 0; hipEvent_t done = nullptr; bool busy = false; };
  std::vector<Lane> lanes;                             // lanes[0].stream == stream
  int next_lane = 0;
  hipEvent_t ev_fork = nullptr;                        // "everything queued so far on the handle's stream", awaited by a lane before it launches
  std::vector<std::vector<hipEvent_t>> marks;          // mpcb_event_record: per slot one event per lane stream
  std::vector<double> tgrid_host;                      // host copy of the time grid (handed to the peers of a device group)
  // multi-GPU: (a) this handle is rank `rank` of `world` processes (mpcb_comm_init_rank), or (b) it leads a group of
  // `peers.size()` devices of this process (mpcb_set_devices; peers[0] is a sibling handle on the leader's own device)
  ncclComm_t comm = nullptr; int world = 1, rank = 0;
  std::vector<mpcb_handle*> peers;
  double* d_gather = nullptr; size_t gather_cap = 0;   // (b): [world, longest shard, nz] all-gather target on this device
  double* d_red = nullptr;                             // small device buffer for mpcb_allreduce
  double* d_tgrid = nullptr; double T0 = 0;            // mpcb_set_time_grid: [N] step lengths on the device, T_0 for the plant step
  hipEvent_t ev_sync = nullptr;                        // mpcb_stream_wait: marks "everything queued so far on this stream"
  size_t gathered_rows = 0;                            // (b): rows per shard block of the last gather, B of that solve
  int64_t gathered_B = 0;
};

namespace {

int fail(mpcb_handle* h, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  if (h) h->err = buf; else g_create_error = buf;
  return code;
}
#define HIP_TRY(h, expr)                                                                         \
  do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(h, MPCB_E_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)

// librccl, loaded on first use so that a process that never forms a group does not pay for (or depend on) it
struct Rccl {
  decltype(&ncclGetUniqueId) GetUniqueId; decltype(&ncclCommInitRank) CommInitRank; decltype(&ncclCommInitAll) CommInitAll;
  decltype(&ncclCommDestroy) CommDestroy; decltype(&ncclAllGather) AllGather; decltype(&ncclAllReduce) AllReduce;
  decltype(&ncclGroupStart) GroupStart; decltype(&ncclGroupEnd) GroupEnd; decltype(&ncclGetErrorString) GetErrorString;
};
std::string g_rccl_error;
const Rccl* rccl() {
  static Rccl R; static int state = 0;                       // 1 loaded, -1 failed
  static std::once_flag once;                                // (handles are driven from several host threads)
  std::call_once(once, [] {
    // MPCB_RCCL_LIB names the library to load instead of the default search (a test points it at a missing file to exercise the error path)
    const char* forced = std::getenv("MPCB_RCCL_LIB");
    void* so = nullptr;
    if (forced && *forced) so = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
    else {
      so = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!so) so = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (!so) so = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    }
    if (!so) { const char* e = dlerror(); g_rccl_error = e ? e : "dlopen failed"; state = -1; return; }
    bool ok = true;
#define MPCB_SYM(field, name) do { R.field = (decltype(R.field))dlsym(so, name); if (!R.field) { ok = false; g_rccl_error = std::string("missing symbol ") + name; } } while (0)
    MPCB_SYM(GetUniqueId, "ncclGetUniqueId"); MPCB_SYM(CommInitRank, "ncclCommInitRank"); MPCB_SYM(CommInitAll, "ncclCommInitAll");
    MPCB_SYM(CommDestroy, "ncclCommDestroy"); MPCB_SYM(AllGather, "ncclAllGather"); MPCB_SYM(AllReduce, "ncclAllReduce");
    MPCB_SYM(GroupStart, "ncclGroupStart"); MPCB_SYM(GroupEnd, "ncclGroupEnd"); MPCB_SYM(GetErrorString, "ncclGetErrorString");
#undef MPCB_SYM
    state = ok ? 1 : -1;
  });
  return state == 1 ? &R : nullptr;
}
#define NCCL_TRY(h, R, expr)                                                                     \
  do { ncclResult_t e_ = (expr); if (e_ != ncclSuccess) return fail(h, MPCB_E_DEVICE, "%s: %s", #expr, (R)->GetErrorString(e_)); } while (0)

int nx_of(const mpcb_config& c) { return c.model == MPCB_MODEL_DYN ? 6 : 4; }
int n_rate(const mpcb_config& c) {
  int n = 0;
  for (int i = 0; i < 2; ++i) if (std::isfinite(c.du_lo[i]) || std::isfinite(c.du_hi[i])) ++n;
  return n;
}

int check_cfg(mpcb_handle* h, const mpcb_config* c) {
  if (!c) return fail(h, MPCB_E_INVALID, "config is NULL");
  if (c->struct_size != sizeof(mpcb_config))
    return fail(h, MPCB_E_INVALID, "mpcb_config.struct_size = %u, this library expects %zu", c->struct_size, sizeof(mpcb_config));
  if (c->model != MPCB_MODEL_KIN && c->model != MPCB_MODEL_DYN) return fail(h, MPCB_E_INVALID, "unknown model %d", c->model);
  if (c->N < 1 || c->N > MPCB_N_MAX) return fail(h, MPCB_E_INVALID, "N = %d outside 1..%d", c->N, MPCB_N_MAX);
  if (c->n_obs < 0 || c->n_obs > MPCB_NOBS_MAX) return fail(h, MPCB_E_INVALID, "n_obs = %d outside 0..%d", c->n_obs, MPCB_NOBS_MAX);
  if (!(c->T > 0) || !(c->tol > 0) || c->max_iter < 0 || !(c->mu_init > 0)) return fail(h, MPCB_E_INVALID, "T, tol, mu_init must be > 0 and max_iter >= 0");
  if (!(c->veh_l > 0)) return fail(h, MPCB_E_INVALID, "veh_l must be > 0");
  if (!(c->dual_inf_tol > 0) || !(c->constr_viol_tol > 0) || !(c->compl_inf_tol > 0) || c->acceptable_iter < 0 || !(c->acceptable_tol > 0) ||
      !(c->acceptable_obj_change_tol >= 0) || !(c->acceptable_constr_viol_tol > 0) || !(c->acceptable_dual_inf_tol > 0) || !(c->acceptable_compl_inf_tol > 0))
    return fail(h, MPCB_E_INVALID, "termination tolerances must be > 0 (mpcb_default_config sets IPOPT's defaults) and acceptable_iter >= 0");
  for (int i = 0; i < 2; ++i) if (!(c->R[i] > 0)) return fail(h, MPCB_E_INVALID, "R must be positive");
  if (c->obs_mode == MPCB_OBS_DCBF && !(c->gamma > 0.0 && c->gamma <= 1.0 + 1e-12))
    return fail(h, MPCB_E_INVALID, "discrete-CBF rows need 0 < gamma <= 1 (kin.py:235 uses 1.0), got %g", c->gamma);
  if (c->obs_mode == MPCB_OBS_DCBF && c->gamma < 1.0 - 1e-12 && (c->model != MPCB_MODEL_KIN || c->obs_terminal))
    return fail(h, MPCB_E_UNSUPPORTED, "general-gamma discrete-CBF rows are implemented for the kinematic model with rows i = 0..N-1");
  if (c->obs_mode != MPCB_OBS_KEEPOUT && c->obs_mode != MPCB_OBS_DCBF) return fail(h, MPCB_E_INVALID, "unknown obs_mode %d", c->obs_mode);
  if (c->obs_mode == MPCB_OBS_DCBF && c->obs_terminal)
    return fail(h, MPCB_E_UNSUPPORTED, "discrete-CBF rows exist for i = 0..N-1 only (row N would need X_{N+1}, kin.py:236-248): obs_terminal must be 0");
  if (c->mu_strategy != MPCB_MU_MONOTONE) return fail(h, MPCB_E_INVALID, "unknown mu_strategy %d (MPCB_MU_MONOTONE is the only one)", c->mu_strategy);
  if (c->integrator != MPCB_INT_EULER && c->integrator != MPCB_INT_RK4) return fail(h, MPCB_E_INVALID, "unknown integrator %d", c->integrator);
  if (c->integrator == MPCB_INT_RK4 && (c->model != MPCB_MODEL_KIN || c->n_obs > 3 || (c->obs_mode == MPCB_OBS_DCBF && c->gamma < 1.0 - 1e-12)))
    return fail(h, MPCB_E_UNSUPPORTED, "MPCB_INT_RK4 is built for the kinematic model with up to 3 obstacles and keep-out / gamma = 1 rows");
  if (c->restoration != 0 && c->restoration != 1) return fail(h, MPCB_E_INVALID, "restoration must be 0 or 1");
  if (c->second_start < 0 || c->second_start > 3) return fail(h, MPCB_E_INVALID, "second_start must be 0, 1, 2 or 3");
  if (!(c->start_steer >= 0.0 && c->start_steer <= 0.5)) return fail(h, MPCB_E_INVALID, "start_steer must be in [0, 0.5] rad");
  if (std::isfinite(c->x_lo[0]) || std::isfinite(c->x_hi[0]) || std::isfinite(c->x_lo[2]) || std::isfinite(c->x_hi[2]))
    return fail(h, MPCB_E_UNSUPPORTED, "state boxes are supported on y, vx (and vy for the dynamic model) (kin.py:97-105, dyn.py:97-110)");
  if (c->model == MPCB_MODEL_KIN) {
    // kinematic kernel: rate row on the steering angle only, rows not interleaved
    if (std::isfinite(c->du_lo[1]) || std::isfinite(c->du_hi[1]))
      return fail(h, MPCB_E_UNSUPPORTED, "kinematic kernel: rate rows are supported on the steering angle only (kin.py:216-217)");
    if (c->rate_interleaved) return fail(h, MPCB_E_UNSUPPORTED, "kinematic kernel: rate rows form one block (kin.py:211-216)");
  } else {
    if (std::isfinite(c->x_lo[5]) || std::isfinite(c->x_hi[5])) return fail(h, MPCB_E_UNSUPPORTED, "dynamic kernel: no box on the yaw rate (dyn.py:109-110)");
    if (!(c->veh_m > 0) || !(c->veh_Iz > 0) || !(c->aopt_f > 0) || !(c->aopt_r > 0)) return fail(h, MPCB_E_INVALID, "vehicle / tyre parameters must be positive");
    if (!(c->x_lo[3] >= 0)) return fail(h, MPCB_E_INVALID, "dynamic model needs vx_min >= 0 (the tyre model divides by vx, dyn.py:156-157)");
  }
  return MPCB_OK;
}

bool is_gen(const mpcb_config& c) { return c.model == MPCB_MODEL_KIN && c.obs_mode == MPCB_OBS_DCBF && c.gamma < 1.0 - 1e-12 && c.n_obs > 0; }
bool is_rk4(const mpcb_config& c) { return c.model == MPCB_MODEL_KIN && c.integrator == MPCB_INT_RK4; }
bool wide_table(const mpcb_config& c) { return is_gen(c) || is_rk4(c); }       // four more rows in the entry table of the kinematic kernels

size_t lds_bytes(const mpcb_config& c, int nz) {
  return (size_t)(c.model == MPCB_MODEL_DYN ? mpcbk::layout_dyn(c.N, false, mpcbk::obs_in_lds(mpcbk::obs_capacity_dyn(c.n_obs))).total
                                             : mpcbk::layout_kin(c.N, nz, false, mpcbk::obs_in_lds(mpcbk::obs_capacity_kin(c.n_obs, is_gen(c))), wide_table(c)).total) * sizeof(double);
}

// oldest recorded pair -> total_ms / last_ms / launches
int harvest_one(mpcb_handle* h) {
  const int i = (h->ev_head - h->ev_pending + 2 * mpcb_handle::EV_RING) % mpcb_handle::EV_RING;
  HIP_TRY(h, hipEventSynchronize(h->ev[i].second));
  float ms = 0;
  HIP_TRY(h, hipEventElapsedTime(&ms, h->ev[i].first, h->ev[i].second));
  h->total_ms += ms; h->last_ms = ms; ++h->launches; --h->ev_pending;
  return MPCB_OK;
}

int collect_timing(mpcb_handle* h) {
  while (h->ev_pending > 0) { int rc = harvest_one(h); if (rc != MPCB_OK) return rc; }
  return MPCB_OK;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, device), not of a handle: two handles of one process can
// share an instantiation with different N.  The limit is therefore kept process-wide and only ever raised.
std::mutex g_lds_mutex;
std::map<std::pair<const void*, int>, size_t> g_lds_limit;

template <class K>
int launch_kernel(mpcb_handle* h, hipStream_t stream, K kernel, const MpcbKArgs& a, size_t lds) {
  if (lds > 48 * 1024) {
    std::lock_guard<std::mutex> lock(g_lds_mutex);
    size_t& cur = g_lds_limit[{(const void*)kernel, h->device}];
    if (lds > cur) {
      HIP_TRY(h, hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      cur = lds;
    }
  }
  hipLaunchKernelGGL(kernel, dim3(a.B), dim3(64), lds, stream, a);
  return MPCB_OK;
}

// the handle's stream waits for every lane that has a launch outstanding: called by everything that may consume the results of,
// or overwrite the inputs of, asynchronous solves (downloads, uploads, collectives, the closed loop, mpcb_sync ...)
int join_lanes(mpcb_handle* h) {
  for (size_t i = 1; i < h->lanes.size(); ++i) {
    auto& L = h->lanes[i];
    if (L.busy) { HIP_TRY(h, hipStreamWaitEvent(h->stream, L.done, 0)); L.busy = false; }
  }
  return MPCB_OK;
}

int ensure_lanes(mpcb_handle* h, int k) {
  if (h->lanes.empty()) { h->lanes.resize(1); h->lanes[0].stream = h->stream; }
  while ((int)h->lanes.size() > k) {
    auto& L = h->lanes.back();
    if (L.d_work) (void)hipFree(L.d_work);
    if (L.d_st_own) (void)hipFree(L.d_st_own);
    if (L.done) (void)hipEventDestroy(L.done);
    if (L.stream && L.stream != h->stream) (void)hipStreamDestroy(L.stream);
    h->lanes.pop_back();
  }
  while ((int)h->lanes.size() < k) {
    mpcb_handle::Lane L;
    HIP_TRY(h, hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
    HIP_TRY(h, hipEventCreateWithFlags(&L.done, hipEventDisableTiming));
    h->lanes.push_back(L);
  }
  if (!h->ev_fork) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
  h->next_lane = 0;
  return MPCB_OK;
}

bool second_pass(const mpcb_config& c) { return c.second_start != 0 && c.init_rollout != 0; }      // a second start exists only after a roll-out start
bool multi_pass(const mpcb_config& c) { return c.restoration != 0 || second_pass(c); }
// cfg.second_start = 3: the order of the passes follows the kind of start — a cold start (z0 = NULL) behaves as 1, a solve with a start
// vector (a warm start, every step of a closed loop) as 2
int second_mode(const mpcb_config& c, const void* z0) { return c.second_start == 3 ? (z0 ? 2 : 1) : c.second_start; }

// both passes of one solve on lane `lane_id` (0 = the handle's own stream)
int launch_solve(mpcb_handle* h, const MpcbKArgs& a_in, int lane_id = 0) {
  MpcbKArgs a = a_in;
  const size_t lds = lds_bytes(h->cfg, h->nz);
  if (lds > 160 * 1024) return fail(h, MPCB_E_UNSUPPORTED, "LDS need %zu B exceeds 160 KiB", lds);
  if (a.B == 0) return MPCB_OK;
  if (h->lanes.empty()) { int rc = ensure_lanes(h, 1); if (rc != MPCB_OK) return rc; }
  mpcb_handle::Lane& L = h->lanes[lane_id];
  const hipStream_t stream = L.stream;
  if (lane_id > 0) {                   // work queued earlier on the handle's stream (uploads, scene sampling) happens before this launch
    HIP_TRY(h, hipEventRecord(h->ev_fork, h->stream));
    HIP_TRY(h, hipStreamWaitEvent(stream, h->ev_fork, 0));
  }
  a.pass = MPCB_PASS_FIRST; a.work = nullptr;
  if (multi_pass(h->cfg)) {
    if (a.B > L.work_cap) {       // grows with the largest batch seen (first call of a given size only)
      HIP_TRY(h, hipStreamSynchronize(stream));
      if (L.d_work) HIP_TRY(h, hipFree(L.d_work));
      if (L.d_st_own) HIP_TRY(h, hipFree(L.d_st_own));
      L.d_work = nullptr; L.d_st_own = nullptr; L.work_cap = 0;
      HIP_TRY(h, hipMalloc(&L.d_work, (size_t)a.B * mpcbk::WK_SIZE * sizeof(double)));
      HIP_TRY(h, hipMalloc(&L.d_st_own, (size_t)a.B * sizeof(int32_t)));
      L.work_cap = a.B;
    }
    a.work = L.d_work;
    if (!a.status) {                 // the passes communicate through the status column
      if (a.st_stride != 1) return fail(h, MPCB_E_INVALID, "a strided iteration history needs a status history");
      a.status = L.d_st_own;
    }
  }
  if (h->ev.empty()) {
    h->ev.resize(mpcb_handle::EV_RING);
    for (auto& p : h->ev) { HIP_TRY(h, hipEventCreate(&p.first)); HIP_TRY(h, hipEventCreate(&p.second)); }
  }
  if (h->ev_pending == mpcb_handle::EV_RING) { int rc = harvest_one(h); if (rc != MPCB_OK) return rc; }
  auto& evp = h->ev[h->ev_head];
  HIP_TRY(h, hipEventRecord(evp.first, stream));
  const int n = h->cfg.n_obs;
  int rc = MPCB_OK;
  auto lean_pass = [&](int pass) -> int {         // the lean main-phase kernel over the whole grid
    a.pass = pass;
    if (h->cfg.model == MPCB_MODEL_DYN) {
      if (n <= 1) rc = launch_kernel(h, stream, mpcb_kernel_dyn<1>, a, lds);
      else if (n <= 3) rc = launch_kernel(h, stream, mpcb_kernel_dyn<3>, a, lds);
      else if (n <= 5) rc = launch_kernel(h, stream, mpcb_kernel_dyn<5>, a, lds);
      else rc = launch_kernel(h, stream, mpcb_kernel_dyn<8>, a, lds);
    } else if (h->cfg.obs_mode == MPCB_OBS_DCBF && h->cfg.gamma < 1.0 - 1e-12 && n > 0) {   // general-gamma CBF rows
      if (n == 1) rc = launch_kernel(h, stream, mpcb_kernel_kin<1, true>, a, lds);
      else if (n <= 3) rc = launch_kernel(h, stream, mpcb_kernel_kin<3, true>, a, lds);
      else rc = launch_kernel(h, stream, mpcb_kernel_kin<8, true>, a, lds);
    } else if (is_rk4(h->cfg)) {                                                               // Runge-Kutta shooting rows
      if (n == 0) rc = launch_kernel(h, stream, mpcb_kernel_kin<0, false, true>, a, lds);
      else if (n == 1) rc = launch_kernel(h, stream, mpcb_kernel_kin<1, false, true>, a, lds);
      else rc = launch_kernel(h, stream, mpcb_kernel_kin<3, false, true>, a, lds);
    } else if (n == 0) rc = launch_kernel(h, stream, mpcb_kernel_kin<0>, a, lds);
    else if (n == 1) rc = launch_kernel(h, stream, mpcb_kernel_kin<1>, a, lds);
    else if (n <= 3) rc = launch_kernel(h, stream, mpcb_kernel_kin<3>, a, lds);
    else if (n <= 5) rc = launch_kernel(h, stream, mpcb_kernel_kin<5>, a, lds);
    else rc = launch_kernel(h, stream, mpcb_kernel_kin<8>, a, lds);
    if (rc != MPCB_OK) return rc;
    HIP_TRY(h, hipGetLastError());
    return MPCB_OK;
  };
  auto resto_pass = [&]() -> int {                // the restoration-pass kernel: instances that ended the pass before it with MPCB_ST_NEEDS_RESTO continue, the others return
    a.pass = MPCB_PASS_RESTO;
    const bool dyn = h->cfg.model == MPCB_MODEL_DYN;
    const size_t lds2 = (size_t)(dyn ? mpcbk::layout_dyn(h->cfg.N, true, mpcbk::obs_in_lds(mpcbk::obs_capacity_dyn(n))).total
                                     : mpcbk::layout_kin(h->cfg.N, h->nz, true, mpcbk::obs_in_lds(mpcbk::obs_capacity_kin(n, is_gen(h->cfg))), wide_table(h->cfg)).total) * sizeof(double);
    if (lds2 > 160 * 1024) return fail(h, MPCB_E_UNSUPPORTED, "LDS need %zu B exceeds 160 KiB", lds2);
    const bool gen = h->cfg.obs_mode == MPCB_OBS_DCBF && h->cfg.gamma < 1.0 - 1e-12 && n > 0;
    if (dyn) {
      if (n <= 1) rc = launch_kernel(h, stream, mpcb_kernel_dyn_resto<1>, a, lds2);
      else if (n <= 3) rc = launch_kernel(h, stream, mpcb_kernel_dyn_resto<3>, a, lds2);
      else if (n <= 5) rc = launch_kernel(h, stream, mpcb_kernel_dyn_resto<5>, a, lds2);
      else rc = launch_kernel(h, stream, mpcb_kernel_dyn_resto<8>, a, lds2);
    } else if (gen) {
      if (n == 1) rc = launch_kernel(h, stream, mpcb_kernel_kin_resto<1, true>, a, lds2);
      else if (n <= 3) rc = launch_kernel(h, stream, mpcb_kernel_kin_resto<3, true>, a, lds2);
      else rc = launch_kernel(h, stream, mpcb_kernel_kin_resto<8, true>, a, lds2);
    } else if (is_rk4(h->cfg)) {
      if (n == 0) rc = launch_kernel(h, stream, mpcb_kernel_kin_resto<0, false, true>, a, lds2);
      else if (n == 1) rc = launch_kernel(h, stream, mpcb_kernel_kin_resto<1, false, true>, a, lds2);
      else rc = launch_kernel(h, stream, mpcb_kernel_kin_resto<3, false, true>, a, lds2);
    } else if (n == 0) rc = launch_kernel(h, stream, mpcb_kernel_kin_resto<0>, a, lds2);
    else if (n == 1) rc = launch_kernel(h, stream, mpcb_kernel_kin_resto<1>, a, lds2);
    else if (n <= 3) rc = launch_kernel(h, stream, mpcb_kernel_kin_resto<3>, a, lds2);
    else if (n <= 5) rc = launch_kernel(h, stream, mpcb_kernel_kin_resto<5>, a, lds2);
    else rc = launch_kernel(h, stream, mpcb_kernel_kin_resto<8>, a, lds2);
    if (rc != MPCB_OK) return rc;
    HIP_TRY(h, hipGetLastError());
    return MPCB_OK;
  };
  // first attempt from the caller's start, its restoration pass; with a second start (cfg.second_start after a roll-out start) the
  // lean kernel once more over the same grid, where only the instances whose first attempt did not succeed run from z = 0, and the
  // restoration pass of that attempt
  const int ss = second_mode(h->cfg, a.z0);
  rc = lean_pass(MPCB_PASS_FIRST);
  // (second start of kind 1: the first attempt's restoration pass is skipped — its instances go straight to the second start)
  if (rc == MPCB_OK && h->cfg.restoration && !(second_pass(h->cfg) && ss == 1)) rc = resto_pass();
  if (rc == MPCB_OK && second_pass(h->cfg)) {
    // (kind 1 on an instantiation that fuses: the second attempt ran inside the first launch)
    const bool fused = ss == 1 && (h->cfg.model == MPCB_MODEL_DYN
                                       ? ((n <= 1 && mpcb_dyn_fuses<1>) || (n > 1 && n <= 3 && mpcb_dyn_fuses<3>))
                                       : (!is_gen(h->cfg) && !is_rk4(h->cfg) && ((n == 0 && mpcb_kin_fuses<0, false, false>) || (n == 1 && mpcb_kin_fuses<1, false, false>) ||
                                                                                   (n > 1 && n <= 3 && mpcb_kin_fuses<3, false, false>))));
    if (!fused) rc = lean_pass(MPCB_PASS_SECOND);
    if (rc == MPCB_OK && h->cfg.restoration) rc = resto_pass();
  }
  if (rc != MPCB_OK) return rc;
  HIP_TRY(h, hipEventRecord(evp.second, stream));
  h->ev_head = (h->ev_head + 1) % mpcb_handle::EV_RING; ++h->ev_pending;
  if (lane_id > 0) { HIP_TRY(h, hipEventRecord(L.done, stream)); L.busy = true; }
  return MPCB_OK;
}

int ensure_scratch(mpcb_handle* h, size_t bytes) {
  if (bytes <= h->d_cap) return MPCB_OK;
  if (h->d_buf) { HIP_TRY(h, hipFree(h->d_buf)); h->d_buf = nullptr; h->d_cap = 0; }
  HIP_TRY(h, hipMalloc(&h->d_buf, bytes));
  h->d_cap = bytes;
  return MPCB_OK;
}

// Sub-allocation of the handle's scratch buffer.  The SAME sequence of take() calls runs twice: first with base = NULL to
// learn the size (so the size formula cannot drift from the carving), then with the allocated base.
struct Carve {
  char* base; size_t off = 0;
  template <class T> T* take(size_t n) { off = (off + 255) & ~size_t(255); T* p = base ? (T*)(base + off) : nullptr; off += n * sizeof(T); return p; }
  size_t bytes() const { return off + 256; }
};

// one launch of the solve over B instances, everything resident on the handle's device, asynchronous on its stream
int solve_on_device(mpcb_handle* h, int32_t B, const double* d_x0, const double* d_xs, const double* d_obs, int32_t obs_kind,
                    const double* d_z0, double* d_z, double* d_obj, int32_t* d_status, int32_t* d_iters, int32_t st_stride,
                    double* d_kkt, double* d_lam_g, double* d_lam_x, int lane_id = 0) {
  if (B < 0 || !d_x0 || !d_xs || !d_z) return fail(h, MPCB_E_INVALID, "B < 0 or a required pointer is NULL");
  if (h->cfg.n_obs > 0 && !d_obs) return fail(h, MPCB_E_INVALID, "n_obs = %d but obs is NULL", h->cfg.n_obs);
  if (obs_kind != MPCB_OBSIN_STATIC && obs_kind != MPCB_OBSIN_PREDICTED) return fail(h, MPCB_E_INVALID, "unknown obs_kind %d", obs_kind);
  HIP_TRY(h, hipSetDevice(h->device));
  MpcbKArgs a;
  a.cfg = h->cfg; a.B = B; a.nz = h->nz; a.ng = h->ng; a.obs_kind = obs_kind;
  a.want_mult = (d_lam_g || d_lam_x) ? 1 : 0; a.trace_instance = -1; a.trace = nullptr; a.st_stride = st_stride; a.tgrid = h->d_tgrid;
  a.x0 = d_x0; a.xs = d_xs; a.obs = d_obs; a.z0 = d_z0;
  a.z = d_z; a.obj = d_obj; a.kkt = d_kkt; a.lam_g = d_lam_g; a.lam_x = d_lam_x; a.status = d_status; a.iters = d_iters;
  return launch_solve(h, a, lane_id);
}

// One asynchronous solve on the next lane in turn: call k on lane k mod K (K = 1: the handle's stream).  Up to K consecutive calls
// are then in flight together, each a full-depth launch that fills the SIMDs the tails of the others leave idle; calls k and k + K
// share a lane and are ordered.  Measured alternatives (C2, 4096 instances, MI355X): cutting every call into K chunks, chunk c on
// lane c: 1.11 M solves/s against 1.48 M (only one batch's worth of workgroups is ever queued); cutting a lone synchronous call
// into chunks: 5.5 ms per call against 4.2 ms for the single launch — so the host-pointer entries stay one launch on lane 0.
int solve_next_lane(mpcb_handle* h, int32_t B, const double* d_x0, const double* d_xs, const double* d_obs, int32_t obs_kind,
                    const double* d_z0, double* d_z, double* d_obj, int32_t* d_status, int32_t* d_iters, double* d_kkt,
                    double* d_lam_g, double* d_lam_x) {
  const int K = h->lanes.empty() ? 1 : (int)h->lanes.size();
  const int lane = K == 1 ? 0 : h->next_lane;
  if (K > 1) h->next_lane = (h->next_lane + 1) % K;
  return solve_on_device(h, B, d_x0, d_xs, d_obs, obs_kind, d_z0, d_z, d_obj, d_status, d_iters, 1, d_kkt, d_lam_g, d_lam_x, lane);
}

}  // namespace

extern "C" {

const char* mpcb_version(void) { return "mpcbatch 0.3 (gfx950, abi 3)"; }

int mpcb_default_config(mpcb_config* cfg, int32_t model, int32_t N, double T) {
  if (!cfg || (model != MPCB_MODEL_KIN && model != MPCB_MODEL_DYN)) return MPCB_E_INVALID;
  mpcb_config c;
  std::memset(&c, 0, sizeof c);
  c.struct_size = sizeof(mpcb_config);
  c.model = model; c.N = N; c.T = T; c.gamma = 1.0;
  c.obs_mode = MPCB_OBS_KEEPOUT; c.max_iter = 100;                       // kin.py:252
  c.mu_strategy = MPCB_MU_MONOTONE; c.init_rollout = 1; c.integrator = MPCB_INT_EULER; c.restoration = 1;
  const double rad = M_PI / 180.0;
  for (int i = 0; i < MPCB_NX_MAX; ++i) { c.x_lo[i] = -INF; c.x_hi[i] = INF; }
  // mpc_parameters.yaml: kinematics_constraints / dynamics_constraints / vehicle_params / tire_params
  c.u_lo[0] = -35.0 * rad; c.u_hi[0] = 35.0 * rad; c.u_lo[1] = -3.0; c.u_hi[1] = 3.0;
  c.x_lo[1] = -1.0; c.x_hi[1] = 5.0; c.x_lo[3] = 0.0; c.x_hi[3] = 40.0;
  c.du_lo[0] = -5.0 * rad * T; c.du_hi[0] = 5.0 * rad * T; c.du_lo[1] = -INF; c.du_hi[1] = INF;
  c.veh_l = 2.6; c.ego_hl = 4.8 / 2; c.ego_hw = 1.8 / 2; c.safe_disl = 1.0; c.safe_disw = 0.5;
  c.veh_m = 1575.0; c.veh_lf = 1.2; c.veh_lr = 1.6; c.veh_Iz = 2875.0;
  c.aopt_f = 0.3490658503988659; c.aopt_r = 0.19198621771937624;
  c.Fymax_f = -50000.0 * c.aopt_f / 2; c.Fymax_r = -50000.0 * c.aopt_r / 2;
  if (model == MPCB_MODEL_KIN) {
    const double Q[4] = {1e1, 1e5, 3e5, 1e4};                             // kin.py:168-172
    for (int i = 0; i < 4; ++i) c.Q[i] = Q[i];
    c.R[0] = c.R[1] = 1e4; c.DR[0] = 1e5; c.DR[1] = 1e2;                  // kin.py:179-184
    c.du0_cost = 1; c.obs_terminal = 0; c.obs_hmin = 0.0; c.rate_interleaved = 0;
  } else {
    const double Q[6] = {10, 1e5, 1e3, 1e3, 1, 1};                        // dyn.py:189-195
    for (int i = 0; i < 6; ++i) c.Q[i] = Q[i];
    c.R[0] = c.R[1] = 1e3; c.DR[0] = 5e3; c.DR[1] = 5e2;                  // dyn.py:204-209
    c.x_lo[4] = -5.0; c.x_hi[4] = 5.0;
    c.du_lo[1] = -3.0 * T; c.du_hi[1] = 1.5 * T;
    c.du0_cost = 0; c.obs_terminal = 1; c.obs_hmin = 1.0; c.obs_sx_fixed = 4.0; c.obs_sy_fixed = 1.0; c.rate_interleaved = 1;
  }
  // IPOPT defaults; mu_init is raised from IPOPT's 0.1 because the roll-out start is already dynamics-feasible
  c.tol = 1e-8; c.mu_init = 10.0; c.bound_push = 0.01; c.bound_frac = 0.01; c.bound_relax = 1e-8; c.max_gradient = 100.0;
  // termination: IPOPT's defaults, and the two options the reference sets (kin.py:252-253)
  c.dual_inf_tol = 1.0; c.constr_viol_tol = 1e-4; c.compl_inf_tol = 1e-4;
  c.acceptable_tol = 1e-8; c.acceptable_obj_change_tol = 1e-6; c.acceptable_iter = 15;
  c.acceptable_constr_viol_tol = 1e-2; c.acceptable_dual_inf_tol = 1e10; c.acceptable_compl_inf_tol = 1e-2;
  c.second_start = 3;
  c.start_steer = 0.03;
  *cfg = c;
  return MPCB_OK;
}

int mpcb_dims(const mpcb_config* c, int32_t* nx, int32_t* nz, int32_t* ng) {
  if (!c) return MPCB_E_INVALID;
  const int n = nx_of(*c);
  if (nx) *nx = n;
  if (nz) *nz = 2 * c->N + n * (c->N + 1);
  if (ng) *ng = n * (c->N + 1) + n_rate(*c) * (c->N - 1) + c->n_obs * (c->obs_terminal ? c->N + 1 : c->N);
  return MPCB_OK;
}

int mpcb_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int mpcb_create(const mpcb_config* cfg, int32_t device, mpcb_handle** out) {
  if (!out) return fail(nullptr, MPCB_E_INVALID, "out is NULL");
  *out = nullptr;
  int rc = check_cfg(nullptr, cfg);
  if (rc != MPCB_OK) return rc;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) return fail(nullptr, MPCB_E_DEVICE, "no HIP device available (%s); libmpcbatch has no CPU path", e != hipSuccess ? hipGetErrorString(e) : "0 devices");
  if (device < 0 || device >= ndev) return fail(nullptr, MPCB_E_INVALID, "device %d outside 0..%d", device, ndev - 1);
  mpcb_handle* h = new mpcb_handle();
  h->cfg = *cfg; h->device = device;
  int nx, nz, ng; mpcb_dims(cfg, &nx, &nz, &ng);
  h->nx = nx; h->nz = nz; h->ng = ng;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
    delete h;
    return fail(nullptr, MPCB_E_DEVICE, "cannot create a stream on device %d", device);
  }
  *out = h;
  return MPCB_OK;
}

int mpcb_destroy(mpcb_handle* h) {
  if (!h) return MPCB_OK;
  (void)hipSetDevice(h->device);
  collect_timing(h);
  for (auto& p : h->ev) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
  if (h->d_buf) (void)hipFree(h->d_buf);
  for (auto* p : h->peers) mpcb_destroy(p);
  if (h->comm) { const Rccl* R = rccl(); if (R) (void)R->CommDestroy(h->comm); }
  if (h->d_gather) (void)hipFree(h->d_gather);
  if (h->d_red) (void)hipFree(h->d_red);
  if (h->ev_sync) (void)hipEventDestroy(h->ev_sync);
  if (h->d_tgrid) (void)hipFree(h->d_tgrid);
  for (auto& L : h->lanes) {
    if (L.stream && L.stream != h->stream) { (void)hipStreamSynchronize(L.stream); (void)hipStreamDestroy(L.stream); }
    if (L.d_work) (void)hipFree(L.d_work);
    if (L.d_st_own) (void)hipFree(L.d_st_own);
    if (L.done) (void)hipEventDestroy(L.done);
  }
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  for (auto& m : h->marks) for (auto e : m) (void)hipEventDestroy(e);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return MPCB_OK;
}

const char* mpcb_last_error(const mpcb_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int mpcb_set_bounds(mpcb_handle* h, const double* lbx, const double* ubx, int32_t nz, const double* lbg, const double* ubg, int32_t ng) {
  if (!h || !lbx || !ubx || !lbg || !ubg) return fail(h, MPCB_E_INVALID, "NULL argument");
  mpcb_config c = h->cfg;
  const int N = c.N, nx = h->nx;
  if (nz != h->nz) return fail(h, MPCB_E_BOUNDS, "len(lbx) = %d, the NLP has %d variables", nz, h->nz);
  // decision-vector boxes must repeat per stage (kin.py:90-105)
  for (int k = 0; k < N; ++k) for (int i = 0; i < 2; ++i)
    if (lbx[2 * k + i] != lbx[i] || ubx[2 * k + i] != ubx[i]) return fail(h, MPCB_E_BOUNDS, "control box differs at stage %d", k);
  for (int k = 0; k <= N; ++k) for (int i = 0; i < nx; ++i)
    if (lbx[2 * N + nx * k + i] != lbx[2 * N + i] || ubx[2 * N + nx * k + i] != ubx[2 * N + i]) return fail(h, MPCB_E_BOUNDS, "state box differs at node %d", k);
  for (int i = 0; i < 2; ++i) { c.u_lo[i] = lbx[i]; c.u_hi[i] = ubx[i]; }
  for (int i = 0; i < nx; ++i) { c.x_lo[i] = lbx[2 * N + i]; c.x_hi[i] = ubx[2 * N + i]; }
  // rows: nx(N+1) equalities, rate rows, obstacle rows [hmin, inf)
  const int n_obs_rows = c.n_obs * (c.obs_terminal ? N + 1 : N);
  const int n_eq = nx * (N + 1);
  const int rate_rows = ng - n_eq - n_obs_rows;
  if (rate_rows < 0 || (rate_rows != 0 && rate_rows % (N - 1 > 0 ? N - 1 : 1) != 0) || (N > 1 && rate_rows / (N - 1) > 2))
    return fail(h, MPCB_E_BOUNDS, "len(lbg) = %d does not match %d equality + k*(N-1) rate + %d obstacle rows", ng, n_eq, n_obs_rows);
  const int nr = (N > 1) ? rate_rows / (N - 1) : 0;
  std::vector<int> kind(ng, 0);   // 0 equality, 1.. rate comp+1, 3 obstacle
  int r = 0;
  if (!c.rate_interleaved) {
    for (int i = 0; i < n_eq; ++i) kind[r++] = 0;
    for (int k = 1; k < N; ++k) for (int q = 0; q < nr; ++q) kind[r++] = 1 + q;
  } else {
    for (int i = 0; i < nx; ++i) kind[r++] = 0;
    for (int k = 0; k < N; ++k) { for (int i = 0; i < nx; ++i) kind[r++] = 0; if (k > 0) for (int q = 0; q < nr; ++q) kind[r++] = 1 + q; }
  }
  for (int i = 0; i < n_obs_rows; ++i) kind[r++] = 3;
  double rl[2] = {-INF, -INF}, ru[2] = {INF, INF}; bool rset[2] = {false, false};
  double ol = 0; bool oset = false;
  for (int i = 0; i < ng; ++i) {
    if (kind[i] == 0) {
      if (lbg[i] != 0.0 || ubg[i] != 0.0) return fail(h, MPCB_E_BOUNDS, "row %d is a dynamics row of g but its bounds are [%g, %g], not [0, 0] (bounds and rows mis-aligned?)", i, lbg[i], ubg[i]);
    } else if (kind[i] == 3) {
      if (std::isfinite(ubg[i]) || !std::isfinite(lbg[i])) return fail(h, MPCB_E_BOUNDS, "row %d is an obstacle row but its bounds are [%g, %g]", i, lbg[i], ubg[i]);
      if (oset && lbg[i] != ol) return fail(h, MPCB_E_BOUNDS, "obstacle rows have different lower bounds");
      ol = lbg[i]; oset = true;
    } else {
      const int q = kind[i] - 1;
      if (rset[q] && (lbg[i] != rl[q] || ubg[i] != ru[q])) return fail(h, MPCB_E_BOUNDS, "rate rows of control %d have different bounds", q);
      if (lbg[i] == 0.0 && ubg[i] == 0.0) return fail(h, MPCB_E_BOUNDS, "row %d is a rate row of g but its bounds are [0, 0] (bounds and rows mis-aligned?)", i);
      rl[q] = lbg[i]; ru[q] = ubg[i]; rset[q] = true;
    }
  }
  // which controls carry rate rows: the configured ones, in order
  int q = 0;
  for (int i = 0; i < 2; ++i) {
    const bool had = std::isfinite(h->cfg.du_lo[i]) || std::isfinite(h->cfg.du_hi[i]);
    if (had) { if (q >= nr) return fail(h, MPCB_E_BOUNDS, "the NLP has rate rows on control %d but lbg has none", i); c.du_lo[i] = rl[q]; c.du_hi[i] = ru[q]; ++q; }
  }
  if (q != nr) return fail(h, MPCB_E_BOUNDS, "lbg carries %d rate rows per stage, the NLP has %d", nr, q);
  if (oset) c.obs_hmin = ol;
  mpcb_config saved = h->cfg;
  int rc = check_cfg(h, &c);
  if (rc != MPCB_OK) { h->cfg = saved; return rc; }
  h->cfg = c;
  for (auto* q : h->peers) q->cfg = c;          // a device group solves every shard with the leader's bounds
  return MPCB_OK;
}

int mpcb_set_time_grid(mpcb_handle* h, const double* T_i, int32_t n) {
  if (!h) return MPCB_E_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  { int rc = join_lanes(h); if (rc != MPCB_OK) return rc; }
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (!T_i || n == 0) {
    if (h->d_tgrid) { HIP_TRY(h, hipFree(h->d_tgrid)); h->d_tgrid = nullptr; }
    h->tgrid_host.clear();
  } else {
    if (n != h->cfg.N) return fail(h, MPCB_E_INVALID, "the time grid has %d entries, the NLP has N = %d stages", n, h->cfg.N);
    for (int i = 0; i < n; ++i) if (!(T_i[i] > 0) || !std::isfinite(T_i[i])) return fail(h, MPCB_E_INVALID, "T_%d = %g must be positive and finite", i, T_i[i]);
    if (!h->d_tgrid) HIP_TRY(h, hipMalloc(&h->d_tgrid, (size_t)h->cfg.N * sizeof(double)));
    HIP_TRY(h, hipMemcpy(h->d_tgrid, T_i, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    h->T0 = T_i[0];
    h->tgrid_host.assign(T_i, T_i + n);
  }
  for (auto* p : h->peers) { int rc = mpcb_set_time_grid(p, T_i, n); if (rc != MPCB_OK) return fail(h, rc, "device %d: %s", p->device, p->err.c_str()); }
  HIP_TRY(h, hipSetDevice(h->device));
  return MPCB_OK;
}

int mpcb_solve_device(mpcb_handle* h, int32_t B, const double* d_x0, const double* d_xs, const double* d_obs, int32_t obs_kind,
                      const double* d_z0, double* d_z, double* d_obj, int32_t* d_status, int32_t* d_iters, double* d_kkt,
                      double* d_lam_g, double* d_lam_x, int32_t sync) {
  if (!h) return MPCB_E_INVALID;
  int rc = solve_next_lane(h, B, d_x0, d_xs, d_obs, obs_kind, d_z0, d_z, d_obj, d_status, d_iters, d_kkt, d_lam_g, d_lam_x);
  if (rc != MPCB_OK) return rc;
  if (sync) return mpcb_sync(h);
  return MPCB_OK;
}

int mpcb_set_inflight(mpcb_handle* h, int32_t k) {
  if (!h) return MPCB_E_INVALID;
  if (k < 1 || k > MPCB_INFLIGHT_MAX) return fail(h, MPCB_E_INVALID, "inflight = %d outside 1..%d", k, MPCB_INFLIGHT_MAX);
  int rc = mpcb_sync(h);
  if (rc != MPCB_OK) return rc;
  for (auto& m : h->marks) { for (auto e : m) (void)hipEventDestroy(e); m.clear(); }
  return ensure_lanes(h, k);
}

}  // extern "C"

namespace {

// One host-pointer solve on one device, in two halves so that a device group can issue every shard before it waits for any:
// issue() uploads the inputs and launches the solve (asynchronous on the handle's stream), collect() queues the downloads.
struct HostSolve {
  mpcb_handle* h; int32_t B; int obs_kind;
  const double *x0, *xs, *obs, *z0; double *z, *obj, *kkt, *lam_g, *lam_x; int32_t *status, *iters;
  double *d_x0, *d_xs, *d_obs, *d_z0, *d_z, *d_obj, *d_kkt, *d_lg, *d_lx; int32_t *d_st, *d_it;
  int issue() {
    HIP_TRY(h, hipSetDevice(h->device));
    { int rc = join_lanes(h); if (rc != MPCB_OK) return rc; }
    const int nx = h->nx, nz = h->nz, ng = h->ng, N = h->cfg.N;
    const size_t n_obs_d = (size_t)B * h->cfg.n_obs * 6 * (obs_kind == MPCB_OBSIN_PREDICTED ? N + 1 : 1);
    auto carve = [&](Carve& cv) {
      d_x0 = cv.take<double>((size_t)B * nx);
      d_xs = cv.take<double>((size_t)B * nx);
      d_obs = n_obs_d ? cv.take<double>(n_obs_d) : nullptr;
      d_z0 = z0 ? cv.take<double>((size_t)B * nz) : nullptr;
      d_z = cv.take<double>((size_t)B * nz);
      d_obj = cv.take<double>(B);
      d_kkt = cv.take<double>((size_t)B * 4);
      d_lg = lam_g ? cv.take<double>((size_t)B * ng) : nullptr;
      d_lx = lam_x ? cv.take<double>((size_t)B * nz) : nullptr;
      d_st = cv.take<int32_t>(B);
      d_it = cv.take<int32_t>(B);
    };
    { Carve dry{nullptr}; carve(dry); int rc = ensure_scratch(h, dry.bytes()); if (rc != MPCB_OK) return rc; }
    { Carve cv{(char*)h->d_buf}; carve(cv); }
    hipStream_t s = h->stream;
    HIP_TRY(h, hipMemcpyAsync(d_x0, x0, (size_t)B * nx * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(d_xs, xs, (size_t)B * nx * 8, hipMemcpyHostToDevice, s));
    if (d_obs) HIP_TRY(h, hipMemcpyAsync(d_obs, obs, n_obs_d * 8, hipMemcpyHostToDevice, s));
    if (d_z0) HIP_TRY(h, hipMemcpyAsync(d_z0, z0, (size_t)B * nz * 8, hipMemcpyHostToDevice, s));
    return solve_on_device(h, B, d_x0, d_xs, d_obs, obs_kind, d_z0, d_z, d_obj, d_st, d_it, 1, d_kkt, d_lg, d_lx);
  }
  int collect() {
    HIP_TRY(h, hipSetDevice(h->device));
    { int rc = join_lanes(h); if (rc != MPCB_OK) return rc; }
    const int nz = h->nz, ng = h->ng;
    hipStream_t s = h->stream;
    if (z) HIP_TRY(h, hipMemcpyAsync(z, d_z, (size_t)B * nz * 8, hipMemcpyDeviceToHost, s));
    if (obj) HIP_TRY(h, hipMemcpyAsync(obj, d_obj, (size_t)B * 8, hipMemcpyDeviceToHost, s));
    if (kkt) HIP_TRY(h, hipMemcpyAsync(kkt, d_kkt, (size_t)B * 4 * 8, hipMemcpyDeviceToHost, s));
    if (status) HIP_TRY(h, hipMemcpyAsync(status, d_st, (size_t)B * 4, hipMemcpyDeviceToHost, s));
    if (iters) HIP_TRY(h, hipMemcpyAsync(iters, d_it, (size_t)B * 4, hipMemcpyDeviceToHost, s));
    if (lam_g) HIP_TRY(h, hipMemcpyAsync(lam_g, d_lg, (size_t)B * ng * 8, hipMemcpyDeviceToHost, s));
    if (lam_x) HIP_TRY(h, hipMemcpyAsync(lam_x, d_lx, (size_t)B * nz * 8, hipMemcpyDeviceToHost, s));
    return MPCB_OK;
  }
};

// mpcb_solve on a device group (mpcb_set_devices): contiguous shards, no data-path collective, one all-gather of z
int solve_group(mpcb_handle* h, int32_t B, const double* x0, const double* xs, const double* obs, int32_t obs_kind, const double* z0,
                double* z, double* obj, int32_t* status, int32_t* iters, double* kkt, double* lam_g, double* lam_x) {
  const int G = (int)h->peers.size(), nx = h->nx, nz = h->nz, ng = h->ng, N = h->cfg.N;
  const size_t obs_row = (size_t)h->cfg.n_obs * 6 * (obs_kind == MPCB_OBSIN_PREDICTED ? N + 1 : 1);
  const Rccl* R = rccl();
  std::vector<HostSolve> part(G);
  int64_t longest = 0;
  for (int g = 0; g < G; ++g) { int64_t lo, hi; mpcb_shard_bounds(B, G, g, &lo, &hi); if (hi - lo > longest) longest = hi - lo; }
  for (int g = 0; g < G; ++g) {
    int64_t lo, hi; mpcb_shard_bounds(B, G, g, &lo, &hi);
    mpcb_handle* p = h->peers[g];
    p->cfg = h->cfg;                                       // (bounds reach the peers in mpcb_set_bounds, the time grid in mpcb_set_time_grid / mpcb_set_devices; this keeps option edits of the leader in step)
    HostSolve& q = part[g];
    q = HostSolve{};
    q.h = p; q.B = (int32_t)(hi - lo); q.obs_kind = obs_kind;
    q.x0 = x0 + lo * nx; q.xs = xs + lo * nx; q.obs = obs ? obs + lo * obs_row : nullptr; q.z0 = z0 ? z0 + lo * nz : nullptr;
    q.z = nullptr;                                         // z comes back from the gathered copy
    q.obj = obj ? obj + lo : nullptr; q.kkt = kkt ? kkt + lo * 4 : nullptr; q.status = status ? status + lo : nullptr;
    q.iters = iters ? iters + lo : nullptr; q.lam_g = lam_g ? lam_g + lo * ng : nullptr; q.lam_x = lam_x ? lam_x + lo * nz : nullptr;
    // all-gather target and a padded send block (shards may differ by one row) on this device
    HIP_TRY(p, hipSetDevice(p->device));
    const size_t need = (size_t)(G + 1) * longest * nz;
    if (need > p->gather_cap) {
      if (p->d_gather) HIP_TRY(p, hipFree(p->d_gather));
      p->d_gather = nullptr; p->gather_cap = 0;
      HIP_TRY(p, hipMalloc(&p->d_gather, need * sizeof(double)));
      p->gather_cap = need;
    }
  }
  // Every shard is issued from its own host thread: the uploads come from pageable caller memory, where hipMemcpyAsync blocks
  // its calling thread until the copy is staged — issued from one thread the shards would start one after the other.
  {
    std::vector<int> rcs(G, MPCB_OK);
    auto issue_shard = [&](int g) {
      mpcb_handle* p = h->peers[g]; HostSolve& q = part[g];
      auto body = [&]() -> int {
        HIP_TRY(p, hipSetDevice(p->device));
        if (q.B > 0) { int rc = q.issue(); if (rc != MPCB_OK) return rc; }
        double* send = p->d_gather + (size_t)G * longest * nz;
        if (q.B < longest) HIP_TRY(p, hipMemsetAsync(send, 0, (size_t)longest * nz * 8, p->stream));
        if (q.B > 0) HIP_TRY(p, hipMemcpyAsync(send, q.d_z, (size_t)q.B * nz * 8, hipMemcpyDeviceToDevice, p->stream));
        return MPCB_OK;
      };
      rcs[g] = body();
    };
    std::vector<std::thread> th;
    for (int g = 1; g < G; ++g) th.emplace_back(issue_shard, g);
    issue_shard(0);
    for (auto& t : th) t.join();
    for (int g = 0; g < G; ++g) if (rcs[g] != MPCB_OK) return fail(h, rcs[g], "device %d: %s", h->peers[g]->device, h->peers[g]->err.c_str());
  }
  // one all-gather over xGMI: every device ends up with every shard's trajectories
  NCCL_TRY(h, R, R->GroupStart());
  for (int g = 0; g < G; ++g) {
    mpcb_handle* p = h->peers[g];
    ncclResult_t e = R->AllGather(p->d_gather + (size_t)G * longest * nz, p->d_gather, (size_t)longest * nz, ncclDouble, p->comm, p->stream);
    if (e != ncclSuccess) { (void)R->GroupEnd(); return fail(h, MPCB_E_DEVICE, "ncclAllGather: %s", R->GetErrorString(e)); }
  }
  NCCL_TRY(h, R, R->GroupEnd());
  for (int g = 0; g < G; ++g) {
    mpcb_handle* p = h->peers[g];
    if (part[g].B > 0) { int rc = part[g].collect(); if (rc != MPCB_OK) return fail(h, rc, "device %d: %s", p->device, p->err.c_str()); }
    p->gathered_rows = (size_t)longest; p->gathered_B = B;
  }
  // the caller's z: the gathered copy of device 0, shard blocks de-padded
  mpcb_handle* p0 = h->peers[0];
  HIP_TRY(p0, hipSetDevice(p0->device));
  for (int g = 0; g < G; ++g) {
    int64_t lo, hi; mpcb_shard_bounds(B, G, g, &lo, &hi);
    if (hi > lo) HIP_TRY(p0, hipMemcpyAsync(z + lo * nz, p0->d_gather + (size_t)g * longest * nz, (size_t)(hi - lo) * nz * 8, hipMemcpyDeviceToHost, p0->stream));
  }
  for (int g = 0; g < G; ++g) { mpcb_handle* p = h->peers[g]; HIP_TRY(p, hipSetDevice(p->device)); HIP_TRY(p, hipStreamSynchronize(p->stream)); }
  HIP_TRY(h, hipSetDevice(h->device));
  return MPCB_OK;
}

}  // namespace

extern "C" {

int mpcb_solve(mpcb_handle* h, int32_t B, const double* x0, const double* xs, const double* obs, int32_t obs_kind, const double* z0,
               double* z, double* obj, int32_t* status, int32_t* iters, double* kkt, double* lam_g, double* lam_x) {
  if (!h) return MPCB_E_INVALID;
  if (B < 0 || !x0 || !xs || !z) return fail(h, MPCB_E_INVALID, "B < 0 or a required pointer is NULL");
  if (h->cfg.n_obs > 0 && !obs) return fail(h, MPCB_E_INVALID, "n_obs = %d but obs is NULL", h->cfg.n_obs);
  if (obs_kind != MPCB_OBSIN_STATIC && obs_kind != MPCB_OBSIN_PREDICTED) return fail(h, MPCB_E_INVALID, "unknown obs_kind %d", obs_kind);
  if (B == 0) return MPCB_OK;
  if (!h->peers.empty()) return solve_group(h, B, x0, xs, obs, obs_kind, z0, z, obj, status, iters, kkt, lam_g, lam_x);
  HostSolve q{};
  q.h = h; q.B = B; q.obs_kind = obs_kind; q.x0 = x0; q.xs = xs; q.obs = obs; q.z0 = z0;
  q.z = z; q.obj = obj; q.kkt = kkt; q.lam_g = lam_g; q.lam_x = lam_x; q.status = status; q.iters = iters;
  int rc = q.issue();
  if (rc != MPCB_OK) return rc;
  rc = q.collect();
  if (rc != MPCB_OK) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MPCB_OK;
}

int mpcb_solve_trace(mpcb_handle* h, const double* x0, const double* xs, const double* obs, int32_t obs_kind, const double* z0,
                     double* z, int32_t* status, int32_t* iters, double* trace) {
  if (!h || !x0 || !xs || !z || !trace) return fail(h, MPCB_E_INVALID, "NULL argument");
  if (h->cfg.n_obs > 0 && !obs) return fail(h, MPCB_E_INVALID, "n_obs = %d but obs is NULL", h->cfg.n_obs);
  HIP_TRY(h, hipSetDevice(h->device));
  { int rc = join_lanes(h); if (rc != MPCB_OK) return rc; }
  const int nx = h->nx, nz = h->nz, N = h->cfg.N;
  const size_t n_obs_d = (size_t)h->cfg.n_obs * 6 * (obs_kind == MPCB_OBSIN_PREDICTED ? N + 1 : 1);
  const size_t n_tr = (size_t)(h->cfg.max_iter + 1) * 8;
  double *d_x0, *d_xs, *d_obs, *d_z0, *d_z, *d_tr; int32_t *d_st, *d_it;
  auto carve = [&](Carve& cv) {
    d_x0 = cv.take<double>(nx); d_xs = cv.take<double>(nx);
    d_obs = n_obs_d ? cv.take<double>(n_obs_d) : nullptr;
    d_z0 = z0 ? cv.take<double>(nz) : nullptr;
    d_z = cv.take<double>(nz); d_tr = cv.take<double>(n_tr);
    d_st = cv.take<int32_t>(1); d_it = cv.take<int32_t>(1);
  };
  { Carve dry{nullptr}; carve(dry); int rc = ensure_scratch(h, dry.bytes()); if (rc != MPCB_OK) return rc; }
  { Carve cv{(char*)h->d_buf}; carve(cv); }
  int rc = MPCB_OK;
  hipStream_t s = h->stream;
  HIP_TRY(h, hipMemcpyAsync(d_x0, x0, nx * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(h, hipMemcpyAsync(d_xs, xs, nx * 8, hipMemcpyHostToDevice, s));
  if (d_obs) HIP_TRY(h, hipMemcpyAsync(d_obs, obs, n_obs_d * 8, hipMemcpyHostToDevice, s));
  if (d_z0) HIP_TRY(h, hipMemcpyAsync(d_z0, z0, nz * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(h, hipMemsetAsync(d_tr, 0, n_tr * 8, s));
  MpcbKArgs a;
  a.cfg = h->cfg; a.B = 1; a.nz = nz; a.ng = h->ng; a.obs_kind = obs_kind; a.want_mult = 0; a.trace_instance = 0; a.trace = d_tr; a.st_stride = 1; a.tgrid = h->d_tgrid;
  a.x0 = d_x0; a.xs = d_xs; a.obs = d_obs; a.z0 = d_z0; a.z = d_z; a.obj = nullptr; a.kkt = nullptr; a.lam_g = nullptr; a.lam_x = nullptr;
  a.status = d_st; a.iters = d_it;
  rc = launch_solve(h, a);
  if (rc != MPCB_OK) return rc;
  HIP_TRY(h, hipMemcpyAsync(z, d_z, nz * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipMemcpyAsync(trace, d_tr, n_tr * 8, hipMemcpyDeviceToHost, s));
  if (status) HIP_TRY(h, hipMemcpyAsync(status, d_st, 4, hipMemcpyDeviceToHost, s));
  if (iters) HIP_TRY(h, hipMemcpyAsync(iters, d_it, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipStreamSynchronize(s));
  return MPCB_OK;
}

// the closed loop; scenes either come from the host (x0, xs, obs_state) or are drawn on the device (sample_kind != 0)
static int closed_loop_impl(mpcb_handle* h, int32_t B, int32_t steps, const double* x0, const double* xs, double* obs_state, int32_t obs_motion,
                            int32_t flags, double* x_hist, double* u_hist, int32_t* status_hist, int32_t* iters_hist,
                            int32_t sample_kind, uint64_t seed, uint64_t first_index, double* x0_out, double* obs0_out) {
  if (!h) return MPCB_E_INVALID;
  if (B < 0 || steps < 0 || (!sample_kind && (!x0 || !xs))) return fail(h, MPCB_E_INVALID, "B < 0, steps < 0 or a required pointer is NULL");
  if (!sample_kind && h->cfg.n_obs > 0 && !obs_state) return fail(h, MPCB_E_INVALID, "n_obs = %d but obs_state is NULL", h->cfg.n_obs);
  if (obs_motion < MPCB_OBSMOVE_STATIC || obs_motion > MPCB_OBSMOVE_CURRENT) return fail(h, MPCB_E_INVALID, "unknown obs_motion %d", obs_motion);
  if (flags & ~(MPCB_CL_HOLD_ON_FAILURE | MPCB_CL_ADVANCE_FIRST_ONLY)) return fail(h, MPCB_E_INVALID, "unknown flags 0x%x", flags);
  const int predict = obs_motion == MPCB_OBSMOVE_PREDICTED;
  if (B == 0 || steps == 0) return MPCB_OK;
  HIP_TRY(h, hipSetDevice(h->device));
  { int rc = join_lanes(h); if (rc != MPCB_OK) return rc; }
  const int nx = h->nx, nz = h->nz, N = h->cfg.N, no = h->cfg.n_obs;
  const size_t n_traj = predict ? (size_t)B * no * (N + 1) * 6 : 0;
  double *d_x0, *d_xs, *d_obs, *d_traj, *d_z0, *d_z, *d_xh, *d_uh; int32_t *d_st, *d_it;
  auto carve = [&](Carve& cv) {
    d_x0 = cv.take<double>((size_t)B * nx);
    d_xs = cv.take<double>((size_t)B * nx);
    d_obs = no ? cv.take<double>((size_t)B * no * 6) : nullptr;
    d_traj = n_traj ? cv.take<double>(n_traj) : nullptr;
    d_z0 = cv.take<double>((size_t)B * nz);
    d_z = cv.take<double>((size_t)B * nz);
    d_xh = cv.take<double>((size_t)B * (steps + 1) * nx);
    d_uh = cv.take<double>((size_t)B * steps * 2);
    d_st = cv.take<int32_t>((size_t)B * steps);
    d_it = cv.take<int32_t>((size_t)B * steps);
  };
  { Carve dry{nullptr}; carve(dry); int rc = ensure_scratch(h, dry.bytes()); if (rc != MPCB_OK) return rc; }
  { Carve cv{(char*)h->d_buf}; carve(cv); }
  hipStream_t s = h->stream;
  if (sample_kind) {
    int rc = mpcb_sample_scenes(h, sample_kind, B, seed, first_index, d_x0, d_xs, d_obs);
    if (rc != MPCB_OK) return rc;
    if (x0_out) HIP_TRY(h, hipMemcpyAsync(x0_out, d_x0, (size_t)B * nx * 8, hipMemcpyDeviceToHost, s));
    if (obs0_out && d_obs) HIP_TRY(h, hipMemcpyAsync(obs0_out, d_obs, (size_t)B * no * 6 * 8, hipMemcpyDeviceToHost, s));
  } else {
    HIP_TRY(h, hipMemcpyAsync(d_x0, x0, (size_t)B * nx * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(d_xs, xs, (size_t)B * nx * 8, hipMemcpyHostToDevice, s));
    if (d_obs) HIP_TRY(h, hipMemcpyAsync(d_obs, obs_state, (size_t)B * no * 6 * 8, hipMemcpyHostToDevice, s));
  }
  HIP_TRY(h, hipMemsetAsync(d_z0, 0, (size_t)B * nz * 8, s));                  // u0 = 0, next_states = 0 (main_cbf_kin_c_sim.py:47-50)
  HIP_TRY(h, hipMemcpy2DAsync(d_xh, (size_t)(steps + 1) * nx * 8, d_x0, (size_t)nx * 8, (size_t)nx * 8, B, hipMemcpyDeviceToDevice, s));
  const int move = obs_motion == MPCB_OBSMOVE_STATIC ? 0 : (flags & MPCB_CL_ADVANCE_FIRST_ONLY) ? 2 : 1;
  const int hold = (flags & MPCB_CL_HOLD_ON_FAILURE) ? 1 : 0;
  const double Tstep = h->d_tgrid ? h->T0 : h->cfg.T;
  for (int t = 0; t < steps; ++t) {
    const double* obs_in = d_obs; int kind = MPCB_OBSIN_STATIC;
    if (predict && no) {
      const int total = B * no;
      hipLaunchKernelGGL(mpcb_predict_obs, dim3((total + 255) / 256), dim3(256), 0, s, total, N, h->cfg.T, h->d_tgrid, d_obs, d_traj);
      obs_in = d_traj; kind = MPCB_OBSIN_PREDICTED;
    }
    // the solve kernel writes status / iters of step t straight into column t of the [B, steps] histories
    int rc = solve_on_device(h, B, d_x0, d_xs, obs_in, kind, d_z0, d_z, nullptr, d_st + t, d_it + t, steps, nullptr, nullptr, nullptr);
    if (rc != MPCB_OK) return rc;
    if (nx == 6)
      hipLaunchKernelGGL(mpcb_advance<6>, dim3((B + 127) / 128), dim3(128), 0, s, h->cfg, B, nz, d_z, d_x0, d_z0, d_obs, d_xh, d_uh,
                         d_st + t, steps, t, steps, move, hold, Tstep);
    else
      hipLaunchKernelGGL(mpcb_advance<4>, dim3((B + 127) / 128), dim3(128), 0, s, h->cfg, B, nz, d_z, d_x0, d_z0, d_obs, d_xh, d_uh,
                         d_st + t, steps, t, steps, move, hold, Tstep);
    HIP_TRY(h, hipGetLastError());
  }
  if (x_hist) HIP_TRY(h, hipMemcpyAsync(x_hist, d_xh, (size_t)B * (steps + 1) * nx * 8, hipMemcpyDeviceToHost, s));
  if (u_hist) HIP_TRY(h, hipMemcpyAsync(u_hist, d_uh, (size_t)B * steps * 2 * 8, hipMemcpyDeviceToHost, s));
  if (status_hist) HIP_TRY(h, hipMemcpyAsync(status_hist, d_st, (size_t)B * steps * 4, hipMemcpyDeviceToHost, s));
  if (iters_hist) HIP_TRY(h, hipMemcpyAsync(iters_hist, d_it, (size_t)B * steps * 4, hipMemcpyDeviceToHost, s));
  if (d_obs && obs_state) HIP_TRY(h, hipMemcpyAsync(obs_state, d_obs, (size_t)B * no * 6 * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipStreamSynchronize(s));
  return MPCB_OK;
}

int mpcb_closed_loop(mpcb_handle* h, int32_t B, int32_t steps, const double* x0, const double* xs, double* obs_state, int32_t obs_motion,
                     int32_t flags, double* x_hist, double* u_hist, int32_t* status_hist, int32_t* iters_hist) {
  return closed_loop_impl(h, B, steps, x0, xs, obs_state, obs_motion, flags, x_hist, u_hist, status_hist, iters_hist, 0, 0, 0, nullptr, nullptr);
}

int mpcb_closed_loop_sampled(mpcb_handle* h, int32_t kind, int32_t B, uint64_t seed, uint64_t first_index, int32_t steps, int32_t obs_motion,
                             int32_t flags, double* x0_out, double* obs0_out, double* x_hist, double* u_hist, int32_t* status_hist, int32_t* iters_hist) {
  if (!h) return MPCB_E_INVALID;
  if (kind != MPCB_SCENES_C2 && kind != MPCB_SCENES_C3 && kind != MPCB_SCENES_C4) return fail(h, MPCB_E_INVALID, "unknown scene kind %d", kind);
  return closed_loop_impl(h, B, steps, nullptr, nullptr, nullptr, obs_motion, flags, x_hist, u_hist, status_hist, iters_hist, kind, seed, first_index, x0_out, obs0_out);
}

int mpcb_sample_scenes(mpcb_handle* h, int32_t kind, int32_t B, uint64_t seed, uint64_t first_index, double* d_x0, double* d_xs, double* d_obs) {
  if (!h) return MPCB_E_INVALID;
  if (B < 0 || !d_x0 || !d_xs || (h->cfg.n_obs > 0 && !d_obs)) return fail(h, MPCB_E_INVALID, "B < 0 or a required pointer is NULL");
  if (kind != MPCB_SCENES_C2 && kind != MPCB_SCENES_C3 && kind != MPCB_SCENES_C4) return fail(h, MPCB_E_INVALID, "unknown scene kind %d", kind);
  if ((kind == MPCB_SCENES_C4) != (h->cfg.model == MPCB_MODEL_DYN)) return fail(h, MPCB_E_INVALID, "scene kind %d does not fit the handle's model", kind);
  if (kind == MPCB_SCENES_C2 && h->cfg.n_obs > 1) return fail(h, MPCB_E_INVALID, "MPCB_SCENES_C2 has one obstacle, the handle has n_obs = %d", h->cfg.n_obs);
  if (B == 0) return MPCB_OK;
  HIP_TRY(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(mpcb_sample_kernel, dim3((B + 127) / 128), dim3(128), 0, h->stream, h->cfg, kind, B, seed, first_index, d_x0, d_xs, d_obs);
  HIP_TRY(h, hipGetLastError());
  return MPCB_OK;
}

int mpcb_predict_obstacles(mpcb_handle* h, int32_t n, int32_t N, double dt, const double* obs, double* traj) {
  if (!h || n < 0 || N < 1 || !(dt > 0) || !obs || !traj) return fail(h, MPCB_E_INVALID, "bad argument");
  if (n == 0) return MPCB_OK;
  HIP_TRY(h, hipSetDevice(h->device));
  { int rc = join_lanes(h); if (rc != MPCB_OK) return rc; }
  double *d_o, *d_t;
  auto carve = [&](Carve& cv) { d_o = cv.take<double>((size_t)n * 6); d_t = cv.take<double>((size_t)n * (N + 1) * 6); };
  { Carve dry{nullptr}; carve(dry); int rc = ensure_scratch(h, dry.bytes()); if (rc != MPCB_OK) return rc; }
  { Carve cv{(char*)h->d_buf}; carve(cv); }
  HIP_TRY(h, hipMemcpyAsync(d_o, obs, (size_t)n * 6 * 8, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(mpcb_predict_obs, dim3((n + 255) / 256), dim3(256), 0, h->stream, n, N, dt, (const double*)nullptr, d_o, d_t);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(traj, d_t, (size_t)n * (N + 1) * 6 * 8, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MPCB_OK;
}

int mpcb_ref_path_window(mpcb_handle* h, int32_t B, double x_start, const double* x0, const double* xs, double T_horizon, double dt,
                         int32_t* last_idx, double* window) {
  if (!h || B < 0 || !x0 || !xs || !last_idx || !window || !(dt > 0) || !(T_horizon > 0)) return fail(h, MPCB_E_INVALID, "bad argument");
  if (B == 0) return MPCB_OK;
  const int N_p = (int)(T_horizon / dt);                                              // RefPathGenerator.py:32
  HIP_TRY(h, hipSetDevice(h->device));
  double *d_x0, *d_xs, *d_w; int32_t* d_li;
  auto carve = [&](Carve& cv) { d_x0 = cv.take<double>((size_t)B * 4); d_xs = cv.take<double>((size_t)B * 4); d_w = cv.take<double>((size_t)B * (N_p + 1) * 4); d_li = cv.take<int32_t>(B); };
  { Carve dry{nullptr}; carve(dry); int rc = ensure_scratch(h, dry.bytes()); if (rc != MPCB_OK) return rc; }
  { Carve cv{(char*)h->d_buf}; carve(cv); }
  hipStream_t s = h->stream;
  HIP_TRY(h, hipMemcpyAsync(d_x0, x0, (size_t)B * 4 * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(h, hipMemcpyAsync(d_xs, xs, (size_t)B * 4 * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(h, hipMemcpyAsync(d_li, last_idx, (size_t)B * 4, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(mpcb_ref_window_kernel, dim3((B + 127) / 128), dim3(128), 0, s, B, x_start, d_x0, d_xs, T_horizon, dt, N_p, d_li, d_w);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(window, d_w, (size_t)B * (N_p + 1) * 4 * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipMemcpyAsync(last_idx, d_li, (size_t)B * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipStreamSynchronize(s));
  return MPCB_OK;
}

// ---- multi-GPU ----------------------------------------------------------------------------------------------------------
int mpcb_shard_bounds(int64_t B, int32_t world, int32_t rank, int64_t* lo, int64_t* hi) {
  if (B < 0 || world < 1 || rank < 0 || rank >= world || !lo || !hi) return MPCB_E_INVALID;
  const int64_t q = B / world, r = B % world;
  *lo = rank * q + (rank < r ? rank : r);
  *hi = *lo + q + (rank < r ? 1 : 0);
  return MPCB_OK;
}

int mpcb_comm_unique_id(void* id128) {
  if (!id128) return MPCB_E_INVALID;
  const Rccl* R = rccl();
  if (!R) return fail(nullptr, MPCB_E_DEVICE, "librccl.so could not be loaded: %s", g_rccl_error.c_str());
  static_assert(sizeof(ncclUniqueId) == MPCB_UNIQUE_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  ncclResult_t e = R->GetUniqueId(&id);
  if (e != ncclSuccess) return fail(nullptr, MPCB_E_DEVICE, "ncclGetUniqueId: %s", R->GetErrorString(e));
  std::memcpy(id128, &id, sizeof id);
  return MPCB_OK;
}

int mpcb_comm_init_rank(mpcb_handle* h, const void* id128, int32_t rank, int32_t world) {
  if (!h || !id128 || world < 1 || rank < 0 || rank >= world) return fail(h, MPCB_E_INVALID, "bad rank / world");
  if (h->comm || !h->peers.empty()) return fail(h, MPCB_E_INVALID, "the handle already belongs to a group");
  const Rccl* R = rccl();
  if (!R) return fail(h, MPCB_E_DEVICE, "librccl.so could not be loaded: %s", g_rccl_error.c_str());
  HIP_TRY(h, hipSetDevice(h->device));
  ncclUniqueId id; std::memcpy(&id, id128, sizeof id);
  NCCL_TRY(h, R, R->CommInitRank(&h->comm, world, id, rank));
  h->world = world; h->rank = rank;
  HIP_TRY(h, hipMalloc(&h->d_red, 64 * sizeof(double)));
  return MPCB_OK;
}

int mpcb_set_devices(mpcb_handle* h, const int32_t* ids, int32_t n) {
  if (!h || !ids || n < 1) return fail(h, MPCB_E_INVALID, "bad device list");
  if (h->comm || !h->peers.empty()) return fail(h, MPCB_E_INVALID, "the handle already belongs to a group");
  int ndev = 0;
  HIP_TRY(h, hipGetDeviceCount(&ndev));
  for (int i = 0; i < n; ++i) {
    if (ids[i] < 0 || ids[i] >= ndev) return fail(h, MPCB_E_INVALID, "device %d outside 0..%d", ids[i], ndev - 1);
    for (int j = 0; j < i; ++j) if (ids[j] == ids[i]) return fail(h, MPCB_E_INVALID, "device %d listed twice", ids[i]);
  }
  const Rccl* R = rccl();
  if (!R) return fail(h, MPCB_E_DEVICE, "librccl.so could not be loaded: %s", g_rccl_error.c_str());
  std::vector<mpcb_handle*> peers(n, nullptr);
  for (int i = 0; i < n; ++i) {
    int rc = mpcb_create(&h->cfg, ids[i], &peers[i]);
    if (rc != MPCB_OK) { for (auto* p : peers) mpcb_destroy(p); return fail(h, rc, "device %d: %s", ids[i], g_create_error.c_str()); }
  }
  std::vector<ncclComm_t> comms(n);
  std::vector<int> devs(ids, ids + n);
  ncclResult_t e = R->CommInitAll(comms.data(), n, devs.data());
  if (e != ncclSuccess) { for (auto* p : peers) mpcb_destroy(p); return fail(h, MPCB_E_DEVICE, "ncclCommInitAll: %s", R->GetErrorString(e)); }
  for (int i = 0; i < n; ++i) { peers[i]->comm = comms[i]; peers[i]->world = n; peers[i]->rank = i; }
  h->peers = peers; h->world = n; h->rank = 0;
  // a time grid set before the group was formed goes to every new peer (bounds travel inside cfg, which mpcb_create copied)
  if (!h->tgrid_host.empty()) {
    for (auto* q : peers) { int rc = mpcb_set_time_grid(q, h->tgrid_host.data(), (int32_t)h->tgrid_host.size()); if (rc != MPCB_OK) return fail(h, rc, "device %d: %s", q->device, q->err.c_str()); }
    HIP_TRY(h, hipSetDevice(h->device));
  }
  return MPCB_OK;
}

int mpcb_comm_info(const mpcb_handle* h, int32_t* world, int32_t* rank) {
  if (!h) return MPCB_E_INVALID;
  if (world) *world = h->world;
  if (rank) *rank = h->rank;
  return MPCB_OK;
}

int mpcb_allgather(mpcb_handle* h, const double* d_send, double* d_recv, uint64_t count) {
  if (!h || !d_send || !d_recv) return fail(h, MPCB_E_INVALID, "NULL argument");
  HIP_TRY(h, hipSetDevice(h->device));
  { int rc = join_lanes(h); if (rc != MPCB_OK) return rc; }
  if (h->world == 1 && !h->comm) {               // no group: the gather of one block is a copy
    if (d_recv != d_send) HIP_TRY(h, hipMemcpyAsync(d_recv, d_send, count * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    return MPCB_OK;
  }
  if (!h->comm) return fail(h, MPCB_E_INVALID, "mpcb_allgather on a group leader: use mpcb_solve, which gathers");
  const Rccl* R = rccl();
  NCCL_TRY(h, R, R->AllGather(d_send, d_recv, count, ncclDouble, h->comm, h->stream));
  return MPCB_OK;
}

int mpcb_allreduce(mpcb_handle* h, double* values, int32_t n, int32_t op) {
  if (!h || !values || n < 1 || n > 64 || (op != 0 && op != 1)) return fail(h, MPCB_E_INVALID, "bad argument (1 <= n <= 64, op 0 = sum, 1 = max)");
  HIP_TRY(h, hipSetDevice(h->device));
  { int rc = join_lanes(h); if (rc != MPCB_OK) return rc; }
  if (!h->comm) { HIP_TRY(h, hipStreamSynchronize(h->stream)); return MPCB_OK; }
  const Rccl* R = rccl();
  HIP_TRY(h, hipMemcpyAsync(h->d_red, values, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  NCCL_TRY(h, R, R->AllReduce(h->d_red, h->d_red, n, ncclDouble, op == 0 ? ncclSum : ncclMax, h->comm, h->stream));
  HIP_TRY(h, hipMemcpyAsync(values, h->d_red, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MPCB_OK;
}

int mpcb_gathered_z(mpcb_handle* h, int32_t index, const double** d_z) {
  if (!h || !d_z || index < 0 || index >= (int)h->peers.size()) return fail(h, MPCB_E_INVALID, "not a group leader or bad index");
  *d_z = h->peers[index]->d_gather;
  return MPCB_OK;
}

int mpcb_dev_alloc(mpcb_handle* h, uint64_t bytes, void** dptr) {
  if (!h || !dptr) return MPCB_E_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipMalloc(dptr, bytes ? bytes : 8));
  return MPCB_OK;
}
int mpcb_dev_free(mpcb_handle* h, void* dptr) {
  if (!h) return MPCB_E_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  { int rc = mpcb_sync(h); if (rc != MPCB_OK) return rc; }       // nothing in flight may still use the buffer
  HIP_TRY(h, hipFree(dptr));
  return MPCB_OK;
}
int mpcb_dev_upload(mpcb_handle* h, void* dptr, const void* src, uint64_t bytes) {
  if (!h || !dptr || !src) return MPCB_E_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  { int rc = join_lanes(h); if (rc != MPCB_OK) return rc; }
  HIP_TRY(h, hipMemcpyAsync(dptr, src, bytes, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MPCB_OK;
}
int mpcb_dev_download(mpcb_handle* h, void* dst, const void* dptr, uint64_t bytes) {
  if (!h || !dptr || !dst) return MPCB_E_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  { int rc = join_lanes(h); if (rc != MPCB_OK) return rc; }
  HIP_TRY(h, hipMemcpyAsync(dst, dptr, bytes, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MPCB_OK;
}
int mpcb_sync(mpcb_handle* h) {
  if (!h) return MPCB_E_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  { int rc = join_lanes(h); if (rc != MPCB_OK) return rc; }
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MPCB_OK;
}

int mpcb_stream_wait(mpcb_handle* h, mpcb_handle* other) {
  if (!h || !other) return MPCB_E_INVALID;
  if (h->device != other->device) return fail(h, MPCB_E_INVALID, "mpcb_stream_wait: the handles live on different devices");
  HIP_TRY(h, hipSetDevice(h->device));
  { int rc = join_lanes(other); if (rc != MPCB_OK) return fail(h, rc, "%s", other->err.c_str()); }   // "everything queued on other" includes its lanes
  if (!other->ev_sync) HIP_TRY(h, hipEventCreateWithFlags(&other->ev_sync, hipEventDisableTiming));
  HIP_TRY(h, hipEventRecord(other->ev_sync, other->stream));
  HIP_TRY(h, hipStreamWaitEvent(h->stream, other->ev_sync, 0));
  return MPCB_OK;
}

int mpcb_event_record(mpcb_handle* h, int32_t slot) {
  if (!h || slot < 0 || slot >= MPCB_EVENT_SLOTS) return fail(h, MPCB_E_INVALID, "slot outside 0..%d", MPCB_EVENT_SLOTS - 1);
  HIP_TRY(h, hipSetDevice(h->device));
  if (h->lanes.empty()) { int rc = ensure_lanes(h, 1); if (rc != MPCB_OK) return rc; }
  if (h->marks.empty()) h->marks.resize(MPCB_EVENT_SLOTS);
  auto& m = h->marks[slot];
  while (m.size() < h->lanes.size()) { hipEvent_t e; HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming)); m.push_back(e); }
  for (size_t i = 0; i < h->lanes.size(); ++i) HIP_TRY(h, hipEventRecord(m[i], h->lanes[i].stream));   // the lanes are NOT joined: later launches stay free to run ahead
  return MPCB_OK;
}

int mpcb_event_wait(mpcb_handle* h, const mpcb_handle* other, int32_t slot) {
  if (!h || !other || slot < 0 || slot >= MPCB_EVENT_SLOTS) return fail(h, MPCB_E_INVALID, "NULL handle or slot outside 0..%d", MPCB_EVENT_SLOTS - 1);
  if (h->device != other->device) return fail(h, MPCB_E_INVALID, "mpcb_event_wait: the handles live on different devices");
  if (other->marks.empty() || other->marks[slot].empty()) return MPCB_OK;       // never recorded: nothing to wait for
  HIP_TRY(h, hipSetDevice(h->device));
  for (auto e : other->marks[slot]) HIP_TRY(h, hipStreamWaitEvent(h->stream, e, 0));
  return MPCB_OK;
}

int mpcb_timing(mpcb_handle* h, int32_t reset, int32_t* launches, double* total_ms, double* last_ms) {
  if (!h) return MPCB_E_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  int rc = collect_timing(h);
  if (rc != MPCB_OK) return rc;
  if (launches) *launches = h->launches;
  if (total_ms) *total_ms = h->total_ms;
  if (last_ms) *last_ms = h->last_ms;
  if (reset) { h->launches = 0; h->total_ms = 0; h->last_ms = 0; }
  return MPCB_OK;
}

int mpcb_model_rhs(const mpcb_config* c, const double* x, const double* u, double* xdot) {
  if (!c || !x || !u || !xdot) return MPCB_E_INVALID;
  if (c->model != MPCB_MODEL_KIN && c->model != MPCB_MODEL_DYN) return MPCB_E_INVALID;
  model_rhs(*c, x, u, xdot);
  return MPCB_OK;
}

}  // extern "C"
