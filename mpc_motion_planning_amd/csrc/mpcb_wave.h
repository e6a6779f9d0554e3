// mpcb_wave.h — the cross-lane primitives the solver kernel is written against.
//
// Device build (hipcc, gfx950): thin wrappers over wave64 intrinsics; one 64-thread workgroup = one
// wavefront = one problem instance, so "sync" is a single-wave barrier that only orders LDS traffic.
//
// MPCB_WAVE_EMU build (g++; used ONLY by tests/emu to step the kernel source on a CPU, never shipped in
// libmpcbatch.so): every lane is a host thread, cross-lane traffic goes through a shared exchange array
// guarded by a barrier.  Reductions use the same xor-butterfly association order as the device code so
// that wave-uniform decisions are identical.
#pragma once

#ifndef MPCB_WAVE_EMU
#include <hip/hip_runtime.h>
#define MPCB_DEV __device__ __forceinline__
// the solve functions are ALWAYS inlined into their kernels: out of line (hipcc 7.2 does that to the five largest instantiations on its
// own) the LDS pointer becomes a generic pointer, every LDS access a FLAT instruction, and the restoration pass of kin<8, GEN> then
// computed wrong costates on the GPU (tools/probe_fuzzcase.py 11 4) while the same source stepped on the CPU was right
#ifdef MPCB_NOINLINE_SOLVE
#define MPCB_DEVFN __device__
#else
#define MPCB_DEVFN __device__ __forceinline__
#endif
#define MPCB_HD __host__ __device__ inline

namespace wv {
MPCB_DEV int lane() { return (int)threadIdx.x; }
// One wavefront per workgroup: the LDS unit executes the DS instructions of a wave in issue order, so a store by one
// lane followed by a load by another lane needs NO s_waitcnt and no s_barrier in between — only the compiler has to keep
// the program order.  Wavefront-scope fences + wave_barrier do exactly that and emit no instruction; the loads that
// follow are then waited for individually (counted lgkmcnt), so independent DS traffic stays in flight across a "sync".
#if defined(MPCB_SYNC_BLOCK)
MPCB_DEV void sync() { __syncthreads(); }
#elif defined(MPCB_SYNC_WAITCNT)
// diagnostic build (tools/sync_ab.sh): every sync drains the wave's outstanding memory operations
MPCB_DEV void sync() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
#else
MPCB_DEV void sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#endif
MPCB_DEV double shfl(double v, int src) { return __shfl(v, src, 64); }
MPCB_DEV int shfl(int v, int src) { return __shfl(v, src, 64); }
// value of lane `src` (wave-uniform index) in every lane
MPCB_DEV double bcast(double v, int src) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
MPCB_DEV double sum(double v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}
MPCB_DEV double max(double v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v = fmax(v, __shfl_xor(v, m, 64));
  return v;
}
MPCB_DEV double min(double v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v = fmin(v, __shfl_xor(v, m, 64));
  return v;
}
MPCB_DEV bool any(bool p) { return __any(p) != 0; }
MPCB_DEV bool all(bool p) { return __all(p) != 0; }

// ---- fused multi-value reductions: NS sums and NM maxima in one interleaved butterfly (independent chains
// give the scheduler ILP).  Steps 1,2,4,8 are DPP moves inside a 16-lane row, 16 and 32 go through ds_bpermute.
// Association order = xor butterfly (quad-mirror steps pair the same partial sums), every lane ends bit-identical.
template <int CTRL> MPCB_DEV double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
template <int NS, int NM> MPCB_DEV void reduce(double* s, double* m) {
#define MPCB_STEP(EXPR)                                                        \
  {                                                                            \
    double ts[NS > 0 ? NS : 1], tm[NM > 0 ? NM : 1];                           \
    _Pragma("unroll") for (int i = 0; i < NS; ++i) { double v = s[i]; ts[i] = EXPR; } \
    _Pragma("unroll") for (int i = 0; i < NM; ++i) { double v = m[i]; tm[i] = EXPR; } \
    _Pragma("unroll") for (int i = 0; i < NS; ++i) s[i] += ts[i];              \
    _Pragma("unroll") for (int i = 0; i < NM; ++i) m[i] = fmax(m[i], tm[i]);   \
  }
  MPCB_STEP(dpp_mov<0xB1>(v))      // quad_perm [1,0,3,2]
  MPCB_STEP(dpp_mov<0x4E>(v))      // quad_perm [2,3,0,1]
  MPCB_STEP(dpp_mov<0x141>(v))     // row_half_mirror
  MPCB_STEP(dpp_mov<0x140>(v))     // row_mirror
  MPCB_STEP(__shfl_xor(v, 16, 64))
  MPCB_STEP(__shfl_xor(v, 32, 64))
#undef MPCB_STEP
}
// the same value, but opaque to the optimiser (breaks common-subexpression reuse across the kernel)
MPCB_DEV int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
// The kernel's own argument block, re-addressed behind an optimisation barrier: fields read through this pointer are loaded (s_load)
// where they are used, each time, instead of at kernel entry — for values needed once per iteration (the termination tolerances)
// that keeps their SGPRs out of the register pressure of the whole solve.  Every solve kernel takes ONE argument, a MpcbKArgs by value.
template <class T> MPCB_DEV const T* late_args(const T&) {
  const unsigned long long p0 = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
  // (readfirstlane: the pointer is uniform, but behind loops whose exits the compiler cannot prove uniform it would otherwise be
  // carried in a VGPR, which the scalar constraint below rejects)
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p0), hi = __builtin_amdgcn_readfirstlane((unsigned)(p0 >> 32));
  asm volatile("" : "+s"(lo), "+s"(hi));
  const unsigned long long p = ((unsigned long long)hi << 32) | lo;
  return (const T*)(const __attribute__((address_space(4))) T*)p;      // constant address space: scalar loads
}
// a wave-uniform double moved to scalar registers
MPCB_DEV double uni(double v) {
  int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
// reciprocal: hardware estimate + two Newton steps (no denormal / overflow rescaling: arguments are slacks and duals)
MPCB_DEV double rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
}  // namespace wv

#else  // ------------------------------------------------------------------ host emulation (tests only)
#include <barrier>
#include <cmath>
#define MPCB_DEV inline
#define MPCB_DEVFN
#define MPCB_HD inline

namespace wv {
struct Emu {
  std::barrier<>* bar;
  double xd[64];
  int xi[64];
};
extern thread_local int t_lane;
extern thread_local Emu* t_emu;
inline int lane() { return t_lane; }
inline void sync() { t_emu->bar->arrive_and_wait(); }
inline double shfl(double v, int src) {
  t_emu->xd[t_lane] = v; sync();
  double r = t_emu->xd[src & 63]; sync();
  return r;
}
inline int shfl(int v, int src) {
  t_emu->xi[t_lane] = v; sync();
  int r = t_emu->xi[src & 63]; sync();
  return r;
}
inline double bcast(double v, int src) { return shfl(v, src); }
template <class Op> inline double reduce(double v, Op op) {
  t_emu->xd[t_lane] = v; sync();
  double t[64];
  for (int i = 0; i < 64; ++i) t[i] = t_emu->xd[i];
  for (int n = 64; n > 1; n >>= 1) for (int i = 0; i < n / 2; ++i) t[i] = op(t[2 * i], t[2 * i + 1]);
  sync();
  return t[0];
}
inline double sum(double v) { return reduce(v, [](double a, double b) { return a + b; }); }
inline double max(double v) { return reduce(v, [](double a, double b) { return std::fmax(a, b); }); }
inline double min(double v) { return reduce(v, [](double a, double b) { return std::fmin(a, b); }); }
inline bool any(bool p) { return sum(p ? 1.0 : 0.0) > 0.0; }
inline bool all(bool p) { return sum(p ? 0.0 : 1.0) == 0.0; }

template <int NS, int NM> inline void reduce(double* s, double* m) {
  for (int i = 0; i < NS; ++i) s[i] = sum(s[i]);
  for (int i = 0; i < NM; ++i) m[i] = max(m[i]);
}
inline int opaque(int v) { return v; }
template <class T> inline const T* late_args(const T& a) { return &a; }
inline double uni(double v) { return v; }
inline double rcp(double x) { return 1.0 / x; }
}  // namespace wv
#endif
