// Microbenchmark for VERDICT r1 item 4: v_mfma_f64_4x4x4_4b_f64 as the engine of the Riccati stage products.
//   (1) operand / result lane maps, found with exact integer data (the guides give none for this shape);
//   (2) issue cost: a dependent chain (D -> C), independent instructions, and MFMA interleaved with v_fma_f64 (do the two pipes overlap?);
//   (3) the chain the stage would run: D of one product re-laid as the B operand of the next (cross-block move) -> MFMA, versus the
//       present form of the same two products (LDS exchange + 4-term FMA chains, mpcb_kernel.h `stage`).
//   make -C tools/ubench && ./mfma_f64_4x4x4      (on a GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

__global__ __launch_bounds__(64) void k_map(double* out) {
  const int lane = threadIdx.x;
  // for every source lane s: B = indicator(lane == s), A = 1 + lane  ->  D tells which A lanes meet which B lane, and where the result lands
  for (int s = 0; s < 64; ++s) {
    const double a = 1.0 + lane, b = (lane == s) ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[s * 64 + lane] = d;
  }
}

template <int MODE>
__global__ __launch_bounds__(64) void k_time(double* out, long long* cyc, int iters) {
  const int lane = threadIdx.x;
  double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3, c0 = 0, c1 = 0, c2 = 0, c3 = 0, f0 = a, f1 = b, f2 = a + b, f3 = a - b;
  const double m = 1.0000001, k = 1e-9;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MODE == 0) {            // dependent chain: D feeds C
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      } else if (MODE == 1) {     // four independent accumulators
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
      } else if (MODE == 2) {     // dependent chain where D feeds the B operand (what W -> M needs), no lane movement
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c0 * 1e-3 + b, 0.0, 0, 0, 0);
      } else if (MODE == 3) {     // one MFMA + four independent v_fma_f64: do they overlap?
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        f0 = __builtin_fma(f0, m, k); f1 = __builtin_fma(f1, m, k); f2 = __builtin_fma(f2, m, k); f3 = __builtin_fma(f3, m, k);
      } else if (MODE == 4) {     // the four v_fma_f64 alone
        f0 = __builtin_fma(f0, m, k); f1 = __builtin_fma(f1, m, k); f2 = __builtin_fma(f2, m, k); f3 = __builtin_fma(f3, m, k);
      } else if (MODE == 5) {     // stage-like: MFMA, result moved across the 16-lane blocks with ds_bpermute (both halves), MFMA on it
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
        const double w = __shfl(c0, (lane + 16) & 63, 64);
        c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, w, c1, 0, 0, 0);
        b = c1 * 1e-9 + b;
      } else if (MODE == 6) {     // the present form of one product pair: value -> LDS -> 4 reads -> 4-term FMA chain, twice
        __shared__ double sm[128];
        sm[lane] = c0; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int r = (lane & 7) * 6;
        const double w = __builtin_fma(sm[r + 3], f3, __builtin_fma(sm[r + 2], f2, __builtin_fma(sm[r + 1], f1, __builtin_fma(sm[r], f0, a))));
        sm[64 + lane] = w; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int q = 64 + (lane >> 3) * 6;
        c0 = __builtin_fma(sm[q + 3], f0, __builtin_fma(sm[q + 2], f1, __builtin_fma(sm[q + 1], f2, __builtin_fma(sm[q], f3, b)))) * 1e-3;
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + lane] = c0 + c1 + c2 + c3 + f0 + f1 + f2 + f3 + b;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char* what, double per) {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 1024 * 64 * 8); (void)hipMalloc(&cyc, 1024 * 8);
  const int iters = 500;
  for (int rep = 0; rep < 2; ++rep) k_time<MODE><<<1024, 64>>>(out, cyc, iters);
  (void)hipDeviceSynchronize();
  long long h[1024]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < 1024; ++i) s += h[i];
  printf("%-92s %8.2f ticks per %s\n", what, s / 1024 / (iters * 16.0) / per, per == 1 ? "unit" : "instruction");
  (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
  double* out; (void)hipMalloc(&out, 64 * 64 * 8);
  k_map<<<1, 64>>>(out); (void)hipDeviceSynchronize();
  static double h[64 * 64]; (void)hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
  printf("lane maps (B = indicator of lane s, A = 1 + lane): for s: list of (result lane <- A lane)\n");
  for (int s = 0; s < 64; s += 1) {
    if (!(s < 6 || s == 16 || s == 17 || s == 20 || s == 32 || s == 48 || s == 63)) continue;
    printf("  s=%2d:", s);
    for (int l = 0; l < 64; ++l) if (h[s * 64 + l] != 0.0) printf(" %d<-%d", l, (int)h[s * 64 + l] - 1);
    printf("\n");
  }
  run<0>("v_mfma_f64_4x4x4 dependent chain (D -> C)", 1);
  run<1>("v_mfma_f64_4x4x4, four independent accumulators", 4);
  run<2>("v_mfma_f64_4x4x4 chain through the B operand (v_fma between)", 1);
  run<3>("1 MFMA + 4 independent v_fma_f64 (unit = the group of 5)", 1);
  run<4>("4 independent v_fma_f64 alone (unit = the group of 4)", 1);
  run<5>("stage-like: MFMA -> cross-block move (ds_bpermute x2) -> MFMA -> fma (unit = one pair of products)", 1);
  run<6>("present form: LDS store, 4 reads, 4-FMA chain, twice (unit = one pair of products)", 1);
  return 0;
}
