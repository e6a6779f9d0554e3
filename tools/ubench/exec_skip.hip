// Microbenchmark: does a wave64 FP64 VALU instruction get cheaper when half (or three quarters) of EXEC is off?
//   hipcc -O3 --offload-arch=gfx950 -o exec_skip exec_skip.hip && ./exec_skip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int ACTIVE>
__global__ __launch_bounds__(64) void k(double* out, long long* cyc, int iters) {
  const int lane = threadIdx.x;
  double a0 = lane * 1e-3 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double m = 1.0000001, c = 1e-9;
  long long t0 = 0, t1 = 0;
  if (lane < ACTIVE) {
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
        a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
      }
    }
    t1 = __builtin_amdgcn_s_memtime();
  }
  out[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int ACTIVE>
void run(const char* name) {
  double* out; long long* cyc;
  hipMalloc(&out, 1024 * 64 * 8); hipMalloc(&cyc, 1024 * 8);
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) k<ACTIVE><<<1024, 64>>>(out, cyc, iters);
  hipDeviceSynchronize();
  long long h[1024]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < 1024; ++i) s += h[i];
  printf("%s: %.2f ticks per v_fma_f64 (64 independent-of-8 chains)\n", name, s / 1024 / (iters * 64.0));
  hipFree(out); hipFree(cyc);
}

int main() {
  run<64>("exec = 64 lanes");
  run<32>("exec = 32 lanes");
  run<16>("exec = 16 lanes");
  run<1>("exec =  1 lane ");
  return 0;
}
