// Microbenchmark: issue cost of v_fma_f64 for 1 / 2 / 4 / 8 independent chains (one wave per SIMD) — the dependent-op latency
// that the serial Riccati / roll-out recursions see.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CH>
__global__ __launch_bounds__(64) void k(double* out, long long* cyc, int iters) {
  const int lane = threadIdx.x;
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = lane * 1e-3 + 1.0 + i;
  const double m = 1.0000001, c = 1e-9;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 64 / CH; ++u) {
#pragma unroll
      for (int j = 0; j < CH; ++j) a[j] = __builtin_fma(a[j], m, c);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * 64 + lane] = s;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CH>
void run() {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 1024 * 64 * 8); (void)hipMalloc(&cyc, 1024 * 8);
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) k<CH><<<1024, 64>>>(out, cyc, iters);
  (void)hipDeviceSynchronize();
  long long h[1024]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < 1024; ++i) s += h[i];
  printf("%d independent chains: %.2f ticks per v_fma_f64\n", CH, s / 1024 / (iters * 64.0));
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() { run<1>(); run<2>(); run<4>(); run<8>(); return 0; }
