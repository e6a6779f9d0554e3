// Microbenchmark: where a stage of the forward roll-out spends its time at one wave per SIMD.  The stage of the kernel
// (mpcb_kernel.h, "forward roll-out") is: 9 per-lane LDS reads for the NEXT stage, a chain of 6 dependent v_fma_f64 whose
// multiplicands are scalar registers, 2 x v_readlane of the result, 2 more FMAs, a predicated LDS store, 4 x v_readlane.
// MODE 0 = all of it; 1 = without the LDS reads; 2 = without the readlanes (operands stay vector registers);
// 3 = only the eight FMAs; 4 = reads + FMAs, no readlanes, no store
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ inline double rdlane(double v, int l) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)b, l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}

template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, long long* cyc, int stages) {
  __shared__ double fw[64 * 31];
  __shared__ double hist[8 * 65];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 31; i += 64) fw[i] = 1e-3 * ((i * 7) % 13) - 5e-3;
  __syncthreads();
  const int li = lane < 6 ? lane : 0;
  int fo[9];
  for (int r = 0; r < 9; ++r) fo[r] = (li * 5 + r * 3) % 31;
  double c[9], n9[9];
  for (int r = 0; r < 9; ++r) n9[r] = 0;
  for (int r = 0; r < 9; ++r) c[r] = fw[fo[r]];
  double v0 = 0.1, v1 = 0.2, v2 = 0.3, v3 = 0.4, v4 = 0.5, v5 = 0.6;
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma clang loop unroll(disable)
  for (int s = 0; s < stages; ++s) {
    if (MODE == 0 || MODE == 4) {
      const double* q = fw + ((s + 1) & 63) * 31;
#pragma unroll
      for (int r = 0; r < 9; ++r) n9[r] = q[fo[r]];
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 5 || MODE == 6 || MODE == 7) {
      const double* q = fw + ((s + 1) & 63) * 31;
      if (lane < (MODE == 5 ? 6 : MODE == 6 ? 16 : 32)) {
#pragma unroll
        for (int r = 0; r < 9; ++r) n9[r] = q[fo[r]];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    double t = __builtin_fma(c[5], v5, __builtin_fma(c[4], v4, __builtin_fma(c[3], v3, __builtin_fma(c[2], v2, __builtin_fma(c[1], v1, __builtin_fma(c[0], v0, c[6]))))));
    double du0, du1;
    if (MODE == 0 || MODE == 1 || MODE >= 5) { du0 = rdlane(t, 4); du1 = rdlane(t, 5); } else { du0 = t; du1 = t * 0.5; }
    const double n = __builtin_fma(c[8], du1, __builtin_fma(c[7], du0, t));
    if (MODE == 0 || MODE == 1 || MODE >= 5) {
      if (lane < 6) hist[li * 65 + (s & 63)] = n;
      v0 = rdlane(n, 0); v1 = rdlane(n, 1); v2 = rdlane(n, 2); v3 = rdlane(n, 3);
    } else { v0 = n; v1 = n; v2 = n; v3 = n; }
    v4 = du0; v5 = du1;
    if (MODE == 0 || MODE >= 4) {
#pragma unroll
      for (int r = 0; r < 9; ++r) c[r] = n9[r];
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + lane] = v0 + v1 + v2 + v3 + v4 + v5 + hist[lane];
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* what, int grid = 1024) {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 1024 * 64 * 8); (void)hipMalloc(&cyc, 1024 * 8);
  const int stages = 3000;
  for (int rep = 0; rep < 2; ++rep) k<MODE><<<grid, 64>>>(out, cyc, stages);
  (void)hipDeviceSynchronize();
  long long h[1024]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < grid; ++i) s += h[i];
  printf("%-70s %.1f ticks per stage (grid %d)\n", what, s / grid / stages, grid);
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  run<0>("full stage (9 LDS reads, 8 FMAs, 12 readlanes, store)");
  run<1>("without the LDS reads");
  run<2>("without the readlanes and the store (vector operands)");
  run<3>("only the eight dependent FMAs");
  run<4>("LDS reads + FMAs, no readlanes");
  run<0>("full stage", 256);
  run<0>("full stage", 1);
  run<1>("without the LDS reads", 1);
  run<3>("only the eight dependent FMAs", 1);
  run<5>("full stage, LDS reads by lanes 0..5 only");
  run<6>("full stage, LDS reads by lanes 0..15 only");
  run<7>("full stage, LDS reads by lanes 0..31 only");
  return 0;
}
