// Microbenchmark: what one instruction of each kind costs a lone wave (one wave per SIMD, the occupancy of the solve kernels):
// issue cost of a run of independent instructions, and the latency of a dependent chain.  Ticks of s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>

enum Kind { FMA_IND, FMA_DEP_MUL, FMA_DEP_ADD, ADD_U32, CNDMASK, READLANE, READLANE_DEP, LDS_R64, LDS_R128, LDS_W64, LDS_W128, BPERM, LDS_R64_DEP,
            BPERM_DEP, LDS_R64_VADDR, DPP_MOV, RCP64, LDS_R64_X2, NKIND };
const char* NAMES[] = {"v_fma_f64, 8 independent chains", "v_fma_f64, dependent through a multiplicand", "v_fma_f64, dependent through the addend",
  "v_add_u32, independent", "v_cndmask_b32, independent", "v_readlane_b32, independent", "v_readlane_b32 -> v_fma_f64 (scalar operand) -> v_readlane (dependent round trip, per pair of readlanes + 1 fma)",
  "ds_read_b64, independent, immediate offsets", "ds_read_b128, independent, immediate offsets", "ds_write_b64, independent", "ds_write_b128, independent",
  "ds_bpermute_b32, independent", "ds_read_b64, address depends on the previous read (latency)", "ds_bpermute_b32, dependent chain (latency)",
  "ds_read_b64 with a v_add_u32 address each", "v_mov_b32 dpp row_shr, dependent chain", "v_rcp_f64 dependent chain", "ds_read2_b64 (two doubles, one instruction), independent"};

template <int KIND>
__global__ __launch_bounds__(64) void k(double* out, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) double sm[64 * 34];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 34; i += 64) sm[i] = (KIND == LDS_R64_DEP) ? 0.0 : 1e-3 * ((i * 7) % 13) + 1.0;
  int* smi = (int*)sm;
  if (KIND == LDS_R64_DEP) for (int i = lane; i < 64 * 34 * 2; i += 64) smi[i] = (lane * 16) % 4096;
  __syncthreads();
  double a[8]; for (int i = 0; i < 8; ++i) a[i] = lane * 1e-3 + 1.0 + i;
  int u[8]; for (int i = 0; i < 8; ++i) u[i] = lane + i;
  const double m = 1.0000001, c = 1e-9;
  double acc = 0; int iacc = 0;
  const double* base = sm + lane * 2;
  int addr = lane * 8;
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma clang loop unroll(disable)
  for (int it = 0; it < iters; ++it) {
    if (KIND == FMA_IND) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = __builtin_fma(a[j], m, c);
    } else if (KIND == FMA_DEP_MUL) {
#pragma unroll
      for (int r = 0; r < 32; ++r) a[0] = __builtin_fma(a[0], m, c);
    } else if (KIND == FMA_DEP_ADD) {
#pragma unroll
      for (int r = 0; r < 32; ++r) a[0] = __builtin_fma(a[1], m, a[0]);
    } else if (KIND == ADD_U32) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[j]) : "v"(lane));
    } else if (KIND == CNDMASK) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[j]) : "v"(lane));
    } else if (KIND == READLANE) {
#pragma unroll
      for (int r = 0; r < 32; ++r) { int s; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(u[r & 7])); iacc += s; }
    } else if (KIND == READLANE_DEP) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long b = __builtin_bit_cast(long long, a[0]);
        const int lo = __builtin_amdgcn_readlane((int)b, 3), hi = __builtin_amdgcn_readlane((int)(b >> 32), 3);
        const double s = __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
        a[0] = __builtin_fma(s, a[1], a[2]);
      }
    } else if (KIND == LDS_R64) {
#pragma unroll
      for (int r = 0; r < 32; ++r) { double v; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(r * 512)); acc += 0; a[r & 7] = v; }
      asm volatile("s_waitcnt lgkmcnt(0)");
    } else if (KIND == LDS_R64_X2) {
#pragma unroll
      for (int r = 0; r < 32; ++r) { double __attribute__((ext_vector_type(2))) v; asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(r * 2), "n"(r * 2 + 65)); a[r & 7] = v.x; }
      asm volatile("s_waitcnt lgkmcnt(0)");
    } else if (KIND == LDS_R128) {
#pragma unroll
      for (int r = 0; r < 32; ++r) { double __attribute__((ext_vector_type(2))) v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr * 2), "n"(r * 512)); a[r & 7] = v.x; }
      asm volatile("s_waitcnt lgkmcnt(0)");
    } else if (KIND == LDS_W64) {
#pragma unroll
      for (int r = 0; r < 32; ++r) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(addr), "v"(a[r & 7]), "n"(r * 512));
      asm volatile("s_waitcnt lgkmcnt(0)");
    } else if (KIND == LDS_W128) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { double __attribute__((ext_vector_type(2))) v = {a[r & 7], a[(r + 1) & 7]}; asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(addr * 2), "v"(v), "n"(r * 1024)); }
      asm volatile("s_waitcnt lgkmcnt(0)");
    } else if (KIND == BPERM) {
#pragma unroll
      for (int r = 0; r < 32; ++r) { int v; asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(v) : "v"(addr / 2), "v"(u[r & 7])); u[(r + 3) & 7] ^= 0; iacc += 0; if (r == 31) { asm volatile("s_waitcnt lgkmcnt(0)"); iacc += v; } }
    } else if (KIND == LDS_R64_DEP) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { long long v; asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr)); addr = (int)v; }
    } else if (KIND == BPERM_DEP) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { int v; asm volatile("ds_bpermute_b32 %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr / 2), "v"(u[0])); u[0] = v; }
    } else if (KIND == LDS_R64_VADDR) {
#pragma unroll
      for (int r = 0; r < 32; ++r) { double v; int ad; asm volatile("v_add_u32 %0, %1, %2" : "=v"(ad) : "v"(addr), "v"(u[r & 7])); asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(ad & 0x3ff8)); a[r & 7] = v; }
      asm volatile("s_waitcnt lgkmcnt(0)");
    } else if (KIND == DPP_MOV) {
#pragma unroll
      for (int r = 0; r < 32; ++r) asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[0]));
    } else if (KIND == RCP64) {
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[0]));
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = acc + iacc + addr; for (int i = 0; i < 8; ++i) s += a[i] + u[i];
  out[blockIdx.x * 64 + lane] = s;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND>
void run(int per_iter, int grid) {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 1024 * 64 * 8); (void)hipMalloc(&cyc, 1024 * 8);
  const int iters = 500;
  for (int rep = 0; rep < 2; ++rep) k<KIND><<<grid, 64>>>(out, cyc, iters);
  (void)hipDeviceSynchronize();
  long long h[1024]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < grid; ++i) s += h[i];
  printf("%6.1f ticks  %s (grid %d)\n", s / grid / ((double)iters * per_iter), NAMES[KIND], grid);
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  for (int grid : {1, 1024}) {
    run<FMA_IND>(32, grid); run<FMA_DEP_MUL>(32, grid); run<FMA_DEP_ADD>(32, grid); run<ADD_U32>(32, grid); run<CNDMASK>(32, grid);
    run<READLANE>(32, grid); run<READLANE_DEP>(16, grid); run<LDS_R64>(32, grid); run<LDS_R64_X2>(32, grid); run<LDS_R128>(32, grid); run<LDS_W64>(32, grid);
    run<LDS_W128>(16, grid); run<BPERM>(32, grid); run<LDS_R64_DEP>(16, grid); run<BPERM_DEP>(16, grid); run<LDS_R64_VADDR>(32, grid);
    run<DPP_MOV>(32, grid); run<RCP64>(16, grid);
  }
  return 0;
}
