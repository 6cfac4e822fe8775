// Diagnostic (not shipped): do packed VALU instructions of one wave issue while another wave of the SAME SIMD keeps the
// matrix pipe busy?  512-thread workgroups, one per CU: waves 0-3 run role A, waves 4-7 role B (wave w sits on SIMD w % 4).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap_bench.hip -o tools/_diag/mfma_valu_overlap_bench
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
// role: 0 idle, 1 MFMA stream (4 independent accumulators), 2 packed-fma stream (8 independent chains),
//       3 one MFMA then 8 packed fma, repeated (one wave feeding both pipes)
template <int RA, int RB>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, float s) {
  const int wave = threadIdx.x >> 6;
  const int role = wave < 4 ? RA : RB;
  f32x4 acc[4];
  f32x2 a[8], b = {s, 0.5f * s}, c = {1.f + s, 1.f - s};
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 8; ++i) a[i] = f32x2{(float)threadIdx.x + i, (float)i};
  const float x = (float)threadIdx.x * s, y = 1.f + s;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (role == 7) __builtin_amdgcn_s_setprio(3);
  if (role == 1 || role == 7) {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
  } else if (role == 2) {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int r = 0; r < 32; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
  } else if (role == 3) {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[r & 3]) : "v"(x), "v"(y));
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
      }
  }
  else if (role == 4) {          // as 3, accumulators in AGPRs
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[r & 3]) : "v"(x), "v"(y));
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
      }
  } else if (role == 5) {        // as 3 with unpacked fma (16 of them)
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[r & 3]) : "v"(x), "v"(y));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i][0]) : "v"(b[0]), "v"(c[0]));
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i][1]) : "v"(b[1]), "v"(c[1]));
        }
      }
  } else if (role == 6) {        // 8 packed fma FIRST, then 4 MFMAs back to back (coarser interleave), x 8
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
      }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
  for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) r += a[i][0] + a[i][1];
  out[blockIdx.x * 512 + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int RA, int RB>
void run(const char* name, float* out, unsigned long long* cyc) {
  const int iters = 500, wgs = 256;
  hipLaunchKernelGGL((k<RA, RB>), dim3(wgs), dim3(512), 0, 0, out, cyc, iters, 1e-3f);
  hipLaunchKernelGGL((k<RA, RB>), dim3(wgs), dim3(512), 0, 0, out, cyc, iters, 1e-3f);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return; }
  static unsigned long long h[256 * 8];
  if (hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return;
  double ma = 0, mb = 0;
  for (int i = 0; i < wgs; ++i) { for (int w = 0; w < 4; ++w) ma += (double)h[i * 8 + w]; for (int w = 4; w < 8; ++w) mb += (double)h[i * 8 + w]; }
  ma /= wgs * 4; mb /= wgs * 4;
  auto per = [&](int role, double m) {
    if (role == 1 || role == 7) printf("  MFMA wave%s: %.1f cycles per MFMA", role == 7 ? " (s_setprio 3)" : "", m / (iters * 32.0));
    if (role == 2) printf("  VALU wave: %.2f cycles per v_pk_fma_f32", m / (iters * 256.0));
    if (role >= 3) printf("  mixed wave: %.1f cycles per (1 MFMA + 8 v_pk_fma_f32)", m / (iters * 32.0));
  };
  printf("%-52s", name); per(RA, ma); per(RB, mb); printf("\n"); fflush(stdout);
}
int main() {
  float* out; unsigned long long* cyc;
  if (hipMalloc(&out, sizeof(float) * 256 * 512) != hipSuccess || hipMalloc(&cyc, sizeof(unsigned long long) * 256 * 8) != hipSuccess) return 1;
  run<1, 0>("MFMA stream alone (one wave per SIMD)", out, cyc);
  run<2, 0>("packed-fma stream alone (one wave per SIMD)", out, cyc);
  run<1, 2>("MFMA wave + packed-fma wave on the same SIMD", out, cyc);
  run<1, 1>("two MFMA waves on the same SIMD", out, cyc);
  run<2, 2>("two packed-fma waves on the same SIMD", out, cyc);
  run<7, 1>("two MFMA waves, the first at s_setprio 3", out, cyc);
  run<7, 2>("MFMA wave at s_setprio 3 + packed-fma wave", out, cyc);
  run<3, 0>("one wave: 1 MFMA + 8 packed fma, repeated", out, cyc);
  run<3, 3>("two such waves on the same SIMD", out, cyc);
  run<4, 0>("one wave, accumulators in AGPRs", out, cyc);
  run<4, 4>("two such waves", out, cyc);
  run<5, 0>("one wave: 1 MFMA + 16 v_fma_f32", out, cyc);
  run<6, 0>("one wave: 32 packed fma then 4 MFMAs (same totals)", out, cyc);
  run<6, 6>("two such waves", out, cyc);
  return 0;
}
