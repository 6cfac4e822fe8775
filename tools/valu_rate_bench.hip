// Diagnostic (not shipped): issue rate of the fp32 VALU forms the conv front-end uses, per wave and with 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate_bench.hip -o tools/_diag/valu_rate_bench && tools/_diag/valu_rate_bench
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x2 = __attribute__((ext_vector_type(2))) float;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters, float s) {
  f32x2 a[8], b = {s, s * 0.5f}, c = {1.0f + s, 1.0f - s};
  float f[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = f32x2{(float)threadIdx.x + i, (float)i};
#pragma unroll
  for (int i = 0; i < 16; ++i) f[i] = (float)threadIdx.x + i;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
        if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[i]) : "v"(b), "v"(c));
        if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(a[i]) : "v"(b), "v"(c));
        if (MODE == 3) {
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[2 * i]) : "v"(b[0]), "v"(c[0]));
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[2 * i + 1]) : "v"(b[1]), "v"(c[1]));
        }
        if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
        if (MODE == 5) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
        if (MODE == 6) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[0]) : "v"(b), "v"(c));          // ONE dependent chain
        if (MODE == 7) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i & 1]) : "v"(b), "v"(c));      // two chains
        if (MODE == 8) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i & 3]) : "v"(b), "v"(c));      // four chains
        if (MODE == 9) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[0]) : "v"(b[0]), "v"(c[0]));       // one dependent chain, unpacked
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc += a[i][0] + a[i][1];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += f[i];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* name, int wgs, float* out, unsigned long long* cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, out, cyc, iters, 1e-3f);
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, out, cyc, iters, 1e-3f);
  hipDeviceSynchronize();
  unsigned long long h[4096];
  hipMemcpy(h, cyc, sizeof(unsigned long long) * wgs, hipMemcpyDeviceToHost);
  double m = 0;
  for (int i = 0; i < wgs; ++i) m += (double)h[i];
  m /= wgs;
  const double ops = (double)iters * 32 * (MODE == 3 ? 2 : 1);     // VALU instructions per wave
  printf("%-44s WGs %4d: %8.0f cycles, %.2f cycles per instruction per wave, %.2f lane-MACs(or ops)/cycle/SIMD\n", name, wgs, m,
         m / ops, (double)iters * 32 * 2 * 64 * (wgs >= 1024 ? 4 : 1) / m);
  fflush(stdout);
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, sizeof(float) * 4096 * 256);
  hipMalloc(&cyc, sizeof(unsigned long long) * 4096);
  for (int wgs : {256, 1024}) {     // one wave per SIMD / four waves per SIMD (256 CUs)
    run<0>("v_pk_fma_f32", wgs, out, cyc);
    run<1>("v_pk_fma_f32 op_sel_hi:[0,1,1] (broadcast lo)", wgs, out, cyc);
    run<2>("v_pk_fma_f32 op_sel:[1,0,0] (broadcast hi)", wgs, out, cyc);
    run<3>("2 x v_fma_f32", wgs, out, cyc);
    run<4>("v_pk_mul_f32", wgs, out, cyc);
    run<5>("v_pk_add_f32", wgs, out, cyc);
    run<6>("v_pk_fma_f32, ONE dependent chain", wgs, out, cyc);
    run<7>("v_pk_fma_f32, two chains", wgs, out, cyc);
    run<8>("v_pk_fma_f32, four chains", wgs, out, cyc);
    run<9>("v_fma_f32, one dependent chain", wgs, out, cyc);
  }
  return 0;
}
