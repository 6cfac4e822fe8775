// Micro-benchmark (diagnostic, not shipped): what does one wave per SIMD sustain for
// v_mfma_f32_16x16x4_f32 in the access pattern of the fused trajectory kernel?
//   variant 0: MFMAs only (operands in registers)
//   variant 1: + B fragments streamed from an L2-resident packed buffer (ring of 3)
//   variant 2: + A fragment from LDS per block
// Reports cycles per MFMA (ideal 32) per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int NT = 8;

template <int VARIANT, int WAVES, int LOADKIND = 0>
__global__ __launch_bounds__(64 * WAVES) void bench(const float* __restrict__ w, float* out, int nkc, int iters,
                                                    unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) float lds[16 * 520];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16 * 520; i += 64 * WAVES) lds[i] = 0.001f * (i % 97);
  __syncthreads();
  f32x4 acc[NT];
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0, 0, 0, 0};
  const float* wp = w + (size_t)wave * nkc * NT * 256 + lane * 4;
  const float* ap = lds + (lane & 15) * 520 + (lane >> 4) * 4;
  f32x4 b[3][NT];
  auto ldw = [&](const float* ptr) -> f32x4 {
    if (LOADKIND == 1) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ptr));
    return *reinterpret_cast<const f32x4*>(ptr);
  };
  f32x4 a = {1.f, 0.5f, 0.25f, 0.125f};
  for (int t = 0; t < NT; ++t) b[0][t] = b[1][t] = b[2][t] = f32x4{0.1f * t, 0.2f, 0.3f, 0.4f};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (VARIANT >= 1) {
      for (int t = 0; t < NT; ++t) b[0][t] = ldw(wp + (0 * NT + t) * 256);
      for (int t = 0; t < NT; ++t) b[1][t] = ldw(wp + (1 * NT + t) * 256);
      for (int t = 0; t < NT; ++t) b[2][t] = ldw(wp + (2 * NT + t) * 256);
    }
#pragma nounroll
    for (int kc = 0; kc + 3 <= nkc; kc += 3) {
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        if (VARIANT == 2) a = *(const f32x4*)(ap + ((kc + s) & 31) * 16);
        if (VARIANT == 3) {
          const f32x4 an = *(const f32x4*)(ap + ((kc + s + 1) & 31) * 16);   // next block's A fragment, in flight early
          // loads for chunk kc+s+3 interleaved 1:1 with the last row of MFMAs: their issue cost hides in the MFMA shadow
          const bool more = kc + s + 3 < nkc;
          const float* src = wp + (size_t)(more ? kc + s + 3 : kc + s) * NT * 256;
#pragma unroll
          for (int e = 0; e < 3; ++e)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[s][t][e], acc[t], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[s][t][3], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            b[s][t] = ldw(src + t * 256);
            __builtin_amdgcn_sched_barrier(0);
          }
          a = an;
        } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[s][t][e], acc[t], 0, 0, 0);
        if (VARIANT >= 1 && kc + s + 3 < nkc)
          for (int t = 0; t < NT; ++t) b[s][t] = ldw(wp + ((size_t)(kc + s + 3) * NT + t) * 256);
        }
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int t = 0; t < NT; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * WAVES + wave] = t1 - t0;
}

template <int V, int W, int LK = 0>
void run(const char* name, const float* w, float* out, unsigned long long* cyc, int nkc, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  bench<V, W, LK><<<256, 64 * W>>>(w, out, nkc, 2, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  bench<V, W, LK><<<256, 64 * W>>>(w, out, nkc, iters, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(256 * W);
  hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * 256 * W, hipMemcpyDeviceToHost);
  double sum = 0; for (auto v : h) sum += v;
  const double nm = (double)iters * (nkc / 3 * 3) * 4 * NT;
  const double flops = 256.0 * W * nm * 2048;
  printf("%-44s waves/WG %d: %.1f cyc/MFMA/wave  (%.1f per SIMD-MFMA)  %.3f ms  %.1f TFLOP/s\n", name, W, sum / h.size() / nm,
         sum / h.size() / nm / (W / 4), ms, flops / ms / 1e9);
}

int main() {
  const int nkc = 32, W = 8;
  float* w; float* out; unsigned long long* cyc;
  const size_t n = (size_t)W * nkc * NT * 256;     // 2 MB: L2 resident, shared by every workgroup
  hipMalloc(&w, n * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  std::vector<float> h(n); for (size_t i = 0; i < n; ++i) h[i] = (float)(rand() % 1000) * 1e-4f;
  hipMemcpy(w, h.data(), n * 4, hipMemcpyHostToDevice);
  const int iters = 200;
  run<0, 4>("MFMA only", w, out, cyc, nkc, iters);
  run<1, 4>("MFMA + B stream from L2 (ring 3)", w, out, cyc, nkc, iters);
  run<2, 4>("MFMA + B stream + A from LDS", w, out, cyc, nkc, iters);
  run<1, 4, 1>("MFMA + B stream, nontemporal loads", w, out, cyc, nkc, iters);
  run<2, 4, 1>("MFMA + B stream nt + A from LDS", w, out, cyc, nkc, iters);
  run<3, 4, 0>("MFMA + B stream + A LDS, loads interleaved", w, out, cyc, nkc, iters);
  run<3, 4, 1>("MFMA + B stream nt + A LDS, interleaved", w, out, cyc, nkc, iters);
  run<0, 8>("MFMA only", w, out, cyc, nkc, iters);
  run<1, 8>("MFMA + B stream from L2 (ring 3)", w, out, cyc, nkc, iters);
  run<2, 8>("MFMA + B stream + A from LDS", w, out, cyc, nkc, iters);
  return 0;
}
