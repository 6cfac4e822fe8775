// Micro-benchmark (diagnostic, not shipped), round 4: the shipped streaming core itself -- csrc/fused_common.h's
// ring_prime / stream_layer, exactly as the whole-step kernel K1 instantiates them -- in isolation: 256 workgroups of
// 4 waves, every wave walking its own section of a packed image, A fragments from LDS, accumulators kept across
// calls.  Separates what the core costs (cycles per v_mfma_f32_16x16x4_f32) from what the rest of K1 adds.
//   layer 2 : stream_layer<8, 32, DEPTH>   (8 tiles per wave, 32 k-chunks)     alternating walk direction per call
//   heads   : stream_layer<6, 32, DEPTH>   (6 tiles per wave)
//   layer 1 : two stream_layer<8, 8, DEPTH> halves, both rings primed up front
// SETS: how many distinct weight images the calls cycle through (1 = always L2-resident; K1 cycles through two
// networks' 2.36 MB images per XCD-L2 of 4 MB).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../l2hmc_amd/csrc/fused_common.h"
using namespace l2hmc;

template <int NT, int NKC, int DEPTH>
__global__ __launch_bounds__(256) void bench_layer(const float* __restrict__ w, float* out, int calls, int sets,
                                                   size_t set_stride, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) float lds[16 * 520];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16 * 520; i += 256) lds[i] = 0.001f * (i % 97);
  __syncthreads();
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0, 0, 0, 0};
  const float* ap = lds + (lane & 15) * 520 + (lane >> 4) * 4;
  BRing<NT, DEPTH> R;
  const float* wp0 = w + (size_t)__builtin_amdgcn_readfirstlane(wave) * NKC * NT * 256;
  ring_prime<NT, DEPTH>(R, wp0, false, NKC);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int set = 0;
  for (int c = 0; c < calls; ++c) {
    const bool zig = (c & 1) != 0;
    const float* wp = wp0 + (size_t)set * set_stride;
    stream_layer<NT, NKC, DEPTH>(R, wp, [&](int kc) { return *reinterpret_cast<const f32x4*>(ap + (kc & 31) * 16); }, acc, zig);
    if (++set == sets) set = 0;
    // as in K1: the next call's ring is primed before the "epilogue" (here: a barrier)
    ring_prime<NT, DEPTH>(R, wp0 + (size_t)set * set_stride, !zig, NKC);
    __syncthreads();
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int t = 0; t < NT; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) s += R.b[d][0][0] * 1e-30f;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int NT, int NKC, int DEPTH>
void run(const char* name, const float* w, float* out, unsigned long long* cyc, int calls, int sets, size_t set_stride) {
  bench_layer<NT, NKC, DEPTH><<<256, 256>>>(w, out, 4, sets, set_stride, cyc);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  bench_layer<NT, NKC, DEPTH><<<256, 256>>>(w, out, calls, sets, set_stride, cyc);
  (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(1024);
  (void)hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * 1024, hipMemcpyDeviceToHost);
  double sum = 0; for (auto v : h) sum += v;
  const double nm = (double)calls * NKC * 4 * NT;
  printf("%-46s depth %d, %d image(s): %.1f cyc/MFMA  %.3f ms  %.1f TFLOP/s\n", name, DEPTH, sets, sum / 1024 / nm, ms,
         1024.0 * nm * 2048 / ms / 1e9);
  fflush(stdout);
}

int main() {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  float* w; float* out; unsigned long long* cyc;
  const size_t set_stride = (size_t)4 * 32 * 8 * 256 + 4096;      // one image: 4 waves x 32 chunks x 8 tiles x 1 KiB = 1 MB (+ pad)
  const int max_sets = 6;
  (void)hipMalloc(&w, set_stride * max_sets * 4); (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 1024 * 8);
  std::vector<float> h(set_stride * max_sets); for (auto& v : h) v = (float)(rand() % 1000) * 1e-4f;
  (void)hipMemcpy(w, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep)
    for (int sets : {1, 5}) {          // 5 x 1 MB images cycled: more than one XCD's L2 holds, like K1's two networks
      run<8, 32, 3>("layer 2 (8 tiles x 32 chunks)", w, out, cyc, 200, sets, set_stride);
      run<8, 32, 4>("layer 2 (8 tiles x 32 chunks)", w, out, cyc, 200, sets, set_stride);
      run<6, 32, 4>("heads (6 tiles x 32 chunks)", w, out, cyc, 200, sets, set_stride);
      run<6, 32, 5>("heads (6 tiles x 32 chunks)", w, out, cyc, 200, sets, set_stride);
      run<8, 8, 3>("first-layer half (8 tiles x 8 chunks)", w, out, cyc, 800, sets, set_stride);
    }
  return 0;
}
