// Micro-benchmark (diagnostic, not shipped): where does the B-fragment stream of the fused trajectory kernel
// lose its ~20 % (32 -> 38 cycles per v_mfma_f32_16x16x4_f32)?  Same loop as tools/mfma_bench.hip variant 1
// (MFMA + B stream, ring of 3, one wave per SIMD, 256 workgroups), with the address stream varied:
//   mode 0: every workgroup walks the same packed buffer in the same order (what the kernel does)
//   mode 1: every workgroup starts its walk at a different k-chunk (rotation by in-XCD workgroup index)
//   mode 2: all loads hit the wave's first 3 chunks (24 KiB per wave: vector-L1 / TA resident) -> issue cost only
//   mode 3: each XCD-local workgroup reads its OWN private copy of the buffer (no line shared between CUs)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int NT = 8;

template <int MODE>
__global__ __launch_bounds__(256) void bench(const float* __restrict__ w, float* out, int nkc, int iters,
                                             unsigned long long* cyc, size_t copy_stride) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 acc[NT];
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0, 0, 0, 0};
  const float* wp = w + (size_t)wave * nkc * NT * 256 + lane * 4;
  if (MODE == 3) wp += (size_t)(blockIdx.x >> 3) * copy_stride;      // blockIdx & 7 ~ XCD, >> 3 = index inside it
  const int rot = MODE == 1 ? (int)(((blockIdx.x >> 3) * 5) % nkc) : 0;
  auto chunk = [&](int kc) -> const float* {
    int c = kc;
    if (MODE == 1) { c += rot; if (c >= nkc) c -= nkc; }
    if (MODE == 2) c = kc % 3;
    return wp + (size_t)c * NT * 256;
  };
  f32x4 b[3][NT];
  const f32x4 a = {1.f, 0.5f, 0.25f, 0.125f};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    for (int s = 0; s < 3; ++s)
      for (int t = 0; t < NT; ++t) b[s][t] = *reinterpret_cast<const f32x4*>(chunk(s) + t * 256);
#pragma nounroll
    for (int kc = 0; kc + 3 <= nkc; kc += 3) {
#pragma unroll
      for (int s = 0; s < 3; ++s) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[s][t][e], acc[t], 0, 0, 0);
        if (kc + s + 3 < nkc)
          for (int t = 0; t < NT; ++t) b[s][t] = *reinterpret_cast<const f32x4*>(chunk(kc + s + 3) + t * 256);
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int t = 0; t < NT; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

// mode 4: the loads of chunk kc+2 are issued DURING block kc, one per 4 MFMAs (sched_group_barrier), into the ring
// slot block kc-1 has just released; branch-free (the tail re-loads the last chunk).  A fragment from LDS as in
// the kernel.  What the compiler makes of the plain ring (mode 0) is 3 blocks of MFMAs followed by 24 loads.
template <int S, int N>
__device__ __forceinline__ void block4(f32x4 (&b)[3][NT], const f32x4 a, f32x4 (&acc)[NT], const float* __restrict__ src) {
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[S][t][e], acc[t], 0, 0, 0);
#pragma unroll
  for (int t = 0; t < NT; ++t) b[N][t] = *reinterpret_cast<const f32x4*>(src + t * 256);
#pragma unroll
  for (int g = 0; g < NT; ++g) {
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // 4 MFMA
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
  }
}

template <bool LDSA>
__global__ __launch_bounds__(256) void bench4(const float* __restrict__ w, float* out, int nkc, int iters,
                                              unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) float lds[16 * 520];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16 * 520; i += 256) lds[i] = 0.001f * (i % 97);
  __syncthreads();
  f32x4 acc[NT];
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0, 0, 0, 0};
  const float* wp = w + (size_t)wave * nkc * NT * 256 + lane * 4;
  const float* ap = lds + (lane & 15) * 520 + (lane >> 4) * 4;
  f32x4 b[3][NT];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    for (int t = 0; t < NT; ++t) b[0][t] = *reinterpret_cast<const f32x4*>(wp + t * 256);
    for (int t = 0; t < NT; ++t) b[1][t] = *reinterpret_cast<const f32x4*>(wp + (NT + t) * 256);
    f32x4 a0 = LDSA ? *(const f32x4*)(ap) : f32x4{1.f, 0.5f, 0.25f, 0.125f}, a1 = a0;
#pragma nounroll
    for (int kc = 0; kc + 3 <= nkc; kc += 3) {
      if (LDSA) a1 = *(const f32x4*)(ap + ((kc + 1) & 31) * 16);
      block4<0, 2>(b, a0, acc, wp + (size_t)min(kc + 2, nkc - 1) * NT * 256);
      if (LDSA) a0 = *(const f32x4*)(ap + ((kc + 2) & 31) * 16);
      block4<1, 0>(b, a1, acc, wp + (size_t)min(kc + 3, nkc - 1) * NT * 256);
      if (LDSA) a1 = *(const f32x4*)(ap + ((kc + 3) & 31) * 16);
      block4<2, 1>(b, a0, acc, wp + (size_t)min(kc + 4, nkc - 1) * NT * 256);
      a0 = a1;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int t = 0; t < NT; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <bool LDSA>
void run4(const char* name, const float* w, float* out, unsigned long long* cyc, int nkc, int iters) {
  bench4<LDSA><<<256, 256>>>(w, out, nkc, 2, cyc);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  bench4<LDSA><<<256, 256>>>(w, out, nkc, iters, cyc);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(1024);
  (void)hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * 1024, hipMemcpyDeviceToHost);
  double sum = 0; for (auto v : h) sum += v;
  const double nm = (double)iters * (nkc / 3 * 3) * 4 * NT;
  printf("%-70s nkc %3d: %.1f cyc/MFMA  %.3f ms  %.1f TFLOP/s  (%.0f GB/s per CU)\n", name, nkc, sum / 1024 / nm, ms,
         1024.0 * nm * 2048 / ms / 1e9, 4.0 * nm / 32 * 8192 / ms / 1e6);
}

// mode 5: TWO waves per SIMD that split K, not N: waves w and w+4 (same SIMD) own the same 8 output tiles and each
// streams half of the k-chunks, so the B bytes, the A reads and the MFMAs per SIMD are what mode 0 has -- but while
// one wave's instruction stream is blocked issuing a load, the other wave's MFMAs keep the pipe busy.
template <bool LDSA>
__global__ __launch_bounds__(512) void bench5(const float* __restrict__ w, float* out, int nkc, int iters,
                                              unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) float lds[16 * 520];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = wave & 3, half = wave >> 2;
  for (int i = threadIdx.x; i < 16 * 520; i += 512) lds[i] = 0.001f * (i % 97);
  __syncthreads();
  f32x4 acc[NT];
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0, 0, 0, 0};
  const int hk = nkc / 2;
  const float* wp = w + ((size_t)col * nkc + (size_t)half * hk) * NT * 256 + lane * 4;
  const float* ap = lds + (lane & 15) * 520 + (lane >> 4) * 4 + half * hk * 16;
  f32x4 b[3][NT];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    for (int s = 0; s < 3; ++s)
      for (int t = 0; t < NT; ++t) b[s][t] = *reinterpret_cast<const f32x4*>(wp + ((size_t)s * NT + t) * 256);
    f32x4 a = f32x4{1.f, 0.5f, 0.25f, 0.125f};
#pragma nounroll
    for (int kc = 0; kc + 3 <= hk; kc += 3) {
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        if (LDSA) a = *(const f32x4*)(ap + ((kc + s) & 15) * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[s][t][e], acc[t], 0, 0, 0);
        if (kc + s + 3 < hk)
          for (int t = 0; t < NT; ++t) b[s][t] = *reinterpret_cast<const f32x4*>(wp + ((size_t)(kc + s + 3) * NT + t) * 256);
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int t = 0; t < NT; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <bool LDSA>
void run5(const char* name, const float* w, float* out, unsigned long long* cyc, int nkc, int iters) {
  bench5<LDSA><<<256, 512>>>(w, out, nkc, 2, cyc);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  bench5<LDSA><<<256, 512>>>(w, out, nkc, iters, cyc);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2048);
  (void)hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * 2048, hipMemcpyDeviceToHost);
  double sum = 0; for (auto v : h) sum += v;
  const int hk = nkc / 2;
  const double nm = (double)iters * (hk / 3 * 3) * 4 * NT;          // MFMAs per wave; a SIMD runs two such waves
  printf("%-70s nkc %3d: %.1f cyc per SIMD-MFMA  %.3f ms  %.1f TFLOP/s  (%.0f GB/s per CU)\n", name, nkc, sum / 2048 / (2 * nm),
         ms, 2048.0 * nm * 2048 / ms / 1e9, 8.0 * nm / 32 * 8192 / ms / 1e6);
}

// mode 6: THIRTY-TWO rows per workgroup -- every weight fragment feeds two MFMAs (row groups 0 and 1), so the load
// issue cost per MFMA halves; 16 accumulator tiles per wave, two A fragments per k-chunk from LDS.
template <bool LDSA>
__global__ __launch_bounds__(256) void bench6(const float* __restrict__ w, float* out, int nkc, int iters,
                                              unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) float lds[32 * 520];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 32 * 520; i += 256) lds[i] = 0.001f * (i % 97);
  __syncthreads();
  f32x4 acc[2][NT];
  for (int g = 0; g < 2; ++g)
    for (int t = 0; t < NT; ++t) acc[g][t] = f32x4{0, 0, 0, 0};
  const float* wp = w + (size_t)wave * nkc * NT * 256 + lane * 4;
  const float* ap = lds + (lane & 15) * 520 + (lane >> 4) * 4;
  f32x4 b[3][NT];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    for (int s = 0; s < 3; ++s)
      for (int t = 0; t < NT; ++t) b[s][t] = *reinterpret_cast<const f32x4*>(wp + ((size_t)s * NT + t) * 256);
    f32x4 a0 = f32x4{1.f, 0.5f, 0.25f, 0.125f}, a1 = f32x4{0.5f, 0.25f, 0.125f, 1.f};
#pragma nounroll
    for (int kc = 0; kc + 3 <= nkc; kc += 3) {
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        if (LDSA) {
          a0 = *(const f32x4*)(ap + ((kc + s) & 31) * 16);
          a1 = *(const f32x4*)(ap + 16 * 520 + ((kc + s) & 31) * 16);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], b[s][t][e], acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], b[s][t][e], acc[1][t], 0, 0, 0);
          }
        if (kc + s + 3 < nkc)
          for (int t = 0; t < NT; ++t) b[s][t] = *reinterpret_cast<const f32x4*>(wp + ((size_t)(kc + s + 3) * NT + t) * 256);
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int g = 0; g < 2; ++g)
    for (int t = 0; t < NT; ++t) s += acc[g][t][0] + acc[g][t][1] + acc[g][t][2] + acc[g][t][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <bool LDSA>
void run6(const char* name, const float* w, float* out, unsigned long long* cyc, int nkc, int iters) {
  bench6<LDSA><<<256, 256>>>(w, out, nkc, 2, cyc);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  bench6<LDSA><<<256, 256>>>(w, out, nkc, iters, cyc);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(1024);
  (void)hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * 1024, hipMemcpyDeviceToHost);
  double sum = 0; for (auto v : h) sum += v;
  const double nm = (double)iters * (nkc / 3 * 3) * 4 * NT * 2;
  printf("%-70s nkc %3d: %.1f cyc/MFMA  %.3f ms  %.1f TFLOP/s  (%.0f GB/s per CU)\n", name, nkc, sum / 1024 / nm, ms,
         1024.0 * nm * 2048 / ms / 1e9, 4.0 * nm / 64 * 8192 / ms / 1e6);
}

template <int MODE>
void run(const char* name, const float* w, float* out, unsigned long long* cyc, int nkc, int iters, size_t stride) {
  bench<MODE><<<256, 256>>>(w, out, nkc, 2, cyc, stride);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  bench<MODE><<<256, 256>>>(w, out, nkc, iters, cyc, stride);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(1024);
  (void)hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * 1024, hipMemcpyDeviceToHost);
  double sum = 0; for (auto v : h) sum += v;
  const double nm = (double)iters * (nkc / 3 * 3) * 4 * NT;
  printf("%-70s nkc %3d: %.1f cyc/MFMA  %.3f ms  %.1f TFLOP/s  (%.0f GB/s per CU)\n", name, nkc, sum / 1024 / nm, ms,
         1024.0 * nm * 2048 / ms / 1e9, 4.0 * nm / 32 * 8192 / ms / 1e6);
}

int main() {
  float* w; float* out; unsigned long long* cyc;
  const int nkc_max = 96;                              // 96 chunks x 8 KiB x 4 waves = 3 MB per copy
  const size_t copy = (size_t)4 * nkc_max * NT * 256;  // floats per copy
  const size_t n = copy * 32;                          // 32 private copies for mode 3 (96 MB)
  (void)hipMalloc(&w, n * 4); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 2048 * 8);
  std::vector<float> h(copy); for (size_t i = 0; i < copy; ++i) h[i] = (float)(rand() % 1000) * 1e-4f;
  for (int c = 0; c < 32; ++c) (void)hipMemcpy(w + c * copy, h.data(), copy * 4, hipMemcpyHostToDevice);
  for (int nkc : {36, 96}) {
    const int iters = nkc == 32 ? 300 : 100;
    run<0>("mode 0: shared buffer, same order in every workgroup (the kernel)", w, out, cyc, nkc, iters, copy);
    run<1>("mode 1: shared buffer, start chunk rotated per workgroup", w, out, cyc, nkc, iters, copy);
    run<2>("mode 2: loads hit 24 KiB per wave (issue cost only)", w, out, cyc, nkc, iters, copy);
    run<3>("mode 3: private copy per in-XCD workgroup (no shared lines)", w, out, cyc, nkc, iters, copy);
    run4<false>("mode 4: loads interleaved 1 per 4 MFMAs (sched_group_barrier), A in registers", w, out, cyc, nkc, iters);
    run4<true>("mode 4: loads interleaved 1 per 4 MFMAs, A fragment from LDS", w, out, cyc, nkc, iters);
    run5<false>("mode 5: two waves per SIMD splitting K, A in registers", w, out, cyc, nkc, iters);
    run5<true>("mode 5: two waves per SIMD splitting K, A fragment from LDS", w, out, cyc, nkc, iters);
    run6<false>("mode 6: 32 rows per workgroup (a weight fragment feeds 2 MFMAs), A in registers", w, out, cyc, nkc, iters);
    run6<true>("mode 6: 32 rows per workgroup, both A fragments from LDS", w, out, cyc, nkc, iters);
  }
  return 0;
}
