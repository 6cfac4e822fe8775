// Micro-benchmark (diagnostic, not shipped): where does the split-k TN weight-gradient product (train.hip,
// gemm_tn_kernel: 128 x 128 tile, 4 waves, 16 rows per stage, v_mfma_f32_32x32x2_f32) lose its ~28 %?
// Same stage loop, pieces switched on one at a time; two workgroups per CU as in the product's launch.
//   variant 0: the 32 MFMAs of a stage only (operands in registers)
//   variant 1: + the 32 operand reads from LDS
//   variant 2: + one barrier per stage
//   variant 3: + the 4 x 16-byte LDS stores of the next stage (from registers)
//   variant 4: + the 4 x 16-byte global loads that feed those stores
// Reports cycles per MFMA per SIMD (ideal 64).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int BKR = 16, LDT = 160;

template <int V, int MF>
__global__ __launch_bounds__(256) void bench(const float* __restrict__ P, const float* __restrict__ Q, int ld,
                                             int stages, float* out, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * BKR * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, r = lane & 31;
  for (int i = tid; i < 2 * 2 * BKR * LDT; i += 256) lds[i] = 0.001f * (i % 97);
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  f32x4 rp[2] = {f32x4{1, 2, 3, 4}, f32x4{1, 2, 3, 4}}, rq[2] = {f32x4{1, 2, 3, 4}, f32x4{1, 2, 3, 4}};
  const int64_t rbeg = (int64_t)(blockIdx.x >> 4) * stages * BKR;      // the 16 tiles of a split share its rows
  const int m0 = (blockIdx.x & 3) * 128;
  float av[BKR / 2][2], bv[BKR / 2][2];
  for (int kk = 0; kk < BKR / 2; ++kk) {
    av[kk][0] = 0.5f + kk; av[kk][1] = 0.25f + kk; bv[kk][0] = 0.125f * kk; bv[kk][1] = 1.f;
  }
  int cur = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma nounroll
  for (int st = 0; st < stages; ++st) {
    if (V >= 4) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = tid + i * 256;
        const int row = c >> 5, col = (c & 31) * 4;
        const int64_t gr = rbeg + (int64_t)st * BKR + row;
        rp[i] = *reinterpret_cast<const f32x4*>(P + gr * ld + m0 + col);
        rq[i] = *reinterpret_cast<const f32x4*>(Q + gr * ld + ((blockIdx.x >> 2) & 3) * 128 + col);
      }
    }
    const float* ps = lds + cur * 2 * BKR * LDT + wm * 64 + r;
    const float* qs = lds + cur * 2 * BKR * LDT + BKR * LDT + wn * 64 + r;
    if (V >= 1) {
#pragma unroll
      for (int kk = 0; kk < BKR / 2; ++kk) {
        const int k = 2 * kk + half;
        av[kk][0] = ps[k * LDT]; av[kk][1] = ps[k * LDT + 32];
        bv[kk][0] = qs[k * LDT]; bv[kk][1] = qs[k * LDT + 32];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int kk = 0; kk < BKR / 2; ++kk) {
      if (MF == 0) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk][0], bv[kk][0], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk][0], bv[kk][1], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk][1], bv[kk][0], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk][1], bv[kk][1], acc[1][1], 0, 0, 0);
      }
    }
    if (V >= 3) {
      float* pd = lds + (cur ^ 1) * 2 * BKR * LDT;
      float* qd = pd + BKR * LDT;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = tid + i * 256;
        const int row = c >> 5, col = (c & 31) * 4;
        *reinterpret_cast<f32x4*>(pd + row * LDT + col) = rp[i];
        *reinterpret_cast<f32x4*>(qd + row * LDT + col) = rq[i];
      }
    }
    if (V >= 2) __syncthreads();
    if (V >= 1) cur ^= 1;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  out[blockIdx.x * 256 + tid] = s + rp[0][0] + rq[1][3];
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

// The transposed staging of round 3 (train.hip): tiles in LDS as [column][k] (stride 20), fragments by ds_read_b128
constexpr int LDK = BKR + 4;
__device__ inline bool isq_of(int t) { return t >= 128; }
template <int V>
__global__ __launch_bounds__(256) void bench_t(const float* __restrict__ P, const float* __restrict__ Q, int ld,
                                               int stages, float* out, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * 128 * LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, r = lane & 31;
  for (int i = tid; i < 2 * 2 * 128 * LDK; i += 256) lds[i] = 0.001f * (i % 97);
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  f32x4 rg[4] = {f32x4{1, 2, 3, 4}, f32x4{1, 2, 3, 4}, f32x4{1, 2, 3, 4}, f32x4{1, 2, 3, 4}};
  const int64_t rbeg = (int64_t)(blockIdx.x >> 4) * stages * BKR;
  const int m0 = (isq_of(threadIdx.x) ? (blockIdx.x >> 2) & 3 : blockIdx.x & 3) * 128;
  const bool isq = tid >= 128;
  const int kg = (lane >> 2) & 3, cg = 16 * (wave & 1) + 4 * (lane >> 4) + (lane & 3);
  const float* gsrc = isq ? Q : P;
  f32x4 av[2][2], bv[2][2];
  for (int s = 0; s < 2; ++s)
    for (int i = 0; i < 2; ++i) { av[s][i] = f32x4{0.5f, 1.f, 2.f, 3.f}; bv[s][i] = f32x4{0.25f, 1.f, 0.5f, 2.f}; }
  int cur = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma nounroll
  for (int st = 0; st < stages; ++st) {
    if (V >= 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t gr = rbeg + (int64_t)st * BKR + 4 * kg + i;
        rg[i] = *reinterpret_cast<const f32x4*>(gsrc + gr * ld + m0 + 4 * cg);
      }
    }
    const float* ps = lds + cur * 2 * 128 * LDK + (wm * 64 + r) * LDK + 4 * half;
    const float* qs = lds + cur * 2 * 128 * LDK + 128 * LDK + (wn * 64 + r) * LDK + 4 * half;
    if (V >= 1) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        av[s][0] = *reinterpret_cast<const f32x4*>(ps + 8 * s);
        av[s][1] = *reinterpret_cast<const f32x4*>(ps + 32 * LDK + 8 * s);
        bv[s][0] = *reinterpret_cast<const f32x4*>(qs + 8 * s);
        bv[s][1] = *reinterpret_cast<const f32x4*>(qs + 32 * LDK + 8 * s);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][0][e], bv[s][0][e], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][0][e], bv[s][1][e], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][1][e], bv[s][0][e], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][1][e], bv[s][1][e], acc[1][1], 0, 0, 0);
      }
    if (V >= 3) {
      float* dst = lds + (cur ^ 1) * 2 * 128 * LDK + (isq ? 128 * LDK : 0) + (4 * cg) * LDK + 4 * kg;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 c = {rg[0][j], rg[1][j], rg[2][j], rg[3][j]};
        *reinterpret_cast<f32x4*>(dst + j * LDK) = c;
      }
    }
    if (V >= 2) __syncthreads();
    if (V >= 1) cur ^= 1;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  out[blockIdx.x * 256 + tid] = s + rg[0][0] + rg[1][3];
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int V, int LAY = 0>
void run(const char* name, const float* P, const float* Q, int ld, int wgs, int stages, float* out,
         unsigned long long* cyc) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  if (LAY) bench_t<V><<<wgs, 256>>>(P, Q, ld, 4, out, cyc); else bench<V, 0><<<wgs, 256>>>(P, Q, ld, 4, out, cyc);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  if (LAY) bench_t<V><<<wgs, 256>>>(P, Q, ld, stages, out, cyc); else bench<V, 0><<<wgs, 256>>>(P, Q, ld, stages, out, cyc);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(wgs * 4);
  (void)hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * wgs * 4, hipMemcpyDeviceToHost);
  double sum = 0; for (auto v : h) sum += v;
  const double nm = (double)stages * 32;                         // MFMAs per wave
  const double per_simd = (double)wgs / 256.0;                   // waves per SIMD
  printf("%-62s WGs %4d: %.1f cyc/MFMA per SIMD  %.3f ms  %.1f TFLOP/s\n", name, wgs,
         sum / h.size() / nm / per_simd, ms, (double)wgs * 4 * nm * 4096 / ms / 1e9);
}

int main() {
  const int ld = 512, stages = 160;
  const int maxw = 1024;
  const size_t rows = (size_t)maxw * stages * BKR;
  float *P, *Q, *out; unsigned long long* cyc;
  if (hipMalloc(&P, rows * ld * 4) != hipSuccess || hipMalloc(&Q, rows * ld * 4) != hipSuccess) return 1;
  (void)hipMalloc(&out, maxw * 256 * 4); (void)hipMalloc(&cyc, maxw * 4 * 8);
  (void)hipMemset(P, 0, rows * ld * 4); (void)hipMemset(Q, 0, rows * ld * 4);
  for (int wgs : {256, 512, 768}) {
    run<0>("0: MFMAs only", P, Q, ld, wgs, stages, out, cyc);
    run<1>("1: + 32 operand reads from LDS per stage", P, Q, ld, wgs, stages, out, cyc);
    run<2>("2: + barrier per stage", P, Q, ld, wgs, stages, out, cyc);
    run<3>("3: + 4 x 16-byte LDS stores per stage", P, Q, ld, wgs, stages, out, cyc);
    run<4>("4: + 4 x 16-byte global loads per stage", P, Q, ld, wgs, stages, out, cyc);
    run<1, 1>("T1: + 8 ds_read_b128 operand reads (transposed tiles)", P, Q, ld, wgs, stages, out, cyc);
    run<2, 1>("T2: + barrier per stage", P, Q, ld, wgs, stages, out, cyc);
    run<3, 1>("T3: + 4 x 16-byte LDS stores per stage (transposing)", P, Q, ld, wgs, stages, out, cyc);
    run<4, 1>("T4: + 4 x 16-byte global loads per stage", P, Q, ld, wgs, stages, out, cyc);
  }
  return 0;
}
