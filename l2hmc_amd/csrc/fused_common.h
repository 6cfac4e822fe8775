// Pieces shared by the whole-trajectory kernels (fused_traj.hip: sampling / taped forward,
// fused_train.hip: reverse pass): the 16-row workgroup geometry, the streaming MFMA core fed by
// fragment-ordered weights, and the hardware exp / tanh forms.
#pragma once
#include "stq_dense.h"
#include <math.h>

namespace l2hmc {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kFM = 16;                   // rows per workgroup
#ifndef L2HMC_FUSED_WAVES
#define L2HMC_FUSED_WAVES 4
#endif
constexpr int kFWaves = L2HMC_FUSED_WAVES;   // waves per workgroup (4 = one per SIMD, 8 = two per SIMD)
constexpr int kFThreads = 64 * kFWaves;   // wave w owns output columns [w*N/kFWaves, (w+1)*N/kFWaves)
constexpr int kTPC = kFThreads / kFM;     // threads per chain in the chain-local passes

// exp / tanh on the hardware exp2 + rcp units (v_exp_f32, v_rcp_f32: ~1 ulp each).  Arguments here are
// eps * S, eps * Q and pre-activations of O(1): |error| <= ~2e-7 relative for exp, ~1.5e-7 absolute for
// tanh -- at the fp32 rounding floor of the quantities they feed, and ~8x cheaper than the libm forms.
#ifdef L2HMC_EXACT_MATH   // diagnostic build only (tools/build_exact.sh): libm forms, to price the hardware forms' error
__device__ __forceinline__ float fast_exp(float x) { return expf(x); }
__device__ __forceinline__ float fast_tanh(float x) { return tanhf(x); }
#else
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float fast_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(-2.88539008177792681f * fabsf(x));      // exp(-2|x|) in (0, 1]
  const float t = (1.f - e) * __builtin_amdgcn_rcpf(1.f + e);
  return copysignf(t, x);
}
#endif

// ---------------------------------------------------------------------------
// streaming GEMM core: acc[t] += A(16 x 16*NKC) . Wpacked, one wave, NT tiles
// ---------------------------------------------------------------------------
// The WEIGHT fragment is the MFMA's first operand and the activation fragment its second: the product is formed
// transposed, C^T[col][row], so lane (q = lane / 16, r = lane % 16) ends up with out[row r][tile * 16 + 4 q + e],
// e = 0..3 -- four CONSECUTIVE COLUMNS of ONE row.  Every epilogue access (bias, masks, x / v / force, the h1 / h2
// rows the next layer reads as its k-contiguous fragments) is then one 16-byte LDS instruction instead of four
// 4-byte ones, and a row's log-det needs two cross-lane steps instead of four.  The packed weight image and the
// activation fragment reads are the same for either operand order (A[i][k] and B[k][j] use the same lane map).
template <int NT>
__device__ __forceinline__ void mfma_block(const f32x4 a, const f32x4 (&b)[NT], f32x4 (&acc)[NT]) {
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[t][e], a[e], acc[t], 0, 0, 0);
}

template <int NT>
__device__ __forceinline__ void load_frags(f32x4 (&b)[NT], const float* __restrict__ wp, int kc) {
#pragma unroll
  for (int t = 0; t < NT; ++t) b[t] = *reinterpret_cast<const f32x4*>(wp + ((size_t)kc * NT + t) * 256);
}

// Three-deep ring of B fragments: chunk kc+3 is requested as soon as chunk kc has been consumed, so every
// load has two full MFMA blocks (~2 x 32 x NT cycles) to come back from L2.  The ring of the NEXT layer is
// primed before the current layer's epilogue and barrier, which hides the pipeline fill.
template <int NT>
struct BRing {
  f32x4 b[3][NT];
};

template <int NT>
__device__ __forceinline__ void ring_prime(BRing<NT>& R, const float* __restrict__ wp) {
  load_frags<NT>(R.b[0], wp, 0);
  load_frags<NT>(R.b[1], wp, 1);
  load_frags<NT>(R.b[2], wp, 2);
}

// wp: this wave's section base + lane * 4.  afrag(kc) returns the lane's A fragment of chunk kc
// (the fragment of the next chunk is fetched from LDS while the current block's MFMAs issue).
template <int NT, int NKC, typename AF>
__device__ __forceinline__ void stream_layer(BRing<NT>& R, const float* __restrict__ wp, AF afrag,
                                             f32x4 (&acc)[NT]) {
  static_assert(NKC >= 3, "ring depth");
  f32x4 a0 = afrag(0), a1;
  int kc = 0;
#pragma nounroll
  for (; kc + 3 <= NKC; kc += 3) {
    a1 = afrag(kc + 1 < NKC ? kc + 1 : NKC - 1);
    mfma_block<NT>(a0, R.b[0], acc);
    if (kc + 3 < NKC) load_frags<NT>(R.b[0], wp, kc + 3);
    a0 = afrag(kc + 2 < NKC ? kc + 2 : NKC - 1);
    mfma_block<NT>(a1, R.b[1], acc);
    if (kc + 4 < NKC) load_frags<NT>(R.b[1], wp, kc + 4);
    a1 = afrag(kc + 3 < NKC ? kc + 3 : NKC - 1);
    mfma_block<NT>(a0, R.b[2], acc);
    if (kc + 5 < NKC) load_frags<NT>(R.b[2], wp, kc + 5);
    a0 = a1;
  }
  if constexpr (NKC % 3 >= 1) {
    if constexpr (NKC % 3 == 2) a1 = afrag(NKC - 1);
    mfma_block<NT>(a0, R.b[0], acc);
  }
  if constexpr (NKC % 3 == 2) mfma_block<NT>(a1, R.b[1], acc);
}

}  // namespace l2hmc
