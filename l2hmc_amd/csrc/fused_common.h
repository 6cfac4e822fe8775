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

// (fast_exp / fast_tanh: common.h)

// Tape traffic of the training kernels (1.2 GB written per forward pass, read once by the reverse pass and the
// weight-gradient products, always from HBM): non-temporal, so that it does not push the weight images -- which
// every workgroup re-reads from L2 at every network call -- out of the 4 MB L2 of its XCD.
#ifdef L2HMC_TAPE_TEMPORAL      // diagnostic build only: default cache policy, to price the non-temporal one
__device__ __forceinline__ void tape_store(float* dst, const f32x4& v) { *reinterpret_cast<f32x4*>(dst) = v; }
__device__ __forceinline__ f32x4 tape_load(const float* src) { return *reinterpret_cast<const f32x4*>(src); }
#else
__device__ __forceinline__ void tape_store(float* dst, const f32x4& v) {
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst));
}
__device__ __forceinline__ f32x4 tape_load(const float* src) {
  return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src));
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

// DEPTH-deep ring of B fragments: chunk kc + DEPTH is requested as soon as chunk kc has been consumed, so every
// load has DEPTH - 1 full MFMA blocks (32 x NT cycles each) to come back from L2 -- measured under this kernel's
// load an L2 hit takes ~1,300 cycles, so 8-tile layers want DEPTH >= 3 and the 6-tile heads DEPTH >= 4.  The ring
// of the NEXT layer is primed before the current layer's epilogue and barrier, which hides the pipeline fill.
template <int NT, int DEPTH = 3>
struct BRing {
  f32x4 b[DEPTH][NT];
};

// rev: the layer's k-chunks are walked from the last to the first (NKC chunks in all).  A network's image is
// streamed in alternating directions on its consecutive calls, so the part of it that the previous call left in L2
// -- its most recently touched end -- is what the next call asks for first.
template <int NT, int DEPTH>
__device__ __forceinline__ void ring_prime(BRing<NT, DEPTH>& R, const float* __restrict__ wp, bool rev = false,
                                           int nkc = 0) {
#pragma unroll
  for (int s = 0; s < DEPTH; ++s) load_frags<NT>(R.b[s], wp, rev ? nkc - 1 - s : s);
}

// wp: this wave's section base + lane * 4.  afrag(kc) returns the lane's A fragment of chunk kc
// (the fragment of the next chunk is fetched from LDS while the current block's MFMAs issue).
template <int NT, int NKC, int DEPTH, typename AF>
__device__ __forceinline__ void stream_layer(BRing<NT, DEPTH>& R, const float* __restrict__ wp, AF afrag,
                                             f32x4 (&acc)[NT], bool rev = false) {
  static_assert(NKC >= DEPTH, "ring depth");
  auto km = [&](int k) { return rev ? NKC - 1 - k : k; };      // position in the walk -> chunk
  f32x4 a0 = afrag(km(0));
  int kc = 0;
#pragma nounroll
  for (; kc + DEPTH <= NKC; kc += DEPTH) {
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) {
      const f32x4 a1 = afrag(km(kc + s + 1 < NKC ? kc + s + 1 : NKC - 1));
      mfma_block<NT>(a0, R.b[s], acc);
      if (kc + s + DEPTH < NKC) load_frags<NT>(R.b[s], wp, km(kc + s + DEPTH));
      a0 = a1;
    }
  }
  constexpr int REM = NKC % DEPTH;        // their fragments were requested by the last full round
#pragma unroll
  for (int s = 0; s < REM; ++s) {
    const f32x4 a1 = afrag(km(NKC - REM + s + 1 < NKC ? NKC - REM + s + 1 : NKC - 1));
    mfma_block<NT>(a0, R.b[s], acc);
    a0 = a1;
  }
}

}  // namespace l2hmc
