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

// (WSection / buf_load16: common.h)  A lane's 16 bytes of the 1 KiB weight fragment number `frag_index` of a section
__device__ __forceinline__ f32x4 lane_frag(const WSection& ws, unsigned frag_index) {
  return buf_load16(ws, (threadIdx.x & 63u) * 16u, frag_index * 1024u);
}

// wp (below): the section base of the image's wave, wave-uniform (kernel argument + readfirstlane'd wave number).
// A wave's tile t of k-chunk kc is fragment kc * NTI + toff + t * TS of that section: NTI = tiles per chunk in the IMAGE.
// A kernel with as many waves as the image was packed for walks its own section (NTI = NT, toff = 0, TS = 1); a kernel
// with twice the waves shares a section between two waves (the 8-wave sampling instance on the 4-wave image:
// layers 1 / 2 toff = 4 (wave & 1), the heads -- tiles [S | T | Q] x 2 -- toff = wave & 1, TS = 2).
template <int NT, int NTI = NT, int TS = 1>
__device__ __forceinline__ void load_frags(f32x4 (&b)[NT], const WSection& ws, int kc, int toff = 0) {
#pragma unroll
  for (int t = 0; t < NT; ++t) b[t] = lane_frag(ws, (unsigned)(kc * NTI + toff + t * TS));
}

// Ring of B fragments, DEPTH slots of NT fragments; chunk k of the walk lives in slot k % DEPTH and every slot is
// re-loaded IN PLACE (same registers: the allocator never has to rotate the ring), but not behind its block as in
// rounds 1-3: left to itself the compiler sank those loads into bursts of 2 * NT behind the FOLLOWING block and
// drained the queue in front of it (ISA of rounds 1-3) -- the wave issued 16 loads back to back while its matrix pipe
// idled, 38 cycles per v_mfma_f32_16x16x4_f32.  Now a block runs in two halves over its tile groups [0, NT / 2) and
// [NT / 2, NT), the order fixed by sched_barrier, with ONE 1 KiB load after every fourth MFMA:
//   first half  (MFMAs on group 0): the previous block's group-1 registers take their next chunk,
//   second half (MFMAs on group 1): this block's own group-0 registers take theirs.
// A load issues while the pipe works on the four MFMAs in front of it: 34-35 cycles per MFMA in isolation
// (tools/mfma_pinned_bench.hip, profiles/r04_mfma_pinned_bench.txt; an LDS ring fed by LDS-DMA loader waves reads 41
// cycles on the consumer side alone and 57-73 with its hand-shake: not built).  Every fragment has DEPTH - 1 blocks
// (32 x NT cycles each) to come back from L2.  The ring of the NEXT layer is primed before the current layer's
// epilogue and barrier, which hides the pipeline fill.
template <int NT, int DEPTH = 3>
struct BRing {
  f32x4 b[DEPTH][NT];
};

template <int NT, int T0, int T1, int NTI = NT, int TS = 1>
__device__ __forceinline__ void load_group(f32x4 (&b)[NT], const WSection& ws, int kc, int toff = 0) {
#pragma unroll
  for (int t = T0; t < T1; ++t) b[t] = lane_frag(ws, (unsigned)(kc * NTI + toff + t * TS));
}

// rev: the layer's k-chunks are walked from the last to the first (NKC chunks in all).  A network's image is
// streamed in alternating directions on its consecutive calls, so the part of it that the previous call left in L2
// -- its most recently touched end -- is what the next call asks for first.
// Primes chunks 0 .. DEPTH - 1 of the walk except the last slot's second tile group, which the first block requests.
template <int NT, int DEPTH, int NTI = NT, int TS = 1>
__device__ __forceinline__ void ring_prime(BRing<NT, DEPTH>& R, const float* __restrict__ wp, bool rev = false,
                                           int nkc = 0, int toff = 0) {
  const WSection ws = wsection(wp);
#pragma unroll
  for (int s = 0; s < DEPTH - 1; ++s) load_frags<NT, NTI, TS>(R.b[s], ws, rev ? nkc - 1 - s : s, toff);
  load_group<NT, 0, (NT + 1) / 2, NTI, TS>(R.b[DEPTH - 1], ws, rev ? nkc - DEPTH : DEPTH - 1, toff);
}

// Half a block: acc[t] += cur[t] (x) a over the chunk's four k-steps for the tiles [T0, T1), e-major (every
// accumulator sees its k in ascending order: the bits do not depend on the interleave); after every fourth MFMA one
// of the fragments [L0, L1) of chunk `kc` goes into dst (LOAD); an odd tile count leaves one more load than the
// shorter half has slots: it goes behind the last MFMA.
// G groups of MPL MFMAs, one load behind each of the first NLD groups (all that are left behind the last one)
template <int G, int NLD, int MPL = 4, int g = 0>
__device__ __forceinline__ void sched_mfma_load_pipeline() {
  if constexpr (g < G) {
    __builtin_amdgcn_sched_group_barrier(0x008, MPL, 0);                                   // MPL MFMA
    if constexpr (g < NLD) __builtin_amdgcn_sched_group_barrier(0x020, g == G - 1 ? NLD - g : 1, 0);   // VMEM read(s)
    sched_mfma_load_pipeline<G, NLD, MPL, g + 1>();
  }
}

template <int NT, int T0, int T1, int L0, int L1, bool LOAD, int NDS = 0, int NTI = NT, int TS = 1>
__device__ __forceinline__ void mfma_half_stream(const f32x4 a, const f32x4 (&cur)[NT], f32x4 (&dst)[NT],
                                                 f32x4 (&acc)[NT], const WSection& wp, int kc, int toff = 0) {
  constexpr int G = T1 - T0, NLD = L1 - L0;
#pragma unroll
  for (int i = 0; i < 4 * G; ++i) {
    const int e = i / G, t = T0 + i % G;
    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[t][e], a[e], acc[t], 0, 0, 0);
  }
  if constexpr (LOAD) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) dst[L0 + j] = lane_frag(wp, (unsigned)(kc * NTI + toff + (L0 + j) * TS));
  }
  // the order the scheduler must emit: the NDS LDS reads of the NEXT block's activation fragment first (left to float
  // they sink to the block's end and the next block waits out the whole LDS latency), then four MFMAs, one load, ...
  if constexpr (NDS > 0) __builtin_amdgcn_sched_group_barrier(0x100, NDS, 0);
  sched_mfma_load_pipeline<G, LOAD ? NLD : 0>();
}

// wp: this wave's section base (wave-uniform).  afrag(kc) returns the lane's A fragment of chunk kc
// (the fragment of the next chunk is fetched from LDS while the current block's MFMAs issue).
template <int NT, int NKC, int DEPTH, int NTI = NT, int TS = 1, typename AF>
__device__ __forceinline__ void stream_layer(BRing<NT, DEPTH>& R, const float* __restrict__ wbase, AF afrag,
                                             f32x4 (&acc)[NT], bool rev = false, int toff = 0) {
  const WSection wp = wsection(wbase);
  static_assert(DEPTH >= 2 && NKC >= DEPTH && NT >= 2, "ring depth / tile groups");
  constexpr int G = (NT + 1) / 2;                              // tiles [0, G) and [G, NT)
  auto km = [&](int k) { return rev ? NKC - 1 - k : k; };      // position in the walk -> chunk
  // block k requests group 1 of chunk k + DEPTH - 1 (first half) and group 0 of chunk k + DEPTH (second half)
  constexpr int MAIN = (NKC - DEPTH) / DEPTH * DEPTH;          // rolled loop: both requests in range
  // Nothing crosses this point: the pipeline below takes ANY vector-memory read of its scheduling region for its load
  // slots -- without the fence it pulls the ring_prime loads a caller issued just above into them (first-layer
  // halves), every MFMA group then waits for the load in front of it (vmcnt(0) per group: measured, 1.9 x slower).
  __builtin_amdgcn_sched_barrier(0);
  f32x4 a0 = afrag(km(0));
  int kc = 0;
#pragma nounroll
  for (; kc < MAIN; kc += DEPTH) {
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) {
      const f32x4 a1 = afrag(km(kc + s + 1));
      mfma_half_stream<NT, 0, G, G, NT, true, 1, NTI, TS>(a0, R.b[s], R.b[(s + DEPTH - 1) % DEPTH], acc, wp,
                                                          km(kc + s + DEPTH - 1), toff);
      mfma_half_stream<NT, G, NT, 0, G, true, 0, NTI, TS>(a0, R.b[s], R.b[s], acc, wp, km(kc + s + DEPTH), toff);
      a0 = a1;
    }
  }
#pragma unroll
  for (int k = MAIN; k < NKC; ++k) {                           // the walk's end: compile-time block numbers
    const int s = (k - MAIN) % DEPTH;
    const f32x4 a1 = afrag(km(k + 1 < NKC ? k + 1 : NKC - 1));
    if (k + DEPTH - 1 < NKC)
      mfma_half_stream<NT, 0, G, G, NT, true, 1, NTI, TS>(a0, R.b[s], R.b[(s + DEPTH - 1) % DEPTH], acc, wp,
                                                          km(k + DEPTH - 1), toff);
    else
      mfma_half_stream<NT, 0, G, G, NT, false, 1>(a0, R.b[s], R.b[(s + DEPTH - 1) % DEPTH], acc, wp, 0);
    if (k + DEPTH < NKC)
      mfma_half_stream<NT, G, NT, 0, G, true, 0, NTI, TS>(a0, R.b[s], R.b[s], acc, wp, km(k + DEPTH), toff);
    else
      mfma_half_stream<NT, G, NT, 0, G, false>(a0, R.b[s], R.b[s], acc, wp, 0);
    a0 = a1;
  }
  __builtin_amdgcn_sched_barrier(0);                           // ... and the next layer's priming loads stay behind it
}

}  // namespace l2hmc
