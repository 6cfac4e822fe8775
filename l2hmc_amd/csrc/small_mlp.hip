// Generic L2HMC integrator on low-dimensional toy targets (MoG / strongly
// correlated Gaussian), one thread per chain, whole trajectory in one launch.
//
// Replaces the tf.while_loop graph of
//   l2hmc/utils/dynamics.py:120-225 (_forward_step/_backward_step), :255-319
//   l2hmc/utils/network.py:89-114   (`network` MLP, ScaleTanh on S and F)
//   l2hmc/utils/distributions.py:32-39,63-68,151-158 (energies; gradient by
//   autodiff there, closed form here)
// These configurations (x_dim 2, 10-50 hidden units) are launch/latency bound.
// Both networks' weights stay in LDS and the chain state in registers; x and v
// are read once and written once per trajectory.  A wave integrates 16 chains:
// the thin first layer runs on the VALU, the hidden layer and the heads on
// 16x16x4 fp32 MFMAs (see small_traj_mfma_kernel below).  The training kernel
// (small_train.hip) keeps the earlier sixteen-lanes-per-chain VALU form of
// small_mlp.h.
#include "small_mlp.h"
#include <atomic>

namespace l2hmc {

__global__ __launch_bounds__(kSmallThreads) void mog_energy_grad_kernel(l2hmc_mog_target t,
                                                                        const float* __restrict__ x,
                                                                        int64_t rows,
                                                                        float* __restrict__ energy,
                                                                        float* __restrict__ grad) {
  extern __shared__ float lds[];
  load_target(t, lds);
  __syncthreads();
  const int64_t r = (int64_t)blockIdx.x * kSmallThreads + threadIdx.x;
  if (r >= rows) return;
  float xv[kMaxDim], g[kMaxDim], E;
#pragma unroll
  for (int d = 0; d < kMaxDim; ++d) xv[d] = d < t.dim ? x[r * t.dim + d] : 0.f;
  energy_grad(lds, t.dim, t.K, t.is_gaussian, 1.f / t.temperature, xv, &E, g);
  if (energy) energy[r] = E;
  if (grad) {
#pragma unroll
    for (int d = 0; d < kMaxDim; ++d)
      if (d < t.dim) grad[r * t.dim + d] = g[d];
  }
}

struct SmallTrajArgs {
  l2hmc_small_plan plan;
  const float* x0; const float* v0; const int* dir; int64_t rows;
  float* x_out; float* v_out; float* sumlogdet; float* p_accept;
  // propose mode (l2hmc_small_propose; prop_B > 0): the kernel owns BOTH directions of its chains -- lane r of a
  // wave is chain r & 7 of the wave's eight, direction r >> 3 --, draws its own Philox streams and finishes
  // utils/sampler.py:28-59 (mix by the direction bit, Metropolis-Hastings) in its epilogue
  int64_t prop_B; uint64_t seed, draw0;
  float* Lx; float* Lv; float* px; float* mh_out;
  unsigned long long* stamps;            // diagnostic builds only (-DL2HMC_STAMPS, class 7), else NULL
};

// Per-phase cycle counters of a wave (diagnostic build only): 0 first layer (VALU), 1 hidden layer (MFMA), 2 heads
// (MFMA + bias + store to the wave's patch), 3 patch round trip + tanh / exp(coeff), 4 target energy / gradient,
// 5 sub-update arithmetic (exp, masks, log-det), 6 total
#ifdef L2HMC_STAMPS
extern unsigned long long* g_stamp_buf;      // stq_dense.hip (l2hmc_debug_set_stamps)
extern int g_stamp_cls;
#define ST_NOW()                                                                              \
  ({                                                                                          \
    unsigned long long t_;                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    t_;                                                                                       \
  })
#define ST_ADD(slot, t0) st[slot] += ST_NOW() - (t0)
#else
#define ST_NOW() 0ull
#define ST_ADD(slot, t0) do {} while (0)
#endif

// element i of the stream l2hmc_fill_uniform / l2hmc_fill_normal writes for (seed, offset)   (capi.hip: fill_kernel)
__device__ __forceinline__ void philox_block_at(uint64_t seed, uint64_t offset, int64_t i, uint32_t c[4]) {
  const uint64_t b = (uint64_t)i >> 2;
  c[0] = (uint32_t)b; c[1] = (uint32_t)(b >> 32); c[2] = (uint32_t)offset; c[3] = (uint32_t)(offset >> 32);
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}
__device__ __forceinline__ float philox_uniform_at(uint64_t seed, uint64_t offset, int64_t i) {
  uint32_t c[4];
  philox_block_at(seed, offset, i, c);
  return (float)(c[i & 3] >> 8) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ float philox_normal_at(uint64_t seed, uint64_t offset, int64_t i) {
  uint32_t c[4];
  philox_block_at(seed, offset, i, c);
  float v[4];
  philox_normal4(c, v);
  return v[i & 3];
}

// =====================================================================================================
// The trajectory kernel.  One WAVE integrates 16 chains and never exchanges anything
// with another wave: lane (q = lane / 16, r = lane % 16) belongs to chain r of the wave; the four lanes of a
// chain hold the same chain state.  Matrix instructions take the WEIGHTS as first operand (16 output units x 4 k)
// and the 16 chains as second, so a product comes out as out[chain r][row 4 q + e of the tile] in register e.
//   layer 1  input vector [a | b | cos t, sin t, 1, 0] (2 MD + 4 entries; the bias rides on the constant 1); a lane
//            produces exactly the hidden units it feeds layer 2 as ITS slice of the k-steps.  Two forms, the same
//            ascending-k fma chain per unit:
//              L1M  on the matrix pipe: lane q supplies entry 4 s1 + q of k-step s1 (2 x NT instructions); unit
//                   16 t + 4 q + e arrives in register e of tile t and is k-step 4 t + e of layer 2 (KSH steps).  For
//                   batches of at most one wave per SIMD, where a wave's 4 N network calls are one chain of
//                   dependent steps (cfg 2: 512 waves on 1024 SIMDs);
//              VALU unit 4 s + q at step s (KS steps), 8 multiply-adds per unit from an LDS record: for chip-filling
//                   batches, where the matrix pipe is the bound (4 x NT fewer instructions per call) and another
//                   wave's MFMAs cover this wave's VALU work.
//            The two forms walk layer 2's k in different orders, so they agree to rounding, not bit for bit: a batch
//            size picks one, and every entry point (propose, trajectory) picks the same for the same rows.
//            (Round 2 had only a VALU form: 19 % of a wave's cycles at cfg 2, profiles/r03_small_traj_stamps.txt.)
//   layer 2  k-steps x NT tiles; the product leaves unit 16 t + 4 q + e in register e of tile t -- the lane's slice
//            for the heads (step s = 4 t + e), again without any exchange;
//   heads    KSH steps (only steps whose smallest unit 16 t + e exists), outputs packed [S | T | Q] by MD; the chain's
//            values reach its four lanes by ds_bpermute_b32 (LDS crossbar: no memory, no barrier) instead of round 2's
//            store / wave barrier / load round trip through an LDS patch.
// Both networks' fragments, biases and coefficients live in registers (one wave per SIMD: 512 per lane).
// =====================================================================================================
using f32x4s = __attribute__((ext_vector_type(4))) float;

// HP: hidden width padded to the 16-wide output tiles.  KS: k-steps of the hidden layer (unit 4 s + q:
// ceil(num_nodes / 4)).  KSH: k-steps of the heads (unit 16 t + 4 q + e at step 4 t + e; only steps whose smallest
// unit 16 t + e exists).  num_nodes 50: 13 and 14 steps instead of 16 and 16.
template <int HP, int MD, int KS_, int KSH_>
struct MfmaNet {
  static constexpr int NT = HP / 16, KS = KS_, KSH = KSH_, NTH = (3 * MD + 15) / 16, K1 = 2 * MD + 4, KS1 = K1 / 4;
  static constexpr int rec = 0;                                 // [HP][K1]: unit u's weights for [a | b | cos, sin, 1, 0]
  static constexpr int w1 = rec + HP * K1;                      // [NT][KS1][64]   (L1M; permuted output rows)
  static constexpr int w2 = w1 + NT * KS1 * 64;                 // VALU form [NT][KS][64] (k = 4 s + q); L1M [NT][KSH][64]
  static constexpr int whd = w2 + NT * (KSH > KS ? KSH : KS) * 64;   // [NTH][KSH][64]
  static constexpr int bh = whd + NTH * KSH * 64;               // [HP]
  static constexpr int bhd = bh + HP;                           // [NTH * 16]  (outputs packed [S | T | Q] by MD)
  static constexpr int es = bhd + NTH * 16;                     // [MD] (padded to 8)
  static constexpr int eq = es + 8;
  static constexpr int size = eq + 8;
};

// weight of hidden unit `u` for entry k of the first layer's input vector [a | b | cos t, sin t, 1, 0]
template <int MD>
__device__ __forceinline__ float l1_weight(const l2hmc_dense_net& n, int dim, int u, int k) {
  if (u >= n.H) return 0.f;
  if (k < MD) return k < dim ? n.w1_t[(size_t)u * 2 * dim + k] : 0.f;
  if (k < 2 * MD) return k - MD < dim ? n.w1_t[(size_t)u * 2 * dim + dim + (k - MD)] : 0.f;
  if (k == 2 * MD) return n.wt[u];
  if (k == 2 * MD + 1) return n.wt[n.H + u];
  if (k == 2 * MD + 2) return n.b1[u];
  return 0.f;
}

// (the image is built by every workgroup in its prologue, which is part of a latency-bound launch: only the
//  first-layer form that will run is filled)
template <int HP, int MD, int KS_, int KSH_, bool L1M>
__device__ void load_net_mfma(const l2hmc_dense_net& n, float* L, int dim) {
  using V = MfmaNet<HP, MD, KS_, KSH_>;
  const int H = n.H, tid = threadIdx.x;
  if constexpr (!L1M) {
    for (int i = tid; i < HP * V::K1; i += kSmallThreads) L[V::rec + i] = l1_weight<MD>(n, dim, i / V::K1, i % V::K1);
  } else {
    for (int i = tid; i < V::NT * V::KS1 * 64; i += kSmallThreads) {
      const int lane = i & 63, s = (i >> 6) % V::KS1, to = (i >> 6) / V::KS1;
      L[V::w1 + i] = l1_weight<MD>(n, dim, 16 * to + (lane & 15), 4 * s + (lane >> 4));
    }
  }
  constexpr int K2 = L1M ? V::KSH : V::KS;         // k-steps of the hidden layer in this form
  for (int i = tid; i < V::NT * K2 * 64; i += kSmallThreads) {
    const int lane = i & 63, s = (i >> 6) % K2, to = (i >> 6) / K2;
    const int out = 16 * to + (lane & 15);
    const int k = L1M ? 16 * (s >> 2) + 4 * (lane >> 4) + (s & 3) : 4 * s + (lane >> 4);
    L[V::w2 + i] = (out < H && k < H) ? n.wh_t[(size_t)out * H + k] : 0.f;
  }
  for (int i = tid; i < V::NTH * V::KSH * 64; i += kSmallThreads) {
    const int lane = i & 63, s = (i >> 6) % V::KSH, th = (i >> 6) / V::KSH;
    const int o = 16 * th + (lane & 15), hd = o / MD, d = o % MD;        // output o = head * MD + component
    const int k = 16 * (s >> 2) + 4 * (lane >> 4) + (s & 3);
    L[V::whd + i] = (hd < 3 && d < dim && k < H) ? n.whd_t[((size_t)hd * dim + d) * H + k] : 0.f;
  }
  for (int i = tid; i < HP; i += kSmallThreads) L[V::bh + i] = i < H ? n.bh[i] : 0.f;
  for (int i = tid; i < V::NTH * 16; i += kSmallThreads) {
    const int hd = i / MD, d = i % MD;
    L[V::bhd + i] = (hd < 3 && d < dim) ? n.bhd[hd * dim + d] : 0.f;
  }
  for (int i = tid; i < 8; i += kSmallThreads) {
    L[V::es + i] = i < dim ? expf(n.coeff_s[i]) : 0.f;
    L[V::eq + i] = i < dim ? expf(n.coeff_q[i]) : 0.f;
  }
}

// The lane's weight fragments of one network in registers (num_nodes 50, x_dim 2: 8 + 52 + 14; both networks: 148 of
// the lone wave's 512): the matrix instructions take their weights straight from registers.  Biases and coefficients:
// in registers in the latency form (L1M: one wave per SIMD by design; 101 -> 92 us per cfg-2 propose), in LDS in the
// throughput form (in registers they pushed it past 256, i.e. from two waves per SIMD to one: 0.57 -> 0.81 ms at
// 65536 chains)
// TW ("twin", round 4; latency form only, HP = 64): TWO waves integrate the same 16 rows.  Both form the whole first
// layer; wave `part` owns the hidden layer's output tiles 2 part, 2 part + 1 -- i.e. the hidden units that are k-steps
// [8 part, 8 part + 8) of the heads -- and the heads' partial sums over those steps; the two partial sums meet through
// LDS once per network call (net_eval_mfma).  Per call a wave issues 2 NT + 2 K2 + 8 NTH matrix instructions instead of
// 2 NT + 4 K2 + KSH NTH (cfg 2: 44 instead of 78), and at cfg 2 the 1024 waves have a SIMD each (512 waves before).
template <int HP, int MD, int KS_, int KSH_>
struct NetRegsTwin {
  using V = MfmaNet<HP, MD, KS_, KSH_>;
  static_assert(V::NT == 4, "the twin form splits four hidden tiles over two waves");
  static constexpr int K2 = V::KSH, KO = 8;                  // hidden-layer k-steps; heads k-steps owned by a wave
  float w1[V::NT * V::KS1];
  float w2[2 * K2];                     // own tiles: [tile-local][s]
  float whd[V::NTH * KO];               // own heads steps: global step 8 part + j (zero beyond KSH)
  float bh[KO];                         // bias of own unit at own step j
  float bhd[V::NTH * 4];
  float es[MD], eq[MD];
  __device__ __forceinline__ void load(const float* L, int lane, int part) {
    const int q = lane >> 4;
#pragma unroll
    for (int j = 0; j < KO; ++j) {
      const int s = 8 * part + j;
      bh[j] = s < V::KSH ? L[V::bh + 16 * (s >> 2) + 4 * q + (s & 3)] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < V::NTH; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) bhd[4 * t + e] = L[V::bhd + 16 * t + 4 * q + e];
#pragma unroll
    for (int d = 0; d < MD; ++d) {
      es[d] = L[V::es + d];
      eq[d] = L[V::eq + d];
    }
#pragma unroll
    for (int i = 0; i < V::NT * V::KS1; ++i) w1[i] = L[V::w1 + i * 64 + lane];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
      for (int s = 0; s < K2; ++s) w2[tl * K2 + s] = L[V::w2 + ((2 * part + tl) * K2 + s) * 64 + lane];
#pragma unroll
    for (int th = 0; th < V::NTH; ++th)
#pragma unroll
      for (int j = 0; j < KO; ++j) {
        const int s = 8 * part + j;
        whd[th * KO + j] = s < V::KSH ? L[V::whd + (th * V::KSH + s) * 64 + lane] : 0.f;
      }
  }
};

// (S, T, Q) of the twin form: as net_eval_mfma's L1M path, with the hidden layer and the heads split over the two waves
// of a group.  xch: this group's exchange patch [2 call parities][2 parts][NTH][64 lanes] of 16 bytes; every wave of the
// workgroup calls this the same number of times (one workgroup barrier per call).
template <int HP, int MD, int KS_, int KSH_>
__device__ __forceinline__ void net_eval_twin(const NetRegsTwin<HP, MD, KS_, KSH_>& W, int dim, int q_tanh,
                                              const float (&a)[MD], const float (&b)[MD], float tc, float ts, int lane,
                                              int part, f32x4s* xch, int parity, float (&S)[MD], float (&T)[MD],
                                              float (&Q)[MD]) {
  using V = MfmaNet<HP, MD, KS_, KSH_>;
  constexpr int NT = V::NT, NTH = V::NTH, KS1 = V::KS1, K2 = V::KSH, KO = 8;
  const int q = lane >> 4, r = lane & 15;
  // ---- layer 1, whole (both waves need every hidden unit as k of layer 2).  The LAST k-step -- [.., cos t, sin t, 1, 0]:
  // time term and bias -- goes first: it depends on nothing the caller computes right before the call (the target's
  // gradient for the momentum network), so its matrix instructions can issue under that arithmetic
  f32x4s acc[NT];
#pragma unroll
  for (int to = 0; to < NT; ++to) acc[to] = f32x4s{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s0 = 0; s0 < KS1; ++s0) {
    const int s = (s0 + KS1 - 1) % KS1;           // KS1 - 1, 0, 1, ...
    float e4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = 4 * s + j;           // compile-time
      e4[j] = k < MD ? a[k < MD ? k : 0] : k < 2 * MD ? b[(k >= MD && k < 2 * MD) ? k - MD : 0]
              : k == 2 * MD ? tc : k == 2 * MD + 1 ? ts : k == 2 * MD + 2 ? 1.f : 0.f;
    }
    const float mine = q == 0 ? e4[0] : q == 1 ? e4[1] : q == 2 ? e4[2] : e4[3];
#pragma unroll
    for (int to = 0; to < NT; ++to)
      acc[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(W.w1[to * KS1 + s], mine, acc[to], 0, 0, 0);
  }
  float h1[K2];
#pragma unroll
  for (int s = 0; s < K2; ++s) h1[s] = fmaxf(acc[s >> 2][s & 3], 0.f);
  // ---- layer 2, own two tiles
  f32x4s a2[2] = {f32x4s{0.f, 0.f, 0.f, 0.f}, f32x4s{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int s = 0; s < K2; ++s)
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
      a2[tl] = __builtin_amdgcn_mfma_f32_16x16x4f32(W.w2[tl * K2 + s], h1[s], a2[tl], 0, 0, 0);
  float h2[KO];
#pragma unroll
  for (int j = 0; j < KO; ++j) h2[j] = fmaxf(a2[j >> 2][j & 3] + W.bh[j], 0.f);
  // ---- heads: partial sums over the own k-steps, exchanged through LDS, added part 0 first in BOTH waves
  f32x4s* mine = xch + (parity * 2 + part) * NTH * 64 + lane;
  const f32x4s* other = xch + (parity * 2 + (part ^ 1)) * NTH * 64 + lane;
  f32x4s hv[NTH];
#pragma unroll
  for (int th = 0; th < NTH; ++th) {
    f32x4s c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
#pragma unroll
    for (int j = 0; j < KO; ++j) {
      if (j & 1) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(W.whd[th * KO + j], h2[j], c1, 0, 0, 0);
      else c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(W.whd[th * KO + j], h2[j], c0, 0, 0, 0);
    }
    hv[th] = c0 + c1;
    mine[th * 64] = hv[th];
  }
  __syncthreads();
#pragma unroll
  for (int th = 0; th < NTH; ++th) {
    const f32x4s o = other[th * 64];
    const f32x4s bias = f32x4s{W.bhd[4 * th], W.bhd[4 * th + 1], W.bhd[4 * th + 2], W.bhd[4 * th + 3]};
    hv[th] = (part ? o + hv[th] : hv[th] + o) + bias;
  }
  float out[3 * MD];
#pragma unroll
  for (int o = 0; o < 3 * MD; ++o) {
    const int src = 16 * ((o & 15) >> 2) + r;
    out[o] = __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(hv[o >> 4][o & 3])));
  }
#pragma unroll
  for (int d = 0; d < MD; ++d) {
    if (d < dim) {
      S[d] = fast_tanh(out[d]) * W.es[d];
      T[d] = out[MD + d];
      Q[d] = (q_tanh ? fast_tanh(out[2 * MD + d]) : out[2 * MD + d]) * W.eq[d];
    }
  }
}

template <int HP, int MD, int KS_, int KSH_, bool L1M>
struct NetRegs {
  using V = MfmaNet<HP, MD, KS_, KSH_>;
  float w1[L1M ? V::NT * V::KS1 : 1];
  static constexpr int K2 = L1M ? V::KSH : V::KS;
  float w2[V::NT * K2];
  float whd[V::NTH * V::KSH];
  float bh[L1M ? V::KSH : 1];           // L1M: bias of unit 16 t + 4 q + e at [4 t + e]
  float bhd[L1M ? V::NTH * 4 : 1];      //      bias of output 16 th + 4 q + e
  float es[L1M ? MD : 1], eq[L1M ? MD : 1];
  __device__ __forceinline__ void load(const float* L, int lane) {
    if constexpr (L1M) {
      const int q = lane >> 4;
#pragma unroll
      for (int s = 0; s < V::KSH; ++s) bh[s] = L[V::bh + 16 * (s >> 2) + 4 * q + (s & 3)];
#pragma unroll
      for (int t = 0; t < V::NTH; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) bhd[4 * t + e] = L[V::bhd + 16 * t + 4 * q + e];
#pragma unroll
      for (int d = 0; d < MD; ++d) {
        es[d] = L[V::es + d];
        eq[d] = L[V::eq + d];
      }
#pragma unroll
      for (int i = 0; i < V::NT * V::KS1; ++i) w1[i] = L[V::w1 + i * 64 + lane];
    }
#pragma unroll
    for (int i = 0; i < V::NT * K2; ++i) w2[i] = L[V::w2 + i * 64 + lane];
#pragma unroll
    for (int i = 0; i < V::NTH * V::KSH; ++i) whd[i] = L[V::whd + i * 64 + lane];
  }
};

// (S, T, Q) = net([a, b, t]) for the 16 chains of this wave; every lane returns its own chain's values.
template <int HP, int MD, int KS_, int KSH_, bool L1M>
__device__ __forceinline__ void net_eval_mfma(const float* L, const NetRegs<HP, MD, KS_, KSH_, L1M>& W, int dim,
                                              int q_tanh, const float (&a)[MD], const float (&b)[MD], float tc, float ts,
                                              int lane, float (&S)[MD], float (&T)[MD], float (&Q)[MD],
                                              [[maybe_unused]] unsigned long long* st = nullptr) {
  using V = MfmaNet<HP, MD, KS_, KSH_>;
  constexpr int NT = V::NT, KS = V::KS, KSH = V::KSH, NTH = V::NTH, KS1 = V::KS1, K1 = V::K1;
  const int q = lane >> 4, r = lane & 15;
  [[maybe_unused]] unsigned long long t0 = ST_NOW();
  // ---- layer 1: h1[s] = relu(pre-activation of unit 4 s + q), the same ascending-k fma chain in both forms
  constexpr int K2 = L1M ? KSH : KS;        // k-steps of the hidden layer: unit 16 t + 4 q + e at step 4 t + e (L1M), 4 s + q (VALU)
  float h1[K2];
  f32x4s acc[NT];
  if constexpr (L1M) {
#pragma unroll
    for (int to = 0; to < NT; ++to) acc[to] = f32x4s{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS1; ++s) {
      float e4[4];      // (built per step: the same values in one K1-entry array cost 6 us per cfg-2 propose)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = 4 * s + j;           // compile-time
        e4[j] = k < MD ? a[k < MD ? k : 0] : k < 2 * MD ? b[(k >= MD && k < 2 * MD) ? k - MD : 0]
                : k == 2 * MD ? tc : k == 2 * MD + 1 ? ts : k == 2 * MD + 2 ? 1.f : 0.f;
      }
      const float mine = q == 0 ? e4[0] : q == 1 ? e4[1] : q == 2 ? e4[2] : e4[3];
#pragma unroll
      for (int to = 0; to < NT; ++to)
        acc[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(W.w1[to * KS1 + s], mine, acc[to], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < K2; ++s) h1[s] = fmaxf(acc[s >> 2][s & 3], 0.f);
  } else {
    float in[K1];
#pragma unroll
    for (int k = 0; k < K1; ++k)
      in[k] = k < MD ? a[k < MD ? k : 0] : k < 2 * MD ? b[(k >= MD && k < 2 * MD) ? k - MD : 0]
              : k == 2 * MD ? tc : k == 2 * MD + 1 ? ts : k == 2 * MD + 2 ? 1.f : 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float* rec = L + V::rec + (4 * s + q) * K1;
      float pre = 0.f;
#pragma unroll
      for (int k4 = 0; k4 < K1; k4 += 4) {
        const f32x4s w = *reinterpret_cast<const f32x4s*>(rec + k4);
#pragma unroll
        for (int j = 0; j < 4; ++j) pre = __builtin_fmaf(w[j], in[k4 + j], pre);
      }
      h1[s] = fmaxf(pre, 0.f);
    }
  }
  ST_ADD(0, t0);
  t0 = ST_NOW();
  // ---- layer 2
#pragma unroll
  for (int to = 0; to < NT; ++to) acc[to] = f32x4s{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < K2; ++s)
#pragma unroll
    for (int to = 0; to < NT; ++to)
      acc[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(W.w2[to * K2 + s], h1[s], acc[to], 0, 0, 0);
  float h2[KSH];
#pragma unroll
  for (int s = 0; s < KSH; ++s) {
    float bias;
    if constexpr (L1M) bias = W.bh[s];
    else bias = L[V::bh + 16 * (s >> 2) + 4 * q + (s & 3)];
    h2[s] = fmaxf(acc[s >> 2][s & 3] + bias, 0.f);
  }
  ST_ADD(1, t0);
  t0 = ST_NOW();
  // ---- heads
  f32x4s hv[NTH];
#pragma unroll
  for (int th = 0; th < NTH; ++th) {
    // two accumulators (even / odd k-steps): a single one would serialise on the 40-cycle dependent latency
    f32x4s c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
#pragma unroll
    for (int s = 0; s < KSH; ++s) {
      if (s & 1) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(W.whd[th * KSH + s], h2[s], c1, 0, 0, 0);
      else c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(W.whd[th * KSH + s], h2[s], c0, 0, 0, 0);
    }
    f32x4s bias;
    if constexpr (L1M) bias = f32x4s{W.bhd[4 * th], W.bhd[4 * th + 1], W.bhd[4 * th + 2], W.bhd[4 * th + 3]};
    else bias = *reinterpret_cast<const f32x4s*>(L + V::bhd + 16 * th + 4 * q);
    hv[th] = c0 + c1 + bias;
  }
  ST_ADD(2, t0);
  t0 = ST_NOW();
  // output o = head * MD + d sits in register o % 4 of tile o / 16 on the lane with q = (o % 16) / 4 of this chain
  float out[3 * MD];
#pragma unroll
  for (int o = 0; o < 3 * MD; ++o) {
    const int src = 16 * ((o & 15) >> 2) + r;
    out[o] = __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(hv[o >> 4][o & 3])));
  }
#pragma unroll
  for (int d = 0; d < MD; ++d) {
    if (d < dim) {
      S[d] = fast_tanh(out[d]) * (L1M ? W.es[L1M ? d : 0] : L[V::es + d]);
      T[d] = out[MD + d];
      Q[d] = (q_tanh ? fast_tanh(out[2 * MD + d]) : out[2 * MD + d]) * (L1M ? W.eq[L1M ? d : 0] : L[V::eq + d]);
    }
  }
#ifdef L2HMC_STAMPS
  asm volatile("" :: "v"(S[0]), "v"(T[0]), "v"(Q[0]));
#endif
  ST_ADD(3, t0);
}

// Target parameters in registers (x_dim <= 2 instance, the reference's toy targets): energy_grad() re-reads
// them from LDS with run-time offsets at every one of its 2 N + 2 calls, behind the network's LDS traffic; here
// they are read once.  Same arithmetic, same order as energy_grad (small_mlp.h).
template <int MD>
struct TargetRegs {
  static constexpr int KM = 2;              // components held (mog_model.py: two; more fall back to energy_grad)
  static constexpr bool kFits = MD <= 2;
  float mu[KM][MD], prec[KM][MD][MD], logc[KM];
  __device__ __forceinline__ void load(const float* Lt, int dim, int K) {
    const TargetView tv = target_view(dim, K);
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      logc[k] = k < K ? Lt[tv.logc + k] : 0.f;
#pragma unroll
      for (int i = 0; i < MD; ++i) {
        mu[k][i] = (k < K && i < dim) ? Lt[tv.mu + k * dim + i] : 0.f;
#pragma unroll
        for (int j = 0; j < MD; ++j)
          prec[k][i][j] = (k < K && i < dim && j < dim) ? Lt[tv.prec + (k * dim + i) * dim + j] : 0.f;
      }
    }
  }
  // E == nullptr: gradient only (the energy's logf is needed at the two ends of a trajectory, not inside it)
  __device__ __forceinline__ void eval(int dim, int K, int is_gaussian, float inv_temp, const float (&x)[MD], float* E,
                                       float (&g)[MD]) const {
    float V[KM];
    float vmax = -INFINITY;
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      if (k < K) {
        float quad = 0.f;
#pragma unroll
        for (int i = 0; i < MD; ++i) {
          if (i < dim) {
            float pd = 0.f;
#pragma unroll
            for (int j = 0; j < MD; ++j)
              if (j < dim) pd += prec[k][i][j] * (x[j] - mu[k][j]);
            quad += (x[i] - mu[k][i]) * pd;
          }
        }
        V[k] = -0.5f * quad + (is_gaussian ? 0.f : logc[k]);
        vmax = fmaxf(vmax, V[k]);
      }
    }
    float sw = 0.f;
#pragma unroll
    for (int d = 0; d < MD; ++d) g[d] = 0.f;
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      if (k < K) {
        const float w = is_gaussian ? 1.f : expf(V[k] - vmax);
        sw += w;
#pragma unroll
        for (int i = 0; i < MD; ++i) {
          if (i < dim) {
            float gi = 0.f;
#pragma unroll
            for (int j = 0; j < MD; ++j)
              if (j < dim) gi += (prec[k][i][j] + prec[k][j][i]) * (x[j] - mu[k][j]);
            g[i] += w * 0.5f * gi;
          }
        }
      }
    }
    if (E) {
      const float e = is_gaussian ? -V[0] : -(vmax + logf(sw));
      *E = e * inv_temp;
    }
#pragma unroll
    for (int d = 0; d < MD; ++d) g[d] = g[d] / sw * inv_temp;
  }
};

template <int HP, int MD, int KS_, int KSH_, bool L1M, bool TW = false>
__global__ __launch_bounds__(kSmallThreads) void small_traj_mfma_kernel(SmallTrajArgs a) {
  using V = MfmaNet<HP, MD, KS_, KSH_>;
  static_assert(!TW || (L1M && V::NT == 4), "the twin form is a latency form of the 64-unit instances");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const l2hmc_small_plan& P = a.plan;
  const int dim = P.x_dim, N = P.trajectory_length;
  const TargetView tv = target_view(P.target.dim, P.target.K);
  float* Lx = lds;
  float* Lv = Lx + V::size;
  float* Lt = Lv + V::size;
  float* Lm = Lt + tv.size;                       // masks [N][dim]
  float* Lts = Lm + ((N * dim + 3) & ~3);         // (cos, sin) of 2 pi step / N, [N][2]
  float* scr_all = Lts + ((2 * N + 3) & ~3);                   // [waves][16 chains][NTH * 16]: hand-off patch of the propose epilogue
  if (!P.hmc) {
    load_net_mfma<HP, MD, KS_, KSH_, L1M>(P.xnet, Lx, dim);
    load_net_mfma<HP, MD, KS_, KSH_, L1M>(P.vnet, Lv, dim);
  }
  load_target(P.target, Lt);
  for (int i = threadIdx.x; i < N * dim; i += kSmallThreads) Lm[i] = P.masks[i];
  for (int i = threadIdx.x; i < N; i += kSmallThreads) {       // utils/dynamics.py:105-110, once instead of per step
    const float arg = 6.28318530717958647692f * (float)i / (float)N;
    Lts[2 * i] = cosf(arg);
    Lts[2 * i + 1] = sinf(arg);
  }
  __syncthreads();                                // the only workgroup barrier of the kernel

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // twin form: waves 2 g, 2 g + 1 of the workgroup integrate the same group g of 16 rows (NetRegsTwin)
  [[maybe_unused]] const int part = TW ? (wave & 1) : 0;
  using WRegs = std::conditional_t<TW, NetRegsTwin<HP, MD, KS_, KSH_>, NetRegs<HP, MD, KS_, KSH_, L1M>>;
  WRegs Wx, Wv;
  if (!P.hmc) {
    if constexpr (TW) {
      Wx.load(Lx, lane, part);
      Wv.load(Lv, lane, part);
    } else {
      Wx.load(Lx, lane);
      Wv.load(Lv, lane);
    }
  }
  float* scr = scr_all + wave * 16 * V::NTH * 16;
  [[maybe_unused]] f32x4s* xch = reinterpret_cast<f32x4s*>(scr_all + (kSmallThreads / 64) * 16 * V::NTH * 16) +
                                 (wave >> 1) * 4 * V::NTH * 64;          // [group][2 parities][2 parts][NTH][64]
  [[maybe_unused]] int ncall = 0;
  const bool prop = a.prop_B > 0;
  const int64_t gw = TW ? (int64_t)blockIdx.x * (kSmallThreads / 128) + (wave >> 1)
                        : (int64_t)blockIdx.x * (kSmallThreads / 64) + wave;
  // trajectory mode: row r of [rows]; propose mode: chain r, direction (lane & 15) >> 3
  const int64_t r = prop ? gw * 8 + (lane & 7) : gw * 16 + (lane & 15);
  const bool live = r < (prop ? a.prop_B : a.rows);
  const int bwd = prop ? ((lane >> 3) & 1) : ((a.dir && live) ? a.dir[r] : 0);
  const float eps = P.eps;
  const float inv_temp = 1.f / P.target.temperature;
  const int isg = P.target.is_gaussian, K = P.target.K;
  [[maybe_unused]] unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  [[maybe_unused]] const unsigned long long st_begin = ST_NOW();

  float x[MD], v[MD];
#pragma unroll
  for (int d = 0; d < MD; ++d) {
    x[d] = (d < dim && live) ? a.x0[r * dim + d] : 0.f;
    if (prop)      // streams draw0 + 1 (forward momenta) and draw0 + 2 (backward), element chain * dim + d
      v[d] = (d < dim && live) ? philox_normal_at(a.seed, a.draw0 + 1 + bwd, r * dim + d) : 0.f;
    else
      v[d] = (d < dim && live) ? a.v0[r * dim + d] : 0.f;
  }
  float x_init[MD];
#pragma unroll
  for (int d = 0; d < MD; ++d) x_init[d] = x[d];
  TargetRegs<MD> tregs;
  const bool treg = TargetRegs<MD>::kFits && K <= TargetRegs<MD>::KM;       // uniform
  if (treg) tregs.load(Lt, dim, K);
  auto target = [&](const float (&xx)[MD], float* E, float (&gg)[MD]) {
    [[maybe_unused]] const unsigned long long tt = ST_NOW();
    float dummy;
    if (treg) tregs.eval(dim, K, isg, inv_temp, xx, E, gg);
    else energy_grad<MD>(Lt, dim, K, isg, inv_temp, xx, E ? E : &dummy, gg);
#ifdef L2HMC_STAMPS
    asm volatile("" :: "v"(gg[0]));
#endif
    ST_ADD(4, tt);
  };
  float g[MD], E0, E1;
  target(x, &E0, g);
  float kin0 = 0.f;
#pragma unroll
  for (int d = 0; d < MD; ++d) kin0 += v[d] * v[d];
  const float H0 = E0 + 0.5f * kin0;

  float logdet = 0.f;
  float S[MD], T[MD], Q[MD], bin[MD];
#pragma unroll
  for (int d = 0; d < MD; ++d) S[d] = T[d] = Q[d] = 0.f;
  for (int it = 0; it < N; ++it) {
    const int step = bwd ? N - 1 - it : it;       // utils/dynamics.py:294-296
    const float tc = Lts[2 * step], ts = Lts[2 * step + 1];
    const float* m = Lm + step * dim;
    for (int half = 0; half < 2; ++half) {
      if (half == 1) {
        for (int sub = 0; sub < 2; ++sub) {       // keep mask m then 1 - m (fwd) / 1 - m then m (bwd)
          const bool keep_is_m = (sub == 0) != (bwd != 0);
#pragma unroll
          for (int d = 0; d < MD; ++d) {
            const float k = d < dim ? (keep_is_m ? m[d] : 1.f - m[d]) : 1.f;
            bin[d] = k * x[d];
          }
          if (!P.hmc) {
            if constexpr (TW) net_eval_twin<HP, MD, KS_, KSH_>(Wx, dim, P.xnet.q_tanh, v, bin, tc, ts, lane, part, xch, ncall++ & 1, S, T, Q);
            else net_eval_mfma<HP, MD, KS_, KSH_, L1M>(Lx, Wx, dim, P.xnet.q_tanh, v, bin, tc, ts, lane, S, T, Q, st);
          }
          [[maybe_unused]] const unsigned long long tu = ST_NOW();
#pragma unroll
          for (int d = 0; d < MD; ++d) {
            if (d < dim) {
              const float k = keep_is_m ? m[d] : 1.f - m[d];
              const float s = (bwd ? -eps : eps) * S[d];
              const float drift = eps * (fast_exp(eps * Q[d]) * v[d] + T[d]);
              const float es_ = fast_exp(s);
              const float upd = bwd ? es_ * (x[d] - drift) : x[d] * es_ + drift;
              x[d] = k * x[d] + (1.f - k) * upd;
              logdet += (1.f - k) * s;
            }
          }
#ifdef L2HMC_STAMPS
          asm volatile("" :: "v"(x[0]), "v"(logdet));
#endif
          ST_ADD(5, tu);
        }
        target(x, nullptr, g);         // (the energy itself is needed only after the last step: below)
      }
      if (!P.hmc) {
        if constexpr (TW) net_eval_twin<HP, MD, KS_, KSH_>(Wv, dim, P.vnet.q_tanh, x, g, tc, ts, lane, part, xch, ncall++ & 1, S, T, Q);
        else net_eval_mfma<HP, MD, KS_, KSH_, L1M>(Lv, Wv, dim, P.vnet.q_tanh, x, g, tc, ts, lane, S, T, Q, st);
      }
      [[maybe_unused]] const unsigned long long tu2 = ST_NOW();
#pragma unroll
      for (int d = 0; d < MD; ++d) {
        if (d < dim) {
          const float s = (bwd ? -0.5f : 0.5f) * eps * S[d];
          const float kick = 0.5f * eps * (fast_exp(eps * Q[d]) * g[d] - T[d]);
          const float es_ = fast_exp(s);
          v[d] = bwd ? es_ * (v[d] + kick) : v[d] * es_ - kick;
          logdet += s;
        }
      }
#ifdef L2HMC_STAMPS
      asm volatile("" :: "v"(v[0]), "v"(logdet));
#endif
      ST_ADD(5, tu2);
    }
  }
#ifdef L2HMC_STAMPS
  if (a.stamps && lane == 0 && part == 0) {
    st[6] = ST_NOW() - st_begin;
    for (int i = 0; i < 8; ++i) a.stamps[gw * 8 + i] = st[i];
  }
#endif
  if (TW && part) return;             // both waves of a group hold the same state: the first one finishes (no barrier follows)
  target(x, &E1, g);
  float kin1 = 0.f;
#pragma unroll
  for (int d = 0; d < MD; ++d) kin1 += v[d] * v[d];
  const float H1 = E1 + 0.5f * kin1;
  if (prop) {
    // the backward half (lanes 8..15) hands its result to the forward half through the wave's LDS patch; lanes
    // 0..7 then mix and accept exactly as l2hmc_mix_accept(strict = 0) does
    const float pacc = accept_from_delta(H0 - H1 + logdet);
    float* slot = scr + (lane & 7) * (2 * MD + 1);
    if (lane >= 8 && lane < 16) {
#pragma unroll
      for (int d = 0; d < MD; ++d) {
        slot[d] = x[d];
        slot[MD + d] = v[d];
      }
      slot[2 * MD] = pacc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (!live || lane >= 8) return;
    const bool fwd = philox_uniform_at(a.seed, a.draw0, r) >= 0.5f;          // sampler.py:33 randint {0, 1}
    const float fm = fwd ? 1.f : 0.f, bm = 1.f - fm;
    const float pm = fm * pacc + bm * slot[2 * MD];
    const bool acc = a.mh_out ? (pm - philox_uniform_at(a.seed, a.draw0 + 3, r) >= 0.f) : false;   // :57-59
    if (a.px) a.px[r] = pm;
#pragma unroll
    for (int d = 0; d < MD; ++d) {
      if (d < dim) {
        const float xp = fm * x[d] + bm * slot[d];
        if (a.Lx) a.Lx[r * dim + d] = xp;
        if (a.Lv) a.Lv[r * dim + d] = fm * v[d] + bm * slot[MD + d];
        if (a.mh_out) a.mh_out[r * dim + d] = acc ? xp : x_init[d];
      }
    }
    return;
  }
  if (!live || lane >= 16) return;                 // the four lanes of a chain hold the same result
#pragma unroll
  for (int d = 0; d < MD; ++d) {
    if (d < dim) {
      a.x_out[r * dim + d] = x[d];
      a.v_out[r * dim + d] = v[d];
    }
  }
  if (a.sumlogdet) a.sumlogdet[r] = logdet;
  if (a.p_accept) a.p_accept[r] = accept_from_delta(H0 - H1 + logdet);   // utils/dynamics.py:312-319
}

template <int HP, int MD, int KS_, int KSH_>
static size_t small_mfma_lds(int dim, int K, int N) {
  return sizeof(float) * (2 * (size_t)MfmaNet<HP, MD, KS_, KSH_>::size + target_view(dim, K).size + ((N * dim + 3) & ~3) +
                          (size_t)((2 * N + 3) & ~3) +
                          (size_t)(kSmallThreads / 64) * 16 * MfmaNet<HP, MD, KS_, KSH_>::NTH * 16 +
                          (size_t)(kSmallThreads / 128) * 4 * MfmaNet<HP, MD, KS_, KSH_>::NTH * 64 * 4);   // twin exchange patches
}

template <int HP, int MD, int KS_, int KSH_, bool L1M, bool TW = false>
static int launch_small_mfma_form(const SmallTrajArgs& a, dim3 grid, hipStream_t st) {
  const l2hmc_small_plan& P = a.plan;
  const size_t lds = small_mfma_lds<HP, MD, KS_, KSH_>(P.x_dim, P.target.K, P.trajectory_length);
  L2HMC_REQUIRE(lds <= 160 * 1024, "small_trajectory: LDS image %zu B too large", lds);
  static DeviceOnce attr_once;   // dynamic LDS beyond 64 KiB needs the opt-in (host-side, not a stream op)
  if (attr_once.pending()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&small_traj_mfma_kernel<HP, MD, KS_, KSH_, L1M, TW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_once.done();
  }
  prof_before(kProfSmall, st);
#ifdef L2HMC_STAMPS
  SmallTrajArgs b = a;
  b.stamps = g_stamp_cls == 7 ? g_stamp_buf : nullptr;
  hipLaunchKernelGGL((small_traj_mfma_kernel<HP, MD, KS_, KSH_, L1M, TW>), grid, dim3(kSmallThreads), lds, st, b);
#else
  hipLaunchKernelGGL((small_traj_mfma_kernel<HP, MD, KS_, KSH_, L1M, TW>), grid, dim3(kSmallThreads), lds, st, a);
#endif
  prof_after(kProfSmall, st);
  L2HMC_CHECK_LAUNCH("small_trajectory");
  return L2HMC_OK;
}

// Three forms, chosen by the number of 16-row groups (l2hmc_small_plan::first_layer_form forces one; they walk sums in
// different groupings and agree to rounding: tests):
//   3  twin: two waves per group, every wave with a SIMD to itself (<= 512 groups on 1024 SIMDs; 64-unit instances)
//   1  first layer on the matrix pipe, one wave per group (<= 1024 groups)
//   2  first layer on the VALU: chip-filling batches
template <int HP, int MD, int KS_, int KSH_>
static int launch_small_mfma(const SmallTrajArgs& a, dim3 grid, hipStream_t st) {
  const int force = a.plan.first_layer_form;
  const int64_t groups = ceil_div(a.rows, 16);
  constexpr bool can_twin = HP == 64;
  if constexpr (can_twin) {
    if (force == 3 || (force == 0 && groups <= 512)) {
      const dim3 g2((unsigned)ceil_div(groups, kSmallThreads / 128));
      return launch_small_mfma_form<HP, MD, KS_, KSH_, true, true>(a, g2, st);
    }
  }
  const bool l1m = (force == 1 || force == 3) ? true : force == 2 ? false : groups <= 1024;
  return l1m ? launch_small_mfma_form<HP, MD, KS_, KSH_, true>(a, grid, st)
             : launch_small_mfma_form<HP, MD, KS_, KSH_, false>(a, grid, st);
}

static int check_target(const l2hmc_mog_target* t) {
  L2HMC_REQUIRE(t != nullptr, "target is NULL");
  L2HMC_REQUIRE(t->dim > 0 && t->dim <= kMaxDim && t->K > 0 && t->K <= kMaxMix,
                "target: dim=%d (max %d), K=%d (max %d)", t->dim, kMaxDim, t->K, kMaxMix);
  L2HMC_REQUIRE(t->mu && t->prec && (t->is_gaussian || t->log_const), "target: NULL parameter pointer");
  L2HMC_REQUIRE(!t->is_gaussian || t->K == 1, "target: gaussian needs K == 1");
  L2HMC_REQUIRE(t->temperature > 0.f, "target: temperature must be > 0");
  return L2HMC_OK;
}

}  // namespace l2hmc

using namespace l2hmc;

extern "C" int l2hmc_mog_energy_grad(const l2hmc_mog_target* tgt, const float* x, int64_t rows, float* energy,
                                     float* grad, l2hmc_stream_t stream) {
  if (int e = check_target(tgt)) return e;
  L2HMC_REQUIRE(rows >= 0, "mog_energy_grad: rows < 0");
  if (rows == 0) return L2HMC_OK;
  L2HMC_REQUIRE(x != nullptr, "mog_energy_grad: x is NULL");
  const size_t lds = sizeof(float) * target_view(tgt->dim, tgt->K).size;
  hipLaunchKernelGGL(mog_energy_grad_kernel, dim3((unsigned)ceil_div(rows, kSmallThreads)), dim3(kSmallThreads),
                     lds, (hipStream_t)stream, *tgt, x, rows, energy, grad);
  L2HMC_CHECK_LAUNCH("mog_energy_grad");
  return L2HMC_OK;
}

static int small_launch(const l2hmc_small_plan* plan, SmallTrajArgs a, l2hmc_stream_t stream);

extern "C" int l2hmc_small_propose(const l2hmc_small_plan* plan, const float* x, int64_t B, uint64_t seed,
                                   uint64_t draw0, float* Lx, float* Lv, float* px, float* x_out,
                                   l2hmc_stream_t stream) {
  L2HMC_REQUIRE(plan != nullptr, "small_propose: plan is NULL");
  L2HMC_REQUIRE(B >= 0, "small_propose: B < 0");
  if (B == 0) return L2HMC_OK;
  L2HMC_REQUIRE(x != nullptr, "small_propose: x is NULL");
  L2HMC_REQUIRE(!plan->hmc, "small_propose: the hmc sampler proposes with the forward trajectory only "
                            "(utils/sampler.py:30-32): use l2hmc_small_trajectory + l2hmc_mix_accept");
  SmallTrajArgs a{};
  a.plan = *plan; a.x0 = x; a.rows = 2 * B;
  a.prop_B = B; a.seed = seed; a.draw0 = draw0; a.Lx = Lx; a.Lv = Lv; a.px = px; a.mh_out = x_out;
  return small_launch(plan, a, stream);
}

extern "C" int l2hmc_small_trajectory(const l2hmc_small_plan* plan, const float* x0, const float* v0,
                                      const int32_t* dir, int64_t rows, float* x_out, float* v_out,
                                      float* sumlogdet, float* p_accept, l2hmc_stream_t stream) {
  L2HMC_REQUIRE(plan != nullptr, "small_trajectory: plan is NULL");
  L2HMC_REQUIRE(rows >= 0, "small_trajectory: rows < 0");
  if (rows == 0) return L2HMC_OK;
  L2HMC_REQUIRE(x0 && v0 && x_out && v_out, "small_trajectory: NULL pointer");
  SmallTrajArgs a{};
  a.plan = *plan; a.x0 = x0; a.v0 = v0; a.dir = dir; a.rows = rows;
  a.x_out = x_out; a.v_out = v_out; a.sumlogdet = sumlogdet; a.p_accept = p_accept;
  return small_launch(plan, a, stream);
}

static int small_launch(const l2hmc_small_plan* plan, SmallTrajArgs a, l2hmc_stream_t stream) {
  if (int e = check_target(&plan->target)) return e;
  const int dim = plan->x_dim, H = plan->num_nodes, N = plan->trajectory_length;
  L2HMC_REQUIRE(dim == plan->target.dim, "small_trajectory: x_dim=%d != target dim=%d", dim, plan->target.dim);
  L2HMC_REQUIRE(N > 0 && plan->masks != nullptr, "small_trajectory: bad trajectory_length / masks");
  L2HMC_REQUIRE(plan->first_layer_form >= 0 && plan->first_layer_form <= 3,
                "small_trajectory: first_layer_form=%d (0 by batch size, 1 matrix pipe, 2 VALU, 3 two waves per group)",
                plan->first_layer_form);
  const int64_t rows = a.rows;
  if (!plan->hmc) {
    L2HMC_REQUIRE(H > 0 && H <= 64, "small_trajectory: num_nodes=%d unsupported (1..64)", H);
    const l2hmc_dense_net* nets[2] = {&plan->xnet, &plan->vnet};
    for (const l2hmc_dense_net* n : nets) {
      L2HMC_REQUIRE(n->D == dim && n->Ka == dim && n->Kb == dim && n->H == H,
                    "small_trajectory: net shape (D=%d Ka=%d Kb=%d H=%d) != (x_dim=%d, num_nodes=%d)", n->D,
                    n->Ka, n->Kb, n->H, dim, H);
      L2HMC_REQUIRE(n->w1_t && n->wt && n->b1 && n->wh_t && n->bh && n->whd_t && n->bhd && n->coeff_s &&
                        n->coeff_q,
                    "small_trajectory: net has NULL weight pointer");
    }
  }
  const bool d2 = dim <= 2;          // the benchmark targets: x_dim 2 instance (chain state entirely in registers)
  const dim3 grid((unsigned)ceil_div(rows, (kSmallThreads / 64) * 16));     // 16 chains per wave
  hipStream_t st = (hipStream_t)stream;
  // (output tiles; hidden-layer k-steps, heads k-steps -- see MfmaNet): up to 16 units (1; 4, 4), up to 50 (4; 13, 14),
  // up to 64 (4; 16, 16)
  const int cls = (plan->hmc || H <= 16) ? 0 : (H <= 50 ? 1 : 2);
  if (d2) {
    if (cls == 0) return launch_small_mfma<16, 2, 4, 4>(a, grid, st);
    if (cls == 1) return launch_small_mfma<64, 2, 13, 14>(a, grid, st);
    return launch_small_mfma<64, 2, 16, 16>(a, grid, st);
  }
  if (cls == 0) return launch_small_mfma<16, kMaxDim, 4, 4>(a, grid, st);
  if (cls == 1) return launch_small_mfma<64, kMaxDim, 13, 14>(a, grid, st);
  return launch_small_mfma<64, kMaxDim, 16, 16>(a, grid, st);
}
