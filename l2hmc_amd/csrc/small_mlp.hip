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
};

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
// with another wave: lane (q = lane / 16, r = lane % 16) belongs to chain r of the wave.
//   layer 1  (K = 2 dim + 2: far too thin for a matrix instruction) on the VALU: the lane evaluates the 4 * NT
//            hidden units k(t, e) = 16 t + 4 q + e -- exactly the values it has to feed the next layer as ITS
//            slice of the k dimension, so no transpose and no LDS row exchange exists at all;
//   layer 2  v_mfma_f32_16x16x4_f32 with the WEIGHTS as the first operand (16 output units x 4 k) and the 16
//            chains as the second: 16 chains x HP x HP is a dense tile with no padding beyond num_nodes -> HP;
//            the product comes out as out[chain r][16 to + 4 q + e], again the lane's own k slice for the heads;
//   heads    the same instruction, N = 3 dim (<= 16 per tile); S, T, Q reach the chain's four lanes through a
//            1 KiB wave-private LDS patch (no barrier: LDS operations of one wave execute in order).
// Weights sit in LDS in fragment order ([tile][k-step][lane]: one conflict-free ds_read_b32 per MFMA).  The
// previous form (16 lanes per chain on the VALU, two workgroup barriers and HP broadcast LDS reads per network call)
// spent its time in LDS issue and barriers: profiles/r02_small_traj_before_after.txt.
// =====================================================================================================
using f32x4s = __attribute__((ext_vector_type(4))) float;

// HP: hidden width padded to the 16-wide output tiles.  KS: k-steps of the hidden layer; its inputs are produced on
// the VALU, so step s simply takes units 4 s + q (q = lane / 16): ceil(num_nodes / 4) steps.  KSH: k-steps of the heads;
// their inputs come out of the hidden layer's MFMAs as unit 16 t + 4 q + e in register e of tile t, so step s = 4 t + e
// takes those, and only steps whose smallest unit 16 t + e exists are run.  num_nodes 50: 13 and 14 steps instead of
// 16 and 16 (no multiplies by the padding up to 64).
template <int HP, int MD, int KS_, int KSH_>
struct MfmaNet {
  static constexpr int NT = HP / 16, KS = KS_, KSH = KSH_, NTH = (3 * MD + 15) / 16, REC = 4 + 2 * MD;
  static constexpr int rec = 0;                                 // [HP][REC]: b1, wt0, wt1, 0, W1a[MD], W1b[MD]
  static constexpr int w2 = rec + HP * REC;                     // [NT][KS][64]
  static constexpr int whd = w2 + NT * KS * 64;                 // [NTH][KSH][64]
  static constexpr int bh = whd + NTH * KSH * 64;               // [HP]
  static constexpr int bhd = bh + HP;                           // [NTH * 16]
  static constexpr int es = bhd + NTH * 16;                     // [MD] (padded to 8)
  static constexpr int eq = es + 8;
  static constexpr int size = eq + 8;
};

template <int HP, int MD, int KS_, int KSH_>
__device__ void load_net_mfma(const l2hmc_dense_net& n, float* L, int dim) {
  using V = MfmaNet<HP, MD, KS_, KSH_>;
  const int H = n.H, tid = threadIdx.x;
  for (int i = tid; i < HP * V::REC; i += kSmallThreads) {
    const int k = i / V::REC, f = i - k * V::REC;
    float val = 0.f;
    if (k < H) {
      if (f == 0) val = n.b1[k];
      else if (f == 1) val = n.wt[k];
      else if (f == 2) val = n.wt[H + k];
      else if (f >= 4 && f < 4 + MD) { if (f - 4 < dim) val = n.w1_t[(size_t)k * 2 * dim + (f - 4)]; }
      else if (f >= 4 + MD) { if (f - 4 - MD < dim) val = n.w1_t[(size_t)k * 2 * dim + dim + (f - 4 - MD)]; }
    }
    L[V::rec + i] = val;
  }
  for (int i = tid; i < V::NT * V::KS * 64; i += kSmallThreads) {
    const int lane = i & 63, s = (i >> 6) % V::KS, to = (i >> 6) / V::KS;
    const int out = 16 * to + (lane & 15), k = 4 * s + (lane >> 4);
    L[V::w2 + i] = (out < H && k < H) ? n.wh_t[(size_t)out * H + k] : 0.f;
  }
  for (int i = tid; i < V::NTH * V::KSH * 64; i += kSmallThreads) {
    const int lane = i & 63, s = (i >> 6) % V::KSH, th = (i >> 6) / V::KSH;
    const int o = 16 * th + (lane & 15), k = 16 * (s >> 2) + 4 * (lane >> 4) + (s & 3);
    L[V::whd + i] = (o < 3 * dim && k < H) ? n.whd_t[(size_t)o * H + k] : 0.f;
  }
  for (int i = tid; i < HP; i += kSmallThreads) L[V::bh + i] = i < H ? n.bh[i] : 0.f;
  for (int i = tid; i < V::NTH * 16; i += kSmallThreads) L[V::bhd + i] = i < 3 * dim ? n.bhd[i] : 0.f;
  for (int i = tid; i < 8; i += kSmallThreads) {
    L[V::es + i] = i < dim ? expf(n.coeff_s[i]) : 0.f;
    L[V::eq + i] = i < dim ? expf(n.coeff_q[i]) : 0.f;
  }
}

// The lane's fragments of the hidden layer and of the heads: with one wave per SIMD the register file (512 per lane)
// has room for both networks' (2 x (NT + NTH) x KS = 130 registers at num_nodes 50), so the matrix instructions
// take their weights straight from registers instead of one ds_read_b32 each, every call
template <int HP, int MD, int KS_, int KSH_>
struct NetRegs {
  using V = MfmaNet<HP, MD, KS_, KSH_>;
  float w2[V::NT * V::KS];
  float whd[V::NTH * V::KSH];
  __device__ __forceinline__ void load(const float* L, int lane) {
#pragma unroll
    for (int i = 0; i < V::NT * V::KS; ++i) w2[i] = L[V::w2 + i * 64 + lane];
#pragma unroll
    for (int i = 0; i < V::NTH * V::KSH; ++i) whd[i] = L[V::whd + i * 64 + lane];
  }
};

// (S, T, Q) = net([a, b, t]) for the 16 chains of this wave; every lane returns its own chain's values.
template <int HP, int MD, int KS_, int KSH_>
__device__ __forceinline__ void net_eval_mfma(const float* L, const NetRegs<HP, MD, KS_, KSH_>& W, int dim, int q_tanh,
                                              const float (&a)[MD], const float (&b)[MD], float tc, float ts,
                                              int lane, float* scr, float (&S)[MD], float (&T)[MD], float (&Q)[MD]) {
  using V = MfmaNet<HP, MD, KS_, KSH_>;
  constexpr int NT = V::NT, KS = V::KS, KSH = V::KSH, NTH = V::NTH, REC = V::REC;
  const int q = lane >> 4, r = lane & 15;
  float h1[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const float* rec = L + V::rec + (4 * s + q) * REC;
    const f32x4s r0 = *reinterpret_cast<const f32x4s*>(rec);
    float pre = r0[0] + (tc * r0[1] + ts * r0[2]);
#pragma unroll
    for (int d4 = 0; d4 < 2 * MD; d4 += 4) {
      const f32x4s w = *reinterpret_cast<const f32x4s*>(rec + 4 + d4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = d4 + j;                        // c < MD: first input, else second (zero weights beyond dim)
        pre += (c < MD ? a[c < MD ? c : 0] : b[c >= MD ? c - MD : 0]) * w[j];
      }
    }
    h1[s] = fmaxf(pre, 0.f);
  }
  f32x4s acc[NT];
#pragma unroll
  for (int to = 0; to < NT; ++to) acc[to] = f32x4s{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int to = 0; to < NT; ++to)
      acc[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(W.w2[to * KS + s], h1[s], acc[to], 0, 0, 0);
  float h2[KSH];
#pragma unroll
  for (int s = 0; s < KSH; ++s) {
    const int to = s >> 2, e = s & 3;
    h2[s] = fmaxf(acc[to][e] + L[V::bh + 16 * to + 4 * q + e], 0.f);
  }
#pragma unroll
  for (int th = 0; th < NTH; ++th) {
    // two accumulators (even / odd k-steps): a single one would serialise on the 40-cycle dependent latency
    f32x4s c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
#pragma unroll
    for (int s = 0; s < KSH; ++s) {
      if (s & 1) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(W.whd[th * KSH + s], h2[s], c1, 0, 0, 0);
      else c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(W.whd[th * KSH + s], h2[s], c0, 0, 0, 0);
    }
    const f32x4s bias = *reinterpret_cast<const f32x4s*>(L + V::bhd + 16 * th + 4 * q);
    *reinterpret_cast<f32x4s*>(scr + r * (NTH * 16) + 16 * th + 4 * q) = c0 + c1 + bias;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const float* mine = scr + r * (NTH * 16);
#pragma unroll
  for (int d = 0; d < MD; ++d) {
    if (d < dim) {
      const float s_ = mine[d], t_ = mine[dim + d], q_ = mine[2 * dim + d];
      S[d] = fast_tanh(s_) * L[V::es + d];
      T[d] = t_;
      Q[d] = (q_tanh ? fast_tanh(q_) : q_) * L[V::eq + d];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();          // the patch is free for the next call
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
    const float e = is_gaussian ? -V[0] : -(vmax + logf(sw));
    *E = e * inv_temp;
#pragma unroll
    for (int d = 0; d < MD; ++d) g[d] = g[d] / sw * inv_temp;
  }
};

template <int HP, int MD, int KS_, int KSH_>
__global__ __launch_bounds__(kSmallThreads) void small_traj_mfma_kernel(SmallTrajArgs a) {
  using V = MfmaNet<HP, MD, KS_, KSH_>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const l2hmc_small_plan& P = a.plan;
  const int dim = P.x_dim, N = P.trajectory_length;
  const TargetView tv = target_view(P.target.dim, P.target.K);
  float* Lx = lds;
  float* Lv = Lx + V::size;
  float* Lt = Lv + V::size;
  float* Lm = Lt + tv.size;                       // masks [N][dim]
  float* scr_all = Lm + ((N * dim + 3) & ~3);     // [waves][16 chains][NTH * 16]
  if (!P.hmc) {
    load_net_mfma<HP, MD, KS_, KSH_>(P.xnet, Lx, dim);
    load_net_mfma<HP, MD, KS_, KSH_>(P.vnet, Lv, dim);
  }
  load_target(P.target, Lt);
  for (int i = threadIdx.x; i < N * dim; i += kSmallThreads) Lm[i] = P.masks[i];
  __syncthreads();                                // the only workgroup barrier of the kernel

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  NetRegs<HP, MD, KS_, KSH_> Wx, Wv;
  if (!P.hmc) {
    Wx.load(Lx, lane);
    Wv.load(Lv, lane);
  }
  float* scr = scr_all + wave * 16 * V::NTH * 16;
  const bool prop = a.prop_B > 0;
  const int64_t gw = (int64_t)blockIdx.x * (kSmallThreads / 64) + wave;
  // trajectory mode: row r of [rows]; propose mode: chain r, direction (lane & 15) >> 3
  const int64_t r = prop ? gw * 8 + (lane & 7) : gw * 16 + (lane & 15);
  const bool live = r < (prop ? a.prop_B : a.rows);
  const int bwd = prop ? ((lane >> 3) & 1) : ((a.dir && live) ? a.dir[r] : 0);
  const float eps = P.eps;
  const float inv_temp = 1.f / P.target.temperature;
  const int isg = P.target.is_gaussian, K = P.target.K;

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
    if (treg) tregs.eval(dim, K, isg, inv_temp, xx, E, gg);
    else energy_grad<MD>(Lt, dim, K, isg, inv_temp, xx, E, gg);
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
    const float arg = 6.28318530717958647692f * (float)step / (float)N;
    const float tc = cosf(arg), ts = sinf(arg);
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
          if (!P.hmc) net_eval_mfma<HP, MD, KS_, KSH_>(Lx, Wx, dim, P.xnet.q_tanh, v, bin, tc, ts, lane, scr, S, T, Q);
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
        }
        target(x, &E1, g);
      }
      if (!P.hmc) net_eval_mfma<HP, MD, KS_, KSH_>(Lv, Wv, dim, P.vnet.q_tanh, x, g, tc, ts, lane, scr, S, T, Q);
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
    }
  }
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
                          (size_t)(kSmallThreads / 64) * 16 * MfmaNet<HP, MD, KS_, KSH_>::NTH * 16);
}

template <int HP, int MD, int KS_, int KSH_>
static int launch_small_mfma(const SmallTrajArgs& a, dim3 grid, hipStream_t st) {
  const l2hmc_small_plan& P = a.plan;
  const size_t lds = small_mfma_lds<HP, MD, KS_, KSH_>(P.x_dim, P.target.K, P.trajectory_length);
  L2HMC_REQUIRE(lds <= 160 * 1024, "small_trajectory: LDS image %zu B too large", lds);
  static DeviceOnce attr_once;   // dynamic LDS beyond 64 KiB needs the opt-in (host-side, not a stream op)
  if (attr_once.pending()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&small_traj_mfma_kernel<HP, MD, KS_, KSH_>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_once.done();
  }
  prof_before(kProfSmall, st);
  hipLaunchKernelGGL((small_traj_mfma_kernel<HP, MD, KS_, KSH_>), grid, dim3(kSmallThreads), lds, st, a);
  prof_after(kProfSmall, st);
  L2HMC_CHECK_LAUNCH("small_trajectory");
  return L2HMC_OK;
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
