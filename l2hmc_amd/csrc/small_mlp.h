// Device helpers shared by the toy-target integrator (small_mlp.hip) and its training kernel
// (small_train.hip): LDS images of the MLP weights and the mixture target, one network evaluation by the
// sixteen lanes of a chain, closed-form energy / gradient.
#pragma once
#include "common.h"


namespace l2hmc {

constexpr int kSmallThreads = 256;   // 16 chains per workgroup
constexpr int kLPC = 16;              // lanes per chain
constexpr int kMaxDim = L2HMC_MAX_SMALL_DIM;
constexpr int kMaxMix = L2HMC_MAX_MIX;

struct SmallNetView {   // offsets (floats) into the per-net LDS image; weight matrices are k-major [k][HP]
  int w1, wt, b1, wh, bh, whd, bhd, es, eq, size;
};

__host__ __device__ inline SmallNetView small_net_view(int HP, int dim) {
  SmallNetView v;
  int o = 0;
  v.w1 = o; o += 2 * dim * HP;      // [2*dim][HP]
  v.wt = o; o += 2 * HP;            // [2][HP]
  v.b1 = o; o += HP;
  v.wh = o; o += HP * HP;           // [k][n]
  v.bh = o; o += HP;
  v.whd = o; o += 3 * dim * HP;     // [3*dim][k]
  v.bhd = o; o += 3 * dim;
  v.es = o; o += dim;
  v.eq = o; o += dim;
  v.size = (o + 3) & ~3;
  return v;
}

template <int HP>
__device__ void load_net(const l2hmc_dense_net& n, float* L, int dim) {
  const SmallNetView v = small_net_view(HP, dim);
  const int H = n.H, tid = threadIdx.x;
  for (int i = tid; i < v.size; i += blockDim.x) L[i] = 0.f;
  __syncthreads();
  // packed global layouts: w1_t [H][2*dim], wh_t [H (out)][H (in)], whd_t [3][dim][H]
  for (int i = tid; i < H * 2 * dim; i += blockDim.x) L[v.w1 + (i % (2 * dim)) * HP + i / (2 * dim)] = n.w1_t[i];
  for (int i = tid; i < 2 * H; i += blockDim.x) L[v.wt + (i / H) * HP + (i % H)] = n.wt[i];
  for (int i = tid; i < H; i += blockDim.x) {
    L[v.b1 + i] = n.b1[i];
    L[v.bh + i] = n.bh[i];
  }
  for (int i = tid; i < H * H; i += blockDim.x) L[v.wh + (i % H) * HP + i / H] = n.wh_t[i];
  for (int i = tid; i < 3 * dim * H; i += blockDim.x) L[v.whd + (i / H) * HP + (i % H)] = n.whd_t[i];
  for (int i = tid; i < 3 * dim; i += blockDim.x) L[v.bhd + i] = n.bhd[i];
  for (int i = tid; i < dim; i += blockDim.x) {
    L[v.es + i] = expf(n.coeff_s[i]);
    L[v.eq + i] = expf(n.coeff_q[i]);
  }
}

// (S, T, Q) = net([a, b, t]) for the chain this lane belongs to.  `sub` = lane within the chain (0..15),
// `hrow` = the chain's HP-float LDS row for the hidden-vector exchange.  Must be called by all threads of the
// workgroup (it contains workgroup barriers).
// MD: compile-time bound on x_dim (2 for the benchmark targets, kMaxDim otherwise); register arrays and the
// unrolled loops are sized by it.
template <int HP, int MD = L2HMC_MAX_SMALL_DIM>
__device__ void net_eval(const float* L, int dim, int q_tanh, const float* a, const float* b, float tc, float ts,
                         int sub, float* hrow, float* S, float* T, float* Q) {
  constexpr int kMaxDim = MD;
  constexpr int UPL = HP / kLPC;           // hidden units per lane: n = sub * UPL + j
  const SmallNetView v = small_net_view(HP, dim);
  const int n0 = sub * UPL;
  float h[UPL];
#pragma unroll
  for (int j = 0; j < UPL; ++j) h[j] = L[v.b1 + n0 + j] + tc * L[v.wt + n0 + j] + ts * L[v.wt + HP + n0 + j];
#pragma unroll
  for (int k = 0; k < kMaxDim; ++k) {
    if (k < dim) {
#pragma unroll
      for (int j = 0; j < UPL; ++j)
        h[j] += a[k] * L[v.w1 + k * HP + n0 + j] + b[k] * L[v.w1 + (dim + k) * HP + n0 + j];
    }
  }
  __syncthreads();                          // previous readers of hrow are done
#pragma unroll
  for (int j = 0; j < UPL; ++j) hrow[n0 + j] = fmaxf(h[j], 0.f);
  __syncthreads();
  float h2[UPL];
#pragma unroll
  for (int j = 0; j < UPL; ++j) h2[j] = L[v.bh + n0 + j];
  // four input units per step: one 16-byte read of the hidden row (broadcast within the chain's 16 lanes) instead of
  // four scalar ones -- the loop is LDS-issue-bound (one weight read per UPL multiply-adds); same summation order
  using hvec4 = __attribute__((ext_vector_type(4))) float;
#pragma unroll 4
  for (int k = 0; k < HP; k += 4) {
    const hvec4 hv = *reinterpret_cast<const hvec4*>(hrow + k);
    const float* w = L + v.wh + k * HP + n0;
#pragma unroll
    for (int j = 0; j < UPL; ++j) {
      h2[j] += hv[0] * w[j];
      h2[j] += hv[1] * w[HP + j];
      h2[j] += hv[2] * w[2 * HP + j];
      h2[j] += hv[3] * w[3 * HP + j];
    }
  }
#pragma unroll
  for (int j = 0; j < UPL; ++j) h2[j] = fmaxf(h2[j], 0.f);
  // heads: k-split over the lanes, 16-lane butterfly
#pragma unroll
  for (int d = 0; d < kMaxDim; ++d) {
    if (d < dim) {
      float ps = 0.f, pt = 0.f, pq = 0.f;
      const float* ws = L + v.whd + (0 * dim + d) * HP + n0;
      const float* wt = L + v.whd + (1 * dim + d) * HP + n0;
      const float* wq = L + v.whd + (2 * dim + d) * HP + n0;
#pragma unroll
      for (int j = 0; j < UPL; ++j) {
        ps += h2[j] * ws[j];
        pt += h2[j] * wt[j];
        pq += h2[j] * wq[j];
      }
      // the chain's sixteen lanes are one DPP row: four VALU steps each instead of four ds_bpermute round trips
      static_assert(kLPC == 16, "row16_sum reduces over the 16 lanes of a chain");
      ps = row16_sum(ps);
      pt = row16_sum(pt);
      pq = row16_sum(pq);
      const float s = ps + L[v.bhd + d], t = pt + L[v.bhd + dim + d], q = pq + L[v.bhd + 2 * dim + d];
      S[d] = tanhf(s) * L[v.es + d];
      T[d] = t;
      Q[d] = (q_tanh ? tanhf(q) : q) * L[v.eq + d];
    }
  }
}

struct TargetView {   // LDS image of l2hmc_mog_target
  int mu, prec, logc, size;
};
__host__ __device__ inline TargetView target_view(int dim, int K) {
  TargetView t;
  t.mu = 0;
  t.prec = K * dim;
  t.logc = t.prec + K * dim * dim;
  t.size = (t.logc + K + 3) & ~3;
  return t;
}

// distributions.py:151-158 (GMM), :63-68 (Gaussian); gradient in closed form.  All loops over the dimension run
// to the compile-time bound MD with a guard, so x / g stay in registers (no dynamically indexed arrays).
template <int MD = L2HMC_MAX_SMALL_DIM>
__device__ inline void energy_grad(const float* Lt, int dim, int K, int is_gaussian, float inv_temp, const float* x,
                                   float* E, float* g) {
  const TargetView tv = target_view(dim, K);
  float V[kMaxMix];
  float vmax = -INFINITY;
#pragma unroll
  for (int k = 0; k < kMaxMix; ++k) {
    if (k < K) {
      float quad = 0.f;
#pragma unroll
      for (int i = 0; i < MD; ++i) {
        if (i < dim) {
          float pd = 0.f;
#pragma unroll
          for (int j = 0; j < MD; ++j)
            if (j < dim) pd += Lt[tv.prec + (k * dim + i) * dim + j] * (x[j] - Lt[tv.mu + k * dim + j]);
          quad += (x[i] - Lt[tv.mu + k * dim + i]) * pd;
        }
      }
      V[k] = -0.5f * quad + (is_gaussian ? 0.f : Lt[tv.logc + k]);
      vmax = fmaxf(vmax, V[k]);
    }
  }
  float sw = 0.f;
#pragma unroll
  for (int d = 0; d < MD; ++d) g[d] = 0.f;
#pragma unroll
  for (int k = 0; k < kMaxMix; ++k) {
    if (k < K) {
      const float w = is_gaussian ? 1.f : expf(V[k] - vmax);
      sw += w;
#pragma unroll
      for (int i = 0; i < MD; ++i) {
        if (i < dim) {
          float gi = 0.f;
#pragma unroll
          for (int j = 0; j < MD; ++j) {
            if (j < dim) {
              const float dj = x[j] - Lt[tv.mu + k * dim + j];
              gi += (Lt[tv.prec + (k * dim + i) * dim + j] + Lt[tv.prec + (k * dim + j) * dim + i]) * dj;
            }
          }
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

__device__ inline void load_target(const l2hmc_mog_target& t, float* Lt) {
  const TargetView tv = target_view(t.dim, t.K);
  for (int i = threadIdx.x; i < t.K * t.dim; i += blockDim.x) Lt[tv.mu + i] = t.mu[i];
  for (int i = threadIdx.x; i < t.K * t.dim * t.dim; i += blockDim.x) Lt[tv.prec + i] = t.prec[i];
  for (int i = threadIdx.x; i < t.K; i += blockDim.x) Lt[tv.logc + i] = t.is_gaussian ? 0.f : t.log_const[i];
}


}  // namespace l2hmc
