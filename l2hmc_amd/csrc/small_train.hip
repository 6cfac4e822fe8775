// Training step of the generic L2HMC sampler on the 2-D toy targets: loss and its gradient with
// respect to both MLPs and the step size, for
//   l2hmc/mog_model.py:324-355        (_create_loss: squared jump distance, x and z chains)
//   l2hmc/utils/sampler.py:28-55      (propose: direction picked per chain)
//   l2hmc/utils/dynamics.py:120-319   (augmented leapfrog, p_accept; eps = exp(alpha) :51-60)
//   l2hmc/utils/network.py:89-114     (network MLP with ScaleTanh on S and F)
// what `AdamOptimizer.minimize(loss)` (mog_model.py:357-363) differentiates.
//
// A chain's loss term depends only on that chain's trajectory, so ONE launch does forward, loss and
// the whole reverse pass for the sixteen chains of a workgroup, entirely on-chip:
//  * forward as small_traj_kernel (sixteen lanes per chain, weights in LDS), keeping only each
//    network call's inputs and the state it updated in an LDS tape (3*dim floats per call);
//  * reverse pass call by call: the network is RE-EVALUATED from the taped inputs (cheaper than
//    taping 2*H activations per call), the sub-update and the three heads are differentiated in
//    registers, the hidden-layer deltas travel through LDS rows (the transposed hidden matrix is
//    simply the global [out][in] layout, so both directions read LDS conflict-free), and the
//    target's Hessian-vector product (closed form for the mixture) closes the loop through
//    grad_energy;
//  * weight gradients: after each call every thread owns a fixed set of weight entries and adds the
//    sixteen chains' outer-product terms in slot order into registers; at the end each workgroup
//    writes its partial gradient (same layout as the packed weights), summed in order afterwards.
// No atomics anywhere: gradients are reproducible.
#include "small_mlp.h"
#include <type_traits>
#include "stq_dense.h"      // diagnostic stamp globals

namespace l2hmc {

constexpr int kSlotsMin = kSmallThreads / kLPC;   // chains per workgroup of the 256-thread instance (sizes the workspace)
// TH: threads per workgroup (256: 16 chains, one wave per SIMD; 512: 32 chains, two waves per SIMD)
// per-slot LDS row: a, b (2 MD) | tc ts | dout (3 MD) | dSS dQQ (2 MD)
constexpr int small_misc(int MD) { return 8 * MD + 8; }

template <int HP, int MD, int TH>
struct SmallAcc {   // per-thread weight-gradient accumulators of one network
  static constexpr int kMaxDim = MD;
  float wh[HP * HP / TH > 0 ? HP * HP / TH : 1];
  float w1[((2 * kMaxDim + 2) * HP + TH - 1) / TH];
  float whd[(3 * kMaxDim * HP + TH - 1) / TH];
  float b1, bh, bhd, cs, cq;
};

template <int HP, int MD, int TH>
__device__ __forceinline__ void acc_zero(SmallAcc<HP, MD, TH>& a) {
#pragma unroll
  for (int i = 0; i < (int)(sizeof(a.wh) / 4); ++i) a.wh[i] = 0.f;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(a.w1) / 4); ++i) a.w1[i] = 0.f;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(a.whd) / 4); ++i) a.whd[i] = 0.f;
  a.b1 = a.bh = a.bhd = a.cs = a.cq = 0.f;
}

// one network call's contribution of the workgroup's sixteen chains, slot order
// Row buffers Rh1 / Rh2 / Rd1 / Rd2 are [unit][slot] (the 16 chains of a unit contiguous): an owner thread reads a
// unit's sixteen values as four 16-byte LDS reads, and since entry e = tid + 256 i keeps its input unit k = tid % HP
// for every i, the h1 column is read once per call for all of the thread's hidden-matrix entries (68 ds_read_b128
// per call where the [slot][unit] layout took 512 scalar reads).  Sums run over the slots in order, as before.
using f32x4t = __attribute__((ext_vector_type(4))) float;
template <int SLOTS>
__device__ __forceinline__ float dot16(const f32x4t (&a)[SLOTS / 4], const float* b16) {
  float s = 0.f;
#pragma unroll
  for (int q4 = 0; q4 < SLOTS / 4; ++q4) {
    const f32x4t b = *reinterpret_cast<const f32x4t*>(b16 + 4 * q4);
#pragma unroll
    for (int e = 0; e < 4; ++e) s += a[q4][e] * b[e];
  }
  return s;
}
template <int HP, int MD, int TH>
__device__ __forceinline__ void acc_add(SmallAcc<HP, MD, TH>& a, int dim, const float* Rh1, const float* Rh2,
                                        const float* Rd1, const float* Rd2, const float* Rm) {
  constexpr int kMaxDim = MD, kSlots = TH / kLPC, kSmallThreads = TH;
  constexpr int SS = kSlots + 4;          // padded row stride: units 4 banks apart modulo 64, 16-byte rows
  static_assert(kSlots % 4 == 0 && kSmallThreads % HP == 0, "slot vectors / fixed input unit per owner thread");
  const int tid = threadIdx.x;
  {
    const int k = tid % HP;                                   // the same for every entry of this thread
    f32x4t hk[kSlots / 4];
#pragma unroll
    for (int q4 = 0; q4 < kSlots / 4; ++q4) hk[q4] = *reinterpret_cast<const f32x4t*>(Rh1 + k * SS + 4 * q4);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(a.wh) / 4); ++i) {
      const int e = tid + i * kSmallThreads;
      if (e < HP * HP) a.wh[i] += dot16<kSlots>(hk, Rd2 + (e / HP) * SS);
    }
  }
#pragma unroll
  for (int i = 0; i < (int)(sizeof(a.w1) / 4); ++i) {
    const int e = tid + i * kSmallThreads;
    if (e < (2 * dim + 2) * HP) {
      const int kin = e / HP, n = e - kin * HP;     // rows: a (dim), b (dim), cos, sin
      const int src = kin < 2 * dim ? kin : 2 * kMaxDim + (kin - 2 * dim);
      f32x4t mv[kSlots / 4];
#pragma unroll
      for (int q4 = 0; q4 < kSlots / 4; ++q4) mv[q4] = *reinterpret_cast<const f32x4t*>(Rm + src * SS + 4 * q4);
      a.w1[i] += dot16<kSlots>(mv, Rd1 + n * SS);
    }
  }
#pragma unroll
  for (int i = 0; i < (int)(sizeof(a.whd) / 4); ++i) {
    const int e = tid + i * kSmallThreads;
    if (e < 3 * dim * HP) {
      const int hd = e / HP, n = e - hd * HP;       // hd = head * dim + d
      const int src = 2 * kMaxDim + 2 + (hd / dim) * kMaxDim + hd % dim;
      f32x4t mv[kSlots / 4];
#pragma unroll
      for (int q4 = 0; q4 < kSlots / 4; ++q4) mv[q4] = *reinterpret_cast<const f32x4t*>(Rm + src * SS + 4 * q4);
      a.whd[i] += dot16<kSlots>(mv, Rh2 + n * SS);
    }
  }
  if (tid < HP) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int sl = 0; sl < kSlots; ++sl) {
      s1 += Rd1[tid * SS + sl];
      s2 += Rd2[tid * SS + sl];
    }
    a.b1 += s1;
    a.bh += s2;
  }
  if (tid < 3 * dim) {
    const int src = 2 * kMaxDim + 2 + (tid / dim) * kMaxDim + tid % dim;
    float s = 0.f;
#pragma unroll
    for (int sl = 0; sl < kSlots; ++sl) s += Rm[src * SS + sl];
    a.bhd += s;
  }
  if (tid < dim) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int sl = 0; sl < kSlots; ++sl) {
      s1 += Rm[(5 * kMaxDim + 2 + tid) * SS + sl];
      s2 += Rm[(6 * kMaxDim + 2 + tid) * SS + sl];
    }
    a.cs += s1;
    a.cq += s2;
  }
}

// flat layout of one network's gradient = the packed weights' segment order
// [w1_t (H x 2dim) | wt (2 x H) | b1 | wh_t (H x H) | bh | whd_t (3 dim x H) | bhd | coeff_s | coeff_q]
__host__ __device__ inline int small_grad_floats(int H, int dim) {
  return H * 2 * dim + 2 * H + H + H * H + H + 3 * dim * H + 3 * dim + 2 * dim;
}

template <int HP, int MD, int TH>
__device__ __forceinline__ void acc_store(const SmallAcc<HP, MD, TH>& a, int H, int dim, float* out) {
  constexpr int kMaxDim = MD, kSmallThreads = TH;
  (void)kMaxDim;
  const int tid = threadIdx.x;
  float* w1 = out;
  float* wt = w1 + H * 2 * dim;
  float* b1 = wt + 2 * H;
  float* wh = b1 + H;
  float* bh = wh + H * H;
  float* whd = bh + H;
  float* bhd = whd + 3 * dim * H;
  float* cs = bhd + 3 * dim;
  float* cq = cs + dim;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(a.wh) / 4); ++i) {
    const int e = tid + i * kSmallThreads;
    if (e < HP * HP) {
      const int n = e / HP, k = e - n * HP;
      if (n < H && k < H) wh[n * H + k] = a.wh[i];
    }
  }
#pragma unroll
  for (int i = 0; i < (int)(sizeof(a.w1) / 4); ++i) {
    const int e = tid + i * kSmallThreads;
    if (e < (2 * dim + 2) * HP) {
      const int kin = e / HP, n = e - kin * HP;
      if (n < H) {
        if (kin < 2 * dim) w1[n * 2 * dim + kin] = a.w1[i];
        else wt[(kin - 2 * dim) * H + n] = a.w1[i];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < (int)(sizeof(a.whd) / 4); ++i) {
    const int e = tid + i * kSmallThreads;
    if (e < 3 * dim * HP) {
      const int hd = e / HP, n = e - hd * HP;
      if (n < H) whd[hd * H + n] = a.whd[i];
    }
  }
  if (tid < H) {
    b1[tid] = a.b1;
    bh[tid] = a.bh;
  }
  if (tid < 3 * dim) bhd[tid] = a.bhd;
  if (tid < dim) {
    cs[tid] = a.cs;
    cq[tid] = a.cq;
  }
}

// network evaluation that also returns this lane's hidden units (post-relu)
template <int HP, int MD>
__device__ void net_eval_keep(const float* L, int dim, int q_tanh, const float* a, const float* b, float tc, float ts,
                              int sub, float* hrow, float* h1, float* h2, float* S, float* T, float* Q) {
  constexpr int kMaxDim = MD;
  constexpr int UPL = HP / kLPC;
  const SmallNetView v = small_net_view(HP, dim);
  const int n0 = sub * UPL;
#pragma unroll
  for (int j = 0; j < UPL; ++j) h1[j] = L[v.b1 + n0 + j] + tc * L[v.wt + n0 + j] + ts * L[v.wt + HP + n0 + j];
#pragma unroll
  for (int k = 0; k < kMaxDim; ++k) {
    if (k < dim) {
#pragma unroll
      for (int j = 0; j < UPL; ++j)
        h1[j] += a[k] * L[v.w1 + k * HP + n0 + j] + b[k] * L[v.w1 + (dim + k) * HP + n0 + j];
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < UPL; ++j) {
    h1[j] = fmaxf(h1[j], 0.f);
    hrow[n0 + j] = h1[j];
  }
  __syncthreads();
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

// Hessian(energy)(x) . u for the mixture / Gaussian target (second derivative of distributions.py:151-158):
//   H = sum_k r_k P_k - sum_k r_k g_k g_k^T + gbar gbar^T,   g_k = P_k (x - mu_k), r = softmax(V), gbar = sum r_k g_k
// with P_k the symmetrised precision; everything divided by the temperature.
template <int MD>
__device__ inline void energy_hvp(const float* Lt, int dim, int K, int is_gaussian, float inv_temp, const float* x,
                                  const float* u, float* out) {
  constexpr int kMaxDim = MD;
  const TargetView tv = target_view(dim, K);
  float V[kMaxMix];
  float vmax = -INFINITY;
#pragma unroll
  for (int k = 0; k < kMaxMix; ++k) {
    if (k < K) {
      float quad = 0.f;
#pragma unroll
      for (int i = 0; i < kMaxDim; ++i) {
        if (i < dim) {
          float pd = 0.f;
#pragma unroll
          for (int j = 0; j < kMaxDim; ++j)
            if (j < dim) pd += Lt[tv.prec + (k * dim + i) * dim + j] * (x[j] - Lt[tv.mu + k * dim + j]);
          quad += (x[i] - Lt[tv.mu + k * dim + i]) * pd;
        }
      }
      V[k] = -0.5f * quad + (is_gaussian ? 0.f : Lt[tv.logc + k]);
      vmax = fmaxf(vmax, V[k]);
    }
  }
  float sw = 0.f;
  float gbar[kMaxDim], acc[kMaxDim];
  float gu_bar = 0.f;
#pragma unroll
  for (int d = 0; d < kMaxDim; ++d) gbar[d] = acc[d] = 0.f;
#pragma unroll
  for (int k = 0; k < kMaxMix; ++k) {
    if (k < K) {
      const float w = is_gaussian ? 1.f : expf(V[k] - vmax);
      sw += w;
      float gk[kMaxDim], pu[kMaxDim];
      float gu = 0.f;
#pragma unroll
      for (int i = 0; i < kMaxDim; ++i) {
        gk[i] = pu[i] = 0.f;
        if (i < dim) {
#pragma unroll
          for (int j = 0; j < kMaxDim; ++j) {
            if (j < dim) {
              const float ps = 0.5f * (Lt[tv.prec + (k * dim + i) * dim + j] + Lt[tv.prec + (k * dim + j) * dim + i]);
              gk[i] += ps * (x[j] - Lt[tv.mu + k * dim + j]);
              pu[i] += ps * u[j];
            }
          }
          gu += gk[i] * u[i];
        }
      }
#pragma unroll
      for (int i = 0; i < kMaxDim; ++i) {
        if (i < dim) {
          acc[i] += w * (pu[i] - (is_gaussian ? 0.f : gk[i] * gu));
          gbar[i] += w * gk[i];
        }
      }
      gu_bar += w * gu;
    }
  }
#pragma unroll
  for (int i = 0; i < kMaxDim; ++i) {
    float h = acc[i] / sw;
    if (!is_gaussian) h += (gbar[i] / sw) * (gu_bar / sw);
    out[i] = i < dim ? h * inv_temp : 0.f;
  }
}

struct SmallTrainArgs {
  l2hmc_small_plan plan;
  const float* x0; const float* v0; const int* dir; int64_t rows;
  float scale, inv_count;
  float* x_out; float* v_out; float* p_accept; float* terms;
  float* part;        // [workgroups][2 * gsize + 1]: xnet gradient | vnet gradient | d loss / d eps
  unsigned long long* stamps;      // diagnostic builds only
};

#ifdef L2HMC_STAMPS
#define ST_NOW()                                                                              \
  ({                                                                                          \
    unsigned long long t_;                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    t_;                                                                                       \
  })
#define ST_ADD(slot, t0) st_[slot] += ST_NOW() - (t0)
#else
#define ST_NOW() 0ull
#define ST_ADD(slot, t0) do {} while (0)
#endif

template <int HP, int MD, int TH>
__global__ __launch_bounds__(TH) void small_train_kernel(SmallTrainArgs a) {
  constexpr int kSlots = TH / kLPC, kSmallThreads = TH, SS = kSlots + 4;
  // diagnostic cycle shares (class 7): 0 prologue, 1 forward, 2 loss, 3 net re-evaluation, 4 sub-update + head deltas,
  // 5 hidden deltas (d2, d1, input gradients), 6 owner pass, 7 total
  [[maybe_unused]] unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  [[maybe_unused]] const unsigned long long st_begin = ST_NOW();
  constexpr int kMaxDim = MD, kMisc = small_misc(MD);      // shadow the library-wide bound for this instance
  constexpr int UPL = HP / kLPC;
  extern __shared__ __attribute__((aligned(16))) float lds[];      // (every carve-out below is a multiple of 16 bytes)
  const l2hmc_small_plan& P = a.plan;
  const int dim = P.x_dim, N = P.trajectory_length, H = P.num_nodes;
  const int ncalls = 4 * N;
  const SmallNetView nv = small_net_view(HP, dim);
  const TargetView tv = target_view(P.target.dim, P.target.K);
  float* Lx = lds;
  float* Lv = Lx + nv.size;
  float* LxT = Lv + nv.size;                       // hidden matrices transposed: [n (out)][k (in)]
  float* LvT = LxT + HP * HP;
  float* Lt = LvT + HP * HP;
  float* Lm = Lt + tv.size;                        // masks [N][dim]
  float* hx = Lm + ((N * dim + 3) & ~3);           // [16][HP] forward hidden-vector exchange
  float* Rh1 = hx + kSlots * HP;                   // backward rows [HP][slots + 4 pad]
  float* Rh2 = Rh1 + SS * HP;
  float* Rd1 = Rh2 + SS * HP;
  float* Rd2 = Rd1 + SS * HP;
  float* Rm = Rd2 + SS * HP;                       // [kMisc][slots + 4 pad]
  float* tape = Rm + SS * kMisc;               // [16][ncalls][3 * dim]: a, b, updated state
  load_net<HP>(P.xnet, Lx, dim);
  load_net<HP>(P.vnet, Lv, dim);
  for (int i = threadIdx.x; i < HP * HP; i += kSmallThreads) {
    const int n = i / HP, k = i - n * HP;
    LxT[i] = (n < H && k < H) ? P.xnet.wh_t[n * H + k] : 0.f;
    LvT[i] = (n < H && k < H) ? P.vnet.wh_t[n * H + k] : 0.f;
  }
  load_target(P.target, Lt);
  for (int i = threadIdx.x; i < N * dim; i += kSmallThreads) Lm[i] = P.masks[i];
  __syncthreads();

  const int lsub = threadIdx.x & (kLPC - 1), slot = threadIdx.x / kLPC;
  const int64_t r = (int64_t)blockIdx.x * kSlots + slot;
  const bool live = r < a.rows;
  float* hrow = hx + slot * HP;
  float* mytape = tape + (size_t)slot * ncalls * 3 * dim;
  const int bwd = (a.dir && live) ? a.dir[r] : 0;
  const float eps = P.eps;
  const float inv_temp = 1.f / P.target.temperature;
  const int isg = P.target.is_gaussian, K = P.target.K;
  const int n0 = lsub * UPL;

  float x[kMaxDim], v[kMaxDim], xs[kMaxDim];
#pragma unroll
  for (int d = 0; d < kMaxDim; ++d) {
    x[d] = (d < dim && live) ? a.x0[r * dim + d] : 0.f;
    v[d] = (d < dim && live) ? a.v0[r * dim + d] : 0.f;
    xs[d] = x[d];
  }
  float g[kMaxDim], E0, E1;
  energy_grad<MD>(Lt, dim, K, isg, inv_temp, x, &E0, g);
  float kin0 = 0.f;
#pragma unroll
  for (int d = 0; d < kMaxDim; ++d) kin0 += v[d] * v[d];
  const float H0 = E0 + 0.5f * kin0;

  ST_ADD(0, st_begin);
  [[maybe_unused]] unsigned long long st_t = ST_NOW();
  // ------------------------------------------------------------------ forward, taping call inputs
  float logdet = 0.f;
  float S[kMaxDim], T[kMaxDim], Q[kMaxDim], bin[kMaxDim], h1[UPL], h2[UPL];
  for (int it = 0; it < N; ++it) {
    const int step = bwd ? N - 1 - it : it;
    const float arg = 6.28318530717958647692f * (float)step / (float)N;
    const float tc = cosf(arg), ts = sinf(arg);
    const float* m = Lm + step * dim;
    for (int half = 0; half < 2; ++half) {
      if (half == 1) {
        for (int sub = 0; sub < 2; ++sub) {
          const bool keep_is_m = (sub == 0) != (bwd != 0);
          float* tp = mytape + (size_t)(it * 4 + 1 + sub) * 3 * dim;
#pragma unroll
          for (int d = 0; d < kMaxDim; ++d) {
            const float k = d < dim ? (keep_is_m ? m[d] : 1.f - m[d]) : 1.f;
            bin[d] = k * x[d];
            if (d < dim && lsub == 0) {
              tp[d] = v[d];
              tp[dim + d] = bin[d];
              tp[2 * dim + d] = x[d];
            }
          }
          net_eval<HP, MD>(Lx, dim, P.xnet.q_tanh, v, bin, tc, ts, lsub, hrow, S, T, Q);
#pragma unroll
          for (int d = 0; d < kMaxDim; ++d) {
            if (d < dim) {
              const float k = keep_is_m ? m[d] : 1.f - m[d];
              const float s = (bwd ? -eps : eps) * S[d];
              const float drift = eps * (expf(eps * Q[d]) * v[d] + T[d]);
              const float upd = bwd ? expf(s) * (x[d] - drift) : x[d] * expf(s) + drift;
              x[d] = k * x[d] + (1.f - k) * upd;
              logdet += (1.f - k) * s;
            }
          }
        }
        energy_grad<MD>(Lt, dim, K, isg, inv_temp, x, &E1, g);
      }
      float* tp = mytape + (size_t)(it * 4 + (half ? 3 : 0)) * 3 * dim;
      if (lsub == 0) {
#pragma unroll
        for (int d = 0; d < kMaxDim; ++d)
          if (d < dim) {
            tp[d] = x[d];
            tp[dim + d] = g[d];
            tp[2 * dim + d] = v[d];
          }
      }
      net_eval<HP, MD>(Lv, dim, P.vnet.q_tanh, x, g, tc, ts, lsub, hrow, S, T, Q);
#pragma unroll
      for (int d = 0; d < kMaxDim; ++d) {
        if (d < dim) {
          const float s = (bwd ? -0.5f : 0.5f) * eps * S[d];
          const float kick = 0.5f * eps * (expf(eps * Q[d]) * g[d] - T[d]);
          v[d] = bwd ? expf(s) * (v[d] + kick) : v[d] * expf(s) - kick;
          logdet += s;
        }
      }
    }
  }
  energy_grad<MD>(Lt, dim, K, isg, inv_temp, x, &E1, g);
  float kin1 = 0.f;
#pragma unroll
  for (int d = 0; d < kMaxDim; ++d) kin1 += v[d] * v[d];
  const float H1 = E1 + 0.5f * kin1;
  const float p = accept_from_delta(H0 - H1 + logdet);

  ST_ADD(1, st_t);
  st_t = ST_NOW();
  // ------------------------------------------------------------------ loss (mog_model.py:336-355)
  float dist2 = 0.f;
#pragma unroll
  for (int d = 0; d < kMaxDim; ++d)
    if (d < dim) dist2 += (xs[d] - x[d]) * (xs[d] - x[d]);
  const float vj = dist2 * p + 1e-4f;
  const float term = a.scale / vj - vj / a.scale;
  if (live && lsub == 0) {
#pragma unroll
    for (int d = 0; d < kMaxDim; ++d)
      if (d < dim) {
        a.x_out[r * dim + d] = x[d];
        a.v_out[r * dim + d] = v[d];
      }
    a.p_accept[r] = p;
    a.terms[r] = term;
  }
  const float dvj = live ? a.inv_count * (-a.scale / (vj * vj) - 1.f / a.scale) : 0.f;
  const float dp = dvj * dist2;
  const float dD = p < 1.f ? dp * p : 0.f;         // through exp(min(., 0))
  float dx[kMaxDim], dv[kMaxDim];
#pragma unroll
  for (int d = 0; d < kMaxDim; ++d) {
    dx[d] = d < dim ? dvj * p * (-2.f) * (xs[d] - x[d]) - dD * g[d] : 0.f;
    dv[d] = d < dim ? -dD * v[d] : 0.f;
  }
  const float dl = dD;
  float deps = 0.f;

  ST_ADD(2, st_t);
  // ------------------------------------------------------------------ reverse pass
  SmallAcc<HP, MD, TH> accX, accV;
  acc_zero(accX);
  acc_zero(accV);
  for (int c = ncalls - 1; c >= 0; --c) {
    const int it = c >> 2, kind = c & 3;
    const bool vcall = kind == 0 || kind == 3;
    const int step = bwd ? N - 1 - it : it;
    const float arg = 6.28318530717958647692f * (float)step / (float)N;
    const float tc = cosf(arg), ts = sinf(arg);
    const float* m = Lm + step * dim;
    const float* tp = mytape + (size_t)c * 3 * dim;
    const float* L = vcall ? Lv : Lx;
    const float* LT = vcall ? LvT : LxT;
    const int q_tanh = vcall ? P.vnet.q_tanh : P.xnet.q_tanh;
    float ain[kMaxDim], st[kMaxDim];
#pragma unroll
    for (int d = 0; d < kMaxDim; ++d) {
      ain[d] = d < dim ? tp[d] : 0.f;
      bin[d] = d < dim ? tp[dim + d] : 0.f;
      st[d] = d < dim ? tp[2 * dim + d] : 0.f;
    }
    st_t = ST_NOW();
    net_eval_keep<HP, MD>(L, dim, q_tanh, ain, bin, tc, ts, lsub, hrow, h1, h2, S, T, Q);
    ST_ADD(3, st_t);
    st_t = ST_NOW();
    const SmallNetView nvw = small_net_view(HP, dim);
    // ---- sub-update backward -> head pre-activation gradients (replicated over the chain's lanes)
    float dS[kMaxDim], dT[kMaxDim], dQ[kMaxDim], dgd[kMaxDim], keep[kMaxDim];
#pragma unroll
    for (int d = 0; d < kMaxDim; ++d) {
      dS[d] = dT[d] = dQ[d] = dgd[d] = 0.f;
      keep[d] = 1.f;
      if (d >= dim) continue;
      const float eq = expf(eps * Q[d]);
      if (vcall) {
        const float vv = st[d], gg = bin[d], u = dv[d], he = 0.5f * eps;
        if (!bwd) {
          const float es = expf(he * S[d]);
          const float ds = u * vv * es + dl;
          dv[d] = u * es;
          dS[d] = ds * he; dT[d] = u * he; dQ[d] = -u * he * eq * gg * eps;
          dgd[d] = -u * he * eq;
          deps += ds * 0.5f * S[d] - u * 0.5f * (eq * gg - T[d]) - u * he * gg * eq * Q[d];
        } else {
          const float es = expf(-he * S[d]);
          const float vp = es * (vv + he * (eq * gg - T[d]));
          const float dw = u * es;
          const float ds = u * vp + dl;
          dv[d] = dw;
          dS[d] = -he * ds; dT[d] = -dw * he; dQ[d] = dw * he * eq * gg * eps;
          dgd[d] = dw * he * eq;
          deps += -0.5f * S[d] * ds + dw * 0.5f * (eq * gg - T[d]) + dw * he * gg * eq * Q[d];
        }
      } else {
        const bool keep_is_m = (kind == 1) != (bwd != 0);
        const float k = keep_is_m ? m[d] : 1.f - m[d], mi = 1.f - k;
        keep[d] = k;
        const float xx = st[d], vv = ain[d], u = dx[d];
        const float dy = mi * u;
        if (!bwd) {
          const float es = expf(eps * S[d]);
          const float ds = dy * xx * es + dl * mi;
          dx[d] = k * u + dy * es;
          dv[d] += dy * eps * eq;
          dS[d] = eps * ds; dT[d] = dy * eps; dQ[d] = dy * eps * eq * vv * eps;
          deps += ds * S[d] + dy * (eq * vv + T[d]) + dy * eps * vv * eq * Q[d];
        } else {
          const float es = expf(-eps * S[d]);
          const float w = xx - eps * (eq * vv + T[d]);
          const float dw = dy * es;
          const float ds = dy * (es * w) + dl * mi;
          dx[d] = k * u + dw;
          dv[d] -= dw * eps * eq;
          dS[d] = -eps * ds; dT[d] = -dw * eps; dQ[d] = -dw * eps * eq * vv * eps;
          deps += -S[d] * ds - dw * (eq * vv + T[d]) - dw * eps * vv * eq * Q[d];
        }
      }
    }
    float dout[3][kMaxDim], dSS[kMaxDim], dQQ[kMaxDim];
#pragma unroll
    for (int d = 0; d < kMaxDim; ++d) {
      dout[0][d] = dout[1][d] = dout[2][d] = dSS[d] = dQQ[d] = 0.f;
      if (d >= dim) continue;
      const float es_ = L[nvw.es + d], eq_ = L[nvw.eq + d];
      const float th = S[d] / es_;
      dout[0][d] = dS[d] * es_ * (1.f - th * th);
      dout[1][d] = dT[d];
      float daq = dQ[d] * eq_;
      if (q_tanh) {
        const float tq = Q[d] / eq_;
        daq *= 1.f - tq * tq;
      }
      dout[2][d] = daq;
      dSS[d] = dS[d] * S[d];
      dQQ[d] = dQ[d] * Q[d];
    }
    ST_ADD(4, st_t);
    st_t = ST_NOW();
    // ---- hidden deltas
    float d2[UPL], d1[UPL];
#pragma unroll
    for (int j = 0; j < UPL; ++j) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < kMaxDim; ++d)
        if (d < dim)
          s += dout[0][d] * L[nvw.whd + (0 * dim + d) * HP + n0 + j] + dout[1][d] * L[nvw.whd + (1 * dim + d) * HP + n0 + j] +
               dout[2][d] * L[nvw.whd + (2 * dim + d) * HP + n0 + j];
      d2[j] = h2[j] > 0.f ? s : 0.f;
    }
    __syncthreads();                               // previous call's owner pass has finished reading the rows
#pragma unroll
    for (int j = 0; j < UPL; ++j) {
      Rd2[(n0 + j) * SS + slot] = d2[j];
      Rh2[(n0 + j) * SS + slot] = h2[j];
      Rh1[(n0 + j) * SS + slot] = h1[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < UPL; ++j) d1[j] = 0.f;
    for (int n = 0; n < HP; ++n) {
      const float dn = Rd2[n * SS + slot];
      const float* w = LT + n * HP + n0;            // [n][k]: this lane's input units are contiguous
#pragma unroll
      for (int j = 0; j < UPL; ++j) d1[j] += dn * w[j];
    }
#pragma unroll
    for (int j = 0; j < UPL; ++j) {
      d1[j] = h1[j] > 0.f ? d1[j] : 0.f;
      Rd1[(n0 + j) * SS + slot] = d1[j];
    }
    // ---- gradient with respect to the two network inputs: k-split over the lanes + butterfly
    float da[kMaxDim], db[kMaxDim];
#pragma unroll
    for (int d = 0; d < kMaxDim; ++d) {
      da[d] = db[d] = 0.f;
      if (d >= dim) continue;
      float pa = 0.f, pb = 0.f;
#pragma unroll
      for (int j = 0; j < UPL; ++j) {
        pa += d1[j] * L[nvw.w1 + d * HP + n0 + j];
        pb += d1[j] * L[nvw.w1 + (dim + d) * HP + n0 + j];
      }
      pa = row16_sum(pa);
      pb = row16_sum(pb);
      da[d] = pa;
      db[d] = pb;
    }
    if (lsub == 0) {
      float* rm = Rm + slot;                // [kMisc][slots + 4 pad]; first-layer input rows compact: a at [0, dim), b at [dim, 2 dim)
#pragma unroll
      for (int d = 0; d < kMaxDim; ++d) {
        if (d < dim) {
          rm[d * SS] = ain[d];
          rm[(dim + d) * SS] = bin[d];
        }
        rm[(2 * kMaxDim + 2 + d) * SS] = dout[0][d];
        rm[(3 * kMaxDim + 2 + d) * SS] = dout[1][d];
        rm[(4 * kMaxDim + 2 + d) * SS] = dout[2][d];
        rm[(5 * kMaxDim + 2 + d) * SS] = dSS[d];
        rm[(6 * kMaxDim + 2 + d) * SS] = dQQ[d];
      }
      rm[(2 * kMaxDim) * SS] = live ? tc : 0.f;
      rm[(2 * kMaxDim + 1) * SS] = live ? ts : 0.f;
    }
    __syncthreads();
    ST_ADD(5, st_t);
    st_t = ST_NOW();
    if (vcall) acc_add<HP, MD, TH>(accV, dim, Rh1, Rh2, Rd1, Rd2, Rm);
    else acc_add<HP, MD, TH>(accX, dim, Rh1, Rh2, Rd1, Rd2, Rm);
    ST_ADD(6, st_t);
    // ---- into the upstream gradients
    if (vcall) {
      float u[kMaxDim], hv[kMaxDim];
#pragma unroll
      for (int d = 0; d < kMaxDim; ++d) u[d] = dgd[d] + db[d];
      energy_hvp<MD>(Lt, dim, K, isg, inv_temp, ain, u, hv);
#pragma unroll
      for (int d = 0; d < kMaxDim; ++d) dx[d] += da[d] + hv[d];
    } else {
#pragma unroll
      for (int d = 0; d < kMaxDim; ++d) {
        dv[d] += da[d];
        dx[d] += keep[d] * db[d];
      }
    }
  }
  // ------------------------------------------------------------------ partial gradients of this workgroup
  const int gsize = small_grad_floats(H, dim);
  float* out = a.part + (size_t)blockIdx.x * (2 * gsize + 1);
  acc_store<HP, MD, TH>(accX, H, dim, out);
  acc_store<HP, MD, TH>(accV, H, dim, out + gsize);
  __syncthreads();
  if (lsub == 0) Rm[slot] = live ? deps : 0.f;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int sl = 0; sl < kSlots; ++sl) s += Rm[sl];
    out[2 * gsize] = s;
  }
#ifdef L2HMC_STAMPS
  if (a.stamps && threadIdx.x == 0) {
    st_[7] = ST_NOW() - st_begin;
    for (int i = 0; i < 8; ++i) a.stamps[blockIdx.x * 8 + i] = st_[i];
  }
#endif
}

__global__ __launch_bounds__(256) void small_reduce_kernel(const float* __restrict__ part, int S, int64_t count,
                                                           float* __restrict__ out) {
  __shared__ float red[4][64];
  const int il = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + il;
  float t = 0.f;
  if (i < count)
    for (int s = sl; s < S; s += 4) t += part[(size_t)s * count + i];
  red[sl][il] = t;
  __syncthreads();
  if (sl == 0 && i < count) out[i] = (red[0][il] + red[1][il]) + (red[2][il] + red[3][il]);
}

template <int HP, int MD, int TH>
static size_t small_train_lds(int dim, int K, int N) {
  constexpr int kMisc = small_misc(MD), kSlots = TH / kLPC;
  return sizeof(float) * (2 * (size_t)small_net_view(HP, dim).size + 2 * (size_t)HP * HP + target_view(dim, K).size +
                          ((N * dim + 3) & ~3) + (size_t)kSlots * HP + 4 * (size_t)(kSlots + 4) * HP + (size_t)(kSlots + 4) * kMisc +
                          (size_t)kSlots * 4 * N * 3 * dim);
}

}  // namespace l2hmc

using namespace l2hmc;

extern "C" size_t l2hmc_small_train_ws_bytes(const l2hmc_small_plan* plan, int64_t rows) {
  if (!plan || rows <= 0 || plan->hmc) return 0;
  const size_t g = small_grad_floats(plan->num_nodes, plan->x_dim);
  return sizeof(float) * (size_t)ceil_div(rows, kSlotsMin) * (2 * g + 1);
}

extern "C" int l2hmc_small_train_step(const l2hmc_small_plan* plan, const float* x0, const float* v0,
                                      const int32_t* dir, int64_t rows, float scale, float inv_count, float* x_out,
                                      float* v_out, float* p_accept, float* terms, float* grads, void* ws,
                                      size_t ws_bytes, l2hmc_stream_t stream) {
  L2HMC_REQUIRE(plan != nullptr, "small_train_step: plan is NULL");
  L2HMC_REQUIRE(!plan->hmc, "small_train_step: hmc plans have no trainable networks");
  const int dim = plan->x_dim, H = plan->num_nodes, N = plan->trajectory_length;
  L2HMC_REQUIRE(plan->target.dim == dim && dim > 0 && dim <= kMaxDim && plan->target.K > 0 && plan->target.K <= kMaxMix,
                "small_train_step: bad target / x_dim");
  L2HMC_REQUIRE(plan->target.mu && plan->target.prec && (plan->target.is_gaussian || plan->target.log_const) &&
                    plan->target.temperature > 0.f,
                "small_train_step: bad target parameters");
  L2HMC_REQUIRE(N > 0 && plan->masks != nullptr && H > 0 && H <= 64, "small_train_step: bad plan (num_nodes 1..64)");
  L2HMC_REQUIRE(rows >= 0 && scale > 0.f, "small_train_step: bad rows / scale");
  if (rows == 0) return L2HMC_OK;
  L2HMC_REQUIRE(x0 && v0 && x_out && v_out && p_accept && terms && grads && ws, "small_train_step: NULL pointer");
  const l2hmc_dense_net* nets[2] = {&plan->xnet, &plan->vnet};
  for (const l2hmc_dense_net* n : nets) {
    L2HMC_REQUIRE(n->D == dim && n->Ka == dim && n->Kb == dim && n->H == H, "small_train_step: net shape mismatch");
    L2HMC_REQUIRE(n->w1_t && n->wt && n->b1 && n->wh_t && n->bh && n->whd_t && n->bhd && n->coeff_s && n->coeff_q,
                  "small_train_step: net has NULL weight pointer");
  }
  if (ws_bytes < l2hmc_small_train_ws_bytes(plan, rows)) {
    set_error("small_train_step: workspace %zu < %zu bytes", ws_bytes, l2hmc_small_train_ws_bytes(plan, rows));
    return L2HMC_ERR_WORKSPACE;
  }
  const int HP = H <= 16 ? 16 : 64;
  const bool d2 = dim <= 2;          // x_dim 2 instance for the benchmark targets
  hipStream_t s = (hipStream_t)stream;
  int nwg = 0;
  // (TH = 512 -- 32 chains, two waves per SIMD, one round of workgroups for 8192 chains -- was measured: 1.59 ms per
  //  step against 1.46 for two rounds of 16-chain workgroups: the kernel is bound by LDS throughput, not latency)
  auto run = [&](auto hp, auto md) -> int {
    constexpr int HPc = decltype(hp)::value, MDc = decltype(md)::value;
    const size_t lds = small_train_lds<HPc, MDc, 256>(dim, plan->target.K, N);
    L2HMC_REQUIRE(lds <= 160 * 1024, "small_train_step: LDS image %zu B too large (trajectory too long?)", lds);
    static DeviceOnce attr_once;
    if (attr_once.pending()) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&small_train_kernel<HPc, MDc, 256>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_once.done();
    }
    SmallTrainArgs a{*plan, x0, v0, dir, rows, scale, inv_count, x_out, v_out, p_accept, terms, static_cast<float*>(ws),
                     nullptr};
#ifdef L2HMC_STAMPS
    a.stamps = g_stamp_cls == 7 ? g_stamp_buf : nullptr;
#endif
    nwg = (int)ceil_div(rows, 256 / kLPC);
    hipLaunchKernelGGL((small_train_kernel<HPc, MDc, 256>), dim3(nwg), dim3(256), lds, s, a);
    return L2HMC_OK;
  };
  using I16 = std::integral_constant<int, 16>;
  using I64 = std::integral_constant<int, 64>;
  using I2 = std::integral_constant<int, 2>;
  using IM = std::integral_constant<int, kMaxDim>;
  int rc;
  if (d2) rc = HP == 16 ? run(I16{}, I2{}) : run(I64{}, I2{});
  else rc = HP == 16 ? run(I16{}, IM{}) : run(I64{}, IM{});
  if (rc) return rc;
  L2HMC_CHECK_LAUNCH("small_train");
  const int64_t count = 2 * (int64_t)small_grad_floats(H, dim) + 1;
  hipLaunchKernelGGL(small_reduce_kernel, dim3((unsigned)ceil_div(count, 64)), dim3(256), 0, s,
                     static_cast<const float*>(ws), nwg, count, grads);
  L2HMC_CHECK_LAUNCH("small_reduce");
  return L2HMC_OK;
}
