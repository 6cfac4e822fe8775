// Leapfrog sub-updates (standalone forms), accept probability, direction
// mixing / Metropolis-Hastings, and the host-side orchestration of the lattice
// trajectory: l2hmc/dynamics/gauge_dynamics.py:195-313, :412-609.
#include "stq_dense.h"
#include <atomic>
#include <math.h>

namespace l2hmc {

// =====================================================================
// standalone sub-updates: one wave per row, S/T/Q may be NULL (= 0, the
// hmc=True nets of gauge_dynamics.py:102-108).  dir: per-row array or NULL
// with `dir_all` for every row.
// =====================================================================
__global__ __launch_bounds__(256) void lf_update_v_kernel(
    const float* v, const float* __restrict__ grad, const float* __restrict__ S,
    const float* __restrict__ T, const float* __restrict__ Q, float eps, const int* __restrict__ dir,
    int dir_all, int64_t rows, int D, float* v_out, float* __restrict__ logdet,
    int accumulate) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int d = dir ? dir[row] : dir_all;
  float ld = 0.f;
  for (int c = lane; c < D; c += kWave) {
    const int64_t i = row * D + c;
    const float Sv = S ? S[i] : 0.f, Tv = T ? T[i] : 0.f, Qv = Q ? Q[i] : 0.f;
    const float s = (d ? -0.5f : 0.5f) * eps * Sv;
    const float kick = 0.5f * eps * (expf(eps * Qv) * grad[i] - Tv);
    const float vv = v[i];
    v_out[i] = d ? expf(s) * (vv + kick) : vv * expf(s) - kick;
    ld += s;
  }
  ld = wave_sum(ld);
  if (lane == 0 && logdet) logdet[row] = accumulate ? logdet[row] + ld : ld;
}

__global__ __launch_bounds__(256) void lf_update_x_kernel(
    const float* x, const float* __restrict__ v, const float* __restrict__ keep_f,
    const float* __restrict__ keep_b, const float* __restrict__ S, const float* __restrict__ T,
    const float* __restrict__ Q, float eps, const int* __restrict__ dir, int dir_all, int64_t rows,
    int D, float* x_out, float* __restrict__ logdet, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int d = dir ? dir[row] : dir_all;
  const float* keep = d ? keep_b : keep_f;
  float ld = 0.f;
  for (int c = lane; c < D; c += kWave) {
    const int64_t i = row * D + c;
    const float Sv = S ? S[i] : 0.f, Tv = T ? T[i] : 0.f, Qv = Q ? Q[i] : 0.f;
    const float k = keep[c];
    const float s = (d ? -eps : eps) * Sv;
    const float drift = eps * (expf(eps * Qv) * v[i] + Tv);
    const float xx = x[i];
    const float upd = d ? expf(s) * (xx - drift) : xx * expf(s) + drift;
    x_out[i] = k * xx + (1.f - k) * upd;
    ld += (1.f - k) * s;
  }
  ld = wave_sum(ld);
  if (lane == 0 && logdet) logdet[row] = accumulate ? logdet[row] + ld : ld;
}

// keep_inv[c] = 1 - keep[c]   (the m-bar of gauge_dynamics.py:671-673), for all steps at once
__global__ void invert_mask_kernel(const float* __restrict__ m, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = 1.f - m[i];
}

// logdet[row] (+)= sum_cb part[row][cb]
__global__ void reduce_parts_kernel(const float* __restrict__ part, int ncb, int64_t rows,
                                    float* __restrict__ out, int accumulate) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  double s = 0.0;       // up to 64 slots at x_dim 2048: the slot sum itself adds no fp32 rounding
  for (int c = 0; c < ncb; ++c) s += (double)part[r * ncb + c];
  out[r] = accumulate ? out[r] + (float)s : (float)s;
}

// gauge_dynamics.py:592-609 from Hamiltonians
__global__ void accept_prob_kernel(const float* __restrict__ h_old, const float* __restrict__ h_new,
                                   const float* __restrict__ sld, int64_t n, float* __restrict__ p) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  p[i] = accept_from_delta(h_old[i] - h_new[i] + sld[i]);
}

// Same from the pieces the trajectory has at hand.  The difference of the two
// O(100) Hamiltonians is formed in fp64 from the fp32 action / kinetic sums so
// that p carries no cancellation error beyond the inputs' own rounding.
__global__ void accept_from_parts_kernel(const float* __restrict__ act0, const float* __restrict__ kin0,
                                         const float* __restrict__ act1, const float* __restrict__ kin1,
                                         const float* __restrict__ ld_part, int ncb, float beta,
                                         int64_t n, float* __restrict__ sumlogdet,
                                         float* __restrict__ p) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double sd = 0.0;
  for (int c = 0; c < ncb; ++c) sd += (double)ld_part[i * ncb + c];
  const float s = (float)sd;
  if (sumlogdet) sumlogdet[i] = s;
  if (p) {
    const double dh = (double)beta * ((double)act0[i] - (double)act1[i]) +
                      ((double)kin0[i] - (double)kin1[i]) + sd;
    p[i] = accept_from_delta(dh);
  }
}

// gauge_dynamics.py:221-257 / utils/sampler.py:33-59
__global__ __launch_bounds__(256) void mix_accept_kernel(
    const float* __restrict__ x, const float* __restrict__ xf, const float* __restrict__ vf,
    const float* __restrict__ pf, const float* __restrict__ xb, const float* __restrict__ vb,
    const float* __restrict__ pb, const float* __restrict__ coin, const float* __restrict__ u,
    int strict, int64_t B, int D, float* __restrict__ x_prop, float* __restrict__ v_prop,
    float* __restrict__ p_out, float* __restrict__ x_out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const bool fwd = strict ? (coin[row] > 0.5f) : (coin[row] != 0.f);
  const float fm = fwd ? 1.f : 0.f, bm = 1.f - fm;
  // keep the reference's arithmetic (mask * a + (1 - mask) * b) so non-finite
  // values in the unselected branch propagate exactly as they do there
  const float p = fm * pf[row] + bm * pb[row];
  const bool acc = u ? (strict ? (p > u[row]) : (p - u[row] >= 0.f)) : false;
  const float am = acc ? 1.f : 0.f;
  if (lane == 0 && p_out) p_out[row] = p;
  for (int c = lane; c < D; c += kWave) {
    const int64_t i = row * D + c;
    const float xp = fm * xf[i] + bm * xb[i];
    if (x_prop) x_prop[i] = xp;
    if (v_prop) v_prop[i] = fm * vf[i] + bm * vb[i];
    if (x_out) x_out[i] = strict ? (am * xp + (1.f - am) * x[i]) : (acc ? xp : x[i]);
  }
}

// np.mod(x, 2*pi) in fp32 (gauge_model.py:1180,1388: the host-side wrap between MCMC steps)
__global__ __launch_bounds__(256) void wrap_angle_kernel(const float* x, int64_t n, float* out) {
  const float two_pi = 6.28318530717958647692f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float r = fmodf(x[i], two_pi);
    if (r < 0.f) r += two_pi;
    out[i] = r;
  }
}

// selected-direction mode: dir = coin > 0.5 ? fwd : bwd; v0 = that direction's momentum
__global__ __launch_bounds__(256) void select_dir_kernel(const float* __restrict__ coin,
                                                         const float* __restrict__ v0_f,
                                                         const float* __restrict__ v0_b, int64_t B, int D,
                                                         int* __restrict__ dir, float* __restrict__ v) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const int d = (coin[row] > 0.5f) ? 0 : 1;
  if (lane == 0) dir[row] = d;
  const float* src = d ? v0_b : v0_f;
  for (int c = lane; c < D; c += kWave) v[row * D + c] = src[row * D + c];
}

__global__ void fill_dir_kernel(int* __restrict__ dir, int64_t B) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 2 * B) dir[i] = (i < B) ? 0 : 1;
}

// selected-direction MH step (x_prop already is the selected trajectory)
__global__ __launch_bounds__(256) void accept_selected_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ xs,
                                                              const float* __restrict__ p,
                                                              const float* __restrict__ u, int64_t B, int D,
                                                              float* __restrict__ x_out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const float am = (p[row] > u[row]) ? 1.f : 0.f;
  for (int c = lane; c < D; c += kWave) {
    const int64_t i = row * D + c;
    x_out[i] = am * xs[i] + (1.f - am) * x[i];
  }
}

// =====================================================================
// host orchestration
// =====================================================================
struct GaugeWs {
  float* h1; float* h2; float* g; float* ld_part; float* mask_inv; float* fa; float* fb;
  float* act0; float* kin0; float* act1; float* kin1;
  float* pre_v; float* pre_x;      // kept first-layer products [rows][H] (NULL when the plan recomputes them)
  int* act_cols; int* act_cnt;     // [num_steps][2][D] / [num_steps][2]: columns a position sub-update moves (stq_dense.h)
  size_t bytes;
};

// Recurring first-layer products of a leapfrog step (gauge_dynamics.py:412-483 evaluates every one of them anew):
//   * the two position sub-updates of a step call XNet on (v, m.x) and (v, m_inv.x) with the SAME v: the momentum
//     half of the first-layer product (and, for ConvNet3D, the conv stack of v) is formed once;
//   * the momentum update that ends step s and the one that starts step s + 1 call VNet on the SAME (x, force(x)) --
//     only the time input differs, and that enters after the product: force, conv stacks and the whole first-layer
//     product of the second call are the first call's.
// The kept values are the fp32 accumulators of an ascending-k fma chain, so the results are bit-identical to the
// recomputing path (L2HMC_PLAN_RECOMPUTE; tests/test_gpu_parity.py::test_kept_products_equal_recomputed).
enum NetCarry { kCarryNone = 0, kCarryVSave, kCarryVUse, kCarryXFirst, kCarrySecond };

static bool use_fused(const l2hmc_gauge_plan* p) {
  return !(p->flags & L2HMC_PLAN_LAYERED) && fused_plan_supported(p);
}
static int gauge_hmax(const l2hmc_gauge_plan* p) { return p->hmc ? 0 : hmax(p->xnet.H, p->vnet.H); }
static int gauge_ncb(const l2hmc_gauge_plan* p) { return (int)ceil_div(2 * p->T * p->X, 32); }

static GaugeWs carve_gauge_ws(const l2hmc_gauge_plan* p, int64_t rows, void* ws) {
  const int D = 2 * p->T * p->X;
  const int H = gauge_hmax(p);
  char* base = static_cast<char*>(ws);
  size_t off = 0;
  auto take = [&](size_t nfloat) {
    float* ptr = reinterpret_cast<float*>(base + off);
    off += align_up(nfloat * sizeof(float), 256);
    return ptr;
  };
  GaugeWs w;
  w.h1 = take((size_t)rows * H);
  w.h2 = take((size_t)rows * H);
  w.g = take((size_t)rows * D);
  w.ld_part = take((size_t)rows * gauge_ncb(p));
  w.mask_inv = take((size_t)p->num_steps * D);
  const size_t nflat = (p->flags & L2HMC_PLAN_CONV3D) ? conv3d_nflat(p->T, p->X, p->xfront.F) : 0;
  w.fa = take((size_t)rows * nflat);
  w.fb = take((size_t)rows * nflat);
  w.act0 = take(rows);
  w.kin0 = take(rows);
  w.act1 = take(rows);
  w.kin1 = take(rows);
  const bool keep = !p->hmc && !(p->flags & L2HMC_PLAN_RECOMPUTE) && dense_net_tileable(&p->xnet) &&
                    dense_net_tileable(&p->vnet);
  w.pre_v = keep ? take((size_t)rows * p->vnet.H) : nullptr;
  w.pre_x = keep ? take((size_t)rows * p->xnet.H) : nullptr;
  w.act_cols = reinterpret_cast<int*>(take((size_t)p->num_steps * 2 * D));
  w.act_cnt = reinterpret_cast<int*>(take((size_t)p->num_steps * 2));
  w.bytes = off;
  return w;
}

static int check_plan(const l2hmc_gauge_plan* p) {
  L2HMC_REQUIRE(p != nullptr, "plan is NULL");
  L2HMC_REQUIRE(p->T > 0 && p->X > 0 && p->num_steps > 0, "plan: bad T/X/num_steps");
  L2HMC_REQUIRE(p->masks != nullptr, "plan: masks is NULL");
  const int D = 2 * p->T * p->X;
  if (!p->hmc) {
    const l2hmc_dense_net* nets[2] = {&p->xnet, &p->vnet};
    for (const l2hmc_dense_net* n : nets) {
      L2HMC_REQUIRE(n->D == D, "plan: net D=%d != lattice x_dim=%d", n->D, D);
      L2HMC_REQUIRE(dense_net_supported(n), "plan: net widths (Ka=%d, Kb=%d, H=%d) must be positive", n->Ka, n->Kb,
                    n->H);
      if (p->flags & L2HMC_PLAN_CONV3D) {
        const l2hmc_conv3d_front* f = (n == &p->xnet) ? &p->xfront : &p->vfront;
        L2HMC_REQUIRE(f->F > 0 && p->T % 4 == 0 && p->X % 4 == 0, "plan: conv3D needs T, X multiples of 4 and F > 0");
        L2HMC_REQUIRE(p->xfront.F == p->vfront.F, "plan: both nets must use the same num_filters");
        const int nf = conv3d_nflat(p->T, p->X, f->F);
        L2HMC_REQUIRE(n->Ka == nf && n->Kb == nf, "plan: conv3D trunk expects Ka=Kb=%d (got %d, %d)", nf, n->Ka,
                      n->Kb);
        L2HMC_REQUIRE(f->w1_a && f->b1_a && f->w2_a && f->b2_a && f->w1_b && f->b1_b && f->w2_b && f->b2_b,
                      "plan: conv front-end has NULL weight pointer");
      } else {
        L2HMC_REQUIRE(n->Ka == D && n->Kb == D, "plan: generic net expects Ka=Kb=x_dim (got %d, %d)", n->Ka,
                      n->Kb);
      }
      L2HMC_REQUIRE(n->w1_t && n->wt && n->b1 && n->wh_t && n->bh && n->whd_t && n->bhd && n->coeff_s &&
                        n->coeff_q,
                    "plan: net has NULL weight pointer");
    }
  }
  return L2HMC_OK;
}

// one S/T/Q evaluation of `net` on (a, b*mask), fused with the v or x update
static int net_update(const l2hmc_gauge_plan* plan, const l2hmc_dense_net* net, const float* a,
                      const float* b, const float* cm_f, const float* cm_b, const int* dir,
                      const float tcs[4], int64_t rows, int mode, float* x, float* v, const float* g,
                      const float* keep_f, const float* keep_b, float eps, const GaugeWs& w, int ncb,
                      hipStream_t stream, int carry = kCarryNone, const int* cols_f = nullptr,
                      const int* cols_b = nullptr, const int* cnt_f = nullptr, const int* cnt_b = nullptr,
                      int64_t dir_split = -1) {
  if (carry == kCarryVUse) {
    // same (x, force) as the call that kept the product: only bias + time term + relu remain of the first layer
    L1FinishArgs f{};
    f.pre = w.pre_v; f.out = w.h1; f.N = net->H; f.rows = rows;
    f.bias = net->b1; f.wt0 = net->wt; f.wt1 = net->wt + net->H;
    f.dir = dir; f.tc_f = tcs[0]; f.ts_f = tcs[1]; f.tc_b = tcs[2]; f.ts_b = tcs[3];
    if (int e = launch_l1_finish(f, stream)) return e;
  } else {
  if (plan->flags & L2HMC_PLAN_CONV3D) {
    // conv_net.py:251-262: both inputs through their conv stacks, then the dense trunk on the features
    const l2hmc_conv3d_front* f = (net == &plan->xnet) ? &plan->xfront : &plan->vfront;
    ConvFrontArgs c{};
    c.T = plan->T; c.X = plan->X; c.F = f->F;
    c.in[0] = a; c.in[1] = b; c.cmask_f = cm_f; c.cmask_b = cm_b; c.dir = dir;
    c.w1[0] = f->w1_a; c.b1[0] = f->b1_a; c.w2[0] = f->w2_a; c.b2[0] = f->b2_a;
    c.w1[1] = f->w1_b; c.b1[1] = f->b1_b; c.w2[1] = f->w2_b; c.b2[1] = f->b2_b;
    c.out[0] = w.fa; c.out[1] = w.fb; c.ldo = net->Ka; c.rows = rows;
    c.only = carry == kCarrySecond ? 2 : 0;       // the first input's features are still in w.fa
    if (int e = launch_conv3d_front(c, stream)) return e;
    a = w.fa; b = w.fb; cm_f = cm_b = nullptr;
  }
  GemmReluArgs l1{};
  l1.A1 = a; l1.lda1 = net->Ka; l1.K1 = net->Ka;
  l1.A2 = b; l1.lda2 = net->Kb;
  l1.cmask_f = cm_f; l1.cmask_b = cm_b;
  l1.dir = dir;
  l1.Wt = net->w1_t; l1.K = net->Ka + net->Kb; l1.N = net->H;
  l1.bias = net->b1; l1.wt0 = net->wt; l1.wt1 = net->wt + net->H;
  l1.tc_f = tcs[0]; l1.ts_f = tcs[1]; l1.tc_b = tcs[2]; l1.ts_b = tcs[3];
  l1.out = w.h1; l1.ldo = net->H; l1.rows = rows;
  if (carry == kCarryVSave) { l1.acc_out = w.pre_v; l1.k_dump = l1.K; }
  if (carry == kCarryXFirst) { l1.acc_out = w.pre_x; l1.k_dump = net->Ka; }
  if (carry == kCarrySecond) { l1.acc_in = w.pre_x; l1.k_begin = net->Ka; }
  if (int e = launch_gemm_relu(l1, stream)) return e;
  }

  GemmReluArgs l2{};
  l2.A1 = w.h1; l2.lda1 = net->H; l2.K1 = net->H;
  l2.Wt = net->wh_t; l2.K = net->H; l2.N = net->H;
  l2.bias = net->bh;
  l2.out = w.h2; l2.ldo = net->H; l2.rows = rows;
  if (int e = launch_gemm_relu(l2, stream)) return e;

  HeadsArgs h{};
  h.A = w.h2; h.lda = net->H; h.K = net->H;
  h.Wt = net->whd_t; h.bhd = net->bhd; h.cs = net->coeff_s; h.cq = net->coeff_q;
  h.q_tanh = net->q_tanh; h.D = net->D; h.rows = rows; h.mode = mode;
  h.x = x; h.v = v; h.g = g; h.dir = dir; h.keep_f = keep_f; h.keep_b = keep_b; h.eps = eps;
  h.ld_part = w.ld_part; h.ncb = ncb;
  if (mode == 2 && dir_split >= 0 && !(plan->flags & L2HMC_PLAN_ALL_COLUMNS)) {
    // only the columns this sub-update moves (the others' S, T, Q are multiplied by 1 - keep = 0)
    h.cols_f = cols_f; h.cols_b = cols_b; h.cnt_f = cnt_f; h.cnt_b = cnt_b; h.dir_split = dir_split;
  }
  return launch_heads(h, stream);
}

// the kept products travel through the aligned tile loads only: every operand of a first layer 16-byte aligned
static bool carry_possible(const l2hmc_gauge_plan* p, const float* x, const float* v, const GaugeWs& w) {
  auto ok = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const int D = 2 * p->T * p->X;
  return w.pre_v != nullptr && (D % 4) == 0 && ok(x) && ok(v) && ok(p->masks) && ok(p->xnet.w1_t) && ok(p->vnet.w1_t) &&
         ok(p->xnet.b1) && ok(p->vnet.b1) && ok(p->xnet.wt) && ok(p->vnet.wt) && (p->xnet.H % 4) == 0 &&
         (p->vnet.H % 4) == 0;
}

// one augmented leapfrog step in place; log-det goes to w.ld_part (+=)
// use_prev: the previous step of this launch kept VNet's first-layer product and the force of the current x;
// keep_last: keep them for the next step
static int leapfrog_step(const l2hmc_gauge_plan* p, float beta, int step, float* x, float* v,
                         const int* dir, int64_t rows, const GaugeWs& w, hipStream_t stream, bool carry = false,
                         bool use_prev = false, bool keep_last = false, int64_t dir_split = -1) {
  const int D = 2 * p->T * p->X;
  const int N = p->num_steps;
  const int sf = step, sb = N - 1 - step;   // gauge_dynamics.py:453-457
  // _format_time (:625-633): fp32 cos/sin of 2*pi*i/N
  float tcs[4];
  {
    const float two_pi = (float)(2.0 * M_PI);
    const float af = two_pi * (float)sf / (float)N, ab = two_pi * (float)sb / (float)N;
    tcs[0] = cosf(af); tcs[1] = sinf(af); tcs[2] = cosf(ab); tcs[3] = sinf(ab);
  }
  const float* m_f = p->masks + (size_t)sf * D;
  const float* m_b = p->masks + (size_t)sb * D;
  const float* mi_f = w.mask_inv + (size_t)sf * D;
  const float* mi_b = w.mask_inv + (size_t)sb * D;
  const int ncb = gauge_ncb(p);
  const unsigned rgrid = (unsigned)ceil_div(rows, 4);

  for (int half = 0; half < 2; ++half) {
    if (half == 1) {
      // position sub-updates between the two momentum half-kicks.
      // forward (:428-438): (m, m_inv) then (m_inv, m); backward (:466-476): (m_inv, m) then (m, m_inv)
      for (int sub = 0; sub < 2; ++sub) {
        const float* kf = sub == 0 ? m_f : mi_f;
        const float* kb = sub == 0 ? mi_b : m_b;
        if (p->hmc) {
          hipLaunchKernelGGL(lf_update_x_kernel, dim3(rgrid), dim3(256), 0, stream, x, v, kf, kb, nullptr,
                             nullptr, nullptr, p->eps, dir, 0, rows, D, x, nullptr, 0);
          L2HMC_CHECK_LAUNCH("lf_update_x");
        } else {
          // active columns: keep = mask -> list 0 of that mask row, keep = 1 - mask -> list 1 (active_cols_kernel)
          const int wf = sub == 0 ? 0 : 1, wb = sub == 0 ? 1 : 0;
          const int* lf = w.act_cols + ((size_t)sf * 2 + wf) * D;
          const int* lb = w.act_cols + ((size_t)sb * 2 + wb) * D;
          if (int e = net_update(p, &p->xnet, v, x, kf, kb, dir, tcs, rows, /*mode x*/ 2, x, v, nullptr, kf, kb,
                                 p->eps, w, ncb, stream, !carry ? kCarryNone : sub == 0 ? kCarryXFirst : kCarrySecond,
                                 lf, lb, w.act_cnt + sf * 2 + wf, w.act_cnt + sb * 2 + wb, dir_split))
            return e;
        }
      }
    }
    // momentum half-kick (:423-425 and :440-442); w.g still holds force(x) when the previous step kept it
    const bool reuse = half == 0 && use_prev && !p->hmc;
    if (!reuse)
      if (int e = launch_u1_action_force(x, rows, p->T, p->X, beta, nullptr, w.g, nullptr, nullptr, stream))
        return e;
    if (p->hmc) {
      hipLaunchKernelGGL(lf_update_v_kernel, dim3(rgrid), dim3(256), 0, stream, v, w.g, nullptr, nullptr,
                         nullptr, p->eps, dir, 0, rows, D, v, nullptr, 0);
      L2HMC_CHECK_LAUNCH("lf_update_v");
    } else {
      if (int e = net_update(p, &p->vnet, x, w.g, nullptr, nullptr, dir, tcs, rows, /*mode v*/ 1, x, v, w.g,
                             nullptr, nullptr, p->eps, w, ncb, stream,
                             reuse ? kCarryVUse : (half == 1 && keep_last) ? kCarryVSave : kCarryNone))
        return e;
    }
  }
  return L2HMC_OK;
}

static int prepare_ws(const l2hmc_gauge_plan* p, int64_t rows, const GaugeWs& w, hipStream_t stream) {
  const int D = 2 * p->T * p->X;
  const int n = p->num_steps * D;
  hipLaunchKernelGGL(invert_mask_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, stream, p->masks,
                     w.mask_inv, n);
  L2HMC_CHECK_LAUNCH("invert_mask");
  if (int e = launch_active_cols(p->masks, p->num_steps, D, w.act_cols, w.act_cnt, stream)) return e;
  if (hipMemsetAsync(w.ld_part, 0, sizeof(float) * (size_t)rows * gauge_ncb(p), stream) != hipSuccess) {
    set_error("hipMemsetAsync(ld_part) failed");
    return L2HMC_ERR_HIP;
  }
  return L2HMC_OK;
}

// x, v: [rows][D] integrated in place over all num_steps; optional sumlogdet / p
// dir_split: rows [0, dir_split) are known to run forward and [dir_split, rows) backward (-1: directions are whatever
// `dir` says per row; then every heads launch forms all columns)
static int trajectory_inplace(const l2hmc_gauge_plan* p, float beta, float* x, float* v, const int* dir,
                              int64_t rows, float* sumlogdet, float* p_accept, const GaugeWs& w,
                              hipStream_t stream, int64_t dir_split = -1) {
  const int D = 2 * p->T * p->X;
  if (use_fused(p))
    return launch_fused_trajectory(p, beta, 0, p->num_steps, x, v, dir, rows, x, v, sumlogdet, 0, p_accept,
                                   stream);
  if (int e = prepare_ws(p, rows, w, stream)) return e;
  if (p_accept) {
    if (int e = launch_u1_action_force(x, rows, p->T, p->X, beta, w.act0, nullptr, nullptr, nullptr, stream))
      return e;
    if (int e = l2hmc_kinetic_energy(v, rows, D, w.kin0, stream)) return e;
  }
  const bool carry = carry_possible(p, x, v, w);
  // (measured and dropped: the two halves of the batch on two streams, so that one half's launch boundaries and kernel
  //  tails overlap the other half's matrix work -- half-size kernels lose more than the overlap returns: cfg 4
  //  9.75 -> 10.18 ms, cfg 3 layered 2.58 -> 2.63 ms)
  for (int step = 0; step < p->num_steps; ++step)
    if (int e = leapfrog_step(p, beta, step, x, v, dir, rows, w, stream, carry, carry && step > 0,
                              carry && step + 1 < p->num_steps, dir ? dir_split : rows))
      return e;
  if (p_accept) {
    if (int e = launch_u1_action_force(x, rows, p->T, p->X, beta, w.act1, nullptr, nullptr, nullptr, stream))
      return e;
    if (int e = l2hmc_kinetic_energy(v, rows, D, w.kin1, stream)) return e;
  }
  if (p_accept || sumlogdet) {
    hipLaunchKernelGGL(accept_from_parts_kernel, dim3((unsigned)ceil_div(rows, 256)), dim3(256), 0, stream,
                       w.act0, w.kin0, w.act1, w.kin1, w.ld_part, gauge_ncb(p), beta, rows, sumlogdet,
                       p_accept);
    L2HMC_CHECK_LAUNCH("accept_from_parts");
  }
  return L2HMC_OK;
}

static int copy_async(void* dst, const void* src, size_t bytes, hipStream_t s) {
  if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) {
    set_error("hipMemcpyAsync failed");
    return L2HMC_ERR_HIP;
  }
  return L2HMC_OK;
}

}  // namespace l2hmc

using namespace l2hmc;

// ---------------------------------------------------------------------------
extern "C" int l2hmc_lf_update_v(const float* v, const float* grad, const float* S, const float* T,
                                 const float* Q, float eps, int32_t dir, int64_t rows, int32_t D,
                                 float* v_out, float* logdet, l2hmc_stream_t stream) {
  L2HMC_REQUIRE(rows >= 0 && D > 0 && (dir == 0 || dir == 1), "lf_update_v: bad arguments");
  if (rows == 0) return L2HMC_OK;
  L2HMC_REQUIRE(v && grad && v_out, "lf_update_v: NULL pointer");
  hipLaunchKernelGGL(lf_update_v_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                     v, grad, S, T, Q, eps, nullptr, dir, rows, D, v_out, logdet, 0);
  L2HMC_CHECK_LAUNCH("lf_update_v");
  return L2HMC_OK;
}

extern "C" int l2hmc_lf_update_x(const float* x, const float* v, const float* keep, const float* S,
                                 const float* T, const float* Q, float eps, int32_t dir, int64_t rows,
                                 int32_t D, float* x_out, float* logdet, l2hmc_stream_t stream) {
  L2HMC_REQUIRE(rows >= 0 && D > 0 && (dir == 0 || dir == 1), "lf_update_x: bad arguments");
  if (rows == 0) return L2HMC_OK;
  L2HMC_REQUIRE(x && v && keep && x_out, "lf_update_x: NULL pointer");
  hipLaunchKernelGGL(lf_update_x_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                     x, v, keep, keep, S, T, Q, eps, nullptr, dir, rows, D, x_out, logdet, 0);
  L2HMC_CHECK_LAUNCH("lf_update_x");
  return L2HMC_OK;
}

extern "C" int l2hmc_accept_prob(const float* h_old, const float* h_new, const float* sumlogdet, int64_t n,
                                 float* p, l2hmc_stream_t stream) {
  L2HMC_REQUIRE(n >= 0, "accept_prob: n < 0");
  if (n == 0) return L2HMC_OK;
  L2HMC_REQUIRE(h_old && h_new && sumlogdet && p, "accept_prob: NULL pointer");
  hipLaunchKernelGGL(accept_prob_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     h_old, h_new, sumlogdet, n, p);
  L2HMC_CHECK_LAUNCH("accept_prob");
  return L2HMC_OK;
}

extern "C" int l2hmc_mix_accept(const float* x, const float* xf, const float* vf, const float* pf,
                                const float* xb, const float* vb, const float* pb, const float* coin,
                                const float* u, int32_t strict, int64_t B, int32_t D, float* x_prop,
                                float* v_prop, float* p, float* x_out, l2hmc_stream_t stream) {
  L2HMC_REQUIRE(B >= 0 && D > 0, "mix_accept: bad shape");
  if (B == 0) return L2HMC_OK;
  L2HMC_REQUIRE(xf && vf && pf && xb && vb && pb && coin, "mix_accept: NULL pointer");
  L2HMC_REQUIRE(x_out == nullptr || (x && u), "mix_accept: x_out needs x and u");
  hipLaunchKernelGGL(mix_accept_kernel, dim3((unsigned)ceil_div(B, 4)), dim3(256), 0, (hipStream_t)stream, x,
                     xf, vf, pf, xb, vb, pb, coin, u, strict, B, D, x_prop, v_prop, p, x_out);
  L2HMC_CHECK_LAUNCH("mix_accept");
  return L2HMC_OK;
}

extern "C" int l2hmc_wrap_angle(const float* x, int64_t n, float* out, l2hmc_stream_t stream) {
  L2HMC_REQUIRE(n >= 0, "wrap_angle: n < 0");
  if (n == 0) return L2HMC_OK;
  L2HMC_REQUIRE(x && out, "wrap_angle: NULL pointer");
  hipLaunchKernelGGL(wrap_angle_kernel, dim3((unsigned)hmin(ceil_div(n, 256), 2048)), dim3(256), 0,
                     (hipStream_t)stream, x, n, out);
  L2HMC_CHECK_LAUNCH("wrap_angle");
  return L2HMC_OK;
}

// ---------------------------------------------------------------------------
extern "C" size_t l2hmc_stq_ws_bytes(int64_t rows, int32_t H) {
  return 2 * align_up((size_t)rows * H * sizeof(float), 256);
}

extern "C" int l2hmc_stq_dense(const l2hmc_dense_net* net, const float* a, const float* b, const float* bmask,
                               float t_cos, float t_sin, int64_t rows, float* S, float* T, float* Q, void* ws,
                               size_t ws_bytes, l2hmc_stream_t stream) {
  L2HMC_REQUIRE(net != nullptr && rows >= 0, "stq_dense: bad arguments");
  if (rows == 0) return L2HMC_OK;
  L2HMC_REQUIRE(dense_net_supported(net), "stq_dense: widths (D=%d, Ka=%d, Kb=%d, H=%d) must be positive", net->D,
                net->Ka, net->Kb, net->H);
  L2HMC_REQUIRE(a && b && S && T && Q && ws, "stq_dense: NULL pointer");
  if (ws_bytes < l2hmc_stq_ws_bytes(rows, net->H)) {
    set_error("stq_dense: workspace %zu < %zu bytes", ws_bytes, l2hmc_stq_ws_bytes(rows, net->H));
    return L2HMC_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  float* h1 = static_cast<float*>(ws);
  float* h2 = reinterpret_cast<float*>(static_cast<char*>(ws) + align_up((size_t)rows * net->H * sizeof(float), 256));
  GemmReluArgs l1{};
  l1.A1 = a; l1.lda1 = net->Ka; l1.K1 = net->Ka;
  l1.A2 = b; l1.lda2 = net->Kb;
  l1.cmask_f = bmask; l1.cmask_b = bmask;
  l1.Wt = net->w1_t; l1.K = net->Ka + net->Kb; l1.N = net->H;
  l1.bias = net->b1; l1.wt0 = net->wt; l1.wt1 = net->wt + net->H;
  l1.tc_f = l1.tc_b = t_cos; l1.ts_f = l1.ts_b = t_sin;
  l1.out = h1; l1.ldo = net->H; l1.rows = rows;
  if (int e = launch_gemm_relu(l1, s)) return e;
  GemmReluArgs l2{};
  l2.A1 = h1; l2.lda1 = net->H; l2.K1 = net->H;
  l2.Wt = net->wh_t; l2.K = net->H; l2.N = net->H;
  l2.bias = net->bh; l2.out = h2; l2.ldo = net->H; l2.rows = rows;
  if (int e = launch_gemm_relu(l2, s)) return e;
  HeadsArgs h{};
  h.A = h2; h.lda = net->H; h.K = net->H;
  h.Wt = net->whd_t; h.bhd = net->bhd; h.cs = net->coeff_s; h.cq = net->coeff_q;
  h.q_tanh = net->q_tanh; h.D = net->D; h.rows = rows; h.mode = 0;
  h.S = S; h.T = T; h.Q = Q;
  return launch_heads(h, s);
}

extern "C" size_t l2hmc_stq_conv3d_ws_bytes(int64_t rows, int32_t H, int32_t T, int32_t X, int32_t F) {
  return l2hmc_stq_ws_bytes(rows, H) + 2 * align_up(sizeof(float) * (size_t)rows * conv3d_nflat(T, X, F), 256);
}

extern "C" int l2hmc_stq_conv3d(const l2hmc_conv3d_front* front, const l2hmc_dense_net* net, int32_t T, int32_t X,
                                const float* a, const float* b, const float* bmask, float t_cos, float t_sin,
                                int64_t rows, float* S, float* Tr, float* Q, void* ws, size_t ws_bytes,
                                l2hmc_stream_t stream) {
  L2HMC_REQUIRE(front && net && rows >= 0 && T > 0 && X > 0, "stq_conv3d: bad arguments");
  if (rows == 0) return L2HMC_OK;
  const int nf = conv3d_nflat(T, X, front->F);
  L2HMC_REQUIRE(net->Ka == nf && net->Kb == nf, "stq_conv3d: trunk expects Ka=Kb=%d (got %d, %d)", nf, net->Ka,
                net->Kb);
  L2HMC_REQUIRE(a && b && S && Tr && Q && ws, "stq_conv3d: NULL pointer");
  const size_t need = l2hmc_stq_conv3d_ws_bytes(rows, net->H, T, X, front->F);
  if (ws_bytes < need) {
    set_error("stq_conv3d: workspace %zu < %zu bytes", ws_bytes, need);
    return L2HMC_ERR_WORKSPACE;
  }
  char* base = static_cast<char*>(ws);
  const size_t fbytes = align_up(sizeof(float) * (size_t)rows * nf, 256);
  float* fa = reinterpret_cast<float*>(base);
  float* fb = reinterpret_cast<float*>(base + fbytes);
  ConvFrontArgs c{};
  c.T = T; c.X = X; c.F = front->F;
  c.in[0] = a; c.in[1] = b; c.cmask_f = bmask; c.cmask_b = bmask;
  c.w1[0] = front->w1_a; c.b1[0] = front->b1_a; c.w2[0] = front->w2_a; c.b2[0] = front->b2_a;
  c.w1[1] = front->w1_b; c.b1[1] = front->b1_b; c.w2[1] = front->w2_b; c.b2[1] = front->b2_b;
  c.out[0] = fa; c.out[1] = fb; c.ldo = nf; c.rows = rows;
  if (int e = launch_conv3d_front(c, (hipStream_t)stream)) return e;
  return l2hmc_stq_dense(net, fa, fb, nullptr, t_cos, t_sin, rows, S, Tr, Q, base + 2 * fbytes,
                         ws_bytes - 2 * fbytes, stream);
}

// ---------------------------------------------------------------------------
extern "C" size_t l2hmc_gauge_ws_bytes(const l2hmc_gauge_plan* plan, int64_t rows) {
  if (!plan || rows < 0) return 0;
  return carve_gauge_ws(plan, rows, nullptr).bytes;
}

extern "C" int l2hmc_gauge_plan_fused(const l2hmc_gauge_plan* plan) {
  if (int e = check_plan(plan)) return -e;
  return use_fused(plan) ? 1 : 0;
}

extern "C" int l2hmc_gauge_leapfrog(const l2hmc_gauge_plan* plan, float beta, int32_t step, float* x, float* v,
                                    const int32_t* dir, int64_t rows, float* logdet, void* ws, size_t ws_bytes,
                                    l2hmc_stream_t stream) {
  if (int e = check_plan(plan)) return e;
  L2HMC_REQUIRE(rows >= 0 && step >= 0 && step < plan->num_steps, "gauge_leapfrog: bad rows/step");
  if (rows == 0) return L2HMC_OK;
  L2HMC_REQUIRE(x && v && ws, "gauge_leapfrog: NULL pointer");
  const GaugeWs w = carve_gauge_ws(plan, rows, ws);
  if (ws_bytes < w.bytes) {
    set_error("gauge_leapfrog: workspace %zu < %zu bytes", ws_bytes, w.bytes);
    return L2HMC_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  if (use_fused(plan))
    return launch_fused_trajectory(plan, beta, step, step + 1, x, v, dir, rows, x, v, logdet, 1, nullptr, s);
  if (int e = prepare_ws(plan, rows, w, s)) return e;
  if (int e = leapfrog_step(plan, beta, step, x, v, dir, rows, w, s, carry_possible(plan, x, v, w), false, false,
                            dir ? -1 : rows))
    return e;
  if (logdet) {
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)ceil_div(rows, 256)), dim3(256), 0, s, w.ld_part,
                       gauge_ncb(plan), rows, logdet, 1);
    L2HMC_CHECK_LAUNCH("reduce_parts");
  }
  return L2HMC_OK;
}

extern "C" int l2hmc_gauge_trajectory(const l2hmc_gauge_plan* plan, float beta, const float* x0,
                                      const float* v0, const int32_t* dir, int64_t rows, float* x_out,
                                      float* v_out, float* sumlogdet, float* p_accept, void* ws,
                                      size_t ws_bytes, l2hmc_stream_t stream) {
  if (int e = check_plan(plan)) return e;
  L2HMC_REQUIRE(rows >= 0, "gauge_trajectory: rows < 0");
  if (rows == 0) return L2HMC_OK;
  L2HMC_REQUIRE(x0 && v0 && x_out && v_out && ws, "gauge_trajectory: NULL pointer");
  const GaugeWs w = carve_gauge_ws(plan, rows, ws);
  if (ws_bytes < w.bytes) {
    set_error("gauge_trajectory: workspace %zu < %zu bytes", ws_bytes, w.bytes);
    return L2HMC_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const size_t nb = sizeof(float) * (size_t)rows * 2 * plan->T * plan->X;
  if (x_out != x0)
    if (int e = copy_async(x_out, x0, nb, s)) return e;
  if (v_out != v0)
    if (int e = copy_async(v_out, v0, nb, s)) return e;
  return trajectory_inplace(plan, beta, x_out, v_out, dir, rows, sumlogdet, p_accept, w, s);
}

// transition workspace = [X | V | p | dir] for R rows, then the trajectory workspace
static size_t transition_head_bytes(int64_t R, int D) {
  return 2 * align_up(sizeof(float) * (size_t)R * D, 256) + align_up(sizeof(float) * (size_t)R, 256) +
         align_up(sizeof(int) * (size_t)R, 256);
}

extern "C" size_t l2hmc_gauge_transition_ws_bytes(const l2hmc_gauge_plan* plan, int64_t B,
                                                  int32_t both_directions) {
  if (!plan || B < 0) return 0;
  const int64_t R = both_directions ? 2 * B : B;
  return transition_head_bytes(R, 2 * plan->T * plan->X) + carve_gauge_ws(plan, R, nullptr).bytes;
}

extern "C" int l2hmc_gauge_transition(const l2hmc_gauge_plan* plan, float beta, const float* x,
                                      const float* v0_f, const float* v0_b, const float* coin, const float* u,
                                      int64_t B, int32_t both_directions, float* x_prop, float* v_prop,
                                      float* p_accept, float* x_out, void* ws, size_t ws_bytes,
                                      l2hmc_stream_t stream) {
  if (int e = check_plan(plan)) return e;
  L2HMC_REQUIRE(B >= 0, "gauge_transition: B < 0");
  if (B == 0) return L2HMC_OK;
  L2HMC_REQUIRE(x && v0_f && v0_b && coin && u && x_prop && v_prop && p_accept && x_out && ws,
                "gauge_transition: NULL pointer");
  const int D = 2 * plan->T * plan->X;
  const int64_t R = both_directions ? 2 * B : B;
  const size_t need = l2hmc_gauge_transition_ws_bytes(plan, B, both_directions);
  if (ws_bytes < need) {
    set_error("gauge_transition: workspace %zu < %zu bytes", ws_bytes, need);
    return L2HMC_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  char* base = static_cast<char*>(ws);
  const size_t xv = align_up(sizeof(float) * (size_t)R * D, 256);
  float* Xw = reinterpret_cast<float*>(base);
  float* Vw = reinterpret_cast<float*>(base + xv);
  float* Pw = reinterpret_cast<float*>(base + 2 * xv);
  int* dirw = reinterpret_cast<int*>(base + 2 * xv + align_up(sizeof(float) * (size_t)R, 256));
  const GaugeWs w = carve_gauge_ws(plan, R, base + transition_head_bytes(R, D));
  const size_t nb = sizeof(float) * (size_t)B * D;

  if (both_directions) {
    // rows [0, B): forward with v0_f; rows [B, 2B): backward with v0_b (gauge_dynamics.py:211-218)
    if (int e = copy_async(Xw, x, nb, s)) return e;
    if (int e = copy_async(Xw + (size_t)B * D, x, nb, s)) return e;
    if (int e = copy_async(Vw, v0_f, nb, s)) return e;
    if (int e = copy_async(Vw + (size_t)B * D, v0_b, nb, s)) return e;
    hipLaunchKernelGGL(fill_dir_kernel, dim3((unsigned)ceil_div(2 * B, 256)), dim3(256), 0, s, dirw, B);
    L2HMC_CHECK_LAUNCH("fill_dir");
    if (int e = trajectory_inplace(plan, beta, Xw, Vw, dirw, R, nullptr, Pw, w, s, /*dir_split*/ B)) return e;
    hipLaunchKernelGGL(mix_accept_kernel, dim3((unsigned)ceil_div(B, 4)), dim3(256), 0, s, x, Xw, Vw, Pw,
                       Xw + (size_t)B * D, Vw + (size_t)B * D, Pw + B, coin, u, 1, B, D, x_prop, v_prop,
                       p_accept, x_out);
    L2HMC_CHECK_LAUNCH("mix_accept");
  } else {
    if (int e = copy_async(x_prop, x, nb, s)) return e;
    hipLaunchKernelGGL(select_dir_kernel, dim3((unsigned)ceil_div(B, 4)), dim3(256), 0, s, coin, v0_f, v0_b, B,
                       D, dirw, v_prop);
    L2HMC_CHECK_LAUNCH("select_dir");
    if (int e = trajectory_inplace(plan, beta, x_prop, v_prop, dirw, R, nullptr, p_accept, w, s)) return e;
    hipLaunchKernelGGL(accept_selected_kernel, dim3((unsigned)ceil_div(B, 4)), dim3(256), 0, s, x, x_prop,
                       p_accept, u, B, D, x_out);
    L2HMC_CHECK_LAUNCH("accept_selected");
  }
  return L2HMC_OK;
}
