// One whole MCMC step on device-resident chains (the inference step harness of
// l2hmc/gauge_model.py:1371-1388 around dynamics/gauge_dynamics.py:195-259):
//   draw momenta / direction coin / MH uniform  ->  both trajectories  ->  mix, accept/reject,
//   per-step observables, wrap to [0, 2 pi), all in three launches when the plan has a fused kernel:
//     step_draws_kernel     Philox: V = [v0_f; v0_b] straight into the stacked trajectory layout, coin | u
//     gauge_traj_fused      rows [0,B) forward, [B,2B) backward, both reading x[r % B] (no copies)
//     finish_step_kernel    one lattice site per thread: mix + MH + observables of input and output + wrap
// The reference pays one session run with a host round trip of the whole batch per step.
#include "stq_dense.h"

namespace l2hmc {

constexpr float kTwoPiF = 6.28318530717958647692f;
constexpr float kPiF = 3.14159265358979323846f;

// elements [0, nV) are standard normals of stream (seed, 2*draw), elements of `cu` uniforms of stream
// (seed, 2*draw+1): bit-identical to l2hmc_fill_normal / l2hmc_fill_uniform with those offsets.
__global__ __launch_bounds__(256) void step_draws_kernel(float* __restrict__ V, int64_t nV, float* __restrict__ cu,
                                                         int64_t ncu, uint64_t seed, uint64_t draw,
                                                         float* __restrict__ sums) {
  if (sums && blockIdx.x == 0 && threadIdx.x == 0) sums[3] = 0.f;      // ticket counter of finish_step_kernel
  const int64_t nbV = (nV + 3) >> 2, nbU = (ncu + 3) >> 2;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nbV + nbU;
       b += (int64_t)gridDim.x * blockDim.x) {
    const bool normal = b < nbV;
    const int64_t blk = normal ? b : b - nbV;
    const uint64_t off = 2 * draw + (normal ? 0 : 1);
    uint32_t c[4] = {(uint32_t)blk, (uint32_t)((uint64_t)blk >> 32), (uint32_t)off, (uint32_t)(off >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    float v[4];
    if (normal) {
      philox_normal4(c, v);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (float)(c[j] >> 8) * (1.0f / 16777216.0f);
    }
    float* out = normal ? V : cu;
    const int64_t n = normal ? nV : ncu, i0 = blk << 2;
    for (int j = 0; j < 4 && i0 + j < n; ++j) out[i0 + j] = v[j];
  }
}

// Selected-direction variant (L2HMC_PLAN_SELECTED_ONLY): chain b needs only the momentum of the direction its coin
// picks.  Philox is counter based, so exactly those blocks of the SAME streams are generated: V[b] = elements
// [(sel*B + b)*D, +D) of the stacked normal stream, with sel = 0 (forward) / 1 (backward) from the coin.
// D % 4 == 0.  Also writes coin | u (as step_draws_kernel) and dir[b].
__global__ __launch_bounds__(256) void step_draws_selected_kernel(float* __restrict__ V, float* __restrict__ cu,
                                                                  int* __restrict__ dir, int64_t B, int D,
                                                                  uint64_t seed, uint64_t draw,
                                                                  float* __restrict__ sums) {
  if (sums && blockIdx.x == 0 && threadIdx.x == 0) sums[3] = 0.f;
  const int d4n = D >> 2;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < B * d4n; w += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = w / d4n;
    const int d4 = (int)(w - b * d4n);
    // coin[b] = uniform element b of stream (seed, 2*draw+1)
    const uint64_t offu = 2 * draw + 1;
    const int64_t cb = b >> 2;
    uint32_t c[4] = {(uint32_t)cb, (uint32_t)((uint64_t)cb >> 32), (uint32_t)offu, (uint32_t)(offu >> 32)};
    philox4x32_10(c, k0, k1);
    const float coin = (float)(c[b & 3] >> 8) * (1.0f / 16777216.0f);
    const int sel = coin > 0.5f ? 0 : 1;                       // gauge_dynamics.py:221-227: coin > 0.5 -> forward
    if (d4 == 0) {
      cu[b] = coin;
      const int64_t ub = (B + b) >> 2;
      uint32_t cu4[4] = {(uint32_t)ub, (uint32_t)((uint64_t)ub >> 32), (uint32_t)offu, (uint32_t)(offu >> 32)};
      philox4x32_10(cu4, k0, k1);
      cu[B + b] = (float)(cu4[(B + b) & 3] >> 8) * (1.0f / 16777216.0f);
      dir[b] = sel;
    }
    const uint64_t offn = 2 * draw;
    const int64_t nb = ((int64_t)(sel ? B + b : b) * D + 4 * d4) >> 2;
    uint32_t cn[4] = {(uint32_t)nb, (uint32_t)((uint64_t)nb >> 32), (uint32_t)offn, (uint32_t)(offn >> 32)};
    philox4x32_10(cn, k0, k1);
    float v[4];
    philox_normal4(cn, v);
    float* out = V + b * D + 4 * d4;
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = v[j];
  }
}

// sites % 64 == 0 and 256 % sites == 0.  Xw/Pw: rows [0,B) forward, [boff, boff+B) backward results
// (boff = B; boff = 0 when only the selected direction was integrated into rows [0,B)).
__global__ __launch_bounds__(256) void finish_step_kernel(const float* x, float* x_next, float* __restrict__ sums,
                                                          float* __restrict__ part, const float* __restrict__ Xw,
                                                          const float* __restrict__ Pw, const float* __restrict__ cu,
                                                          int64_t B, int64_t boff, int T, int X, int cpw,
                                                          float* __restrict__ px,
                                                          float* __restrict__ actions, float* __restrict__ plaqs,
                                                          float* __restrict__ charges, float* __restrict__ dq) {
  __shared__ float2 xs[256];
  __shared__ float red[4][4];
  const int sites = T * X;
  const int tid = threadIdx.x;
  const int c = tid / sites, site = tid - c * sites;
  const int i = site / X, j = site - i * X;
  const int base = c * sites;
  const int n_jp = base + i * X + ((j + 1 == X) ? 0 : j + 1);
  const int n_ip = base + ((i + 1 == T) ? 0 : i + 1) * X + j;
  const int64_t row = (int64_t)blockIdx.x * cpw + c;
  const bool live = row < B;
  float2 xin = make_float2(0.f, 0.f), xo = xin;
  float p = 0.f;
  if (live) {
    // gauge_dynamics.py:221-257, arithmetic kept as mask * a + (1 - mask) * b
    const float fm = cu[row] > 0.5f ? 1.f : 0.f, bm = 1.f - fm;
    p = fm * Pw[row] + bm * Pw[boff + row];
    const float am = p > cu[B + row] ? 1.f : 0.f;
    const float2* x2 = reinterpret_cast<const float2*>(x);
    const float2* w2 = reinterpret_cast<const float2*>(Xw);
    xin = x2[row * sites + site];
    const float2 xf = w2[row * sites + site], xb = w2[(boff + row) * sites + site];
    const float xp0 = fm * xf.x + bm * xb.x, xp1 = fm * xf.y + bm * xb.y;
    xo.x = am * xp0 + (1.f - am) * xin.x;
    xo.y = am * xp1 + (1.f - am) * xin.y;
  }
  const float inv2pi = 0.15915494309189533577f;
  // observables of the step's INPUT samples (gauge_model.py:256-266) ...
  xs[tid] = xin;
  __syncthreads();
  const float Pin = xin.x - xin.y - xs[n_jp].x + xs[n_ip].y;
  float sn, cs;
  fast_sincos(Pin, &sn, &cs);
  float a = wave_sum(1.f - cs), q = wave_sum(cs), ch = wave_sum(Pin - kTwoPiF * floorf((Pin + kPiF) * inv2pi));
  __syncthreads();
  // ... and the topological charge of the output for |dQ| (:718-725)
  xs[tid] = xo;
  __syncthreads();
  const float Pout = xo.x - xo.y - xs[n_jp].x + xs[n_ip].y;
  float cho = wave_sum(Pout - kTwoPiF * floorf((Pout + kPiF) * inv2pi));
  if (sites != kWave) {
    if ((tid & 63) == 0) {
      red[tid >> 6][0] = a;
      red[tid >> 6][1] = q;
      red[tid >> 6][2] = ch;
      red[tid >> 6][3] = cho;
    }
    __syncthreads();
    if (site == 0) {
      a = q = ch = cho = 0.f;
      for (int w = base / kWave; w < (base + sites) / kWave; ++w) {
        a += red[w][0];
        q += red[w][1];
        ch += red[w][2];
        cho += red[w][3];
      }
    }
  }
  if (live && site == 0) {
    if (px) px[row] = p;
    if (actions) actions[row] = a;
    if (plaqs) plaqs[row] = q / (float)sites;
    if (charges) charges[row] = ch * inv2pi;
    if (dq) dq[row] = fabsf(ch * inv2pi - cho * inv2pi);
  }
  if (sums) {
    // [sum p_accept, sum |dQ|, chains] for dist.StepStats, in a fixed order and without a further launch: every
    // workgroup leaves its partial sums in `part`, the last one to arrive (ticket in sums[3]) adds them up
    __shared__ float bs[2][4];
    __shared__ float fin[2][256];
    __shared__ int last;
    if (site == 0) {
      bs[0][c] = live ? p : 0.f;
      bs[1][c] = live ? fabsf(ch * inv2pi - cho * inv2pi) : 0.f;
    }
    __syncthreads();
    if (tid == 0) {
      float a0 = 0.f, a1 = 0.f;
      for (int k = 0; k < cpw; ++k) {
        a0 += bs[0][k];
        a1 += bs[1][k];
      }
      part[2 * blockIdx.x] = a0;
      part[2 * blockIdx.x + 1] = a1;
      __threadfence();
      last = atomicAdd(reinterpret_cast<int*>(sums + 3), 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (last) {
      __threadfence();
      float a0 = 0.f, a1 = 0.f;
      for (int b = tid; b < (int)gridDim.x; b += 256) {
        a0 += part[2 * b];
        a1 += part[2 * b + 1];
      }
      fin[0][tid] = a0;
      fin[1][tid] = a1;
      __syncthreads();
      for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) {
          fin[0][tid] += fin[0][tid + st];
          fin[1][tid] += fin[1][tid + st];
        }
        __syncthreads();
      }
      if (tid == 0) {
        sums[0] = fin[0][0];
        sums[1] = fin[1][0];
        sums[2] = (float)B;
      }
    }
  }
  if (live) {
    float2 w;                                     // np.mod(x_out, 2 pi), gauge_model.py:1388
    w.x = fmodf(xo.x, kTwoPiF);
    w.y = fmodf(xo.y, kTwoPiF);
    if (w.x < 0.f) w.x += kTwoPiF;
    if (w.y < 0.f) w.y += kTwoPiF;
    reinterpret_cast<float2*>(x_next)[row * sites + site] = w;
  }
}

__global__ void charge_diff_kernel(const float* __restrict__ q_in, const float* __restrict__ q_out, int64_t B,
                                   float* __restrict__ charges, float* __restrict__ dq) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  if (charges) charges[i] = q_in[i];
  if (dq) dq[i] = fabsf(q_in[i] - q_out[i]);
}

// general path: [sum p, sum |dQ|, B] from the per-chain arrays, one workgroup, fixed order
__global__ __launch_bounds__(256) void step_sums_kernel(const float* __restrict__ px, const float* __restrict__ dq,
                                                        int64_t B, float* __restrict__ sums) {
  __shared__ float red[2][4];
  float a = 0.f, b = 0.f;
  for (int64_t i = threadIdx.x; i < B; i += 256) {
    a += px[i];
    b += dq[i];
  }
  a = wave_sum(a);
  b = wave_sum(b);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = a;
    red[1][threadIdx.x >> 6] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    sums[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    sums[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    sums[2] = (float)B;
  }
}

}  // namespace l2hmc

using namespace l2hmc;

static size_t step_head_bytes(int64_t B, int D) {
  return 2 * align_up(sizeof(float) * (size_t)2 * B * D, 256) + align_up(sizeof(float) * (size_t)2 * B, 256) +
         align_up(sizeof(float) * (size_t)2 * B, 256);
}

extern "C" size_t l2hmc_gauge_mcmc_step_ws_bytes(const l2hmc_gauge_plan* plan, int64_t B) {
  if (!plan || B < 0) return 0;
  return step_head_bytes(B, 2 * plan->T * plan->X) + l2hmc_gauge_transition_ws_bytes(plan, B, 1) +
         3 * align_up(sizeof(float) * (size_t)B * 2 * plan->T * plan->X, 256);
}

extern "C" int l2hmc_gauge_mcmc_step(const l2hmc_gauge_plan* plan, float beta, float* x, int64_t B, uint64_t seed,
                                     uint64_t draw, float* px, float* actions, float* plaqs, float* charges,
                                     float* charge_diff, void* ws, size_t ws_bytes, l2hmc_stream_t stream) {
  return l2hmc_gauge_mcmc_step_ex(plan, beta, x, x, B, seed, draw, px, actions, plaqs, charges, charge_diff, nullptr,
                                  ws, ws_bytes, stream);
}

extern "C" int l2hmc_gauge_mcmc_step_ex(const l2hmc_gauge_plan* plan, float beta, const float* x_in, float* x_next,
                                        int64_t B, uint64_t seed, uint64_t draw, float* px, float* actions,
                                        float* plaqs, float* charges, float* charge_diff, float* step_sums, void* ws,
                                        size_t ws_bytes, l2hmc_stream_t stream) {
  L2HMC_REQUIRE(plan != nullptr && B >= 0, "gauge_mcmc_step: bad arguments");
  if (B == 0) return L2HMC_OK;
  L2HMC_REQUIRE(x_in && x_next && ws, "gauge_mcmc_step: NULL pointer");
  const float* x = x_in;
  const size_t need = l2hmc_gauge_mcmc_step_ws_bytes(plan, B);
  if (ws_bytes < need) {
    set_error("gauge_mcmc_step: workspace %zu < %zu bytes", ws_bytes, need);
    return L2HMC_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const int T = plan->T, X = plan->X, D = 2 * T * X, sites = T * X;
  char* base = static_cast<char*>(ws);
  const size_t xv = align_up(sizeof(float) * (size_t)2 * B * D, 256);
  float* Xw = reinterpret_cast<float*>(base);
  float* Vw = reinterpret_cast<float*>(base + xv);
  float* Pw = reinterpret_cast<float*>(base + 2 * xv);
  float* cu = reinterpret_cast<float*>(base + 2 * xv + align_up(sizeof(float) * (size_t)2 * B, 256));
  char* rest = base + step_head_bytes(B, D);
  size_t rest_bytes = ws_bytes - step_head_bytes(B, D);

  const bool fused = !(plan->flags & L2HMC_PLAN_LAYERED) && fused_plan_supported(plan);
  const bool fast_finish = sites % kWave == 0 && 256 % sites == 0;
  const bool selected = (plan->flags & L2HMC_PLAN_SELECTED_ONLY) != 0;
  if (fused && fast_finish && selected) {
    // half the rows: the momentum of the chosen direction only (same Philox streams), per-row direction from the coin
    int* dirs = reinterpret_cast<int*>(Pw + B);                     // second half of the [2B] accept buffer
    hipLaunchKernelGGL(step_draws_selected_kernel, dim3((unsigned)hmin(ceil_div(B * (D >> 2), 256), 4096)), dim3(256),
                       0, s, Vw, cu, dirs, B, D, seed, draw, step_sums);
    L2HMC_CHECK_LAUNCH("step_draws_selected");
    if (int e = launch_fused_trajectory(plan, beta, 0, plan->num_steps, x, Vw, dirs, B, Xw, Vw, nullptr, 0, Pw, s))
      return e;
    const int cpw = 256 / sites;
    hipLaunchKernelGGL(finish_step_kernel, dim3((unsigned)ceil_div(B, cpw)), dim3(256), 0, s, x, x_next, step_sums, Vw,
                       Xw, Pw, cu, B, (int64_t)0, T, X, cpw, px, actions, plaqs, charges, charge_diff);
    L2HMC_CHECK_LAUNCH("finish_step");
    return L2HMC_OK;
  }
  // momenta of both directions, coin | u  (tf.random_normal :269, tf.random_uniform :223,:246)
  const int64_t nblk = (((int64_t)2 * B * D + 3) >> 2) + ((2 * B + 3) >> 2);
  hipLaunchKernelGGL(step_draws_kernel, dim3((unsigned)hmin(ceil_div(nblk, 256), 4096)), dim3(256), 0, s, Vw,
                     (int64_t)2 * B * D, cu, 2 * B, seed, draw, step_sums);
  L2HMC_CHECK_LAUNCH("step_draws");

  if (fused && fast_finish) {
    if (int e = launch_fused_trajectory(plan, beta, 0, plan->num_steps, x, Vw, nullptr, 2 * B, Xw, Vw, nullptr, 0, Pw,
                                        s, /*x_mod=*/B, /*dir_split=*/B))
      return e;
    const int cpw = 256 / sites;
    hipLaunchKernelGGL(finish_step_kernel, dim3((unsigned)ceil_div(B, cpw)), dim3(256), 0, s, x, x_next, step_sums, Vw,
                       Xw, Pw, cu, B, B, T, X, cpw, px, actions, plaqs, charges, charge_diff);
    L2HMC_CHECK_LAUNCH("finish_step");
    return L2HMC_OK;
  }
  // general path (no fused kernel for this plan, or an odd lattice): the same step through the public ops
  const size_t bd = align_up(sizeof(float) * (size_t)B * D, 256);
  float* x_prop = reinterpret_cast<float*>(rest);
  float* v_prop = reinterpret_cast<float*>(rest + bd);
  float* x_out = reinterpret_cast<float*>(rest + 2 * bd);
  rest += 3 * bd;
  rest_bytes -= 3 * bd;
  L2HMC_REQUIRE(!step_sums || (px && charge_diff), "gauge_mcmc_step: step_sums needs px and charge_diff on this path");
  float* pp = px ? px : Xw;        // Xw / Pw of the head are free here: the transition carves its own copies
  if (int e = l2hmc_gauge_transition(plan, beta, x, Vw, Vw + (size_t)B * D, cu, cu + B, B, selected ? 0 : 1, x_prop,
                                     v_prop, pp,
                                     x_out, rest, rest_bytes, stream))
    return e;
  float* q_in = Pw;
  float* q_out = Pw + B;
  if (int e = l2hmc_u1_action_force(x, B, T, X, beta, actions, nullptr, plaqs, q_in, stream)) return e;
  if (int e = l2hmc_u1_action_force(x_out, B, T, X, beta, nullptr, nullptr, nullptr, q_out, stream)) return e;
  hipLaunchKernelGGL(charge_diff_kernel, dim3((unsigned)ceil_div(B, 256)), dim3(256), 0, s, q_in, q_out, B, charges,
                     charge_diff);
  L2HMC_CHECK_LAUNCH("charge_diff");
  if (step_sums) {
    hipLaunchKernelGGL(step_sums_kernel, dim3(1), dim3(256), 0, s, px, charge_diff, B, step_sums);
    L2HMC_CHECK_LAUNCH("step_sums");
  }
  return l2hmc_wrap_angle(x_out, (int64_t)B * D, x_next, stream);
}
