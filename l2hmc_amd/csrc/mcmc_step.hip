// One whole MCMC step on device-resident chains (the inference step harness of
// l2hmc/gauge_model.py:1371-1388 around dynamics/gauge_dynamics.py:195-259):
//   draw momenta / direction coin / MH uniform  ->  both trajectories  ->  mix, accept/reject,
//   per-step observables, wrap to [0, 2 pi).  Plans with a whole-trajectory kernel run the step in ONE launch
//   (launch_fused_step, fused_traj.hip: the kernel draws its own Philox streams and finishes the step in its
//   epilogue); other plans go through the public ops below with the draws of step_draws_kernel.
// The reference pays one session run with a host round trip of the whole batch per step.
#include "stq_dense.h"

namespace l2hmc {

// elements [0, nV) are standard normals of stream (seed, 2*draw), elements of `cu` uniforms of stream
// (seed, 2*draw+1): bit-identical to l2hmc_fill_normal / l2hmc_fill_uniform with those offsets.
__global__ __launch_bounds__(256) void step_draws_kernel(float* __restrict__ V, int64_t nV, float* __restrict__ cu,
                                                         int64_t ncu, uint64_t seed, uint64_t draw,
                                                         float* __restrict__ sums) {
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
  const int T = plan->T, X = plan->X, D = 2 * T * X;
  char* base = static_cast<char*>(ws);
  const size_t xv = align_up(sizeof(float) * (size_t)2 * B * D, 256);
  float* Xw = reinterpret_cast<float*>(base);
  float* Vw = reinterpret_cast<float*>(base + xv);
  float* Pw = reinterpret_cast<float*>(base + 2 * xv);
  float* cu = reinterpret_cast<float*>(base + 2 * xv + align_up(sizeof(float) * (size_t)2 * B, 256));
  char* rest = base + step_head_bytes(B, D);
  size_t rest_bytes = ws_bytes - step_head_bytes(B, D);

  const bool fused = !(plan->flags & L2HMC_PLAN_LAYERED) && fused_plan_supported(plan);
  const bool selected = (plan->flags & L2HMC_PLAN_SELECTED_ONLY) != 0;
  if (fused) {
    // ONE launch: the whole-trajectory kernel draws, integrates, mixes, accepts, measures and wraps (fused_traj.hip)
    return launch_fused_step(plan, beta, x, x_next, B, seed, draw, selected ? 0 : 1, px, actions, plaqs, charges,
                             charge_diff, step_sums, Xw /* 2 * workgroups floats of scratch */, s);
  }
  // momenta of both directions, coin | u  (tf.random_normal :269, tf.random_uniform :223,:246)
  const int64_t nblk = (((int64_t)2 * B * D + 3) >> 2) + ((2 * B + 3) >> 2);
  hipLaunchKernelGGL(step_draws_kernel, dim3((unsigned)hmin(ceil_div(nblk, 256), 4096)), dim3(256), 0, s, Vw,
                     (int64_t)2 * B * D, cu, 2 * B, seed, draw, nullptr);
  L2HMC_CHECK_LAUNCH("step_draws");

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

extern "C" int l2hmc_gauge_transition_draw(const l2hmc_gauge_plan* plan, float beta, const float* x, int64_t B,
                                           uint64_t seed, uint64_t draw, float* x_prop, float* v_prop,
                                           float* p_accept, float* x_out, void* ws, size_t ws_bytes,
                                           l2hmc_stream_t stream) {
  L2HMC_REQUIRE(plan != nullptr && B >= 0, "gauge_transition_draw: bad arguments");
  if (B == 0) return L2HMC_OK;
  L2HMC_REQUIRE(x && x_prop && v_prop && p_accept && x_out && ws, "gauge_transition_draw: NULL pointer");
  const size_t need = l2hmc_gauge_mcmc_step_ws_bytes(plan, B);
  if (ws_bytes < need) {
    set_error("gauge_transition_draw: workspace %zu < %zu bytes", ws_bytes, need);
    return L2HMC_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const int D = 2 * plan->T * plan->X;
  const bool selected = (plan->flags & L2HMC_PLAN_SELECTED_ONLY) != 0;
  if (!(plan->flags & L2HMC_PLAN_LAYERED) && fused_plan_supported(plan))
    return launch_fused_step(plan, beta, x, nullptr, B, seed, draw, selected ? 0 : 1, p_accept, nullptr, nullptr,
                             nullptr, nullptr, nullptr, nullptr, s, x_prop, v_prop, x_out);
  // other plans: the same draws into the workspace, then the public transition
  char* base = static_cast<char*>(ws);
  const size_t xv = align_up(sizeof(float) * (size_t)2 * B * D, 256);
  float* Vw = reinterpret_cast<float*>(base + xv);
  float* cu = reinterpret_cast<float*>(base + 2 * xv + align_up(sizeof(float) * (size_t)2 * B, 256));
  const int64_t nblk = (((int64_t)2 * B * D + 3) >> 2) + ((2 * B + 3) >> 2);
  hipLaunchKernelGGL(step_draws_kernel, dim3((unsigned)hmin(ceil_div(nblk, 256), 4096)), dim3(256), 0, s, Vw,
                     (int64_t)2 * B * D, cu, 2 * B, seed, draw, nullptr);
  L2HMC_CHECK_LAUNCH("step_draws");
  char* rest = base + step_head_bytes(B, D);
  return l2hmc_gauge_transition(plan, beta, x, Vw, Vw + (size_t)B * D, cu, cu + B, B, selected ? 0 : 1, x_prop, v_prop,
                                p_accept, x_out, rest, ws_bytes - step_head_bytes(B, D), stream);
}
