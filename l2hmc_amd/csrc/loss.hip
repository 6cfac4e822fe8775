// Forward value of the lattice training loss, per chain: l2hmc/gauge_model.py:766-795 (`_calc_loss`) with
// `_create_metric_fn` (:632-657), `project_angle_approx` (:94-108, N=5 => n = 1..4) and
// `_calc_top_charges_diff(fft=True)` (:701-725).  One workgroup per chain; the three plaquette stencils
// (x, x_, z) and the two metric sums are reduced by fixed shuffle trees.  As in the reference, both
// auxiliary terms compare z with the proposal of x (x_), quirk Q9.
// The derivative of this loss through both accept probabilities is `loss_bwd_kernel` in train.hip.
#include "common.h"

namespace l2hmc {

__device__ __forceinline__ float metric_val(int kind, float a, float b) {
  switch (kind) {
    case 0: return fabsf(a - b);                                   // 'l1'
    case 1: return (a - b) * (a - b);                              // 'l2'
    case 2: return fabsf(cosf(a) - cosf(b));                       // 'cos'
    case 3: { const float d = cosf(a) - cosf(b); return d * d; }   // 'cos2'
    default: return 1.f - cosf(a - b);                             // 'cos_diff'
  }
}

// sum_{n=1}^{4} (-2/n)(-1)^n sin(n a)
__device__ __forceinline__ float project_angle_approx(float a) {
  return 2.f * sinf(a) - sinf(2.f * a) + (2.f / 3.f) * sinf(3.f * a) - 0.5f * sinf(4.f * a);
}

struct LossArgs {
  const float* x; const float* x_prop; const float* z; const float* px; const float* pz;
  int64_t B; int T, X, metric;
  float loss_scale, aux_weight, std_weight, charge_weight;
  float* terms;
};

__global__ __launch_bounds__(256) void gauge_loss_kernel(LossArgs p) {
  __shared__ float red[4][5];
  const int64_t b = blockIdx.x;
  const int sites = p.T * p.X, D = 2 * sites, X = p.X, T = p.T;
  const float* x = p.x + b * D;
  const float* xp = p.x_prop + b * D;
  const float* z = p.z + b * D;
  float m_x = 0.f, m_z = 0.f, q_x = 0.f, q_xp = 0.f, q_z = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) {
    m_x += metric_val(p.metric, x[d], xp[d]);
    m_z += metric_val(p.metric, z[d], xp[d]);
  }
  for (int s = threadIdx.x; s < sites; s += 256) {
    const int i = s / X, j = s - i * X;
    const int jp = (j + 1 == X) ? 0 : j + 1, ip = (i + 1 == T) ? 0 : i + 1;
    const int a0 = 2 * s, a1 = 2 * s + 1, a2 = 2 * (i * X + jp), a3 = 2 * (ip * X + j) + 1;
    q_x += project_angle_approx(x[a0] - x[a1] - x[a2] + x[a3]);
    q_xp += project_angle_approx(xp[a0] - xp[a1] - xp[a2] + xp[a3]);
    q_z += project_angle_approx(z[a0] - z[a1] - z[a2] + z[a3]);
  }
  float v[5] = {m_x, m_z, q_x, q_xp, q_z};
#pragma unroll
  for (int k = 0; k < 5; ++k) v[k] = wave_sum(v[k]);
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < 5; ++k) red[threadIdx.x >> 6][k] = v[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 0; k < 5; ++k) v[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    const float eps = 1e-3f, inv2pi = 0.15915494309189533577f;
    const float px = p.px[b], pz = p.pz[b], ls = p.loss_scale;
    const float x_std = v[0] * px + eps;
    const float z_std = p.aux_weight * (v[1] * pz + eps);
    const float std_loss = p.std_weight * (ls * (1.f / x_std + 1.f / z_std) - (x_std + z_std) / ls);
    const float xq = px * fabsf(v[2] * inv2pi - v[3] * inv2pi) + eps;
    const float zq = p.aux_weight * (pz * fabsf(v[4] * inv2pi - v[3] * inv2pi) + eps);
    p.terms[b] = std_loss + p.charge_weight * (xq + zq);
  }
}

}  // namespace l2hmc

using namespace l2hmc;

extern "C" int l2hmc_gauge_loss_terms(const float* x, const float* x_prop, const float* px, const float* z,
                                      const float* pz, int64_t B, int32_t T, int32_t X, int32_t metric,
                                      float loss_scale, float aux_weight, float std_weight, float charge_weight,
                                      float* terms, l2hmc_stream_t stream) {
  L2HMC_REQUIRE(B >= 0 && T > 0 && X > 0 && metric >= 0 && metric <= 4, "gauge_loss_terms: bad arguments");
  if (B == 0) return L2HMC_OK;
  L2HMC_REQUIRE(x && x_prop && px && z && pz && terms, "gauge_loss_terms: NULL pointer");
  L2HMC_REQUIRE(loss_scale != 0.f, "gauge_loss_terms: loss_scale must be non-zero");
  LossArgs a{x, x_prop, z, px, pz, B, T, X, metric, loss_scale, aux_weight, std_weight, charge_weight, terms};
  hipLaunchKernelGGL(gauge_loss_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, a);
  L2HMC_CHECK_LAUNCH("gauge_loss_terms");
  return L2HMC_OK;
}
