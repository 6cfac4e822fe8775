/* Pure-C host program on the C ABI of libl2hmc_hip.so -- no Python, no PyTorch: the boundary a maintainer binds
 * (cgo / JNI / ctypes alike) is plain pointers and sizes.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ examples/c_abi_demo.c -Iinclude -I/opt/rocm/include \
 *       -Ll2hmc_amd -l:libl2hmc_hip.so -L/opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,$PWD/l2hmc_amd -Wl,-rpath,/opt/rocm/lib -o c_abi_demo && ./c_abi_demo
 * (plain C99 with gcc: the header needs no C++ and no HIP compiler; the HIP runtime is only used here to allocate
 * and copy device memory)
 *
 * 1. U(1) action of a cold start (all links 0) must be 0, its plaquette 1 (gauge_model_graph_mode.ipynb check);
 * 2. a GenericNet-shaped sampler with small random weights runs MCMC steps through l2hmc_gauge_mcmc_step and
 *    the mean accept probability comes back in (0, 1].
 * Exit code 0 on success. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "l2hmc_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK(x) do { int rc_ = (x); if (rc_ != L2HMC_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, l2hmc_last_error()); return 3; } } while (0)

static float* dev_floats(size_t n) {
  float* p = NULL;
  if (hipMalloc((void**)&p, n * sizeof(float)) != hipSuccess) return NULL;
  hipMemset(p, 0, n * sizeof(float));
  return p;
}

/* N(0, std^2) weights straight on the device: fill_normal then an in-place scale done by a tiny host round trip
 * (this demo is not about speed) */
static int dev_normal(float* p, size_t n, float std, uint64_t seed) {
  CHECK(l2hmc_fill_normal(p, (int64_t)n, seed, 0, NULL));
  float* h = (float*)malloc(n * sizeof(float));
  CHECK_HIP(hipMemcpy(h, p, n * sizeof(float), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n; ++i) h[i] *= std;
  CHECK_HIP(hipMemcpy(p, h, n * sizeof(float), hipMemcpyHostToDevice));
  free(h);
  return 0;
}

static int make_net(l2hmc_dense_net* n, int D, int H, uint64_t seed) {
  n->D = D; n->H = H; n->Ka = D; n->Kb = D; n->q_tanh = 0; n->reserved = 0; n->packed = NULL;
  float *w1 = dev_floats((size_t)H * 2 * D), *wt = dev_floats(2 * (size_t)H), *b1 = dev_floats(H);
  float *wh = dev_floats((size_t)H * H), *bh = dev_floats(H), *whd = dev_floats((size_t)3 * D * H);
  float *bhd = dev_floats(3 * (size_t)D), *cs = dev_floats(D), *cq = dev_floats(D);
  if (!w1 || !wt || !b1 || !wh || !bh || !whd || !bhd || !cs || !cq) return 2;
  if (dev_normal(w1, (size_t)H * 2 * D, sqrtf(2.6f / 3.f / D), seed + 1)) return 3;
  if (dev_normal(wt, 2 * (size_t)H, sqrtf(2.6f / 3.f / 2), seed + 2)) return 3;
  if (dev_normal(wh, (size_t)H * H, sqrtf(2.6f / H), seed + 3)) return 3;
  if (dev_normal(whd, (size_t)3 * D * H, sqrtf(2.6f * 0.001f / H), seed + 4)) return 3;
  n->w1_t = w1; n->wt = wt; n->b1 = b1; n->wh_t = wh; n->bh = bh; n->whd_t = whd; n->bhd = bhd;
  n->coeff_s = cs; n->coeff_q = cq;
  size_t pb = l2hmc_dense_pack_bytes(n);     /* fused whole-trajectory kernel image, if this shape has one */
  if (pb) {
    float* pk = dev_floats(pb / sizeof(float));
    if (!pk) return 2;
    CHECK(l2hmc_dense_pack(n, pk, NULL));
    n->packed = pk;
  }
  return 0;
}

int main(void) {
  const int T = 8, X = 8, D = 2 * T * X, H = 4 * D, NLF = 5;
  const int64_t B = 256;
  if (l2hmc_abi_version() != L2HMC_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }

  /* ---- 1. cold start: action 0, plaquette 1, charge 0 */
  float* x = dev_floats((size_t)B * D);
  float *act = dev_floats(B), *plq = dev_floats(B), *chg = dev_floats(B);
  if (!x || !act || !plq || !chg) return 2;
  CHECK(l2hmc_u1_action_force(x, B, T, X, 2.0f, act, NULL, plq, chg, NULL));
  float h3[3];
  CHECK_HIP(hipMemcpy(&h3[0], act, sizeof(float), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(&h3[1], plq, sizeof(float), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(&h3[2], chg, sizeof(float), hipMemcpyDeviceToHost));
  printf("cold start: action %.6f  avg_plaq %.6f  top_charge %.6f\n", h3[0], h3[1], h3[2]);
  if (fabsf(h3[0]) > 1e-6f || fabsf(h3[1] - 1.f) > 1e-6f || fabsf(h3[2]) > 1e-6f) return 4;

  /* ---- 2. a sampler: masks, two networks, MCMC steps on device-resident chains */
  float* hm = (float*)calloc((size_t)NLF * D, sizeof(float));
  srand(42);
  for (int s = 0; s < NLF; ++s) {            /* D/2 ones per step (gauge_dynamics.py:651-661) */
    int placed = 0;
    while (placed < D / 2) {
      int i = rand() % D;
      if (hm[s * D + i] == 0.f) { hm[s * D + i] = 1.f; ++placed; }
    }
  }
  float* masks = dev_floats((size_t)NLF * D);
  CHECK_HIP(hipMemcpy(masks, hm, (size_t)NLF * D * sizeof(float), hipMemcpyHostToDevice));
  free(hm);
  l2hmc_gauge_plan plan;
  memset(&plan, 0, sizeof(plan));
  plan.T = T; plan.X = X; plan.num_steps = NLF; plan.hmc = 0; plan.eps = 0.1f; plan.flags = 0; plan.masks = masks;
  if (make_net(&plan.xnet, D, H, 100) || make_net(&plan.vnet, D, H, 200)) return 5;

  CHECK(l2hmc_fill_uniform(x, B * D, 7, 0, NULL));          /* hot start in [0, 1) is enough for the demo */
  size_t wsb = l2hmc_gauge_mcmc_step_ws_bytes(&plan, B);
  void* ws = NULL;
  CHECK_HIP(hipMalloc(&ws, wsb));
  float* px = dev_floats(B);
  float* hpx = (float*)malloc(B * sizeof(float));
  double mean = 0.0;
  const int steps = 20;
  for (int s = 0; s < steps; ++s) {
    CHECK(l2hmc_gauge_mcmc_step(&plan, 2.0f, x, B, /*seed*/ 42, /*draw*/ (uint64_t)s, px, act, plq, chg, NULL, ws, wsb,
                                NULL));
    CHECK_HIP(hipMemcpy(hpx, px, B * sizeof(float), hipMemcpyDeviceToHost));   /* synchronises */
    double m = 0.0;
    for (int64_t i = 0; i < B; ++i) m += hpx[i];
    mean += m / B / steps;
  }
  CHECK_HIP(hipMemcpy(hpx, plq, B * sizeof(float), hipMemcpyDeviceToHost));
  double pl = 0.0;
  for (int64_t i = 0; i < B; ++i) pl += hpx[i] / B;
  printf("sampler: %d MCMC steps of %lld chains (fused kernel: %s), mean accept %.4f, <plaq> of the last inputs %.4f\n",
         steps, (long long)B, plan.xnet.packed ? "yes" : "no", mean, pl);
  if (!(mean > 0.0 && mean <= 1.0) || !(pl > -1.0 && pl < 1.0)) return 6;

  /* ---- 3. argument checking happens on the host: a bad shape is refused, nothing is launched */
  int rc = l2hmc_u1_action_force(x, B, 0, X, 2.0f, act, NULL, NULL, NULL, NULL);
  printf("bad lattice extent -> rc %d (\"%s\")\n", rc, l2hmc_last_error());
  if (rc != L2HMC_ERR_ARG) return 7;
  puts("c_abi_demo OK");
  return 0;
}
