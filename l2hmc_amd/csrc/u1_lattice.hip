// 2D U(1) lattice kernels for gfx950: plaquette sums, action, force, observables.
//
// Replaces the ~9 stock TF ops (4 strided slices, 2 rolls, cos, reduce, and the
// autodiff graph twice that size) of
//   l2hmc/lattice/lattice.py:337-362, :285-313
//   l2hmc/gauge_model.py:659-725
//   l2hmc/dynamics/gauge_dynamics.py:698-709 (force = autodiff of beta*S)
// with one pass: a chain's links are read once (coalesced), its plaquette
// neighbourhood is staged in LDS with the periodic wrap resolved there, sin P is
// shared through LDS for the two links that need each plaquette, and the
// per-chain sums use a fixed shuffle tree (bit-reproducible).
//
// HBM-bound: 8*D + 4..12 bytes per chain (read x, write force + scalars).
#include "common.h"

namespace l2hmc {

constexpr int kLatThreads = 256;
constexpr float kTwoPi = 6.28318530717958647692f;
constexpr float kPi = 3.14159265358979323846f;

// One workgroup handles `cpw` consecutive chains (cpw*sites <= 256 threads busy)
// or, for sites >= 256, one chain with several sites per thread.
__global__ __launch_bounds__(kLatThreads) void u1_action_force_kernel(
    const float* __restrict__ x, int64_t rows, int T, int X, float beta, int cpw,
    float* __restrict__ action, float* __restrict__ force, float* __restrict__ avg_plaq,
    float* __restrict__ top_charge) {
  extern __shared__ float lds[];
  const int sites = T * X;
  const int D = 2 * sites;
  float* xs = lds;                 // [cpw][sites][2]
  float* sp = lds + cpw * D;       // [cpw][sites]   sin P
  float* red = sp + cpw * sites;   // [256][3] or [4][3] partials

  const int tid = threadIdx.x;
  const int64_t row0 = (int64_t)blockIdx.x * cpw;
  const int nrow = (int)min((int64_t)cpw, rows - row0);
  const int nflt = nrow * D;
  const float* src = x + row0 * D;

  // coalesced copy of nrow*D contiguous floats
  if ((D & 3) == 0) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(xs);
    for (int i = tid; i < nflt / 4; i += kLatThreads) d4[i] = s4[i];
  } else {
    for (int i = tid; i < nflt; i += kLatThreads) xs[i] = src[i];
  }
  __syncthreads();

  const int work = nrow * sites;
  // index arithmetic without run-time divisions where the shape allows it: one chain per workgroup (sites >= 256)
  // has c = 0, and a power-of-two X turns site / X into a shift
  const bool one_chain = cpw == 1;
  const int xsh = (X & (X - 1)) == 0 ? 31 - __clz(X) : -1;
  float a_act = 0.f, a_plq = 0.f, a_chg = 0.f;
  for (int s = tid; s < work; s += kLatThreads) {
    const int c = one_chain ? 0 : s / sites;
    const int site = s - c * sites;
    const int i = xsh >= 0 ? site >> xsh : site / X, j = site - i * X;
    const float* xc = xs + c * D;
    const int jp = (j + 1 == X) ? 0 : j + 1;
    const int ip = (i + 1 == T) ? 0 : i + 1;
    // gauge_model.py:676-679: x0[i,j] - x1[i,j] - x0[i,j+1] + x1[i+1,j]
    const float P = xc[2 * site] - xc[2 * site + 1] - xc[2 * (i * X + jp)] + xc[2 * (ip * X + j) + 1];
    float sn, cs;
    fast_sincos(P, &sn, &cs);
    sp[s] = sn;
    a_act += 1.f - cs;
    a_plq += cs;
    a_chg += P - kTwoPi * floorf((P + kPi) / kTwoPi);   // project_angle, gauge_model.py:78-80
  }

  // per-chain reduction, deterministic
  const bool wave_aligned = (sites % kWave) == 0;
  if (wave_aligned) {
    a_act = wave_sum(a_act);
    a_plq = wave_sum(a_plq);
    a_chg = wave_sum(a_chg);
    const int w = tid >> 6;
    if ((tid & 63) == 0) {
      red[w * 3 + 0] = a_act;
      red[w * 3 + 1] = a_plq;
      red[w * 3 + 2] = a_chg;
    }
  } else {
    red[tid * 3 + 0] = a_act;
    red[tid * 3 + 1] = a_plq;
    red[tid * 3 + 2] = a_chg;
  }
  __syncthreads();
  if (tid < nrow) {
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
    int lo, hi;
    if (wave_aligned) {
      const int wpc = (sites >= kLatThreads) ? (kLatThreads / kWave) : (sites / kWave);
      lo = tid * wpc;
      hi = lo + wpc;
    } else if (sites >= kLatThreads) {
      lo = 0;
      hi = kLatThreads;
    } else {
      lo = tid * sites;
      hi = lo + sites;
    }
    for (int k = lo; k < hi; ++k) {
      r0 += red[k * 3 + 0];
      r1 += red[k * 3 + 1];
      r2 += red[k * 3 + 2];
    }
    const int64_t r = row0 + tid;
    if (action) action[r] = r0;
    if (avg_plaq) avg_plaq[r] = r1 / (float)sites;
    if (top_charge) top_charge[r] = r2 / kTwoPi;
  }

  if (force) {
    float2* dst = reinterpret_cast<float2*>(force + row0 * D);
    for (int s = tid; s < work; s += kLatThreads) {
      const int c = one_chain ? 0 : s / sites;
      const int site = s - c * sites;
      const int i = xsh >= 0 ? site >> xsh : site / X, j = site - i * X;
      const float* spc = sp + c * sites;
      const int jm = (j == 0) ? X - 1 : j - 1;
      const int im = (i == 0) ? T - 1 : i - 1;
      const float sP = spc[site];
      float2 g;
      g.x = beta * (sP - spc[i * X + jm]);       // d/dx0[i,j]
      g.y = beta * (-sP + spc[im * X + j]);      // d/dx1[i,j]
      dst[s] = g;
    }
  }
}

// Fast path for even X with 64 ... 1024 sites per chain: every thread owns the two sites (i, j), (i, j + 1) of a row
// -- ONE 16-byte load and ONE 16-byte store per thread -- and every workgroup exactly one group of chains (one-shot
// grid, no persistence).  tools/stream_shape_bench.hip measured the shapes on MI355X for this read-once /
// write-once stream with an LDS neighbour exchange and two barriers: 8 bytes per thread 5.4-5.5 TB/s (persistent
// or one-shot; the previous form of this kernel), 16 bytes per thread one-shot 6.4 TB/s -- the same as an
// element-wise kernel without any exchange (profiles/r02_stream_shape_bench.txt).
// Thread l of a chain: row i = l / (X/2), column pair jh = l % (X/2).
//   P[i,j] = x0[i,j] - x1[i,j] - x0[i,j+1] + x1[i+1,j];  dS/dx0[i,j] = beta (sin P[i,j] - sin P[i,j-1]),
//   dS/dx1[i,j] = beta (-sin P[i,j] + sin P[i-1,j])   (lattice.py:285-362, gauge_dynamics.py:592-609)
template <bool SCALARS, int THREADS>
__global__ __launch_bounds__(THREADS) void u1_pair_kernel(
    const float* __restrict__ x, int64_t rows, int T, int X, float beta, int cpw,
    float* __restrict__ action, float* __restrict__ force, float* __restrict__ avg_plaq,
    float* __restrict__ top_charge) {
  __shared__ float2 x1s[THREADS];            // (x1[i,j], x1[i,j+1])
  __shared__ float x0s[THREADS];             // x0[i,j]
  __shared__ float2 sps[THREADS];            // (sin P[i,j], sin P[i,j+1])
  __shared__ float red[THREADS / kWave][2];
  const int sites = T * X, lpc = sites >> 1, X2 = X >> 1;
  const int tid = threadIdx.x;
  // lpc is a power of two on this path; X / 2 usually is (shift instead of the ~30-instruction integer division)
  const int lsh = 31 - __builtin_clz(lpc);
  const int c = tid >> lsh, l = tid & (lpc - 1);
  const int i = (X2 & (X2 - 1)) == 0 ? l >> (31 - __builtin_clz(X2)) : l / X2, jh = l - i * X2;
  const int base = c * lpc;
  const int n_right = base + i * X2 + ((jh + 1 == X2) ? 0 : jh + 1);
  const int n_left = base + i * X2 + ((jh == 0) ? X2 - 1 : jh - 1);
  const int n_up = base + ((i + 1 == T) ? 0 : i + 1) * X2 + jh;
  const int n_down = base + ((i == 0) ? T - 1 : i - 1) * X2 + jh;
  const float inv_two_pi = 0.15915494309189533577f;
  const int64_t row = (int64_t)blockIdx.x * cpw + c;
  const bool live = row < rows;
  const float4 a = reinterpret_cast<const float4*>(x)[(live ? row : rows - 1) * lpc + l];
  x1s[tid] = make_float2(a.y, a.w);
  x0s[tid] = a.x;
  __syncthreads();
  const float2 up = x1s[n_up];
  const float PA = a.x - a.y - a.z + up.x;
  const float PB = a.z - a.w - x0s[n_right] + up.y;
  float sA, cA = 0.f, sB, cB = 0.f;
  fast_sincos(PA, &sA, &cA);
  fast_sincos(PB, &sB, &cB);
  sps[tid] = make_float2(sA, sB);
  if (SCALARS) {
    // S = sum (1 - cos P) = sites - sum cos P: one reduction serves action and plaquette
    const float qv = cA + cB;
    const float cv = (PA - kTwoPi * floorf((PA + kPi) * inv_two_pi)) + (PB - kTwoPi * floorf((PB + kPi) * inv_two_pi));
    if (lpc == 32) {                         // two chains per wave
      const float q = wave_half_sums(qv), ch = wave_half_sums(cv);
      if ((tid & 31) == 31 && live) {
        if (action) action[row] = (float)sites - q;
        if (avg_plaq) avg_plaq[row] = q / (float)sites;
        if (top_charge) top_charge[row] = ch * inv_two_pi;
      }
    } else {
      const float q = wave_sum(qv), ch = wave_sum(cv);
      if (lpc == kWave) {
        if ((tid & 63) == 0 && live) {
          if (action) action[row] = (float)sites - q;
          if (avg_plaq) avg_plaq[row] = q / (float)sites;
          if (top_charge) top_charge[row] = ch * inv_two_pi;
        }
      } else if ((tid & 63) == 0) {
        red[tid >> 6][0] = q;
        red[tid >> 6][1] = ch;
      }
    }
  }
  __syncthreads();
  if (SCALARS && lpc > kWave && l == 0 && live) {
    const int w0 = base / kWave, nw = lpc / kWave;
    float q = 0.f, ch = 0.f;
    for (int w = 0; w < nw; ++w) {
      q += red[w0 + w][0];
      ch += red[w0 + w][1];
    }
    if (action) action[row] = (float)sites - q;
    if (avg_plaq) avg_plaq[row] = q / (float)sites;
    if (top_charge) top_charge[row] = ch * inv_two_pi;
  }
  if (force && live) {
    const float2 dn = sps[n_down];
    float4 g;
    g.x = beta * (sA - sps[n_left].y);
    g.y = beta * (-sA + dn.x);
    g.z = beta * (sB - sA);
    g.w = beta * (-sB + dn.y);
    reinterpret_cast<float4*>(force)[row * lpc + l] = g;
  }
}

__global__ __launch_bounds__(256) void u1_plaq_sums_kernel(const float* __restrict__ x, int64_t rows,
                                                           int T, int X, float* __restrict__ plaq) {
  const int sites = T * X;
  const int64_t total = rows * sites;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
       g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = g / sites;
    const int site = (int)(g - r * sites);
    const int i = site / X, j = site - i * X;
    const int jp = (j + 1 == X) ? 0 : j + 1;
    const int ip = (i + 1 == T) ? 0 : i + 1;
    const float* xc = x + r * 2 * sites;
    plaq[g] = xc[2 * site] - xc[2 * site + 1] - xc[2 * (i * X + jp)] + xc[2 * (ip * X + j) + 1];
  }
}

// one wave per row: 0.5 * sum v^2
__global__ __launch_bounds__(256) void kinetic_kernel(const float* __restrict__ v, int64_t rows, int D,
                                                      float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* vr = v + r * D;
  float acc = 0.f;
  for (int d = lane; d < D; d += kWave) {
    const float t = vr[d];
    acc += t * t;
  }
  acc = wave_sum(acc);
  if (lane == 0) out[r] = 0.5f * acc;
}

int launch_u1_action_force(const float* x, int64_t rows, int T, int X, float beta, float* action,
                           float* force, float* avg_plaq, float* top_charge, hipStream_t stream) {
  const int sites = T * X;
  // pair kernel: even X, sites / 2 lanes per chain dividing the workgroup (256 threads; 512 for 1024 sites), rows
  // 16-byte aligned (sites * 8 bytes per row is, the bases must be)
  const int lpc = sites / 2;
  const bool aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(force)) & 15) == 0;
  if (X % 2 == 0 && aligned && (lpc == 32 || lpc == 64 || lpc == 128 || lpc == 256 || lpc == 512)) {
    const int threads = lpc == 512 ? 512 : 256;      // (512 threads for the smaller lattices too: measured 0-7 % slower)
    const int cpwf = threads / lpc;
    const int64_t ngroups = ceil_div(rows, cpwf);
    L2HMC_REQUIRE(ngroups < (1ll << 31), "u1_action_force: too many rows");
    const bool scalars = action || avg_plaq || top_charge;
    prof_before(kProfU1, stream);
#define L2HMC_U1_PAIR(SC, TH)                                                                                   \
  hipLaunchKernelGGL((u1_pair_kernel<SC, TH>), dim3((unsigned)ngroups), dim3(TH), 0, stream, x, rows, T, X, beta, \
                     cpwf, action, force, avg_plaq, top_charge)
    if (scalars) {
      if (threads == 512) L2HMC_U1_PAIR(true, 512); else L2HMC_U1_PAIR(true, 256);
    } else {
      if (threads == 512) L2HMC_U1_PAIR(false, 512); else L2HMC_U1_PAIR(false, 256);
    }
#undef L2HMC_U1_PAIR
    prof_after(kProfU1, stream);
    L2HMC_CHECK_LAUNCH("u1_action_force");
    return L2HMC_OK;
  }
  const int cpw = sites >= kLatThreads ? 1 : kLatThreads / sites;
  const size_t lds = sizeof(float) * ((size_t)cpw * 3 * sites + 3 * kLatThreads);
  L2HMC_REQUIRE(lds <= 160 * 1024, "u1_action_force: lattice %dx%d does not fit LDS", T, X);
  const int64_t grid = ceil_div(rows, cpw);
  prof_before(kProfU1, stream);
  hipLaunchKernelGGL(u1_action_force_kernel, dim3((unsigned)grid), dim3(kLatThreads), lds, stream, x,
                     rows, T, X, beta, cpw, action, force, avg_plaq, top_charge);
  prof_after(kProfU1, stream);
  L2HMC_CHECK_LAUNCH("u1_action_force");
  return L2HMC_OK;
}

}  // namespace l2hmc

using namespace l2hmc;

extern "C" int l2hmc_u1_action_force(const float* x, int64_t rows, int32_t T, int32_t X, float beta,
                                     float* action, float* force, float* avg_plaq, float* top_charge,
                                     l2hmc_stream_t stream) {
  L2HMC_REQUIRE(rows >= 0 && T > 0 && X > 0, "u1_action_force: bad shape rows=%lld T=%d X=%d",
                (long long)rows, T, X);
  if (rows == 0) return L2HMC_OK;
  L2HMC_REQUIRE(x != nullptr, "u1_action_force: x is NULL");
  L2HMC_REQUIRE(ceil_div(rows, 1) < (1ll << 31), "u1_action_force: too many rows");
  return launch_u1_action_force(x, rows, T, X, beta, action, force, avg_plaq, top_charge,
                                (hipStream_t)stream);
}

extern "C" int l2hmc_u1_plaq_sums(const float* x, int64_t rows, int32_t T, int32_t X, float* plaq,
                                  l2hmc_stream_t stream) {
  L2HMC_REQUIRE(rows >= 0 && T > 0 && X > 0, "u1_plaq_sums: bad shape");
  if (rows == 0) return L2HMC_OK;
  L2HMC_REQUIRE(x && plaq, "u1_plaq_sums: NULL pointer");
  const int64_t total = rows * T * X;
  const int64_t grid = hmin(ceil_div(total, 256), 2048);
  hipLaunchKernelGGL(u1_plaq_sums_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, rows,
                     T, X, plaq);
  L2HMC_CHECK_LAUNCH("u1_plaq_sums");
  return L2HMC_OK;
}

extern "C" int l2hmc_kinetic_energy(const float* v, int64_t rows, int32_t D, float* out,
                                    l2hmc_stream_t stream) {
  L2HMC_REQUIRE(rows >= 0 && D > 0, "kinetic_energy: bad shape");
  if (rows == 0) return L2HMC_OK;
  L2HMC_REQUIRE(v && out, "kinetic_energy: NULL pointer");
  hipLaunchKernelGGL(kinetic_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                     v, rows, D, out);
  L2HMC_CHECK_LAUNCH("kinetic_energy");
  return L2HMC_OK;
}
