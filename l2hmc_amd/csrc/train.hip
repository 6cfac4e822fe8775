// Training path of the lattice sampler on gfx950: the loss's gradient with respect to every network
// weight and the step size -- what tf.gradients(loss, dynamics.variables) builds in
//   l2hmc/gauge_model.py:799-830 (_calc_loss_and_grads)
// for the graph of
//   l2hmc/dynamics/gauge_dynamics.py:195-313, :412-609 (transition kernel, sub-updates)
//   l2hmc/network/generic_net.py:129-146              (S/T/Q network)
//   l2hmc/gauge_model.py:728-797                      (loss)
// and the Adam update of tf.train.AdamOptimizer (gauge_model.py:942-969).
//
// Design (reverse mode by hand, no autograd engine):
//  * forward ("taped") pass: the layered kernels of stq_dense.hip with every intermediate a network
//    call produces kept in HBM -- first-layer input [a | b*mask], h1, h2, (S,T,Q) and the state the
//    sub-update consumed.  At the benchmark size that is 31 MB per network call, 1.3 GB per
//    trajectory of a 288 GB part; nothing is recomputed.
//  * backward pass: sub-updates in reverse order.  Per network call one element-wise kernel turns the
//    upstream (dx, dv, dlogdet) into the pre-activation gradients of the three heads, three "NT"
//    MFMA products (same tiles as the forward, epilogue = relu gate) carry them back to the inputs,
//    and the force's Hessian-vector product closes the loop through grad_action.
//  * weight gradients are NOT formed per call: the deltas are taped too, and after the loop one
//    split-k "TN" MFMA product per weight matrix contracts over all calls x rows at once (81920
//    rows at the benchmark size), written in the k-contiguous layout the weights are stored in, so
//    the optimiser is one element-wise pass over a flat buffer and the data-parallel all-reduce is
//    one bucket per network.
//  * every reduction runs in a fixed order (no float atomics): results are reproducible.
#include "stq_dense.h"
#include <math.h>

namespace l2hmc {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

// =====================================================================
// forward helpers
// =====================================================================

// in[row] = [a | b * mask(dir)],  st[row] = state   (one wave per row)
__global__ __launch_bounds__(256) void tape_in_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      const float* __restrict__ cm_f,
                                                      const float* __restrict__ cm_b,
                                                      const int* __restrict__ dir,
                                                      const float* __restrict__ state, int64_t rows, int D,
                                                      float* __restrict__ in, float* __restrict__ st) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int d = dir ? dir[row] : 0;
  const float* cm = cm_f ? (d ? cm_b : cm_f) : nullptr;
  for (int c = lane; c < D; c += kWave) {
    in[row * 2 * D + c] = a[row * D + c];
    in[row * 2 * D + D + c] = b[row * D + c] * (cm ? cm[c] : 1.f);
    st[row * D + c] = state[row * D + c];
  }
}

// sub-update from materialised S/T/Q (gauge_dynamics.py:486-590), in place, logdet += (one wave per row)
__global__ __launch_bounds__(256) void train_update_kernel(int mode, const float* __restrict__ stq, int64_t plane,
                                                           const float* __restrict__ g,
                                                           const float* __restrict__ keep_f,
                                                           const float* __restrict__ keep_b,
                                                           const int* __restrict__ dir, float eps, int64_t rows,
                                                           int D, float* x, float* v, float* ld) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int d = dir ? dir[row] : 0;
  float acc = 0.f;
  for (int c = lane; c < D; c += kWave) {
    const int64_t i = row * D + c;
    const float S = stq[i], T = stq[plane + i], Q = stq[2 * plane + i];
    if (mode == 1) {
      const float s = (d ? -0.5f : 0.5f) * eps * S;
      const float kick = 0.5f * eps * (expf(eps * Q) * g[i] - T);
      const float vv = v[i];
      v[i] = d ? expf(s) * (vv + kick) : vv * expf(s) - kick;
      acc += s;
    } else {
      const float k = (d ? keep_b : keep_f)[c];
      const float s = (d ? -eps : eps) * S;
      const float drift = eps * (expf(eps * Q) * v[i] + T);
      const float xx = x[i];
      const float upd = d ? expf(s) * (xx - drift) : xx * expf(s) + drift;
      x[i] = k * xx + (1.f - k) * upd;
      acc += (1.f - k) * s;
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) ld[row] += acc;
}

__global__ void train_accept_kernel(const float* __restrict__ act0, const float* __restrict__ kin0,
                                    const float* __restrict__ act1, const float* __restrict__ kin1,
                                    const float* __restrict__ ld, float beta, int64_t rows,
                                    float* __restrict__ sumlogdet, float* __restrict__ p) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  const double h0 = (double)beta * act0[i] + kin0[i], h1 = (double)beta * act1[i] + kin1[i];
  if (sumlogdet) sumlogdet[i] = ld[i];
  if (p) p[i] = accept_from_delta((float)(h0 - h1 + (double)ld[i]));
}

__global__ void invert_mask_train_kernel(const float* __restrict__ m, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = 1.f - m[i];
}

// =====================================================================
// backward of one sub-update: upstream (dx, dv, dlogdet) -> head pre-activation gradients,
// direct force gradient, coefficient / step-size partials.  A thread owns columns and walks its
// block's rows in order, so the per-block column sums are deterministic.
// =====================================================================
constexpr int kUpdRows = 8;

struct UpdBwdArgs {
  int mode;                         // 1: momentum update, 2: position update
  int64_t rows; int D; float eps;
  const float* stq; int64_t plane;  // forward S, T, Q
  const float* st;                  // state the update consumed: v (mode 1) / x (mode 2)
  const float* in;                  // [rows][2D]: (x, g) in mode 1, (v, m*x) in mode 2
  const float* keep_f; const float* keep_b;
  const int* dir;
  const float* cs; const float* cq; int q_tanh;
  const float* dld;                 // [rows]
  float* dx; float* dv;             // in/out [rows][D]
  float* dg;                        // mode 1 out [rows][D]
  float* dout;                      // [rows][3D]
  float* dcs_part; float* dcq_part; // [nblk][D], +=
  float* deps_part;                 // [nblk], +=
};

__global__ __launch_bounds__(256) void update_bwd_kernel(UpdBwdArgs p) {
  __shared__ float red[4];
  const int D = p.D;
  const int64_t r0 = (int64_t)blockIdx.x * kUpdRows;
  const int64_t r1 = r0 + kUpdRows < p.rows ? r0 + kUpdRows : p.rows;
  const float eps = p.eps;
  float deps = 0.f;
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    const float ecs = expf(p.cs[c]), ecq = expf(p.cq[c]);
    float dcs = 0.f, dcq = 0.f;
    for (int64_t row = r0; row < r1; ++row) {
      const int64_t i = row * D + c;
      const int d = p.dir ? p.dir[row] : 0;
      const float S = p.stq[i], T = p.stq[p.plane + i], Q = p.stq[2 * p.plane + i];
      const float dl = p.dld[row];
      const float eq = expf(eps * Q);
      float dS, dT, dQ;
      if (p.mode == 1) {
        const float v = p.st[i], g = p.in[row * 2 * D + D + c], u = p.dv[i];
        const float he = 0.5f * eps;
        if (!d) {
          const float es = expf(he * S);
          const float ds = u * v * es + dl;
          p.dv[i] = u * es;
          dS = ds * he; dT = u * he; dQ = -u * he * eq * g * eps;
          p.dg[i] = -u * he * eq;
          deps += ds * 0.5f * S - u * 0.5f * (eq * g - T) - u * he * g * eq * Q;
        } else {
          const float es = expf(-he * S);
          const float kick = he * (eq * g - T);
          const float vp = es * (v + kick);
          const float dw = u * es;
          const float ds = u * vp + dl;
          p.dv[i] = dw;
          dS = -he * ds; dT = -dw * he; dQ = dw * he * eq * g * eps;
          p.dg[i] = dw * he * eq;
          deps += -0.5f * S * ds + dw * 0.5f * (eq * g - T) + dw * he * g * eq * Q;
        }
      } else {
        const float k = (d ? p.keep_b : p.keep_f)[c], mi = 1.f - k;
        const float x = p.st[i], v = p.in[row * 2 * D + c], u = p.dx[i];
        const float dy = mi * u;
        if (!d) {
          const float es = expf(eps * S);
          const float ds = dy * x * es + dl * mi;
          p.dx[i] = k * u + dy * es;
          p.dv[i] += dy * eps * eq;
          dS = eps * ds; dT = dy * eps; dQ = dy * eps * eq * v * eps;
          deps += ds * S + dy * (eq * v + T) + dy * eps * v * eq * Q;
        } else {
          const float es = expf(-eps * S);
          const float w = x - eps * (eq * v + T);
          const float dw = dy * es;
          const float ds = dy * (es * w) + dl * mi;
          p.dx[i] = k * u + dw;
          p.dv[i] -= dw * eps * eq;
          dS = -eps * ds; dT = -dw * eps; dQ = -dw * eps * eq * v * eps;
          deps += -S * ds - dw * (eq * v + T) - dw * eps * v * eq * Q;
        }
      }
      // through tanh(.) * exp(coeff) (generic_net.py:139-144)
      const float th = S / ecs;
      float daq = dQ * ecq;
      if (p.q_tanh) {
        const float tq = Q / ecq;
        daq *= 1.f - tq * tq;
      }
      dcs += dS * S;
      dcq += dQ * Q;
      float* o = p.dout + row * 3 * D + c;
      o[0] = dS * ecs * (1.f - th * th);
      o[D] = dT;
      o[2 * D] = daq;
    }
    p.dcs_part[(int64_t)blockIdx.x * D + c] += dcs;
    p.dcq_part[(int64_t)blockIdx.x * D + c] += dcq;
  }
  deps = wave_sum(deps);
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) red[wave] = deps;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    p.deps_part[blockIdx.x] += t;
  }
}

// position-network inputs (v, m*x): dv += din[:, :D], dx += m * din[:, D:]
__global__ __launch_bounds__(256) void xnet_in_bwd_kernel(const float* __restrict__ din,
                                                          const float* __restrict__ cm_f,
                                                          const float* __restrict__ cm_b,
                                                          const int* __restrict__ dir, int64_t rows, int D,
                                                          float* dx, float* dv) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* cm = (dir && dir[row]) ? cm_b : cm_f;
  for (int c = lane; c < D; c += kWave) {
    dv[row * D + c] += din[row * 2 * D + c];
    dx[row * D + c] += cm[c] * din[row * 2 * D + D + c];
  }
}

// momentum-network inputs (x, g = beta * grad_action(x)):
//   dx += din[:, :D] + beta * Hess(action)(x) . (dg + din[:, D:])
// The Hessian-vector product has the force's stencil with sin(P) replaced by cos(P) * P[u]
// (lattice.py:246-262 differentiated once more).  One wave per chain, chain staged in LDS.
__global__ __launch_bounds__(256) void vnet_in_bwd_kernel(const float* __restrict__ din,
                                                          const float* __restrict__ dg,
                                                          const float* __restrict__ in, float beta, int T,
                                                          int X, int64_t rows, float* dx) {
  extern __shared__ float lds[];
  const int D = 2 * T * X, sites = T * X;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  float* xs = lds + (size_t)wave * (2 * D + sites);
  float* us = xs + D;
  float* cp = us + D;
  if (row < rows) {
    for (int c = lane; c < D; c += kWave) {
      xs[c] = in[row * 2 * D + c];
      us[c] = dg[row * D + c] + din[row * 2 * D + D + c];
    }
  }
  __syncthreads();
  if (row < rows) {
    for (int s = lane; s < sites; s += kWave) {
      const int i = s / X, j = s - i * X;
      const int jr = (j + 1 == X) ? 0 : j + 1, id = (i + 1 == T) ? 0 : i + 1;
      const int e = 2 * s, er = 2 * (i * X + jr), ed = 2 * (id * X + j);
      const float P = xs[e] - xs[e + 1] - xs[er] + xs[ed + 1];
      const float Pu = us[e] - us[e + 1] - us[er] + us[ed + 1];
      cp[s] = cosf(P) * Pu;
    }
  }
  __syncthreads();
  if (row < rows) {
    for (int s = lane; s < sites; s += kWave) {
      const int i = s / X, j = s - i * X;
      const int jl = (j == 0) ? X - 1 : j - 1, iu = (i == 0) ? T - 1 : i - 1;
      const float c = cp[s];
      const float h0 = c - cp[i * X + jl], h1 = -c + cp[iu * X + j];
      dx[row * D + 2 * s] += din[row * 2 * D + 2 * s] + beta * h0;
      dx[row * D + 2 * s + 1] += din[row * 2 * D + 2 * s + 1] + beta * h1;
    }
  }
}

// column sums of a taped [calls * rows][n] array (bias gradients), optionally also weighted by the
// (cos, sin) time input of each (call, row) -- the t_layer kernel's gradient (generic_net.py:131).
// HBM-bound: every thread streams 16-byte pieces of consecutive rows; a workgroup covers up to 1024
// columns (blockIdx.x) of one row chunk (blockIdx.y).  part: [S][3][n] (plain, cos-, sin-weighted).
// One block = column group bx of row chunk by; sm: [nsteps][2] (cos, sin) + 256 * 12 floats of reduction scratch.
constexpr int kColsumMaxS = 512;
struct ColsumArgs {
  const float* src; int64_t Rt; int n;
  int64_t rows; int nsteps; const int* dir; int timed;
  int64_t chunk;        // rows per block
  float* part;          // [S][3][n]
  int ncg, S;           // column groups (of 1024), row chunks; ncg * S blocks in all (0: no column sums)
};
__device__ __forceinline__ void colsum_block(const ColsumArgs& c, int bx, int by, float* sm) {
  const float* __restrict__ src = c.src;
  const int* __restrict__ dir = c.dir;
  const int64_t Rt = c.Rt, rows = c.rows, chunk = c.chunk;
  const int n = c.n, nsteps = c.nsteps, timed = c.timed;
  float* __restrict__ part = c.part;
  float* tab = sm;
  float* red = sm + 2 * nsteps;
  if (timed) {
    for (int i = threadIdx.x; i < nsteps; i += blockDim.x) {
      const float ang = (float)(2.0 * M_PI) * (float)i / (float)nsteps;
      tab[2 * i] = cosf(ang);
      tab[2 * i + 1] = sinf(ang);
    }
  }
  __syncthreads();
  const int c0 = bx * 1024;
  const int nc4 = (n - c0 < 1024 ? n - c0 : 1024) / 4;   // 16-byte pieces per row in this column group
  const int rpp = 256 / nc4 > 0 ? 256 / nc4 : 1;          // rows per pass
  const int rl = threadIdx.x / nc4, cl = threadIdx.x - rl * nc4;
  const bool active = rl < rpp;
  const int64_t rb = (int64_t)by * chunk;
  const int64_t re = rb + chunk < Rt ? rb + chunk : Rt;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0;
  if (active) {
    // (call, row) of the taped row, advanced incrementally: one 64-bit division per thread instead of one per row
    int64_t call = timed ? (rb + rl) / rows : 0, row = timed ? (rb + rl) - call * rows : 0;
#pragma unroll 4
    for (int64_t rr = rb + rl; rr < re; rr += rpp) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + rr * n + c0 + cl * 4);
      s0 += v;
      if (timed) {
        const int step = (int)(call >> 1);
        const int i = (dir && dir[row]) ? nsteps - 1 - step : step;
        s1 += tab[2 * i] * v;
        s2 += tab[2 * i + 1] * v;
        row += rpp;
        while (row >= rows) {
          row -= rows;
          ++call;
        }
      }
    }
    float* o = red + (size_t)(rl * nc4 + cl) * 12;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = s0[e];
      o[4 + e] = s1[e];
      o[8 + e] = s2[e];
    }
  }
  __syncthreads();
  if (threadIdx.x < nc4) {
    float* o = part + (size_t)by * 3 * n;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t0 = 0.f, t1 = 0.f, t2 = 0.f;
      for (int q = 0; q < rpp; ++q) {
        const float* r = red + (size_t)(q * nc4 + threadIdx.x) * 12;
        t0 += r[e];
        t1 += r[4 + e];
        t2 += r[8 + e];
      }
      const int col = c0 + threadIdx.x * 4 + e;
      o[col] = t0;
      o[n + col] = t1;
      o[2 * n + col] = t2;
    }
  }
}

__global__ __launch_bounds__(256) void colsum_kernel(ColsumArgs c) {
  extern __shared__ float sm[];
  colsum_block(c, blockIdx.x, blockIdx.y, sm);
}

// =====================================================================
// weight gradients: C[m][n] = sum_r P[r][m] * Q[r][n]   ("TN", contraction over rows)
// 128 x 128 tile per workgroup, 16 rows per stage, fp32 32x32x2 MFMAs.  Tiles are staged TRANSPOSED
// ([m][r], column stride 20 floats): the operands of lane (k-half, m) for four consecutive products are
// one conflict-free ds_read_b128 (8 fragment reads per wave and stage instead of 32 four-byte ones).  The
// contraction is split over `splits` workgroups per tile; partials are summed in a fixed order afterwards.
// =====================================================================
struct GemmTnArgs {
  const float* P; int ldp; int M;
  const float* Q; int ldq; int N;
  int64_t R;            // contraction length
  int64_t chunk;        // rows per split (multiple of 16)
  float* part;          // [splits][M][N]
  int mt, nt;
  // The bias gradient that belongs to this weight gradient (column sums of P) rides along as cs.ncg * cs.S extra
  // workgroups behind the product's: HBM-bound blocks that share the CUs with the matrix-pipe-bound ones (a third
  // workgroup per CU fits beside the product's two), so the sums cost no time of their own.
  ColsumArgs cs;
  int nprod;            // workgroups of the product proper (tiles * splits)
};

__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTnArgs p) {
  constexpr int BM = 128, BN = 128, BKR = 16, LDK = BKR + 4;
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * BN * LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, r = lane & 31;
  const int tiles = p.mt * p.nt;
  if (p.cs.ncg) {                                        // uniform per workgroup
    if ((int)blockIdx.x >= p.nprod) {
      const int b = blockIdx.x - p.nprod;
      colsum_block(p.cs, b % p.cs.ncg, b / p.cs.ncg, lds);
      return;
    }
  }
  // (an XCD-aware order -- all tiles of one row range on one XCD -- was measured: no change, the Infinity Cache
  // already serves the cross-XCD re-reads)
  const int split = blockIdx.x / tiles, tile = blockIdx.x - split * tiles;
  const int m0 = (tile / p.nt) * BM, n0 = (tile % p.nt) * BN;
  const int64_t rbeg = (int64_t)split * p.chunk;
  const int64_t rend = rbeg + p.chunk < p.R ? rbeg + p.chunk : p.R;

  // staging: a thread owns a 4-row x 4-column block of ONE operand (threads 0..127: P, 128..255: Q): four 16-byte
  // global loads (one per row; 16 consecutive lanes cover 256 contiguous bytes of a row), the 4 x 4 transpose is a
  // renaming of registers, four 16-byte LDS stores.  Lane -> block: k-group (lane >> 2) & 3, column group
  // 16 * (wave & 1) + 4 * (lane >> 4) + (lane & 3).  The fragment reads (ds_read_b128: four groups of 16 lanes, banks
  // modulo 64) see 16 columns of one k-group at column stride 20: slot 5 c (mod 16), all different.  The stores
  // (ds_write_b128: groups of 8 consecutive lanes, banks modulo 32) are 2-way conflicted with this mapping
  // (SQ_LDS_BANK_CONFLICT = a third of the LDS-active cycles); the conflict-free mapping -- k-group lane & 3, column
  // group lane >> 2 -- was measured: 281 -> 283 us per product (a store is bound by moving its registers to the LDS,
  // 13 cycles, not by the array, and consecutive lanes then load from four different rows).  This one stays.
  const bool isq = tid >= 128;
  const int kg = (lane >> 2) & 3, cg = 16 * (wave & 1) + 4 * (lane >> 4) + (lane & 3);
  const float* gsrc = isq ? p.Q : p.P;
  const int ldg = isq ? p.ldq : p.ldp;
  const int gcol = (isq ? n0 : m0) + 4 * cg;
  const bool col_ok = gcol < (isq ? p.N : p.M);
  // (Loads two stages ahead of the products -- two register slots, loop unrolled by two -- were measured: 164 VGPRs,
  // two workgroups per CU instead of three, 295 -> 319 us per product.  One stage ahead it stays.)
  // Buffer loads (common.h: buf_load16): a scalar resource re-based at the stage's first row, one 32-bit offset per
  // lane and row (< 16 rows of ldg floats); a row past the end or a column group past the operand's width points its
  // lane beyond the resource's window and reads 0, so the four loads of a stage are unconditional.
  f32x4 rg[4];
  unsigned goff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) goff[i] = ((unsigned)(4 * kg + i) * (unsigned)ldg + (unsigned)gcol) * 4u;
  auto load_stage = [&](int64_t rr) {
    const WSection rs = wsection(gsrc + rr * ldg);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = col_ok && rr + 4 * kg + i < rend;
      rg[i] = buf_load16(rs, ok ? goff[i] : 0xfffffff0u, 0u);
    }
  };
  auto store_stage = [&](int buf) {
    float* dst = lds + buf * 2 * BN * LDK + (isq ? BN * LDK : 0) + (4 * cg) * LDK + 4 * kg;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 c = {rg[0][j], rg[1][j], rg[2][j], rg[3][j]};
      *reinterpret_cast<f32x4*>(dst + j * LDK) = c;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (rbeg < rend) {
    load_stage(rbeg);
    store_stage(0);
    __syncthreads();
    int cur = 0;
    for (int64_t rr = rbeg; rr < rend; rr += BKR) {
      const bool more = rr + BKR < rend;
      if (more) load_stage(rr + BKR);
      const float* ps = lds + cur * 2 * BN * LDK + (wm * 64 + r) * LDK + 4 * half;
      const float* qs = lds + cur * 2 * BN * LDK + BN * LDK + (wn * 64 + r) * LDK + 4 * half;
      // The 8 operand fragments of the stage (16-byte reads: 4 contraction indices each) are fetched before its first
      // product.  tools/tn_loop_bench.hip takes the loop apart (profiles/r03_tn_loop_bench.txt): MFMAs alone 143-147
      // TFLOP/s, + these reads 130-132 (round 2: 32 four-byte reads from row-major tiles, 125-128), + barrier and stage
      // stores 130-134, + the stage's global loads 116-121: every register a load writes costs the matrix pipe about
      // four cycles at any occupancy.  In the training step: 294 -> 282 us per product on the same box.
      f32x4 av[2][2], bv[2][2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        av[s][0] = *reinterpret_cast<const f32x4*>(ps + 8 * s);
        av[s][1] = *reinterpret_cast<const f32x4*>(ps + 32 * LDK + 8 * s);
        bv[s][0] = *reinterpret_cast<const f32x4*>(qs + 8 * s);
        bv[s][1] = *reinterpret_cast<const f32x4*>(qs + 32 * LDK + 8 * s);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) {                    // contraction indices 8 s + e (lanes 0..31), 8 s + 4 + e
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][0][e], bv[s][0][e], acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][0][e], bv[s][1][e], acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][1][e], bv[s][0][e], acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][1][e], bv[s][1][e], acc[1][1], 0, 0, 0);
        }
      if (more) store_stage(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }
  float* out = p.part + (size_t)split * p.M * p.N;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + r;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (row < p.M && col < p.N) out[(size_t)row * p.N + col] = acc[i][j][e];
      }
    }
}

// out[i] = sum_s part[s][i]  (fixed order: four interleaved partial sums, then their sum)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, int S, int64_t count,
                                                              float* __restrict__ out) {
  __shared__ float red[4][64];
  const int il = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + il;
  float t = 0.f;
  if (i < count) {
#pragma unroll 8
    for (int s = sl; s < S; s += 4) t += part[(size_t)s * count + i];
  }
  red[sl][il] = t;
  __syncthreads();
  if (sl == 0 && i < count) out[i] = (red[0][il] + red[1][il]) + (red[2][il] + red[3][il]);
}

// The same sum for partials laid out [S][3][n] (colsum_block): plane blockIdx.y goes to its own destination.
struct PlaneOuts { float* o[3]; };
__global__ __launch_bounds__(256) void reduce_planes_kernel(const float* __restrict__ part, int S, int n,
                                                            PlaneOuts outs) {
  __shared__ float red[4][64];
  const int il = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + il;
  const float* src = part + (size_t)blockIdx.y * n;
  float t = 0.f;
  if (i < n) {
#pragma unroll 8
    for (int s = sl; s < S; s += 4) t += src[(size_t)s * 3 * n + i];
  }
  red[sl][il] = t;
  __syncthreads();
  if (sl == 0 && i < n) outs.o[blockIdx.y][i] = (red[0][il] + red[1][il]) + (red[2][il] + red[3][il]);
}

// Both reductions behind one weight-gradient product in ONE launch: blocks [0, nb1) sum the product's split-k partials
// (as reduce_partials_kernel), the rest the column-sum planes (as reduce_planes_kernel: block (x, plane)).
__global__ __launch_bounds__(256) void reduce_product_and_planes_kernel(const float* __restrict__ part, int S,
                                                                        int64_t count, float* __restrict__ out,
                                                                        int nb1, const float* __restrict__ cpart,
                                                                        int cS, int n, PlaneOuts outs) {
  __shared__ float red[4][64];
  const int il = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const bool first = (int)blockIdx.x < nb1;                     // uniform per block
  float t = 0.f;
  int64_t i;
  float* dst;
  if (first) {
    i = (int64_t)blockIdx.x * 64 + il;
    if (i < count) {
#pragma unroll 8
      for (int s = sl; s < S; s += 4) t += part[(size_t)s * count + i];
    }
    dst = (i < count) ? out + i : nullptr;
  } else {
    const int b = blockIdx.x - nb1, nbx = (n + 63) / 64;
    const int plane = b / nbx;
    i = (int64_t)(b - plane * nbx) * 64 + il;
    const float* src = cpart + (size_t)plane * n;
    if (i < n) {
#pragma unroll 8
      for (int s = sl; s < cS; s += 4) t += src[(size_t)s * 3 * n + i];
    }
    dst = (i < n) ? outs.o[plane] + i : nullptr;
  }
  red[sl][il] = t;
  __syncthreads();
  if (sl == 0 && dst) *dst = (red[0][il] + red[1][il]) + (red[2][il] + red[3][il]);
}

// out[c][r] = in[r][c]
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, int R, int Cn,
                                                        float* __restrict__ out) {
  __shared__ float t[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8)
    if (r0 + k < R && c0 + tx < Cn) t[k][tx] = in[(size_t)(r0 + k) * Cn + c0 + tx];
  __syncthreads();
  for (int k = ty; k < 32; k += 8)
    if (c0 + k < Cn && r0 + tx < R) out[(size_t)(c0 + k) * R + r0 + tx] = t[tx][k];
}

// =====================================================================
// loss backward (gauge_model.py:728-797) + the accept probability's dependence on the final state
// (gauge_dynamics.py:284-313, :592-609):  d loss / d (x_N, v_N, sumlogdet) for the 2B stacked chains
// (rows [0,B): chains started at x, rows [B,2B): chains started at the auxiliary z).
// One wave per (x, z) pair.
// =====================================================================
struct LossBwdArgs {
  int T, X; int64_t B; float beta;
  const float* x0;   // [2B][D] initial states (x rows, then z rows)
  const float* xN; const float* vN;   // [2B][D] proposed states
  const float* p;    // [2B]
  int metric; float loss_scale, aux_weight, std_weight, charge_weight, inv_count;
  float* terms;      // [B] per-chain loss (forward value) or NULL
  float* dxN; float* dvN;   // [2B][D]
  float* dld;        // [2B]
};

__device__ __forceinline__ float metric_val(int m, float a, float b) {
  switch (m) {
    case 0: return fabsf(a - b);
    case 1: return (a - b) * (a - b);
    case 2: return fabsf(cosf(a) - cosf(b));
    case 3: { const float c = cosf(a) - cosf(b); return c * c; }
    default: return 1.f - cosf(a - b);
  }
}
// d metric(a, b) / d b
__device__ __forceinline__ float metric_db(int m, float a, float b) {
  switch (m) {
    case 0: return a > b ? -1.f : (a < b ? 1.f : 0.f);
    case 1: return -2.f * (a - b);
    case 2: { const float c = cosf(a) - cosf(b); return (c > 0.f ? 1.f : (c < 0.f ? -1.f : 0.f)) * sinf(b); }
    case 3: return 2.f * (cosf(a) - cosf(b)) * sinf(b);
    default: return -sinf(a - b);
  }
}
// 4-term Fourier series of the plaquette projection and its derivative (lattice.py:108-128)
__device__ __forceinline__ void proj_series(float P, float& f, float& df) {
  f = 0.f; df = 0.f;
  float sgn = -1.f;
#pragma unroll
  for (int n = 1; n <= 4; ++n) {
    f += (-2.f / n) * sgn * sinf(n * P);
    df += -2.f * sgn * cosf(n * P);
    sgn = -sgn;
  }
}

__device__ __forceinline__ float plaq_at(const float* xs, int s, int T, int X) {
  const int i = s / X, j = s - i * X;
  const int jr = (j + 1 == X) ? 0 : j + 1, id = (i + 1 == T) ? 0 : i + 1;
  return xs[2 * s] - xs[2 * s + 1] - xs[2 * (i * X + jr)] + xs[2 * (id * X + j) + 1];
}

__global__ __launch_bounds__(64) void loss_bwd_kernel(LossBwdArgs p) {
  extern __shared__ float lds[];
  const int T = p.T, X = p.X, sites = T * X, D = 2 * sites;
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.x;
  float* xs = lds;            // [D] staged chain
  float* cp = xs + D;         // [sites] d q / d plaquette of x'
  float* sp = cp + sites;     // [sites] sin(plaquette)
  const float kTwoPiInv = 0.15915494309189535f;
  const float* x = p.x0 + b * D;
  const float* z = p.x0 + (p.B + b) * D;
  const float* xp = p.xN + b * D;

  auto charge_of = [&](const float* src) {
    __syncthreads();
    for (int c = lane; c < D; c += 64) xs[c] = src[c];
    __syncthreads();
    float q = 0.f;
    for (int s = lane; s < sites; s += 64) {
      float f, df;
      proj_series(plaq_at(xs, s, T, X), f, df);
      q += f;
    }
    return wave_sum(q) * kTwoPiInv;
  };
  const float qx = charge_of(x);
  const float qz = charge_of(z);
  // proposed x-chain state: charge, d charge / d plaquette, sin(plaquette) for the force
  __syncthreads();
  for (int c = lane; c < D; c += 64) xs[c] = xp[c];
  __syncthreads();
  float qp = 0.f;
  for (int s = lane; s < sites; s += 64) {
    const float P = plaq_at(xs, s, T, X);
    float f, df;
    proj_series(P, f, df);
    qp += f;
    cp[s] = df * kTwoPiInv;
    sp[s] = sinf(P);
  }
  qp = wave_sum(qp) * kTwoPiInv;
  float mx = 0.f, mz = 0.f;
  for (int c = lane; c < D; c += 64) {
    mx += metric_val(p.metric, x[c], xs[c]);
    mz += metric_val(p.metric, z[c], xs[c]);
  }
  mx = wave_sum(mx);
  mz = wave_sum(mz);
  const float px = p.p[b], pz = p.p[p.B + b];
  const float e = 1e-3f;                       // gauge_model.py:745
  const float ls = p.loss_scale, aw = p.aux_weight, sw = p.std_weight, cw = p.charge_weight;
  const float xstd = mx * px + e, zstd = aw * (mz * pz + e);
  const float dqx = qx - qp, dqz = qz - qp;
  const float xq = px * fabsf(dqx) + e, zq = aw * (pz * fabsf(dqz) + e);
  if (p.terms && lane == 0)
    p.terms[b] = sw * (ls * (1.f / xstd + 1.f / zstd) - (xstd + zstd) / ls) + cw * (xq + zq);
  const float w = p.inv_count;
  const float ax = w * sw * (-ls / (xstd * xstd) - 1.f / ls);
  const float az = w * sw * (-ls / (zstd * zstd) - 1.f / ls) * aw;
  const float dpx = ax * mx + w * cw * fabsf(dqx);
  const float dpz = az * mz + w * cw * aw * fabsf(dqz);
  // d / d q(x'):  xq, zq depend on |q - q'|
  const float sgx = dqx > 0.f ? 1.f : (dqx < 0.f ? -1.f : 0.f), sgz = dqz > 0.f ? 1.f : (dqz < 0.f ? -1.f : 0.f);
  const float dqp = -w * cw * (px * sgx + aw * pz * sgz);
  // accept probability: p = exp(min(H0 - H1 + ld, 0)) -> d p / d(.) = p where p < 1
  const float ddx = (px < 1.f) ? dpx * px : 0.f;
  const float ddz = (pz < 1.f) ? dpz * pz : 0.f;
  if (lane == 0) {
    p.dld[b] = ddx;
    p.dld[p.B + b] = ddz;
  }
  __syncthreads();
  for (int s = lane; s < sites; s += 64) {
    const int i = s / X, j = s - i * X;
    const int jl = (j == 0) ? X - 1 : j - 1, iu = (i == 0) ? T - 1 : i - 1;
    const int sl = i * X + jl, su = iu * X + j;
#pragma unroll
    for (int mu = 0; mu < 2; ++mu) {
      const int c = 2 * s + mu;
      const float dq = mu == 0 ? cp[s] - cp[sl] : -cp[s] + cp[su];
      const float frc = p.beta * (mu == 0 ? sp[s] - sp[sl] : -sp[s] + sp[su]);
      const float a = xs[c];
      p.dxN[b * D + c] = ax * px * metric_db(p.metric, x[c], a) + az * pz * metric_db(p.metric, z[c], a) +
                         dqp * dq - ddx * frc;
      p.dvN[b * D + c] = -ddx * p.vN[b * D + c];
    }
  }
  // auxiliary chain: only its accept probability enters the loss
  __syncthreads();
  const float* zp = p.xN + (p.B + b) * D;
  for (int c = lane; c < D; c += 64) xs[c] = zp[c];
  __syncthreads();
  for (int s = lane; s < sites; s += 64) sp[s] = sinf(plaq_at(xs, s, T, X));
  __syncthreads();
  for (int s = lane; s < sites; s += 64) {
    const int i = s / X, j = s - i * X;
    const int jl = (j == 0) ? X - 1 : j - 1, iu = (i == 0) ? T - 1 : i - 1;
    const int sl = i * X + jl, su = iu * X + j;
    const int64_t o = (p.B + b) * D + 2 * s;
    p.dxN[o] = -ddz * p.beta * (sp[s] - sp[sl]);
    p.dxN[o + 1] = -ddz * p.beta * (-sp[s] + sp[su]);
    p.dvN[o] = -ddz * p.vN[o];
    p.dvN[o + 1] = -ddz * p.vN[o + 1];
  }
}

// =====================================================================
// optimiser: tf.train.AdamOptimizer.apply_gradients with optional clip_by_global_norm
// (gauge_model.py:826-827, :942-969)
// =====================================================================
// sum of squares in a fixed order: one workgroup, 1024 threads.  Elements in [tri_lo, tri_hi) count
// three times (the packed first-layer bias stands for the reference's three bias variables).
__global__ __launch_bounds__(1024) void sumsq_kernel(const float* __restrict__ g, int64_t n, int64_t tri_lo,
                                                     int64_t tri_hi, float* out, int accumulate) {
  __shared__ float red[16];
  float t = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const float v = g[i];
    t += ((i >= tri_lo && i < tri_hi) ? 3.f : 1.f) * v * v;
  }
  t = wave_sum(t);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += red[i];
    *out = accumulate ? *out + s : s;
  }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   float lr_t, float b1, float b2, float eps,
                                                   const float* __restrict__ gnorm_sq, float clip,
                                                   int64_t tri_lo, int64_t tri_hi) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float scale = 1.f;
  if (gnorm_sq) {
    // tf.clip_by_global_norm: g * clip / max(norm, clip)
    const float norm = sqrtf(*gnorm_sq);
    scale = clip / fmaxf(norm, clip);
  }
  const float gi = g[i] * scale;
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float step = lr_t * mi / (sqrtf(vi) + eps);
  w[i] -= ((i >= tri_lo && i < tri_hi) ? 3.f : 1.f) * step;
}

// =====================================================================
// host orchestration
// =====================================================================
struct NetTape {
  float *in, *h1, *h2, *stq, *st, *d1, *d2, *dout;   // [calls][rows][.]
  float *dcs_part, *dcq_part;                         // [nblk][D]
  float *w1_n, *wh_n, *whd_n;                         // weights transposed for the backward-data products
  float *bpack;                                       // fragment-ordered transposed weights (fused reverse pass)
  unsigned *gate;                                     // relu masks in fragment order (fused kernels)
  float *feat;                                        // ConvNet3D: [calls][rows][Ka+Kb] front-end features
  float *conv_part;                                   // ConvNet3D: [workgroups][2][filter-gradient slot]
};
struct TrainWs {
  NetTape x, v;
  float *mask_inv, *ld, *act0, *kin0, *act1, *kin1, *g, *dg, *din, *dfeat, *part, *eps_part;
  float* zero_base; size_t zero_bytes;                // dcs_part / dcq_part of both nets + eps_part
  size_t bytes;
};

static inline int64_t imax64(int64_t a, int64_t b) { return a > b ? a : b; }
static inline size_t smax(size_t a, size_t b) { return a > b ? a : b; }
static int64_t upd_blocks(int64_t rows) { return ceil_div(rows, kUpdRows); }
// S row chunks of a column-sum pass over Rt taped rows
static int colsum_chunks(int64_t Rt) { return (int)hmin(kColsumMaxS, imax64(1, Rt / 128)); }

static int tn_splits(int mt, int nt, int64_t R) {
  // two workgroups per CU; 768 and 1024 measured the same, 256 is 4 % slower (profiles/r02_tn_loop_bench.txt)
  int s = (int)imax64(1, 512 / ((int64_t)mt * nt));
  const int64_t maxs = imax64(1, ceil_div(R, 256));   // at least 256 rows per split
  return (int)hmin(s, maxs);
}

static TrainWs carve_train_ws(const l2hmc_gauge_plan* p, int64_t rows, void* ws) {
  TrainWs w{};
  const int D = 2 * p->T * p->X;
  const int C = 2 * p->num_steps;
  char* base = static_cast<char*>(ws);
  size_t off = 0;
  auto take = [&](size_t nfloats) {
    float* q = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off += align_up(nfloats * sizeof(float), 256);
    return q;
  };
  const l2hmc_dense_net* nets[2] = {&p->xnet, &p->vnet};
  const l2hmc_conv3d_front* fronts[2] = {&p->xfront, &p->vfront};
  const bool conv = (p->flags & L2HMC_PLAN_CONV3D) != 0;
  NetTape* tapes[2] = {&w.x, &w.v};
  size_t part_max = 0;
  int kin_max = 2 * D;
  for (int k = 0; k < 2; ++k) {
    const int H = nets[k]->H;
    const int Kin = nets[k]->Ka + nets[k]->Kb;         // first-layer fan-in: 2D (generic) or 2*nflat (ConvNet3D)
    kin_max = hmax(kin_max, Kin);
    NetTape& t = *tapes[k];
    const size_t cr = (size_t)C * rows;
    t.in = take(cr * 2 * D);
    t.h1 = take(cr * H);
    t.h2 = take(cr * H);
    t.stq = take(cr * 3 * D);
    t.st = take(cr * D);
    t.d1 = take(cr * H);
    t.d2 = take(cr * H);
    t.dout = take(cr * 3 * D);
    t.feat = conv ? take(cr * Kin) : nullptr;
    t.conv_part = conv ? take((size_t)ceil_div(rows, conv3d_cpw(p->T, p->X, fronts[k]->F)) * 2 *
                              conv3d_bwd_part_floats(fronts[k]->F))
                       : nullptr;
    t.bpack = take(fused_bwd_pack_floats(nets[k]));
    t.gate = reinterpret_cast<unsigned*>(take((size_t)C * 2 * ceil_div(rows, 16) * 512));   // up to 512 threads per workgroup (ConvNet3D instance)
    t.w1_n = take((size_t)Kin * H);
    t.wh_n = take((size_t)H * H);
    t.whd_n = take((size_t)3 * D * H);
    // split-k partials of the three weight-gradient products and the column sums
    const int64_t R = (int64_t)cr;
    auto need = [&](int M, int N) {
      const int mt = (int)ceil_div(M, 128), nt = (int)ceil_div(N, 128);
      return (size_t)tn_splits(mt, nt, R) * M * N + (size_t)colsum_chunks(R) * 3 * M;   // product + column sums
    };
    part_max = smax(part_max, smax(need(H, Kin), smax(need(H, H), need(3 * D, H))));
    if (conv) part_max = smax(part_max, 2 * conv3d_bwd_part_floats(fronts[k]->F));
    part_max = smax(part_max, (size_t)(kColsumMaxS + 1) * 3 * hmax(H, 3 * D));
  }
  w.mask_inv = take((size_t)p->num_steps * D);
  w.ld = take(rows);
  w.act0 = take(rows); w.kin0 = take(rows); w.act1 = take(rows); w.kin1 = take(rows);
  w.g = take((size_t)rows * D);
  w.dg = take((size_t)rows * D);
  w.din = take((size_t)rows * 2 * D);
  w.dfeat = conv ? take((size_t)rows * kin_max) : nullptr;
  w.part = take(part_max);
  // the += accumulators of the reverse pass, contiguous: one memset clears them all
  const size_t zero_off = off;
  w.zero_base = take(0);
  for (int k = 0; k < 2; ++k) {
    tapes[k]->dcs_part = take((size_t)upd_blocks(rows) * D);
    tapes[k]->dcq_part = take((size_t)upd_blocks(rows) * D);
  }
  w.eps_part = take(upd_blocks(rows));
  w.zero_bytes = off - zero_off;
  w.bytes = off;
  return w;
}

static int check_train_plan(const l2hmc_gauge_plan* p) {
  L2HMC_REQUIRE(p != nullptr, "train: NULL plan");
  L2HMC_REQUIRE(p->T > 0 && p->X > 0 && p->num_steps > 0 && p->masks != nullptr, "train: bad plan");
  L2HMC_REQUIRE(!p->hmc, "train: hmc plans have no trainable networks");
  const bool conv = (p->flags & L2HMC_PLAN_CONV3D) != 0;
  const int D = 2 * p->T * p->X;
  const l2hmc_dense_net* nets[2] = {&p->xnet, &p->vnet};
  const l2hmc_conv3d_front* fronts[2] = {&p->xfront, &p->vfront};
  for (int k = 0; k < 2; ++k) {
    const l2hmc_dense_net* n = nets[k];
    const int kin = conv ? conv3d_nflat(p->T, p->X, fronts[k]->F) : D;
    // the training kernels (taped forward, TN weight-gradient products, column sums) stage 32-wide k-tiles only
    L2HMC_REQUIRE(dense_net_tileable(n) && n->D % 32 == 0 && n->D == D && n->Ka == kin && n->Kb == kin,
                  "train: network widths (D=%d Ka=%d Kb=%d H=%d) must be multiples of 32 with Ka = Kb = %d",
                  n->D, n->Ka, n->Kb, n->H, kin);
    if (conv) {
      const l2hmc_conv3d_front* f = fronts[k];
      L2HMC_REQUIRE(f->F > 0 && f->w1_a && f->b1_a && f->w2_a && f->b2_a && f->w1_b && f->b1_b && f->w2_b && f->b2_b,
                    "train: NULL conv front-end pointer");
    }
    L2HMC_REQUIRE(n->w1_t && n->wt && n->b1 && n->wh_t && n->bh && n->whd_t && n->bhd && n->coeff_s && n->coeff_q,
                  "train: NULL weight pointer");
  }
  return L2HMC_OK;
}

static void step_times(int N, int step, float tcs[4]) {
  const float two_pi = (float)(2.0 * M_PI);
  const float af = two_pi * (float)step / (float)N, ab = two_pi * (float)(N - 1 - step) / (float)N;
  tcs[0] = cosf(af); tcs[1] = sinf(af); tcs[2] = cosf(ab); tcs[3] = sinf(ab);
}

// forward of one network call with everything taped, then the sub-update
static int taped_call(const l2hmc_gauge_plan* p, const l2hmc_dense_net* net, const NetTape& t, int call, int mode,
                      const float* a, const float* b, const float* cm_f, const float* cm_b, const float* state,
                      const float* keep_f, const float* keep_b, const int* dir, const float tcs[4], int64_t rows,
                      float* x, float* v, const TrainWs& w, hipStream_t s) {
  const int D = net->D, H = net->H;
  const size_t cr = (size_t)call * rows;
  float* in = t.in + cr * 2 * D;
  float* h1 = t.h1 + cr * H;
  float* h2 = t.h2 + cr * H;
  float* stq = t.stq + cr * 3 * D;
  const unsigned rgrid = (unsigned)ceil_div(rows, 4);
  hipLaunchKernelGGL(tape_in_kernel, dim3(rgrid), dim3(256), 0, s, a, b, cm_f, cm_b, dir, state, rows, D, in,
                     t.st + cr * D);
  L2HMC_CHECK_LAUNCH("tape_in");
  const int Kin = net->Ka + net->Kb;
  const float* first_in = in;
  if (p->flags & L2HMC_PLAN_CONV3D) {
    // conv_net.py:251-262 on the taped (already masked) inputs; the features are taped for the first layer's
    // weight gradient
    const l2hmc_conv3d_front* f = (net == &p->xnet) ? &p->xfront : &p->vfront;
    float* feat = t.feat + cr * Kin;
    ConvFrontArgs c{};
    c.T = p->T; c.X = p->X; c.F = f->F;
    c.in[0] = in; c.in[1] = in + D; c.ldi = 2 * D; c.dir = nullptr;
    c.w1[0] = f->w1_a; c.b1[0] = f->b1_a; c.w2[0] = f->w2_a; c.b2[0] = f->b2_a;
    c.w1[1] = f->w1_b; c.b1[1] = f->b1_b; c.w2[1] = f->w2_b; c.b2[1] = f->b2_b;
    c.out[0] = feat; c.out[1] = feat + net->Ka; c.ldo = Kin; c.rows = rows;
    if (int e = launch_conv3d_front(c, s)) return e;
    first_in = feat;
  }
  GemmReluArgs l1{};
  l1.A1 = first_in; l1.lda1 = Kin; l1.K1 = Kin;
  l1.dir = dir;
  l1.Wt = net->w1_t; l1.K = Kin; l1.N = H;
  l1.bias = net->b1; l1.wt0 = net->wt; l1.wt1 = net->wt + H;
  l1.tc_f = tcs[0]; l1.ts_f = tcs[1]; l1.tc_b = tcs[2]; l1.ts_b = tcs[3];
  l1.out = h1; l1.ldo = H; l1.rows = rows;
  if (int e = launch_gemm_relu(l1, s)) return e;
  GemmReluArgs l2{};
  l2.A1 = h1; l2.lda1 = H; l2.K1 = H;
  l2.Wt = net->wh_t; l2.K = H; l2.N = H;
  l2.bias = net->bh; l2.out = h2; l2.ldo = H; l2.rows = rows;
  if (int e = launch_gemm_relu(l2, s)) return e;
  HeadsArgs h{};
  h.A = h2; h.lda = H; h.K = H;
  h.Wt = net->whd_t; h.bhd = net->bhd; h.cs = net->coeff_s; h.cq = net->coeff_q;
  h.q_tanh = net->q_tanh; h.D = D; h.rows = rows; h.mode = kHeadsMaterialise;
  h.S = stq; h.T = stq + (size_t)rows * D; h.Q = stq + 2 * (size_t)rows * D;
  if (int e = launch_heads(h, s)) return e;
  hipLaunchKernelGGL(train_update_kernel, dim3(rgrid), dim3(256), 0, s, mode, stq, (int64_t)rows * D, w.g, keep_f,
                     keep_b, dir, p->eps, rows, D, x, v, w.ld);
  L2HMC_CHECK_LAUNCH("train_update");
  return L2HMC_OK;
}

// backward of one network call: heads' pre-activation gradients (in t.dout) -> t.d2, t.d1, w.din
static int call_backward_conv(const l2hmc_gauge_plan* p, const l2hmc_dense_net* net, const NetTape& t, int call,
                              int64_t rows, const TrainWs& w, hipStream_t s);

static int call_backward_data(const l2hmc_gauge_plan* p, const l2hmc_dense_net* net, const NetTape& t, int call,
                              int64_t rows, const TrainWs& w, hipStream_t s) {
  const int D = net->D, H = net->H, Kin = net->Ka + net->Kb;
  const bool conv = (p->flags & L2HMC_PLAN_CONV3D) != 0;
  const size_t cr = (size_t)call * rows;
  // d2 = (dout . Whd) gated by h2 > 0:   B operand [H][3D] = transpose of whd_t [3D][H]
  GemmReluArgs g2{};
  g2.kind = 3;
  g2.A1 = t.dout + cr * 3 * D; g2.lda1 = 3 * D; g2.K1 = 3 * D; g2.K = 3 * D;
  g2.Wt = t.whd_n; g2.N = H;
  g2.gate = t.h2 + cr * H; g2.ldg = H;
  g2.out = t.d2 + cr * H; g2.ldo = H; g2.rows = rows;
  if (int e = launch_gemm_relu(g2, s)) return e;
  // d1 = (d2 . Wh) gated by h1 > 0:      B operand [H_in][H_out] = transpose of wh_t [out][in]
  GemmReluArgs g1{};
  g1.kind = 3;
  g1.A1 = t.d2 + cr * H; g1.lda1 = H; g1.K1 = H; g1.K = H;
  g1.Wt = t.wh_n; g1.N = H;
  g1.gate = t.h1 + cr * H; g1.ldg = H;
  g1.out = t.d1 + cr * H; g1.ldo = H; g1.rows = rows;
  if (int e = launch_gemm_relu(g1, s)) return e;
  // din = d1 . W1:                        B operand [2D][H] = transpose of w1_t [H][2D]
  GemmReluArgs g0{};
  g0.kind = 4;
  g0.A1 = t.d1 + cr * H; g0.lda1 = H; g0.K1 = H; g0.K = H;
  g0.Wt = t.w1_n; g0.N = Kin;
  g0.out = conv ? w.dfeat : w.din; g0.ldo = Kin; g0.rows = rows;
  if (int e = launch_gemm_relu(g0, s)) return e;
  if (!conv) return L2HMC_OK;
  return call_backward_conv(p, net, t, call, rows, w, s);
}

// d loss / d features (w.dfeat) through the conv front-end to the raw inputs (w.din) and the filters (t.conv_part)
static int call_backward_conv(const l2hmc_gauge_plan* p, const l2hmc_dense_net* net, const NetTape& t, int call,
                              int64_t rows, const TrainWs& w, hipStream_t s) {
  const int D = net->D, Kin = net->Ka + net->Kb;
  const size_t cr = (size_t)call * rows;
  const l2hmc_conv3d_front* f = (net == &p->xnet) ? &p->xfront : &p->vfront;
  ConvBwdArgs b{};
  b.T = p->T; b.X = p->X; b.F = f->F;
  b.in = t.in + cr * 2 * D; b.ldi = 2 * D;
  b.dfeat = w.dfeat; b.ldf = Kin;
  b.w1[0] = f->w1_a; b.b1[0] = f->b1_a; b.w2[0] = f->w2_a; b.b2[0] = f->b2_a;
  b.w1[1] = f->w1_b; b.b1[1] = f->b1_b; b.w2[1] = f->w2_b; b.b2[1] = f->b2_b;
  b.din = w.din; b.ldd = 2 * D; b.part = t.conv_part; b.rows = rows;
  return launch_conv3d_front_bwd(b, s);
}

static ColsumArgs colsum_args(const float* src, int64_t Rt, int n, int64_t rows, int nsteps, const int* dir,
                              int timed, float* part) {
  ColsumArgs c{};
  c.src = src; c.Rt = Rt; c.n = n; c.rows = rows; c.nsteps = nsteps; c.dir = dir; c.timed = timed;
  c.S = colsum_chunks(Rt);
  c.chunk = ceil_div(Rt, c.S);
  c.ncg = (int)ceil_div(n, 1024);
  c.part = part;
  return c;
}

// partials [S][3][n] -> the one or three planes, each summed in a fixed order straight into its gradient slot
static int colsum_finish(const ColsumArgs& c, float* out_plain, float* out_cos, float* out_sin, hipStream_t s) {
  const PlaneOuts outs{{out_plain, out_cos, out_sin}};
  hipLaunchKernelGGL(reduce_planes_kernel, dim3((unsigned)ceil_div(c.n, 64), c.timed ? 3 : 1), dim3(256), 0, s,
                     c.part, c.S, c.n, outs);
  L2HMC_CHECK_LAUNCH("reduce_planes");
  return L2HMC_OK;
}

// out = P^T . Q; with b_plain != NULL also the column sums of P -- the bias gradient of the same layer (b_cos /
// b_sin: weighted by each taped row's time input, the t_layer kernel's gradient) -- as extra workgroups of the
// same launch (see GemmTnArgs::cs)
static int gemm_tn(const float* P, int M, const float* Q, int N, int64_t R, float* out, const TrainWs& w,
                   hipStream_t s, float* b_plain = nullptr, float* b_cos = nullptr, float* b_sin = nullptr,
                   int64_t rows = 0, int nsteps = 0, const int* dir = nullptr) {
  GemmTnArgs a{};
  a.P = P; a.ldp = M; a.M = M; a.Q = Q; a.ldq = N; a.N = N; a.R = R;
  a.mt = (int)ceil_div(M, 128); a.nt = (int)ceil_div(N, 128);
  const int splits = tn_splits(a.mt, a.nt, R);
  a.chunk = (int64_t)align_up((size_t)ceil_div(R, splits), 16);
  a.part = w.part;
  a.nprod = a.mt * a.nt * splits;
  const bool timed = b_cos != nullptr;
  if (b_plain) {
    L2HMC_REQUIRE(M % 4 == 0, "gemm_tn: column sums need a width (%d) that is a multiple of 4", M);
    L2HMC_REQUIRE(!timed || (b_sin && rows > 0 && nsteps > 0 && 2 * nsteps + 256 * 12 <= 2 * 2 * 16 * 160),
                  "gemm_tn: bad column-sum arguments");
    a.cs = colsum_args(P, R, M, rows, nsteps, dir, timed ? 1 : 0, w.part + (size_t)splits * M * N);
  }
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(a.nprod + a.cs.ncg * a.cs.S), dim3(256), 0, s, a);
  L2HMC_CHECK_LAUNCH("gemm_tn");
  const int64_t count = (int64_t)M * N;
  if (b_plain) {      // product partials and column-sum planes in one launch
    const PlaneOuts outs{{b_plain, b_cos, b_sin}};
    const int nb1 = (int)ceil_div(count, 64), nb2 = (int)ceil_div(M, 64) * (timed ? 3 : 1);
    hipLaunchKernelGGL(reduce_product_and_planes_kernel, dim3((unsigned)(nb1 + nb2)), dim3(256), 0, s, w.part, splits,
                       count, out, nb1, a.cs.part, a.cs.S, M, outs);
    L2HMC_CHECK_LAUNCH("reduce_product_and_planes");
    return L2HMC_OK;
  }
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)ceil_div(count, 64)), dim3(256), 0, s, w.part, splits,
                     count, out);
  L2HMC_CHECK_LAUNCH("reduce_partials");
  return L2HMC_OK;
}

// stand-alone column sums (deltas whose weight gradient is not a TN product of this file)
static int colsum(const float* src, int64_t Rt, int n, int64_t rows, int nsteps, const int* dir, int timed,
                  float* out_plain, float* out_cos, float* out_sin, const TrainWs& w, hipStream_t s) {
  L2HMC_REQUIRE(n % 4 == 0, "colsum: width %d must be a multiple of 4", n);
  const ColsumArgs c = colsum_args(src, Rt, n, rows, nsteps, dir, timed, w.part);
  hipLaunchKernelGGL(colsum_kernel, dim3(c.ncg, c.S), dim3(256), sizeof(float) * (2 * nsteps + 256 * 12), s, c);
  L2HMC_CHECK_LAUNCH("colsum");
  return colsum_finish(c, out_plain, out_cos, out_sin, s);
}

}  // namespace l2hmc

using namespace l2hmc;

extern "C" size_t l2hmc_gauge_train_ws_bytes(const l2hmc_gauge_plan* plan, int64_t rows) {
  if (!plan || rows <= 0 || plan->hmc) return 0;
  return carve_train_ws(plan, rows, nullptr).bytes;
}

extern "C" int l2hmc_gauge_train_forward(const l2hmc_gauge_plan* plan, float beta, const float* x0, const float* v0,
                                         const int32_t* dir, int64_t rows, float* x_out, float* v_out,
                                         float* sumlogdet, float* p_accept, void* ws, size_t ws_bytes,
                                         l2hmc_stream_t stream) {
  if (int e = check_train_plan(plan)) return e;
  L2HMC_REQUIRE(rows >= 0, "train_forward: rows < 0");
  if (rows == 0) return L2HMC_OK;
  L2HMC_REQUIRE(x0 && v0 && x_out && v_out && ws, "train_forward: NULL pointer");
  const TrainWs w = carve_train_ws(plan, rows, ws);
  if (ws_bytes < w.bytes) {
    set_error("train_forward: workspace %zu < %zu bytes", ws_bytes, w.bytes);
    return L2HMC_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const int D = 2 * plan->T * plan->X, N = plan->num_steps;
  const size_t nb = sizeof(float) * (size_t)rows * D;
  if ((x_out != x0 && hipMemcpyAsync(x_out, x0, nb, hipMemcpyDeviceToDevice, s) != hipSuccess) ||
      (v_out != v0 && hipMemcpyAsync(v_out, v0, nb, hipMemcpyDeviceToDevice, s) != hipSuccess) ||
      hipMemsetAsync(w.ld, 0, sizeof(float) * rows, s) != hipSuccess) {
    set_error("train_forward: copy failed");
    return L2HMC_ERR_HIP;
  }
  hipLaunchKernelGGL(invert_mask_train_kernel, dim3((unsigned)ceil_div(N * D, 256)), dim3(256), 0, s, plan->masks,
                     w.mask_inv, N * D);
  L2HMC_CHECK_LAUNCH("invert_mask");
  float* x = x_out;
  float* v = v_out;
  if (fused_train_forward_supported(plan)) {
    // whole-trajectory kernel (fused_traj.hip) writing the same tape: one launch instead of ~6 per network call
    const FusedTape tx{w.x.in, w.x.h1, w.x.h2, w.x.stq, w.x.st, w.x.feat, w.x.gate}, tv{w.v.in, w.v.h1, w.v.h2, w.v.stq, w.v.st, w.v.feat, w.v.gate};
    return launch_fused_trajectory(plan, beta, 0, N, x, v, dir, rows, x, v, sumlogdet, 0, p_accept, s, 0, 0, &tx, &tv);
  }
  if (int e = launch_u1_action_force(x, rows, plan->T, plan->X, beta, w.act0, nullptr, nullptr, nullptr, s)) return e;
  if (int e = l2hmc_kinetic_energy(v, rows, D, w.kin0, stream)) return e;
  for (int step = 0; step < N; ++step) {
    float tcs[4];
    step_times(N, step, tcs);
    const int sf = step, sb = N - 1 - step;
    const float* m_f = plan->masks + (size_t)sf * D;
    const float* m_b = plan->masks + (size_t)sb * D;
    const float* mi_f = w.mask_inv + (size_t)sf * D;
    const float* mi_b = w.mask_inv + (size_t)sb * D;
    for (int half = 0; half < 2; ++half) {
      if (half == 1) {
        for (int sub = 0; sub < 2; ++sub) {
          // forward rows (gauge_dynamics.py:428-438): (m, m_inv) then (m_inv, m); backward rows (:466-476) swapped
          const float* kf = sub == 0 ? m_f : mi_f;
          const float* kb = sub == 0 ? mi_b : m_b;
          if (int e = taped_call(plan, &plan->xnet, w.x, 2 * step + sub, 2, v, x, kf, kb, x, kf, kb, dir, tcs, rows,
                                 x, v, w, s))
            return e;
        }
      }
      if (int e = launch_u1_action_force(x, rows, plan->T, plan->X, beta, nullptr, w.g, nullptr, nullptr, s)) return e;
      if (int e = taped_call(plan, &plan->vnet, w.v, 2 * step + half, 1, x, w.g, nullptr, nullptr, v, nullptr,
                             nullptr, dir, tcs, rows, x, v, w, s))
        return e;
    }
  }
  if (int e = launch_u1_action_force(x, rows, plan->T, plan->X, beta, w.act1, nullptr, nullptr, nullptr, s)) return e;
  if (int e = l2hmc_kinetic_energy(v, rows, D, w.kin1, stream)) return e;
  hipLaunchKernelGGL(train_accept_kernel, dim3((unsigned)ceil_div(rows, 256)), dim3(256), 0, s, w.act0, w.kin0,
                     w.act1, w.kin1, w.ld, beta, rows, sumlogdet, p_accept);
  L2HMC_CHECK_LAUNCH("train_accept");
  return L2HMC_OK;
}

static int train_backward_impl(const l2hmc_gauge_plan* plan, float beta, const int32_t* dir, int64_t rows,
                               float* dx, float* dv, const float* dlogdet,
                               const l2hmc_dense_grads* gx, const l2hmc_dense_grads* gv,
                               const l2hmc_conv3d_grads* gxf, const l2hmc_conv3d_grads* gvf, float* deps,
                               void* ws, size_t ws_bytes, l2hmc_stream_t stream, l2hmc_bucket_fn on_bucket,
                               void* user) {
  if (int e = check_train_plan(plan)) return e;
  L2HMC_REQUIRE(rows > 0 && dx && dv && dlogdet && gx && gv && deps && ws, "train_backward: bad arguments");
  const bool conv = (plan->flags & L2HMC_PLAN_CONV3D) != 0;
  L2HMC_REQUIRE(!conv || (gxf && gvf), "train_backward: ConvNet3D plans need the front-end gradient structs");
  const TrainWs w = carve_train_ws(plan, rows, ws);
  if (ws_bytes < w.bytes) {
    set_error("train_backward: workspace %zu < %zu bytes", ws_bytes, w.bytes);
    return L2HMC_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const int D = 2 * plan->T * plan->X, N = plan->num_steps, C = 2 * N;
  const int64_t nblk = upd_blocks(rows);
  if (hipMemsetAsync(w.zero_base, 0, w.zero_bytes, s) != hipSuccess) {
    set_error("train_backward: memset failed");
    return L2HMC_ERR_HIP;
  }
  if (conv) {
    const l2hmc_conv3d_front* fr[2] = {&plan->xfront, &plan->vfront};
    const NetTape* tp[2] = {&w.x, &w.v};
    for (int k = 0; k < 2; ++k) {
      const size_t n = (size_t)ceil_div(rows, conv3d_cpw(plan->T, plan->X, fr[k]->F)) * 2 *
                       conv3d_bwd_part_floats(fr[k]->F);
      if (hipMemsetAsync(tp[k]->conv_part, 0, sizeof(float) * n, s) != hipSuccess) {
        set_error("train_backward: memset failed");
        return L2HMC_ERR_HIP;
      }
    }
  }
  const bool fused_bwd = fused_train_supported(plan);     // the forward call took the same branch and left the relu masks
  // ConvNet3D plans with a single-call trunk kernel: element-wise phase + the three backward-data products of a call
  // in ONE launch (fused_train.hip) instead of four
  const bool trunk = !fused_bwd && conv && trunk_bwd_supported(&plan->xnet) && trunk_bwd_supported(&plan->vnet);
  if (!fused_bwd && !trunk) {
    // weights as the layered backward-data products read them (k = output unit contiguous), once per pass
    // (the fused reverse pass packs its own fragment-ordered images instead)
    const l2hmc_dense_net* nets[2] = {&plan->xnet, &plan->vnet};
    const NetTape* tapes[2] = {&w.x, &w.v};
    for (int k = 0; k < 2; ++k) {
      const int H = nets[k]->H;
      const dim3 tb(256);
      hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)ceil_div(H, 32), (unsigned)ceil_div(3 * D, 32)), tb, 0, s,
                         nets[k]->whd_t, 3 * D, H, tapes[k]->whd_n);
      hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)ceil_div(H, 32), (unsigned)ceil_div(H, 32)), tb, 0, s,
                         nets[k]->wh_t, H, H, tapes[k]->wh_n);
      const int Kin = nets[k]->Ka + nets[k]->Kb;
      hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)ceil_div(Kin, 32), (unsigned)ceil_div(H, 32)), tb, 0, s,
                         nets[k]->w1_t, H, Kin, tapes[k]->w1_n);
      L2HMC_CHECK_LAUNCH("transpose");
    }
  }
  const unsigned rgrid = (unsigned)ceil_div(rows, 4);
  const size_t vlds = sizeof(float) * 4 * (size_t)(2 * D + D / 2);
  L2HMC_REQUIRE(vlds <= 160 * 1024, "train_backward: lattice too large for the force-Hessian kernel's LDS");
  if (vlds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vnet_in_bwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)vlds);
  const int ublock = (int)hmin(256, (int64_t)align_up((size_t)D, 64));
  if (trunk) {
    if (int e = launch_trunk_bwd_pack(&plan->xnet, w.x.bpack, s)) return e;
    if (int e = launch_trunk_bwd_pack(&plan->vnet, w.v.bpack, s)) return e;
  }
  auto upd = [&](const l2hmc_dense_net* net, const NetTape& t, int call, int mode, const float* kf,
                 const float* kb) -> int {
    const size_t cr = (size_t)call * rows;
    if (trunk) {
      const int H = net->H;
      TrunkBwdArgs a{};
      a.mode = mode; a.eps = plan->eps; a.pk = t.bpack;
      a.cs = net->coeff_s; a.cq = net->coeff_q; a.q_tanh = net->q_tanh;
      a.keep_f = kf; a.keep_b = kb; a.dir = dir; a.rows = rows;
      a.stq = t.stq + cr * 3 * D; a.plane = (int64_t)rows * D;
      a.st = t.st + cr * D; a.in = t.in + cr * 2 * D; a.h1 = t.h1 + cr * H; a.h2 = t.h2 + cr * H;
      a.dld = dlogdet; a.dx = dx; a.dv = dv; a.dg = w.dg;
      a.dout = t.dout + cr * 3 * D; a.d2 = t.d2 + cr * H; a.d1 = t.d1 + cr * H;
      a.dfeat = w.dfeat;
      a.dcs_part = t.dcs_part; a.dcq_part = t.dcq_part; a.deps_part = w.eps_part;
      if (int e = launch_trunk_bwd(a, s)) return e;
      return call_backward_conv(plan, net, t, call, rows, w, s);
    }
    UpdBwdArgs a{};
    a.mode = mode; a.rows = rows; a.D = D; a.eps = plan->eps;
    a.stq = t.stq + cr * 3 * D; a.plane = (int64_t)rows * D;
    a.st = t.st + cr * D; a.in = t.in + cr * 2 * D;
    a.keep_f = kf; a.keep_b = kb; a.dir = dir;
    a.cs = net->coeff_s; a.cq = net->coeff_q; a.q_tanh = net->q_tanh;
    a.dld = dlogdet; a.dx = dx; a.dv = dv; a.dg = w.dg;
    a.dout = t.dout + cr * 3 * D;
    a.dcs_part = t.dcs_part; a.dcq_part = t.dcq_part; a.deps_part = w.eps_part;
    hipLaunchKernelGGL(update_bwd_kernel, dim3((unsigned)nblk), dim3(ublock), 0, s, a);
    L2HMC_CHECK_LAUNCH("update_bwd");
    return call_backward_data(plan, net, t, call, rows, w, s);
  };
  int64_t ncoef = nblk;            // workgroups that wrote coefficient / step-size partials
  if (fused_bwd) {
    // one launch for the whole reverse data path (fused_train.hip)
    const FusedTape tx{w.x.in, w.x.h1, w.x.h2, w.x.stq, w.x.st, w.x.feat, w.x.gate}, tv{w.v.in, w.v.h1, w.v.h2, w.v.stq, w.v.st, w.v.feat, w.v.gate};
    float* const dxs_[3] = {w.x.dout, w.x.d2, w.x.d1};
    float* const dvs_[3] = {w.v.dout, w.v.d2, w.v.d1};
    float* const coef[4] = {w.x.dcs_part, w.x.dcq_part, w.v.dcs_part, w.v.dcq_part};
    if (int e = launch_fused_train_backward(plan, beta, dir, rows, dx, dv, dlogdet, tx, tv, dxs_, dvs_, w.x.bpack,
                                            w.v.bpack, coef, w.eps_part, s))
      return e;
    ncoef = ceil_div(rows, 16);
  } else {
    for (int step = N - 1; step >= 0; --step) {
      const int sf = step, sb = N - 1 - step;
      const float* m_f = plan->masks + (size_t)sf * D;
      const float* m_b = plan->masks + (size_t)sb * D;
      const float* mi_f = w.mask_inv + (size_t)sf * D;
      const float* mi_b = w.mask_inv + (size_t)sb * D;
      for (int half = 1; half >= 0; --half) {
        const int vc = 2 * step + half;
        if (int e = upd(&plan->vnet, w.v, vc, 1, nullptr, nullptr)) return e;
        hipLaunchKernelGGL(vnet_in_bwd_kernel, dim3(rgrid), dim3(256), vlds, s, w.din,
                           w.dg, w.v.in + (size_t)vc * rows * 2 * D, beta, plan->T, plan->X, rows, dx);
        L2HMC_CHECK_LAUNCH("vnet_in_bwd");
        if (half == 1) {
          for (int sub = 1; sub >= 0; --sub) {
            const float* kf = sub == 0 ? m_f : mi_f;
            const float* kb = sub == 0 ? mi_b : m_b;
            if (int e = upd(&plan->xnet, w.x, 2 * step + sub, 2, kf, kb)) return e;
            hipLaunchKernelGGL(xnet_in_bwd_kernel, dim3(rgrid), dim3(256), 0, s, w.din, kf, kb, dir, rows, D, dx, dv);
            L2HMC_CHECK_LAUNCH("xnet_in_bwd");
          }
        }
      }
    }
  }
  // ---- weight gradients from the tape: one contraction over all calls x rows per matrix
  const l2hmc_dense_net* nets[2] = {&plan->xnet, &plan->vnet};
  const NetTape* tapes[2] = {&w.x, &w.v};
  const l2hmc_dense_grads* gr[2] = {gx, gv};
  const int64_t Rt = (int64_t)C * rows;
  for (int k = 0; k < 2; ++k) {
    const int H = nets[k]->H;
    const NetTape& t = *tapes[k];
    const l2hmc_dense_grads* g = gr[k];
    L2HMC_REQUIRE(g->w1_t && g->wt && g->b1 && g->wh_t && g->bh && g->whd_t && g->bhd && g->coeff_s && g->coeff_q,
                  "train_backward: NULL gradient pointer");
    const int Kin = nets[k]->Ka + nets[k]->Kb;
    // three gradient buckets per network, each finished (weights, then the biases that go with them) before the
    // next product starts, so a caller can put bucket b on the wire while bucket b + 1 is being computed:
    //   3k + 0: [w1_t | wt | b1]      3k + 1: [wh_t | bh]      3k + 2: [whd_t | bhd | coeff_s | coeff_q]
    if (2 * N + 256 * 12 <= 2 * 2 * 16 * 160) {
      // the bias gradients (column sums of the deltas) ride in the launches of the weight-gradient products
      if (int e = gemm_tn(t.d1, H, conv ? t.feat : t.in, Kin, Rt, g->w1_t, w, s, g->b1, g->wt, g->wt + H, rows, N, dir))
        return e;
      if (on_bucket) on_bucket(user, 3 * k + 0);
      if (int e = gemm_tn(t.d2, H, t.h1, H, Rt, g->wh_t, w, s, g->bh)) return e;
      if (on_bucket) on_bucket(user, 3 * k + 1);
      if (int e = gemm_tn(t.dout, 3 * D, t.h2, H, Rt, g->whd_t, w, s, g->bhd)) return e;
    } else {
      if (int e = gemm_tn(t.d1, H, conv ? t.feat : t.in, Kin, Rt, g->w1_t, w, s)) return e;
      if (int e = colsum(t.d1, Rt, H, rows, N, dir, 1, g->b1, g->wt, g->wt + H, w, s)) return e;
      if (on_bucket) on_bucket(user, 3 * k + 0);
      if (int e = gemm_tn(t.d2, H, t.h1, H, Rt, g->wh_t, w, s)) return e;
      if (int e = colsum(t.d2, Rt, H, rows, N, dir, 0, g->bh, nullptr, nullptr, w, s)) return e;
      if (on_bucket) on_bucket(user, 3 * k + 1);
      if (int e = gemm_tn(t.dout, 3 * D, t.h2, H, Rt, g->whd_t, w, s)) return e;
      if (int e = colsum(t.dout, Rt, 3 * D, rows, N, dir, 0, g->bhd, nullptr, nullptr, w, s)) return e;
    }
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)ceil_div(D, 64)), dim3(256), 0, s, t.dcs_part,
                       (int)ncoef, (int64_t)D, g->coeff_s);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)ceil_div(D, 64)), dim3(256), 0, s, t.dcq_part,
                       (int)ncoef, (int64_t)D, g->coeff_q);
    L2HMC_CHECK_LAUNCH("reduce_partials");
    if (on_bucket) on_bucket(user, 3 * k + 2);
  }
  if (conv) {
    const l2hmc_conv3d_front* fr[2] = {&plan->xfront, &plan->vfront};
    const l2hmc_conv3d_grads* gf[2] = {gxf, gvf};
    for (int k = 0; k < 2; ++k) {
      const int F = fr[k]->F;
      const size_t ps = conv3d_bwd_part_floats(F);
      const int nwg = (int)ceil_div(rows, conv3d_cpw(plan->T, plan->X, F));
      const l2hmc_conv3d_grads* g = gf[k];
      L2HMC_REQUIRE(g->w1_a && g->b1_a && g->w2_a && g->b2_a && g->w1_b && g->b1_b && g->w2_b && g->b2_b,
                    "train_backward: NULL conv gradient pointer");
      hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)ceil_div(2 * ps, 64)), dim3(256), 0, s,
                         tapes[k]->conv_part, nwg, (int64_t)(2 * ps), w.part);
      L2HMC_CHECK_LAUNCH("reduce_partials");
      float* dst[8] = {g->w1_a, g->b1_a, g->w2_a, g->b2_a, g->w1_b, g->b1_b, g->w2_b, g->b2_b};
      const size_t len[4] = {(size_t)18 * F, (size_t)F, (size_t)16 * F * F, (size_t)2 * F};
      size_t off = 0;
      for (int q = 0; q < 8; ++q) {
        if (hipMemcpyAsync(dst[q], w.part + off, sizeof(float) * len[q & 3], hipMemcpyDeviceToDevice, s) != hipSuccess) {
          set_error("train_backward: copy failed");
          return L2HMC_ERR_HIP;
        }
        off += len[q & 3];
      }
    }
  }
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, s, w.eps_part, (int)ncoef, (int64_t)1, deps);
  L2HMC_CHECK_LAUNCH("reduce_partials");
  if (on_bucket) on_bucket(user, L2HMC_GRAD_BUCKET_REST);
  return L2HMC_OK;
}

extern "C" int l2hmc_gauge_train_backward(const l2hmc_gauge_plan* plan, float beta, const int32_t* dir, int64_t rows,
                                          float* dx, float* dv, const float* dlogdet,
                                          const l2hmc_dense_grads* gx, const l2hmc_dense_grads* gv,
                                          const l2hmc_conv3d_grads* gxf, const l2hmc_conv3d_grads* gvf, float* deps,
                                          void* ws, size_t ws_bytes, l2hmc_stream_t stream) {
  return train_backward_impl(plan, beta, dir, rows, dx, dv, dlogdet, gx, gv, gxf, gvf, deps, ws, ws_bytes, stream,
                             nullptr, nullptr);
}

extern "C" int l2hmc_gauge_train_backward_buckets(const l2hmc_gauge_plan* plan, float beta, const int32_t* dir,
                                                  int64_t rows, float* dx, float* dv, const float* dlogdet,
                                                  const l2hmc_dense_grads* gx, const l2hmc_dense_grads* gv,
                                                  const l2hmc_conv3d_grads* gxf, const l2hmc_conv3d_grads* gvf,
                                                  float* deps, void* ws, size_t ws_bytes, l2hmc_stream_t stream,
                                                  l2hmc_bucket_fn on_bucket, void* user) {
  return train_backward_impl(plan, beta, dir, rows, dx, dv, dlogdet, gx, gv, gxf, gvf, deps, ws, ws_bytes, stream,
                             on_bucket, user);
}

extern "C" int l2hmc_gauge_loss_backward(int32_t T, int32_t X, float beta, const float* x0, const float* xN,
                                         const float* vN, const float* p, int64_t B, int32_t metric,
                                         float loss_scale, float aux_weight, float std_weight, float charge_weight,
                                         float inv_count, float* terms, float* dxN, float* dvN, float* dlogdet,
                                         l2hmc_stream_t stream) {
  L2HMC_REQUIRE(T > 0 && X > 0 && B >= 0 && metric >= 0 && metric <= 4, "loss_backward: bad arguments");
  if (B == 0) return L2HMC_OK;
  L2HMC_REQUIRE(x0 && xN && vN && p && dxN && dvN && dlogdet, "loss_backward: NULL pointer");
  LossBwdArgs a{};
  a.T = T; a.X = X; a.B = B; a.beta = beta; a.x0 = x0; a.xN = xN; a.vN = vN; a.p = p; a.metric = metric;
  a.loss_scale = loss_scale; a.aux_weight = aux_weight; a.std_weight = std_weight; a.charge_weight = charge_weight;
  a.inv_count = inv_count; a.terms = terms; a.dxN = dxN; a.dvN = dvN; a.dld = dlogdet;
  const size_t lds = sizeof(float) * (size_t)(2 * T * X + 2 * T * X);
  L2HMC_REQUIRE(lds <= 64 * 1024, "loss_backward: lattice too large for one workgroup's LDS");
  hipLaunchKernelGGL(loss_bwd_kernel, dim3((unsigned)B), dim3(64), lds, (hipStream_t)stream, a);
  L2HMC_CHECK_LAUNCH("loss_bwd");
  return L2HMC_OK;
}

extern "C" int l2hmc_grad_sumsq(const float* g, int64_t n, int64_t tri_lo, int64_t tri_hi, float* out,
                                int32_t accumulate, l2hmc_stream_t stream) {
  L2HMC_REQUIRE(n >= 0 && out, "grad_sumsq: bad arguments");
  L2HMC_REQUIRE(n == 0 || g, "grad_sumsq: NULL gradient");
  hipLaunchKernelGGL(sumsq_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, g, n, tri_lo, tri_hi, out, accumulate);
  L2HMC_CHECK_LAUNCH("sumsq");
  return L2HMC_OK;
}

extern "C" int l2hmc_adam_step(float* w, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1,
                               float beta2, float eps, const float* gnorm_sq, float clip, int64_t tri_lo,
                               int64_t tri_hi, l2hmc_stream_t stream) {
  L2HMC_REQUIRE(n >= 0, "adam_step: n < 0");
  if (n == 0) return L2HMC_OK;
  L2HMC_REQUIRE(w && g && m && v, "adam_step: NULL pointer");
  L2HMC_REQUIRE(gnorm_sq == nullptr || clip > 0.f, "adam_step: clip value must be positive");
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, w, g, m, v, n,
                     lr_t, beta1, beta2, eps, gnorm_sq, clip, tri_lo, tri_hi);
  L2HMC_CHECK_LAUNCH("adam");
  return L2HMC_OK;
}
