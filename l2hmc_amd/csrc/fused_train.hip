// Whole-trajectory REVERSE pass for the lattice training step (gfx950), the counterpart of
// gauge_traj_fused_kernel: one launch walks every network call of
//   l2hmc/dynamics/gauge_dynamics.py:412-590 (leapfrog sub-updates) + network/generic_net.py:129-146
// backwards for a tile of 16 chain-rows per workgroup and produces what tf.gradients
// (gauge_model.py:825) needs from the data path: the pre-activation gradients of the three heads and of
// both hidden layers for every call (written to the delta tape the split-k weight-gradient products of
// train.hip contract afterwards), the per-column coefficient gradients and d loss / d eps.
//
// Same machine mapping as the forward kernel: d loss / d (x, v) of the 16 chains live in LDS for the
// whole pass; per call an element-wise phase differentiates the sub-update (reads S, T, Q and the
// consumed state from the forward tape), then three streamed MFMA products carry the head gradients
// back through the network -- delta2 = (dout . Whd) gated by h2, delta1 = (delta2 . Wh) gated by h1,
// din = delta1 . W1 -- with the TRANSPOSED weights pre-packed in the fragment order a wave consumes
// (pack_fused_bwd_kernel), so the weight stream per call is the same 2.36 MB as in the forward pass
// and the same L2 -> CU fabric roofline applies.  The force's Hessian-vector product (momentum calls)
// is the plaquette stencil with sin P -> cos P . P[u], chain-local in LDS.
// All reductions have a fixed order: results are reproducible.
#include "fused_common.h"
#include <stdlib.h>

namespace l2hmc {

// backward weight images, fragment order [wave][k-chunk][n-tile][lane][4] per section:
//   B1: K = 3D (head, d), N = H   value = whd_t[k][n]      delta2 = dout   . Whd
//   B2: K = H,            N = H   value = wh_t[k][n]       delta1 = delta2 . Wh
//   B3: K = H,            N = 2D  value = w1_t[k][n]       din    = delta1 . W1
__global__ void pack_fused_bwd_kernel(l2hmc_dense_net n, float* __restrict__ out) {
  const int D = n.D, H = n.H, K1 = n.Ka + n.Kb;
  const size_t P1 = (size_t)3 * D * H, P2 = (size_t)H * H, P3 = (size_t)H * K1;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < P1 + P2 + P3;
       i += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(i & 3), lane = (int)((i >> 2) & 63);
    int sec;
    size_t rest;
    if (i < P1) { sec = 0; rest = i >> 8; }
    else if (i < P1 + P2) { sec = 1; rest = (i - P1) >> 8; }
    else { sec = 2; rest = (i - P1 - P2) >> 8; }
    const int N = sec == 2 ? K1 : H, K = sec == 0 ? 3 * D : H;
    const int NT = N / (16 * kFWaves), KC = K / 16;
    const int t = (int)(rest % NT);
    rest /= NT;
    const int kc = (int)(rest % KC), w = (int)(rest / KC);
    const int col = (w * NT + t) * 16 + (lane & 15);
    const int k = kc * 16 + (lane >> 4) * 4 + j;
    const float* src = sec == 0 ? n.whd_t : sec == 1 ? n.wh_t : n.w1_t;
    out[i] = src[(size_t)k * N + col];
  }
}

struct FusedBwdArgs {
  int T, X, num_steps;
  float eps, beta;
  const float* masks;
  const float* pk_x; const float* pk_v;          // backward images
  const float* cs_x; const float* cq_x; const float* cs_v; const float* cq_v;
  int qtanh_x, qtanh_v;
  const int* dir; int64_t rows;
  FusedTape tx, tv;                              // forward tape (read only here)
  float* dout_x; float* d2_x; float* d1_x;       // delta tapes [calls][rows][3D | H | H]
  float* dout_v; float* d2_v; float* d1_v;
  float* dx; float* dv;                          // [rows][D], in/out
  const float* dld;                              // [rows]
  float* dcs_x; float* dcq_x; float* dcs_v; float* dcq_v;   // [workgroups][D]
  float* deps;                                   // [workgroups]
};

template <int D, int H>
struct FusedBwdCfg {
  static constexpr int SX = D + 8, SO = 3 * D + 8, SH = H + 8;
  static constexpr int NT1 = H / (16 * kFWaves), NT3 = 2 * D / (16 * kFWaves);
  static constexpr int KCO = 3 * D / 16, KCH = H / 16;
  static constexpr size_t P1 = (size_t)3 * D * H, P2 = (size_t)H * H, P3 = (size_t)H * 2 * D;
  static constexpr int SP = D / 2 + 4;
  static constexpr int LDS_FLOATS = 4 * kFM * SX + kFM * SO + 2 * kFM * SH + kFM * SP + 2 * D + 4 * D + 2 * kFM + 8;
};

template <int D, int H>
__global__ __launch_bounds__(kFThreads) void gauge_train_bwd_fused_kernel(FusedBwdArgs p) {
  using Cfg = FusedBwdCfg<D, H>;
  constexpr int SX = Cfg::SX, SO = Cfg::SO, SH = Cfg::SH, NT1 = Cfg::NT1, NT3 = Cfg::NT3, SP = Cfg::SP;
  constexpr int sites = D / 2;
  static_assert(NT1 <= 8, "relu masks are one 32-bit word per lane");
  static_assert(kTPC * 8 == D, "phase A maps 16 threads x 8 columns onto a chain");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* dxs = lds;                       // [16][SX] d loss / d x
  float* dvs = dxs + kFM * SX;            // [16][SX] d loss / d v
  float* xs = dvs + kFM * SX;             // [16][SX] x of the current momentum call (Hessian-vector product)
  float* us = xs + kFM * SX;              // [16][SX] d loss / d force
  float* dos = us + kFM * SX;             // [16][SO] head pre-activation gradients; later din [16][2D+8]
  float* d2s = dos + kFM * SO;            // [16][SH]
  float* d1s = d2s + kFM * SH;            // [16][SH]
  float* sp = d1s + kFM * SH;             // [16][SP] cos P . P[u]
  float* skm = sp + kFM * SP;             // [2][D] masks of this step (forward row, backward row)
  float* ec = skm + 2 * D;                // exp(cs_x) exp(cq_x) exp(cs_v) exp(cq_v)  [4][D]
  float* sdl = ec + 4 * D;                // [16] d loss / d sumlogdet
  int* sdir = reinterpret_cast<int*>(sdl + kFM);   // [16]
  float* red = reinterpret_cast<float*>(sdir + kFM);   // [8]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, r = lane & 15;
  const int64_t row0 = (int64_t)blockIdx.x * kFM;
  const int nrow = (int)min((int64_t)kFM, p.rows - row0);
  const float eps = p.eps;
  const int fc = tid / kTPC, fl = tid % kTPC;      // chain, lane-in-chain
  const int c0 = fl * 8;                           // this thread's 8 columns in the element-wise phases
  const bool live = fc < nrow;
  const int T = p.T, X = p.X, N = p.num_steps;
  const int xsh = 31 - __clz(X);       // X is a power of two whenever this kernel is chosen (fused_plan_supported)

  for (int i = tid; i < kFM * (D / 4); i += kFThreads) {
    const int rr = i / (D / 4), c4 = (i - rr * (D / 4)) * 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    if (rr < nrow) {
      a = *reinterpret_cast<const f32x4*>(p.dx + (row0 + rr) * D + c4);
      b = *reinterpret_cast<const f32x4*>(p.dv + (row0 + rr) * D + c4);
    }
    *reinterpret_cast<f32x4*>(dxs + rr * SX + c4) = a;
    *reinterpret_cast<f32x4*>(dvs + rr * SX + c4) = b;
  }
  for (int i = tid; i < D; i += kFThreads) {
    ec[i] = expf(p.cs_x[i]);
    ec[D + i] = expf(p.cq_x[i]);
    ec[2 * D + i] = expf(p.cs_v[i]);
    ec[3 * D + i] = expf(p.cq_v[i]);
  }
  if (tid < kFM) {
    sdl[tid] = tid < nrow ? p.dld[row0 + tid] : 0.f;
    sdir[tid] = (tid < nrow && p.dir) ? p.dir[row0 + tid] : 0;
  }
  __syncthreads();
  const int d = sdir[fc];
  const float dl = sdl[fc];

  float acs_x[8], acq_x[8], acs_v[8], acq_v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acs_x[k] = acq_x[k] = acs_v[k] = acq_v[k] = 0.f;
  float deps = 0.f;

  // [16][W] LDS rows -> a [rows][W] tape (coalesced 16-byte stores)
  auto tape_rows = [&](float* dst, size_t first_row, const float* src, int W, int stride) {
    for (int i = tid; i < kFM * (W / 4); i += kFThreads) {
      const int rr = i / (W / 4), c4 = (i - rr * (W / 4)) * 4;
      if (rr < nrow)
        tape_store(dst + (first_row + rr) * W + c4, *reinterpret_cast<const f32x4*>(src + rr * stride + c4));
    }
  };

  // The tape of a call (S, T, Q, consumed state, both inputs: 12 x 16 bytes per thread) does not depend on anything
  // this kernel computes, so the NEXT call's values are requested from HBM right after the current call's
  // element-wise phase has consumed its own and travel under the three streamed products.
  struct TapeRegs { f32x4 vS[2], vT[2], vQ[2], vst[2], va[2], vb[2]; };
  auto load_tape = [&](int cidx_, bool is_v_, TapeRegs& t) {
    const FusedTape& tp_ = is_v_ ? p.tv : p.tx;
    const size_t plane = (size_t)p.rows * D;
    const size_t tcr = (size_t)cidx_ * (size_t)p.rows + (size_t)row0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      t.vS[h] = t.vT[h] = t.vQ[h] = t.vst[h] = t.va[h] = t.vb[h] = z;
      if (live) {
        const float* sq = tp_.stq + (size_t)cidx_ * 3 * plane + ((size_t)row0 + fc) * D + c0 + 4 * h;
        t.vS[h] = tape_load(sq);
        t.vT[h] = tape_load(sq + plane);
        t.vQ[h] = tape_load(sq + 2 * plane);
        t.vst[h] = tape_load(tp_.st + (tcr + fc) * D + c0 + 4 * h);
        t.va[h] = tape_load(tp_.in + (tcr + fc) * (2 * D) + c0 + 4 * h);
        t.vb[h] = tape_load(tp_.in + (tcr + fc) * (2 * D) + D + c0 + 4 * h);
      }
    }
  };
  TapeRegs tnext;
  load_tape(2 * (N - 1) + 1, true, tnext);          // the first call of the reverse pass: step N-1, call 3

  for (int step = N - 1; step >= 0; --step) {
    const int sf = step, sb = N - 1 - step;
    __syncthreads();
    for (int i = tid; i < D; i += kFThreads) {
      skm[i] = p.masks[(size_t)sf * D + i];
      skm[D + i] = p.masks[(size_t)sb * D + i];
    }
    __syncthreads();
#pragma nounroll
    for (int call = 3; call >= 0; --call) {
      const bool is_v = call == 0 || call == 3;
      const int sub = call == 2 ? 1 : 0;
      const int cidx = 2 * step + (call >= 2 ? 1 : 0);
      const FusedTape& tp = is_v ? p.tv : p.tx;
      const float* pk = is_v ? p.pk_v : p.pk_x;
      const size_t tcr0 = (size_t)cidx * (size_t)p.rows + (size_t)row0;
      const float* ecs = ec + (is_v ? 2 * D : 0);
      const float* ecq = ecs + D;
      const int q_tanh = is_v ? p.qtanh_v : p.qtanh_x;
      float* dout_t = is_v ? p.dout_v : p.dout_x;

      // relu masks of this call's h1 / h2, written by the taped forward kernel in this lane's fragment order
      const unsigned gate1 = tp.gate[((size_t)(cidx * 2 + 0) * gridDim.x + blockIdx.x) * kFThreads + tid];
      const unsigned gate2 = tp.gate[((size_t)(cidx * 2 + 1) * gridDim.x + blockIdx.x) * kFThreads + tid];
      // ================= phase A: the sub-update, element-wise (gauge_dynamics.py:486-590 differentiated)
      {
        float S[8], Tt[8], Q[8], st[8], ia[8], ib[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            S[4 * h + k] = tnext.vS[h][k]; Tt[4 * h + k] = tnext.vT[h][k]; Q[4 * h + k] = tnext.vQ[h][k];
            st[4 * h + k] = tnext.vst[h][k]; ia[4 * h + k] = tnext.va[h][k]; ib[4 * h + k] = tnext.vb[h][k];
          }
        }
        {
          // request the next call's tape now (reverse order: calls 3, 2, 1, 0 of a step, then step - 1)
          const int ncall = call == 0 ? 3 : call - 1, nstep = call == 0 ? step - 1 : step;
          if (nstep >= 0) load_tape(2 * nstep + (ncall >= 2 ? 1 : 0), ncall == 0 || ncall == 3, tnext);
        }
        float o_s[8], o_t[8], o_q[8];
        // The thread's 8 columns of every LDS row it touches move as two 16-byte pieces (scalar accesses at a column
        // stride of 8 floats between lanes are 8-way bank conflicts: SQ_LDS_BANK_CONFLICT was 56 % of this kernel's
        // LDS-active cycles), the element-wise arithmetic runs on registers.
        float udx[8], udv[8], mfr[8], mbr[8], ecsr[8], ecqr[8], dgdr[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(dxs + fc * SX + c0 + 4 * h);
          const f32x4 a1 = *reinterpret_cast<const f32x4*>(dvs + fc * SX + c0 + 4 * h);
          const f32x4 a2 = *reinterpret_cast<const f32x4*>(skm + c0 + 4 * h);
          const f32x4 a3 = *reinterpret_cast<const f32x4*>(skm + D + c0 + 4 * h);
          const f32x4 a4 = *reinterpret_cast<const f32x4*>(ecs + c0 + 4 * h);
          const f32x4 a5 = *reinterpret_cast<const f32x4*>(ecq + c0 + 4 * h);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            udx[4 * h + k] = a0[k]; udv[4 * h + k] = a1[k]; mfr[4 * h + k] = a2[k]; mbr[4 * h + k] = a3[k];
            ecsr[4 * h + k] = a4[k]; ecqr[4 * h + k] = a5[k]; dgdr[4 * h + k] = 0.f;
          }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float eq = fast_exp(eps * Q[k]);
          float dS, dT, dQ;
          if (is_v) {
            // st = v before the kick, ib = force, ia = x
            const float vv = st[k], gg = ib[k], u = udv[k], he = 0.5f * eps;
            float dgd;
            if (!d) {
              const float es = fast_exp(he * S[k]);
              const float ds = u * vv * es + dl;
              udv[k] = u * es;
              dS = ds * he; dT = u * he; dQ = -u * he * eq * gg * eps;
              dgd = -u * he * eq;
              deps += ds * 0.5f * S[k] - u * 0.5f * (eq * gg - Tt[k]) - u * he * gg * eq * Q[k];
            } else {
              const float es = fast_exp(-he * S[k]);
              const float vp = es * (vv + he * (eq * gg - Tt[k]));
              const float dw = u * es;
              const float ds = u * vp + dl;
              udv[k] = dw;
              dS = -he * ds; dT = -dw * he; dQ = dw * he * eq * gg * eps;
              dgd = dw * he * eq;
              deps += -0.5f * S[k] * ds + dw * 0.5f * (eq * gg - Tt[k]) + dw * he * gg * eq * Q[k];
            }
            dgdr[k] = dgd;
          } else {
            // st = x before the update, ia = v; keep mask per direction and sub-update
            const float mf = mfr[k], mb = mbr[k];
            const float kk = sub == 0 ? (d ? 1.f - mb : mf) : (d ? mb : 1.f - mf), mi = 1.f - kk;
            const float xx = st[k], vv = ia[k], u = udx[k];
            const float dy = mi * u;
            if (!d) {
              const float es = fast_exp(eps * S[k]);
              const float ds = dy * xx * es + dl * mi;
              udx[k] = kk * u + dy * es;
              udv[k] += dy * eps * eq;
              dS = eps * ds; dT = dy * eps; dQ = dy * eps * eq * vv * eps;
              deps += ds * S[k] + dy * (eq * vv + Tt[k]) + dy * eps * vv * eq * Q[k];
            } else {
              const float es = fast_exp(-eps * S[k]);
              const float w = xx - eps * (eq * vv + Tt[k]);
              const float dw = dy * es;
              const float ds = dy * (es * w) + dl * mi;
              udx[k] = kk * u + dw;
              udv[k] -= dw * eps * eq;
              dS = -eps * ds; dT = -dw * eps; dQ = -dw * eps * eq * vv * eps;
              deps += -S[k] * ds - dw * (eq * vv + Tt[k]) - dw * eps * vv * eq * Q[k];
            }
          }
          // through tanh(.) * exp(coeff) (generic_net.py:139-144)
          const float es_ = ecsr[k], eq_ = ecqr[k];
          const float th = S[k] / es_;
          float daq = dQ * eq_;
          if (q_tanh) {
            const float tq = Q[k] / eq_;
            daq *= 1.f - tq * tq;
          }
          o_s[k] = dS * es_ * (1.f - th * th);
          o_t[k] = dT;
          o_q[k] = daq;
          if (is_v) { acs_v[k] += dS * S[k]; acq_v[k] += dQ * Q[k]; }
          else { acs_x[k] += dS * S[k]; acq_x[k] += dQ * Q[k]; }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          *reinterpret_cast<f32x4*>(dvs + fc * SX + c0 + 4 * h) =
              f32x4{udv[4 * h], udv[4 * h + 1], udv[4 * h + 2], udv[4 * h + 3]};
          if (is_v) {
            *reinterpret_cast<f32x4*>(us + fc * SX + c0 + 4 * h) =
                f32x4{dgdr[4 * h], dgdr[4 * h + 1], dgdr[4 * h + 2], dgdr[4 * h + 3]};
            *reinterpret_cast<f32x4*>(xs + fc * SX + c0 + 4 * h) =
                f32x4{ia[4 * h], ia[4 * h + 1], ia[4 * h + 2], ia[4 * h + 3]};
          } else {
            *reinterpret_cast<f32x4*>(dxs + fc * SX + c0 + 4 * h) =
                f32x4{udx[4 * h], udx[4 * h + 1], udx[4 * h + 2], udx[4 * h + 3]};
          }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 vs_ = {o_s[4 * h], o_s[4 * h + 1], o_s[4 * h + 2], o_s[4 * h + 3]};
          const f32x4 vt_ = {o_t[4 * h], o_t[4 * h + 1], o_t[4 * h + 2], o_t[4 * h + 3]};
          const f32x4 vq_ = {o_q[4 * h], o_q[4 * h + 1], o_q[4 * h + 2], o_q[4 * h + 3]};
          float* lo = dos + fc * SO + c0 + 4 * h;
          *reinterpret_cast<f32x4*>(lo) = vs_;
          *reinterpret_cast<f32x4*>(lo + D) = vt_;
          *reinterpret_cast<f32x4*>(lo + 2 * D) = vq_;
          if (live) {
            float* go = dout_t + (tcr0 + fc) * (3 * D) + c0 + 4 * h;
            tape_store(go, vs_);
            tape_store(go + D, vt_);
            tape_store(go + 2 * D, vq_);
          }
        }
      }
      __syncthreads();

      // ================= phase B: delta2 = (dout . Whd) gated by h2 > 0
      const int wv = __builtin_amdgcn_readfirstlane(wave);    // provably uniform: the weight loads' base stays in SGPRs
      const float* wpb1 = pk + (size_t)wv * Cfg::KCO * NT1 * 256;
      const float* wpb2 = pk + Cfg::P1 + (size_t)wv * Cfg::KCH * NT1 * 256;
      const float* wpb3 = pk + Cfg::P1 + Cfg::P2 + (size_t)wv * Cfg::KCH * NT3 * 256;
      BRing<NT1, 4> R2;
      {
        BRing<NT1, 4> R1;
        ring_prime<NT1, 4>(R1, wpb1);
        f32x4 acc[NT1];
#pragma unroll
        for (int t = 0; t < NT1; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* a = dos + r * SO + q * 4;
        stream_layer<NT1, Cfg::KCO, 4>(
            R1, wpb1, [&](int kc) { return *reinterpret_cast<const f32x4*>(a + kc * 16); }, acc);
        ring_prime<NT1, 4>(R2, wpb2);
#pragma unroll
        for (int t = 0; t < NT1; ++t) {       // lane (q, r): row r, columns c0 .. c0 + 3 (fused_common.h)
          const int c0 = (wave * NT1 + t) * 16 + q * 4;
          f32x4 g;
#pragma unroll
          for (int e = 0; e < 4; ++e) g[e] = ((gate2 >> (t * 4 + e)) & 1u) ? acc[t][e] : 0.f;
          *reinterpret_cast<f32x4*>(d2s + r * SH + c0) = g;
        }
      }
      __syncthreads();
      tape_rows(is_v ? p.d2_v : p.d2_x, tcr0, d2s, H, SH);

      // ================= phase C: delta1 = (delta2 . Wh) gated by h1 > 0
      BRing<NT3, 5> R3;
      {
        f32x4 acc[NT1];
#pragma unroll
        for (int t = 0; t < NT1; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* a = d2s + r * SH + q * 4;
        stream_layer<NT1, Cfg::KCH, 4>(
            R2, wpb2, [&](int kc) { return *reinterpret_cast<const f32x4*>(a + kc * 16); }, acc);
        ring_prime<NT3, 5>(R3, wpb3);
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
          const int c0 = (wave * NT1 + t) * 16 + q * 4;
          f32x4 g;
#pragma unroll
          for (int e = 0; e < 4; ++e) g[e] = ((gate1 >> (t * 4 + e)) & 1u) ? acc[t][e] : 0.f;
          *reinterpret_cast<f32x4*>(d1s + r * SH + c0) = g;
        }
      }
      __syncthreads();
      tape_rows(is_v ? p.d1_v : p.d1_x, tcr0, d1s, H, SH);

      // ================= phase D: din = delta1 . W1   -> dis ([16][2D], aliases dos)
      float* dis = dos;
      constexpr int SI = SO;
      {
        f32x4 acc[NT3];
#pragma unroll
        for (int t = 0; t < NT3; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* a = d1s + r * SH + q * 4;
        stream_layer<NT3, Cfg::KCH, 5>(
            R3, wpb3, [&](int kc) { return *reinterpret_cast<const f32x4*>(a + kc * 16); }, acc);
#pragma unroll
        for (int t = 0; t < NT3; ++t)
          *reinterpret_cast<f32x4*>(dis + r * SI + (wave * NT3 + t) * 16 + q * 4) = acc[t];
      }
      __syncthreads();

      // ================= phase E: into d loss / d (x, v)
      if (!is_v) {
        // inputs (v, keep (.) x): gauge_dynamics.py:515-517
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int c = c0 + k;
          const float mf = skm[c], mb = skm[D + c];
          const float kk = sub == 0 ? (d ? 1.f - mb : mf) : (d ? mb : 1.f - mf);
          dvs[fc * SX + c] += dis[fc * SI + c];
          dxs[fc * SX + c] += kk * dis[fc * SI + D + c];
        }
      } else {
        // inputs (x, force = beta * grad_action(x)): gauge_dynamics.py:493-495, :698-709
#pragma unroll
        for (int k = 0; k < 8; ++k) us[fc * SX + c0 + k] += dis[fc * SI + D + c0 + k];
        __syncthreads();
        const float* xc = xs + fc * SX;
        const float* uc = us + fc * SX;
        for (int s = fl; s < sites; s += kTPC) {
          const int i = s >> xsh, j = s & (X - 1);            // X is a power of two (T * X = 64)
          const int jp = (j + 1 == X) ? 0 : j + 1, ip = (i + 1 == T) ? 0 : i + 1;
          const int e0 = 2 * s, er = 2 * (i * X + jp), ed = 2 * (ip * X + j);
          const float P = xc[e0] - xc[e0 + 1] - xc[er] + xc[ed + 1];
          const float Pu = uc[e0] - uc[e0 + 1] - uc[er] + uc[ed + 1];
          float sn_, cs_;
          fast_sincos(P, &sn_, &cs_);       // (the polynomial pair of common.h, ~1e-7: as in the forward stencil)
          sp[fc * SP + s] = cs_ * Pu;
        }
        __syncthreads();
        const float* spc = sp + fc * SP;
        for (int s = fl; s < sites; s += kTPC) {
          const int i = s >> xsh, j = s & (X - 1);            // X is a power of two (T * X = 64)
          const int jm = (j == 0) ? X - 1 : j - 1, im = (i == 0) ? T - 1 : i - 1;
          const float c = spc[s];
          dxs[fc * SX + 2 * s] += dis[fc * SI + 2 * s] + p.beta * (c - spc[i * X + jm]);
          dxs[fc * SX + 2 * s + 1] += dis[fc * SI + 2 * s + 1] + p.beta * (-c + spc[im * X + j]);
        }
      }
      __syncthreads();
    }
  }

  // ---- write back d loss / d (x_0, v_0), coefficient and step-size partials -----------------------
  for (int i = tid; i < kFM * (D / 4); i += kFThreads) {
    const int rr = i / (D / 4), c4 = (i - rr * (D / 4)) * 4;
    if (rr < nrow) {
      *reinterpret_cast<f32x4*>(p.dx + (row0 + rr) * D + c4) = *reinterpret_cast<const f32x4*>(dxs + rr * SX + c4);
      *reinterpret_cast<f32x4*>(p.dv + (row0 + rr) * D + c4) = *reinterpret_cast<const f32x4*>(dvs + rr * SX + c4);
    }
  }
  auto col_reduce = [&](const float (&acc)[8], float* out) {     // sum over the 16 chains, fixed order
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) d2s[fc * SH + c0 + k] = live ? acc[k] : 0.f;
    __syncthreads();
    if (tid < D) {
      float s = 0.f;
#pragma unroll
      for (int rr = 0; rr < kFM; ++rr) s += d2s[rr * SH + tid];
      out[(size_t)blockIdx.x * D + tid] = s;
    }
  };
  col_reduce(acs_x, p.dcs_x);
  col_reduce(acq_x, p.dcq_x);
  col_reduce(acs_v, p.dcs_v);
  col_reduce(acq_v, p.dcq_v);
  deps = live ? deps : 0.f;
  deps = wave_sum(deps);
  __syncthreads();
  if (lane == 0) red[wave] = deps;
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
    for (int w = 0; w < kFWaves; ++w) s += red[w];
    p.deps[blockIdx.x] = s;
  }
}

// =====================================================================================================
// ONE network call of the reverse pass in one launch, for plans whose front-end keeps the reverse pass layered
// (ConvNet3D: its front-end's reverse pass wants many resident waves, conv3d_front.hip).  Replaces, per call,
// update_bwd_kernel + three gemm_relu launches of a few GFLOP each (one workgroup per CU: 12-16 us each,
// latency-bound) by the element-wise phase and the three streamed products of the whole-trajectory kernel above on
// 16 rows per workgroup: d loss / d (x, v) come from and go back to HBM, the relu gates are read from the taped
// h1 / h2 rows (the forward kernel's mask words follow ITS wave count), the first layer's input gradient
// (d loss / d features) goes out for conv3d_front_bwd_kernel, coefficient and step-size partials accumulate (+=)
// in the workgroup's slots.  Same arithmetic and summation order per element as the layered kernels.
// =====================================================================================================

template <int D, int H, int K1>
struct TrunkBwdCfg {
  static constexpr int SO = 3 * D + 8, SH = H + 8;
  static constexpr int NT1 = H / (16 * kFWaves), NT3 = K1 / (16 * kFWaves);
  static constexpr int KCO = 3 * D / 16, KCH = H / 16;
  static constexpr size_t P1 = (size_t)3 * D * H, P2 = (size_t)H * H;
  static constexpr int LDS_FLOATS = kFM * SO + 2 * kFM * SH + 2 * D + kFM + 8;
  static_assert(NT1 >= 1 && NT3 >= 1 && kTPC * 8 == D, "tile / thread mapping");
};

template <int D, int H, int K1>
__global__ __launch_bounds__(kFThreads) void gauge_trunk_bwd_kernel(TrunkBwdArgs p) {
  using Cfg = TrunkBwdCfg<D, H, K1>;
  constexpr int SO = Cfg::SO, SH = Cfg::SH, NT1 = Cfg::NT1, NT3 = Cfg::NT3;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* dos = lds;                       // [16][SO] head pre-activation gradients
  float* d2s = dos + kFM * SO;            // [16][SH]
  float* d1s = d2s + kFM * SH;            // [16][SH]
  float* ec = d1s + kFM * SH;             // exp(cs) exp(cq)  [2][D]
  int* sdir = reinterpret_cast<int*>(ec + 2 * D);      // [16]
  float* red = reinterpret_cast<float*>(sdir + kFM);   // [8]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, r = lane & 15;
  const int64_t row0 = (int64_t)blockIdx.x * kFM;
  const int nrow = (int)min((int64_t)kFM, p.rows - row0);
  const float eps = p.eps;
  const int fc = tid / kTPC, fl = tid % kTPC;      // chain, lane-in-chain
  const int c0 = fl * 8;                           // this thread's 8 columns in the element-wise phase
  const bool live = fc < nrow;
  const bool is_v = p.mode == 1;

  for (int i = tid; i < D; i += kFThreads) {
    ec[i] = expf(p.cs[i]);
    ec[D + i] = expf(p.cq[i]);
  }
  if (tid < kFM) sdir[tid] = (tid < nrow && p.dir) ? p.dir[row0 + tid] : 0;
  __syncthreads();
  const int d = sdir[fc];
  const int64_t grow = row0 + (live ? fc : 0);     // (dead rows read row 0 of the tile and write nothing)
  const float dl = live ? p.dld[grow] : 0.f;

  // ---- phase A: the sub-update, element-wise (gauge_dynamics.py:486-590 differentiated; update_bwd_kernel's math)
  float acs[8], acq[8], deps = 0.f;
  {
    float S[8], Tt[8], Q[8], st[8], ia[8], ib[8], u[8], uv[8];
    const float* sq = p.stq + grow * D + c0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x4 vS = *reinterpret_cast<const f32x4*>(sq + 4 * h);
      const f32x4 vT = *reinterpret_cast<const f32x4*>(sq + p.plane + 4 * h);
      const f32x4 vQ = *reinterpret_cast<const f32x4*>(sq + 2 * p.plane + 4 * h);
      const f32x4 vst = *reinterpret_cast<const f32x4*>(p.st + grow * D + c0 + 4 * h);
      const f32x4 va = *reinterpret_cast<const f32x4*>(p.in + grow * (2 * D) + c0 + 4 * h);
      const f32x4 vb = *reinterpret_cast<const f32x4*>(p.in + grow * (2 * D) + D + c0 + 4 * h);
      const f32x4 vx = *reinterpret_cast<const f32x4*>(p.dx + grow * D + c0 + 4 * h);
      const f32x4 vv = *reinterpret_cast<const f32x4*>(p.dv + grow * D + c0 + 4 * h);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        S[4 * h + k] = vS[k]; Tt[4 * h + k] = vT[k]; Q[4 * h + k] = vQ[k]; st[4 * h + k] = vst[k];
        ia[4 * h + k] = va[k]; ib[4 * h + k] = vb[k]; u[4 * h + k] = vx[k]; uv[4 * h + k] = vv[k];
      }
    }
    float o_s[8], o_t[8], o_q[8], ndx[8], ndv[8], ndg[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = c0 + k;
      const float eq = expf(eps * Q[k]);
      float dS, dT, dQ;
      ndx[k] = u[k]; ndv[k] = uv[k]; ndg[k] = 0.f;
      if (is_v) {
        // st = v before the kick, ib = force
        const float vv = st[k], gg = ib[k], uu = uv[k], he = 0.5f * eps;
        if (!d) {
          const float es = expf(he * S[k]);
          const float ds = uu * vv * es + dl;
          ndv[k] = uu * es;
          dS = ds * he; dT = uu * he; dQ = -uu * he * eq * gg * eps;
          ndg[k] = -uu * he * eq;
          deps += ds * 0.5f * S[k] - uu * 0.5f * (eq * gg - Tt[k]) - uu * he * gg * eq * Q[k];
        } else {
          const float es = expf(-he * S[k]);
          const float kick = he * (eq * gg - Tt[k]);
          const float vp = es * (vv + kick);
          const float dw = uu * es;
          const float ds = uu * vp + dl;
          ndv[k] = dw;
          dS = -he * ds; dT = -dw * he; dQ = dw * he * eq * gg * eps;
          ndg[k] = dw * he * eq;
          deps += -0.5f * S[k] * ds + dw * 0.5f * (eq * gg - Tt[k]) + dw * he * gg * eq * Q[k];
        }
      } else {
        // st = x before the update, ia = v
        const float kk = (d ? p.keep_b : p.keep_f)[c], mi = 1.f - kk;
        const float xx = st[k], vv = ia[k], uu = u[k];
        const float dy = mi * uu;
        if (!d) {
          const float es = expf(eps * S[k]);
          const float ds = dy * xx * es + dl * mi;
          ndx[k] = kk * uu + dy * es;
          ndv[k] = uv[k] + dy * eps * eq;
          dS = eps * ds; dT = dy * eps; dQ = dy * eps * eq * vv * eps;
          deps += ds * S[k] + dy * (eq * vv + Tt[k]) + dy * eps * vv * eq * Q[k];
        } else {
          const float es = expf(-eps * S[k]);
          const float w = xx - eps * (eq * vv + Tt[k]);
          const float dw = dy * es;
          const float ds = dy * (es * w) + dl * mi;
          ndx[k] = kk * uu + dw;
          ndv[k] = uv[k] - dw * eps * eq;
          dS = -eps * ds; dT = -dw * eps; dQ = -dw * eps * eq * vv * eps;
          deps += -S[k] * ds - dw * (eq * vv + Tt[k]) - dw * eps * vv * eq * Q[k];
        }
      }
      // through tanh(.) * exp(coeff) (generic_net.py:139-144)
      const float es_ = ec[c], eq_ = ec[D + c];
      const float th = S[k] / es_;
      float daq = dQ * eq_;
      if (p.q_tanh) {
        const float tq = Q[k] / eq_;
        daq *= 1.f - tq * tq;
      }
      o_s[k] = dS * es_ * (1.f - th * th);
      o_t[k] = dT;
      o_q[k] = daq;
      acs[k] = dS * S[k];
      acq[k] = dQ * Q[k];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x4 vs_ = {o_s[4 * h], o_s[4 * h + 1], o_s[4 * h + 2], o_s[4 * h + 3]};
      const f32x4 vt_ = {o_t[4 * h], o_t[4 * h + 1], o_t[4 * h + 2], o_t[4 * h + 3]};
      const f32x4 vq_ = {o_q[4 * h], o_q[4 * h + 1], o_q[4 * h + 2], o_q[4 * h + 3]};
      float* lo = dos + fc * SO + c0 + 4 * h;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(lo) = live ? vs_ : z;
      *reinterpret_cast<f32x4*>(lo + D) = live ? vt_ : z;
      *reinterpret_cast<f32x4*>(lo + 2 * D) = live ? vq_ : z;
      if (live) {
        float* go = p.dout + grow * (3 * D) + c0 + 4 * h;
        *reinterpret_cast<f32x4*>(go) = vs_;
        *reinterpret_cast<f32x4*>(go + D) = vt_;
        *reinterpret_cast<f32x4*>(go + 2 * D) = vq_;
        *reinterpret_cast<f32x4*>(p.dx + grow * D + c0 + 4 * h) = f32x4{ndx[4 * h], ndx[4 * h + 1], ndx[4 * h + 2], ndx[4 * h + 3]};
        *reinterpret_cast<f32x4*>(p.dv + grow * D + c0 + 4 * h) = f32x4{ndv[4 * h], ndv[4 * h + 1], ndv[4 * h + 2], ndv[4 * h + 3]};
        if (is_v)
          *reinterpret_cast<f32x4*>(p.dg + grow * D + c0 + 4 * h) = f32x4{ndg[4 * h], ndg[4 * h + 1], ndg[4 * h + 2], ndg[4 * h + 3]};
      }
    }
  }
  __syncthreads();

  const float* pk = p.pk;
  const int wv = __builtin_amdgcn_readfirstlane(wave);        // provably uniform: the weight loads' base stays in SGPRs
  const float* wpb1 = pk + (size_t)wv * Cfg::KCO * NT1 * 256;
  const float* wpb2 = pk + Cfg::P1 + (size_t)wv * Cfg::KCH * NT1 * 256;
  const float* wpb3 = pk + Cfg::P1 + Cfg::P2 + (size_t)wv * Cfg::KCH * NT3 * 256;
  const bool rlive = r < nrow;
  const int64_t rrow = row0 + (rlive ? r : 0);    // lane (q, r): row r, four consecutive columns (fused_common.h)
  // ---- phase B: delta2 = (dout . Whd) gated by h2 > 0
  BRing<NT1, 4> R2;
  {
    BRing<NT1, 4> R1;
    ring_prime<NT1, 4>(R1, wpb1);
    f32x4 acc[NT1];
#pragma unroll
    for (int t = 0; t < NT1; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* a = dos + r * SO + q * 4;
    stream_layer<NT1, Cfg::KCO, 4>(
        R1, wpb1, [&](int kc) { return *reinterpret_cast<const f32x4*>(a + kc * 16); }, acc);
    ring_prime<NT1, 4>(R2, wpb2);
#pragma unroll
    for (int t = 0; t < NT1; ++t) {
      const int cc = (wave * NT1 + t) * 16 + q * 4;
      const f32x4 hv = *reinterpret_cast<const f32x4*>(p.h2 + rrow * H + cc);
      f32x4 g;
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = (rlive && hv[e] > 0.f) ? acc[t][e] : 0.f;
      *reinterpret_cast<f32x4*>(d2s + r * SH + cc) = g;
      if (rlive) *reinterpret_cast<f32x4*>(p.d2 + rrow * H + cc) = g;
    }
  }
  __syncthreads();
  // ---- phase C: delta1 = (delta2 . Wh) gated by h1 > 0
  BRing<NT3, 4> R3;
  {
    f32x4 acc[NT1];
#pragma unroll
    for (int t = 0; t < NT1; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* a = d2s + r * SH + q * 4;
    stream_layer<NT1, Cfg::KCH, 4>(
        R2, wpb2, [&](int kc) { return *reinterpret_cast<const f32x4*>(a + kc * 16); }, acc);
    ring_prime<NT3, 4>(R3, wpb3);
#pragma unroll
    for (int t = 0; t < NT1; ++t) {
      const int cc = (wave * NT1 + t) * 16 + q * 4;
      const f32x4 hv = *reinterpret_cast<const f32x4*>(p.h1 + rrow * H + cc);
      f32x4 g;
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = (rlive && hv[e] > 0.f) ? acc[t][e] : 0.f;
      *reinterpret_cast<f32x4*>(d1s + r * SH + cc) = g;
      if (rlive) *reinterpret_cast<f32x4*>(p.d1 + rrow * H + cc) = g;
    }
  }
  __syncthreads();
  // ---- phase D: d loss / d (first-layer inputs) = delta1 . W1, straight to HBM
  {
    f32x4 acc[NT3];
#pragma unroll
    for (int t = 0; t < NT3; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* a = d1s + r * SH + q * 4;
    stream_layer<NT3, Cfg::KCH, 4>(
        R3, wpb3, [&](int kc) { return *reinterpret_cast<const f32x4*>(a + kc * 16); }, acc);
    if (rlive) {
#pragma unroll
      for (int t = 0; t < NT3; ++t)
        *reinterpret_cast<f32x4*>(p.dfeat + rrow * K1 + (wave * NT3 + t) * 16 + q * 4) = acc[t];
    }
  }
  // ---- coefficient / step-size partials: sum over the tile's chains in order, += into the workgroup's slots
  auto col_reduce = [&](const float (&a8)[8], float* out) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) d2s[fc * SH + c0 + k] = live ? a8[k] : 0.f;
    __syncthreads();
    if (tid < D) {
      float sum = 0.f;
#pragma unroll
      for (int rr = 0; rr < kFM; ++rr) sum += d2s[rr * SH + tid];
      out[(size_t)blockIdx.x * D + tid] += sum;
    }
  };
  col_reduce(acs, p.dcs_part);
  col_reduce(acq, p.dcq_part);
  deps = live ? deps : 0.f;
  deps = wave_sum(deps);
  __syncthreads();
  if (lane == 0) red[wave] = deps;
  __syncthreads();
  if (tid == 0) {
    float sum = 0.f;
    for (int w = 0; w < kFWaves; ++w) sum += red[w];
    p.deps_part[blockIdx.x] += sum;
  }
}

// plans whose reverse pass stays layered but whose dense trunk has a single-call kernel: the 8x8 ConvNet3D trunk
int trunk_bwd_supported(const l2hmc_dense_net* n) {
  return n->D == 128 && n->H == 256 && n->Ka + n->Kb == 128;
}

int launch_trunk_bwd_pack(const l2hmc_dense_net* n, float* pack, hipStream_t stream) {
  hipLaunchKernelGGL(pack_fused_bwd_kernel, dim3(256), dim3(256), 0, stream, *n, pack);
  L2HMC_CHECK_LAUNCH("pack_fused_bwd");
  return L2HMC_OK;
}

int launch_trunk_bwd(TrunkBwdArgs& a, hipStream_t stream) {
  using Cfg = TrunkBwdCfg<128, 256, 128>;
  static DeviceOnce attr_once;
  const size_t lds = sizeof(float) * Cfg::LDS_FLOATS;
  if (attr_once.pending()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gauge_trunk_bwd_kernel<128, 256, 128>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      set_error("trunk backward: cannot reserve %zu B of LDS", lds);
      return L2HMC_ERR_HIP;
    }
    attr_once.done();
  }
  hipLaunchKernelGGL((gauge_trunk_bwd_kernel<128, 256, 128>), dim3((unsigned)ceil_div(a.rows, kFM)), dim3(kFThreads),
                     lds, stream, a);
  L2HMC_CHECK_LAUNCH("gauge_trunk_bwd");
  return L2HMC_OK;
}

size_t fused_bwd_pack_floats(const l2hmc_dense_net* n) {
  return (size_t)3 * n->D * n->H + (size_t)n->H * n->H + (size_t)n->H * (n->Ka + n->Kb);
}

// taped whole-trajectory forward: every plan with a fused kernel.  Fused reverse pass: GenericNet plans only --
// a ConvNet3D front-end's reverse pass is branchy scalar work that needs many resident waves to hide its LDS
// latency, and this kernel runs one wave per SIMD (measured in-kernel: 12.9 ms against 4.4 ms for the
// standalone conv3d_front_bwd_kernel over the same 40 calls), so conv plans keep the layered reverse pass.
int fused_train_forward_supported(const l2hmc_gauge_plan* p) {
  return !(p->flags & L2HMC_PLAN_LAYERED) && fused_plan_supported(p);
}
int fused_train_supported(const l2hmc_gauge_plan* p) {
  return !(p->flags & L2HMC_PLAN_CONV3D) && fused_train_forward_supported(p);
}

int launch_fused_train_backward(const l2hmc_gauge_plan* p, float beta, const int* dir, int64_t rows, float* dx,
                                float* dv, const float* dld, const FusedTape& tx, const FusedTape& tv,
                                float* const deltas_x[3], float* const deltas_v[3], float* pack_x, float* pack_v,
                                float* const coef_parts[4], float* deps_part, hipStream_t stream) {
  using Cfg = FusedBwdCfg<128, 512>;
  L2HMC_REQUIRE(fused_train_supported(p), "fused training backward: unsupported plan");
  static DeviceOnce attr_once;
  const size_t lds = sizeof(float) * Cfg::LDS_FLOATS;
  if (attr_once.pending()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gauge_train_bwd_fused_kernel<128, 512>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      set_error("fused training backward: cannot reserve %zu B of LDS", lds);
      return L2HMC_ERR_HIP;
    }
    attr_once.done();
  }
  hipLaunchKernelGGL(pack_fused_bwd_kernel, dim3(1024), dim3(256), 0, stream, p->xnet, pack_x);
  hipLaunchKernelGGL(pack_fused_bwd_kernel, dim3(1024), dim3(256), 0, stream, p->vnet, pack_v);
  L2HMC_CHECK_LAUNCH("pack_fused_bwd");
  FusedBwdArgs a{};
  a.T = p->T; a.X = p->X; a.num_steps = p->num_steps; a.eps = p->eps; a.beta = beta; a.masks = p->masks;
  a.pk_x = pack_x; a.pk_v = pack_v;
  a.cs_x = p->xnet.coeff_s; a.cq_x = p->xnet.coeff_q; a.cs_v = p->vnet.coeff_s; a.cq_v = p->vnet.coeff_q;
  a.qtanh_x = p->xnet.q_tanh; a.qtanh_v = p->vnet.q_tanh;
  a.dir = dir; a.rows = rows; a.tx = tx; a.tv = tv;
  a.dout_x = deltas_x[0]; a.d2_x = deltas_x[1]; a.d1_x = deltas_x[2];
  a.dout_v = deltas_v[0]; a.d2_v = deltas_v[1]; a.d1_v = deltas_v[2];
  a.dx = dx; a.dv = dv; a.dld = dld;
  a.dcs_x = coef_parts[0]; a.dcq_x = coef_parts[1]; a.dcs_v = coef_parts[2]; a.dcq_v = coef_parts[3];
  a.deps = deps_part;
  hipLaunchKernelGGL((gauge_train_bwd_fused_kernel<128, 512>), dim3((unsigned)ceil_div(rows, kFM)), dim3(kFThreads),
                     lds, stream, a);
  L2HMC_CHECK_LAUNCH("gauge_train_bwd_fused");
  return L2HMC_OK;
}

}  // namespace l2hmc
