// Sub-tile form of the whole-trajectory kernel (fused_traj.hip) for batches that cannot put a 16-row tile on every CU.
//
// The 16-row kernel's cost per workgroup does not depend on how many of its 16 rows are live: the weight stream and the
// v_mfma_f32_16x16x4_f32 count are per tile, so 256 chains take as long as 2048 (profiles/r03_batch_sweep.txt).  The
// only f32 matrix instruction with fewer rows is v_mfma_f32_4x4x1_16B_f32: 16 blocks of a 4 x 4 x 1 product, the first
// operand broadcast from one block to all (CBSZ = 4, ABID selects the block).  Here the FIRST operand is the
// activations -- lane l of one register holds act[row l % 4][k0 + l / 4], i.e. 4 rows x 16 k, and ABID picks the k --
// and the second the weights, lane l = output column: one instruction is 4 rows x 64 columns x 1 k in 8 cycles, a
// quarter of the matrix cycles of a 16-row tile for a 4-row group.  What remains is the weight stream (2.36 MB per
// network call and workgroup whatever the rows), so small tiles are L2-bandwidth-bound, not matrix-bound.
//
// Same algorithm, same arithmetic ORDER as fused_traj.hip (l2hmc/dynamics/gauge_dynamics.py:261-313, :412-609;
// network/generic_net.py:129-146): a product's k runs in the order the 16-row form's instructions take it (within a
// 16-k chunk: k = 4 q + e for e = 0..3, q = 0..3), epilogue expressions are the same, the log-det partial sums are added
// in the same grouping, the Philox streams are indexed by chain -- the results are bit-identical to the 16-row form
// (tests/test_gpu_parity.py::test_subtile_and_32_row_forms_equal_16_row_form).
//
// Geometry: 4 waves; wave w owns output columns [128 w, 128 w + 128) of layers 1 / 2 (two 64-column blocks) and columns
// [32 w, 32 w + 32) of each head (block 0: lanes 0-31 S, lanes 32-63 T; block 1: lanes 0-31 Q); a lane ends up with ONE
// column and 4 rows per row group (register i = row).  ROWS = 4, 8 or 12 rows per workgroup (1 to 3 row groups sharing
// every weight fragment).  GenericNet on the 8x8 lattice (D = 128, H = 512), sampling only (no tape, no ConvNet3D).
#include "fused_common.h"
#include "fused_args.h"

namespace l2hmc {

namespace {
constexpr int kD = 128, kH = 512;
constexpr int kWaves4 = 4, kThreads4 = 256;
constexpr int kSX4 = kD + 8, kSH4 = kH + 8;
constexpr size_t kP1 = (size_t)2 * kD * kH, kP2 = (size_t)kH * kH, kP4H = (size_t)4 * 32 * 2 * 4 * 256;   // padded heads
constexpr int kKC1 = 2 * kD / 16, kKC2 = kH / 16;           // 16-k chunks: layer 1 (both inputs), layers 2 / heads
constexpr int kNC = 4 * kH + 5 * kD;                        // per-net constants in LDS (as fused_traj.hip)

// [wave][chunk][block][m][lane][4]: W[column(wave, block, lane)][16 chunk + 4 m + j]
__global__ void pack_fused4_kernel(l2hmc_dense_net n, float* __restrict__ out) {
  const size_t total = kP1 + kP2 + kP4H;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int layer = i < kP1 ? 0 : (i < kP1 + kP2 ? 1 : 2);
    const size_t o = i - (layer == 0 ? 0 : layer == 1 ? kP1 : kP1 + kP2);
    const int j = (int)(o & 3), lane = (int)((o >> 2) & 63), m = (int)((o >> 8) & 3), cb = (int)((o >> 10) & 1);
    const size_t rest = o >> 11;
    const int KC = layer == 0 ? kKC1 : kKC2;
    const int kc = (int)(rest % KC), w = (int)(rest / KC);
    const int k = kc * 16 + 4 * m + j;
    float val = 0.f;
    if (layer == 0) {
      val = n.w1_t[(size_t)(w * 128 + cb * 64 + lane) * (2 * kD) + k];
    } else if (layer == 1) {
      val = n.wh_t[(size_t)(w * 128 + cb * 64 + lane) * kH + k];
    } else {
      const int hd = cb == 0 ? (lane >> 5) : (lane < 32 ? 2 : -1);
      if (hd >= 0) val = n.whd_t[((size_t)hd * kD + w * 32 + (lane & 31)) * kH + k];
    }
    out[i] = val;
  }
}

template <int ABID>
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, ABID, 0);    // A broadcast from block ABID to all 16
}

// one 16-k chunk: RG row groups x 2 column blocks, k in the order the 16-row form's instructions take it
template <int RG>
__device__ __forceinline__ void chunk4(const float (&a)[RG], const f32x4 (&wf)[2][4], f32x4 (&acc)[RG][2]) {
#define L2HMC_K4(E, Q)                                                                        \
  _Pragma("unroll") for (int cb = 0; cb < 2; ++cb) _Pragma("unroll") for (int g = 0; g < RG; ++g) \
      acc[g][cb] = mfma4<4 * (Q) + (E)>(a[g], wf[cb][Q][E], acc[g][cb]);
#define L2HMC_E4(E) L2HMC_K4(E, 0) L2HMC_K4(E, 1) L2HMC_K4(E, 2) L2HMC_K4(E, 3)
  L2HMC_E4(0) L2HMC_E4(1) L2HMC_E4(2) L2HMC_E4(3)
#undef L2HMC_E4
#undef L2HMC_K4
}

// acc += rows(arow, stride) . W^T over NCH chunks of this wave's section; wbase = section base (wave-uniform: the
// fragments are buffer loads, fused_common.h: lane_frag)
// rev: the chunks are walked from the last to the first -- the 16-row form streams layers 2 and 3 of a network in
// alternating directions on its consecutive calls (fused_common.h), and the order of k is part of the result's bits
template <int RG, int NCH>
__device__ __forceinline__ void stream4(const float* __restrict__ wbase, const float* arow, int stride, int lane,
                                        f32x4 (&acc)[RG][2], bool rev = false) {
  constexpr int DEPTH = 3;
  f32x4 ring[DEPTH][2][4];
  const WSection ws = wsection(wbase);
  auto load = [&](f32x4 (&dst)[2][4], int kw) {
    const int kc = rev ? NCH - 1 - kw : kw;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int m = 0; m < 4; ++m) dst[cb][m] = lane_frag(ws, (unsigned)((kc * 2 + cb) * 4 + m));
  };
  const float* ap = arow + (lane & 3) * stride + (lane >> 2);
  auto afrag = [&](float (&a)[RG], int kw) {
    const int kc = rev ? NCH - 1 - kw : kw;
#pragma unroll
    for (int g = 0; g < RG; ++g) a[g] = ap[4 * g * stride + kc * 16];
  };
#pragma unroll
  for (int s = 0; s < DEPTH; ++s) load(ring[s], s < NCH ? s : NCH - 1);
  float a0[RG];
  afrag(a0, 0);
  static_assert(NCH >= DEPTH, "ring depth");
  int kc = 0;
#pragma nounroll
  for (; kc + DEPTH <= NCH; kc += DEPTH) {
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) {
      float a1[RG];
      afrag(a1, kc + s + 1 < NCH ? kc + s + 1 : NCH - 1);
      chunk4<RG>(a0, ring[s], acc);
      if (kc + s + DEPTH < NCH) load(ring[s], kc + s + DEPTH);
#pragma unroll
      for (int g = 0; g < RG; ++g) a0[g] = a1[g];
    }
  }
  constexpr int REM = NCH % DEPTH;
#pragma unroll
  for (int s = 0; s < REM; ++s) {
    float a1[RG];
    afrag(a1, NCH - REM + s + 1 < NCH ? NCH - REM + s + 1 : NCH - 1);
    chunk4<RG>(a0, ring[s], acc);
#pragma unroll
    for (int g = 0; g < RG; ++g) a0[g] = a1[g];
  }
}

template <int ROWS>
struct F4Cfg {
  static constexpr int RG = ROWS / 4;
  static constexpr int TPC = 16;                             // threads per chain in the chain-local passes, as the 16-row form
  static constexpr int SP = kD / 2 + 4;
  // xs vs gs | h1 h2 | consts x2 | sinP | masks | ldw | dir | step scratch | log-det staging [waves][ROWS][32]
  static constexpr int LDS_FLOATS = 3 * ROWS * kSX4 + 2 * ROWS * kSH4 + 2 * kNC + ROWS * SP + 2 * kD + kWaves4 * ROWS +
                                    ROWS + 8 * ROWS + kWaves4 * ROWS * 32 +
                                    (2 * kThreads4 > ROWS * kSX4 ? 2 * kThreads4 - ROWS * kSX4 : 0);   // final-sum scratch fits vs
};

template <int ROWS>
__global__ __launch_bounds__(kThreads4) void gauge_traj_fused4_kernel(FusedArgs p) {
  using Cfg = F4Cfg<ROWS>;
  constexpr int RG = Cfg::RG, kTPC = Cfg::TPC, SX = kSX4, SH = kSH4, SP = Cfg::SP, D = kD, H = kH;
  constexpr int sites = D / 2;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xs = lds;                         // [ROWS][SX] position
  float* vs = xs + ROWS * SX;              // [ROWS][SX] momentum  (+ tail: scratch of the final step sums)
  float* gs = vs + (2 * kThreads4 > ROWS * SX ? 2 * kThreads4 : ROWS * SX);   // [ROWS][SX] force, or keep (.) x
  float* h1 = gs + ROWS * SX;              // [ROWS][SH]
  float* h2 = h1 + ROWS * SH;              // [ROWS][SH]
  float* cx = h2 + ROWS * SH;              // XNet constants [NC]
  float* cv = cx + kNC;
  float* sp = cv + kNC;                    // [ROWS][SP] sin P
  float* skm = sp + ROWS * SP;             // [2][D] masks of this step: forward row, backward row
  float* ldw = skm + 2 * D;                // [waves][ROWS] log-det partial sums per wave
  int* sdir = reinterpret_cast<int*>(ldw + kWaves4 * ROWS);   // [ROWS]
  float* stp = reinterpret_cast<float*>(sdir + ROWS);        // step mode: coin[R] u[R] p_row[R] obs[R][4]
  float* sst = stp + 8 * ROWS;             // [waves][ROWS][32] per-element log-det terms of a call (ordered sum)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  const int nrow = (int)min((int64_t)ROWS, p.rows - row0);
  const float eps = p.eps;

  // ---- stage chain state and constants (as fused_traj.hip, kFM -> ROWS) ----
  const bool STEPM = p.step_B > 0;
  const int cpw = STEPM ? (p.step_both ? ROWS / 2 : ROWS) : ROWS;
  float* scoin = stp;
  float* su = stp + ROWS;
  float* spx = stp + 2 * ROWS;
  float* sobs = stp + 3 * ROWS;            // [ROWS][4]
  auto philox_u01 = [&](uint64_t elem, uint64_t stream) {
    const uint64_t b = elem >> 2;
    uint32_t c[4] = {(uint32_t)b, (uint32_t)(b >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
    philox4x32_10(c, (uint32_t)p.step_seed, (uint32_t)(p.step_seed >> 32));
    return (float)(c[elem & 3] >> 8) * (1.0f / 16777216.0f);
  };
  if (STEPM) {
    if (tid < cpw) {
      const int64_t chain = (int64_t)blockIdx.x * cpw + tid;
      const bool lv = chain < p.step_Bl;          // (streams are indexed by the chain's place in the WHOLE batch)
      scoin[tid] = lv ? philox_u01((uint64_t)(p.step_chain0 + chain), 2 * p.step_draw + 1) : 1.f;
      su[tid] = lv ? philox_u01((uint64_t)(p.step_B + p.step_chain0 + chain), 2 * p.step_draw + 1) : 1.f;
    }
    __syncthreads();
    for (int i = tid; i < ROWS * (D / 4); i += kThreads4) {
      const int rr = i / (D / 4), c4 = (i - rr * (D / 4)) * 4;
      const int k = p.step_both ? (rr >= ROWS / 2 ? rr - ROWS / 2 : rr) : rr;
      const int64_t chain = (int64_t)blockIdx.x * cpw + k;
      const int dsel = p.step_both ? (rr >= ROWS / 2 ? 1 : 0) : (scoin[k] > 0.5f ? 0 : 1);
      f32x4 xv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
      if (chain < p.step_Bl) {
        xv = *reinterpret_cast<const f32x4*>(p.x0 + chain * D + c4);
        const uint64_t nb = (((uint64_t)dsel * (uint64_t)p.step_B + (uint64_t)(p.step_chain0 + chain)) * D + c4) >> 2;
        uint32_t c[4] = {(uint32_t)nb, (uint32_t)(nb >> 32), (uint32_t)(2 * p.step_draw), (uint32_t)((2 * p.step_draw) >> 32)};
        philox4x32_10(c, (uint32_t)p.step_seed, (uint32_t)(p.step_seed >> 32));
        float nv[4];
        philox_normal4(c, nv);
        vv = f32x4{nv[0], nv[1], nv[2], nv[3]};
      }
      *reinterpret_cast<f32x4*>(xs + rr * SX + c4) = xv;
      *reinterpret_cast<f32x4*>(vs + rr * SX + c4) = vv;
    }
  } else {
    for (int i = tid; i < ROWS * (D / 4); i += kThreads4) {
      const int rr = i / (D / 4), c4 = (i - rr * (D / 4)) * 4;
      f32x4 xv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
      if (rr < nrow) {
        const int64_t xr = p.x_mod > 0 ? (row0 + rr) % p.x_mod : row0 + rr;
        xv = *reinterpret_cast<const f32x4*>(p.x0 + xr * D + c4);
        vv = *reinterpret_cast<const f32x4*>(p.v0 + (row0 + rr) * D + c4);
      }
      *reinterpret_cast<f32x4*>(xs + rr * SX + c4) = xv;
      *reinterpret_cast<f32x4*>(vs + rr * SX + c4) = vv;
    }
  }
  auto load_consts = [&](const l2hmc_dense_net& n, float* c) {
    for (int i = tid; i < H; i += kThreads4) {
      c[i] = n.b1[i];
      c[H + i] = n.wt[i];
      c[2 * H + i] = n.wt[H + i];
      c[3 * H + i] = n.bh[i];
    }
    for (int i = tid; i < 3 * D; i += kThreads4) c[4 * H + i] = n.bhd[i];
    for (int i = tid; i < D; i += kThreads4) {
      c[4 * H + 3 * D + i] = expf(n.coeff_s[i]);
      c[4 * H + 4 * D + i] = expf(n.coeff_q[i]);
    }
  };
  load_consts(p.xnet, cx);
  load_consts(p.vnet, cv);
  if (tid < ROWS) {
    int d = 0;
    if (STEPM) d = p.step_both ? (tid >= ROWS / 2 ? 1 : 0) : (scoin[tid] > 0.5f ? 0 : 1);
    else if (tid < nrow) d = p.dir ? p.dir[row0 + tid] : (p.dir_split > 0 && row0 + tid >= p.dir_split) ? 1 : 0;
    sdir[tid] = d;
  }
  if (tid < kWaves4 * ROWS) ldw[tid] = 0.f;
  __syncthreads();

  // direction of the rows this lane's registers belong to: register i of row group g = row 4 g + i
  int dirr[RG][4];
#pragma unroll
  for (int g = 0; g < RG; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) dirr[g][i] = sdir[4 * g + i];

  // ---- chain-local passes: kTPC consecutive threads per chain ----
  // Chain-local passes (force, action, kinetic energy, plaquette sums): 16 threads per chain with the 16-row form's
  // grouping (terms strided by 16, butterfly over 16 lanes) -- the same bits as there; threads beyond 16 ROWS idle.
  constexpr int kSumT = 16;
  const bool cact = tid < kTPC * ROWS;                 // (idle threads: lane-in-chain beyond every loop bound)
  const int fc = cact ? tid / kTPC : 0, fl = cact ? tid % kTPC : (1 << 20);
  auto chain_sum = [&](float v) {
#pragma unroll
    for (int off = kSumT / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
  };
  const int T = p.T, X = p.X;
  const int xsh = 31 - __clz(X);
  auto force_pass = [&]() -> float {
    const float* xc = xs + fc * SX;
    float act = 0.f;
    for (int s = fl; s < sites && fl < kSumT; s += kSumT) {
      const int i = s >> xsh, j = s & (X - 1);
      const int jp = (j + 1 == X) ? 0 : j + 1, ip = (i + 1 == T) ? 0 : i + 1;
      const float P = xc[2 * s] - xc[2 * s + 1] - xc[2 * (i * X + jp)] + xc[2 * (ip * X + j) + 1];
      float sn, cs;
      fast_sincos(P, &sn, &cs);
      sp[fc * SP + s] = sn;
      act += 1.f - cs;
    }
    act = chain_sum(act);
    __syncthreads();
    float* gc = gs + fc * SX;
    const float* spc = sp + fc * SP;
    for (int s = fl; s < sites; s += kTPC) {
      const int i = s >> xsh, j = s & (X - 1);
      const int jm = (j == 0) ? X - 1 : j - 1, im = (i == 0) ? T - 1 : i - 1;
      const float sP = spc[s];
      gc[2 * s] = p.beta * (sP - spc[i * X + jm]);
      gc[2 * s + 1] = p.beta * (-sP + spc[im * X + j]);
    }
    __syncthreads();
    return act;
  };
  auto kinetic_pass = [&]() -> float {
    const float* vc = vs + fc * SX;
    float k = 0.f;
    for (int d = fl; d < D && fl < kSumT; d += kSumT) k += vc[d] * vc[d];
    return 0.5f * chain_sum(k);
  };
  const float act0 = force_pass();
  const float kin0 = kinetic_pass();

  f32x4 keep_v[RG][2], keep_x[RG][2];
  bool keep_v_valid = false;

  // l1: 0 compute both halves; 1 as 0, keep the product in keep_v; 2 take keep_v; 3 snapshot the first half in keep_x;
  //     4 start from keep_x, second half only   (fused_traj.hip)
  auto net_update = [&](const l2hmc_dense_net& net, const float* cn, const float* in1, int mode, int sub,
                        bool prep_next_mask, int l1, const float (&tcr)[RG][4], const float (&tsr)[RG][4], int callidx) {
    const bool zig = (callidx & 1) != 0;                   // as fused_traj.hip: layers 2 / 3 alternate their direction
    const float* pk = net.packed + (kP1 + kP2 + (size_t)3 * kD * kH);      // the sub-tile image follows the 16-row one
    const int wv = __builtin_amdgcn_readfirstlane(wave);      // provably uniform: the weight loads' base stays in SGPRs
    const float* wp1 = pk + (size_t)wv * kKC1 * 2 * 4 * 256;
    const float* wp2 = pk + kP1 + (size_t)wv * kKC2 * 2 * 4 * 256;
    const float* wph = pk + kP1 + kP2 + (size_t)wv * kKC2 * 2 * 4 * 256;
    // ----- layer 1
    {
      f32x4 acc[RG][2];
      if (l1 == 2) {
#pragma unroll
        for (int g = 0; g < RG; ++g)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) acc[g][cb] = keep_v[g][cb];
      } else {
        constexpr int KH = kKC1 / 2;
        if (l1 == 4) {
#pragma unroll
          for (int g = 0; g < RG; ++g)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) acc[g][cb] = keep_x[g][cb];
        } else {
#pragma unroll
          for (int g = 0; g < RG; ++g)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) acc[g][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
          stream4<RG, KH>(wp1, in1, SX, lane, acc);
          if (l1 == 3) {
#pragma unroll
            for (int g = 0; g < RG; ++g)
#pragma unroll
              for (int cb = 0; cb < 2; ++cb) keep_x[g][cb] = acc[g][cb];
          }
        }
        stream4<RG, KH>(wp1 + (size_t)KH * 2 * 4 * 256, gs, SX, lane, acc);
        if (l1 == 1) {
#pragma unroll
          for (int g = 0; g < RG; ++g)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) keep_v[g][cb] = acc[g][cb];
        }
      }
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int c = wave * 128 + cb * 64 + lane;
        const float b = cn[c], w0 = cn[H + c], w1 = cn[2 * H + c];
#pragma unroll
        for (int g = 0; g < RG; ++g)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            h1[(4 * g + i) * SH + c] = fmaxf(acc[g][cb][i] + b + (tcr[g][i] * w0 + tsr[g][i] * w1), 0.f);
      }
    }
    __syncthreads();
    // ----- layer 2
    {
      f32x4 acc[RG][2];
#pragma unroll
      for (int g = 0; g < RG; ++g)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[g][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
      stream4<RG, kKC2>(wp2, h1, SH, lane, acc, zig);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int c = wave * 128 + cb * 64 + lane;
        const float b = cn[3 * H + c];
#pragma unroll
        for (int g = 0; g < RG; ++g)
#pragma unroll
          for (int i = 0; i < 4; ++i) h2[(4 * g + i) * SH + c] = fmaxf(acc[g][cb][i] + b, 0.f);
      }
    }
    __syncthreads();
    // ----- heads + update: block 0 = S (lanes 0-31) | T (lanes 32-63), block 1 = Q (lanes 0-31)
    {
      f32x4 acc[RG][2];
#pragma unroll
      for (int g = 0; g < RG; ++g)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[g][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
      stream4<RG, kKC2>(wph, h2, SH, lane, acc, zig);
      const float* bhd = cn + 4 * H;
      const float* es = bhd + 3 * D;
      const float* eq = es + D;
      const int cl = lane & 31;
      const int c = wave * 32 + cl;                       // this lane's column (lanes 32-63 mirror 0-31 and stay idle)
      const float b_s = bhd[c], b_t = bhd[D + c], b_q = bhd[2 * D + c], e_s = es[c], e_q = eq[c];
      const float mf = skm[c], mb = skm[D + c];
      float* myst = sst + wave * ROWS * 32;
#pragma unroll
      for (int g = 0; g < RG; ++g) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = 4 * g + i;
          const float aT = __shfl(acc[g][0][i], cl + 32, 64);      // the T product of this column sits 32 lanes up
          const float S = fast_tanh(acc[g][0][i] + b_s) * e_s;
          const float Tt = aT + b_t;
          const float qq = acc[g][1][i] + b_q;
          const float Q = (net.q_tanh ? fast_tanh(qq) : qq) * e_q;
          const int d = dirr[g][i];
          const int idx = row * SX + c;
          float term;
          if (mode == 1) {
            const float gg = gs[idx], v = vs[idx];
            const float s = (d ? -0.5f : 0.5f) * eps * S;
            const float kick = 0.5f * eps * (fast_exp(eps * Q) * gg - Tt);
            const float es_ = fast_exp(s);
            const float vn = d ? es_ * (v + kick) : v * es_ - kick;
            term = s;
            if (lane < 32) {
              vs[idx] = vn;
              if (prep_next_mask) gs[idx] = (d ? 1.f - mb : mf) * xs[idx];
            }
          } else {
            const float x = xs[idx], v = vs[idx];
            const float keep = sub == 0 ? (d ? 1.f - mb : mf) : (d ? mb : 1.f - mf);
            const float s = (d ? -eps : eps) * S;
            const float drift = eps * (fast_exp(eps * Q) * v + Tt);
            const float es_ = fast_exp(s);
            const float upd = d ? es_ * (x - drift) : x * es_ + drift;
            const float xn = keep * x + (1.f - keep) * upd;
            term = (1.f - keep) * s;
            if (lane < 32) {
              xs[idx] = xn;
              if (prep_next_mask) gs[idx] = (1.f - keep) * xn;
            }
          }
          if (lane < 32) myst[row * 32 + cl] = term;
        }
      }
      // row's log-det share of this wave, added in the 16-row form's order: l_q = sum over t, e of column
      // 16 t + 4 q + e, then (l0 + l1) + (l2 + l3)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (lane < ROWS) {
        const float* tr = myst + lane * 32;
        float lq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float a = 0.f;
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) a += tr[16 * t + 4 * q + e];
          lq[q] = a;
        }
        ldw[wave * ROWS + lane] += (lq[0] + lq[1]) + (lq[2] + lq[3]);
      }
    }
    __syncthreads();
  };

  // ---- leapfrog steps ----
  const float two_pi = 6.28318530717958647692f;
  for (int step = p.step_begin; step < p.step_end; ++step) {
    const int sf = step, sb = p.num_steps - 1 - step;
    const float af = two_pi * (float)sf / (float)p.num_steps, ab = two_pi * (float)sb / (float)p.num_steps;
    const float tcf = cosf(af), tsf = sinf(af), tcb = cosf(ab), tsb = sinf(ab);
    float tcr[RG][4], tsr[RG][4];
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        tcr[g][i] = dirr[g][i] ? tcb : tcf;
        tsr[g][i] = dirr[g][i] ? tsb : tsf;
      }
    for (int i = tid; i < D; i += kThreads4) {
      skm[i] = p.masks[(size_t)sf * D + i];
      skm[D + i] = p.masks[(size_t)sb * D + i];
    }
    __syncthreads();
#pragma nounroll
    for (int call = 0; call < 4; ++call) {
      const bool is_v = call == 0 || call == 3;
      if (call == 3) (void)force_pass();
      const int l1 = call == 0 ? (keep_v_valid ? 2 : 0) : call == 1 ? 3 : call == 2 ? 4 : 1;
      net_update(is_v ? p.vnet : p.xnet, is_v ? cv : cx, is_v ? xs : vs, is_v ? 1 : 2, call == 2 ? 1 : 0, call < 2, l1,
                 tcr, tsr, 2 * step + (call == 0 || call == 1 ? 0 : 1));
    }
    keep_v_valid = true;
  }

  // ---- epilogue: energies, accept probability, write back (as fused_traj.hip, kFM -> ROWS) ----
  const float act1 = force_pass();
  const float kin1 = kinetic_pass();
  if (STEPM) {
    if (fl == 0) {
      float sld = 0.f;
#pragma unroll
      for (int w = 0; w < kWaves4; ++w) sld += ldw[w * ROWS + fc];
      const double dh = (double)p.beta * ((double)act0 - (double)act1) + ((double)kin0 - (double)kin1) + (double)sld;
      spx[fc] = accept_from_delta(dh);
    }
    __syncthreads();
    float* gin = gs;
    float* gout = h1;
    for (int i = tid; i < cpw * (D / 4); i += kThreads4) {
      const int k = i / (D / 4), c4 = (i - k * (D / 4)) * 4;
      const int64_t chain = (int64_t)blockIdx.x * cpw + k;
      f32x4 xin = {0.f, 0.f, 0.f, 0.f};
      if (chain < p.step_Bl) xin = *reinterpret_cast<const f32x4*>(p.x0 + chain * D + c4);
      f32x4 xp;
      float pk;
      if (p.step_both) {
        const float fm = scoin[k] > 0.5f ? 1.f : 0.f, bm = 1.f - fm;
        pk = fm * spx[k] + bm * spx[ROWS / 2 + k];
        const f32x4 xf = *reinterpret_cast<const f32x4*>(xs + k * SX + c4);
        const f32x4 xb = *reinterpret_cast<const f32x4*>(xs + (ROWS / 2 + k) * SX + c4);
        xp = fm * xf + bm * xb;
      } else {
        pk = spx[k];
        xp = *reinterpret_cast<const f32x4*>(xs + k * SX + c4);
      }
      const float am = pk > su[k] ? 1.f : 0.f;
      const f32x4 xo = am * xp + (1.f - am) * xin;
      *reinterpret_cast<f32x4*>(gin + k * SX + c4) = xin;
      *reinterpret_cast<f32x4*>(gout + k * SX + c4) = xo;
      if (c4 == 0) sobs[k * 4 + 3] = pk;
      if (chain < p.step_Bl) {
        if (p.step_xprop) *reinterpret_cast<f32x4*>(p.step_xprop + chain * D + c4) = xp;
        if (p.step_xout) *reinterpret_cast<f32x4*>(p.step_xout + chain * D + c4) = xo;
        if (p.step_vprop) {
          f32x4 vp = *reinterpret_cast<const f32x4*>(vs + k * SX + c4);
          if (p.step_both) {
            const float fm = scoin[k] > 0.5f ? 1.f : 0.f, bm = 1.f - fm;
            vp = fm * vp + bm * *reinterpret_cast<const f32x4*>(vs + (ROWS / 2 + k) * SX + c4);
          }
          *reinterpret_cast<f32x4*>(p.step_vprop + chain * D + c4) = vp;
        }
      }
    }
    __syncthreads();
    auto plaq_sums = [&](const float* xc, float& scos, float& sproj) {
      const float inv2pi = 0.15915494309189533577f;
      float a = 0.f, b = 0.f;
      for (int st = fl; st < sites && fl < kSumT; st += kSumT) {
        const int i = st >> xsh, j = st & (X - 1);
        const int jp = (j + 1 == X) ? 0 : j + 1, ip = (i + 1 == T) ? 0 : i + 1;
        const float P = xc[2 * st] - xc[2 * st + 1] - xc[2 * (i * X + jp)] + xc[2 * (ip * X + j) + 1];
        float sn, cs;
        fast_sincos(P, &sn, &cs);
        a += cs;
        b += P - 6.28318530717958647692f * floorf((P + 3.14159265358979323846f) * inv2pi);
      }
      scos = chain_sum(a);
      sproj = chain_sum(b);
    };
    if (p.step_both) {
      float a, b;
      plaq_sums(fc < ROWS / 2 ? gin + fc * SX : gout + (fc - ROWS / 2) * SX, a, b);
      if (fl == 0) {
        if (fc < ROWS / 2) { sobs[fc * 4 + 0] = a; sobs[fc * 4 + 1] = b; }
        else sobs[(fc - ROWS / 2) * 4 + 2] = b;
      }
    } else {
      float a, b, c_, d_;
      plaq_sums(gin + fc * SX, a, b);
      plaq_sums(gout + fc * SX, c_, d_);
      if (fl == 0) { sobs[fc * 4 + 0] = a; sobs[fc * 4 + 1] = b; sobs[fc * 4 + 2] = d_; }
    }
    __syncthreads();
    const float inv2pi = 0.15915494309189533577f;
    if (tid < cpw) {
      const int64_t chain = (int64_t)blockIdx.x * cpw + tid;
      if (chain < p.step_Bl) {
        const float q_in = sobs[tid * 4 + 1] * inv2pi, q_out = sobs[tid * 4 + 2] * inv2pi;
        if (p.step_px) p.step_px[chain] = sobs[tid * 4 + 3];
        if (p.step_act) p.step_act[chain] = (float)sites - sobs[tid * 4 + 0];
        if (p.step_plq) p.step_plq[chain] = sobs[tid * 4 + 0] / (float)sites;
        if (p.step_chg) p.step_chg[chain] = q_in;
        if (p.step_dq) p.step_dq[chain] = fabsf(q_in - q_out);
      }
    }
    if (p.step_sums) {
      int* last = reinterpret_cast<int*>(spx);
      if (tid == 0) {
        float a0 = 0.f, a1 = 0.f;
        for (int k = 0; k < cpw; ++k) {
          if ((int64_t)blockIdx.x * cpw + k < p.step_Bl) {
            a0 += sobs[k * 4 + 3];
            a1 += fabsf(sobs[k * 4 + 1] * inv2pi - sobs[k * 4 + 2] * inv2pi);
          }
        }
        p.step_part[2 * blockIdx.x] = a0;
        p.step_part[2 * blockIdx.x + 1] = a1;
        __threadfence();
        *last = atomicAdd(reinterpret_cast<int*>(p.step_sums + 3), 1) == (int)gridDim.x - 1;
      }
      __syncthreads();
      if (*last) {
        __threadfence();
        float a0 = 0.f, a1 = 0.f;
        for (int b = tid; b < (int)gridDim.x; b += kThreads4) {
          a0 += p.step_part[2 * b];
          a1 += p.step_part[2 * b + 1];
        }
        float* fin = vs;                                  // [2][kThreads4] scratch (vs is dead; sized for it above)
        fin[tid] = a0;
        fin[kThreads4 + tid] = a1;
        __syncthreads();
        for (int st = kThreads4 / 2; st > 0; st >>= 1) {
          if (tid < st) {
            fin[tid] += fin[tid + st];
            fin[kThreads4 + tid] += fin[kThreads4 + tid + st];
          }
          __syncthreads();
        }
        if (tid == 0) {
          p.step_sums[0] = p.step_sums_acc ? p.step_sums[0] + fin[0] : fin[0];
          p.step_sums[1] = p.step_sums_acc ? p.step_sums[1] + fin[kThreads4] : fin[kThreads4];
          p.step_sums[2] = (float)p.step_B;
          *reinterpret_cast<int*>(p.step_sums + 3) = 0;
        }
      }
    }
    for (int i = tid; p.step_x_next && i < cpw * (D / 4); i += kThreads4) {
      const int k = i / (D / 4), c4 = (i - k * (D / 4)) * 4;
      const int64_t chain = (int64_t)blockIdx.x * cpw + k;
      if (chain < p.step_Bl) {
        f32x4 w = *reinterpret_cast<const f32x4*>(gout + k * SX + c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float tp = 6.28318530717958647692f;
          float m_ = fmaf(-tp, floorf(w[e] * 0.15915494309189533577f), w[e]);
          if (m_ < 0.f) m_ += tp;
          if (m_ >= tp) m_ -= tp;
          w[e] = m_;
        }
        *reinterpret_cast<f32x4*>(p.step_x_next + chain * D + c4) = w;
      }
    }
    return;
  }
  if (fl == 0 && fc < nrow) {
    float sld = 0.f;
#pragma unroll
    for (int w = 0; w < kWaves4; ++w) sld += ldw[w * ROWS + fc];
    const int64_t rr = row0 + fc;
    if (p.logdet) p.logdet[rr] = p.logdet_accumulate ? p.logdet[rr] + sld : sld;
    if (p.p_accept) {
      const double dh = (double)p.beta * ((double)act0 - (double)act1) + ((double)kin0 - (double)kin1) + (double)sld;
      p.p_accept[rr] = accept_from_delta(dh);
    }
  }
  for (int i = tid; i < ROWS * (D / 4); i += kThreads4) {
    const int rr = i / (D / 4), c4 = (i - rr * (D / 4)) * 4;
    if (rr < nrow) {
      *reinterpret_cast<f32x4*>(p.x_out + (row0 + rr) * D + c4) = *reinterpret_cast<const f32x4*>(xs + rr * SX + c4);
      *reinterpret_cast<f32x4*>(p.v_out + (row0 + rr) * D + c4) = *reinterpret_cast<const f32x4*>(vs + rr * SX + c4);
    }
  }
}

}  // namespace

size_t fused4_pack_floats(const l2hmc_dense_net* n) {
  return (n->D == kD && n->H == kH && n->Ka == kD && n->Kb == kD) ? kP1 + kP2 + kP4H : 0;
}

int launch_fused4_pack(const l2hmc_dense_net* n, float* image4, hipStream_t stream) {
  hipLaunchKernelGGL(pack_fused4_kernel, dim3(1024), dim3(256), 0, stream, *n, image4);
  L2HMC_CHECK_LAUNCH("dense_pack (sub-tile image)");
  return L2HMC_OK;
}

// rows per workgroup of the sub-tile form, or 0 where 16-row tiles already cover the CUs
int fused4_rows_per_wg(int64_t rows, int cus) {
  // (measured and dropped: two 4-row workgroups per CU for 1024 < rows <= 2048 -- 512 workgroups streaming the weights
  //  ask the L2s for ~55 TB/s: 1.61 ms against 1.07 ms for one 8-row workgroup per CU)
  if (rows <= 4 * (int64_t)cus) return 4;
  if (rows <= 8 * (int64_t)cus) return 8;
  if (rows <= 12 * (int64_t)cus) return 12;
  return 0;
}

int launch_fused4(const FusedArgs& a, int rows_per_wg, hipStream_t stream) {
  static DeviceOnce once;
  const size_t lds4 = sizeof(float) * F4Cfg<4>::LDS_FLOATS, lds8 = sizeof(float) * F4Cfg<8>::LDS_FLOATS,
               lds12 = sizeof(float) * F4Cfg<12>::LDS_FLOATS;
  if (once.pending()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gauge_traj_fused4_kernel<4>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gauge_traj_fused4_kernel<8>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gauge_traj_fused4_kernel<12>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds12) != hipSuccess) {
      set_error("fused sub-tile kernel: cannot reserve %zu B of LDS", lds12);
      return L2HMC_ERR_HIP;
    }
    once.done();
  }
  L2HMC_REQUIRE(rows_per_wg == 4 || rows_per_wg == 8 || rows_per_wg == 12, "fused sub-tile kernel: %d rows per workgroup",
                rows_per_wg);
  const dim3 grid((unsigned)ceil_div(a.rows, rows_per_wg));
  prof_before(kProfFused, stream);
  if (rows_per_wg == 4)
    hipLaunchKernelGGL((gauge_traj_fused4_kernel<4>), grid, dim3(kThreads4), lds4, stream, a);
  else if (rows_per_wg == 8)
    hipLaunchKernelGGL((gauge_traj_fused4_kernel<8>), grid, dim3(kThreads4), lds8, stream, a);
  else
    hipLaunchKernelGGL((gauge_traj_fused4_kernel<12>), grid, dim3(kThreads4), lds12, stream, a);
  prof_after(kProfFused, stream);
  L2HMC_CHECK_LAUNCH("gauge_traj_fused4");
  return L2HMC_OK;
}

}  // namespace l2hmc
