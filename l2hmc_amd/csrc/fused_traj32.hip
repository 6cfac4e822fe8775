// 32-row form of the whole-trajectory kernel (fused_traj.hip) for batches of more than one round of 16-row tiles.
//
// With 16 rows per workgroup every weight fragment a wave loads feeds ONE v_mfma_f32_16x16x4_f32 per k, and the issue
// cost of that load (about 19 cycles of the wave's single instruction stream per global_load_dwordx4) caps the product
// loops at 0.83 of the matrix pipe (tools/mfma_stream_bench.hip: 38.1 cycles per MFMA against 32).  Here a workgroup
// owns 32 rows = two row groups that share every fragment: one load per eight MFMAs, 35.0-35.8 cycles per MFMA in the
// same bench (profiles/r03_mfma_stream_bench.txt).  It needs 32 rows per CU, i.e. more than 4096 rows per GPU, so it
// takes the batches the 16-row form would need two or more rounds for.  Round 4: buffer loads with the pinned
// interleave (fused_common.h), then 8 waves per workgroup on the 4-wave image (8 accumulator tiles per wave).
//
// Same algorithm, same packed weight image, same arithmetic ORDER as fused_traj.hip
// (l2hmc/dynamics/gauge_dynamics.py:261-313, :412-609; network/generic_net.py:129-146): a row's accumulators see
// their k in the same order, epilogue expressions, log-det grouping, chain-local sums (16 lanes per chain) and the
// Philox indexing are the 16-row form's -- results are bit-identical
// (tests/test_gpu_parity.py::test_subtile_and_32_row_forms_equal_16_row_form).
//
// LDS: x, v, force rows (3 x 32 x 136), ONE hidden buffer (32 x 520; the second layer's output overwrites its input
// behind an extra barrier -- two buffers would need 218 KB), constants, masks, scratch: 153.7 KB.  GenericNet on the
// 8x8 lattice (D = 128, H = 512), sampling only (no tape, no ConvNet3D).
#include "fused_common.h"
#include "fused_args.h"

namespace l2hmc {

namespace {
constexpr int kD = 128, kH = 512;
constexpr int kR32 = 32, kG32 = 2;                         // rows per workgroup, row groups of 16
// Two waves per SIMD since round 4 (fused_traj.hip: FusedCfg has the story): 8 waves read the 4-wave image, two waves
// sharing a section; the chain-local passes and the fixed-order reductions stay on the first 256 threads (their sums
// are part of the result's bits).
constexpr int kIW32 = 4;                                   // waves the packed image is laid out for
constexpr int kW32 = 8, kT32 = 64 * kW32;                  // waves, threads
constexpr int kRW32 = kW32 / kIW32;                        // waves per image section
constexpr int kCT32 = 256;                                 // threads of the chain-local passes and reductions
constexpr int kSX = kD + 8, kSH = kH + 8, kSP = kD / 2 + 4;
constexpr int kNT1 = kH / (16 * kW32), kNTH = kD / (16 * kW32);      // 4 tiles per wave in layers 1 / 2, 1 per head
constexpr int kNTI1 = kH / (16 * kIW32), kNTIH = kD / (16 * kIW32);  // ... 8 and 2 per image section
static_assert(kRW32 == 2 && kNTH == 1, "a shared heads section is walked with one tile per head and wave");
constexpr int kKC1 = 2 * kD / 16, kKC2 = kH / 16;
constexpr size_t kP1 = (size_t)2 * kD * kH, kP2 = (size_t)kH * kH;
constexpr int kNC = 4 * kH + 5 * kD;
constexpr int kTPC = 16;                                   // threads per chain in the chain-local passes (two passes of 16 chains)
constexpr int kLds32 = 3 * kR32 * kSX + kR32 * kSH + 2 * kNC + kR32 * kSP + 2 * kD + kIW32 * kR32 /*ldw*/ +
                       kIW32 * kG32 * 64 /*ldx*/ + kR32 + 8 * kR32;
static_assert(kLds32 * sizeof(float) <= 160 * 1024, "LDS of the 32-row form");

// Half a block for both row groups (fused_common.h: mfma_half_stream): every weight fragment cur[t] feeds two MFMAs,
// so one load goes behind every EIGHTH MFMA.  e-major per accumulator, as the 16-row form.
template <int NT, int T0, int T1, int L0, int L1, bool LOAD, int NDS = 0, int NTI = NT, int TS = 1>
__device__ __forceinline__ void mfma_half_stream2(const f32x4 a0, const f32x4 a1, const f32x4 (&cur)[NT], f32x4 (&dst)[NT],
                                                  f32x4 (&acc0)[NT], f32x4 (&acc1)[NT], const WSection& wp, int kc,
                                                  int toff = 0) {
  constexpr int G = T1 - T0, NLD = L1 - L0;
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int t = T0; t < T1; ++t) {
      acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[t][e], a0[e], acc0[t], 0, 0, 0);
      acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[t][e], a1[e], acc1[t], 0, 0, 0);
    }
  if constexpr (LOAD) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) dst[L0 + j] = lane_frag(wp, (unsigned)(kc * NTI + toff + (L0 + j) * TS));
  }
  if constexpr (NDS > 0) __builtin_amdgcn_sched_group_barrier(0x100, NDS, 0);     // the next block's two A fragments first
  sched_mfma_load_pipeline<G, LOAD ? NLD : 0, 8>();
}

// stream_layer (fused_common.h) for two row groups; ap = the lane's fragment address in row group 0, gstride = floats
// between the groups' rows
// NTI, TS, toff: the wave's tiles in a shared image section (fused_common.h: load_frags)
template <int NT, int NKC, int DEPTH, int NTI = NT, int TS = 1>
__device__ __forceinline__ void stream_layer2(BRing<NT, DEPTH>& R, const float* __restrict__ wbase, const float* ap,
                                              int gstride, f32x4 (&acc0)[NT], f32x4 (&acc1)[NT], bool rev = false,
                                              int toff = 0) {
  const WSection wp = wsection(wbase);
  static_assert(DEPTH >= 2 && NKC >= DEPTH && NT >= 2, "ring depth / tile groups");
  constexpr int G = (NT + 1) / 2;
  auto km = [&](int k) { return rev ? NKC - 1 - k : k; };
  auto af = [&](int kc, int g) { return *reinterpret_cast<const f32x4*>(ap + g * gstride + kc * 16); };
  constexpr int MAIN = (NKC - DEPTH) / DEPTH * DEPTH;
  __builtin_amdgcn_sched_barrier(0);                           // (fused_common.h: stream_layer)
  f32x4 a0 = af(km(0), 0), c0 = af(km(0), 1);
  int kc = 0;
#pragma nounroll
  for (; kc < MAIN; kc += DEPTH) {
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) {
      const int kn = km(kc + s + 1);
      const f32x4 a1 = af(kn, 0), c1 = af(kn, 1);
      mfma_half_stream2<NT, 0, G, G, NT, true, 2, NTI, TS>(a0, c0, R.b[s], R.b[(s + DEPTH - 1) % DEPTH], acc0, acc1, wp,
                                                        km(kc + s + DEPTH - 1), toff);
      mfma_half_stream2<NT, G, NT, 0, G, true, 0, NTI, TS>(a0, c0, R.b[s], R.b[s], acc0, acc1, wp, km(kc + s + DEPTH), toff);
      a0 = a1;
      c0 = c1;
    }
  }
#pragma unroll
  for (int k = MAIN; k < NKC; ++k) {
    const int s = (k - MAIN) % DEPTH;
    const int kn = km(k + 1 < NKC ? k + 1 : NKC - 1);
    const f32x4 a1 = af(kn, 0), c1 = af(kn, 1);
    if (k + DEPTH - 1 < NKC)
      mfma_half_stream2<NT, 0, G, G, NT, true, 2, NTI, TS>(a0, c0, R.b[s], R.b[(s + DEPTH - 1) % DEPTH], acc0, acc1, wp,
                                                        km(k + DEPTH - 1), toff);
    else
      mfma_half_stream2<NT, 0, G, G, NT, false, 2>(a0, c0, R.b[s], R.b[(s + DEPTH - 1) % DEPTH], acc0, acc1, wp, 0);
    if (k + DEPTH < NKC)
      mfma_half_stream2<NT, G, NT, 0, G, true, 0, NTI, TS>(a0, c0, R.b[s], R.b[s], acc0, acc1, wp, km(k + DEPTH), toff);
    else
      mfma_half_stream2<NT, G, NT, 0, G, false>(a0, c0, R.b[s], R.b[s], acc0, acc1, wp, 0);
    a0 = a1;
    c0 = c1;
  }
  __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(kT32) void gauge_traj_fused32_kernel(FusedArgs p) {
  constexpr int ROWS = kR32, SX = kSX, SH = kSH, SP = kSP, D = kD, H = kH, NT1 = kNT1, NTH = kNTH;
  constexpr int sites = D / 2;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xs = lds;                         // [32][SX] position
  float* vs = xs + ROWS * SX;              // [32][SX] momentum
  float* gs = vs + ROWS * SX;              // [32][SX] second net input: force, or keep (.) x
  float* hh = gs + ROWS * SX;              // [32][SH] h1, then h2 in place
  float* cx = hh + ROWS * SH;              // XNet constants [NC]
  float* cv = cx + kNC;
  float* sp = cv + kNC;                    // [32][SP] sin P
  float* skm = sp + ROWS * SP;             // [2][D] masks of this step: forward row, backward row
  float* ldw = skm + 2 * D;                // [image waves][32] log-det partial sums per image wave
  float* ldx = ldw + kIW32 * ROWS;         // [image waves][2 row groups][64] lane sums an even wave hands to its odd partner
  int* sdir = reinterpret_cast<int*>(ldx + kIW32 * kG32 * 64);   // [32]
  float* stp = reinterpret_cast<float*>(sdir + ROWS);      // step mode: coin[32] u[32] p_row[32] obs[32][4]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, r = lane & 15;
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  const int nrow = (int)min((int64_t)ROWS, p.rows - row0);
  const float eps = p.eps;

  // ---- stage chain state and constants (as fused_traj.hip, kFM -> 32) ----
  const bool STEPM = p.step_B > 0;
  const int cpw = STEPM ? (p.step_both ? ROWS / 2 : ROWS) : ROWS;
  float* scoin = stp;
  float* su = stp + ROWS;
  float* spx = stp + 2 * ROWS;
  float* sobs = stp + 3 * ROWS;            // [32][4]
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
    for (int i = tid; i < ROWS * (D / 4); i += kT32) {
      const int rr = i / (D / 4), c4 = (i - rr * (D / 4)) * 4;
      const int k = p.step_both ? (rr & (ROWS / 2 - 1)) : rr;
      const int64_t chain = (int64_t)blockIdx.x * cpw + k;
      const int dsel = p.step_both ? (rr >= ROWS / 2 ? 1 : 0) : (scoin[k] > 0.5f ? 0 : 1);   // gauge_dynamics.py:221-227
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
    for (int i = tid; i < ROWS * (D / 4); i += kT32) {
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
    for (int i = tid; i < H; i += kT32) {
      c[i] = n.b1[i];
      c[H + i] = n.wt[i];
      c[2 * H + i] = n.wt[H + i];
      c[3 * H + i] = n.bh[i];
    }
    for (int i = tid; i < 3 * D; i += kT32) c[4 * H + i] = n.bhd[i];
    for (int i = tid; i < D; i += kT32) {
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
  if (tid < kIW32 * ROWS) ldw[tid] = 0.f;
  __syncthreads();

  // direction of the rows this lane owns in its C fragments: row 16 g + r
  const int dirl[kG32] = {sdir[r], sdir[16 + r]};

  // ---- chain-local passes: 16 consecutive threads per chain, two passes of 16 chains (chain = 16 h + tid / 16);
  //      the terms of a chain are strided by 16 and summed by a butterfly over 16 lanes, as in the 16-row form
  // (the threads beyond 256 walk empty loops there and take part in the barriers only)
  const bool own = tid < kCT32;
  const int fc0 = own ? tid / kTPC : 0, fl = tid % kTPC;
  const int sites_l = own ? sites : 0, D_l = own ? D : 0;
  auto chain_sum = [&](float v) {
#pragma unroll
    for (int off = kTPC / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
  };
  const int T = p.T, X = p.X;
  const int xsh = 31 - __clz(X);
  // force (beta * dS/dx) into gs; act[h] = the action of chain 16 h + fc0 (all 16 lanes of the chain)
  auto force_pass = [&](float (&act)[kG32]) {
#pragma unroll
    for (int h = 0; h < kG32; ++h) {
      const int fc = 16 * h + fc0;
      const float* xc = xs + fc * SX;
      float a = 0.f;
      for (int s = fl; s < sites_l; s += kTPC) {
        const int i = s >> xsh, j = s & (X - 1);
        const int jp = (j + 1 == X) ? 0 : j + 1, ip = (i + 1 == T) ? 0 : i + 1;
        const float P = xc[2 * s] - xc[2 * s + 1] - xc[2 * (i * X + jp)] + xc[2 * (ip * X + j) + 1];
        float sn, cs;
        fast_sincos(P, &sn, &cs);
        sp[fc * SP + s] = sn;
        a += 1.f - cs;
      }
      act[h] = chain_sum(a);
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < kG32; ++h) {
      const int fc = 16 * h + fc0;
      float* gc = gs + fc * SX;
      const float* spc = sp + fc * SP;
      for (int s = fl; s < sites_l; s += kTPC) {
        const int i = s >> xsh, j = s & (X - 1);
        const int jm = (j == 0) ? X - 1 : j - 1, im = (i == 0) ? T - 1 : i - 1;
        const float sP = spc[s];
        gc[2 * s] = p.beta * (sP - spc[i * X + jm]);
        gc[2 * s + 1] = p.beta * (-sP + spc[im * X + j]);
      }
    }
    __syncthreads();
  };
  auto kinetic_pass = [&](float (&kin)[kG32]) {
#pragma unroll
    for (int h = 0; h < kG32; ++h) {
      const float* vc = vs + (16 * h + fc0) * SX;
      float k = 0.f;
      for (int d = fl; d < D_l; d += kTPC) k += vc[d] * vc[d];
      kin[h] = 0.5f * chain_sum(k);
    }
  };
  float act0[kG32], kin0[kG32];
  force_pass(act0);                        // also leaves the force of x0 in gs
  kinetic_pass(kin0);

  // recurring first-layer products kept in registers (fused_traj.hip: keep_v, keep_x)
  f32x4 keep_v[kG32][NT1], keep_x[kG32][NT1];
  bool keep_v_valid = false;

  // l1: 0 = compute both halves; 1 = as 0 and store the raw product in keep_v; 2 = take keep_v, no GEMM;
  //     3 = compute, snapshot the first-input half into keep_x; 4 = start from keep_x, second half only.
  auto net_update = [&](const l2hmc_dense_net& net, const float* cn, const float* in1, int mode, int sub,
                        bool prep_next_mask, int l1, const float (&tcr)[kG32], const float (&tsr)[kG32], int callidx) {
    const float* pk = net.packed;
    const bool zig = (callidx & 1) != 0;   // layers 2 and 3 alternate their direction per call (fused_common.h)
    const int wv = __builtin_amdgcn_readfirstlane(wave);      // provably uniform: the weight loads' base stays in SGPRs
    const int wimg = wv / kRW32, wsub = wv - wimg * kRW32;    // image section, and this wave's share of it
    const float* wp1 = pk + (size_t)wimg * kKC1 * kNTI1 * 256;
    const float* wp2 = pk + kP1 + (size_t)wimg * kKC2 * kNTI1 * 256;
    const float* wph = pk + kP1 + kP2 + (size_t)wimg * kKC2 * 3 * kNTIH * 256;
    const int to1 = wsub * NT1, toh = wsub * NTH;              // first tile of this wave in a chunk of the section
    constexpr int TSH = kNTIH;                                 // tile stride of a wave's heads fragments
    float ld_s[kG32][4];                                       // the odd wave's log-det terms (the mask factor is re-read)
    constexpr int DP1 = 2, DP2 = 3, DPH = 3;   // (ring depths; with two waves per SIMD they no longer matter: fused_traj.hip)
    BRing<NT1, DP2> R2;
    BRing<3 * NTH, DPH> R3;
    // ----- layer 1: two half-K streams (first input rows, then the second-input rows in gs)
    {
      constexpr int KH = kKC1 / 2;
      f32x4 acc[kG32][NT1];
      if (l1 == 2) {
#pragma unroll
        for (int g = 0; g < kG32; ++g)
#pragma unroll
          for (int t = 0; t < NT1; ++t) acc[g][t] = keep_v[g][t];
      } else {
        BRing<NT1, DP1> RA;
        const float* wpb = wp1 + (size_t)KH * kNTI1 * 256;
        if (l1 == 4) {
#pragma unroll
          for (int g = 0; g < kG32; ++g)
#pragma unroll
            for (int t = 0; t < NT1; ++t) acc[g][t] = keep_x[g][t];
        } else {
          ring_prime<NT1, DP1, kNTI1>(RA, wp1, false, 0, to1);
#pragma unroll
          for (int g = 0; g < kG32; ++g)
#pragma unroll
            for (int t = 0; t < NT1; ++t) acc[g][t] = f32x4{0.f, 0.f, 0.f, 0.f};
          stream_layer2<NT1, KH, DP1, kNTI1>(RA, wp1, in1 + r * SX + q * 4, 16 * SX, acc[0], acc[1], false, to1);
          if (l1 == 3) {
#pragma unroll
            for (int g = 0; g < kG32; ++g)
#pragma unroll
              for (int t = 0; t < NT1; ++t) keep_x[g][t] = acc[g][t];
          }
        }
        ring_prime<NT1, DP1, kNTI1>(RA, wpb, false, 0, to1);
        stream_layer2<NT1, KH, DP1, kNTI1>(RA, wpb, gs + r * SX + q * 4, 16 * SX, acc[0], acc[1], false, to1);
        if (l1 == 1) {
#pragma unroll
          for (int g = 0; g < kG32; ++g)
#pragma unroll
            for (int t = 0; t < NT1; ++t) keep_v[g][t] = acc[g][t];
        }
      }
      ring_prime<NT1, DP2, kNTI1>(R2, wp2, zig, kKC2, to1);      // layer-2 weights start flowing under the epilogue + barrier
#pragma unroll
      for (int t = 0; t < NT1; ++t) {
        const int c0 = (wave * NT1 + t) * 16 + q * 4;          // this lane: rows r and 16 + r, columns c0 .. c0 + 3
        const f32x4 b = *reinterpret_cast<const f32x4*>(cn + c0);
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(cn + H + c0);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(cn + 2 * H + c0);
#pragma unroll
        for (int g = 0; g < kG32; ++g) {
          f32x4 hv;
#pragma unroll
          for (int e = 0; e < 4; ++e) hv[e] = fmaxf(acc[g][t][e] + b[e] + (tcr[g] * w0[e] + tsr[g] * w1[e]), 0.f);
          *reinterpret_cast<f32x4*>(hh + (16 * g + r) * SH + c0) = hv;
        }
      }
    }
    __syncthreads();
    // ----- layer 2 (its output overwrites its input once every wave has finished reading it)
    {
      f32x4 acc[kG32][NT1];
#pragma unroll
      for (int g = 0; g < kG32; ++g)
#pragma unroll
        for (int t = 0; t < NT1; ++t) acc[g][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      stream_layer2<NT1, kKC2, DP2, kNTI1>(R2, wp2, hh + r * SH + q * 4, 16 * SH, acc[0], acc[1], zig, to1);
      ring_prime<3 * NTH, DPH, 3 * kNTIH, TSH>(R3, wph, zig, kKC2, toh);
      __syncthreads();
#pragma unroll
      for (int t = 0; t < NT1; ++t) {
        const int c0 = (wave * NT1 + t) * 16 + q * 4;
        const f32x4 b = *reinterpret_cast<const f32x4*>(cn + 3 * H + c0);
#pragma unroll
        for (int g = 0; g < kG32; ++g) {
          f32x4 hv;
#pragma unroll
          for (int e = 0; e < 4; ++e) hv[e] = fmaxf(acc[g][t][e] + b[e], 0.f);
          *reinterpret_cast<f32x4*>(hh + (16 * g + r) * SH + c0) = hv;
        }
      }
    }
    __syncthreads();
    // ----- heads + update
    {
      f32x4 acc[kG32][3 * NTH];
#pragma unroll
      for (int g = 0; g < kG32; ++g)
#pragma unroll
        for (int t = 0; t < 3 * NTH; ++t) acc[g][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      stream_layer2<3 * NTH, kKC2, DPH, 3 * kNTIH, TSH>(R3, wph, hh + r * SH + q * 4, 16 * SH, acc[0], acc[1], zig, toh);
      const float* bhd = cn + 4 * H;
      const float* es = bhd + 3 * D;
      const float* eq = es + D;
#pragma unroll
      for (int g = 0; g < kG32; ++g) {
        float ld = 0.f;                     // this lane's share of row (16 g + r)'s log-det
        const int d = dirl[g];
#pragma unroll
        for (int t = 0; t < NTH; ++t) {
          const int c0 = wave * (D / kW32) + t * 16 + q * 4;
          const f32x4 b_s = *reinterpret_cast<const f32x4*>(bhd + c0);
          const f32x4 b_t = *reinterpret_cast<const f32x4*>(bhd + D + c0);
          const f32x4 b_q = *reinterpret_cast<const f32x4*>(bhd + 2 * D + c0);
          const f32x4 e_s = *reinterpret_cast<const f32x4*>(es + c0);
          const f32x4 e_q = *reinterpret_cast<const f32x4*>(eq + c0);
          const f32x4 mf = *reinterpret_cast<const f32x4*>(skm + c0);
          const f32x4 mb = *reinterpret_cast<const f32x4*>(skm + D + c0);
          const int idx = (16 * g + r) * SX + c0;
          f32x4 S, Tt, Q;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            S[e] = fast_tanh(acc[g][0 * NTH + t][e] + b_s[e]) * e_s[e];
            Tt[e] = acc[g][1 * NTH + t][e] + b_t[e];
            const float qq = acc[g][2 * NTH + t][e] + b_q[e];
            Q[e] = (net.q_tanh ? fast_tanh(qq) : qq) * e_q[e];
          }
          if (mode == 1) {
            // gauge_dynamics.py:497-506 (fwd), :549-559 (bwd)
            const f32x4 gg = *reinterpret_cast<const f32x4*>(gs + idx);
            const f32x4 v = *reinterpret_cast<const f32x4*>(vs + idx);
            f32x4 vn;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float s = (d ? -0.5f : 0.5f) * eps * S[e];
              const float kick = 0.5f * eps * (fast_exp(eps * Q[e]) * gg[e] - Tt[e]);
              const float es_ = fast_exp(s);
              vn[e] = d ? es_ * (v[e] + kick) : v[e] * es_ - kick;
              ld += s;
              ld_s[g][e] = s;
            }
            *reinterpret_cast<f32x4*>(vs + idx) = vn;
            if (prep_next_mask) {
              const f32x4 x = *reinterpret_cast<const f32x4*>(xs + idx);
              f32x4 kx;
#pragma unroll
              for (int e = 0; e < 4; ++e) kx[e] = (d ? 1.f - mb[e] : mf[e]) * x[e];
              *reinterpret_cast<f32x4*>(gs + idx) = kx;
            }
          } else {
            // gauge_dynamics.py:519-531 (fwd), :574-584 (bwd); keep mask per direction and sub-update
            const f32x4 x = *reinterpret_cast<const f32x4*>(xs + idx);
            const f32x4 v = *reinterpret_cast<const f32x4*>(vs + idx);
            f32x4 xn, kx;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float keep = sub == 0 ? (d ? 1.f - mb[e] : mf[e]) : (d ? mb[e] : 1.f - mf[e]);
              const float s = (d ? -eps : eps) * S[e];
              const float drift = eps * (fast_exp(eps * Q[e]) * v[e] + Tt[e]);
              const float es_ = fast_exp(s);
              const float upd = d ? es_ * (x[e] - drift) : x[e] * es_ + drift;
              xn[e] = keep * x[e] + (1.f - keep) * upd;
              ld += (1.f - keep) * s;
              ld_s[g][e] = s;
              kx[e] = (1.f - keep) * xn[e];
            }
            *reinterpret_cast<f32x4*>(xs + idx) = xn;
            if (prep_next_mask) *reinterpret_cast<f32x4*>(gs + idx) = kx;
          }
        }
        // The bits of the 4-wave form are those of ONE chain of adds per lane over both waves' head columns, then the
        // cross-lane steps: the even wave hands its lane sum over, the odd wave continues the chain with its own four
        // terms behind the barrier below (ld_k, ld_s; fused_traj.hip has the same hand-off) and does the rest.
        if (wsub == 0) ldx[(wimg * kG32 + g) * 64 + lane] = ld;
      }
    }
    __syncthreads();
    if (wsub == 1) {
#pragma unroll
      for (int g = 0; g < kG32; ++g) {
        float ld = ldx[(wimg * kG32 + g) * 64 + lane];
        const int d = dirl[g];
        const int c0 = wave * (D / kW32) + q * 4;                        // (NTH = 1: the wave's one head tile)
        const f32x4 mf = *reinterpret_cast<const f32x4*>(skm + c0);
        const f32x4 mb = *reinterpret_cast<const f32x4*>(skm + D + c0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {                                    // the statements of the chain above, verbatim
          const float s = ld_s[g][e];
          if (mode == 1) {
            ld += s;
          } else {
            const float keep = sub == 0 ? (d ? 1.f - mb[e] : mf[e]) : (d ? mb[e] : 1.f - mf[e]);
            ld += (1.f - keep) * s;
          }
        }
        // the row's log-det share of this image wave: lanes r, r + 16, r + 32, r + 48 (fixed order: bit-reproducible)
        ld += __shfl_xor(ld, 16, 64);
        ld += __shfl_xor(ld, 32, 64);
        if (q == 0) ldw[wimg * ROWS + 16 * g + r] += ld;
      }
    }
  };

  // ---- leapfrog steps ----
  const float two_pi = 6.28318530717958647692f;
  for (int step = p.step_begin; step < p.step_end; ++step) {
    const int sf = step, sb = p.num_steps - 1 - step;       // gauge_dynamics.py:453-457
    const float af = two_pi * (float)sf / (float)p.num_steps, ab = two_pi * (float)sb / (float)p.num_steps;
    const float tcf = cosf(af), tsf = sinf(af), tcb = cosf(ab), tsb = sinf(ab);
    const float tcr[kG32] = {dirl[0] ? tcb : tcf, dirl[1] ? tcb : tcf};
    const float tsr[kG32] = {dirl[0] ? tsb : tsf, dirl[1] ? tsb : tsf};
    for (int i = tid; i < D; i += kT32) {
      skm[i] = p.masks[(size_t)sf * D + i];
      skm[D + i] = p.masks[(size_t)sb * D + i];
    }
    __syncthreads();
#pragma nounroll
    for (int call = 0; call < 4; ++call) {
      const bool is_v = call == 0 || call == 3;
      if (call == 3) {
        float dummy[kG32];
        force_pass(dummy);                                     // force at the new position
      }
      const int l1 = call == 0 ? (keep_v_valid ? 2 : 0) : call == 1 ? 3 : call == 2 ? 4 : 1;
      net_update(is_v ? p.vnet : p.xnet, is_v ? cv : cx, is_v ? xs : vs, is_v ? 1 : 2, call == 2 ? 1 : 0, call < 2, l1,
                 tcr, tsr, 2 * step + (call == 0 || call == 1 ? 0 : 1));
    }
    keep_v_valid = true;
  }

  // ---- epilogue: energies, accept probability, write back (as fused_traj.hip, kFM -> 32, two passes of 16 chains) ----
  float act1[kG32], kin1[kG32];
  force_pass(act1);
  kinetic_pass(kin1);
  if (STEPM) {
    if (own && fl == 0) {
#pragma unroll
      for (int h = 0; h < kG32; ++h) {
        const int fc = 16 * h + fc0;
        float sld = 0.f;
#pragma unroll
        for (int w = 0; w < kIW32; ++w) sld += ldw[w * ROWS + fc];
        const double dh = (double)p.beta * ((double)act0[h] - (double)act1[h]) + ((double)kin0[h] - (double)kin1[h]) +
                          (double)sld;
        spx[fc] = accept_from_delta(dh);
      }
    }
    __syncthreads();
    // ---- mix the two directions, Metropolis-Hastings; x_in -> gs rows, x_out -> hh rows (both free now)
    float* gin = gs;
    float* gout = hh;
    for (int i = tid; i < cpw * (D / 4); i += kT32) {
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
      const float am = pk > su[k] ? 1.f : 0.f;                       // strict >, quirk Q5
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
    // ---- observables of the step's INPUT samples and the charge of its output
    auto plaq_sums = [&](const float* xc, float& scos, float& sproj) {
      const float inv2pi = 0.15915494309189533577f;
      float a = 0.f, b = 0.f;
      for (int st = fl; st < sites_l; st += kTPC) {
        const int i = st >> xsh, j = st & (X - 1);
        const int jp = (j + 1 == X) ? 0 : j + 1, ip = (i + 1 == T) ? 0 : i + 1;
        const float P = xc[2 * st] - xc[2 * st + 1] - xc[2 * (i * X + jp)] + xc[2 * (ip * X + j) + 1];
        float sn, cs;
        fast_sincos(P, &sn, &cs);
        a += cs;
        b += P - 6.28318530717958647692f * floorf((P + 3.14159265358979323846f) * inv2pi);   // project_angle
      }
      scos = chain_sum(a);
      sproj = chain_sum(b);
    };
#pragma unroll
    for (int h = 0; h < kG32; ++h) {
      const int fc = 16 * h + fc0;
      if (p.step_both) {
        float a, b;
        plaq_sums(fc < ROWS / 2 ? gin + fc * SX : gout + (fc - ROWS / 2) * SX, a, b);
        if (own && fl == 0) {
          if (fc < ROWS / 2) { sobs[fc * 4 + 0] = a; sobs[fc * 4 + 1] = b; }
          else sobs[(fc - ROWS / 2) * 4 + 2] = b;
        }
      } else {
        float a, b, c_, d_;
        plaq_sums(gin + fc * SX, a, b);
        plaq_sums(gout + fc * SX, c_, d_);
        if (own && fl == 0) { sobs[fc * 4 + 0] = a; sobs[fc * 4 + 1] = b; sobs[fc * 4 + 2] = d_; }
      }
    }
    __syncthreads();
    const float inv2pi = 0.15915494309189533577f;
    if (tid < cpw) {
      const int64_t chain = (int64_t)blockIdx.x * cpw + tid;
      if (chain < p.step_Bl) {
        const float q_in = sobs[tid * 4 + 1] * inv2pi, q_out = sobs[tid * 4 + 2] * inv2pi;
        if (p.step_px) p.step_px[chain] = sobs[tid * 4 + 3];
        if (p.step_act) p.step_act[chain] = (float)sites - sobs[tid * 4 + 0];      // sum (1 - cos P)
        if (p.step_plq) p.step_plq[chain] = sobs[tid * 4 + 0] / (float)sites;
        if (p.step_chg) p.step_chg[chain] = q_in;
        if (p.step_dq) p.step_dq[chain] = fabsf(q_in - q_out);
      }
    }
    if (p.step_sums) {
      int* last = reinterpret_cast<int*>(spx);            // spx is free again
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
        float a0 = 0.f, a1 = 0.f;                         // (256 partial sums and their tree, whatever the thread count)
        for (int b = tid; own && b < (int)gridDim.x; b += kCT32) {
          a0 += p.step_part[2 * b];
          a1 += p.step_part[2 * b + 1];
        }
        float* fin = vs;                                  // [2][256] scratch (vs is dead)
        if (own) {
          fin[tid] = a0;
          fin[kCT32 + tid] = a1;
        }
        __syncthreads();
        for (int st = kCT32 / 2; st > 0; st >>= 1) {
          if (tid < st) {
            fin[tid] += fin[tid + st];
            fin[kCT32 + tid] += fin[kCT32 + tid + st];
          }
          __syncthreads();
        }
        if (tid == 0) {
          p.step_sums[0] = p.step_sums_acc ? p.step_sums[0] + fin[0] : fin[0];
          p.step_sums[1] = p.step_sums_acc ? p.step_sums[1] + fin[kCT32] : fin[kCT32];
          p.step_sums[2] = (float)p.step_B;
          *reinterpret_cast<int*>(p.step_sums + 3) = 0;
        }
      }
    }
    // ---- np.mod(x_out, 2 pi) (gauge_model.py:1388) and the write-back of the chains' new state
    for (int i = tid; p.step_x_next && i < cpw * (D / 4); i += kT32) {
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
  if (own && fl == 0) {
#pragma unroll
    for (int h = 0; h < kG32; ++h) {
      const int fc = 16 * h + fc0;
      if (fc < nrow) {
        float sld = 0.f;
#pragma unroll
        for (int w = 0; w < kIW32; ++w) sld += ldw[w * ROWS + fc];
        const int64_t rr = row0 + fc;
        if (p.logdet) p.logdet[rr] = p.logdet_accumulate ? p.logdet[rr] + sld : sld;
        if (p.p_accept) {
          const double dh = (double)p.beta * ((double)act0[h] - (double)act1[h]) + ((double)kin0[h] - (double)kin1[h]) +
                            (double)sld;
          p.p_accept[rr] = accept_from_delta(dh);
        }
      }
    }
  }
  for (int i = tid; i < ROWS * (D / 4); i += kT32) {
    const int rr = i / (D / 4), c4 = (i - rr * (D / 4)) * 4;
    if (rr < nrow) {
      *reinterpret_cast<f32x4*>(p.x_out + (row0 + rr) * D + c4) = *reinterpret_cast<const f32x4*>(xs + rr * SX + c4);
      *reinterpret_cast<f32x4*>(p.v_out + (row0 + rr) * D + c4) = *reinterpret_cast<const f32x4*>(vs + rr * SX + c4);
    }
  }
}

}  // namespace

int fused32_supported(const l2hmc_dense_net* n) {
  return n->D == kD && n->H == kH && n->Ka == kD && n->Kb == kD;
}

int launch_fused32(const FusedArgs& a, hipStream_t stream) {
  static DeviceOnce once;
  const size_t lds = sizeof(float) * kLds32;
  if (once.pending()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gauge_traj_fused32_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      set_error("fused 32-row kernel: cannot reserve %zu B of LDS", lds);
      return L2HMC_ERR_HIP;
    }
    once.done();
  }
  const dim3 grid((unsigned)ceil_div(a.rows, kR32));
  prof_before(kProfFused, stream);
  hipLaunchKernelGGL(gauge_traj_fused32_kernel, grid, dim3(kT32), lds, stream, a);
  prof_after(kProfFused, stream);
  L2HMC_CHECK_LAUNCH("gauge_traj_fused32");
  return L2HMC_OK;
}

}  // namespace l2hmc
