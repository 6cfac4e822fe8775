// Whole-trajectory persistent kernel for the lattice integrator (gfx950).
//
// One launch integrates ALL leapfrog steps of
//   l2hmc/dynamics/gauge_dynamics.py:261-313 (transition_kernel) with
//   :412-483 (_forward_lf/_backward_lf), :486-590 (sub-updates), :592-609 (accept
//   prob), network/generic_net.py:129-146 (S/T/Q nets) and the U(1) force
//   (:698-709 over lattice/lattice.py:337-362)
// for a tile of 16 chain-rows per workgroup.  Chains are independent and the
// plaquette stencil is local to a chain, so a workgroup never talks to another
// one: x, v, force, both hidden activations and the log-det sums live in LDS
// for the whole trajectory (read from / written to HBM exactly once), and the
// only steady-state traffic is the weight stream.
//
// Why 16 rows: rows / 256 CUs = 16 at the benchmark shape (2 directions x 2048
// chains), which is exactly one 16x16x4 fp32 MFMA tile in M.  Every CU then
// streams every weight once per network call: (2DH + H^2 + 3DH) * 4 B = 2.36 MB
// per 18.9 MFLOP of MFMA work, i.e. ~77 GB/s per CU at the fp32 MFMA peak --
// the L2 -> CU fabric roofline and the MFMA roofline coincide.
// Weights are therefore pre-packed (l2hmc_dense_pack) into the exact order a
// wave consumes them -- [wave][k-chunk][n-tile][lane][4] -- so each B-fragment
// load is one fully coalesced 1 KiB global_load_dwordx4 straight into VGPRs (no
// LDS round trip: a wave's columns are not shared with other waves), double
// buffered one k-chunk (16 k) ahead.  The A operand (16 rows of activations) is
// shared by the 4 waves and read from LDS with conflict-free ds_read_b128
// (row stride = K + 8 floats).
#include "fused_common.h"
#include "fused_args.h"
#include <atomic>
#include <stdlib.h>

namespace l2hmc {

// D = x_dim, H = hidden width, KA = width of each first-layer input (x_dim for GenericNet; the flattened conv
// features for ConvNet3D), CONV = the two inputs go through the conv front-end first (8x8 lattice, F = 8).
template <int D, int H, int KA = D, bool CONV = false>
struct FusedCfg {
  // Waves per workgroup: TWO per SIMD in every instance.  ConvNet3D plans since round 2 (their VALU conv stage is
  // latency-bound at one: 0.987 -> 0.883 ms per step).  The GenericNet kernel used to be faster with one wave per SIMD
  // (rounds 1-3: 1.655 against 1.745 ms); with the weight stream of round 4 (buffer loads, pinned interleave) two win --
  // one wave's epilogues and barriers lie under the other's matrix instructions: 1.431 -> 1.38 ms per step.
  // IMGW: waves the packed weight image is laid out for (pack_fused_kernel).  The 8-wave GenericNet instances read the
  // SAME 4-wave image as the 32-row form and the reverse kernel: two waves share a section (fused_common.h: load_frags),
  // and the taped forward writes its relu-gate words in the 4-wave lane order the reverse kernel reads (fused_train.hip).
  static constexpr int IMGW = CONV ? 2 * kFWaves : kFWaves;
  static constexpr int WAVES = 2 * kFWaves;
  static constexpr int RW = WAVES / IMGW;      // waves per image section
  static constexpr int THREADS = 64 * WAVES;   // wave w owns output columns [w*N/WAVES, (w+1)*N/WAVES)
  // threads per chain in the chain-local passes (force, kinetic energy, observables): their sums are part of the
  // result's bits, so the GenericNet instances keep 16 whatever their wave count (the other threads idle there)
  static constexpr int TPC = CONV ? THREADS / kFM : 16;
  static constexpr int SX = D + 8;             // LDS row stride of x / v / second-input rows
  static constexpr int SA = KA + 8;            // LDS row stride of the conv feature rows
  static constexpr int SH = H + 8;             // LDS row stride of h1 / h2
  static constexpr int NT1 = H / (16 * WAVES);     // 16-column tiles per wave, layers 1 and 2
  static constexpr int NTH = D / (16 * WAVES);     // tiles per wave per head
  static constexpr int NTI1 = H / (16 * IMGW);     // ... and per image section (= NT1, NTH unless two waves share it)
  static constexpr int NTIH = D / (16 * IMGW);
  static_assert(NT1 >= 1 && NT1 <= 8 && NTH >= 1, "every wave needs at least one tile per layer");
  static_assert(RW == 1 || NTH == 1, "a shared heads section is walked with one tile per head and wave");
  static constexpr int KC1 = 2 * KA / 16;      // k-chunks (16 k each), layer 1
  static constexpr int KC2 = H / 16;           // k-chunks, layers 2 and heads
  static constexpr size_t P1 = (size_t)2 * KA * H;  // packed floats per section
  static constexpr size_t P2 = (size_t)H * H;
  static constexpr size_t PH = (size_t)3 * D * H;
  // per-net constants kept in LDS: b1[H] wt[2H] bh[H] bhd[3D] exp(cs)[D] exp(cq)[D]
  static constexpr int NC = 4 * H + 5 * D;
  // conv front-end (CONV only): F = 8 filters on the 8x8 lattice
  static constexpr int CF = 8, CL = 8;
  static constexpr int CW = 18 * CF + CF + 8 * CF * CF + 2 * CF;            // one (net, input) filter set
  static constexpr int CXIN = (CL + 2) * (CL + 2) * 2;                      // haloed chain
  static constexpr int CP1 = (CL / 2 + 1) * (CL / 2 + 1) * CF;              // pooled conv1 map, zero halo
  static constexpr int CONV_FLOATS = CONV ? 2 * kFM * SA + 4 * CW + kFM * (CXIN + CP1) : 0;
  static constexpr int LDS_FLOATS = 3 * kFM * SX + 2 * kFM * SH + 2 * NC + kFM * (D / 2 + 4) /*sinP*/ +
                                    2 * D /*masks*/ + WAVES * kFM /*ldw*/ + (RW > 1 ? IMGW * 64 : 0) /*ldx*/ + kFM /*dir*/ +
                                    8 * kFM /*step mode*/ + CONV_FLOATS;
};

// ---------------------------------------------------------------------------
// weight packing (device side, once per weight update)
// ---------------------------------------------------------------------------
__global__ void pack_fused_kernel(l2hmc_dense_net n, float* __restrict__ out, int waves) {
  const int kFWaves = waves;                 // waves per workgroup of the kernel that will read this image
  const int D = n.D, H = n.H, K1 = n.Ka + n.Kb;
  const size_t P1 = (size_t)K1 * H, P2 = (size_t)H * H, PH = (size_t)3 * D * H;
  const int NT1 = H / (16 * kFWaves), NTH = D / (16 * kFWaves);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < P1 + P2 + PH;
       i += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(i & 3), lane = (int)((i >> 2) & 63);
    size_t rest;
    float val;
    if (i < P1 + P2) {
      const bool first = i < P1;
      const int K = first ? K1 : H;
      rest = (first ? i : i - P1) >> 8;               // ((w * KC + kc) * NT1 + t)
      const int t = (int)(rest % NT1);
      rest /= NT1;
      const int KC = K / 16;
      const int kc = (int)(rest % KC), w = (int)(rest / KC);
      const int col = (w * NT1 + t) * 16 + (lane & 15);
      const int k = kc * 16 + (lane >> 4) * 4 + j;
      val = first ? n.w1_t[(size_t)col * K1 + k] : n.wh_t[(size_t)col * H + k];
    } else {
      rest = (i - P1 - P2) >> 8;                      // (((w * KC2 + kc) * 3 + hd) * NTH + t)
      const int t = (int)(rest % NTH);
      rest /= NTH;
      const int hd = (int)(rest % 3);
      rest /= 3;
      const int KC = H / 16;
      const int kc = (int)(rest % KC), w = (int)(rest / KC);
      const int col = w * (D / kFWaves) + t * 16 + (lane & 15);
      const int k = kc * 16 + (lane >> 4) * 4 + j;
      val = n.whd_t[((size_t)hd * D + col) * H + k];
    }
    out[i] = val;
  }
}

#ifdef L2HMC_STAMPS
#define FT_NOW()                                                                              \
  ({                                                                                          \
    unsigned long long t_;                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    t_;                                                                                       \
  })
#define FT_ADD(slot, t0) ft[slot] += FT_NOW() - (t0)
#else
#define FT_NOW() 0ull
#define FT_ADD(slot, t0) do {} while (0)
#endif

// (struct FusedArgs: fused_args.h, shared with the sub-tile form in fused_traj4.hip)

#ifdef L2HMC_STAMPS
int g_fused_stagger = 0;
extern "C" void l2hmc_debug_set_stagger(int cycles) { g_fused_stagger = cycles; }
#endif

// TAPE: training instantiation (GenericNet plans) that also writes the per-call tape of train.hip
template <int D, int H, int KA, bool CONV, bool TAPE = false>
__global__ __launch_bounds__((FusedCfg<D, H, KA, CONV>::THREADS)) void gauge_traj_fused_kernel(FusedArgs p) {
  using Cfg = FusedCfg<D, H, KA, CONV>;
  constexpr int kFWaves = Cfg::WAVES, kFThreads = Cfg::THREADS, kTPC = Cfg::TPC;   // this instance's geometry
  constexpr int SX = Cfg::SX, SH = Cfg::SH, SA = Cfg::SA, NT1 = Cfg::NT1, NTH = Cfg::NTH;
  constexpr int RW = Cfg::RW, IMGW = Cfg::IMGW, NTI1 = Cfg::NTI1, NTIH = Cfg::NTIH;
  constexpr int TSH = RW > 1 ? NTIH : 1;       // tile stride of a wave's heads fragments in a shared section
  constexpr int sites = D / 2;
  constexpr int SP = sites + 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xs = lds;                         // [16][SX] position
  float* vs = xs + kFM * SX;               // [16][SX] momentum
  float* gs = vs + kFM * SX;               // [16][SX] second net input: force, or keep (.) x
  float* h1 = gs + kFM * SX;               // [16][SH]
  float* h2 = h1 + kFM * SH;               // [16][SH]
  float* cx = h2 + kFM * SH;               // XNet constants [NC]
  float* cv = cx + Cfg::NC;                // VNet constants [NC]
  float* sp = cv + Cfg::NC;                // [16][SP] sin P
  float* skm = sp + kFM * SP;              // [2][D]  masks of this step: forward row, backward row
  float* ldw = skm + 2 * D;                // [IMGW][16] log-det partial sums per image wave; behind them (RW > 1 only)
                                           // [IMGW][64] the lane sums an even wave hands to its odd partner
  float* ldx = ldw + IMGW * kFM;
  int* sdir = reinterpret_cast<int*>(ldw + kFWaves * kFM + (RW > 1 ? IMGW * 64 : 0));   // [16]
  float* stp = reinterpret_cast<float*>(sdir + kFM);  // step mode: coin[16] u[16] p_row[16] obs[16][4]
  // ConvNet3D front-end state (CONV only; zero-sized otherwise)
  float* fa = stp + 8 * kFM;                          // [16][SA] features of the first input
  float* fb = fa + kFM * SA;                          // [16][SA] features of the second input
  float* cwl = fb + kFM * SA;                         // [net x|v][input a|b][CW] filters
  float* cxin = cwl + 4 * Cfg::CW;                    // [16][CXIN] haloed chains
  float* cp1 = cxin + kFM * Cfg::CXIN;                // [16][CP1]  pooled conv1 maps

  // diagnostic cycle shares: 0-2 gemm L1/L2/heads, 3-5 their epilogues, 6 barriers, 7 force, 8 mask pass, 9 total
  [[maybe_unused]] unsigned long long ft[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  [[maybe_unused]] const unsigned long long ft_start = FT_NOW();
#ifdef L2HMC_STAMPS
  unsigned long long rt0;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0)::"memory");
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, r = lane & 15;
  const int64_t row0 = (int64_t)blockIdx.x * kFM;
  const int nrow = (int)min((int64_t)kFM, p.rows - row0);
  const float eps = p.eps;

#ifdef L2HMC_STAMPS
  if (p.stagger > 0) {
    // (diagnostic build) spread the workgroups of an XCD (blockIdx & 7 labels the XCD group) over the weight stream
    const long long delay = (long long)((blockIdx.x >> 3) & 31) * p.stagger;
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    while ((long long)__builtin_amdgcn_s_memtime() - t0 < delay) __builtin_amdgcn_s_sleep(16);
  }
#endif
  // ---- stage chain state and constants ------------------------------------
  const bool STEPM = p.step_B > 0;
  const int cpw = STEPM ? (p.step_both ? kFM / 2 : kFM) : kFM;           // chains per workgroup in step mode
  float* scoin = stp;                    // [16] direction coin per chain slot
  float* su = stp + kFM;                 // [16] MH uniform
  float* spx = stp + 2 * kFM;            // [16] accept probability per row
  float* sobs = stp + 3 * kFM;           // [16][4] sum cos P (in), sum project P (in), sum project P (out), p
  auto philox_u01 = [&](uint64_t elem, uint64_t stream) {        // element `elem` of l2hmc_fill_uniform's stream
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
    for (int i = tid; i < kFM * (D / 4); i += kFThreads) {
      const int rr = i / (D / 4), c4 = (i - rr * (D / 4)) * 4;
      const int k = p.step_both ? (rr & (kFM / 2 - 1)) : rr;
      const int64_t chain = (int64_t)blockIdx.x * cpw + k;
      const int dsel = p.step_both ? (rr >= kFM / 2 ? 1 : 0) : (scoin[k] > 0.5f ? 0 : 1);   // gauge_dynamics.py:221-227
      f32x4 xv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
      if (chain < p.step_Bl) {
        xv = *reinterpret_cast<const f32x4*>(p.x0 + chain * D + c4);
        // momentum of (direction dsel, chain): elements [(dsel * B + chain) * D, + D) of the normal stream
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
    for (int i = tid; i < kFM * (D / 4); i += kFThreads) {
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
    for (int i = tid; i < H; i += kFThreads) {
      c[i] = n.b1[i];
      c[H + i] = n.wt[i];
      c[2 * H + i] = n.wt[H + i];
      c[3 * H + i] = n.bh[i];
    }
    for (int i = tid; i < 3 * D; i += kFThreads) c[4 * H + i] = n.bhd[i];
    for (int i = tid; i < D; i += kFThreads) {
      c[4 * H + 3 * D + i] = expf(n.coeff_s[i]);
      c[4 * H + 4 * D + i] = expf(n.coeff_q[i]);
    }
  };
  load_consts(p.xnet, cx);
  load_consts(p.vnet, cv);
  if constexpr (CONV) {
    constexpr int F = Cfg::CF, F2 = 2 * Cfg::CF;
    auto load_filters = [&](const float* w1, const float* b1, const float* w2, const float* b2, float* dst) {
      for (int i = tid; i < 18 * F; i += kFThreads) dst[i] = w1[i];
      for (int i = tid; i < F; i += kFThreads) dst[18 * F + i] = b1[i];
      for (int i = tid; i < 4 * F * F2; i += kFThreads) {      // Keras [di][dj][dd][c][g]: keep dd = 0
        const int g = i % F2, c = (i / F2) % F, tap = i / (F2 * F);
        dst[19 * F + i] = w2[((size_t)(tap * 2) * F + c) * F2 + g];
      }
      for (int i = tid; i < F2; i += kFThreads) dst[19 * F + 4 * F * F2 + i] = b2[i];
    };
    load_filters(p.xfront.w1_a, p.xfront.b1_a, p.xfront.w2_a, p.xfront.b2_a, cwl + 0 * Cfg::CW);
    load_filters(p.xfront.w1_b, p.xfront.b1_b, p.xfront.w2_b, p.xfront.b2_b, cwl + 1 * Cfg::CW);
    load_filters(p.vfront.w1_a, p.vfront.b1_a, p.vfront.w2_a, p.vfront.b2_a, cwl + 2 * Cfg::CW);
    load_filters(p.vfront.w1_b, p.vfront.b1_b, p.vfront.w2_b, p.vfront.b2_b, cwl + 3 * Cfg::CW);
    for (int i = tid; i < kFM * (Cfg::CXIN + Cfg::CP1); i += kFThreads) cxin[i] = 0.f;   // halos stay zero
  }
  if (tid < kFM) {
    int d = 0;
    if (STEPM) d = p.step_both ? (tid >= kFM / 2 ? 1 : 0) : (scoin[tid] > 0.5f ? 0 : 1);
    else if (tid < nrow) d = p.dir ? p.dir[row0 + tid] : (p.dir_split > 0 && row0 + tid >= p.dir_split) ? 1 : 0;
    sdir[tid] = d;
  }
  if (tid < IMGW * kFM) ldw[tid] = 0.f;
  __syncthreads();

  const int dirl = sdir[r];           // direction of the row this lane owns in a C fragment (fused_common.h)

  // ---- chain-local passes: kTPC consecutive threads per chain ------------------
  // (GenericNet instances: kTPC = 16 whatever the wave count; the threads beyond 16 chains x 16 walk empty loops and
  //  take part in the barriers only)
  const bool own = tid < kFM * kTPC;
  const int fc = own ? tid / kTPC : 0, fl = tid % kTPC;      // chain, lane-in-chain
  const int sites_l = own ? sites : 0, D_l = own ? D : 0;    // loop bounds of the chain-local passes
  auto chain_sum = [&](float v) {
#pragma unroll
    for (int off = kTPC / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
  };
  const int T = p.T, X = p.X;
  const int xsh = 31 - __clz(X);       // sites = 64 and X divides it: shifts instead of run-time divisions
  // force (beta * dS/dx) into gs; returns this chain's action (all 16 lanes of the chain)
  auto force_pass = [&]() -> float {
    const float* xc = xs + fc * SX;
    float act = 0.f;
    for (int s = fl; s < sites_l; s += kTPC) {
      const int i = s >> xsh, j = s & (X - 1);            // X is a power of two (T * X = 64)
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
    for (int s = fl; s < sites_l; s += kTPC) {
      const int i = s >> xsh, j = s & (X - 1);            // X is a power of two (T * X = 64)
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
    for (int d = fl; d < D_l; d += kTPC) k += vc[d] * vc[d];
    return 0.5f * chain_sum(k);
  };

  const float act0 = force_pass();     // also leaves the force of x0 in gs
  const float kin0 = kinetic_pass();

  // ---- one network evaluation + fused sub-update ------------------------------
  // in1: first input rows (LDS, stride SX); second input is always gs.
  // mode 1: momentum update (uses gs as the force), mode 2: position update with keep masks.
  // ConvNet3D front-end on a [16][SX] LDS array (network/conv_net.py:251-262; same arithmetic as
  // conv3d_front_kernel): conv1(3,3,2)+relu+pool -> cp1, conv2(2,2,[2])+relu+pool -> dst [16][SA].
  [[maybe_unused]] auto conv_features = [&](const float* src, const float* cw, float* dst) {
    constexpr int F = Cfg::CF, F2 = 2 * Cfg::CF, L = Cfg::CL, LP = L + 2, L2 = L / 2, L2P = L2 + 1, L4 = L / 4;
    const float* w1 = cw;
    const float* b1 = cw + 18 * F;
    const float* w2 = b1 + F;
    const float* b2 = w2 + 4 * F * F2;
    for (int i = tid; i < kFM * D; i += kFThreads) {
      const int c = i / D, e = i - c * D;
      const int site = e >> 1, mu = e & 1, ii = site / L, jj = site - ii * L;
      cxin[c * Cfg::CXIN + ((ii + 1) * LP + jj + 1) * 2 + mu] = src[c * SX + e];
    }
    __syncthreads();
    // conv1 + relu + pool.  A thread owns a PAIR of filters (2 fp, 2 fp + 1) of one pooled position: the pair's 18
    // taps sit in registers for all of its positions (the workgroup size is a multiple of F / 2, so fp never
    // changes), the 4 x 4 x 2 input patch under the 2 x 2 pooling window is read once (16 ds_read_b64 instead of
    // 72 + 72 scalar reads), and every multiply-add is a v_pk_fma_f32 over the filter pair.
    {
      using f32x2 = __attribute__((ext_vector_type(2))) float;
      constexpr int FP = F / 2;
      static_assert(kFThreads % FP == 0 && F % 2 == 0, "filter pairs must stay with their threads");
      const int fp = tid % FP;
      f32x2 k0[9], k1[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        k0[t] = *reinterpret_cast<const f32x2*>(w1 + (t * 2 + 0) * F + 2 * fp);
        k1[t] = *reinterpret_cast<const f32x2*>(w1 + (t * 2 + 1) * F + 2 * fp);
      }
      const f32x2 bias = *reinterpret_cast<const f32x2*>(b1 + 2 * fp);
      for (int idx = tid; idx < kFM * L2 * L2 * FP; idx += kFThreads) {
        int rr = idx / FP;
        const int J = rr % L2;
        rr /= L2;
        const int I = rr % L2, c = rr / L2;
        f32x2 px[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int bb = 0; bb < 4; ++bb)
            px[a][bb] = *reinterpret_cast<const f32x2*>(cxin + c * Cfg::CXIN + ((2 * I + a) * LP + 2 * J + bb) * 2);
        f32x2 m = {-INFINITY, -INFINITY};
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
          for (int bb = 0; bb < 2; ++bb) {
            f32x2 v0 = bias, v1 = bias;              // the two depth positions of conv_net.py's (3, 3, 2) filter
#pragma unroll
            for (int di = 0; di < 3; ++di) {
#pragma unroll
              for (int dj = 0; dj < 3; ++dj) {
                const f32x2 xx = px[a + di][bb + dj];
                const f32x2 x0 = {xx[0], xx[0]}, x1 = {xx[1], xx[1]};
                v0 += x0 * k0[di * 3 + dj];      // (conv3d_front.hip's order: three fused multiply-adds per tap)
                v0 += x1 * k1[di * 3 + dj];
                v1 += x1 * k0[di * 3 + dj];
              }
            }
            m[0] = fmaxf(m[0], fmaxf(v0[0], v1[0]));
            m[1] = fmaxf(m[1], fmaxf(v0[1], v1[1]));
          }
        }
        *reinterpret_cast<f32x2*>(cp1 + c * Cfg::CP1 + (I * L2P + J) * F + 2 * fp) = f32x2{fmaxf(m[0], 0.f), fmaxf(m[1], 0.f)};
      }
    }
    __syncthreads();
    // conv2 + relu + pool, again a pair of filters (2 gp, 2 gp + 1) per thread: the 2 x 2 outputs of the pooling
    // window share a 3 x 3 patch of the pooled conv1 map -- per 4 input channels, 9 ds_read_b128 (patch) + 16
    // ds_read_b64 (taps of the pair) feed 64 v_pk_fma_f32
    {
      using f32x2 = __attribute__((ext_vector_type(2))) float;
      constexpr int GP = F2 / 2;
      for (int idx = tid; idx < kFM * L4 * L4 * GP; idx += kFThreads) {
        const int gp = idx % GP;
        int rr = idx / GP;
        const int J2 = rr % L4;
        rr /= L4;
        const int I2 = rr % L4, c = rr / L4;
        const f32x2 bias = *reinterpret_cast<const f32x2*>(b2 + 2 * gp);
        f32x2 acc[2][2] = {{bias, bias}, {bias, bias}};
        const float* pbase = cp1 + c * Cfg::CP1 + ((2 * I2) * L2P + 2 * J2) * F;
        for (int ch4 = 0; ch4 < F; ch4 += 4) {
          f32x4 w[3][3];
#pragma unroll
          for (int wi = 0; wi < 3; ++wi)
#pragma unroll
            for (int wj = 0; wj < 3; ++wj)
              w[wi][wj] = *reinterpret_cast<const f32x4*>(pbase + (wi * L2P + wj) * F + ch4);
#pragma unroll
          for (int di = 0; di < 2; ++di)
#pragma unroll
            for (int dj = 0; dj < 2; ++dj)
#pragma unroll
              for (int cc = 0; cc < 4; ++cc) {
                const f32x2 kw = *reinterpret_cast<const f32x2*>(w2 + ((di * 2 + dj) * F + ch4 + cc) * F2 + 2 * gp);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                  for (int bb = 0; bb < 2; ++bb) {
                    const float xv = w[a + di][bb + dj][cc];
                    acc[a][bb] += f32x2{xv, xv} * kw;
                  }
              }
        }
        const float m0 = fmaxf(fmaxf(acc[0][0][0], acc[0][1][0]), fmaxf(acc[1][0][0], acc[1][1][0]));
        const float m1 = fmaxf(fmaxf(acc[0][0][1], acc[0][1][1]), fmaxf(acc[1][0][1], acc[1][1][1]));
        *reinterpret_cast<f32x2*>(dst + c * SA + (I2 * L4 + J2) * F2 + 2 * gp) = f32x2{fmaxf(m0, 0.f), fmaxf(m1, 0.f)};
      }
    }
    __syncthreads();
  };

  // First-layer pre-activations that recur unchanged and are kept in registers instead of being recomputed
  // (bit-identical results, 8.3 % fewer weight bytes and MFMAs per leapfrog step):
  //   keep_v: VNet's whole first-layer product.  The second half-kick of step s and the first half-kick of
  //           step s+1 see the same (x, force); only the time term differs, and that is added in the epilogue.
  //   keep_x: XNet's product with its FIRST input (v), identical for the two position sub-updates of a step.
  f32x4 keep_v[NT1], keep_x[NT1];
  bool keep_v_valid = false;

  // l1: 0 = compute both halves; 1 = as 0 and store the raw product in keep_v; 2 = take keep_v, no GEMM;
  //     3 = compute, snapshot the first-input half into keep_x; 4 = start from keep_x, second half only.
  auto net_update = [&](const l2hmc_dense_net& net, const float* cn, const float* in1, int mode, int sub,
                        bool prep_next_mask, int l1, bool is_vnet, const float tcr, const float tsr,
                        int callidx) {
    const float* pk = net.packed;
    // layers 2 and 3 of a network are streamed in alternating directions on its consecutive calls (fused_common.h)
    const bool zig = (callidx & 1) != 0;
    // training tape (generic plans): this call's inputs and the state its sub-update consumes
    [[maybe_unused]] const FusedTape& tp = is_vnet ? p.tv : p.tx;
    [[maybe_unused]] const size_t tcr0 = (size_t)callidx * (size_t)p.rows + (size_t)row0;     // first taped row of this workgroup
    if constexpr (TAPE) {
      {
        const float* stsrc = mode == 1 ? vs : xs;
        for (int i = tid; i < kFM * (D / 4); i += kFThreads) {
          const int rr = i / (D / 4), c4 = (i - rr * (D / 4)) * 4;
          if (rr < nrow) {
            float* dst = tp.in + (tcr0 + rr) * (2 * D);
            tape_store(dst + c4, *reinterpret_cast<const f32x4*>(in1 + rr * SX + c4));
            tape_store(dst + D + c4, *reinterpret_cast<const f32x4*>(gs + rr * SX + c4));
            tape_store(tp.st + (tcr0 + rr) * D + c4, *reinterpret_cast<const f32x4*>(stsrc + rr * SX + c4));
          }
        }
      }
    }
    [[maybe_unused]] auto tape_rows = [&](float* dst, const float* src) {     // [16][H] LDS rows -> tape
      for (int i = tid; i < kFM * (H / 4); i += kFThreads) {
        const int rr = i / (H / 4), c4 = (i - rr * (H / 4)) * 4;
        if (rr < nrow)
          tape_store(dst + (tcr0 + rr) * H + c4, *reinterpret_cast<const f32x4*>(src + rr * SH + c4));
      }
    };
    const int wv = __builtin_amdgcn_readfirstlane(wave);      // provably uniform: the weight loads' base stays in SGPRs
    const int wimg = wv / RW, wsub = wv - wimg * RW;          // image section, and this wave's share of it
    const float* wp1 = pk + (size_t)wimg * Cfg::KC1 * NTI1 * 256;
    const float* wp2 = pk + Cfg::P1 + (size_t)wimg * Cfg::KC2 * NTI1 * 256;
    const float* wph = pk + Cfg::P1 + Cfg::P2 + (size_t)wimg * Cfg::KC2 * 3 * NTIH * 256;
    const int to1 = wsub * NT1, toh = wsub * NTH;             // first tile of this wave in a chunk of the section
    [[maybe_unused]] float ld_k[4] = {0.f, 0.f, 0.f, 0.f}, ld_s[4] = {0.f, 0.f, 0.f, 0.f};   // (RW > 1: the odd wave's log-det terms)
#ifndef L2HMC_DP1                                 // (A/B builds: tools/build_variant.sh)
#define L2HMC_DP1 3
#define L2HMC_DP2 3
#define L2HMC_DPH 4
#endif
    constexpr int DP1 = L2HMC_DP1, DP2 = L2HMC_DP2, DPH = L2HMC_DPH;   // ring depths (fused_common.h): first-layer halves, layer 2, heads
    BRing<NT1, DP2> R2;
    BRing<3 * NTH, DPH> R3;
    // ----- layer 1: two half-K streams (first input rows, then the second-input rows in gs)
    {
      constexpr int KH = Cfg::KC1 / 2;
      f32x4 acc[NT1];
      [[maybe_unused]] unsigned long long t0 = FT_NOW();
      // inputs of the dense trunk: the LDS rows themselves, or their conv features
      const float* src1 = in1;
      const float* src2 = gs;
      int s1 = SX;
      if constexpr (CONV) {
        const float* cwn = cwl + (is_vnet ? 2 : 0) * Cfg::CW;
        [[maybe_unused]] const unsigned long long tc0 = FT_NOW();
        if (l1 == 0 || l1 == 1 || l1 == 3) conv_features(in1, cwn, fa);            // first input (conv_v*)
        if (l1 != 2) conv_features(gs, cwn + Cfg::CW, fb);                          // second input (conv_x*)
        FT_ADD(8, tc0);                                                             // (slot 8: conv front-end; also inside slot 0)
        src1 = fa;
        src2 = fb;
        s1 = SA;
        if constexpr (TAPE) {      // features (also the reused ones still sitting in fa / fb) -> tape, for d loss / d W1
          for (int i = tid; i < kFM * (2 * KA / 4); i += kFThreads) {
            const int rr = i / (2 * KA / 4), c4 = (i - rr * (2 * KA / 4)) * 4;
            if (rr < nrow)
              tape_store(tp.feat + (tcr0 + rr) * (2 * KA) + c4,
                         *reinterpret_cast<const f32x4*>((c4 < KA ? fa + rr * SA + c4 : fb + rr * SA + (c4 - KA))));
          }
        }
      }
      if (l1 == 2) {
#pragma unroll
        for (int t = 0; t < NT1; ++t) acc[t] = keep_v[t];
      } else {
        BRing<NT1, DP1> RA, RB;
        const float* wpb = wp1 + (size_t)KH * NTI1 * 256;
        ring_prime<NT1, DP1, NTI1>(RB, wpb, false, 0, to1);          // both halves' first fragments are requested up front
        if (l1 == 4) {
#pragma unroll
          for (int t = 0; t < NT1; ++t) acc[t] = keep_x[t];
        } else {
          ring_prime<NT1, DP1, NTI1>(RA, wp1, false, 0, to1);
#pragma unroll
          for (int t = 0; t < NT1; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
          const float* a1 = src1 + r * s1 + q * 4;
          stream_layer<NT1, KH, DP1, NTI1>(
              RA, wp1, [&](int kc) { return *reinterpret_cast<const f32x4*>(a1 + kc * 16); }, acc, false, to1);
          if (l1 == 3) {
#pragma unroll
            for (int t = 0; t < NT1; ++t) keep_x[t] = acc[t];
          }
        }
        const float* a2 = src2 + r * s1 + q * 4;
        stream_layer<NT1, KH, DP1, NTI1>(
            RB, wpb, [&](int kc) { return *reinterpret_cast<const f32x4*>(a2 + kc * 16); }, acc, false, to1);
        if (l1 == 1) {
#pragma unroll
          for (int t = 0; t < NT1; ++t) keep_v[t] = acc[t];
        }
      }
      ring_prime<NT1, DP2, NTI1>(R2, wp2, zig, Cfg::KC2, to1);      // layer-2 weights start flowing under the epilogue + barrier
      FT_ADD(0, t0);
      t0 = FT_NOW();
      [[maybe_unused]] unsigned gmask = 0;
#pragma unroll
      for (int t = 0; t < NT1; ++t) {
        const int c0 = (wave * NT1 + t) * 16 + q * 4;          // this lane: row r, columns c0 .. c0 + 3
        const f32x4 b = *reinterpret_cast<const f32x4*>(cn + c0);
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(cn + H + c0);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(cn + 2 * H + c0);
        f32x4 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          hv[e] = fmaxf(acc[t][e] + b[e] + (tcr * w0[e] + tsr * w1[e]), 0.f);
          if constexpr (TAPE) gmask |= (hv[e] > 0.f ? 1u : 0u) << (t * 4 + e);
        }
        *reinterpret_cast<f32x4*>(h1 + r * SH + c0) = hv;
      }
      if constexpr (TAPE) {
        // one 32-bit word per lane of the 4-wave image wave (the reverse kernel's layout); two waves of a section write its halves
        const size_t word = ((size_t)(callidx * 2 + 0) * gridDim.x + blockIdx.x) * (64 * IMGW) + wimg * 64 + lane;
        if constexpr (RW == 1) tp.gate[word] = gmask;
        else reinterpret_cast<unsigned short*>(tp.gate)[word * 2 + wsub] = (unsigned short)gmask;
      }
      FT_ADD(3, t0);
    }
    {
      [[maybe_unused]] const unsigned long long tb = FT_NOW();
      __syncthreads();
      FT_ADD(6, tb);
    }
    if constexpr (TAPE) tape_rows(tp.h1, h1);
    // ----- layer 2
    {
      f32x4 acc[NT1];
#pragma unroll
      for (int t = 0; t < NT1; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* a = h1 + r * SH + q * 4;
      [[maybe_unused]] unsigned long long t0 = FT_NOW();
      stream_layer<NT1, Cfg::KC2, DP2, NTI1>(
          R2, wp2, [&](int kc) { return *reinterpret_cast<const f32x4*>(a + kc * 16); }, acc, zig, to1);
      ring_prime<3 * NTH, DPH, 3 * NTIH, TSH>(R3, wph, zig, Cfg::KC2, toh);
      FT_ADD(1, t0);
      t0 = FT_NOW();
      [[maybe_unused]] unsigned gmask = 0;
#pragma unroll
      for (int t = 0; t < NT1; ++t) {
        const int c0 = (wave * NT1 + t) * 16 + q * 4;
        const f32x4 b = *reinterpret_cast<const f32x4*>(cn + 3 * H + c0);
        f32x4 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          hv[e] = fmaxf(acc[t][e] + b[e], 0.f);
          if constexpr (TAPE) gmask |= (hv[e] > 0.f ? 1u : 0u) << (t * 4 + e);
        }
        *reinterpret_cast<f32x4*>(h2 + r * SH + c0) = hv;
      }
      if constexpr (TAPE) {
        // one 32-bit word per lane of the 4-wave image wave (the reverse kernel's layout); two waves of a section write its halves
        const size_t word = ((size_t)(callidx * 2 + 1) * gridDim.x + blockIdx.x) * (64 * IMGW) + wimg * 64 + lane;
        if constexpr (RW == 1) tp.gate[word] = gmask;
        else reinterpret_cast<unsigned short*>(tp.gate)[word * 2 + wsub] = (unsigned short)gmask;
      }
      FT_ADD(4, t0);
    }
    {
      [[maybe_unused]] const unsigned long long tb = FT_NOW();
      __syncthreads();
      FT_ADD(6, tb);
    }
    if constexpr (TAPE) tape_rows(tp.h2, h2);
    // ----- heads + update
    {
      f32x4 acc[3 * NTH];
#pragma unroll
      for (int t = 0; t < 3 * NTH; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* a = h2 + r * SH + q * 4;
      [[maybe_unused]] unsigned long long t0 = FT_NOW();
      stream_layer<3 * NTH, Cfg::KC2, DPH, 3 * NTIH, TSH>(
          R3, wph, [&](int kc) { return *reinterpret_cast<const f32x4*>(a + kc * 16); }, acc, zig, toh);
      FT_ADD(2, t0);
      t0 = FT_NOW();
      float ld = 0.f;                       // this lane's share of row r's log-det
      const float* bhd = cn + 4 * H;
      const float* es = bhd + 3 * D;
      const float* eq = es + D;
      const int d = dirl;
#pragma unroll
      for (int t = 0; t < NTH; ++t) {
        const int c0 = wave * (D / kFWaves) + t * 16 + q * 4;      // row r, columns c0 .. c0 + 3
        const f32x4 b_s = *reinterpret_cast<const f32x4*>(bhd + c0);
        const f32x4 b_t = *reinterpret_cast<const f32x4*>(bhd + D + c0);
        const f32x4 b_q = *reinterpret_cast<const f32x4*>(bhd + 2 * D + c0);
        const f32x4 e_s = *reinterpret_cast<const f32x4*>(es + c0);
        const f32x4 e_q = *reinterpret_cast<const f32x4*>(eq + c0);
        const f32x4 mf = *reinterpret_cast<const f32x4*>(skm + c0);
        const f32x4 mb = *reinterpret_cast<const f32x4*>(skm + D + c0);
        const int idx = r * SX + c0;
        f32x4 S, Tt, Q;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          S[e] = fast_tanh(acc[0 * NTH + t][e] + b_s[e]) * e_s[e];
          Tt[e] = acc[1 * NTH + t][e] + b_t[e];
          const float qq = acc[2 * NTH + t][e] + b_q[e];
          Q[e] = (net.q_tanh ? fast_tanh(qq) : qq) * e_q[e];
        }
        if constexpr (TAPE) {
          if (r < nrow) {
            const size_t plane = (size_t)p.rows * D;
            float* o = tp.stq + (size_t)callidx * 3 * plane + ((size_t)row0 + r) * D + c0;
            tape_store(o, S);
            tape_store(o + plane, Tt);
            tape_store(o + 2 * plane, Q);
          }
        }
        if (mode == 1) {
          // gauge_dynamics.py:497-506 (fwd), :549-559 (bwd)
          const f32x4 g = *reinterpret_cast<const f32x4*>(gs + idx);
          const f32x4 v = *reinterpret_cast<const f32x4*>(vs + idx);
          f32x4 vn;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float s = (d ? -0.5f : 0.5f) * eps * S[e];
            const float kick = 0.5f * eps * (fast_exp(eps * Q[e]) * g[e] - Tt[e]);
            const float es_ = fast_exp(s);
            vn[e] = d ? es_ * (v[e] + kick) : v[e] * es_ - kick;
            ld += s;
            if constexpr (RW > 1) { ld_k[e] = 1.f; ld_s[e] = s; }
          }
          *reinterpret_cast<f32x4*>(vs + idx) = vn;
          // the next net call is the first position sub-update: its second input is keep (.) x
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
            if constexpr (RW > 1) { ld_k[e] = 1.f - keep; ld_s[e] = s; }
            kx[e] = (1.f - keep) * xn[e];
          }
          *reinterpret_cast<f32x4*>(xs + idx) = xn;
          // second sub-update follows: its keep mask is the complement (gauge_dynamics.py:434-437, :472-475)
          if (prep_next_mask) *reinterpret_cast<f32x4*>(gs + idx) = kx;
        }
      }
      // row r's log-det share of this wave: lanes r, r + 16, r + 32, r + 48 (fixed order: bit-reproducible)
      if constexpr (RW == 1) {
        ld += __shfl_xor(ld, 16, 64);
        ld += __shfl_xor(ld, 32, 64);
        if (q == 0) ldw[wave * kFM + r] += ld;
      } else {
        // Two waves share an image wave's columns.  The bits of the 4-wave forms are those of ONE chain of adds per
        // lane over both waves' tiles, then the cross-lane steps: the even wave hands its lane sum over, the odd wave
        // continues the chain with its own four terms behind the barrier below (ld_k, ld_s) and does the rest.
        if (wsub == 0) ldx[wimg * 64 + lane] = ld;
      }
      FT_ADD(5, t0);
    }
    {
      [[maybe_unused]] const unsigned long long tb = FT_NOW();
      __syncthreads();
      FT_ADD(6, tb);
    }
    if constexpr (RW > 1) {
      if (wsub == 1) {
        float ld = ldx[wimg * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) ld += ld_k[e] * ld_s[e];      // (the same contracted multiply-add as the chain above)
        ld += __shfl_xor(ld, 16, 64);
        ld += __shfl_xor(ld, 32, 64);
        if (q == 0) ldw[wimg * kFM + r] += ld;
      }
    }
  };

  // ---- leapfrog steps -----------------------------------------------------------
  const float two_pi = 6.28318530717958647692f;
  for (int step = p.step_begin; step < p.step_end; ++step) {
    const int sf = step, sb = p.num_steps - 1 - step;       // gauge_dynamics.py:453-457
    const float af = two_pi * (float)sf / (float)p.num_steps, ab = two_pi * (float)sb / (float)p.num_steps;
    const float tcf = cosf(af), tsf = sinf(af), tcb = cosf(ab), tsb = sinf(ab);
    const float tcr = dirl ? tcb : tcf, tsr = dirl ? tsb : tsf;      // time encoding of this lane's row
    for (int i = tid; i < D; i += kFThreads) {
      skm[i] = p.masks[(size_t)sf * D + i];
      skm[D + i] = p.masks[(size_t)sb * D + i];
    }
    // (gs holds the force of the current x: from the prologue or the previous step's last kick)
    __syncthreads();
    // the four network calls of a leapfrog step run through ONE copy of the code (runtime parameters,
    // wave-uniform branches): the kernel stays well inside the instruction cache
#pragma nounroll
    for (int call = 0; call < 4; ++call) {
      const bool is_v = call == 0 || call == 3;
      if (call == 3) {
        [[maybe_unused]] const unsigned long long tf = FT_NOW();
        (void)force_pass();                                    // force at the new position
        FT_ADD(7, tf);
      }
      // call 0: momentum half-kick (+ keep (.) x into gs)      call 1: position sub-update 1 (+ complement mask)
      // call 2: position sub-update 2                          call 3: second momentum half-kick (product kept)
      const int l1 = call == 0 ? (keep_v_valid ? 2 : 0) : call == 1 ? 3 : call == 2 ? 4 : 1;
      net_update(is_v ? p.vnet : p.xnet, is_v ? cv : cx, is_v ? xs : vs, is_v ? 1 : 2, call == 2 ? 1 : 0,
                 call < 2, l1, is_v, tcr, tsr, 2 * step + (call == 0 || call == 1 ? 0 : 1));
    }
    keep_v_valid = true;
  }

  // ---- epilogue: energies, accept probability, write back -------------------------
  const float act1 = force_pass();
  const float kin1 = kinetic_pass();
  if (STEPM) {
    if (own && fl == 0) {
      float sld = 0.f;
#pragma unroll
      for (int w = 0; w < IMGW; ++w) sld += ldw[w * kFM + fc];
      const double dh = (double)p.beta * ((double)act0 - (double)act1) + ((double)kin0 - (double)kin1) + (double)sld;
      spx[fc] = accept_from_delta(dh);
    }
    __syncthreads();
    // ---- mix the two directions, Metropolis-Hastings (gauge_dynamics.py:221-257, arithmetic kept as
    //      mask * a + (1 - mask) * b); x_in -> gs rows, x_out -> h1 rows (both free now)
    float* gin = gs;
    float* gout = h1;
    for (int i = tid; i < cpw * (D / 4); i += kFThreads) {
      const int k = i / (D / 4), c4 = (i - k * (D / 4)) * 4;
      const int64_t chain = (int64_t)blockIdx.x * cpw + k;
      f32x4 xin = {0.f, 0.f, 0.f, 0.f};
      if (chain < p.step_Bl) xin = *reinterpret_cast<const f32x4*>(p.x0 + chain * D + c4);
      f32x4 xp;
      float pk;
      if (p.step_both) {
        const float fm = scoin[k] > 0.5f ? 1.f : 0.f, bm = 1.f - fm;
        pk = fm * spx[k] + bm * spx[kFM / 2 + k];
        const f32x4 xf = *reinterpret_cast<const f32x4*>(xs + k * SX + c4);
        const f32x4 xb = *reinterpret_cast<const f32x4*>(xs + (kFM / 2 + k) * SX + c4);
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
      if (chain < p.step_Bl) {                                       // apply_transition's own outputs (:259)
        if (p.step_xprop) *reinterpret_cast<f32x4*>(p.step_xprop + chain * D + c4) = xp;
        if (p.step_xout) *reinterpret_cast<f32x4*>(p.step_xout + chain * D + c4) = xo;
        if (p.step_vprop) {
          f32x4 vp = *reinterpret_cast<const f32x4*>(vs + k * SX + c4);
          if (p.step_both) {
            const float fm = scoin[k] > 0.5f ? 1.f : 0.f, bm = 1.f - fm;
            vp = fm * vp + bm * *reinterpret_cast<const f32x4*>(vs + (kFM / 2 + k) * SX + c4);
          }
          *reinterpret_cast<f32x4*>(p.step_vprop + chain * D + c4) = vp;
        }
      }
    }
    __syncthreads();
    // ---- observables of the step's INPUT samples (gauge_model.py:256-266) and the charge of its output (:718-725)
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
    if (p.step_both) {
      float a, b;
      plaq_sums(fc < kFM / 2 ? gin + fc * SX : gout + (fc - kFM / 2) * SX, a, b);
      if (own && fl == 0) {
        if (fc < kFM / 2) { sobs[fc * 4 + 0] = a; sobs[fc * 4 + 1] = b; }
        else sobs[(fc - kFM / 2) * 4 + 2] = b;
      }
    } else {
      float a, b, c_, d_;
      plaq_sums(gin + fc * SX, a, b);
      plaq_sums(gout + fc * SX, c_, d_);
      if (own && fl == 0) { sobs[fc * 4 + 0] = a; sobs[fc * 4 + 1] = b; sobs[fc * 4 + 2] = d_; }
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
      // [sum p_accept, sum |dQ|, chains] in a fixed order and without a further launch: every workgroup leaves its
      // partial sums in step_part, the last one to arrive (ticket in step_sums[3]) adds them up and resets the ticket
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
        float a0 = 0.f, a1 = 0.f;
        for (int b = tid; b < (int)gridDim.x; b += kFThreads) {
          a0 += p.step_part[2 * b];
          a1 += p.step_part[2 * b + 1];
        }
        float* fin = vs;                                  // [2][kFThreads] scratch (vs is dead)
        fin[tid] = a0;
        fin[kFThreads + tid] = a1;
        __syncthreads();
        for (int st = kFThreads / 2; st > 0; st >>= 1) {
          if (tid < st) {
            fin[tid] += fin[tid + st];
            fin[kFThreads + tid] += fin[kFThreads + tid + st];
          }
          __syncthreads();
        }
        if (tid == 0) {
          p.step_sums[0] = p.step_sums_acc ? p.step_sums[0] + fin[0] : fin[0];          // (a batch cut into two launches)
          p.step_sums[1] = p.step_sums_acc ? p.step_sums[1] + fin[kFThreads] : fin[kFThreads];
          p.step_sums[2] = (float)p.step_B;
          *reinterpret_cast<int*>(p.step_sums + 3) = 0;
        }
      }
    }
    // ---- np.mod(x_out, 2 pi) (gauge_model.py:1388) and the write-back of the chains' new state
    for (int i = tid; p.step_x_next && i < cpw * (D / 4); i += kFThreads) {
      const int k = i / (D / 4), c4 = (i - k * (D / 4)) * 4;
      const int64_t chain = (int64_t)blockIdx.x * cpw + k;
      if (chain < p.step_Bl) {
        f32x4 w = *reinterpret_cast<const f32x4*>(gout + k * SX + c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float tp = 6.28318530717958647692f;
          float m_ = fmaf(-tp, floorf(w[e] * 0.15915494309189533577f), w[e]);       // w - 2 pi floor(w / 2 pi)
          if (m_ < 0.f) m_ += tp;
          if (m_ >= tp) m_ -= tp;
          w[e] = m_;
        }
        *reinterpret_cast<f32x4*>(p.step_x_next + chain * D + c4) = w;
      }
    }
    return;
  }
  if (own && fl == 0 && fc < nrow) {
    float sld = 0.f;

#pragma unroll
    for (int w = 0; w < IMGW; ++w) sld += ldw[w * kFM + fc];      // fixed order: bit-reproducible
    const int64_t rr = row0 + fc;
    if (p.logdet) p.logdet[rr] = p.logdet_accumulate ? p.logdet[rr] + sld : sld;
    if (p.p_accept) {
      // gauge_dynamics.py:592-609; the O(100) Hamiltonians are differenced in fp64
      const double dh = (double)p.beta * ((double)act0 - (double)act1) + ((double)kin0 - (double)kin1) +
                        (double)sld;
      p.p_accept[rr] = accept_from_delta(dh);
    }
  }
#ifdef L2HMC_STAMPS
  if (p.stamps && tid == 0) {
    unsigned long long rt1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1)::"memory");
    ft[9] = FT_NOW() - ft_start;
    for (int i = 0; i < 10; ++i) p.stamps[blockIdx.x * 12 + i] = ft[i];
    p.stamps[blockIdx.x * 12 + 10] = rt0;
    p.stamps[blockIdx.x * 12 + 11] = rt1;
  }
#endif
  for (int i = tid; i < kFM * (D / 4); i += kFThreads) {
    const int rr = i / (D / 4), c4 = (i - rr * (D / 4)) * 4;
    if (rr < nrow) {
      *reinterpret_cast<f32x4*>(p.x_out + (row0 + rr) * D + c4) = *reinterpret_cast<const f32x4*>(xs + rr * SX + c4);
      *reinterpret_cast<f32x4*>(p.v_out + (row0 + rr) * D + c4) = *reinterpret_cast<const f32x4*>(vs + rr * SX + c4);
    }
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
// shapes with a whole-trajectory kernel: GenericNet on D=128 (H=512), and the dense trunk of ConvNet3D on the
// 8x8 lattice (features 64+64, H=256)
static int fused_generic_net(const l2hmc_dense_net* n) {
  return n->D == 128 && n->H == 512 && n->Ka == 128 && n->Kb == 128;
}
static int fused_conv_net(const l2hmc_dense_net* n) {
  return n->D == 128 && n->H == 256 && n->Ka == 64 && n->Kb == 64;
}
int fused_net_supported(const l2hmc_dense_net* n) { return fused_generic_net(n) || fused_conv_net(n); }

// L2HMC_PLAN_TILES16_ONLY keeps every batch on the 16-row form (A/B and the bit-identity test)
static bool subtile_enabled(const l2hmc_gauge_plan* p) { return !(p->flags & L2HMC_PLAN_TILES16_ONLY); }
// CUs of the current device (one 16-row workgroup each per round); asked once per device
static int device_cu_count() {
  static std::atomic<int> cached[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  int n = cached[dev].load(std::memory_order_relaxed);
  if (n <= 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}
int fused_plan_supported(const l2hmc_gauge_plan* p) {
  if (p->hmc || !p->xnet.packed || !p->vnet.packed || 2 * p->T * p->X != 128 || (p->X & (p->X - 1)) != 0) return 0;
  if (p->flags & L2HMC_PLAN_CONV3D)
    return fused_conv_net(&p->xnet) && fused_conv_net(&p->vnet) && p->T == 8 && p->X == 8 && p->xfront.F == 8 &&
           p->vfront.F == 8;
  return fused_generic_net(&p->xnet) && fused_generic_net(&p->vnet);
}

int launch_fused_trajectory(const l2hmc_gauge_plan* p, float beta, int step_begin, int step_end,
                            const float* x0, const float* v0, const int* dir, int64_t rows, float* x_out,
                            float* v_out, float* logdet, int logdet_accumulate, float* p_accept,
                            hipStream_t stream, int64_t x_mod, int64_t dir_split, const FusedTape* tape_x,
                            const FusedTape* tape_v) {
  const bool conv = (p->flags & L2HMC_PLAN_CONV3D) != 0;
  using CfgG = FusedCfg<128, 512, 128, false>;
  using CfgC = FusedCfg<128, 256, 64, true>;
  static DeviceOnce attr_once;
  const bool tape = tape_x && tape_v;
  const size_t lds = sizeof(float) * (conv ? CfgC::LDS_FLOATS : CfgG::LDS_FLOATS);
  if (attr_once.pending()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gauge_traj_fused_kernel<128, 512, 128, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(sizeof(float) * CfgG::LDS_FLOATS)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gauge_traj_fused_kernel<128, 256, 64, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(sizeof(float) * CfgC::LDS_FLOATS)) != hipSuccess) {
      set_error("fused trajectory: cannot reserve %zu B of LDS", lds);
      return L2HMC_ERR_HIP;
    }
    attr_once.done();
  }
  FusedArgs a{};
  a.T = p->T; a.X = p->X; a.num_steps = p->num_steps; a.step_begin = step_begin; a.step_end = step_end;
  a.eps = p->eps; a.beta = beta; a.masks = p->masks; a.xnet = p->xnet; a.vnet = p->vnet;
  a.xfront = p->xfront; a.vfront = p->vfront;
  a.x0 = x0; a.v0 = v0; a.dir = dir; a.rows = rows; a.x_out = x_out; a.v_out = v_out;
  a.x_mod = x_mod; a.dir_split = dir_split;
  a.logdet = logdet; a.logdet_accumulate = logdet_accumulate; a.p_accept = p_accept;
  if (tape) {
    L2HMC_REQUIRE(step_begin == 0, "fused trajectory: taping needs the whole trajectory");
    L2HMC_REQUIRE(!conv || (tape_x->feat && tape_v->feat), "fused trajectory: ConvNet3D taping needs the feature tape");
    L2HMC_REQUIRE(tape_x->in && tape_x->h1 && tape_x->h2 && tape_x->stq && tape_x->st && tape_v->in && tape_v->h1 &&
                      tape_v->h2 && tape_v->stq && tape_v->st && tape_x->gate && tape_v->gate,
                  "fused trajectory: NULL tape pointer");
    a.tx = *tape_x;
    a.tv = *tape_v;
    static DeviceOnce tape_once;
    if (tape_once.pending()) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gauge_traj_fused_kernel<128, 512, 128, false, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(sizeof(float) * CfgG::LDS_FLOATS)) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(&gauge_traj_fused_kernel<128, 256, 64, true, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(sizeof(float) * CfgC::LDS_FLOATS)) != hipSuccess) {
        set_error("fused trajectory: cannot reserve %zu B of LDS", lds);
        return L2HMC_ERR_HIP;
      }
      tape_once.done();
    }
  }
#ifdef L2HMC_STAMPS
  a.stamps = g_stamp_cls == 5 ? g_stamp_buf : nullptr;
  a.stagger = g_fused_stagger;
#endif
  // batches that cannot put a 16-row tile on every CU: the sub-tile form (4 or 8 rows per workgroup, same bits)
  if (!conv && !tape && subtile_enabled(p)) {
    if (const int rpw = fused4_rows_per_wg(rows, device_cu_count())) return launch_fused4(a, rpw, stream);
    // more than one round of 16-row workgroups: the 32-row form (each weight fragment feeds two MFMAs)
    if (rows > (int64_t)kFM * device_cu_count()) return launch_fused32(a, stream);
  }
  const dim3 grid((unsigned)ceil_div(rows, kFM));
  prof_before(kProfFused, stream);
  if (conv && tape)
    hipLaunchKernelGGL((gauge_traj_fused_kernel<128, 256, 64, true, true>), grid, dim3(CfgC::THREADS), lds, stream, a);
  else if (conv)
    hipLaunchKernelGGL((gauge_traj_fused_kernel<128, 256, 64, true>), grid, dim3(CfgC::THREADS), lds, stream, a);
  else if (tape)
    hipLaunchKernelGGL((gauge_traj_fused_kernel<128, 512, 128, false, true>), grid, dim3(CfgG::THREADS), lds, stream, a);
  else
    hipLaunchKernelGGL((gauge_traj_fused_kernel<128, 512, 128, false>), grid, dim3(CfgG::THREADS), lds, stream, a);
  prof_after(kProfFused, stream);
  L2HMC_CHECK_LAUNCH("gauge_traj_fused");
  return L2HMC_OK;
}

// One launch = one whole MCMC step of B chains (see FusedArgs::step_*).  part: 2 * ceil(B / cpw) floats of scratch.
// The launches of one MCMC step over `rows_all` chain-rows (see launch_fused_step): at most three {rows, rows per
// workgroup} parts, every cut at an even row count.  forms = the sub-tile and 32-row forms may be used.
struct StepPart { int64_t rows; int rpw; };
static int plan_step_parts(int64_t rows_all, bool forms, int cus, StepPart (&parts)[3]) {
  int n = 0;
  if (!forms) {
    parts[n++] = {rows_all, kFM};
  } else if (const int all = fused4_rows_per_wg(rows_all, cus)) {
    parts[n++] = {rows_all, all};
  } else {
    const int64_t round16 = (int64_t)kFM * cus;                          // rows in one full round of 16-row workgroups
    const int64_t main32 = rows_all / (2 * round16) * (2 * round16), rem = rows_all - main32;
    const int64_t over = rem - round16;                                  // rows beyond one more 16-row round
    if (rem > round16 && !(over <= 8 * (int64_t)cus && fused4_rows_per_wg(over, cus))) {
      parts[n++] = {rows_all, 32};                                       // the rest fills most of another 32-row round
    } else {
      if (main32 > 0) parts[n++] = {main32, 32};
      if (rem > round16) {
        parts[n++] = {round16, kFM};
        parts[n++] = {over, fused4_rows_per_wg(over, cus)};
      } else if (rem > 0) {
        const int sub = fused4_rows_per_wg(rem, cus);
        parts[n++] = {rem, sub ? sub : kFM};
      }
    }
  }
  return n;
}
// host logic only, no device call: the launch plan for `rows_all` rows on `cus` CUs
extern "C" int l2hmc_gauge_step_plan(int64_t rows_all, int32_t cus, int64_t* rows_out, int32_t* rpw_out) {
  if (rows_all <= 0 || cus <= 0 || !rows_out || !rpw_out) {
    set_error("gauge_step_plan: rows_all=%lld cus=%d (both > 0) and two output arrays of 3 entries", (long long)rows_all, cus);
    return L2HMC_ERR_ARG;
  }
  StepPart parts[3];
  const int n = plan_step_parts(rows_all, true, cus, parts);
  for (int i = 0; i < n; ++i) {
    rows_out[i] = parts[i].rows;
    rpw_out[i] = parts[i].rpw;
  }
  return n;
}

int launch_fused_step(const l2hmc_gauge_plan* p, float beta, const float* x_in, float* x_next, int64_t B,
                      uint64_t seed, uint64_t draw, int both, float* px, float* actions, float* plaqs, float* charges,
                      float* dq, float* step_sums, float* part, hipStream_t stream, float* x_prop, float* v_prop,
                      float* x_out) {
  const bool conv = (p->flags & L2HMC_PLAN_CONV3D) != 0;
  using CfgG = FusedCfg<128, 512, 128, false>;
  using CfgC = FusedCfg<128, 256, 64, true>;
  static DeviceOnce step_once;
  const size_t lds = sizeof(float) * (conv ? CfgC::LDS_FLOATS : CfgG::LDS_FLOATS);
  if (step_once.pending()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gauge_traj_fused_kernel<128, 512, 128, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(sizeof(float) * CfgG::LDS_FLOATS)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gauge_traj_fused_kernel<128, 256, 64, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(sizeof(float) * CfgC::LDS_FLOATS)) != hipSuccess) {
      set_error("fused step: cannot reserve %zu B of LDS", lds);
      return L2HMC_ERR_HIP;
    }
    step_once.done();
  }
  L2HMC_REQUIRE(x_in && (x_next || x_out) && B > 0 && (!step_sums || part), "fused step: bad arguments");
  // Which form runs which chains (GenericNet 8x8 plans; all forms give the same bits).  The 16-row form covers
  // 16 * (number of CUs) rows per round of workgroups (1.58 ms at the benchmark dynamics), so a batch one chain past a
  // round costs a whole further round.  The batch is cut into at most three parts, launched one after the other on the
  // same stream (a part: chains [c0, c0 + n), global chain indices for the Philox streams, pointers moved to its first
  // chain, the step's sums added to the earlier parts'):
  //   - whole rounds of 32-row workgroups (fused_traj32.hip: twice the rows of a 16-row round in 1.87 x its time);
  //   - what is left: nothing / a sub-tile launch (fused_traj4.hip: up to 3072 rows in 0.9-1.5 ms) / one 16-row round /
  //     a 16-row round and a sub-tile launch (when the rest beyond the round is small) / one more 32-row round;
  //   - a batch that is small as a whole is one sub-tile launch, a batch of at most one round one 16-row launch.
  const int ndir = both ? 2 : 1;
  StepPart parts[3];
  const int nparts = plan_step_parts(B * ndir, !conv && subtile_enabled(p), device_cu_count(), parts);
  auto part_of = [&](int64_t c0, int64_t nb, int rpw, int accumulate) {
    const int cpw = both ? rpw / 2 : rpw;
    const int64_t D = 2 * (int64_t)p->T * p->X;
    FusedArgs a{};
    a.T = p->T; a.X = p->X; a.num_steps = p->num_steps; a.step_begin = 0; a.step_end = p->num_steps;
    a.eps = p->eps; a.beta = beta; a.masks = p->masks; a.xnet = p->xnet; a.vnet = p->vnet;
    a.xfront = p->xfront; a.vfront = p->vfront;
    a.x0 = x_in + c0 * D; a.rows = ceil_div(nb, cpw) * rpw;
    a.step_x_next = x_next ? x_next + c0 * D : nullptr;
    a.step_xprop = x_prop ? x_prop + c0 * D : nullptr;
    a.step_vprop = v_prop ? v_prop + c0 * D : nullptr;
    a.step_xout = x_out ? x_out + c0 * D : nullptr;
    a.step_B = B; a.step_Bl = nb; a.step_chain0 = c0; a.step_sums_acc = accumulate;
    a.step_seed = seed; a.step_draw = draw; a.step_both = both;
    a.step_px = px ? px + c0 : nullptr; a.step_act = actions ? actions + c0 : nullptr;
    a.step_plq = plaqs ? plaqs + c0 : nullptr; a.step_chg = charges ? charges + c0 : nullptr;
    a.step_dq = dq ? dq + c0 : nullptr;
    a.step_sums = step_sums; a.step_part = part;
#ifdef L2HMC_STAMPS
    a.stamps = g_stamp_cls == 5 ? g_stamp_buf : nullptr;
    a.stagger = g_fused_stagger;
#endif
    return a;
  };
  auto launch16 = [&](const FusedArgs& a) {
    const unsigned nwg = (unsigned)(a.rows / kFM);
    prof_before(kProfFused, stream);
    if (conv)
      hipLaunchKernelGGL((gauge_traj_fused_kernel<128, 256, 64, true>), dim3(nwg), dim3(CfgC::THREADS), lds, stream, a);
    else
      hipLaunchKernelGGL((gauge_traj_fused_kernel<128, 512, 128, false>), dim3(nwg), dim3(CfgG::THREADS), lds, stream, a);
    prof_after(kProfFused, stream);
    L2HMC_CHECK_LAUNCH("gauge_traj_fused (step)");
    return L2HMC_OK;
  };
  int64_t c0 = 0;
  for (int i = 0; i < nparts; ++i) {
    const int64_t nb = parts[i].rows / ndir;                             // (every cut is at an even row count)
    const FusedArgs a = part_of(c0, nb, parts[i].rpw, i > 0);
    if (int e = parts[i].rpw == 32 ? launch_fused32(a, stream)
                : parts[i].rpw == kFM ? launch16(a) : launch_fused4(a, parts[i].rpw, stream))
      return e;
    c0 += nb;
  }
  return L2HMC_OK;
}

}  // namespace l2hmc

using namespace l2hmc;

extern "C" size_t l2hmc_dense_pack_bytes(const l2hmc_dense_net* net) {
  if (!net || !fused_net_supported(net)) return 0;
  // the 16-row form's image, then (GenericNet plans) the sub-tile form's (fused_traj4.hip)
  return sizeof(float) * ((size_t)(net->Ka + net->Kb) * net->H + (size_t)net->H * net->H + (size_t)3 * net->D * net->H +
                          fused4_pack_floats(net));
}

extern "C" int l2hmc_dense_pack(const l2hmc_dense_net* net, float* packed, l2hmc_stream_t stream) {
  L2HMC_REQUIRE(net != nullptr && packed != nullptr, "dense_pack: NULL pointer");
  L2HMC_REQUIRE(fused_net_supported(net), "dense_pack: shape (D=%d, H=%d, Ka=%d, Kb=%d) has no fused kernel",
                net->D, net->H, net->Ka, net->Kb);
  L2HMC_REQUIRE(net->w1_t && net->wh_t && net->whd_t, "dense_pack: NULL weight pointer");
  const int waves = fused_conv_net(net) ? FusedCfg<128, 256, 64, true>::IMGW : FusedCfg<128, 512, 128, false>::IMGW;
  hipLaunchKernelGGL(pack_fused_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, *net, packed, waves);
  L2HMC_CHECK_LAUNCH("dense_pack");
  if (fused4_pack_floats(net))
    return launch_fused4_pack(net, packed + (size_t)(net->Ka + net->Kb) * net->H + (size_t)net->H * net->H +
                                        (size_t)3 * net->D * net->H, (hipStream_t)stream);
  return L2HMC_OK;
}
