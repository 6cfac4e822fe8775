// Convolutional front-end of ConvNet3D (channels_last) for gfx950:
//   l2hmc/network/conv_net.py:247-262  reshape_5D -> Conv3D(F,(3,3,2),same,relu) -> MaxPool3D(2,s2,same)
//                                      -> Conv3D(2F,(2,2,2),same,relu) -> MaxPool3D(2,s2,same) -> flatten
// applied to each of the two network inputs with its own filters (conv_v*, conv_x*).
// Keras/TF conventions restated (they are not in the reference tree): zero padding, 'same' pads
// floor((k-1)/2) before and the rest after, pooling windows are clipped at the border, flatten is
// row-major over (h, w, depth, channel).  On the link-direction axis (size 2) the (.,.,2) kernel sees
// one real and one padded tap for output depth 1, and after the first pool that axis has size 1, so
// only the dd=0 slice of the second kernel ever touches data.
//
// One launch handles both inputs (blockIdx.y).  A chain is a few hundred floats: it is staged in LDS
// with a zero halo, conv1+relu+pool1 and conv2+relu+pool2 run out of LDS (max and relu commute), and
// the flattened features go straight into the k-contiguous layout the dense trunk's first GEMM reads.
// This stage is <20 % of the network's FLOPs and VALU-bound; filters are read with
// the channel index on the lane (conflict-free), inputs as LDS broadcasts.
#include <algorithm>
#include <atomic>
#include <type_traits>

#include "stq_dense.h"

namespace l2hmc {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int kConvThreads = 256;

// FT / LT: compile-time filter count and (square) lattice extent, 0 = take them from the arguments.  The index
// arithmetic of every phase divides by F, X/2, ...: with constants these are shifts, with run-time values
// ~30-instruction sequences that dominate the kernel.
#ifdef L2HMC_STAMPS
#define CF_STAMP(i)                                                                          \
  do {                                                                                       \
    if (p.stamps && threadIdx.x == 0) {                                                      \
      unsigned long long t_;                                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                     \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
      __builtin_amdgcn_sched_barrier(0);                                                     \
      p.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = t_;               \
    }                                                                                        \
  } while (0)
#else
#define CF_STAMP(i) do {} while (0)
#endif

template <int FT, int LT>
__global__ __launch_bounds__(kConvThreads) __attribute__((amdgpu_waves_per_eu(4, 8))) void conv3d_front_kernel(ConvFrontArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int which = p.only ? p.only - 1 : blockIdx.y;   // 0: first input, 1: second input
  const int T = LT ? LT : p.T, X = LT ? LT : p.X, F = FT ? FT : p.F, F2 = 2 * F;
  const int D = 2 * T * X;
  const int TP = T + 2, XP = X + 2;               // conv1 halo (pad 1 / 1)
  const int T2 = T / 2, X2 = X / 2, T2P = T2 + 1, X2P = X2 + 1;   // conv2 pad (0 / 1)
  const int T4 = T / 4, X4 = X / 4;
  const int cpw = p.cpw;
  // channel stride of the pooled conv1 map: F + 4 in the compile-time instances, whose second convolution runs on the
  // matrix pipe (16-byte fragment reads of 16 positions: rows of 20 / 12 floats fall on distinct bank groups)
  const int FS = FT ? F + 4 : F;
  const int tid = threadIdx.x;
  const int64_t row0 = (int64_t)blockIdx.x * cpw;
  const int nrow = (int)min((int64_t)cpw, p.rows - row0);
  using f32x2 = __attribute__((ext_vector_type(2))) float;
  float* out = p.out[which];
  if constexpr (FT > 0) {
    // Compile-time shapes (the BASELINE configurations).  Round 4: NO weight goes through LDS -- a wave's conv1 taps
    // are wave-uniform (scalar loads, SGPR operands) and a lane's conv2 fragments are read from the (L2-resident) Keras
    // tensor straight into registers; only the halo cells are zeroed (disjoint from the staged interior: no barrier in
    // between) and the chain is staged in 16-byte pieces.  LDS per workgroup is the two maps alone, so the launcher
    // packs enough chains into a workgroup for ONE round of workgroups (launch_conv3d_front).
    // Before (weights packed into LDS by every workgroup, whole-map zero fill, three barriers): 37.3 / 22.2 us per
    // launch (both inputs / one) at 16 x 16, F = 16 with 40-46 % of the wave cycles parked
    // (profiles/r03_pmc_cfg4_kernels.txt); now 25.5 / 14.6 us (profiles/r04_conv_fwd_stamps.txt).
    constexpr int FP = FT / 2;
    CF_STAMP(0);
#ifdef L2HMC_STAMPS
    if (p.stamps && threadIdx.x == 0) {              // where this workgroup runs: HW_ID | XCC_ID << 32, and the wall clock
      unsigned hw, xcc;
      unsigned long long rt;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt)::"memory");
      p.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + 6] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
      p.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + 7] = rt;
    }
#endif
    float* xin = lds;                                // [cpw][TP][XP][2]
    float* p1 = xin + cpw * TP * XP * 2;             // [cpw][T2P][X2P][FS]
    const float* gw1 = p.w1[which];
    const float* gw2 = p.w2[which];
    // ---- stage the chains (zero halo): element quad e .. e + 3 = sites s, s + 1 of one lattice row, both links
    {
      const float* in = p.in[which];
      const bool masked = which == 1 && p.cmask_f != nullptr;
      for (int i = tid; i < nrow * (D / 4); i += kConvThreads) {
        const int c = i / (D / 4), e = (i - c * (D / 4)) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(in + (row0 + c) * p.ldi + e);
        if (masked) {
          const int d = p.dir ? p.dir[row0 + c] : 0;
          v *= *reinterpret_cast<const f32x4*>((d ? p.cmask_b : p.cmask_f) + e);
        }
        const int site = e >> 1;
        const int ii = site / X, jj = site - ii * X;
        float* dst = xin + ((c * TP + ii + 1) * XP + jj + 1) * 2;       // 8-byte aligned (jj is even)
        *reinterpret_cast<f32x2*>(dst) = f32x2{v[0], v[1]};
        *reinterpret_cast<f32x2*>(dst + 2) = f32x2{v[2], v[3]};
      }
      constexpr int NH = 2 * (LT + 2) + 2 * LT;        // halo cells of a chain: rows 0 and TP - 1, columns 0 and XP - 1
      for (int i = tid; i < cpw * NH; i += kConvThreads) {
        const int c = i / NH, h = i - c * NH;
        const int hi = h < XP ? 0 : h < 2 * XP ? TP - 1 : h < 2 * XP + T ? 1 + (h - 2 * XP) : 1 + (h - 2 * XP - T);
        const int hj = h < XP ? h : h < 2 * XP ? h - XP : h < 2 * XP + T ? 0 : XP - 1;
        *reinterpret_cast<f32x2*>(xin + ((c * TP + hi) * XP + hj) * 2) = f32x2{0.f, 0.f};
      }
      constexpr int NP = (LT / 2 + 1) + LT / 2;        // pad cells of a pooled map: row T2 and column X2
      constexpr int FS4 = (FT + 4) / 4;
      for (int i = tid; i < cpw * NP * FS4; i += kConvThreads) {
        const int c = i / (NP * FS4), rest = i - c * (NP * FS4);
        const int h = rest / FS4, ch = (rest - h * FS4) * 4;
        const int hi = h < X2P ? T2 : h - X2P, hj = h < X2P ? h : X2;
        *reinterpret_cast<f32x4*>(p1 + ((c * T2P + hi) * X2P + hj) * FS + ch) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    // conv2 operands of a lane: filter 16 gt + r, tap q, channels 4 s .. 4 s + 3 of the Keras tensor
    // [di][dj][dd = 0][c][g], requested in front of the second barrier (earlier they cost the conv1 stage registers)
    constexpr int PT = (LT / 2) * (LT / 2) / 16;        // 16-position tiles per chain (16 x 16: 4; 8 x 8: 1)
    constexpr int GT = 2 * FT / 16;                     // 16-filter tiles (F = 16: 2; F = 8: 1)
    constexpr int NS = FT / 4;                          // k-steps per tap group (4 channels each)
    constexpr int RW = 16 / (LT / 2) < 1 ? 1 : 16 / (LT / 2);   // rows of the position grid per tile (2 or 4)
    static_assert(LT / 2 <= 8 && (LT / 2) * (LT / 2) % 16 == 0 && (2 * FT) % 16 == 0 && FT % 4 == 0,
                  "a 16-position tile must hold whole 2 x 2 pooling windows");
    const int lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, r = lane & 15;
    const int di = q >> 1, dj = q & 1;
    // ---- conv1 (VALU) and conv2 (matrix pipe), pipelined over groups of chains.  A group is as many chains as give
    // every wave ONE conv2 tile (16 x 16: one chain, 8 x 8: four); the conv2 MFMAs of group g and the conv1 arithmetic
    // of group g + 1 are independent and share a phase between two barriers.  A wave still issues them one after the
    // other -- on this part no VALU instruction issues under the wave's OWN matrix instruction, only under another
    // wave's, at half rate (profiles/r04_mfma_valu_overlap_bench.txt) -- so what the phase split buys is that the
    // products of one wave can run beside the conv1 arithmetic of the others.  (Per CU and two-input launch at 16 x 16
    // the matrix pipe has 16 k cycles of work and the VALU 12-16 k; the counters of the shipped kernel read 27 % / 31 %
    // of the SIMD cycles; the four workgroups of a CU finish between 10 and 23 us after a common start: oldest first.)
    // conv1 (3,3,2) + relu + pool (2,2,2): a thread owns a PAIR of filters, so every multiply-add is a v_pk_fma_f32,
    // and the 4 x 4 x 2 input patch under a pooling window is read once.
    // conv2 (2,2,[2]) + relu + pool: per chain a [T2 X2 positions] x [4 F] x [2 F] product.  v_mfma_f32_16x16x4_f32
    // with the WEIGHTS as first operand (16 filters x 4 k) and 16 positions as second: lane (q, r) supplies tap
    // q = (di, dj) of position r, channels 4 s .. 4 s + 3 (one ds_read_b128 per step s, element e = channel 4 s + e), and
    // receives out[position r][filters 16 gt + 4 q .. + 3].  A wave takes one 16-position tile with every filter tile;
    // the 2 x 2 pooling partners of a position are lanes r ^ 1 and r ^ X2.
    constexpr int CELLS = (LT / 2) * (LT / 2);
    constexpr int CG = 4 / PT;                                   // chains per group
    constexpr int IT1 = CG * CELLS * FP / kConvThreads;          // conv1 items per thread and group (16 x 16: 2; 8 x 8: 1)
    static_assert(FT == 16 || FT == 8, "tap strides of the scalar loads");
    static_assert(CG * PT == kConvThreads / 64 && CG * CELLS * FP == IT1 * kConvThreads && CG * CELLS == 64 &&
                      FP == IT1 * (kConvThreads / 64),
                  "one conv2 tile per wave and group; a wave's lanes are the cells of a group, its filter pairs uniform");
    const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably uniform: the taps go through the scalar cache
    auto conv1_group = [&](int g, auto&& before_item) {
#pragma unroll
      for (int it = 0; it < IT1; ++it) {
        before_item(it);
        // this wave: filter pair fp of every cell of the group (lane = chain-in-group x cell); the pair's 18 taps are
        // wave-uniform -- scalar loads, SGPR operands of the packed multiply-adds, no vector register
        const int fp = wvu * IT1 + it;
        // (scalar loads by hand: behind the kernel's own stores the compiler only issues vector loads, and hoisting
        //  both items' taps out of the group loop wants 76 SGPRs)
        unsigned long long kk[18], kb;
        const float* tb = gw1 + 2 * fp;
        const float* bb1 = p.b1[which] + 2 * fp;
#define L2HMC_TAP_LOADS(ST)                                                                                          \
  asm volatile("s_load_dwordx2 %0, %19, 0\n\ts_load_dwordx2 %1, %19, " #ST "*1\n\ts_load_dwordx2 %2, %19, " #ST "*2\n\t"     \
               "s_load_dwordx2 %3, %19, " #ST "*3\n\ts_load_dwordx2 %4, %19, " #ST "*4\n\ts_load_dwordx2 %5, %19, " #ST "*5\n\t"  \
               "s_load_dwordx2 %6, %19, " #ST "*6\n\ts_load_dwordx2 %7, %19, " #ST "*7\n\ts_load_dwordx2 %8, %19, " #ST "*8\n\t"  \
               "s_load_dwordx2 %9, %19, " #ST "*9\n\ts_load_dwordx2 %10, %19, " #ST "*10\n\ts_load_dwordx2 %11, %19, " #ST "*11\n\t" \
               "s_load_dwordx2 %12, %19, " #ST "*12\n\ts_load_dwordx2 %13, %19, " #ST "*13\n\ts_load_dwordx2 %14, %19, " #ST "*14\n\t" \
               "s_load_dwordx2 %15, %19, " #ST "*15\n\ts_load_dwordx2 %16, %19, " #ST "*16\n\ts_load_dwordx2 %17, %19, " #ST "*17\n\t" \
               "s_load_dwordx2 %18, %20, 0\n\ts_waitcnt lgkmcnt(0)"                                                   \
               : "=&s"(kk[0]), "=&s"(kk[1]), "=&s"(kk[2]), "=&s"(kk[3]), "=&s"(kk[4]), "=&s"(kk[5]), "=&s"(kk[6]),       \
                 "=&s"(kk[7]), "=&s"(kk[8]), "=&s"(kk[9]), "=&s"(kk[10]), "=&s"(kk[11]), "=&s"(kk[12]), "=&s"(kk[13]),   \
                 "=&s"(kk[14]), "=&s"(kk[15]), "=&s"(kk[16]), "=&s"(kk[17]), "=&s"(kb)                                   \
               : "s"(tb), "s"(bb1)                                                                                    \
               : "memory")
        if constexpr (FT == 16) L2HMC_TAP_LOADS(64);      // tap (t, d) of the pair: 4 F bytes apart
        else L2HMC_TAP_LOADS(32);
#undef L2HMC_TAP_LOADS
        f32x2 k0[9], k1[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          k0[t] = __builtin_bit_cast(f32x2, kk[2 * t]);
          k1[t] = __builtin_bit_cast(f32x2, kk[2 * t + 1]);
        }
        const f32x2 bias1 = __builtin_bit_cast(f32x2, kb);
        const int cell = (tid & 63) % CELLS;
        const int I = cell / X2, J = cell % X2, c = g * CG + (tid & 63) / CELLS;
        f32x2 px[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int bb = 0; bb < 4; ++bb)
            px[a][bb] = *reinterpret_cast<const f32x2*>(xin + ((c * TP + 2 * I + a) * XP + 2 * J + bb) * 2);
        f32x2 m = {-INFINITY, -INFINITY};
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
          for (int bb = 0; bb < 2; ++bb) {
            f32x2 v0 = bias1, v1 = bias1;    // output depth 0 sees (mu = 0, mu = 1), depth 1 sees (mu = 1, pad)
#pragma unroll
            for (int ti = 0; ti < 3; ++ti) {
#pragma unroll
              for (int tj = 0; tj < 3; ++tj) {
                const f32x2 xx = px[a + ti][bb + tj];
                const f32x2 x0 = {xx[0], xx[0]}, x1 = {xx[1], xx[1]};
                v0 += x0 * k0[ti * 3 + tj];       // (three fused multiply-adds per tap)
                v0 += x1 * k1[ti * 3 + tj];
                v1 += x1 * k0[ti * 3 + tj];
              }
            }
            m[0] = fmaxf(m[0], fmaxf(v0[0], v1[0]));
            m[1] = fmaxf(m[1], fmaxf(v0[1], v1[1]));
          }
        }
        *reinterpret_cast<f32x2*>(p1 + ((c * T2P + I) * X2P + J) * FS + 2 * fp) = f32x2{fmaxf(m[0], 0.f), fmaxf(m[1], 0.f)};
        __builtin_amdgcn_sched_barrier(0);       // (one item's patch registers at a time)
      }
    };
    f32x4 wf[GT][NS], bias2[GT], acc[GT];
    const int c2l = wave / PT, pt = wave - c2l * PT;   // this wave's conv2 tile in a group: chain-in-group, tile
    const int I = pt * RW + r / X2, J = r % X2;         // this lane's position
    // the k-steps [S0, S1) of the wave's tile of group g
    auto conv2_products = [&](int g, auto s0, auto s1) {
      constexpr int S0 = decltype(s0)::value, S1 = decltype(s1)::value;
      const float* pb = p1 + (((g * CG + c2l) * T2P + I + di) * X2P + J + dj) * FS;
      if constexpr (S0 == 0) {
#pragma unroll
        for (int gt = 0; gt < GT; ++gt) acc[gt] = bias2[gt];
      }
#pragma unroll
      for (int sx = S0; sx < S1; ++sx) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(pb + 4 * sx);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int gt = 0; gt < GT; ++gt)
            acc[gt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[gt][sx][e], a[e], acc[gt], 0, 0, 0);
      }
    };
    auto conv2_finish = [&](int g) {      // relu + 2 x 2 max-pool across the position lanes (max and relu commute)
      const int c = g * CG + c2l;
#pragma unroll
      for (int gt = 0; gt < GT; ++gt) {
        f32x4 m = acc[gt];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          m[e] = fmaxf(m[e], __shfl_xor(m[e], 1, 64));
          m[e] = fmaxf(m[e], __shfl_xor(m[e], X2 >= 16 ? 0 : X2, 64));
          m[e] = fmaxf(m[e], 0.f);
        }
        if ((J & 1) == 0 && (I & 1) == 0 && c < nrow) {
          float* o = out + (row0 + c) * p.ldo + ((I >> 1) * X4 + (J >> 1)) * F2 + 16 * gt + 4 * q;
          *reinterpret_cast<f32x4*>(o) = m;
        }
      }
    };
    CF_STAMP(1);
    __syncthreads();
    CF_STAMP(2);
    conv1_group(0, [](int) {});
    CF_STAMP(3);
    // conv2 operands of a lane: filter 16 gt + r, tap q, channels 4 s .. 4 s + 3 of the Keras tensor [di][dj][dd = 0][c][g]
#pragma unroll
    for (int gt = 0; gt < GT; ++gt)
#pragma unroll
      for (int sx = 0; sx < NS; ++sx)
#pragma unroll
        for (int e = 0; e < 4; ++e) wf[gt][sx][e] = gw2[((size_t)(q * 2) * F + 4 * sx + e) * F2 + 16 * gt + r];
#pragma unroll
    for (int gt = 0; gt < GT; ++gt) bias2[gt] = *reinterpret_cast<const f32x4*>(p.b2[which] + 16 * gt + 4 * q);
    __syncthreads();
    CF_STAMP(4);
    const int ng = (nrow + CG - 1) / CG;      // (a group's missing chains are computed on whatever their LDS holds; never stored)
    using std::integral_constant;
    static_assert(NS % IT1 == 0, "the tile's k-steps are dealt evenly over the conv1 items they run under");
    for (int g = 0; g + 1 < ng; ++g) {
      conv1_group(g + 1, [&](int it) {          // (it is a compile-time constant after unrolling)
        if (it == 0) conv2_products(g, integral_constant<int, 0>{}, integral_constant<int, NS / IT1>{});
        else conv2_products(g, integral_constant<int, NS / IT1>{}, integral_constant<int, NS>{});
      });
      conv2_finish(g);
      __syncthreads();
    }
    conv2_products(ng - 1, integral_constant<int, 0>{}, integral_constant<int, NS>{});
    conv2_finish(ng - 1);
    CF_STAMP(5);
#ifdef L2HMC_STAMPS
    if (p.stamps && threadIdx.x == 0) {
      unsigned long long rt;
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt)::"memory");
      p.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + 15] = rt;
    }
#endif
    return;
  }
  // ---- run-time shapes: filters and maps through LDS
  float* w1 = lds;                                 // [3][3][2][F]
  float* b1 = w1 + 18 * F;
  float* w2 = b1 + F;                              // [2][2][F][2F]   (dd = 0 slice)
  float* b2 = w2 + 4 * F * F2;
  float* xin = b2 + F2;                            // [cpw][TP][XP][2]
  float* p1 = xin + cpw * TP * XP * 2;             // [cpw][T2P][X2P][FS]

  const float* gw1 = p.w1[which];
  const float* gw2 = p.w2[which];
  for (int i = tid; i < 18 * F; i += kConvThreads) w1[i] = gw1[i];
  for (int i = tid; i < F; i += kConvThreads) b1[i] = p.b1[which][i];
  for (int i = tid; i < 4 * F * F2; i += kConvThreads) {
    // Keras kernel [di][dj][dd][c][g]: keep dd = 0
    const int g = i % F2, c = (i / F2) % F, tap = i / (F2 * F);
    w2[i] = gw2[((size_t)(tap * 2 + 0) * F + c) * F2 + g];
  }
  for (int i = tid; i < F2; i += kConvThreads) b2[i] = p.b2[which][i];
  for (int i = tid; i < cpw * TP * XP * 2; i += kConvThreads) xin[i] = 0.f;
  for (int i = tid; i < cpw * T2P * X2P * FS; i += kConvThreads) p1[i] = 0.f;
  __syncthreads();
  const float* in = p.in[which];
  const bool masked = which == 1 && p.cmask_f != nullptr;
  for (int i = tid; i < nrow * D; i += kConvThreads) {
    const int c = i / D, e = i - c * D;
    float v = in[(row0 + c) * p.ldi + e];
    if (masked) {
      const int d = p.dir ? p.dir[row0 + c] : 0;
      v *= (d ? p.cmask_b : p.cmask_f)[e];
    }
    const int site = e >> 1, mu = e & 1;
    const int ii = site / X, jj = site - ii * X;
    xin[((c * TP + ii + 1) * XP + jj + 1) * 2 + mu] = v;
  }
  __syncthreads();

  // ---- conv1 (3,3,2) + relu + pool (2,2,2): pooled output (c, I, J, f), f on the lane
  const int n1 = nrow * T2 * X2 * F;
  for (int idx = tid; idx < n1; idx += kConvThreads) {
    const int f = idx % F;
    int r = idx / F;
    const int J = r % X2;
    r /= X2;
    const int I = r % T2, c = r / T2;
    const float bias = b1[f];
    float m = -INFINITY;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        const int i = 2 * I + a, j = 2 * J + bb;             // un-haloed output site
        float v0 = bias, v1 = bias;
#pragma unroll
        for (int di = 0; di < 3; ++di) {
#pragma unroll
          for (int dj = 0; dj < 3; ++dj) {
            const float* px = xin + ((c * TP + i + di) * XP + j + dj) * 2;   // halo shifts by +1, tap by -1
            const float x0 = px[0], x1 = px[1];
            const float k0 = w1[((di * 3 + dj) * 2 + 0) * F + f], k1 = w1[((di * 3 + dj) * 2 + 1) * F + f];
            v0 += x0 * k0;                 // output depth 0 sees (mu=0, mu=1); one multiply-add each, in this order,
            v0 += x1 * k1;                 // in EVERY implementation of conv1 (forward, reverse pass, K1c)
            v1 += x1 * k0;                 // output depth 1 sees (mu=1, pad)
          }
        }
        m = fmaxf(m, fmaxf(v0, v1));
      }
    }
    p1[((c * T2P + I) * X2P + J) * F + f] = fmaxf(m, 0.f);
  }
  __syncthreads();

  // ---- conv2 (2,2,[2]) + relu + pool (2,2,[1]): output (c, I2, J2, g), g on the lane
  const int n2 = nrow * T4 * X4 * F2;
  for (int idx = tid; idx < n2; idx += kConvThreads) {
    const int g = idx % F2;
    int r = idx / F2;
    const int J2 = r % X4;
    r /= X4;
    const int I2 = r % T4, c = r / T4;
    const float bias = b2[g];
    // 2x2 outputs of the pooling window share a 3x3 patch of the pooled conv1 map: per 4 input channels,
    // 9 ds_read_b128 (patch, broadcast across the g lanes) + 16 weight reads feed 64 FMAs
    float acc[2][2] = {{bias, bias}, {bias, bias}};
    const float* pbase = p1 + ((c * T2P + 2 * I2) * X2P + 2 * J2) * F;
    for (int ch4 = 0; ch4 < F; ch4 += 4) {
      f32x4 w[3][3];
#pragma unroll
      for (int wi = 0; wi < 3; ++wi)
#pragma unroll
        for (int wj = 0; wj < 3; ++wj)
          w[wi][wj] = *reinterpret_cast<const f32x4*>(pbase + (wi * X2P + wj) * F + ch4);
#pragma unroll
      for (int di = 0; di < 2; ++di)
#pragma unroll
        for (int dj = 0; dj < 2; ++dj)
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            const float kw = w2[((di * 2 + dj) * F + ch4 + cc) * F2 + g];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
              for (int bb = 0; bb < 2; ++bb) acc[a][bb] += w[a + di][bb + dj][cc] * kw;
          }
    }
    const float m = fmaxf(fmaxf(acc[0][0], acc[0][1]), fmaxf(acc[1][0], acc[1][1]));
    out[(row0 + c) * p.ldo + (I2 * X4 + J2) * F2 + g] = fmaxf(m, 0.f);
  }
}

int conv3d_nflat(int T, int X, int F) { return (T / 4) * (X / 4) * 2 * F; }

// CUs of the current device; asked once per device
static int conv_cu_count() {
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

int launch_conv3d_front(ConvFrontArgs& a, hipStream_t stream) {
  L2HMC_REQUIRE(a.T % 4 == 0 && a.X % 4 == 0 && a.F > 0 && a.F % 4 == 0,
                "conv3d front-end: T=%d X=%d F=%d must be multiples of 4", a.T, a.X, a.F);
  if (a.ldi == 0) a.ldi = 2 * a.T * a.X;
  // compile-time instances (pooled map with channel stride F + 4, no weights in LDS, 16-byte staging loads): the
  // BASELINE shapes with 16-byte-aligned operands; everything else takes the run-time form
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  bool inst = ((a.F == 8 && a.T == 8 && a.X == 8) || (a.F == 16 && a.T == 16 && a.X == 16)) && a.ldi % 4 == 0 &&
              a.ldo % 4 == 0 && (!a.cmask_f || (al16(a.cmask_f) && al16(a.cmask_b)));
  for (int w = 0; w < 2 && inst; ++w)
    if (a.only == 0 || a.only == w + 1)
      inst = al16(a.in[w]) && al16(a.out[w]) && al16(a.w1[w]) && al16(a.b1[w]) && al16(a.w2[w]) && al16(a.b2[w]);
  const int per_chain = (a.T / 2) * (a.X / 2) * a.F;
  a.cpw = per_chain >= 1024 ? 1 : (1024 / per_chain > 8 ? 8 : 1024 / per_chain);   // amortise filter loads / barriers
  if (per_chain >= 1024 && per_chain < 4096) a.cpw = 2;      // 16 x 16, F = 16: 32.7 -> 31.0 us per launch (two chains share the filter load)
  const size_t maps = (size_t)(a.T + 2) * (a.X + 2) * 2 + (size_t)(a.T / 2 + 1) * (a.X / 2 + 1) * (inst ? a.F + 4 : a.F);
  if (inst) {
    // the maps are all of a workgroup's LDS: as many chains per workgroup as keep (about) four workgroups per CU
    // in ONE round -- the batch over 4 x CUs workgroups per input, at most what 40 KB hold, at least the default
    const int64_t want = ceil_div(a.rows * (a.only ? 1 : 2), (int64_t)4 * conv_cu_count());
    const int64_t fit = (int64_t)(40 * 1024 / sizeof(float) / maps);
    a.cpw = (int)std::max<int64_t>(a.cpw, std::min<int64_t>(want, fit));
    const int cg = a.T == 8 ? 4 : 1;            // (kernel: CG, chains per pipeline group)
    a.cpw = (a.cpw + cg - 1) / cg * cg;
  }
  const size_t lds = sizeof(float) * (inst ? (size_t)a.cpw * maps
                                           : (size_t)18 * a.F + a.F + (size_t)8 * a.F * a.F + 2 * a.F + (size_t)a.cpw * maps);
  L2HMC_REQUIRE(lds <= 160 * 1024, "conv3d front-end: %zu B of LDS needed", lds);
  L2HMC_REQUIRE(a.only >= 0 && a.only <= 2, "conv3d front-end: bad input selector %d", a.only);
  const dim3 grid((unsigned)ceil_div(a.rows, a.cpw), a.only ? 1 : 2);
  static DeviceOnce attr_once;
  if (attr_once.pending()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_front_kernel<8, 8>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_front_kernel<16, 16>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_front_kernel<0, 0>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_once.done();
  }
#ifdef L2HMC_STAMPS
  a.stamps = g_stamp_cls == 7 ? g_stamp_buf : nullptr;
#endif
  prof_before(kProfConvFront, stream);
  if (inst && a.F == 8)
    hipLaunchKernelGGL((conv3d_front_kernel<8, 8>), grid, dim3(kConvThreads), lds, stream, a);
  else if (inst)
    hipLaunchKernelGGL((conv3d_front_kernel<16, 16>), grid, dim3(kConvThreads), lds, stream, a);
  else
    hipLaunchKernelGGL((conv3d_front_kernel<0, 0>), grid, dim3(kConvThreads), lds, stream, a);
  prof_after(kProfConvFront, stream);
  L2HMC_CHECK_LAUNCH("conv3d_front");
  return L2HMC_OK;
}

// =====================================================================================
// Backward of the front-end for the training path (train.hip): given d loss / d features of one
// network call, the gradient with respect to the raw inputs and to the filters / biases.
// The forward is recomputed from the taped inputs (a chain is a few hundred floats; keeping the
// pooling winners of every call in HBM would cost more than recomputing them in LDS).
// Max-pooling routes each pooled cell's gradient to its single winner, so everything downstream is
// sparse: the kernel records the winner of every pooled cell (arg1: one of 2x2x2 conv1 outputs,
// arg2: one of 2x2 conv2 outputs; 255 = the relu killed it) and every later phase is a GATHER in
// which a thread owns its output element -- no atomics, fixed summation order.
// Filter gradients are accumulated (+=) into this workgroup's own slot of `part`
// ([workgroup][input][18F | F | 16F^2 (Keras layout, dd = 1 rows stay 0) | 2F]); the same workgroup
// sees the same chains in every call, and the slots are summed in order afterwards.
// =====================================================================================
// (latency-bound phases want resident waves: with the tap loop of phase 4 unrolled over one axis only the (8, 8)
// instance needs 80 registers -- fully unrolled it took 242, i.e. two workgroups per CU)
// NTHR: threads per workgroup.  The phases are chains of dependent LDS reads; at 16 x 16 / F = 16 a chain's maps take
// 59 KB (two workgroups per CU), so that instance runs 512 threads per workgroup to have four waves per SIMD in flight.
template <int FT, int LT, int NTHR = kConvThreads>
__global__ __launch_bounds__(NTHR) __attribute__((amdgpu_waves_per_eu(4, 8))) void conv3d_front_bwd_kernel(
    ConvBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int which = blockIdx.y;
  const int T = LT ? LT : p.T, X = LT ? LT : p.X, F = FT ? FT : p.F, F2 = 2 * F;
  const int D = 2 * T * X;
  const int TP = T + 2, XP = X + 2;
  const int T2 = T / 2, X2 = X / 2, T2P = T2 + 1, X2P = X2 + 1;
  const int T4 = T / 4, X4 = X / 4;
  const int cpw = p.cpw;
  float* w1 = lds;                                 // [3][3][2][F]
  float* b1 = w1 + 18 * F;
  float* w2 = b1 + F;                              // [2][2][F][2F]   (dd = 0 slice)
  float* b2 = w2 + 4 * F * F2;
  L2HMC_STAMP(0);
  float* xin = b2 + F2;                            // [cpw][TP][XP][2]
  float* p1 = xin + cpw * TP * XP * 2;             // [cpw][T2P][X2P][F]
  float* dpre1 = p1 + cpw * T2P * X2P * F;         // [cpw][T2][X2][F]   gradient at the winning conv1 output
  // dense gradient maps over the PRE-pooling outputs (the winner of a pooling cell carries its gradient, the other
  // positions hold 0), with a zero halo so that the transposed convolutions below need no bounds checks:
  // (the two depths of a position share one pooling cell, so at most one of them is the winner: G1 keeps ONE value
  //  per (position, filter) and dep1 says which depth it belongs to -- half the LDS, more resident workgroups)
  float* G1 = dpre1 + cpw * T2 * X2 * F;           // [cpw][TP][XP][F]    conv1 output (i, j) at (i + 1, j + 1)
  const int nent = 18 * F + F + 4 * F * F2 + F2;   // compact gradient entries: w1 | b1 | w2 (dd = 0 slice) | b2
  // (compile-time instances pad a G2 position to 2F + 4 floats: the matrix-pipe form of phase 3 reads it with 16-byte
  //  loads at a stride of one position per lane, and 36 / 20 floats spread sixteen lanes over all 64 banks)
  const int G2S = FT > 0 ? F2 + 4 : F2;
  float* G2 = G1 + cpw * max(TP * XP * F, nent);   // [cpw][T2P][X2P][G2S] conv2 output (i2, j2) at (i2 + 1, j2 + 1)
  float* pw = G1;                                  // [cpw][nent] per-chain contributions: phase 5, when G1 is dead
  unsigned char* arg1 = reinterpret_cast<unsigned char*>(G2 + cpw * T2P * X2P * G2S);  // [cpw][T2][X2][F] winners
  unsigned char* dep1 = arg1 + ((cpw * T2 * X2 * F + 15) & ~15);                      // [cpw][TP][XP][F] depth of G1
  const int tid = threadIdx.x;
  const int64_t row0 = (int64_t)blockIdx.x * cpw;
  const int nrow = (int)min((int64_t)cpw, p.rows - row0);

  const float* gw1 = p.w1[which];
  const float* gw2 = p.w2[which];
  for (int i = tid; i < 18 * F; i += NTHR) w1[i] = gw1[i];
  for (int i = tid; i < F; i += NTHR) b1[i] = p.b1[which][i];
  for (int i = tid; i < 4 * F * F2; i += NTHR) {
    const int g = i % F2, c = (i / F2) % F, tap = i / (F2 * F);
    w2[i] = gw2[((size_t)(tap * 2 + 0) * F + c) * F2 + g];
  }
  for (int i = tid; i < F2; i += NTHR) b2[i] = p.b2[which][i];
  {
    // xin | p1 | dpre1 | G1 | G2 are contiguous and each a multiple of four floats: one 16-byte sweep clears the halos
    const int nz4 = (int)((G2 + cpw * T2P * X2P * G2S) - xin) / 4;
    for (int i = tid; i < nz4; i += NTHR) reinterpret_cast<f32x4*>(xin)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // this workgroup's gradient slot is read now and written at the very end: its round trip hides under the phases
  const size_t psize = (size_t)18 * F + F + (size_t)16 * F * F + F2;
  float* part = p.part + ((size_t)blockIdx.x * 2 + which) * psize;
  auto slot_of = [&](int ent) -> size_t {                // slot layout: Keras kernels, dd = 1 rows of w2 stay 0
    if (ent < 19 * F) return (size_t)ent;
    if (ent < 19 * F + 4 * F * F2) {
      const int e2 = ent - 19 * F;
      const int g = e2 % F2, ch = (e2 / F2) % F, tap = e2 / (F2 * F);
      return (size_t)19 * F + ((size_t)(tap * 2 + 0) * F + ch) * F2 + g;
    }
    return (size_t)19 * F + (size_t)16 * F * F + (ent - 19 * F - 4 * F * F2);
  };
  // entries per thread held across the kernel
  constexpr int kSlotRegs = FT > 0 ? (18 * FT + FT + 8 * FT * FT + 2 * FT + NTHR - 1) / NTHR : 12;
  float slot_old[kSlotRegs];
#pragma unroll
  for (int k = 0; k < kSlotRegs; ++k) {
    const int ent = tid + k * NTHR;
    slot_old[k] = ent < nent ? part[slot_of(ent)] : 0.f;
  }
  __syncthreads();
  const float* in = p.in + which * D;
  for (int i = tid; i < nrow * D; i += NTHR) {
    const int c = i / D, e = i - c * D;
    const int site = e >> 1, mu = e & 1;
    const int ii = site / X, jj = site - ii * X;
    xin[((c * TP + ii + 1) * XP + jj + 1) * 2 + mu] = in[(row0 + c) * p.ldi + e];
  }
  __syncthreads();

  L2HMC_STAMP(1);
  // ---- phase 1: conv1 + pool1, remember the winner of each pooled cell
  const int n1 = nrow * T2 * X2 * F;
  for (int idx = tid; idx < n1; idx += NTHR) {
    const int f = idx % F;
    int r = idx / F;
    const int J = r % X2;
    r /= X2;
    const int I = r % T2, c = r / T2;
    const float bias = b1[f];
    float m = -INFINITY;
    int code = 255;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        const int i = 2 * I + a, j = 2 * J + bb;
        float v0 = bias, v1 = bias;
#pragma unroll
        for (int di = 0; di < 3; ++di) {
#pragma unroll
          for (int dj = 0; dj < 3; ++dj) {
            const float* px = xin + ((c * TP + i + di) * XP + j + dj) * 2;
            const float x0 = px[0], x1 = px[1];
            const float k0 = w1[((di * 3 + dj) * 2 + 0) * F + f], k1 = w1[((di * 3 + dj) * 2 + 1) * F + f];
            v0 += x0 * k0;                 // (the forward kernel's order: the pooling winners are ITS winners)
            v0 += x1 * k1;
            v1 += x1 * k0;
          }
        }
        if (v0 > m) { m = v0; code = (a * 2 + bb) * 2; }
        if (v1 > m) { m = v1; code = (a * 2 + bb) * 2 + 1; }
      }
    }
    p1[((c * T2P + I) * X2P + J) * F + f] = fmaxf(m, 0.f);
    arg1[idx] = (unsigned char)(m > 0.f ? code : 255);
  }
  __syncthreads();

  L2HMC_STAMP(2);
  // ---- phase 2: conv2 + pool2; the surviving feature's gradient goes to its winner in G2, zeros to the other three
  const int n2 = nrow * T4 * X4 * F2;
  for (int idx = tid; idx < n2; idx += NTHR) {
    const int g = idx % F2;
    int r = idx / F2;
    const int J2 = r % X4;
    r /= X4;
    const int I2 = r % T4, c = r / T4;
    const float bias = b2[g];
    float acc[2][2] = {{bias, bias}, {bias, bias}};
    const float* pbase = p1 + ((c * T2P + 2 * I2) * X2P + 2 * J2) * F;
    for (int ch4 = 0; ch4 < F; ch4 += 4) {
      f32x4 w[3][3];
#pragma unroll
      for (int wi = 0; wi < 3; ++wi)
#pragma unroll
        for (int wj = 0; wj < 3; ++wj)
          w[wi][wj] = *reinterpret_cast<const f32x4*>(pbase + (wi * X2P + wj) * F + ch4);
#pragma unroll
      for (int di = 0; di < 2; ++di)
#pragma unroll
        for (int dj = 0; dj < 2; ++dj)
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            const float kw = w2[((di * 2 + dj) * F + ch4 + cc) * F2 + g];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
              for (int bb = 0; bb < 2; ++bb) acc[a][bb] += w[a + di][bb + dj][cc] * kw;
          }
    }
    float m = -INFINITY;
    int code = 255;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int bb = 0; bb < 2; ++bb)
        if (acc[a][bb] > m) { m = acc[a][bb]; code = a * 2 + bb; }
    const float dv = m > 0.f ? p.dfeat[(row0 + c) * p.ldf + which * (T4 * X4 * F2) + (I2 * X4 + J2) * F2 + g] : 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int bb = 0; bb < 2; ++bb)
        G2[((c * T2P + 2 * I2 + a + 1) * X2P + 2 * J2 + bb + 1) * G2S + g] = (code == a * 2 + bb) ? dv : 0.f;
  }
  __syncthreads();

  L2HMC_STAMP(3);
  // ---- phase 3: transposed conv2 (dense, no winner tests): gradient of the pooled conv1 map, gated by its own relu /
  // winner, then spread over the cell's eight pre-pooling outputs in G1.  conv2 output (i2, j2) reads p1(i2 + di,
  // j2 + dj), so p1 cell (I, J) collects output (I - di, J - dj) through tap (di, dj).
  // Compile-time instances run the product on the matrix pipe (round 4): per chain a [T2 X2 positions] x [4 taps x 2F
  // filters] x [F channels] product, v_mfma_f32_16x16x4_f32 on tiles of 16 positions x 16 channels, lane (q, r) reading
  // sixteen bytes of position r's G2 row (four consecutive filters = four k-steps) per tap and 16-filter group, the
  // filter fragments held in registers for all of a wave's tiles.  The VALU form below issued two 16-byte LDS reads per
  // four multiply-adds and was LDS-bandwidth-bound: 25 % of the kernel at 16 x 16 (profiles/r04_conv_bwd_stamps.txt).
  if constexpr (FT > 0) {
    constexpr int F_ = FT, F2_ = 2 * FT, X2_ = LT / 2, NP = (LT / 2) * (LT / 2), MT3 = NP / 16, NS = F2_ / 16;
    static_assert(NP % 16 == 0 && F_ <= 16, "transposed conv2 tiles");
    const int lane = tid & 63, wave = tid >> 6, q = lane >> 4, r = lane & 15;
    // (the filter fragments w2[tap][channel r][16 s + 4 q ...] are re-read per tap: held for all four taps they push
    //  the kernel past 128 registers, i.e. from two workgroups per CU to one -- measured: no faster than the VALU form)
    for (int item = wave; item < nrow * MT3; item += NTHR / 64) {
      const int c = item / MT3, mt = item - c * MT3;
      const int pr = mt * 16 + r, Ir = pr / X2_, Jr = pr - Ir * X2_;          // this lane's operand row (a p1 cell)
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tap = 0; tap < 4; ++tap) {
        const float* gq = G2 + ((c * T2P + Ir - (tap >> 1) + 1) * X2P + Jr - (tap & 1) + 1) * G2S + 4 * q;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const f32x4 a4 = *reinterpret_cast<const f32x4*>(gq + 16 * s);
          const f32x4 w4 = r < F_ ? *reinterpret_cast<const f32x4*>(w2 + (tap * F_ + r) * F2_ + 16 * s + 4 * q)
                                  : f32x4{0.f, 0.f, 0.f, 0.f};          // zero columns beyond the F channels
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e], w4[e], acc, 0, 0, 0);
        }
      }
      // C layout: row 4 q + e = position, column r = channel
      if (r < F_) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int pp = mt * 16 + 4 * q + e, I = pp / X2_, J = pp - I * X2_;
          const int idx = ((c * T2 + I) * X2 + J) * F + r;
          const int code = arg1[idx];
          const float dp = code != 255 ? acc[e] : 0.f;
          dpre1[idx] = dp;
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
              const int pos = ((c * TP + 2 * I + a + 1) * XP + 2 * J + bb + 1) * F + r;
              G1[pos] = ((code >> 1) == a * 2 + bb) ? dp : 0.f;
              dep1[pos] = (unsigned char)(code & 1);
            }
        }
      }
    }
  }
  for (int idx = tid; idx < (FT > 0 ? 0 : n1); idx += NTHR) {
    const int ch = idx % F;
    int r = idx / F;
    const int J = r % X2;
    r /= X2;
    const int I = r % T2, c = r / T2;
    const int code = arg1[idx];
    float dp = 0.f;
    if (code != 255) {
      f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int di = 0; di < 2; ++di)
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) {
          const float* gq = G2 + ((c * T2P + I - di + 1) * X2P + J - dj + 1) * G2S;
          const float* kw = w2 + ((di * 2 + dj) * F + ch) * F2;
          for (int g4 = 0; g4 < F2; g4 += 4)
            s4 += *reinterpret_cast<const f32x4*>(gq + g4) * *reinterpret_cast<const f32x4*>(kw + g4);
        }
      dp = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    }
    dpre1[idx] = dp;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        const int pos = ((c * TP + 2 * I + a + 1) * XP + 2 * J + bb + 1) * F + ch;
        G1[pos] = ((code >> 1) == a * 2 + bb) ? dp : 0.f;       // (code 255 >> 1 matches nothing; dp is 0 then anyway)
        dep1[pos] = (unsigned char)(code & 1);
      }
  }
  __syncthreads();

  L2HMC_STAMP(4);
  // ---- phase 4: transposed conv1 (dense): gradient of the raw input, both link directions of a site per thread.
  // conv1 output (i, j) reads x(i + di - 1, j + dj - 1) (zero padding: no periodic wrap, as the forward); depth 0
  // sees (mu = 0, mu = 1) through (k0, k1) = w1[tap][dd = 0 | 1], depth 1 sees mu = 1 through k0.
  // A site's sum over the filters is shared by FS adjacent lanes (compile-time instances with one chain per workgroup
  // have fewer sites than threads: 256 sites on 512 threads at 16 x 16), each walking every FS-th group of four
  // filters; the lanes' partial sums meet in a fixed-order butterfly.
  constexpr int FS = (FT > 0 && (FT / 4) % 2 == 0 && LT * LT <= NTHR / 2) ? 2 : 1;
  for (int sidx0 = tid; sidx0 < nrow * T * X * FS; sidx0 += NTHR) {
    const int sidx = sidx0 / FS, fpart = sidx0 - sidx * FS;
    const int c = sidx / (T * X), site = sidx - c * (T * X);
    const int ip = site / X, jp = site - ip * X;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int di = 0; di < 3; ++di)
#pragma unroll
      for (int dj = 0; dj < 3; ++dj) {
        const int pos = ((c * TP + ip - di + 2) * XP + jp - dj + 2) * F;                // output (ip - di + 1, jp - dj + 1)
        const float* k = w1 + (di * 3 + dj) * 2 * F;
        for (int f4 = 4 * fpart; f4 < F; f4 += 4 * FS) {
          const f32x4 gv = *reinterpret_cast<const f32x4*>(G1 + pos + f4);
          const unsigned dm = *reinterpret_cast<const unsigned*>(dep1 + pos + f4);        // four depth bytes
          const f32x4 k0 = *reinterpret_cast<const f32x4*>(k + f4), k1 = *reinterpret_cast<const f32x4*>(k + F + f4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool d1 = (dm >> (8 * e)) & 1u;
            a0[e] += d1 ? 0.f : gv[e] * k0[e];                  // depth 0 -> mu = 0 through k0
            a1[e] += gv[e] * (d1 ? k0[e] : k1[e]);              // depth 0 -> mu = 1 through k1, depth 1 -> mu = 1 through k0
          }
        }
      }
    float s0 = (a0[0] + a0[1]) + (a0[2] + a0[3]), s1 = (a1[0] + a1[1]) + (a1[2] + a1[3]);
    if constexpr (FS == 2) {             // (nrow * T * X * FS is a multiple of 2: both lanes of a pair are here)
      s0 += __shfl_xor(s0, 1, 64);
      s1 += __shfl_xor(s1, 1, 64);
    }
    if (fpart == 0) {
      float* o = p.din + (row0 + c) * p.ldd + which * D + 2 * site;
      o[0] = s0;
      o[1] = s1;
    }
  }

  __syncthreads();                   // pw aliases G1
  L2HMC_STAMP(5);
  // ---- phase 5: filter / bias gradients, every sum in a fixed order.  Each work item writes ONE chain's
  // contribution to LDS; one owner thread per entry then adds the chains in order into the workgroup's slot.
  //   w1 / b1: item (chain, tap, f) walks the chain's pooling cells of filter f once and feeds both dd entries;
  //   w2 / b2: item (chain, tap, ch, 4 g) is a dense sum over the conv2 output positions (G2 holds zeros off-winner).
  // conv1 filter / bias gradient.  Every pooling cell is a chain of dependent LDS reads (winner byte -> the input pair
  // it points at), and an item that walks all T2 X2 cells of its (chain, tap, filter) is latency-bound with 9 F of the
  // workgroup's threads busy (36 % of the kernel at 16 x 16 after phases 3 and 5b went to the matrix pipe).  Where
  // the dead G1 map has room behind pw (the 16 x 16 instance), an item owns ONE ROW of cells -- T2 times the items,
  // chains of X2 instead of T2 X2 -- and leaves a partial sum; owners then add the T2 partials in row order.
  constexpr bool kRowItems = FT > 0 && (LT + 2) * (LT + 2) * FT - (19 * FT + 8 * FT * FT + 2 * FT) >= (LT / 2) * 19 * FT;
  if constexpr (kRowItems) {
    constexpr int F_ = FT, T2_ = LT / 2, X2_ = LT / 2;
    float* wpart = pw + cpw * nent;                          // [cpw][T2][19][F]: 18 filter rows + the bias row
    const int nit = nrow * T2_ * 9 * F_;
    for (int item = tid; item < nit; item += NTHR) {
      const int f = item % F_;
      int rr = item / F_;
      const int tap = rr % 9;
      rr /= 9;
      const int I = rr % T2_, c = rr / T2_;
      const int di = tap / 3, dj = tap - di * 3;
      float s0 = 0.f, s1 = 0.f, sb = 0.f;
#pragma unroll
      for (int J = 0; J < X2_; ++J) {
        const int cell = ((c * T2_ + I) * X2_ + J) * F_ + f;
        const int code = arg1[cell] & 7;                     // (a dead cell has dpre1 = 0 and decodes to a valid position)
        const int a = code >> 2, bb = (code >> 1) & 1, depth = code & 1;
        const float* px = xin + ((c * TP + 2 * I + a + di) * XP + 2 * J + bb + dj) * 2;
        const float dpv = dpre1[cell], x0 = px[0], x1 = px[1];
        s0 += dpv * (depth ? x1 : x0);
        s1 += depth ? 0.f : dpv * x1;
        sb += dpv;
      }
      float* o = wpart + (c * T2_ + I) * 19 * F_ + f;
      o[(tap * 2 + 0) * F_] = s0;
      o[(tap * 2 + 1) * F_] = s1;
      if (tap == 0) o[18 * F_] = sb;
    }
    __syncthreads();
    for (int ent = tid; ent < nrow * 19 * F_; ent += NTHR) {
      const int c = ent / (19 * F_), e = ent - c * 19 * F_;
      float sum = 0.f;
#pragma unroll
      for (int I = 0; I < T2_; ++I) sum += wpart[(c * T2_ + I) * 19 * F_ + e];
      pw[c * nent + e] = sum;                                // entries [0, 18 F): conv1 filter, [18 F, 19 F): its bias
    }
  }
  {
    const int nw1 = kRowItems ? 0 : nrow * 9 * F;
    for (int item = tid; item < nw1; item += NTHR) {
      const int f = item % F;
      int r = item / F;
      const int tap = r % 9, c = r / 9;
      const int di = tap / 3, dj = tap - di * 3;
      float s0 = 0.f, s1 = 0.f;
      // (one flat loop, unrolled: every cell is a chain of dependent LDS reads -- winner byte, then the input pair it
      //  points at -- and the cells are independent; eight in flight instead of one)
#pragma unroll 2
      for (int IJ = 0; IJ < T2 * X2; ++IJ) {
        const int I = IJ / X2, J = IJ - I * X2;
        // branch-free: a dead cell (code 255) has dpre1 = 0 and decodes to a valid position
        const int cell = (c * T2 * X2 + IJ) * F + f;
        const int code = arg1[cell] & 7;
        const int a = code >> 2, bb = (code >> 1) & 1, depth = code & 1;
        const float* px = xin + ((c * TP + 2 * I + a + di) * XP + 2 * J + bb + dj) * 2;
        const float dpv = dpre1[cell], x0 = px[0], x1 = px[1];
        s0 += dpv * (depth ? x1 : x0);           // dd = 0: depth 0 reads mu = 0, depth 1 reads mu = 1
        s1 += depth ? 0.f : dpv * x1;            // dd = 1: only depth 0 (mu = 1); depth 1 reads the padding
      }
      pw[c * nent + (tap * 2 + 0) * F + f] = s0;
      pw[c * nent + (tap * 2 + 1) * F + f] = s1;
    }
    for (int item = tid; item < (kRowItems ? 0 : nrow * F); item += NTHR) {
      const int f = item % F, c = item / F;
      float sb = 0.f;
      for (int cell = c * T2 * X2; cell < (c + 1) * T2 * X2; ++cell) sb += dpre1[cell * F + f];
      pw[c * nent + 18 * F + f] = sb;
    }
    // conv2 filter gradient: per chain and tap dW[ch][g] = sum over the conv2 output positions of
    // p1(pos + tap)[ch] * G2(pos)[g] -- a product with the POSITIONS as contraction index.  Compile-time instances
    // run it on the matrix pipe (round 4): v_mfma_f32_16x16x4_f32 with rows = (tap, channel) (one tap per 16-row
    // tile at F = 16, two at F = 8), columns = 16 output filters, k = four consecutive positions; both operands are
    // read as they lie in LDS (a lane's operand is one float: sixteen lanes read sixteen consecutive channels /
    // filters of one position) -- 48 four-byte LDS reads per 32 MFMAs where the VALU form below issues two reads
    // per four multiply-adds (it was 41 % of the kernel at 16 x 16 by in-kernel stamps, profiles/r04_conv_bwd_stamps.txt).
    if constexpr (FT > 0) {
      constexpr int F_ = FT, F2_ = 2 * FT, X2_ = LT / 2, NP = (LT / 2) * (LT / 2);
      constexpr int TPM = 16 / F_, MTL = 4 / TPM, NTN = F2_ / 16;
      static_assert(16 % F_ == 0 && F2_ % 16 == 0 && NP % 4 == 0 && X2_ % 4 == 0, "conv2 filter gradient tiles");
      const int lane = tid & 63, wave = tid >> 6, q = lane >> 4, r = lane & 15;
      for (int item = wave; item < nrow * MTL; item += NTHR / 64) {
        const int c = item / MTL, mt = item - c * MTL;
        const int tap = mt * TPM + r / F_, ch = r % F_;          // this lane's operand row
        const float* ap = p1 + ((c * T2P + (tap >> 1)) * X2P + (tap & 1)) * F_ + ch;
        const float* bp = G2 + ((c * T2P + 1) * X2P + 1) * G2S + r;
        f32x4 acc[NTN];
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int ks = 0; ks < NP / 4; ++ks) {
          const int k = 4 * ks + q, i2 = k / X2_, j2 = k - i2 * X2_;      // position k of the chain's conv2 output
          const float av = ap[(i2 * X2P + j2) * F_];
#pragma unroll
          for (int nt = 0; nt < NTN; ++nt)
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bp[(i2 * X2P + j2) * G2S + 16 * nt], acc[nt], 0, 0, 0);
        }
        // C layout: row 4 q + e, column r
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int mrow = 4 * q + e, tap_o = mt * TPM + mrow / F_, ch_o = mrow % F_;
#pragma unroll
          for (int nt = 0; nt < NTN; ++nt) pw[c * nent + 19 * F_ + (tap_o * F_ + ch_o) * F2_ + 16 * nt + r] = acc[nt][e];
        }
      }
    }
    const int g4n = F2 / 4;
    const int nw2 = FT > 0 ? 0 : nrow * 4 * F * g4n;
    for (int item = tid; item < nw2; item += NTHR) {
      const int g4 = (item % g4n) * 4;
      int r = item / g4n;
      const int ch = r % F;
      r /= F;
      const int tap = r % 4, c = r / 4;
      const int di = tap >> 1, dj = tap & 1;
      f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
      for (int i2 = 0; i2 < T2; ++i2)
        for (int j2 = 0; j2 < X2; ++j2)
          s4 += *reinterpret_cast<const f32x4*>(G2 + ((c * T2P + i2 + 1) * X2P + j2 + 1) * G2S + g4) *
                p1[((c * T2P + i2 + di) * X2P + j2 + dj) * F + ch];
      *reinterpret_cast<f32x4*>(pw + c * nent + 19 * F + (tap * F + ch) * F2 + g4) = s4;
    }
    for (int item = tid; item < nrow * F2; item += NTHR) {
      const int g = item % F2, c = item / F2;
      float sb = 0.f;
      for (int i2 = 0; i2 < T2; ++i2)
        for (int j2 = 0; j2 < X2; ++j2) sb += G2[((c * T2P + i2 + 1) * X2P + j2 + 1) * G2S + g];
      pw[c * nent + 19 * F + 4 * F * F2 + g] = sb;
    }
  }
  __syncthreads();
  L2HMC_STAMP(6);
#pragma unroll
  for (int k = 0; k < kSlotRegs; ++k) {
    const int ent = tid + k * NTHR;
    if (ent < nent) {
      float sum = 0.f;
      for (int c = 0; c < nrow; ++c) sum += pw[c * nent + ent];
      part[slot_of(ent)] = slot_old[k] + sum;
    }
  }
  for (int ent = tid + kSlotRegs * NTHR; ent < nent; ent += NTHR) {     // wider filter sets
    float sum = 0.f;
    for (int c = 0; c < nrow; ++c) sum += pw[c * nent + ent];
    part[slot_of(ent)] += sum;
  }
  L2HMC_STAMP(7);
}

size_t conv3d_bwd_part_floats(int F) { return (size_t)18 * F + F + (size_t)16 * F * F + 2 * F; }
// chains per workgroup of the BACKWARD kernel: enough work items (~2048 pooled conv1 cells) to amortise the
// filter loads, the barriers and the read-modify-write of the workgroup's gradient slot
int conv3d_cpw(int T, int X, int F) {
  const int per_chain = (T / 2) * (X / 2) * F;
  const int c = 512 / per_chain;      // (measured at the 8x8 / F = 8 shape: 256 cells 5.08 ms per training step, 512 5.02, 1024 5.37, 2048 6.47)
  return c < 1 ? 1 : (c > 16 ? 16 : c);
}

int launch_conv3d_front_bwd(ConvBwdArgs& a, hipStream_t stream) {
  L2HMC_REQUIRE(a.T % 4 == 0 && a.X % 4 == 0 && a.F > 0 && a.F % 4 == 0,
                "conv3d front-end backward: T=%d X=%d F=%d must be multiples of 4", a.T, a.X, a.F);
  a.cpw = conv3d_cpw(a.T, a.X, a.F);
#ifdef L2HMC_STAMPS
  a.stamps = g_stamp_cls == 6 ? g_stamp_buf : nullptr;
#endif
  const size_t F = a.F, F2 = 2 * F, cpw = a.cpw, TP = a.T + 2, XP = a.X + 2, T2 = a.T / 2, X2 = a.X / 2;
  const size_t cells1 = cpw * T2 * X2 * F, nent = 18 * F + F + 4 * F * F2 + F2;
  const bool fixed = (a.F == 8 && a.T == 8 && a.X == 8) || (a.F == 16 && a.T == 16 && a.X == 16);   // compile-time instances
  const size_t g2s = fixed ? F2 + 4 : F2;                                      // (kernel: G2S)
  const size_t lds = sizeof(float) * (18 * F + F + 4 * F * F2 + F2 +           // filters
                                      cpw * TP * XP * 2 +                       // xin
                                      cpw * (T2 + 1) * (X2 + 1) * F +           // p1
                                      cells1 +                                  // dpre1
                                      cpw * (TP * XP * F > nent ? TP * XP * F : nent) +   // G1, later pw
                                      cpw * (T2 + 1) * (X2 + 1) * g2s) +        // G2
                     align_up(cells1, 16) + align_up(cpw * TP * XP * F, 16);    // arg1, dep1
  L2HMC_REQUIRE(lds <= 160 * 1024, "conv3d front-end backward: %zu B of LDS needed", lds);
  const dim3 grid((unsigned)ceil_div(a.rows, a.cpw), 2);
  static DeviceOnce attr_once;
  if (attr_once.pending()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_front_bwd_kernel<8, 8>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_front_bwd_kernel<16, 16, 512>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_front_bwd_kernel<0, 0>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_once.done();
  }
  if (a.F == 8 && a.T == 8 && a.X == 8)
    hipLaunchKernelGGL((conv3d_front_bwd_kernel<8, 8>), grid, dim3(kConvThreads), lds, stream, a);
  else if (a.F == 16 && a.T == 16 && a.X == 16)
    hipLaunchKernelGGL((conv3d_front_bwd_kernel<16, 16, 512>), grid, dim3(512), lds, stream, a);
  else
    hipLaunchKernelGGL((conv3d_front_bwd_kernel<0, 0>), grid, dim3(kConvThreads), lds, stream, a);
  L2HMC_CHECK_LAUNCH("conv3d_front_bwd");
  return L2HMC_OK;
}

}  // namespace l2hmc
