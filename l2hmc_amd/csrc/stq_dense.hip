// Dense scale/translate/transform network on fp32 MFMA (gfx950), with the
// leapfrog sub-update fused into the heads' epilogue.
//
// Replaces, per network call, the 6 matmuls + ~10 element-wise TF kernels of
//   l2hmc/network/generic_net.py:129-146 (and the dense trunk of
//   network/conv_net.py:264-280, utils/network.py:89-114)
// and, in the fused modes, the ~12 element-wise ops + reduce_sum of
//   l2hmc/dynamics/gauge_dynamics.py:486-508, :511-534, :537-561, :565-590.
//
// Three launches per network call, every operand k-contiguous ("NT" GEMM):
//   L1    h1 = relu([a | b*mask] . W1^T + b1 + t.Wt)     K = Ka + Kb
//   L2    h2 = relu(h1 . Wh^T + bh)                       K = H
//   heads (S,T,Q) = h2 . Whd^T + bhd  -> tanh/exp(coeff) -> v or x update
//         + per-row log-det partial sums (wave shuffles, fixed order).
// S, T, Q never touch HBM in the fused modes; h1/h2 are [rows][H] scratch that
// stays L2/MALL-resident at the benchmark sizes.
//
// Arithmetic is exact fp32: v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 are
// bitwise a k-ordered fmaf chain (no TF32/xf32 on gfx950), which is what the
// 1e-5 parity bar needs.  Tiles are sized for 64-wide waves: 4 waves per
// workgroup, one per SIMD; LDS rows are padded so the 16-byte fragment reads
// are bank-conflict free (stride 36 floats for the 32-row fragments, 40 for the
// 16-row ones).  Tile ids are remapped so tiles sharing activation rows land on
// one XCD (shared L2).
#include "stq_dense.h"

namespace l2hmc {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BK = 32;            // k-tile granularity every width must be a multiple of
constexpr int kGemmThreads = 256;  // 4 waves

// =====================================================================
// L1 / L2:  out = relu(A . Wt^T + bias [+ t-term])
// =====================================================================


// bias and time term of the first layer, in ONE place: gemm_relu_kernel<., 1> and l1_finish_kernel must round alike
__device__ __forceinline__ float l1_preact(float acc, float bj, float tc, float ts, float w0, float w1) {
  float h = acc + bj;
  h += tc * w0 + ts * w1;
  return h;
}

// KIND 1: first layer (two k-contiguous sources, optional column mask, time term)
// KIND 2: hidden layer (single source, plain bias)
// KIND 3: backward-data through a relu layer: out = (A . Wt^T) where gate > 0, else 0   (no bias)
// KIND 4: plain product out = A . Wt^T
// RAGGED: any K, K1 and row strides (widths that are not multiples of 32, e.g. a 6x6 lattice: x_dim 72, H 288;
// or rows that are not 16-byte aligned, x_dim 50): every staged element is loaded on its own under a bounds
// check and the k-loop runs over ceil(K / BK) zero-padded tiles -- the same arithmetic (zeros add nothing), a
// slower load path that only odd shapes take.
template <int BM, int KIND, int BK, int BN = 128, bool RAGGED = false>
__global__ __launch_bounds__(kGemmThreads) __attribute__((amdgpu_waves_per_eu(2))) void gemm_relu_kernel(GemmReluArgs p) {
  constexpr int CPR = BK / 4;              // 16-byte chunks per staged row
  // Aligned shapes (DMA): the tiles go from global memory STRAIGHT into LDS (buffer_load ... lds, 1 KiB per wave
  // instruction, no staging registers, no ds_write; round 4 -- the no-staging experiment of
  // profiles/r04_cfg4_experiments.txt put what staging costs at 7-12 % of the kernel).  Such a load fills 1 KiB of LDS
  // in lane order, so rows cannot be padded; instead chunk c of row R sits at position c ^ swz(R), the lane asking for
  // the global chunk that belongs at ITS position.  swz is even (a fragment's two half-wave chunks 2 kq, 2 kq + 1 stay
  // neighbours) and spreads the 32 rows x 2 halves of a fragment read evenly over the 16 bank groups.
  // Otherwise: staged through registers into padded rows (36 / 68 floats: an odd number of 16-B slots).
  // k-tiles of 32 (the 128-row instances of cfg 5) keep the register path: measured 754 against 748 ms per cfg-5 step.
  constexpr bool DMA = !RAGGED && BK == 64;
  constexpr int LDK = DMA ? BK : BK + 4;
  constexpr int RP = 64 / BK;              // rows per 256 bytes of LDS
  auto swz = [](int R) { return DMA ? 2 * ((R / RP) & (CPR / 2 - 1)) : 0; };
  constexpr int MT = BM / 64;              // 32x32 tiles per wave along M (wave grid 2 x 2)
  constexpr int WN = BN / 2;               // columns per wave: 64 (BN = 128) or 32 (BN = 64)
  constexpr int NT = WN / 32;
  constexpr int A_CH = BM * (BK / 4) / kGemmThreads;
  constexpr int B_CH = BN * (BK / 4) / kGemmThreads;
  constexpr int STAGE = (BM + BN) * LDK;   // one A|B buffer pair
  constexpr int LDC = BN + 4;              // row stride of the output tile staged by the epilogue (floats)
  constexpr int LDS_FLOATS = (RAGGED || 2 * STAGE >= BM * LDC) ? 2 * STAGE : BM * LDC;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

  L2HMC_STAMP_REAL(4);
  L2HMC_STAMP(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int half = lane >> 5, r = lane & 31;
  const int wv = __builtin_amdgcn_readfirstlane(wave);     // provably uniform (LDS addresses of the tile loads)

  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int mt_id = tile / p.ntiles, nt_id = tile - mt_id * p.ntiles;
  const int64_t m0 = (int64_t)mt_id * BM;
  const int n0 = nt_id * BN;

  // --- staging coordinates (fixed per thread)
  int a_row[A_CH], a_kc[A_CH];
  bool a_ok[A_CH];
  int a_dir[A_CH];
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    const int c = tid + i * kGemmThreads;
    a_row[i] = c / CPR;
    a_kc[i] = ((c % CPR) ^ swz(a_row[i])) * 4;          // the k-chunk that belongs at LDS position c
    a_ok[i] = (m0 + a_row[i]) < p.rows;
    a_dir[i] = (KIND == 1 && p.dir && a_ok[i]) ? p.dir[m0 + a_row[i]] : 0;
  }
  int b_row[B_CH], b_kc[B_CH];
  bool b_ok[B_CH];
#pragma unroll
  for (int i = 0; i < B_CH; ++i) {
    const int c = tid + i * kGemmThreads;
    b_row[i] = c / CPR;
    b_kc[i] = ((c % CPR) ^ swz(b_row[i])) * 4;
    b_ok[i] = (n0 + b_row[i]) < p.N;
  }

  // Tile loads of the aligned path are UNCONDITIONAL and nothing consumes them before store_tile (after the k-tile's
  // MFMAs): rows / columns past the end are clamped to the last valid one (their products land in accumulator rows /
  // columns that are never stored), the source of a first-layer tile is a pointer select, and the column mask of the
  // second input is multiplied in at store time.  (The earlier form -- loads under `if (k0 < K1) ... else ...` with
  // the mask multiply right behind them -- made the compiler drain the load queue between the four A loads of
  // every second-input tile: the first layer ran 15 % behind the same-shape hidden layer.)
  // They are BUFFER loads (common.h: buf_load16): one scalar resource per source, based at this tile's first row, a
  // scalar k offset, and ONE 32-bit offset per lane and chunk -- next to MFMAs a load with a 64-bit per-lane address
  // costs the matrix pipe more.
  f32x4 ra[A_CH], rb[B_CH];
  [[maybe_unused]] f32x4 rm[A_CH];
  [[maybe_unused]] bool tile_masked = false;
  [[maybe_unused]] unsigned a_off1[A_CH], a_off2[A_CH], b_off[B_CH];      // bytes from the tile's base (launcher: < 2^31)
  [[maybe_unused]] WSection rs_a1, rs_a2, rs_b;
  [[maybe_unused]] WDesc ds_a1, ds_a2, ds_b;
  [[maybe_unused]] const unsigned lds0 = lds_byte_address(lds);
  if constexpr (!RAGGED) {
    rs_a1 = wsection(p.A1 + m0 * p.lda1);
    rs_a2 = wsection((KIND != 2 && p.A2) ? p.A2 + m0 * p.lda2 : p.A1 + m0 * p.lda1);
    rs_b = wsection(p.Wt + (int64_t)n0 * p.K);
    ds_a1 = wdesc(p.A1 + m0 * p.lda1);
    ds_a2 = wdesc((KIND != 2 && p.A2) ? p.A2 + m0 * p.lda2 : p.A1 + m0 * p.lda1);
    ds_b = wdesc(p.Wt + (int64_t)n0 * p.K);
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const unsigned lr = (unsigned)((a_ok[i] ? m0 + a_row[i] : p.rows - 1) - m0);     // clamped row, tile-local
      a_off1[i] = (lr * (unsigned)p.lda1 + (unsigned)a_kc[i]) * 4u;
      a_off2[i] = (lr * (unsigned)(KIND != 2 ? p.lda2 : p.lda1) + (unsigned)a_kc[i]) * 4u;
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      const unsigned lc = (unsigned)((b_ok[i] ? n0 + b_row[i] : p.N - 1) - n0);
      b_off[i] = (lc * (unsigned)p.K + (unsigned)b_kc[i]) * 4u;
    }
  }
  // buf: the LDS buffer the tile is for (DMA: the loads go there at once; the barrier at the end of the previous
  // k-tile is what freed it)
  auto load_tile = [&](int kt, [[maybe_unused]] int buf) {
    const int k0 = kt * BK;
    if constexpr (RAGGED) {
#pragma unroll
      for (int i = 0; i < A_CH; ++i) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (a_ok[i]) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int k = k0 + a_kc[i] + j;
            if (k < p.K1) {
              v[j] = p.A1[(m0 + a_row[i]) * p.lda1 + k];
            } else if (k < p.K) {
              float t = p.A2[(m0 + a_row[i]) * p.lda2 + (k - p.K1)];
              if (p.cmask_f) t *= (a_dir[i] ? p.cmask_b : p.cmask_f)[k - p.K1];
              v[j] = t;
            }
          }
        }
        ra[i] = v;
      }
#pragma unroll
      for (int i = 0; i < B_CH; ++i) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (b_ok[i]) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int k = k0 + b_kc[i] + j;
            if (k < p.K) v[j] = p.Wt[(int64_t)(n0 + b_row[i]) * p.K + k];
          }
        }
        rb[i] = v;
      }
    } else {
      const bool second = KIND != 2 && k0 >= p.K1;            // uniform
      const int kk0 = second ? k0 - p.K1 : k0;
      [[maybe_unused]] const unsigned la = lds0 + (unsigned)(buf * STAGE) * 4u + (unsigned)wv * 1024u;
      [[maybe_unused]] const unsigned lb = la + (unsigned)(BM * LDK) * 4u;
      if constexpr (KIND == 1) tile_masked = second && p.cmask_f != nullptr;
      if constexpr (!DMA) {
        if (second) {
#pragma unroll
          for (int i = 0; i < A_CH; ++i) ra[i] = buf_load16(rs_a2, a_off2[i], (unsigned)kk0 * 4u);
        } else {
#pragma unroll
          for (int i = 0; i < A_CH; ++i) ra[i] = buf_load16(rs_a1, a_off1[i], (unsigned)kk0 * 4u);
        }
        if constexpr (KIND == 1) {
          if (tile_masked) {
#pragma unroll
            for (int i = 0; i < A_CH; ++i)
              rm[i] = *reinterpret_cast<const f32x4*>((a_dir[i] ? p.cmask_b : p.cmask_f) + kk0 + a_kc[i]);
          }
        }
#pragma unroll
        for (int i = 0; i < B_CH; ++i) rb[i] = buf_load16(rs_b, b_off[i], (unsigned)k0 * 4u);
        return;
      }
      if (KIND == 1 && tile_masked) {
        // the second input's column mask is multiplied in on the way: this tile's A goes through registers
#pragma unroll
        for (int i = 0; i < A_CH; ++i) ra[i] = buf_load16(rs_a2, a_off2[i], (unsigned)kk0 * 4u);
#pragma unroll
        for (int i = 0; i < A_CH; ++i)
          rm[i] = *reinterpret_cast<const f32x4*>((a_dir[i] ? p.cmask_b : p.cmask_f) + kk0 + a_kc[i]);
      } else if (second) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) lds_dma16(ds_a2, la + (unsigned)i * 4096u, a_off2[i], (unsigned)kk0 * 4u);
      } else {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) lds_dma16(ds_a1, la + (unsigned)i * 4096u, a_off1[i], (unsigned)kk0 * 4u);
      }
#pragma unroll
      for (int i = 0; i < B_CH; ++i) lds_dma16(ds_b, lb + (unsigned)i * 4096u, b_off[i], (unsigned)k0 * 4u);
    }
  };
  // the tile is complete in LDS when this returns (the caller's barrier publishes it)
  auto store_tile = [&](int buf) {
    float* ab = lds + buf * STAGE;
    float* bb = ab + BM * LDK;
    if constexpr (DMA) {
      if constexpr (KIND == 1) {
        if (tile_masked) {
#pragma unroll
          for (int i = 0; i < A_CH; ++i)                       // position (tid + 256 i): the chunk a_kc was chosen for
            *reinterpret_cast<f32x4*>(ab + (tid + i * kGemmThreads) * 4) = ra[i] * rm[i];
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's loads-to-LDS have landed
    } else {
      if constexpr (KIND == 1 && !RAGGED) {
        if (tile_masked) {
#pragma unroll
          for (int i = 0; i < A_CH; ++i) ra[i] *= rm[i];
        }
      }
#pragma unroll
      for (int i = 0; i < A_CH; ++i)
        *reinterpret_cast<f32x4*>(ab + a_row[i] * LDK + a_kc[i]) = ra[i];
#pragma unroll
      for (int i = 0; i < B_CH; ++i)
        *reinterpret_cast<f32x4*>(bb + b_row[i] * LDK + b_kc[i]) = rb[i];
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // raw accumulators <-> [rows][N] in the C layout of the 32x32 MFMA (a lane's 32-column run is 128 contiguous bytes).
  // An element's address = a UNIFORM row pointer (scalar registers) + ONE per-lane offset: written so that the 64
  // accesses of a thread do not each hold a 64-bit address in vector registers (the first form did: 256 VGPRs + 160
  // AGPRs, one workgroup per CU, and the first layer ran at 0.62-0.70 where the same-shape hidden layer reaches 0.85).
  auto acc_io = [&](const float* src, float* dst) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = n0 + wn * WN + j * 32 + r;
      const bool cok = col < p.N;
      const unsigned loff = (unsigned)(4 * half) * (unsigned)p.N + (unsigned)col;      // < 2^31: 4 rows of N <= 2^28 floats
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int64_t urow = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2);   // uniform over the wave
          const bool ok = cok && urow + 4 * half < p.rows;
          if (src) {
            const float* rp = src + urow * p.N;
            if (ok) acc[i][j][e] = rp[loff];
          } else {
            float* wp = dst + urow * p.N;
            if (ok) wp[loff] = acc[i][j][e];
          }
        }
    }
  };
  int kt0 = 0, kt_dump = -1;
  if constexpr (KIND == 1) {
    if (p.acc_in) {
      kt0 = p.k_begin / BK;
      acc_io(p.acc_in, nullptr);
    }
    if (p.acc_out) kt_dump = p.k_dump / BK;
  }

  const int nk = RAGGED ? (p.K + BK - 1) / BK : p.K / BK;
  load_tile(kt0, 0);
  store_tile(0);
  __syncthreads();
  L2HMC_STAMP(1);
  // position of a fragment's chunk pair (2 kq, 2 kq + 1) in this lane's rows: (8 kq) ^ xa / xb floats into the row
  // (rows 32 apart share their swizzle)
  const int xa = swz(wm * (BM / 2) + r) * 4, xb = swz(wn * WN + r) * 4;
  for (int kt = kt0; kt < nk; ++kt) {
    const int cur = (kt - kt0) & 1;
    if (kt + 1 < nk) load_tile(kt + 1, cur ^ 1);   // in flight under the MFMAs below
    const float* as = lds + cur * STAGE + (wm * (BM / 2) + r) * LDK + half * 4;
    const float* bs = lds + cur * STAGE + BM * LDK + (wn * WN + r) * LDK + half * 4;
    // the fragments of k-step group kq + 1 are read from LDS while group kq's MFMAs issue (two register sets): read
    // and used in the same group -- what the loop did through round 3 -- every group of MFMAs waited out the LDS
    // latency of the reads in front of it (ISA: four ds_read_b128, s_waitcnt lgkmcnt, eight MFMAs, ...)
    f32x4 af[2][MT], bf[2][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) af[0][i] = *reinterpret_cast<const f32x4*>(as + i * 32 * LDK + (0 ^ xa));
#pragma unroll
    for (int j = 0; j < NT; ++j) bf[0][j] = *reinterpret_cast<const f32x4*>(bs + j * 32 * LDK + (0 ^ xb));
#pragma unroll
    for (int kq = 0; kq < BK / 8; ++kq) {
      const int cb = kq & 1, nb = cb ^ 1;
      if (kq + 1 < BK / 8) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
          af[nb][i] = *reinterpret_cast<const f32x4*>(as + i * 32 * LDK + (((kq + 1) * 8) ^ xa));
#pragma unroll
        for (int j = 0; j < NT; ++j)
          bf[nb][j] = *reinterpret_cast<const f32x4*>(bs + j * 32 * LDK + (((kq + 1) * 8) ^ xb));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cb][i][e], bf[cb][j][e], acc[i][j], 0, 0, 0);
      // pinned order (left alone the scheduler sinks the reads back in front of their use): next group's reads, then
      // this group's MFMAs
      if (kq + 1 < BK / 8) __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * MT * NT, 0);
    }
    if constexpr (KIND == 1) {
      if (kt + 1 == kt_dump) acc_io(nullptr, p.acc_out);   // uniform; at most once
    }
    if (kt + 1 < nk) store_tile(cur ^ 1);
    __syncthreads();
  }

  L2HMC_STAMP(2);
  // --- epilogue: + bias (+ t.Wt), relu, store.  C layout of 32x32 MFMA:
  // col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
  // Aligned shapes: the tile goes through LDS (the stage buffers are free) and leaves as 16-byte pieces of whole rows
  // -- a wave's store instruction covers 2 x 512 contiguous bytes instead of 2 x 128.
  static_assert(RAGGED || BM * LDC <= LDS_FLOATS, "the output tile must fit the stage buffers");
  const bool staged = !RAGGED && (p.ldo % 4) == 0 && (p.N % 4) == 0 && ((reinterpret_cast<uintptr_t>(p.out) & 15) == 0);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = n0 + wn * WN + j * 32 + r;
    const bool cok = col < p.N;
    const float bj = (KIND <= 2 && cok) ? p.bias[col] : 0.f;
    const float w0 = (KIND == 1 && cok && p.wt0) ? p.wt0[col] : 0.f;
    const float w1 = (KIND == 1 && cok && p.wt0) ? p.wt1[col] : 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int lrow = wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        const int64_t row = m0 + lrow;
        if (row < p.rows && cok) {
          float h;
          if (KIND == 1) {
            const int d = p.dir ? p.dir[row] : 0;
            h = l1_preact(acc[i][j][e], bj, d ? p.tc_b : p.tc_f, d ? p.ts_b : p.ts_f, w0, w1);
          } else {
            h = acc[i][j][e] + bj;
          }
          if (KIND == 3) h = p.gate[row * p.ldg + col] > 0.f ? h : 0.f;
          h = KIND <= 2 ? fmaxf(h, 0.f) : h;
          if (staged) lds[lrow * LDC + wn * WN + j * 32 + r] = h;
          else p.out[row * p.ldo + col] = h;
        }
      }
    }
  }
  if (staged) {
    __syncthreads();
    constexpr int C4 = BN / 4;                          // 16-byte pieces per tile row
#pragma unroll 4
    for (int c = tid; c < BM * C4; c += kGemmThreads) {
      const int lrow = c / C4, c4 = (c - lrow * C4) * 4;
      const int64_t row = m0 + lrow;
      if (row < p.rows && n0 + c4 < p.N)                // (N is a multiple of 4: a piece is inside or outside)
        *reinterpret_cast<f32x4*>(p.out + row * p.ldo + n0 + c4) = *reinterpret_cast<const f32x4*>(lds + lrow * LDC + c4);
    }
  }
  L2HMC_STAMP(3);
  L2HMC_STAMP_REAL(5);
}

// h1 = relu(pre + bias + t . Wt): the first layer's epilogue on a product kept from an earlier call (the momentum
// network sees the same (x, force) at the end of one leapfrog step and at the start of the next; only t differs)
__global__ __launch_bounds__(256) void l1_finish_kernel(L1FinishArgs p) {
  const int n4 = p.N >> 2;
  const int64_t total = p.rows * n4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / n4;
    const int c = (int)(i - row * n4) * 4;
    const int d = p.dir ? p.dir[row] : 0;
    const float tc = d ? p.tc_b : p.tc_f, ts = d ? p.ts_b : p.ts_f;
    const f32x4 a = *reinterpret_cast<const f32x4*>(p.pre + row * p.N + c);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + c);
    f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = {0.f, 0.f, 0.f, 0.f};
    if (p.wt0) {
      w0 = *reinterpret_cast<const f32x4*>(p.wt0 + c);
      w1 = *reinterpret_cast<const f32x4*>(p.wt1 + c);
    }
    f32x4 h;
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j] = fmaxf(l1_preact(a[j], b[j], tc, ts, w0[j], w1[j]), 0.f);
    *reinterpret_cast<f32x4*>(p.out + row * p.N + c) = h;
  }
}

int launch_l1_finish(const L1FinishArgs& a, hipStream_t stream) {
  L2HMC_REQUIRE(a.pre && a.out && a.bias && a.N > 0 && a.N % 4 == 0 && a.rows > 0, "l1_finish: bad arguments");
  const int64_t total = a.rows * (a.N >> 2);
  hipLaunchKernelGGL(l1_finish_kernel, dim3((unsigned)hmin(ceil_div(total, 256), 8192)), dim3(256), 0, stream, a);
  L2HMC_CHECK_LAUNCH("l1_finish");
  return L2HMC_OK;
}

// =====================================================================
// heads: (S,T,Q) = h2 . Whd^T + bhd, then materialise or fused v/x update.
// Workgroup tile: 64 rows x 32 output columns x 3 heads, so S, T and Q of one
// (row, col) sit in the same lane.  16x16x4 MFMAs: wave w owns rows 16w..16w+15.
// =====================================================================


// per-column constants and per-element epilogue shared by the heads kernels: (S, T, Q) from the three products,
// then materialise or the fused sub-update; returns the element's log-det contribution
struct HeadsCol { float b_s, b_t, b_q, e_s, e_q, kf, kb; };
__device__ __forceinline__ HeadsCol heads_col(const HeadsArgs& p, int col, bool cok) {
  HeadsCol c;
  c.b_s = cok ? p.bhd[col] : 0.f;
  c.b_t = cok ? p.bhd[p.D + col] : 0.f;
  c.b_q = cok ? p.bhd[2 * p.D + col] : 0.f;
  c.e_s = cok ? expf(p.cs[col]) : 0.f;
  c.e_q = cok ? expf(p.cq[col]) : 0.f;
  c.kf = c.kb = 0.f;
  if (p.mode == kHeadsUpdateX && cok) {
    c.kf = p.keep_f[col];
    c.kb = p.keep_b[col];
  }
  return c;
}
__device__ __forceinline__ float heads_element(const HeadsArgs& p, const HeadsCol& c, int64_t row, int col, float aS,
                                               float aT, float aQ) {
  const float S = fast_tanh(aS + c.b_s) * c.e_s;
  const float T = aT + c.b_t;
  float Q = aQ + c.b_q;
  Q = (p.q_tanh ? fast_tanh(Q) : Q) * c.e_q;
  const int64_t idx = row * p.D + col;
  if (p.mode == kHeadsMaterialise) {
    p.S[idx] = S;
    p.T[idx] = T;
    p.Q[idx] = Q;
    return 0.f;
  }
  const int d = p.dir ? p.dir[row] : 0;
  const float eps = p.eps;
  if (p.mode == kHeadsUpdateV) {
    // gauge_dynamics.py:497-506 (fwd), :549-559 (bwd)
    const float g = p.g[idx], v = p.v[idx];
    const float s = (d ? -0.5f : 0.5f) * eps * S;
    const float tq = eps * Q;
    const float kick = 0.5f * eps * (fast_exp(tq) * g - T);
    p.v[idx] = d ? fast_exp(s) * (v + kick) : v * fast_exp(s) - kick;
    return s;
  }
  // gauge_dynamics.py:519-531 (fwd), :574-584 (bwd)
  const float keep = d ? c.kb : c.kf;
  const float x = p.x[idx], v = p.v[idx];
  const float s = (d ? -eps : eps) * S;
  const float tq = eps * Q;
  const float drift = eps * (fast_exp(tq) * v + T);
  const float upd = d ? fast_exp(s) * (x - drift) : x * fast_exp(s) + drift;
  p.x[idx] = keep * x + (1.f - keep) * upd;
  return (1.f - keep) * s;
}

// BM = 32: half-height row tiles for launches with few live tiles (the active-column form of a small batch: cfg 4's
// position sub-updates have 256 live 64-row tiles, one per CU, and a lone workgroup keeps the matrix pipe half busy)
template <int BK, bool RAGGED = false, int BM = 64>
__global__ __launch_bounds__(kGemmThreads) void heads_kernel(HeadsArgs p) {
  constexpr int BNH = 32, NB = 3 * BNH;
  constexpr int RG = BM / 32;             // 16-row groups per wave (wave grid 2 x 2: BM / 2 rows x 16 columns per wave)
  constexpr int CPR = BK / 4;
  // Aligned shapes with k-tiles of 32 (DMA): the tiles go from global memory straight into LDS (gemm_relu_kernel has
  // the story).  Rows are 128 unpadded bytes; row R keeps its two 64-byte halves (k-steps kh = 0, 1) swapped when
  // (R >> 1) & 1: the 16 rows x 4 quarter-chunks of a fragment read then fall on 16 distinct 16-byte bank groups.
  // Otherwise rows of 40 / 72 floats (= 8 mod 64): conflict-free for the 16-row fragment reads.
  constexpr bool DMA = !RAGGED && BK == 32;
  constexpr int LDK = DMA ? BK : BK + 8;
  auto swz = [](int R) { return DMA ? 4 * ((R >> 1) & 1) : 0; };     // in 16-byte chunks
  constexpr int A_CH = BM * (BK / 4) / kGemmThreads;
  constexpr int B_CH = NB * (BK / 4) / kGemmThreads;
  constexpr int STAGE = (BM + NB) * LDK;
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

  L2HMC_STAMP_REAL(4);
  L2HMC_STAMP(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, r = lane & 15;
  const int wm = wave >> 1, wn = wave & 1;
  const int wv = __builtin_amdgcn_readfirstlane(wave);

  // Active-column form (HeadsArgs::cols_f): the tiles are walked column block by column block in launch order, so the
  // column blocks past the end of the list -- which leave at once -- come last and the live workgroups spread over
  // all CUs (row-major order interleaves live and empty tiles: the live ones doubled up on half the CUs).
  const int tile = p.cols_f ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
  const int mt_id = p.cols_f ? tile % p.mtiles : tile / p.ntiles;
  const int nt_id = p.cols_f ? tile / p.mtiles : tile - mt_id * p.ntiles;
  const int64_t m0 = (int64_t)mt_id * BM;
  const int n0 = nt_id * BNH;
  // output column n of this tile is column cols[n] of the heads
  const int* cols = p.cols_f ? (m0 >= p.dir_split ? p.cols_b : p.cols_f) : nullptr;
  const int ncol = cols ? *(m0 >= p.dir_split ? p.cnt_b : p.cnt_f) : p.D;
  if (n0 >= ncol) return;                  // (uniform; before any barrier)

  int a_row[A_CH], a_kc[A_CH];
  bool a_ok[A_CH];
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    const int c = tid + i * kGemmThreads;
    a_row[i] = c / CPR;
    a_kc[i] = ((c % CPR) ^ swz(a_row[i])) * 4;           // the k-chunk that belongs at LDS position c
    a_ok[i] = (m0 + a_row[i]) < p.rows;
  }
  int b_row[B_CH], b_kc[B_CH];
  int64_t b_src[B_CH];
  bool b_ok[B_CH];
#pragma unroll
  for (int i = 0; i < B_CH; ++i) {
    const int c = tid + i * kGemmThreads;
    b_row[i] = c / CPR;                     // 0..95 = head * 32 + nn
    b_kc[i] = ((c % CPR) ^ swz(b_row[i])) * 4;
    const int hd = b_row[i] >> 5, nn = b_row[i] & 31;
    b_ok[i] = (n0 + nn) < ncol;
    const int wcol = cols ? cols[b_ok[i] ? n0 + nn : ncol - 1] : n0 + nn;
    b_src[i] = ((int64_t)hd * p.D + wcol) * p.K;
  }
  // aligned shapes: the tile loads are buffer loads (common.h: buf_load16) -- A based at this tile's first row, the
  // weights at their start (launcher: 3 D K floats < 2 GiB); a chunk outside the tile / the column list points its
  // lane beyond the resource's window and reads 0, so every load is unconditional
  [[maybe_unused]] WSection rs_a, rs_b;
  [[maybe_unused]] WDesc ds_a, ds_b;
  [[maybe_unused]] unsigned a_boff[A_CH], b_boff[B_CH];
  [[maybe_unused]] const unsigned lds0 = lds_byte_address(lds);
  if constexpr (!RAGGED) {
    rs_a = wsection(p.A + m0 * p.lda);
    rs_b = wsection(p.Wt);
    ds_a = wdesc(p.A + m0 * p.lda);
    ds_b = wdesc(p.Wt);
#pragma unroll
    for (int i = 0; i < A_CH; ++i)
      a_boff[i] = a_ok[i] ? ((unsigned)a_row[i] * (unsigned)p.lda + (unsigned)a_kc[i]) * 4u : 0xfffffff0u;
#pragma unroll
    for (int i = 0; i < B_CH; ++i) b_boff[i] = b_ok[i] ? ((unsigned)b_src[i] + (unsigned)b_kc[i]) * 4u : 0xfffffff0u;
  }

  f32x4 ra[A_CH], rb[B_CH];
  auto load_tile = [&](int kt, [[maybe_unused]] int buf) {
    const int k0 = kt * BK;
    if constexpr (DMA) {
      const unsigned la = lds0 + (unsigned)(buf * STAGE) * 4u + (unsigned)wv * 1024u, lb = la + (unsigned)(BM * LDK) * 4u;
#pragma unroll
      for (int i = 0; i < A_CH; ++i) lds_dma16(ds_a, la + (unsigned)i * 4096u, a_boff[i], (unsigned)k0 * 4u);
#pragma unroll
      for (int i = 0; i < B_CH; ++i) lds_dma16(ds_b, lb + (unsigned)i * 4096u, b_boff[i], (unsigned)k0 * 4u);
      return;
    }
    if constexpr (!RAGGED) {
#pragma unroll
      for (int i = 0; i < A_CH; ++i) ra[i] = buf_load16(rs_a, a_boff[i], (unsigned)k0 * 4u);
#pragma unroll
      for (int i = 0; i < B_CH; ++i) rb[i] = buf_load16(rs_b, b_boff[i], (unsigned)k0 * 4u);
      return;
    }
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (a_ok[i]) {
        if constexpr (RAGGED) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (k0 + a_kc[i] + j < p.K) v[j] = p.A[(m0 + a_row[i]) * p.lda + k0 + a_kc[i] + j];
        } else {
          v = *reinterpret_cast<const f32x4*>(p.A + (m0 + a_row[i]) * p.lda + k0 + a_kc[i]);
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (b_ok[i]) {
        if constexpr (RAGGED) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (k0 + b_kc[i] + j < p.K) v[j] = p.Wt[b_src[i] + k0 + b_kc[i] + j];
        } else {
          v = *reinterpret_cast<const f32x4*>(p.Wt + b_src[i] + k0 + b_kc[i]);
        }
      }
      rb[i] = v;
    }
  };
  auto store_tile = [&](int buf) {
    if constexpr (DMA) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's loads-to-LDS have landed
      return;
    }
    float* ab = lds + buf * STAGE;
    float* bb = ab + BM * LDK;
#pragma unroll
    for (int i = 0; i < A_CH; ++i)
      *reinterpret_cast<f32x4*>(ab + a_row[i] * LDK + a_kc[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < B_CH; ++i)
      *reinterpret_cast<f32x4*>(bb + b_row[i] * LDK + b_kc[i]) = rb[i];
  };

  f32x4 acc[3][RG];
#pragma unroll
  for (int h = 0; h < 3; ++h)
#pragma unroll
    for (int j = 0; j < RG; ++j) acc[h][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = RAGGED ? (p.K + BK - 1) / BK : p.K / BK;
  load_tile(0, 0);
  store_tile(0);
  __syncthreads();
  L2HMC_STAMP(1);
  // (rows 16 apart and the three heads' rows, 32 apart, share their swizzle)
  const int xa = swz(wm * (BM / 2) + r) * 4, xb = swz(wn * 16 + r) * 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1, cur ^ 1);
    // wave (wm, wn): rows [32 wm, 32 wm + 32) x columns [16 wn, 16 wn + 16) of all three heads -- per 16 k two row
    // fragments and three column fragments feed 24 MFMAs (5 ds_read_b128; a wave owning 16 rows x all 96 columns
    // read 7)
    const float* as = lds + cur * STAGE + (wm * (BM / 2) + r) * LDK + q * 4;
    const float* bs = lds + cur * STAGE + BM * LDK + (wn * 16 + r) * LDK + q * 4;
    // (fragments of group kh + 1 are read while group kh's MFMAs issue: gemm_relu_kernel)
    f32x4 af[2][RG], bf[2][3];
#pragma unroll
    for (int i = 0; i < RG; ++i) af[0][i] = *reinterpret_cast<const f32x4*>(as + i * 16 * LDK + (0 ^ xa));
#pragma unroll
    for (int h = 0; h < 3; ++h) bf[0][h] = *reinterpret_cast<const f32x4*>(bs + h * 32 * LDK + (0 ^ xb));
#pragma unroll
    for (int kh = 0; kh < BK / 16; ++kh) {
      const int cb = kh & 1, nb = cb ^ 1;
      if (kh + 1 < BK / 16) {
#pragma unroll
        for (int i = 0; i < RG; ++i)
          af[nb][i] = *reinterpret_cast<const f32x4*>(as + i * 16 * LDK + (((kh + 1) * 16) ^ xa));
#pragma unroll
        for (int h = 0; h < 3; ++h)
          bf[nb][h] = *reinterpret_cast<const f32x4*>(bs + h * 32 * LDK + (((kh + 1) * 16) ^ xb));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int h = 0; h < 3; ++h)
#pragma unroll
          for (int i = 0; i < RG; ++i)
            acc[h][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cb][i][e], bf[cb][h][e], acc[h][i], 0, 0, 0);
      if (kh + 1 < BK / 16) __builtin_amdgcn_sched_group_barrier(0x100, RG + 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 12 * RG, 0);
    }
    if (kt + 1 < nk) store_tile(cur ^ 1);
    __syncthreads();
  }

  L2HMC_STAMP(2);
  // --- epilogue.  C layout of 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg.
  // acc[h][i]: rows m0 + 32 wm + 16 i + 4 q + e, column n0 + 16 wn + r (of the active list, if there is one)
  const int cidx = n0 + wn * 16 + r;
  const bool cok = cidx < ncol;
  const int col = cols ? cols[cok ? cidx : ncol - 1] : cidx;
  const HeadsCol c = heads_col(p, col, cok);
  float ld[RG][4];
#pragma unroll
  for (int i = 0; i < RG; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) ld[i][e] = 0.f;
#pragma unroll
  for (int i = 0; i < RG; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t row = m0 + wm * (BM / 2) + i * 16 + q * 4 + e;
      if (row >= p.rows || !cok) continue;
      ld[i][e] = heads_element(p, c, row, col, acc[0][i][e], acc[1][i][e], acc[2][i][e]);
    }
  if (p.mode != kHeadsMaterialise && p.ld_part) {
    // a row's 32 columns of this workgroup sit in two waves (wn = 0, 1): each reduces its 16 lanes, the second hands
    // its sums over through LDS (the stage buffers are free), the first adds them in a fixed order
    float t[RG][4];
#pragma unroll
    for (int i = 0; i < RG; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) t[i][e] = row16_sum(ld[i][e]);   // DPP on the VALU
    float* hand = lds;                                    // [wm][BM / 2 rows]
    if (wn == 1 && r == 0) {
#pragma unroll
      for (int i = 0; i < RG; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) hand[wm * (BM / 2) + i * 16 + q * 4 + e] = t[i][e];
    }
    __syncthreads();
    if (wn == 0 && r == 0) {
#pragma unroll
      for (int i = 0; i < RG; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int64_t row = m0 + wm * (BM / 2) + i * 16 + q * 4 + e;
          if (row < p.rows) p.ld_part[row * p.ncb + nt_id] += t[i][e] + hand[wm * (BM / 2) + i * 16 + q * 4 + e];
        }
    }
  }
  L2HMC_STAMP(3);
  L2HMC_STAMP_REAL(5);
}

// =====================================================================
// heads for grids that fill the chip (cfg 5: 4096 rows x 2048 columns): 128 rows x 64 output columns x 3 heads per
// workgroup on 32x32x2 MFMAs.  Wave (wm, wn) owns 64 rows x 32 columns of ALL THREE heads (S, T, Q of an element
// still share a lane): per 8 k it reads 2 + 3 fragments for 24 MFMAs of 64 cycles -- 0.10 ds_read_b128 per 1024
// multiply-adds against 0.29 in heads_kernel (7 reads per 24 MFMAs of 32 cycles), which is what held that kernel
// at 0.73 of the MFMA rate where the same-shape hidden layer reaches 0.86.  k-tiles of 16 keep the double buffer at
// 41 KB, so two workgroups share a CU with independent barriers.  Rows are stored without padding and the four
// 16-byte chunks of row R are XOR-swizzled with (R >> 2) & 3: the staging stores (4 rows x 4 chunks per 16 lanes) and
// the fragment reads (16 rows, one chunk) both touch 16 distinct 4-bank slots.  (Round 3 began with row stride 20:
// SQ_LDS_BANK_CONFLICT = a third of the LDS-active cycles, from the stores.)
// =====================================================================
__global__ __launch_bounds__(kGemmThreads) void heads32_kernel(HeadsArgs p) {
  constexpr int BM = 128, BNH = 64, NB = 3 * BNH, BK = 16;
  constexpr int CPR = BK / 4, LDK = BK;              // no pad: 16-byte chunk j of row R sits at slot j ^ ((R >> 2) & 3)
  constexpr int A_CH = BM * CPR / kGemmThreads;      // 2
  constexpr int B_CH = NB * CPR / kGemmThreads;      // 3
  constexpr int STAGE = (BM + NB) * LDK;
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int half = lane >> 5, r = lane & 31;
  const int wv = __builtin_amdgcn_readfirstlane(wave);

  // Active-column form (HeadsArgs::cols_f): the tiles are walked column block by column block in launch order, so the
  // column blocks past the end of the list -- which leave at once -- come last and the live workgroups spread over
  // all CUs (row-major order interleaves live and empty tiles: the live ones doubled up on half the CUs).
  const int tile = p.cols_f ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
  const int mt_id = p.cols_f ? tile % p.mtiles : tile / p.ntiles;
  const int nt_id = p.cols_f ? tile / p.mtiles : tile - mt_id * p.ntiles;
  const int64_t m0 = (int64_t)mt_id * BM;
  const int n0 = nt_id * BNH;
  // output column n of this tile is column cols[n] of the heads
  const int* cols = p.cols_f ? (m0 >= p.dir_split ? p.cols_b : p.cols_f) : nullptr;
  const int ncol = cols ? *(m0 >= p.dir_split ? p.cnt_b : p.cnt_f) : p.D;
  if (n0 >= ncol) return;                  // (uniform; before any barrier)

  // unconditional, clamped tile loads (rows past the end repeat the last row: their products are never stored),
  // straight into LDS (common.h: lds_dma16; gemm_relu_kernel has the story): the lane at LDS position c of a tile
  // asks for the chunk that belongs there -- A based at this tile's first row, the weights at their start
  const WDesc ds_a = wdesc(p.A + m0 * p.lda), ds_b = wdesc(p.Wt);
  const unsigned lds0 = lds_byte_address(lds);
  unsigned a_src[A_CH];
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    const int c = tid + i * kGemmThreads;
    const int row = c / CPR, kc = ((c % CPR) ^ ((row >> 2) & 3)) * 4;
    const int64_t g = (m0 + row) < p.rows ? m0 + row : p.rows - 1;
    a_src[i] = ((unsigned)(g - m0) * (unsigned)p.lda + (unsigned)kc) * 4u;
  }
  unsigned b_src[B_CH];
#pragma unroll
  for (int i = 0; i < B_CH; ++i) {
    const int c = tid + i * kGemmThreads;
    const int row = c / CPR, kc = ((c % CPR) ^ ((row >> 2) & 3)) * 4;     // row = head * 64 + nn
    const int hd = row >> 6, nn = row & 63;
    const int cidx = (n0 + nn) < ncol ? n0 + nn : ncol - 1;
    const int col = cols ? cols[cidx] : cidx;
    b_src[i] = (unsigned)((((int64_t)hd * p.D + col) * p.K + kc) * 4);
  }
  auto load_tile = [&](int kt, int buf) {
    const unsigned la = lds0 + (unsigned)(buf * STAGE) * 4u + (unsigned)wv * 1024u, lb = la + (unsigned)(BM * LDK) * 4u;
#pragma unroll
    for (int i = 0; i < A_CH; ++i) lds_dma16(ds_a, la + (unsigned)i * 4096u, a_src[i], (unsigned)(kt * BK) * 4u);
#pragma unroll
    for (int i = 0; i < B_CH; ++i) lds_dma16(ds_b, lb + (unsigned)i * 4096u, b_src[i], (unsigned)(kt * BK) * 4u);
  };
  auto store_tile = [&](int) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };   // this wave's loads have landed

  f32x16 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int h = 0; h < 3; ++h)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][h][e] = 0.f;

  const int nk = p.K / BK;
  load_tile(0, 0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1, cur ^ 1);
    // (every row offset below is a multiple of 32, so the swizzle term (row >> 2) & 3 is that of r alone)
    const int sw = (r >> 2) & 3;
    const float* as = lds + cur * STAGE + (wm * 64 + r) * LDK;
    const float* bs = lds + cur * STAGE + (BM + wn * 32 + r) * LDK;
    // (reading group kq + 1's fragments under group kq's MFMAs, as gemm_relu_kernel does, was measured here: the two
    //  k-step groups of a 16-deep k-tile give it nothing to hide and the second register set costs: 2241 -> 2374 us at cfg 5)
#pragma unroll
    for (int kq = 0; kq < BK / 8; ++kq) {
      const int slot = ((2 * kq + half) ^ sw) << 2;
      f32x4 af[2], bf[3];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const f32x4*>(as + i * 32 * LDK + slot);
#pragma unroll
      for (int h = 0; h < 3; ++h) bf[h] = *reinterpret_cast<const f32x4*>(bs + h * BNH * LDK + slot);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int h = 0; h < 3; ++h)
            acc[i][h] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[h][e], acc[i][h], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(cur ^ 1);
    __syncthreads();
  }

  // --- epilogue.  C layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
  const int cidx = n0 + wn * 32 + r;
  const bool cok = cidx < ncol;
  const int col = cols ? cols[cok ? cidx : ncol - 1] : cidx;
  const HeadsCol c = heads_col(p, col, cok);
  const int slot = (n0 >> 5) + wn;                   // this wave's 32-column log-det slot (ncb = ceil(D / 32))
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int64_t row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
      float ld = 0.f;
      if (row < p.rows && cok) ld = heads_element(p, c, row, col, acc[i][0][e], acc[i][1][e], acc[i][2][e]);
      if (p.mode != kHeadsMaterialise && p.ld_part) {
        const float t = wave_half_sums(ld);            // lanes 31 / 63 hold the sums over the 32 columns of a half
        if (r == 31 && row < p.rows) p.ld_part[row * p.ncb + slot] += t;
      }
    }
  }
}

// ---------------------------------------------------------------------
// host-side launchers (used by the C ABI in capi.hip)
// ---------------------------------------------------------------------
#ifdef L2HMC_STAMPS
unsigned long long* g_stamp_buf = nullptr;
int g_stamp_cls = 0;
extern "C" void l2hmc_debug_set_stamps(unsigned long long* buf, int cls) {
  g_stamp_buf = buf;
  g_stamp_cls = cls;
}
#endif

static int check_ptr16(const void* p, const char* what) {
  L2HMC_REQUIRE(p != nullptr, "%s is NULL", what);
  L2HMC_REQUIRE((reinterpret_cast<uintptr_t>(p) & 15) == 0, "%s is not 16-byte aligned", what);
  return L2HMC_OK;
}

// any positive widths: multiples of 32 take the 16-byte staged loads, everything else the RAGGED instantiation
int dense_net_supported(const l2hmc_dense_net* n) { return n->Ka > 0 && n->Kb > 0 && n->H > 0 && n->D > 0; }
int dense_net_tileable(const l2hmc_dense_net* n) {
  return dense_net_supported(n) && (n->Ka % BK) == 0 && (n->Kb % BK) == 0 && (n->H % BK) == 0;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int launch_gemm_relu(GemmReluArgs& a, hipStream_t stream) {
  L2HMC_REQUIRE(a.K > 0 && a.K1 >= 0 && a.K1 <= a.K && a.N > 0 && a.ldo > 0, "gemm: bad shape K=%d K1=%d N=%d", a.K,
                a.K1, a.N);
  L2HMC_REQUIRE(a.A1 && a.Wt && (a.K1 == a.K || a.A2), "gemm: NULL operand");
  const bool ragged = a.K % BK != 0 || a.K1 % BK != 0 || a.lda1 % 4 != 0 || (a.A2 != nullptr && a.lda2 % 4 != 0) ||
                      !aligned16(a.A1) || (a.K1 < a.K && !aligned16(a.A2)) || !aligned16(a.Wt) ||
                      (a.cmask_f && (!aligned16(a.cmask_f) || !aligned16(a.cmask_b)));
  L2HMC_REQUIRE(((!a.acc_in && !a.acc_out) || (!ragged && a.kind == 0)) &&
                    (!a.acc_in || (a.k_begin % BK == 0 && a.k_begin > 0 && a.k_begin < a.K)) &&
                    (!a.acc_out || (a.k_dump % BK == 0 && a.k_dump > 0 && a.k_dump <= a.K)) &&
                    (!(a.acc_in && a.acc_out) || a.k_dump > a.k_begin),       // (a dump at or before the resume point would be skipped)
                "gemm: a kept first-layer product needs tile-aligned widths (k_begin=%d, k_dump=%d, K=%d)", a.k_begin,
                a.k_dump, a.K);
  if (ragged) {
    a.ntiles = (int)ceil_div(a.N, 128);
    a.mtiles = (int)ceil_div(a.rows, 64);
    const dim3 grid(a.mtiles * a.ntiles);
    const bool first1 = a.K1 < a.K || a.wt0 != nullptr || a.cmask_f != nullptr;
    if (a.kind == 3) {
      L2HMC_REQUIRE(a.gate != nullptr, "gemm: bad backward-data descriptor");
      hipLaunchKernelGGL((gemm_relu_kernel<64, 3, 32, 128, true>), grid, dim3(kGemmThreads), 0, stream, a);
    } else if (a.kind == 4) {
      hipLaunchKernelGGL((gemm_relu_kernel<64, 4, 32, 128, true>), grid, dim3(kGemmThreads), 0, stream, a);
    } else if (first1) {
      hipLaunchKernelGGL((gemm_relu_kernel<64, 1, 32, 128, true>), grid, dim3(kGemmThreads), 0, stream, a);
    } else {
      hipLaunchKernelGGL((gemm_relu_kernel<64, 2, 32, 128, true>), grid, dim3(kGemmThreads), 0, stream, a);
    }
    L2HMC_CHECK_LAUNCH("gemm_relu (ragged)");
    return L2HMC_OK;
  }
#ifdef L2HMC_STAMPS
  a.stamps = (g_stamp_cls == ((a.K1 < a.K || a.wt0 != nullptr || a.cmask_f != nullptr) ? 1 : 2)) ? g_stamp_buf : nullptr;
#endif
  // the aligned tile loads are buffer loads with 32-bit byte offsets inside a tile of at most 128 rows
  L2HMC_REQUIRE((int64_t)128 * hmax(hmax(a.lda1, a.lda2), a.K) * 4 < (int64_t)1 << 31,
                "gemm: row stride %d too long for the tile loads", hmax(hmax(a.lda1, a.lda2), a.K));
  a.ntiles = (int)ceil_div(a.N, 128);
  // 128-row tiles once they still fill the chip (>= 2 tiles per CU), else 64-row tiles
  const int64_t t128 = ceil_div(a.rows, 128) * a.ntiles;
  const bool first = a.K1 < a.K || a.wt0 != nullptr || a.cmask_f != nullptr || a.acc_in != nullptr ||
                     a.acc_out != nullptr;
  const int cls = first ? kProfGemmL1 : kProfGemmL2;
  if (a.kind >= 3) {
    // backward-data products of the training path (train.hip): same tiles, different epilogue
    L2HMC_REQUIRE(!first && (a.kind == 4 || a.gate != nullptr), "gemm: bad backward-data descriptor");
    const bool big = t128 >= 512;
    a.mtiles = (int)ceil_div(a.rows, big ? 128 : 64);
    const dim3 grid(a.mtiles * a.ntiles);
    const bool deep3 = !big && (a.K % 64 == 0) && a.mtiles * a.ntiles <= 256;
    if (deep3) {
      a.ntiles = (int)ceil_div(a.N, 64);
      const dim3 g64(a.mtiles * a.ntiles);
      if (a.kind == 3) hipLaunchKernelGGL((gemm_relu_kernel<64, 3, 64, 64>), g64, dim3(kGemmThreads), 0, stream, a);
      else hipLaunchKernelGGL((gemm_relu_kernel<64, 4, 64, 64>), g64, dim3(kGemmThreads), 0, stream, a);
    } else if (big) {
      if (a.kind == 3) hipLaunchKernelGGL((gemm_relu_kernel<128, 3, 32>), grid, dim3(kGemmThreads), 0, stream, a);
      else hipLaunchKernelGGL((gemm_relu_kernel<128, 4, 32>), grid, dim3(kGemmThreads), 0, stream, a);
    } else {
      if (a.kind == 3) hipLaunchKernelGGL((gemm_relu_kernel<64, 3, 32>), grid, dim3(kGemmThreads), 0, stream, a);
      else hipLaunchKernelGGL((gemm_relu_kernel<64, 4, 32>), grid, dim3(kGemmThreads), 0, stream, a);
    }
    L2HMC_CHECK_LAUNCH("gemm_bwd_data");
    return L2HMC_OK;
  }
  prof_before(cls, stream);
  // grids that cannot fill the chip with 64 x 128 tiles (<= 256 of them) run 64 x 64 tiles with 64-deep k-tiles:
  // half the barriers, twice the workgroups, 70 KB of LDS each => two co-resident workgroups per CU
  const bool deep = (a.K % 64 == 0) && (a.K1 % 64 == 0) && ceil_div(a.rows, 64) * a.ntiles <= 256 &&
                    (!a.acc_in || a.k_begin % 64 == 0) && (!a.acc_out || a.k_dump % 64 == 0);
  if (t128 >= 512) {
    a.mtiles = (int)ceil_div(a.rows, 128);
    const dim3 grid(a.mtiles * a.ntiles);
    if (first) hipLaunchKernelGGL((gemm_relu_kernel<128, 1, 32>), grid, dim3(kGemmThreads), 0, stream, a);
    else hipLaunchKernelGGL((gemm_relu_kernel<128, 2, 32>), grid, dim3(kGemmThreads), 0, stream, a);
  } else {
    a.mtiles = (int)ceil_div(a.rows, 64);
    const dim3 grid(a.mtiles * a.ntiles);
    if (deep) {
      // 64 x 64 tiles: twice the workgroups, 70 KB of LDS each => two co-resident workgroups per CU with
      // independent barriers (two waves per SIMD) instead of one
      a.ntiles = (int)ceil_div(a.N, 64);
      const dim3 g64(a.mtiles * a.ntiles);
      if (first) hipLaunchKernelGGL((gemm_relu_kernel<64, 1, 64, 64>), g64, dim3(kGemmThreads), 0, stream, a);
      else hipLaunchKernelGGL((gemm_relu_kernel<64, 2, 64, 64>), g64, dim3(kGemmThreads), 0, stream, a);
    } else {
      if (first) hipLaunchKernelGGL((gemm_relu_kernel<64, 1, 32>), grid, dim3(kGemmThreads), 0, stream, a);
      else hipLaunchKernelGGL((gemm_relu_kernel<64, 2, 32>), grid, dim3(kGemmThreads), 0, stream, a);
    }
  }
  prof_after(cls, stream);
  L2HMC_CHECK_LAUNCH("gemm_relu");
  return L2HMC_OK;
}

// Ascending lists of the columns a position sub-update does not hold fixed: block (s, which) walks mask row s in
// chunks of 256 columns and compacts those with keep != 1, where keep = mask (which 0) or 1 - mask (which 1).
__global__ __launch_bounds__(256) void active_cols_kernel(const float* __restrict__ masks, int D, int* __restrict__ lists,
                                                          int* __restrict__ counts) {
  __shared__ int wsum[4];
  __shared__ int base;
  const int s = blockIdx.x >> 1, which = blockIdx.x & 1;
  const float* m = masks + (size_t)s * D;
  int* out = lists + ((size_t)s * 2 + which) * D;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  for (int c0 = 0; c0 < D; c0 += 256) {
    const int c = c0 + threadIdx.x;
    bool act = false;
    if (c < D) {
      const float keep = which == 0 ? m[c] : 1.f - m[c];     // (invert_mask_kernel forms 1 - mask the same way)
      act = keep != 1.f;
    }
    const unsigned long long b = __ballot(act);
    if (lane == 0) wsum[wave] = __popcll(b);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (act) out[off + __popcll(b & ((1ull << lane) - 1ull))] = c;
    __syncthreads();
    if (threadIdx.x == 0) base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) counts[s * 2 + which] = base;
}

int launch_active_cols(const float* masks, int num_steps, int D, int* lists, int* counts, hipStream_t stream) {
  hipLaunchKernelGGL(active_cols_kernel, dim3((unsigned)(2 * num_steps)), dim3(256), 0, stream, masks, D, lists, counts);
  L2HMC_CHECK_LAUNCH("active_cols");
  return L2HMC_OK;
}

int launch_heads(HeadsArgs& a, hipStream_t stream) {
  L2HMC_REQUIRE(a.K > 0 && a.D > 0 && a.A && a.Wt, "heads: bad arguments");
  const bool ragged = a.K % BK != 0 || a.lda % 4 != 0 || !aligned16(a.A) || !aligned16(a.Wt);
  // (the aligned tile loads are buffer loads with 32-bit byte offsets: from the weights' start, from a tile's first row)
  L2HMC_REQUIRE(ragged || ((int64_t)3 * a.D * a.K * 4 < ((int64_t)1 << 31) && (int64_t)128 * a.lda * 4 < ((int64_t)1 << 31)),
                "heads: D=%d K=%d lda=%d too large for the tile loads", a.D, a.K, a.lda);
  // the active-column form needs row tiles of ONE direction: the split between forward and backward rows on a tile edge
  auto split_ok = [&](int bm) { return a.dir_split >= a.rows || a.dir_split % bm == 0; };
  if (a.cols_f && (a.mode != kHeadsUpdateX || !a.cols_b || !a.cnt_f || !a.cnt_b)) a.cols_f = nullptr;
  if (ragged) {
    a.cols_f = nullptr;
    a.mtiles = (int)ceil_div(a.rows, 64);
    a.ntiles = (int)ceil_div(a.D, 32);
    L2HMC_REQUIRE(a.ld_part == nullptr || a.ncb == a.ntiles, "heads: ncb=%d != %d", a.ncb, a.ntiles);
    hipLaunchKernelGGL((heads_kernel<32, true>), dim3(a.mtiles * a.ntiles), dim3(kGemmThreads), 0, stream, a);
    L2HMC_CHECK_LAUNCH("heads (ragged)");
    return L2HMC_OK;
  }
#ifdef L2HMC_STAMPS
  a.stamps = g_stamp_cls == 3 ? g_stamp_buf : nullptr;
#endif
  // grids of at least two 128 x 64 tiles per CU: the 32x32x2 form (fewer fragment reads per MFMA)
  if (a.K % 16 == 0 && a.D % 64 == 0 && ceil_div(a.rows, 128) * (a.D / 64) >= 512 &&
      (a.ld_part == nullptr || a.ncb == a.D / 32)) {
    a.mtiles = (int)ceil_div(a.rows, 128);
    a.ntiles = a.D / 64;
    if (a.cols_f && !split_ok(128)) a.cols_f = nullptr;
    prof_before(kProfHeads, stream);
    hipLaunchKernelGGL(heads32_kernel, dim3(a.mtiles * a.ntiles), dim3(kGemmThreads), 0, stream, a);
    prof_after(kProfHeads, stream);
    L2HMC_CHECK_LAUNCH("heads32");
    return L2HMC_OK;
  }
  a.mtiles = (int)ceil_div(a.rows, 64);
  a.ntiles = (int)ceil_div(a.D, 32);
  L2HMC_REQUIRE(a.ld_part == nullptr || a.ncb == a.ntiles, "heads: ncb=%d != %d", a.ncb, a.ntiles);
  if (a.cols_f && !split_ok(64)) a.cols_f = nullptr;
  prof_before(kProfHeads, stream);
  // active-column launches keep about half of their column blocks: with fewer than two live 64-row tiles per CU, 32-row
  // tiles put two workgroups on every CU again
  // (32-row tiles for the all-columns launches as well -- 1024 workgroups, three per CU -- were measured: 69 -> 72 us)
  if (a.cols_f && (int64_t)a.mtiles * a.ntiles / 2 < 512 && split_ok(32) && a.rows % 32 == 0) {
    a.mtiles = (int)ceil_div(a.rows, 32);
    hipLaunchKernelGGL((heads_kernel<32, false, 32>), dim3(a.mtiles * a.ntiles), dim3(kGemmThreads), 0, stream, a);
    prof_after(kProfHeads, stream);
    L2HMC_CHECK_LAUNCH("heads (32-row tiles)");
    return L2HMC_OK;
  }
  if (a.K % 64 == 0 && a.mtiles * a.ntiles <= 256)
    hipLaunchKernelGGL(heads_kernel<64>, dim3(a.mtiles * a.ntiles), dim3(kGemmThreads), 0, stream, a);
  else
    hipLaunchKernelGGL(heads_kernel<32>, dim3(a.mtiles * a.ntiles), dim3(kGemmThreads), 0, stream, a);
  prof_after(kProfHeads, stream);
  L2HMC_CHECK_LAUNCH("heads");
  return L2HMC_OK;
}

}  // namespace l2hmc
