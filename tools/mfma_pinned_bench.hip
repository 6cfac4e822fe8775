// Micro-benchmark (diagnostic, not shipped), round 4: the weight stream of the whole-step kernel K1 with the SCHEDULE
// PINNED BY HAND.  tools/mfma_stream_bench.hip (rounds 2-3) left the order of loads, waits and MFMAs to the compiler
// and read 38 cycles per v_mfma_f32_16x16x4_f32 "wherever the load is placed"; its ISA shows why that experiment could
// not separate issue cost from latency: the compiler drains the queue (s_waitcnt vmcnt(0)) at every loop header and
// sinks the ring's loads into bursts of 16.  Here every load is an inline-asm instruction the compiler does not
// track, every wait an explicit s_waitcnt with the slot's registers tied to it, and sched_barrier fixes the
// interleave: ONE load per G MFMAs, issued while the matrix pipe is busy.
//   mode 7: B stream from global memory (L2), ring of D slots in registers, vmcnt((D - 2) * NT) before each block
//   mode 8: B stream from a static LDS image (ds_read_b128 per 4 MFMAs) -- what the consumer side of an LDS ring costs
//   mode 9: LDS ring filled by LDS-DMA (global_load_lds_dwordx4) from loader waves, FULL / FREE words in LDS
// A fragments always come from LDS (as in the kernel).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int NT = 8;

#define SB() __builtin_amdgcn_sched_barrier(0)

// Bounded wait on an LDS generation counter: a handshake bug must never hang the GPU.  After ~0.1 s of spinning the
// waiter sets the workgroup's `dead` word; every later wait of every wave then falls through at once and the host
// reports the run as aborted.
__device__ __forceinline__ void wait_ge(volatile int* word, int want, volatile int* dead) {
  int spins = 0;
  while (*word < want) {
    if (*dead) return;
    if (++spins > (1 << 21)) { *dead = 1; return; }
    __builtin_amdgcn_s_sleep(1);
  }
}

__device__ __forceinline__ void gload(f32x4& d, unsigned voff, const float* sbase, int imm) {
  // saddr form: uniform 64-bit base in SGPRs + one 32-bit per-lane offset
  switch (imm) {   // the offset field is an immediate
#define C(I) case I: asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #I : "=v"(d) : "v"(voff), "s"(sbase)); break;
    C(-4096) C(-3072) C(-2048) C(-1024) C(0) C(1024) C(2048) C(3072)
#undef C
  }
}

__device__ __forceinline__ void gload64(f32x4& d, const float* vaddr, int imm) {
  // 64-bit per-lane address (what the compiler emits for the shipped kernel's ordinary loads)
  switch (imm) {
#define C(I) case I: asm volatile("global_load_dwordx4 %0, %1, off offset:" #I : "=v"(d) : "v"(vaddr)); break;
    C(-4096) C(-3072) C(-2048) C(-1024) C(0) C(1024) C(2048) C(3072)
#undef C
  }
}

template <int N>
__device__ __forceinline__ void wait_vm(f32x4 (&b)[NT]) {
  asm volatile("s_waitcnt vmcnt(%8)"
               : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7])
               : "n"(N));
}

// one block: 32 MFMAs on slot `cur`, the 8 loads of the chunk at `src` interleaved one per G = 4 MFMAs into `nxt`
template <bool LOAD>
__device__ __forceinline__ void block_pinned(const f32x4 a, f32x4 (&cur)[NT], f32x4 (&nxt)[NT], f32x4 (&acc)[NT],
                                             unsigned voff, const float* src) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int t = 4 * h; t < 4 * h + 4; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[t][e], a[e], acc[t], 0, 0, 0);
      SB();
      if (LOAD) gload(nxt[e * 2 + h], voff, src, (e * 2 + h) * 1024 - 4096);
      SB();
    }
  }
}

template <int D>
__global__ __launch_bounds__(256) void bench7(const float* __restrict__ w, float* out, int nkc, int iters,
                                              unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) float lds[16 * 520];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16 * 520; i += 256) lds[i] = 0.001f * (i % 97);
  __syncthreads();
  f32x4 acc[NT];
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0, 0, 0, 0};
  const float* wbase = w + (size_t)__builtin_amdgcn_readfirstlane(wave) * nkc * NT * 256 + 1024;   // + 4096 bytes
  const unsigned voff = lane * 16;
  const float* ap = lds + (lane & 15) * 520 + (lane >> 4) * 4;
  f32x4 b[D][NT];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    // prime D - 1 slots
#pragma unroll
    for (int s = 0; s < D - 1; ++s)
#pragma unroll
      for (int t = 0; t < NT; ++t) gload(b[s][t], voff, wbase + (size_t)s * NT * 256, t * 1024 - 4096);
    f32x4 a0 = *(const f32x4*)(ap);
#pragma nounroll
    for (int kc = 0; kc + D <= nkc; kc += D) {
#pragma unroll
      for (int s = 0; s < D; ++s) {
        const f32x4 a1 = *(const f32x4*)(ap + ((kc + s + 1) & 31) * 16);
        // chunk kc + s + D - 1 goes into the slot block kc + s - 1 released; the tail re-loads the last chunk
        const int nx = min(kc + s + D - 1, nkc - 1);
        wait_vm<(D - 2) * NT>(b[s]);
        block_pinned<true>(a0, b[s], b[(s + D - 1) % D], acc, voff, wbase + (size_t)nx * NT * 256);
        a0 = a1;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int t = 0; t < NT; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  for (int d = 0; d < D; ++d) s += b[d][0][0] * 1e-30f;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

// mode 7 variants (bisecting what separates mode 7 from the shipped core): ADDR64 = per-lane 64-bit addresses instead
// of SGPR base + 32-bit offset; GROUP4 = tiles in two groups of four (acc re-used after 4 MFMAs, the A operand kept
// for 4 consecutive MFMAs) with the ring re-loaded in place half a block late, as csrc/fused_common.h does
template <int D, bool ADDR64, bool GROUP4>
__global__ __launch_bounds__(256) void bench7v(const float* __restrict__ w, float* out, int nkc, int iters,
                                               unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) float lds[16 * 520];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16 * 520; i += 256) lds[i] = 0.001f * (i % 97);
  __syncthreads();
  f32x4 acc[NT];
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0, 0, 0, 0};
  const float* wbase = w + (size_t)__builtin_amdgcn_readfirstlane(wave) * nkc * NT * 256 + 1024;
  const unsigned voff = lane * 16;
  const float* vbase = wbase + lane * 4;
  const float* ap = lds + (lane & 15) * 520 + (lane >> 4) * 4;
  f32x4 b[D][NT];
  auto ld = [&](f32x4& d, int chunk, int tile) {
    if (ADDR64) gload64(d, vbase + (size_t)chunk * NT * 256, tile * 1024 - 4096);
    else gload(d, voff, wbase + (size_t)chunk * NT * 256, tile * 1024 - 4096);
  };
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (GROUP4) {
#pragma unroll
      for (int s = 0; s < D - 1; ++s)
#pragma unroll
        for (int t = 0; t < NT; ++t) ld(b[s][t], s, t);
#pragma unroll
      for (int t = 0; t < 4; ++t) ld(b[D - 1][t], D - 1, t);
    } else {
#pragma unroll
      for (int s = 0; s < D - 1; ++s)
#pragma unroll
        for (int t = 0; t < NT; ++t) ld(b[s][t], s, t);
    }
    f32x4 a0 = *(const f32x4*)(ap);
#pragma nounroll
    for (int kc = 0; kc + D <= nkc; kc += D) {
#pragma unroll
      for (int s = 0; s < D; ++s) {
        const f32x4 a1 = *(const f32x4*)(ap + ((kc + s + 1) & 31) * 16);
        if (GROUP4) {
          // in flight behind this block's fragments: (D - 1) blocks' worth of loads minus the half just issued
          wait_vm<(D - 1) * NT - 4>(b[s]);
          const int n1 = min(kc + s + D - 1, nkc - 1), n2 = min(kc + s + D, nkc - 1);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s][t][e], a0[e], acc[t], 0, 0, 0);
            SB();
            ld(b[(s + D - 1) % D][4 + e], n1, 4 + e);
            SB();
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int t = 4; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s][t][e], a0[e], acc[t], 0, 0, 0);
            SB();
            ld(b[s][e], n2, e);
            SB();
          }
        } else {
          const int nx = min(kc + s + D - 1, nkc - 1);
          wait_vm<(D - 2) * NT>(b[s]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
              for (int t = 4 * h; t < 4 * h + 4; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s][t][e], a0[e], acc[t], 0, 0, 0);
              SB();
              ld(b[(s + D - 1) % D][e * 2 + h], nx, e * 2 + h);
              SB();
            }
          }
        }
        a0 = a1;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int t = 0; t < NT; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  for (int d = 0; d < D; ++d) s += b[d][0][0] * 1e-30f;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

// reference for the value check: same arithmetic, same order per accumulator, loads and waits left to the compiler
__global__ __launch_bounds__(256) void bench_ref(const float* __restrict__ w, float* out, int nkc, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[16 * 520];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16 * 520; i += 256) lds[i] = 0.001f * (i % 97);
  __syncthreads();
  f32x4 acc[NT];
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0, 0, 0, 0};
  const float* wp = w + (size_t)wave * nkc * NT * 256 + lane * 4;
  const float* ap = lds + (lane & 15) * 520 + (lane >> 4) * 4;
  for (int it = 0; it < iters; ++it)
    for (int kc = 0; kc < nkc; ++kc) {
      const f32x4 a = *(const f32x4*)(ap + (kc & 31) * 16);
      for (int e = 0; e < 4; ++e)
        for (int t = 0; t < NT; ++t) {
          const f32x4 b = *(const f32x4*)(wp + ((size_t)kc * NT + t) * 256);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[e], a[e], acc[t], 0, 0, 0);
        }
    }
  float s = 0;
  for (int t = 0; t < NT; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// mode 10: the pinned interleave of mode 7 with ORDINARY loads: the compiler tracks them and places the waits itself
// (no untracked register writes: what the shipped kernel can use safely); branch-free, the tail re-loads the last chunk
template <int D>
__global__ __launch_bounds__(256) void bench10(const float* __restrict__ w, float* out, int nkc, int iters,
                                               unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) float lds[16 * 520];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16 * 520; i += 256) lds[i] = 0.001f * (i % 97);
  __syncthreads();
  f32x4 acc[NT];
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0, 0, 0, 0};
  const float* wp = w + (size_t)wave * nkc * NT * 256 + lane * 4;
  const float* ap = lds + (lane & 15) * 520 + (lane >> 4) * 4;
  f32x4 b[D][NT];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < D - 1; ++s)
#pragma unroll
      for (int t = 0; t < NT; ++t) b[s][t] = *(const f32x4*)(wp + ((size_t)s * NT + t) * 256);
    f32x4 a0 = *(const f32x4*)(ap);
#pragma nounroll
    for (int kc = 0; kc + D <= nkc; kc += D) {
#pragma unroll
      for (int s = 0; s < D; ++s) {
        const f32x4 a1 = *(const f32x4*)(ap + ((kc + s + 1) & 31) * 16);
        const int nx = min(kc + s + D - 1, nkc - 1);
        const float* src = wp + (size_t)nx * NT * 256;
        SB();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int t = 4 * h; t < 4 * h + 4; ++t)
              acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s][t][e], a0[e], acc[t], 0, 0, 0);
            SB();
            b[(s + D - 1) % D][e * 2 + h] = *(const f32x4*)(src + (e * 2 + h) * 256);
            SB();
          }
        }
        a0 = a1;
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int t = 0; t < NT; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

// mode 8: the B fragments of a block come from a static LDS image (3 chunks per wave), read one ds_read_b128 per 4
// MFMAs one block ahead (double buffer in registers)
__global__ __launch_bounds__(256) void bench8(float* out, int nkc, int iters, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) float lds[16 * 520 + 4 * 3 * NT * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16 * 520 + 4 * 3 * NT * 256; i += 256) lds[i] = 0.001f * (i % 97);
  __syncthreads();
  f32x4 acc[NT];
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0, 0, 0, 0};
  const float* ap = lds + (lane & 15) * 520 + (lane >> 4) * 4;
  const float* bp = lds + 16 * 520 + wave * 3 * NT * 256 + lane * 4;
  f32x4 b[2][NT];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < NT; ++t) b[0][t] = *(const f32x4*)(bp + t * 256);
    f32x4 a0 = *(const f32x4*)(ap);
#pragma nounroll
    for (int kc = 0; kc + 2 <= nkc; kc += 2) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const f32x4 a1 = *(const f32x4*)(ap + ((kc + s + 1) & 31) * 16);
        const float* src = bp + ((kc + s + 1) % 3) * NT * 256;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int t = 4 * h; t < 4 * h + 4; ++t)
              acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s][t][e], a0[e], acc[t], 0, 0, 0);
            SB();
            b[s ^ 1][e * 2 + h] = *(const f32x4*)(src + (e * 2 + h) * 256);
            SB();
          }
        }
        a0 = a1;
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int t = 0; t < NT; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

// mode 9: LDS ring per consumer wave, filled by its own loader wave (wave w + 4, same SIMD) with LDS-DMA.
// A ring slot holds TS tiles of one k-chunk (TS KiB; a chunk = NT / TS consecutive slots), NS slots per wave.
// full[w][slot] / freed[w][slot] are generation counters in LDS: the loader publishes a slot once vmcnt says its DMA has
// landed (PIPE slots stay in flight behind it), the consumer releases it once the fragments are in registers.
template <int NS, int TS>
__global__ __launch_bounds__(512) void bench9(const float* __restrict__ w, float* out, int nkc, int iters,
                                              unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int SUB = NT / TS;                               // slots per chunk
  float* ring = lds + 16 * 520;                              // [4 waves][NS][TS * 256]
  volatile int* full = (volatile int*)(ring + 4 * NS * TS * 256);     // [4][NS]
  volatile int* freed = full + 4 * NS;                                // [4][NS]
  volatile int* dead = freed + 4 * NS;                                // [1]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cw = wave & 3;
  for (int i = threadIdx.x; i < 16 * 520; i += 512) lds[i] = 0.001f * (i % 97);
  if (threadIdx.x < 8 * NS + 1) ((int*)full)[threadIdx.x] = 0;
  __syncthreads();
  const int total = nkc * iters * SUB;                       // slots streamed per wave
  constexpr int PIPE = NS >= 6 ? 3 : NS >= 4 ? 2 : 1;
  if (wave >= 4) {
    // ---- loader: unit u -> slot u % NS, generation u / NS + 1
    const float* src = w + (size_t)cw * nkc * NT * 256 + lane * 4;
    float* dst = ring + (size_t)cw * NS * TS * 256;
    const int per_pass = nkc * SUB;
    int up = 0;
    for (int u = 0; u < total; ++u) {
      const int slot = u % NS, gen = u / NS;
      if (gen > 0) wait_ge(&freed[cw * NS + slot], gen, dead);
#pragma unroll
      for (int t = 0; t < TS; ++t)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((size_t)up * TS + t) * 256),
                                         (__attribute__((address_space(3))) void*)(dst + (slot * TS + t) * 256), 16, 0, 0);
      if (++up == per_pass) up = 0;
      if (u >= PIPE) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIPE * TS) : "memory");
        if (lane == 0) full[cw * NS + (u - PIPE) % NS] = (u - PIPE) / NS + 1;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int u = total - PIPE; u < total; ++u)
      if (u >= 0 && lane == 0) full[cw * NS + u % NS] = u / NS + 1;
    return;
  }
  // ---- consumer
  f32x4 acc[NT];
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0, 0, 0, 0};
  const float* ap = lds + (lane & 15) * 520 + (lane >> 4) * 4;
  const float* rp = ring + (size_t)cw * NS * TS * 256 + lane * 4;
  f32x4 b[2][TS];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  wait_ge(&full[cw * NS + 0], 1, dead);
#pragma unroll
  for (int t = 0; t < TS; ++t) b[0][t] = *(const f32x4*)(rp + t * 256);
  f32x4 a0 = *(const f32x4*)(ap);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) freed[cw * NS + 0] = 1;                     // unit 0 is in registers: its slot is free again
  int un = 1;                                                // the unit read during the current one
  for (int c = 0; c < nkc * iters; c += 2) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const f32x4 a1 = *(const f32x4*)(ap + ((c + s + 1) & 31) * 16);
#pragma unroll
      for (int sub = 0; sub < SUB; ++sub) {
        constexpr int dummy = 0; (void)dummy;
        const int p = (s * SUB + sub) & 1;                   // register buffer of this unit (compile time)
        const int slotn = un % NS, genn = un / NS + 1;
        const bool more = un < total;
        if (more) wait_ge(&full[cw * NS + slotn], genn, dead);
        const float* src = rp + slotn * TS * 256;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int h = 0; h < TS / 4; ++h) {
#pragma unroll
            for (int t = 4 * h; t < 4 * h + 4; ++t)
              acc[sub * TS + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[p][t][e], a0[e], acc[sub * TS + t], 0, 0, 0);
            SB();
            if (more && e * (TS / 4) + h < TS) b[p ^ 1][e * (TS / 4) + h] = *(const f32x4*)(src + (e * (TS / 4) + h) * 256);
            SB();
          }
        }
        if (more) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (lane == 0) freed[cw * NS + slotn] = genn;
        }
        ++un;
      }
      a0 = a1;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int t = 0; t < NT; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = *dead ? ~0ull : t1 - t0;
}

static void report(const char* name, unsigned long long* cyc, int nkc, int iters, float ms, int used_nkc) {
  std::vector<unsigned long long> h(1024);
  (void)hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * 1024, hipMemcpyDeviceToHost);
  double sum = 0;
  for (auto v : h) {
    if (v == ~0ull) { printf("%-78s nkc %3d: ABORTED (a ring wait timed out: handshake bug)\n", name, nkc); fflush(stdout); return; }
    sum += v;
  }
  const double nm = (double)iters * used_nkc * 4 * NT;
  printf("%-78s nkc %3d: %.1f cyc/MFMA  %.3f ms  %.1f TFLOP/s  (%.0f GB/s per CU)\n", name, nkc, sum / 1024 / nm, ms,
         1024.0 * nm * 2048 / ms / 1e9, 4.0 * nm / 32 * 8192 / ms / 1e6);
  fflush(stdout);
}

static std::vector<float> g_ref;
static void make_ref(const float* w, float* out, int nkc, int iters) {
  bench_ref<<<256, 256>>>(w, out, nkc, iters);
  (void)hipDeviceSynchronize();
  g_ref.resize(256 * 256);
  (void)hipMemcpy(g_ref.data(), out, g_ref.size() * 4, hipMemcpyDeviceToHost);
}
static const char* check(const float* out) {
  std::vector<float> h(256 * 256);
  (void)hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
  for (size_t i = 0; i < h.size(); ++i)
    if (h[i] != g_ref[i]) return "VALUES DIFFER from the reference kernel";
  return "values equal the reference kernel bit for bit";
}

template <int D>
void run10(const char* name, const float* w, float* out, unsigned long long* cyc, int nkc, int iters) {
  bench10<D><<<256, 256>>>(w, out, nkc, 2, cyc);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  bench10<D><<<256, 256>>>(w, out, nkc, iters, cyc);
  (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  report(name, cyc, nkc, iters, ms, nkc / D * D);
  printf("        %s\n", check(out));
}

template <int D, bool ADDR64, bool GROUP4>
void run7v(const char* name, const float* w, float* out, unsigned long long* cyc, int nkc, int iters) {
  bench7v<D, ADDR64, GROUP4><<<256, 256>>>(w, out, nkc, 2, cyc);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  bench7v<D, ADDR64, GROUP4><<<256, 256>>>(w, out, nkc, iters, cyc);
  (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  report(name, cyc, nkc, iters, ms, nkc / D * D);
  printf("        %s\n", check(out));
}

template <int D>
void run7(const char* name, const float* w, float* out, unsigned long long* cyc, int nkc, int iters) {
  bench7<D><<<256, 256>>>(w, out, nkc, 2, cyc);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  bench7<D><<<256, 256>>>(w, out, nkc, iters, cyc);
  (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  report(name, cyc, nkc, iters, ms, nkc / D * D);
  printf("        %s\n", check(out));
}

void run8(const char* name, float* out, unsigned long long* cyc, int nkc, int iters) {
  (void)hipFuncSetAttribute((const void*)bench8, hipFuncAttributeMaxDynamicSharedMemorySize, 0);
  bench8<<<256, 256>>>(out, nkc, 2, cyc);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  bench8<<<256, 256>>>(out, nkc, iters, cyc);
  (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  report(name, cyc, nkc, iters, ms, nkc / 2 * 2);
}

template <int NS, int TS>
void run9(const char* name, const float* w, float* out, unsigned long long* cyc, int nkc, int iters) {
  const size_t smem = (16 * 520 + 4 * NS * TS * 256 + 8 * NS + 4) * 4;
  (void)hipFuncSetAttribute((const void*)bench9<NS, TS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  bench9<NS, TS><<<256, 512, smem>>>(w, out, nkc, 2, cyc);
  if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return; }
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  bench9<NS, TS><<<256, 512, smem>>>(w, out, nkc, iters, cyc);
  (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  char full[200]; snprintf(full, sizeof full, "%s [%zu KB LDS for the ring]", name, (size_t)4 * NS * TS);
  report(full, cyc, nkc, iters, ms, nkc);
}

int main() {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  float* w; float* out; unsigned long long* cyc;
  const int nkc_max = 96;
  const size_t copy = (size_t)4 * nkc_max * NT * 256;
  (void)hipMalloc(&w, copy * 4 + 65536); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 2048 * 8);
  std::vector<float> h(copy); for (size_t i = 0; i < copy; ++i) h[i] = (float)(rand() % 1000) * 1e-4f;
  (void)hipMemcpy(w, h.data(), copy * 4, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep)
    for (int nkc : {36, 96}) {
      const int iters = 100;
      make_ref(w, out, nkc, iters);
      run10<3>("mode 10: pinned interleave, ORDINARY loads (compiler waits), ring of 3", w, out, cyc, nkc, iters);
      run10<4>("mode 10: pinned interleave, ORDINARY loads (compiler waits), ring of 4", w, out, cyc, nkc, iters);
      run7v<3, true, false>("mode 7 + 64-bit per-lane addresses, ring of 3", w, out, cyc, nkc, iters);
      run7v<3, false, true>("mode 7 + tile groups of 4, in-place reload half a block late, ring of 3", w, out, cyc, nkc, iters);
      run7v<3, true, true>("mode 7 + both, ring of 3", w, out, cyc, nkc, iters);
      run7<3>("mode 7: pinned schedule, 1 asm load per 4 MFMAs, ring of 3, vmcnt(8)", w, out, cyc, nkc, iters);
      run7<4>("mode 7: pinned schedule, 1 asm load per 4 MFMAs, ring of 4, vmcnt(16)", w, out, cyc, nkc, iters);
      run7<6>("mode 7: pinned schedule, 1 asm load per 4 MFMAs, ring of 6, vmcnt(32)", w, out, cyc, nkc, iters);
      run8("mode 8: B from a static LDS image, 1 ds_read_b128 per 4 MFMAs (pinned)", out, cyc, nkc, iters);
      run9<2, 8>("mode 9: LDS-DMA ring, loader wave per consumer, 2 slots x 8 KiB per wave", w, out, cyc, nkc, iters);
      run9<3, 8>("mode 9: LDS-DMA ring, 3 slots x 8 KiB per wave", w, out, cyc, nkc, iters);
      run9<4, 4>("mode 9: LDS-DMA ring, 4 slots x 4 KiB per wave", w, out, cyc, nkc, iters);
      run9<6, 4>("mode 9: LDS-DMA ring, 6 slots x 4 KiB per wave", w, out, cyc, nkc, iters);
    }
  return 0;
}
