// Shared host/device helpers for libl2hmc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>

#include "../../include/l2hmc_hip.h"

namespace l2hmc {

void set_error(const char* fmt, ...);

#define L2HMC_REQUIRE(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      ::l2hmc::set_error(__VA_ARGS__);           \
      return L2HMC_ERR_ARG;                      \
    }                                            \
  } while (0)

#define L2HMC_CHECK_LAUNCH(what)                                              \
  do {                                                                        \
    hipError_t e_ = hipGetLastError();                                        \
    if (e_ != hipSuccess) {                                                   \
      ::l2hmc::set_error("%s: %s", what, hipGetErrorString(e_));              \
      return L2HMC_ERR_HIP;                                                   \
    }                                                                         \
  } while (0)

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t hmin(int64_t a, int64_t b) { return a < b ? a : b; }
static inline int hmax(int a, int b) { return a > b ? a : b; }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

constexpr int kWave = 64;  // CDNA wavefront

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device only: a launch site keeps one
// of these and repeats the opt-in the first time it runs on each device (bit per ordinal; a race between two
// threads repeats an idempotent call, nothing worse).
class DeviceOnce {
  std::atomic<uint64_t> mask_[4] = {};      // 256 device ordinals
  static bool slot(int* word, uint64_t* bit) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 256) return false;
    *word = d >> 6;
    *bit = 1ull << (d & 63);
    return true;
  }
 public:
  bool pending() const {                     // true: the opt-in has not been made on this device yet
    int w; uint64_t b;
    if (!slot(&w, &b)) return true;
    return (mask_[w].load(std::memory_order_acquire) & b) == 0;
  }
  void done() {
    int w; uint64_t b;
    if (slot(&w, &b)) mask_[w].fetch_or(b, std::memory_order_release);
  }
};

// ---- XCD-aware tile id: blocks b, b+8, ... share an XCD, give each XCD a
// contiguous run of logical tiles (bijective for any grid size).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, slot = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

// Optional per-kernel-class timing with HIP events on the launch stream
// (l2hmc_profile_begin/_end; used by bench.py's roofline pass, off otherwise).
enum ProfClass { kProfNone = 0, kProfGemmL1 = 1, kProfGemmL2 = 2, kProfHeads = 3, kProfU1 = 4, kProfFused = 5,
                 kProfConvFront = 6, kProfSmall = 7, kProfLast = kProfSmall };
void prof_before(int cls, hipStream_t stream);
void prof_after(int cls, hipStream_t stream);

// sum across the 64 lanes of a wave; every lane gets the total (fixed order => deterministic).  Six DPP steps on
// the VALU -- quad swaps, half-row and row mirrors, then the two row broadcasts that fold the four 16-lane rows into
// lane 63 -- and one v_readlane: no LDS-crossbar traffic, where the __shfl_xor butterfly compiles to six dependent
// ds_bpermute_b32 round trips (what kept the observable-producing stencil kernels at half the HBM rate).
// ---------------------------------------------------------------------------------------------------------------
// Streaming loads next to MFMAs: BUFFER loads.  The section's base sits in a scalar buffer resource, a wave-uniform
// byte offset in a scalar register, and the lane supplies ONE 32-bit offset
// (buffer_load_dwordx4 v, v_off, s[rsrc], s_off offen).  With a 64-bit per-lane address instead (global_load with two
// address registers per lane -- what the compiler makes of pointer arithmetic, even on a provably uniform base) every
// load costs the matrix pipe ~3 cycles per v_mfma_f32_16x16x4_f32 more: 37.5 against 34.8 cycles per MFMA in
// tools/mfma_pinned_bench.hip (profiles/r04_mfma_pinned_bench.txt) -- this, more than the load's placement, is the
// "issue cost" rounds 2-3 measured.  Reads beyond the resource's 2 GiB window return 0 (never a fault); the lane and
// scalar offsets must stay below 2^31 bytes (callers base the resource at their tile).
using u32x4_t = __attribute__((ext_vector_type(4))) unsigned;
using f32x4_t = __attribute__((ext_vector_type(4))) float;
struct WSection {
  __amdgpu_buffer_rsrc_t rs;
};
__device__ __forceinline__ WSection wsection(const float* __restrict__ base) {
  return {__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000)};   // raw, stride 0
}
__device__ __forceinline__ f32x4_t buf_load16(const WSection& ws, unsigned lane_off_bytes, unsigned uniform_off_bytes) {
  return __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(ws.rs, lane_off_bytes, uniform_off_bytes, 0));
}

// The same source as a plain descriptor (four scalar registers) for the loads that go STRAIGHT to LDS: issued by
// hand, because the compiler orders every LDS read behind a load-to-LDS it knows of (it cannot tell the buffer being
// filled from the one being read) -- the caller waits with s_waitcnt vmcnt(0) before the barrier that publishes the tile.
struct WDesc {
  u32x4_t d;
};
__device__ __forceinline__ WDesc wdesc(const float* __restrict__ base) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  return {u32x4_t{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a),
                  (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu)), 0x7fffffffu, 0x00020000u}};
}
// 16 bytes per lane from base + lane_off_bytes + uniform_off_bytes to LDS byte address lds_addr (wave-uniform) + 16 * lane
__device__ __forceinline__ void lds_dma16(const WDesc& w, unsigned lds_addr, unsigned lane_off_bytes, unsigned uniform_off_bytes) {
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_addr), "v"(lane_off_bytes), "s"(w.d), "s"(uniform_off_bytes)
               : "memory");      // (m0 is reserved: the compiler keeps nothing in it)
}
__device__ __forceinline__ unsigned lds_byte_address(const float* p) {
  return (unsigned)reinterpret_cast<unsigned long>((const __attribute__((address_space(3))) float*)p);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_take(float v) {          // lanes of rows outside ROW_MASK receive 0
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_take<0xB1, 0xf>(v);        // quad_perm [1,0,3,2]
  v += dpp_take<0x4E, 0xf>(v);        // quad_perm [2,3,0,1]
  v += dpp_take<0x141, 0xf>(v);       // row_half_mirror
  v += dpp_take<0x140, 0xf>(v);       // row_mirror: every lane holds its row's sum
  v += dpp_take<0x142, 0xa>(v);       // row_bcast:15 into rows 1 and 3
  v += dpp_take<0x143, 0xc>(v);       // row_bcast:31 into rows 2 and 3: lane 63 holds the total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// sum over each 16-lane row (lanes 16 k .. 16 k + 15), returned in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_take<0xB1, 0xf>(v);
  v += dpp_take<0x4E, 0xf>(v);
  v += dpp_take<0x141, 0xf>(v);
  v += dpp_take<0x140, 0xf>(v);
  return v;
}

// sums over lanes 0..31 and 32..63 separately: valid in lanes 31 and 63 of the returned value
__device__ __forceinline__ float wave_half_sums(float v) {
  v += dpp_take<0xB1, 0xf>(v);
  v += dpp_take<0x4E, 0xf>(v);
  v += dpp_take<0x141, 0xf>(v);
  v += dpp_take<0x140, 0xf>(v);
  v += dpp_take<0x142, 0xa>(v);       // row_bcast:15: lanes of row 1 hold rows 0 + 1, lanes of row 3 rows 2 + 3
  return v;
}

// Philox4x32-10 (Salmon et al., SC'11) and the word -> float maps shared by every generator in the library
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += W0;
    k1 += W1;
  }
}

__device__ __forceinline__ float u01_open(uint32_t w) {   // (0, 1): 24 bits, centred
  return ((float)(w >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

// sin and cos of an angle in radians to ~1e-7 absolute (the accuracy of libm's sincosf, which the 1e-5 bar needs;
// v_sin_f32 / v_cos_f32 alone are ~1e-6): Cody-Waite reduction by pi/2 in three parts, then the degree-7 / 8
// minimax polynomials on [-pi/4, pi/4].  ~25 VALU instructions against ~110 for sincosf with its large-argument
// path; the stencil kernels were VALU-bound on exactly that (profiles/r02_u1_*).  |x| >= 8192 (never reached by
// wrapped links, possible for an arbitrary caller) falls back to libm.
__device__ __forceinline__ void fast_sincos(float x, float* sn, float* cs) {
  if (__builtin_expect(!(fabsf(x) < 8192.f), 0)) {
    sincosf(x, sn, cs);
    return;
  }
  const float n = rintf(x * 0.63661977236758134308f);
  float r = fmaf(n, -1.5703125f, x);
  r = fmaf(n, -4.837512969970703125e-4f, r);
  r = fmaf(n, -7.54978995489188216e-8f, r);
  const float z = r * r;
  const float s = fmaf(r * z, fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), r);
  const float c = fmaf(z * z, fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f),
                       fmaf(-0.5f, z, 1.f));
  const int q = (int)n;
  const float a = (q & 1) ? c : s, b = (q & 1) ? s : c;
  *sn = (q & 2) ? -a : a;
  *cs = ((q + 1) & 2) ? -b : b;
}

// four standard normals from one Philox block (Box-Muller on word pairs)
__device__ __forceinline__ void philox_normal4(const uint32_t c[4], float v[4]) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float u1 = u01_open(c[2 * h]), u2 = u01_open(c[2 * h + 1]);
    const float rad = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    fast_sincos(6.28318530717958647692f * u2, &sn, &cs);
    v[2 * h] = rad * cs;
    v[2 * h + 1] = rad * sn;
  }
}

// exp / tanh on the hardware exp2 + rcp units (v_exp_f32, v_rcp_f32: ~1 ulp each).  Arguments here are
// eps * S, eps * Q and pre-activations of O(1): |error| <= ~2e-7 relative for exp, ~1.5e-7 absolute for
// tanh -- at the fp32 rounding floor of the quantities they feed, and ~8x cheaper than the libm forms.
#ifdef L2HMC_EXACT_MATH   // diagnostic build only (tools/build_exact.sh): libm forms, to price the hardware forms' error
__device__ __forceinline__ float fast_exp(float x) { return expf(x); }
__device__ __forceinline__ float fast_tanh(float x) { return tanhf(x); }
#else
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float fast_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(-2.88539008177792681f * fabsf(x));      // exp(-2|x|) in (0, 1]
  const float t = (1.f - e) * __builtin_amdgcn_rcpf(1.f + e);
  return copysignf(t, x);
}
#endif

// exp(min(dh, 0)) with the reference's NaN semantics: tf.minimum propagates NaN and
// gauge_dynamics.py:609 / utils/dynamics.py:319 then map every non-finite result to 0.
template <typename T>
__device__ __forceinline__ float accept_from_delta(T dh) {
  if (dh != dh) return 0.f;
  const float pr = expf((float)(dh < (T)0 ? dh : (T)0));
  return isfinite(pr) ? pr : 0.f;
}

}  // namespace l2hmc
