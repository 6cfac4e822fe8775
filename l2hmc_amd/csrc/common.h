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

// Optional per-kernel-class timing with HIP events on the launch stream
// (l2hmc_profile_begin/_end; used by bench.py's roofline pass, off otherwise).
enum ProfClass { kProfNone = 0, kProfGemmL1 = 1, kProfGemmL2 = 2, kProfHeads = 3, kProfU1 = 4, kProfFused = 5 };
void prof_before(int cls, hipStream_t stream);
void prof_after(int cls, hipStream_t stream);

// sum across the 64 lanes of a wave; every lane gets the total (fixed tree => deterministic)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
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

// four standard normals from one Philox block (Box-Muller on word pairs)
__device__ __forceinline__ void philox_normal4(const uint32_t c[4], float v[4]) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float u1 = u01_open(c[2 * h]), u2 = u01_open(c[2 * h + 1]);
    const float rad = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.28318530717958647692f * u2, &sn, &cs);
    v[2 * h] = rad * cs;
    v[2 * h + 1] = rad * sn;
  }
}

// exp(min(dh, 0)) with the reference's NaN semantics: tf.minimum propagates NaN and
// gauge_dynamics.py:609 / utils/dynamics.py:319 then map every non-finite result to 0.
template <typename T>
__device__ __forceinline__ float accept_from_delta(T dh) {
  if (dh != dh) return 0.f;
  const float pr = expf((float)(dh < (T)0 ? dh : (T)0));
  return isfinite(pr) ? pr : 0.f;
}

}  // namespace l2hmc
