// Shared host/device helpers for libl2hmc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

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

// exp(min(dh, 0)) with the reference's NaN semantics: tf.minimum propagates NaN and
// gauge_dynamics.py:609 / utils/dynamics.py:319 then map every non-finite result to 0.
template <typename T>
__device__ __forceinline__ float accept_from_delta(T dh) {
  if (dh != dh) return 0.f;
  const float pr = expf((float)(dh < (T)0 ? dh : (T)0));
  return isfinite(pr) ? pr : 0.f;
}

}  // namespace l2hmc
