// Error plumbing of the C ABI + the counter-based RNG.
#include "common.h"
#include <string.h>
#include <mutex>
#include <utility>
#include <vector>

namespace l2hmc {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---------------------------------------------------------------------
// per-kernel-class event timing (host side only; never active unless asked)
// ---------------------------------------------------------------------
// One process-wide recorder (a measurement aid, documented as such in the header); a mutex keeps concurrent
// launching threads from corrupting the event list -- the disarmed fast path is one relaxed atomic load.
static std::atomic<int> g_prof_cls{kProfNone};
static std::mutex g_prof_mu;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_events;
static size_t g_prof_used = 0;
static thread_local bool g_prof_open = false;      // this thread recorded a start event that awaits its stop

void prof_before(int cls, hipStream_t stream) {
  if (cls != g_prof_cls.load(std::memory_order_relaxed)) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_open = false;
  if (g_prof_used == g_prof_events.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
    g_prof_events.emplace_back(a, b);
  }
  (void)hipEventRecord(g_prof_events[g_prof_used].first, stream);
  g_prof_open = true;
}

void prof_after(int cls, hipStream_t stream) {
  if (cls != g_prof_cls.load(std::memory_order_relaxed)) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (g_prof_open && g_prof_used < g_prof_events.size()) (void)hipEventRecord(g_prof_events[g_prof_used++].second, stream);
  g_prof_open = false;
}

// ---------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11).  Block b of the stream is
// philox(counter = {b_lo, b_hi, offset_lo, offset_hi}, key = {seed_lo, seed_hi});
// element i of a fill is word (i & 3) of block (i >> 2).
// ---------------------------------------------------------------------
template <bool NORMAL>
__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ out, int64_t n, uint64_t seed,
                                                   uint64_t offset) {
  const int64_t nblk = (n + 3) >> 2;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblk;
       b += (int64_t)gridDim.x * blockDim.x) {
    uint32_t c[4] = {(uint32_t)b, (uint32_t)((uint64_t)b >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    float v[4];
    if (NORMAL) {
      philox_normal4(c, v);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (float)(c[j] >> 8) * (1.0f / 16777216.0f);   // [0, 1)
    }
    const int64_t i0 = b << 2;
    if (i0 + 3 < n && ((reinterpret_cast<uintptr_t>(out) & 15) == 0)) {
      *reinterpret_cast<float4*>(out + i0) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      for (int j = 0; j < 4 && i0 + j < n; ++j) out[i0 + j] = v[j];
    }
  }
}

template <bool NORMAL>
static int launch_fill(float* out, int64_t n, uint64_t seed, uint64_t offset, hipStream_t s) {
  L2HMC_REQUIRE(n >= 0, "fill: n < 0");
  if (n == 0) return L2HMC_OK;
  L2HMC_REQUIRE(out != nullptr, "fill: out is NULL");
  const int64_t nblk = (n + 3) >> 2;
  const int64_t grid = hmin(ceil_div(nblk, 256), 4096);
  hipLaunchKernelGGL(fill_kernel<NORMAL>, dim3((unsigned)grid), dim3(256), 0, s, out, n, seed, offset);
  L2HMC_CHECK_LAUNCH("fill");
  return L2HMC_OK;
}

}  // namespace l2hmc

using namespace l2hmc;

extern "C" int l2hmc_abi_version(void) { return L2HMC_ABI_VERSION; }
extern "C" const char* l2hmc_last_error(void) { return g_err; }

extern "C" int l2hmc_profile_begin(int32_t kernel_class) {
  L2HMC_REQUIRE(kernel_class >= kProfNone && kernel_class <= kProfLast, "profile_begin: unknown class %d",
                kernel_class);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_cls.store(kernel_class, std::memory_order_relaxed);
  g_prof_used = 0;
  return L2HMC_OK;
}

extern "C" int l2hmc_profile_end(double* total_ms, int64_t* launches) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_cls.store(kProfNone, std::memory_order_relaxed);
  double tot = 0.0;
  for (size_t i = 0; i < g_prof_used; ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(g_prof_events[i].second) != hipSuccess ||
        hipEventElapsedTime(&ms, g_prof_events[i].first, g_prof_events[i].second) != hipSuccess) {
      set_error("profile_end: event query failed");
      return L2HMC_ERR_HIP;
    }
    tot += ms;
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = (int64_t)g_prof_used;
  g_prof_used = 0;
  return L2HMC_OK;
}

extern "C" int l2hmc_fill_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, l2hmc_stream_t stream) {
  return launch_fill<true>(out, n, seed, offset, (hipStream_t)stream);
}
extern "C" int l2hmc_fill_uniform(float* out, int64_t n, uint64_t seed, uint64_t offset, l2hmc_stream_t stream) {
  return launch_fill<false>(out, n, seed, offset, (hipStream_t)stream);
}
