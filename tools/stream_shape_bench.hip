// Micro-benchmark (diagnostic, not shipped): which structural element of the U(1) stencil kernel (u1_lattice.hip,
// u1_fast_kernel) keeps a read-once / write-once stream at ~4.7 TB/s when an element-wise kernel of the same bytes
// reaches ~6.2 TB/s?  Each variant reads rows x 128 floats and writes as many.
//   0: one 8-byte element per thread, one-shot grid (f = sin-like polynomial of x)
//   1: one 16-byte element per thread, one-shot grid
//   2: 8 bytes per thread, persistent workgroups (grid 2048, grid-stride, next element prefetched)
//   3: = 2 + LDS round trip with two workgroup barriers per element (the stencil's neighbour exchange)
//   4: = 3 but one-shot grid (one group per workgroup, no prefetch)
//   5: 16 bytes per thread + LDS round trip + two barriers, one-shot grid
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ float work(float p) {          // a handful of VALU ops, like the polynomial sincos
  const float q = p * 0.15915494f;
  const float r = p - 6.2831853f * floorf(q + 0.5f);
  const float r2 = r * r;
  return r * (1.f + r2 * (-0.16666667f + r2 * (0.0083333310f + r2 * -0.00019840874f)));
}

__global__ __launch_bounds__(256) void v0(const float2* __restrict__ x, float2* __restrict__ f, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const float2 a = x[i];
    f[i] = make_float2(work(a.x - a.y), work(a.y));
  }
}
__global__ __launch_bounds__(256) void v1(const float4* __restrict__ x, float4* __restrict__ f, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const float4 a = x[i];
    f[i] = make_float4(work(a.x - a.y), work(a.y), work(a.z - a.w), work(a.w));
  }
}
template <bool LDS>
__global__ __launch_bounds__(256) void v2(const float2* __restrict__ x, float2* __restrict__ f, int64_t ngroups) {
  __shared__ float2 xs[2][256];
  __shared__ float sp[2][256];
  const int tid = threadIdx.x, nb = (tid + 1) & 255, nb2 = (tid + 8) & 255;
  int64_t g = blockIdx.x;
  float2 nxt = x[(g < ngroups ? g : ngroups - 1) * 256 + tid];
  int buf = 0;
  for (; g < ngroups; g += gridDim.x, buf ^= 1) {
    const float2 a = nxt;
    const int64_t g2 = g + gridDim.x;
    nxt = x[(g2 < ngroups ? g2 : ngroups - 1) * 256 + tid];
    float s;
    float2 out;
    if (LDS) {
      xs[buf][tid] = a;
      __syncthreads();
      s = work(a.x - a.y - xs[buf][nb].x + xs[buf][nb2].y);
      sp[buf][tid] = s;
      __syncthreads();
      out = make_float2(s - sp[buf][nb], sp[buf][nb2] - s);
    } else {
      out = make_float2(work(a.x - a.y), work(a.y));
    }
    f[g * 256 + tid] = out;
  }
}
__global__ __launch_bounds__(256) void v4(const float2* __restrict__ x, float2* __restrict__ f, int64_t ngroups) {
  __shared__ float2 xs[256];
  __shared__ float sp[256];
  const int tid = threadIdx.x, nb = (tid + 1) & 255, nb2 = (tid + 8) & 255;
  const int64_t g = blockIdx.x;
  const float2 a = x[g * 256 + tid];
  xs[tid] = a;
  __syncthreads();
  const float s = work(a.x - a.y - xs[nb].x + xs[nb2].y);
  sp[tid] = s;
  __syncthreads();
  f[g * 256 + tid] = make_float2(s - sp[nb], sp[nb2] - s);
}
__global__ __launch_bounds__(256) void v5(const float4* __restrict__ x, float4* __restrict__ f, int64_t ngroups) {
  __shared__ float4 xs[256];
  __shared__ float2 sp[256];
  const int tid = threadIdx.x, nb = (tid + 1) & 255, nb2 = (tid + 4) & 255;
  const int64_t g = blockIdx.x;
  const float4 a = x[g * 256 + tid];
  xs[tid] = a;
  __syncthreads();
  const float s0 = work(a.x - a.y - a.z + xs[nb2].y);
  const float s1 = work(a.z - a.w - xs[nb].x + xs[nb2].w);
  sp[tid] = make_float2(s0, s1);
  __syncthreads();
  const float2 m = sp[(tid + 255) & 255], u = sp[(tid + 252) & 255];
  f[g * 256 + tid] = make_float4(s0 - m.y, u.x - s0, s1 - s0, u.y - s1);
}

template <class F>
void timeit(const char* name, double bytes, F launch) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  const int iters = 20;
  for (int i = 0; i < iters; ++i) launch();
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= iters;
  printf("%-78s %.3f ms  %.0f GB/s (%.2f of 8 TB/s)\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
}

int main() {
  const int64_t rows = 1 << 21, D = 128;
  const int64_t nfl = rows * D;
  float *x, *f;
  if (hipMalloc(&x, nfl * 4) != hipSuccess || hipMalloc(&f, nfl * 4) != hipSuccess) return 1;
  (void)hipMemset(x, 0, nfl * 4);
  const double bytes = 2.0 * nfl * 4;
  const int64_t n2 = nfl / 2, n4 = nfl / 4;
  timeit("0: 8 B per thread, one-shot grid", bytes, [&] { v0<<<(unsigned)(n2 / 256), 256>>>((float2*)x, (float2*)f, n2); });
  timeit("1: 16 B per thread, one-shot grid", bytes, [&] { v1<<<(unsigned)(n4 / 256), 256>>>((float4*)x, (float4*)f, n4); });
  timeit("2: 8 B per thread, persistent (2048 workgroups), next element prefetched", bytes,
         [&] { v2<false><<<2048, 256>>>((float2*)x, (float2*)f, n2 / 256); });
  timeit("3: = 2 + LDS round trip and two barriers per element", bytes,
         [&] { v2<true><<<2048, 256>>>((float2*)x, (float2*)f, n2 / 256); });
  timeit("4: 8 B per thread + LDS round trip and two barriers, one-shot grid", bytes,
         [&] { v4<<<(unsigned)(n2 / 256), 256>>>((float2*)x, (float2*)f, n2 / 256); });
  timeit("5: 16 B per thread + LDS round trip and two barriers, one-shot grid", bytes,
         [&] { v5<<<(unsigned)(n4 / 256), 256>>>((float4*)x, (float4*)f, n4 / 256); });
  return 0;
}
