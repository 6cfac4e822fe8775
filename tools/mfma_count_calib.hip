// Calibration (diagnostic, not shipped): a kernel that issues an EXACTLY known number of
// v_mfma_f32_16x16x4_f32 so that the SQ MFMA counters read under `rocprofv3 --pmc` can be converted to
// instruction counts on this chip: 1024 workgroups-waves x ITER x 32 MFMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ __launch_bounds__(256) void calib_mfma_kernel(float* out, int iters) {
  f32x4 acc[8];
  for (int t = 0; t < 8; ++t) acc[t] = f32x4{0, 0, 0, 0};
  const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
#pragma nounroll
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b + t, acc[t], 0, 0, 0);
  }
  float s = 0;
  for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 256 * 4);
  const int iters = 10000;                       // per wave: 10000 x 32 = 320000 MFMAs; 1024 waves -> 3.2768e8
  for (int rep = 0; rep < 3; ++rep) calib_mfma_kernel<<<256, 256>>>(out, iters);
  hipDeviceSynchronize();
  printf("calib_mfma_kernel: 3 launches x %.0f v_mfma_f32_16x16x4_f32 (= %.4e FLOP) each\n", 1024.0 * iters * 32,
         1024.0 * iters * 32 * 2048);
  return 0;
}
