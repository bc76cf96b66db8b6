// Does v_mfma_f32_16x16x4_f32 overlap with f32 VALU work of a co-resident wave
// on the same SIMD?  512-thread blocks = 2 waves per SIMD.
//   mode 0: all waves MFMA      mode 1: all waves VALU
//   mode 2: waves 0-3 MFMA, waves 4-7 VALU (one of each per SIMD)
//   mode 3: every wave alternates 16 MFMA / 64 VALU (in-wave interleave)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(int mode, int iters, float* out) {
  const int wave = threadIdx.x >> 6;
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  float v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3, v4 = a + 4, v5 = a + 5, v6 = a + 6, v7 = a + 7;
  const bool do_mfma = mode == 0 || (mode == 2 && wave < 4) || mode == 3;
  const bool do_valu = mode == 1 || (mode == 2 && wave >= 4) || mode == 3;
  for (int i = 0; i < iters; ++i) {
    if (do_mfma) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
      }
    }
    if (do_valu) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v0 = fmaf(v0, b, a); v1 = fmaf(v1, b, a); v2 = fmaf(v2, b, a); v3 = fmaf(v3, b, a);
        v4 = fmaf(v4, b, a); v5 = fmaf(v5, b, a); v6 = fmaf(v6, b, a); v7 = fmaf(v7, b, a);
      }
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}
int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int mode = 0; mode < 4; ++mode) {
    k<<<256, 512>>>(mode, 100, out);
    hipEventRecord(e0);
    k<<<256, 512>>>(mode, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d: %.3f ms  (16 MFMA + 64 VALU-fma per iter per wave where enabled)\n", mode, ms);
  }
  return 0;
}
