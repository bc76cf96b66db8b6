// Ceiling of the segment-max access pattern on this chip: 15 M rows of 512 B read
// through a random permutation (one wave-half per row, float4 per lane, non-temporal),
// nothing written - vs the same rows read in order.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <random>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void rd(const float* __restrict__ x, const int* __restrict__ perm,
                                          long n, float* out, int use_perm) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
  v4f acc = {0, 0, 0, 0};
  // 2 rows per wave-instruction (32 lanes x 16 B = 512 B), 8 in flight
  for (long r0 = wave * 16; r0 < n; r0 += nw * 16) {
    v4f v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long j = r0 + u * 2 + (lane >> 5);
      const long r = j < n ? (use_perm ? perm[j] : j) : 0;
      v[u] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(x + r * 128 + (lane & 31) * 4));
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = acc + v[u];   // stands in for the max
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[0] = 1.f;
}
int main() {
  const long n = 15000000;
  float* x; int* perm; float* out;
  hipMalloc(&x, n * 512); hipMalloc(&perm, n * 4); hipMalloc(&out, 4);
  hipMemset(x, 0, n * 512);
  std::vector<int> p(n);
  for (long i = 0; i < n; ++i) p[i] = (int)i;
  std::mt19937 g(1);
  std::shuffle(p.begin(), p.end(), g);
  hipMemcpy(perm, p.data(), n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int use_perm = 0; use_perm < 2; ++use_perm)
    for (int grid : {2048, 4096, 8192}) {
      rd<<<grid, 256>>>(x, perm, n, out, use_perm);
      hipEventRecord(e0);
      for (int i = 0; i < 5; ++i) rd<<<grid, 256>>>(x, perm, n, out, use_perm);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
      printf("%s rows, grid %d: %.3f ms  %.0f GB/s\n", use_perm ? "permuted" : "in-order", grid, ms,
             n * 516.0 / ms / 1e6);
    }
  return 0;
}
