// f32 atomic-add throughput of the chip for the attention backward's scatter patterns:
// 900 M lane-atomics (= dk / dv of 7.03 M edges x 128 columns) into a [428 571][192] f32 buffer
// at random rows.  Patterns, per wave-instruction of 64 lanes:
//   0: one row, 64 consecutive floats (256-byte run)
//   1: four rows, 16 consecutive floats each (64-byte runs)           <- the kernels' layout
//   2: sixteen rows, 4 floats each (16-byte runs, stride 16 B inside a 64-byte segment)
//   3: 64 random dwords
//   4: pattern 1 with plain stores instead of atomics
//   5: pattern 0 with plain stores
// Build: hipcc --offload-arch=gfx950 -O3 -o atomic_rate atomic_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ uint32_t hash(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
template <int PAT>
__global__ __launch_bounds__(256) void k(float* buf, int nrows, long ninstr) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
  for (long i = wave; i < ninstr; i += nw) {
    const uint32_t h = hash((uint32_t)i * 2654435761u + 12345u);
    long off;
    if (PAT == 0 || PAT == 5) {
      off = (long)(h % nrows) * 192 + 64 + ((h >> 20) & 1) * 64 + lane;
    } else if (PAT == 1 || PAT == 4) {
      const uint32_t hr = hash(h + (lane >> 4));
      off = (long)(hr % nrows) * 192 + 64 + ((h >> 20) & 7) * 16 + (lane & 15);
    } else if (PAT == 2) {
      const uint32_t hr = hash(h + (lane & 15));
      off = (long)(hr % nrows) * 192 + 64 + ((h >> 20) & 7) * 16 + 4 * (lane >> 4) + ((h >> 24) & 3);
    } else {
      const uint32_t hr = hash(h * 64 + lane);
      off = (long)(hr % nrows) * 192 + (hr >> 24) % 192;
    }
    if (PAT >= 4) buf[off] = 1.0f; else unsafeAtomicAdd(buf + off, 1.0f);
  }
}
template <int PAT>
void run(float* buf, int nrows, long ninstr, const char* name) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    k<PAT><<<256 * 8, 256>>>(buf, nrows, ninstr);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (rep == 2) printf("%-40s %8.3f ms  %7.1f G lane-ops/s\n", name, ms, ninstr * 64 / ms / 1e6);
  }
}
int main() {
  const int nrows = 428571;
  const long ninstr = 900000000L / 64;
  float* buf;
  hipMalloc(&buf, (size_t)nrows * 192 * 4);
  hipMemset(buf, 0, (size_t)nrows * 192 * 4);
  run<0>(buf, nrows, ninstr, "atomic: 1 row x 64 floats");
  run<1>(buf, nrows, ninstr, "atomic: 4 rows x 16 floats");
  run<2>(buf, nrows, ninstr, "atomic: 16 rows x 4 floats (strided)");
  run<3>(buf, nrows, ninstr, "atomic: 64 random dwords");
  run<4>(buf, nrows, ninstr, "store:  4 rows x 16 floats");
  run<5>(buf, nrows, ninstr, "store:  1 row x 64 floats");
  return 0;
}
