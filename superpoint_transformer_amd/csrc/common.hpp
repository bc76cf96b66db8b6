// Shared helpers for the gfx950 kernels of libspt_hip.so.
// CDNA4 only: wavefront = 64 lanes, hard-coded.
#pragma once

#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/spt_hip.h"

#define SPT_WAVE 64

namespace spt {

// ---- error channel (thread-local, never throws across the C ABI) ----------
char* last_error_buf();
int fail(int code, const char* fmt, ...);

#define SPT_CHECK_ARG(cond, msg)                                   \
  do {                                                             \
    if (!(cond)) return ::spt::fail(-1, "%s: %s", __func__, msg);  \
  } while (0)

#define SPT_CHECK_LAUNCH()                                                    \
  do {                                                                        \
    hipError_t e__ = hipGetLastError();                                       \
    if (e__ != hipSuccess)                                                    \
      return ::spt::fail(-2, "%s: launch failed: %s", __func__,               \
                         hipGetErrorString(e__));                             \
  } while (0)

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// ---- fused MLP layers over the rows of SEVERAL graphs in one launch ----------------------------
// run r = rows [r0[r], r1[r]) of graph g[r]; blockIdx.y selects the run, every per-graph
// coefficient table is indexed by g[r], per-block / per-wave partial tables are laid out run by
// run.  n == 0: the legacy one-range launch (the kernel's own r0 / r1 / table pointers).
constexpr int FMLP_MAX_RUNS = 16;
struct FmlpRuns {
  int n;
  int g[FMLP_MAX_RUNS];
  int64_t r0[FMLP_MAX_RUNS], r1[FMLP_MAX_RUNS];
};
// partial tables of graph b: records [start[b], start[b] + count[b]) (runs sorted by graph)
struct FmlpGroups {
  int start[FMLP_MAX_RUNS], count[FMLP_MAX_RUNS];
};

// ---- device helpers ---------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ uint64_t lanemask_lt() {
  return (1ull << (threadIdx.x & 63)) - 1ull;
}

template <typename T>
__device__ __forceinline__ T wave_reduce_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float wave_reduce_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

// Inclusive scan across the 64 lanes of a wave.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = __shfl_up(v, o, 64);
    if ((threadIdx.x & 63) >= o) v += t;
  }
  return v;
}

// Grid sizing for streaming kernels: enough workgroups to fill 256 CUs x 8,
// grid-stride the rest (guide: Guideline 11).
static inline int stream_grid(int64_t work_items, int per_block) {
  int64_t b = ceil_div(work_items, per_block);
  const int64_t cap = 256 * 16;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace spt
