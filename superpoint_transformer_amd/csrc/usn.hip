// Fused UnitSphereNorm (src/nn/norm.py:67-138).
//
// The reference runs scatter-min, scatter-max, scatter-mean (or cat + weighted
// scatter-sum, src/utils/scatter.py:17-38), two gathers and the elementwise
// normalisation as ~7 launches over `pos`.  Here: one segment pass over the
// CSR view producing (center, diameter) per segment, one coalesced row pass
// applying them.  Bounding boxes are bit-exact (min/max are order-free);
// centres accumulate in f64, so they are at least as accurate as the f32
// scatter they replace.
#include <math.h>
#include <stdlib.h>

#include "common.hpp"

namespace spt {

struct Box {
  float mn[3], mx[3];
  double sw, sx[3];
};

__device__ __forceinline__ void box_init(Box& b) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    b.mn[k] = INFINITY;
    b.mx[k] = -INFINITY;
    b.sx[k] = 0.0;
  }
  b.sw = 0.0;
}

__device__ __forceinline__ void box_add(Box& b, float x, float y, float z, double w) {
  b.mn[0] = fminf(b.mn[0], x); b.mx[0] = fmaxf(b.mx[0], x);
  b.mn[1] = fminf(b.mn[1], y); b.mx[1] = fmaxf(b.mx[1], y);
  b.mn[2] = fminf(b.mn[2], z); b.mx[2] = fmaxf(b.mx[2], z);
  b.sw += w;
  b.sx[0] += w * (double)x;
  b.sx[1] += w * (double)y;
  b.sx[2] += w * (double)z;
}

__device__ __forceinline__ void box_merge_xor(Box& b, int o) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    b.mn[k] = fminf(b.mn[k], __shfl_xor(b.mn[k], o, 64));
    b.mx[k] = fmaxf(b.mx[k], __shfl_xor(b.mx[k], o, 64));
    b.sx[k] += __shfl_xor(b.sx[k], o, 64);
  }
  b.sw += __shfl_xor(b.sw, o, 64);
}

__device__ __forceinline__ double row_weight(const float* wf, const int64_t* wi, int64_t r) {
  if (wi) return (double)(float)wi[r];  // reference: w.float() (scatter.py:26)
  if (wf) return (double)wf[r];
  return 1.0;
}

// cd4 (nullable): the same four values once more as ONE 16-byte record [cx, cy, cz, diameter] per
// segment - the assemble pass gathers them per row, and two gathers (12 + 4 bytes from two arrays)
// are two 64-byte sectors per row where the tables do not fit an XCD's L2 (round 6)
__device__ __forceinline__ void box_finish(const Box& b, int cnt, float* center, float* diam,
                                           float4* cd4 = nullptr) {
  // norm.py:118-126: empty segment -> min = max = 0 -> diameter 0, centre 0;
  // weighted mean divides by the weight sum, 0 replaced by 1 (scatter.py:35)
  float d = 0.f;
  if (cnt > 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) d = fmaxf(d, b.mx[k] - b.mn[k]);
  }
  const double den = (b.sw == 0.0) ? 1.0 : b.sw;
#pragma unroll
  for (int k = 0; k < 3; ++k) center[k] = (float)(b.sx[k] / den);
  *diam = d;
  if (cd4) *cd4 = make_float4(center[0], center[1], center[2], d);
}

// One lane group of 2^g_log2 lanes per segment.
__global__ __launch_bounds__(256) void usn_stats_group_kernel(
    const float* __restrict__ pos, const int32_t* __restrict__ perm,
    const int32_t* __restrict__ rowptr, const float* __restrict__ wf,
    const int64_t* __restrict__ wi, int64_t num_seg, int g_log2,
    float* __restrict__ center, float* __restrict__ diam, float4* __restrict__ cd4 = nullptr) {
  const int lane = threadIdx.x & 63;
  const int g = 1 << g_log2;
  const int spw = 64 >> g_log2;
  const int slot = lane >> g_log2;
  const int lg = lane & (g - 1);
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t sbase = wave * spw; sbase < num_seg; sbase += nwaves * spw) {
    const int64_t s = sbase + slot;
    const bool sv = s < num_seg;
    const int start = sv ? rowptr[s] : 0;
    const int end = sv ? rowptr[s + 1] : 0;
    Box b;
    box_init(b);
    // four rows of the lane's stride per trip: the four row ids are requested together, then the
    // twelve coordinates (round 5: the loop was perm -> pos one row at a time, two dependent
    // round trips per row with ~2 rows per lane - pure latency; same rows, same order of adds)
    for (int j0 = start + lg; j0 < end; j0 += 4 * g) {
      int64_t r[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + u * g;
        ok[u] = j < end;
        const int jc = ok[u] ? j : start;                     // (a valid position: unconditional load)
        r[u] = perm ? perm[jc] : jc;
      }
      float px[4], py[4], pz[4];
      double pw[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        px[u] = pos[r[u] * 3];
        py[u] = pos[r[u] * 3 + 1];
        pz[u] = pos[r[u] * 3 + 2];
        pw[u] = row_weight(wf, wi, r[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (ok[u]) box_add(b, px[u], py[u], pz[u], pw[u]);
    }
    for (int o = 1; o < g; o <<= 1) box_merge_xor(b, o);
    if (sv && lg == 0) box_finish(b, end - start, center + s * 3, diam + s, cd4 ? cd4 + s : nullptr);
  }
}

// One workgroup per segment (few, huge segments: idx=None / per-cloud norms).
__global__ __launch_bounds__(256) void usn_stats_block_kernel(
    const float* __restrict__ pos, const int32_t* __restrict__ perm,
    const int32_t* __restrict__ rowptr, const float* __restrict__ wf,
    const int64_t* __restrict__ wi, int64_t num_seg,
    float* __restrict__ center, float* __restrict__ diam) {
  __shared__ Box part[4];
  for (int64_t s = blockIdx.x; s < num_seg; s += gridDim.x) {
    const int start = rowptr[s], end = rowptr[s + 1];
    Box b;
    box_init(b);
    for (int j = start + threadIdx.x; j < end; j += blockDim.x) {
      const int64_t r = perm ? perm[j] : j;
      box_add(b, pos[r * 3], pos[r * 3 + 1], pos[r * 3 + 2], row_weight(wf, wi, r));
    }
    for (int o = 1; o < 64; o <<= 1) box_merge_xor(b, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < 4; ++w) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          b.mn[k] = fminf(b.mn[k], part[w].mn[k]);
          b.mx[k] = fmaxf(b.mx[k], part[w].mx[k]);
          b.sx[k] += part[w].sx[k];
        }
        b.sw += part[w].sw;
      }
      box_finish(b, end - start, center + s * 3, diam + s);
    }
  }
}

// Few, HUGE segments (idx = None at the top level: one segment of 178 571 nodes at scene S) with a
// caller workspace: blockIdx.x = slice of the segment, blockIdx.y = segment; every block writes its
// partial box, a second kernel merges the P partials of a segment in slice order (deterministic).
// The one-block-per-segment kernel above walks such a segment with 256 threads: 0.40 ms at scene S.
__global__ __launch_bounds__(256) void usn_stats_split_kernel(
    const float* __restrict__ pos, const int32_t* __restrict__ perm,
    const int32_t* __restrict__ rowptr, const float* __restrict__ wf,
    const int64_t* __restrict__ wi, int P, Box* __restrict__ part_out) {
  __shared__ Box part[4];
  const int64_t s = blockIdx.y;
  const int64_t start = rowptr[s], end = rowptr[s + 1];
  const int64_t chunk = (end - start + P - 1) / P;
  const int64_t a = start + (int64_t)blockIdx.x * chunk;
  const int64_t b_end = (a + chunk < end) ? a + chunk : end;
  Box b;
  box_init(b);
  for (int64_t j = a + threadIdx.x; j < b_end; j += blockDim.x) {
    const int64_t r = perm ? perm[j] : j;
    box_add(b, pos[r * 3], pos[r * 3 + 1], pos[r * 3 + 2], row_weight(wf, wi, r));
  }
  for (int o = 1; o < 64; o <<= 1) box_merge_xor(b, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = b;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        b.mn[k] = fminf(b.mn[k], part[w].mn[k]);
        b.mx[k] = fmaxf(b.mx[k], part[w].mx[k]);
        b.sx[k] += part[w].sx[k];
      }
      b.sw += part[w].sw;
    }
    part_out[s * P + blockIdx.x] = b;
  }
}
__global__ __launch_bounds__(256) void usn_stats_merge_kernel(
    const Box* __restrict__ part, const int32_t* __restrict__ rowptr, int64_t num_seg, int P,
    float* __restrict__ center, float* __restrict__ diam) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= num_seg) return;
  Box b = part[s * P];
  for (int p = 1; p < P; ++p) {
    const Box q = part[s * P + p];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      b.mn[k] = fminf(b.mn[k], q.mn[k]);
      b.mx[k] = fmaxf(b.mx[k], q.mx[k]);
      b.sx[k] += q.sx[k];
    }
    b.sw += q.sw;
  }
  box_finish(b, rowptr[s + 1] - rowptr[s], center + s * 3, diam + s);
}

// out[i] = (pos[i] - center[idx[i]]) / (diam[idx[i]] + 1e-2)    norm.py:132-136
__global__ __launch_bounds__(256) void usn_apply_kernel(
    const float* __restrict__ pos, const int64_t* __restrict__ idx,
    const float* __restrict__ center, const float* __restrict__ diam, int64_t n,
    float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int64_t s = idx ? idx[i] : 0;
    const float d = diam[s] + 1e-2f;
    out[i * 3 + 0] = (pos[i * 3 + 0] - center[s * 3 + 0]) / d;
    out[i * 3 + 1] = (pos[i * 3 + 1] - center[s * 3 + 1]) / d;
    out[i * 3 + 2] = (pos[i * 3 + 2] - center[s * 3 + 2]) / d;
  }
}

// out[i] = [diam[idx[i]] | (pos[i] - center[idx[i]]) / (diam[idx[i]] + 1e-2) | x[i]]: the stage input
// [diameter_parent | normalized_pos | x] of src/nn/stage.py:249-271 written in ONE pass (the
// normalised positions, the gathered diameter and the copy of x never exist on their own).
// One thread per 16-byte chunk of the output (C = 4 + cx floats per row, cx % 4 == 0): a wave
// writes 1 KB contiguously; chunk 0 of a row is formed from (idx, pos, center, diam), the others
// are copies of x.  (A wave-per-64-rows variant with all header loads in one lane group measured
// slower: two dependent round trips per block with nothing else in flight.)
struct f3 { float x, y, z; };
// CC4 > 0: the row's chunk count 1 + cx / 4 as a compile-time constant (3: the 8 point features of
// level 0, 17 / 33: 64 / 128 segment features) - the chunk -> (row, chunk of row) split is then a
// multiply-shift instead of a 20-instruction integer division per thread.
template <typename I, int CC4 = 0>
__global__ __launch_bounds__(256) void usn_assemble_kernel(
    const float* __restrict__ pos, const int64_t* __restrict__ idx,
    const float* __restrict__ center, const float* __restrict__ diam,
    const float* __restrict__ x, int cx4, int64_t n, float* __restrict__ out) {
  const I c4 = CC4 > 0 ? (I)CC4 : (I)(cx4 + 1);
  const I total = (I)n * c4;
  const I stride = (I)gridDim.x * blockDim.x;
  for (I q = (I)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += stride) {
    const I i = q / c4;
    const int j = (int)(q - i * c4);
    float4 v;
    if (j == 0) {
      const int64_t s = idx ? idx[i] : 0;
      const f3 p = *reinterpret_cast<const f3*>(pos + (int64_t)i * 3);
      const f3 ce = *reinterpret_cast<const f3*>(center + s * 3);
      const float dm = diam[s], d = dm + 1e-2f;
      v = make_float4(dm, (p.x - ce.x) / d, (p.y - ce.y) / d, (p.z - ce.z) / d);
    } else {
      v = *(reinterpret_cast<const float4*>(x) + ((int64_t)i * cx4 + (j - 1)));
    }
    *(reinterpret_cast<float4*>(out) + (int64_t)q) = v;
  }
}

// Round 6: the same rows, 256 (wide rows: 64 / 32) at a time per workgroup in two phases.  Phase 1: one THREAD per row
// forms the row's header chunk [diam | normalised pos] (idx -> centre / diameter: two dependent
// round trips, all 256 lanes busy with them instead of every third one) into LDS; phase 2: one
// thread per 16-byte output chunk as before - the header out of LDS, the copies of x from loads
// that were issued BEFORE phase 1 (they travel under its two round trips).  Same values bit for
// bit (the same expressions per element).
template <int CC4>
__global__ __launch_bounds__(256) void usn_assemble_rows_kernel(
    const float* __restrict__ pos, const int64_t* __restrict__ idx,
    const float* __restrict__ center, const float* __restrict__ diam,
    const float* __restrict__ x, int64_t n, float* __restrict__ out,
    const float4* __restrict__ cd4 = nullptr) {
  constexpr int R = CC4 <= 4 ? 256 : (CC4 <= 20 ? 64 : 32), CX4 = CC4 - 1;   // ~4 chunks per thread
  constexpr int PER = (R * CC4 + 255) / 256;             // output chunks per thread and block of rows
  __shared__ float4 hdr[R];
  const int t = threadIdx.x;
  const int64_t nblk = (n + R - 1) / R;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int64_t row0 = blk * R;
    const int rows = (int)((n - row0) < R ? (n - row0) : R);
    // the x chunks of this thread's output chunks: requested first
    float4 xv[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int q = t + 256 * k;
      const int r = q / CC4, j = q - r * CC4;
      // (unconditional: a header chunk / a chunk past the end reads a valid chunk it never uses)
      const int rc = r < rows ? r : rows - 1, jc = j > 0 ? j - 1 : 0;
      xv[k] = *(reinterpret_cast<const float4*>(x) + ((row0 + rc) * CX4 + jc));
    }
    if (t < rows) {
      const int64_t i = row0 + t;
      const int64_t sg = idx ? idx[i] : 0;
      const f3 p = *reinterpret_cast<const f3*>(pos + i * 3);
      f3 ce;
      float dm;
      if (cd4) {                                         // one 16-byte gather per row
        const float4 t4 = cd4[sg];
        ce.x = t4.x; ce.y = t4.y; ce.z = t4.z;
        dm = t4.w;
      } else {
        ce = *reinterpret_cast<const f3*>(center + sg * 3);
        dm = diam[sg];
      }
      const float d = dm + 1e-2f;
      hdr[t] = make_float4(dm, (p.x - ce.x) / d, (p.y - ce.y) / d, (p.z - ce.z) / d);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int q = t + 256 * k;
      const int r = q / CC4, j = q - r * CC4;
      if (q < rows * CC4) {
        float4 v = xv[k];
        if (j == 0) v = hdr[r];                          // (values, not a select between two address spaces)
        *(reinterpret_cast<float4*>(out) + (row0 * CC4 + q)) = v;
      }
    }
    __syncthreads();                                     // hdr is rewritten by the next block of rows
  }
}

}  // namespace spt

using namespace spt;

// slices per segment of the split statistics (few huge segments): ~2 048 rows per block, the whole
// launch within 4 096 blocks and 65 535 segments
// SPT_USN_ROWS=0: the one-thread-per-chunk assemble kernel of round 3 (A/B switch)
static bool usn_rows_kernel() {
  static const bool on = [] { const char* e = getenv("SPT_USN_ROWS"); return e ? atoi(e) != 0 : true; }();
  return on;
}
static int usn_slices(int64_t n, int64_t num_seg) {
  if (num_seg < 1 || num_seg > 65535) return 1;
  const int64_t avg = n / num_seg;
  int64_t p = ceil_div(avg, (int64_t)2048);
  const int64_t cap = 4096 / num_seg > 1 ? 4096 / num_seg : 1;
  if (p > cap) p = cap;
  return (int)(p < 1 ? 1 : p);
}
extern "C" size_t spt_unit_sphere_workspace_bytes(int64_t n, int64_t num_seg) {
  return (size_t)(num_seg > 0 ? num_seg : 1) * usn_slices(n, num_seg) * sizeof(Box) + 256;
}

static int usn_launch(const float* pos, const int64_t* idx, const int32_t* perm,
                      const int32_t* rowptr, const float* w_f32, const int64_t* w_i64, int64_t n,
                      int64_t num_seg, float* pos_out, const float* x, int cx, float* xcat,
                      float* diam, float* center, hipStream_t stream, void* ws = nullptr,
                      size_t ws_bytes = 0);

extern "C" int spt_unit_sphere_norm_f32(const float* pos, const int64_t* idx,
                                        const int32_t* perm, const int32_t* rowptr,
                                        const float* w_f32, const int64_t* w_i64,
                                        int64_t n, int64_t num_seg, float* pos_out,
                                        float* diam, float* center,
                                        spt_stream_t stream_) {
  SPT_CHECK_ARG(n == 0 || pos_out, "null pos_out");
  return usn_launch(pos, idx, perm, rowptr, w_f32, w_i64, n, num_seg, pos_out, nullptr, 0, nullptr,
                    diam, center, (hipStream_t)stream_);
}

// UnitSphereNorm + the stage's input assembly: xcat [n, 4 + cx] = [diam[idx] | pos_normalised | x]
// (x [n, cx] row-major, cx % 4 == 0, 16-byte aligned), diam / center as above.
extern "C" int spt_unit_sphere_assemble_f32(const float* pos, const int64_t* idx,
                                            const int32_t* perm, const int32_t* rowptr,
                                            const float* w_f32, const int64_t* w_i64, int64_t n,
                                            int64_t num_seg, const float* x, int cx, float* xcat,
                                            float* diam, float* center, spt_stream_t stream_) {
  SPT_CHECK_ARG(cx >= 4 && cx % 4 == 0, "x needs a multiple of 4 columns");
  SPT_CHECK_ARG(n == 0 || (x && xcat), "null x / xcat");
  SPT_CHECK_ARG(((uintptr_t)x | (uintptr_t)xcat) % 16 == 0, "x / xcat must be 16-byte aligned");
  return usn_launch(pos, idx, perm, rowptr, w_f32, w_i64, n, num_seg, nullptr, x, cx, xcat, diam,
                    center, (hipStream_t)stream_);
}

// The two entries above with a caller workspace (spt_unit_sphere_workspace_bytes): segments of
// thousands of rows (idx = NULL at the top level of a scene) are reduced by several workgroups
// each - partial boxes in the workspace, merged in a fixed order - instead of one.
extern "C" int spt_unit_sphere_norm_ws_f32(const float* pos, const int64_t* idx,
                                           const int32_t* perm, const int32_t* rowptr,
                                           const float* w_f32, const int64_t* w_i64, int64_t n,
                                           int64_t num_seg, float* pos_out, float* diam,
                                           float* center, void* ws, size_t ws_bytes,
                                           spt_stream_t stream_) {
  SPT_CHECK_ARG(n == 0 || pos_out, "null pos_out");
  return usn_launch(pos, idx, perm, rowptr, w_f32, w_i64, n, num_seg, pos_out, nullptr, 0, nullptr,
                    diam, center, (hipStream_t)stream_, ws, ws_bytes);
}
extern "C" int spt_unit_sphere_assemble_ws_f32(const float* pos, const int64_t* idx,
                                               const int32_t* perm, const int32_t* rowptr,
                                               const float* w_f32, const int64_t* w_i64, int64_t n,
                                               int64_t num_seg, const float* x, int cx, float* xcat,
                                               float* diam, float* center, void* ws,
                                               size_t ws_bytes, spt_stream_t stream_) {
  SPT_CHECK_ARG(cx >= 4 && cx % 4 == 0, "x needs a multiple of 4 columns");
  SPT_CHECK_ARG(n == 0 || (x && xcat), "null x / xcat");
  SPT_CHECK_ARG(((uintptr_t)x | (uintptr_t)xcat) % 16 == 0, "x / xcat must be 16-byte aligned");
  return usn_launch(pos, idx, perm, rowptr, w_f32, w_i64, n, num_seg, nullptr, x, cx, xcat, diam,
                    center, (hipStream_t)stream_, ws, ws_bytes);
}

static int usn_launch(const float* pos, const int64_t* idx, const int32_t* perm,
                      const int32_t* rowptr, const float* w_f32, const int64_t* w_i64, int64_t n,
                      int64_t num_seg, float* pos_out, const float* x, int cx, float* xcat,
                      float* diam, float* center, hipStream_t stream, void* ws, size_t ws_bytes) {
  SPT_CHECK_ARG(n >= 0 && num_seg >= 1, "bad shape");
  SPT_CHECK_ARG(rowptr && diam && center, "null pointer");
  SPT_CHECK_ARG(n == 0 || pos, "null pos");
  SPT_CHECK_ARG(!(w_f32 && w_i64), "pass at most one weight array");
  SPT_CHECK_ARG(idx || num_seg == 1, "idx may be null only for a single segment");
  const int64_t avg = n / num_seg;
  const int P = usn_slices(n, num_seg);
  float4* cd4 = nullptr;
  if (avg >= 2048 && P > 1 && ws && ws_bytes >= (size_t)num_seg * P * sizeof(Box)) {
    usn_stats_split_kernel<<<dim3(P, (unsigned)num_seg), 256, 0, stream>>>(
        pos, perm, rowptr, w_f32, w_i64, P, (Box*)ws);
    usn_stats_merge_kernel<<<(int)ceil_div(num_seg, 256), 256, 0, stream>>>(
        (const Box*)ws, rowptr, num_seg, P, center, diam);
  } else if (avg >= 2048) {
    const int grid = (int)(num_seg < 4096 ? num_seg : 4096);
    usn_stats_block_kernel<<<grid, 256, 0, stream>>>(pos, perm, rowptr, w_f32, w_i64,
                                                     num_seg, center, diam);
  } else {
    int g_log2 = 0;
    while (g_log2 < 6 && (((int64_t)4) << g_log2) <= avg) ++g_log2;  // ~4+ rows per lane
    const int spw = 64 >> g_log2;
    const int grid = stream_grid(ceil_div(num_seg, spw), 4);
    // the packed [centre | diameter] records for the assemble pass, in the (here unused) workspace
    if (xcat && idx && ws && ws_bytes >= (size_t)num_seg * sizeof(float4) + 256 && usn_rows_kernel())
      cd4 = reinterpret_cast<float4*>(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    usn_stats_group_kernel<<<grid, 256, 0, stream>>>(pos, perm, rowptr, w_f32, w_i64,
                                                     num_seg, g_log2, center, diam, cd4);
  }
  if (n > 0 && xcat)
  {
    const int64_t chunks = n * (cx / 4 + 1);
    const bool i32 = chunks + (int64_t)256 * 4096 < ((int64_t)1 << 32);   // 32-bit index arithmetic
    const int64_t rblk = ceil_div(n, (int64_t)(cx == 8 ? 256 : (cx == 64 ? 64 : 32)));
    const int rgrid = (int)(rblk < 256 * 16 ? rblk : 256 * 16);
    if (cx == 8 && usn_rows_kernel())
      usn_assemble_rows_kernel<3><<<rgrid, 256, 0, stream>>>(pos, idx, center, diam, x, n, xcat, cd4);
    else if (cx == 64 && usn_rows_kernel())
      usn_assemble_rows_kernel<17><<<rgrid, 256, 0, stream>>>(pos, idx, center, diam, x, n, xcat, cd4);
    else if (cx == 128 && usn_rows_kernel())
      usn_assemble_rows_kernel<33><<<rgrid, 256, 0, stream>>>(pos, idx, center, diam, x, n, xcat, cd4);
    else if (i32 && cx == 8)
      usn_assemble_kernel<uint32_t, 3><<<stream_grid(chunks, 256), 256, 0, stream>>>(
          pos, idx, center, diam, x, cx / 4, n, xcat);
    else if (i32 && cx == 64)
      usn_assemble_kernel<uint32_t, 17><<<stream_grid(chunks, 256), 256, 0, stream>>>(
          pos, idx, center, diam, x, cx / 4, n, xcat);
    else if (i32 && cx == 128)
      usn_assemble_kernel<uint32_t, 33><<<stream_grid(chunks, 256), 256, 0, stream>>>(
          pos, idx, center, diam, x, cx / 4, n, xcat);
    else if (i32)
      usn_assemble_kernel<uint32_t><<<stream_grid(chunks, 256), 256, 0, stream>>>(
          pos, idx, center, diam, x, cx / 4, n, xcat);
    else
      usn_assemble_kernel<int64_t><<<stream_grid(chunks, 256), 256, 0, stream>>>(
          pos, idx, center, diam, x, cx / 4, n, xcat);
  }
  else if (n > 0)
    usn_apply_kernel<<<stream_grid(n, 256), 256, 0, stream>>>(pos, idx, center, diam, n,
                                                              pos_out);
  SPT_CHECK_LAUNCH();
  return 0;
}
