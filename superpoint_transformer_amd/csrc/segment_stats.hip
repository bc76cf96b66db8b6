// Per-segment statistics of the segment-level preprocessing:
//   * spt_segment_std_f32               torch_scatter.scatter_std (unbiased) as called
//                                        by SegmentFeatures, src/transforms/graph.py:285
//   * spt_segment_mean_orientation_f32   scatter_mean_orientation,
//                                        src/utils/scatter.py:249-300
// Both stream the rows of a segment through its CSR view (perm, rowptr): one wave per
// segment, lanes stride the rows, f64 accumulators, wave reductions, no atomics.  The
// reference spends 3 (std) and 5 (orientation) scatter launches plus as many gathers
// and [N, D] temporaries on each.
#include "common.hpp"

namespace spt {
namespace segstats {

constexpr int CB = 8;   // channels per sweep of the std kernel

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// std[s, c] = sqrt( sum_i (x[i,c] - mean[s,c])^2 / (max(cnt - 1, 1) + 1e-6) )
// (torch_scatter's scatter_std: two passes, unbiased, clamped denominator + 1e-6)
__global__ __launch_bounds__(256) void segment_std_kernel(
    const float* __restrict__ x, const int32_t* __restrict__ perm,
    const int32_t* __restrict__ rowptr, int64_t num_seg, int c, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t s = wave; s < num_seg; s += nwaves) {
    const int lo = rowptr[s], hi = rowptr[s + 1];
    const int cnt = hi - lo;
    for (int c0 = 0; c0 < c; c0 += CB) {
      const int cw = (c - c0 < CB) ? (c - c0) : CB;
      double sum[CB];
#pragma unroll
      for (int q = 0; q < CB; ++q) sum[q] = 0.0;
      for (int j = lo + lane; j < hi; j += 64) {
        const float* row = x + (int64_t)(perm ? perm[j] : j) * c + c0;
#pragma unroll
        for (int q = 0; q < CB; ++q)
          if (q < cw) sum[q] += (double)row[q];
      }
      float mean[CB];
#pragma unroll
      for (int q = 0; q < CB; ++q)
        mean[q] = (float)(wave_sum_f64(sum[q]) / (double)(cnt > 0 ? cnt : 1));
      double dev[CB];
#pragma unroll
      for (int q = 0; q < CB; ++q) dev[q] = 0.0;
      for (int j = lo + lane; j < hi; j += 64) {
        const float* row = x + (int64_t)(perm ? perm[j] : j) * c + c0;
#pragma unroll
        for (int q = 0; q < CB; ++q)
          if (q < cw) {
            const float d = row[q] - mean[q];        // f32 like (src - mean[index])
            dev[q] += (double)(d * d);
          }
      }
      const float denom = (float)((cnt - 1 > 1) ? (cnt - 1) : 1) + 1e-6f;
#pragma unroll
      for (int q = 0; q < CB; ++q) {
        const double t = wave_sum_f64(dev[q]);
        if (lane == 0 && q < cw) out[s * c + c0 + q] = sqrtf((float)t / denom);
      }
    }
  }
}

// order-preserving map f32 -> u32 (ascending), for packed (value, position) minima
__device__ __forceinline__ uint32_t ordered_bits(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ void unit_orientation(const float* __restrict__ o, float v[3]) {
  // x /= ||x|| + 1e-4 ; clamp to [-1, 1]                 (scatter.py:268-269)
  const float nrm = sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]) + 1e-4f;
#pragma unroll
  for (int k = 0; k < 3; ++k) v[k] = fminf(fmaxf(o[k] / nrm, -1.f), 1.f);
}

__global__ __launch_bounds__(256) void mean_orientation_kernel(
    const float* __restrict__ orient, const int32_t* __restrict__ perm,
    const int32_t* __restrict__ rowptr, int64_t num_seg, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t s = wave; s < num_seg; s += nwaves) {
    const int lo = rowptr[s], hi = rowptr[s + 1];
    const int cnt = hi - lo;
    // pass 1: mean elevation angle and the first row of smallest angle (scatter.py:272-283)
    double phi_sum = 0.0;
    uint64_t best = ~0ull;
    for (int j = lo + lane; j < hi; j += 64) {
      float v[3];
      unit_orientation(orient + (int64_t)(perm ? perm[j] : j) * 3, v);
      const float phi = asinf(v[2]);
      phi_sum += (double)phi;
      const uint64_t key = ((uint64_t)ordered_bits(phi) << 32) | (uint32_t)(j - lo);
      best = key < best ? key : best;
    }
    phi_sum = wave_sum_f64(phi_sum);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint64_t t = __shfl_xor((unsigned long long)best, o, 64);
      best = t < best ? t : best;
    }
    float m[3] = {0.f, 0.f, 0.f};
    if (cnt > 0) {
      const float phi_mean = (float)(phi_sum / (double)cnt);
      const bool horizontal = phi_mean < 0.78539816339744830962f;   // pi / 4 in f32
      float ref[3];
      const int jr = lo + (int)(uint32_t)best;
      unit_orientation(orient + (int64_t)(perm ? perm[jr] : jr) * 3, ref);
      // pass 2: flip the rows opposing the reference row, average       (scatter.py:284-290)
      double acc[3] = {0.0, 0.0, 0.0};
      for (int j = lo + lane; j < hi; j += 64) {
        float v[3];
        unit_orientation(orient + (int64_t)(perm ? perm[j] : j) * 3, v);
        const float dot = v[0] * ref[0] + v[1] * ref[1] + v[2] * ref[2];
        const float sgn = (horizontal && dot < 0.f) ? -1.f : 1.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[k] += (double)(sgn * v[k]);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) m[k] = (float)(wave_sum_f64(acc[k]) / (double)cnt);
    }
    // normalise, clamp, express towards z+                               (scatter.py:293-298)
    const float nrm = sqrtf(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]) + 1e-4f;
#pragma unroll
    for (int k = 0; k < 3; ++k) m[k] = fminf(fmaxf(m[k] / nrm, -1.f), 1.f);
    if (m[2] < 0.f) { m[0] = -m[0]; m[1] = -m[1]; m[2] = -m[2]; }
    if (lane < 3) out[s * 3 + lane] = (lane == 0) ? m[0] : (lane == 1) ? m[1] : m[2];
  }
}

}  // namespace segstats
}  // namespace spt

using namespace spt;
using namespace spt::segstats;

static int waves_grid(int64_t num_seg) {
  int64_t b = ceil_div(num_seg, 4);
  if (b > 256 * 16) b = 256 * 16;
  return (int)(b < 1 ? 1 : b);
}

extern "C" int spt_segment_std_f32(const float* x, const int32_t* perm, const int32_t* rowptr,
                                   int64_t num_seg, int c, float* out, spt_stream_t stream) {
  SPT_CHECK_ARG(num_seg >= 0 && c >= 1, "bad shape");
  if (num_seg == 0) return 0;
  SPT_CHECK_ARG(x && rowptr && out, "null pointer");
  segment_std_kernel<<<waves_grid(num_seg), 256, 0, (hipStream_t)stream>>>(x, perm, rowptr,
                                                                          num_seg, c, out);
  SPT_CHECK_LAUNCH();
  return 0;
}

extern "C" int spt_segment_mean_orientation_f32(const float* orientation, const int32_t* perm,
                                                const int32_t* rowptr, int64_t num_seg,
                                                float* out, spt_stream_t stream) {
  SPT_CHECK_ARG(num_seg >= 0, "bad shape");
  if (num_seg == 0) return 0;
  SPT_CHECK_ARG(orientation && rowptr && out, "null pointer");
  mean_orientation_kernel<<<waves_grid(num_seg), 256, 0, (hipStream_t)stream>>>(
      orientation, perm, rowptr, num_seg, out);
  SPT_CHECK_LAUNCH();
  return 0;
}
