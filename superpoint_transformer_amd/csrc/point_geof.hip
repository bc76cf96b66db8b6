// Point geometric features (replaces pgeof.compute_features, called at
// src/utils/geometry.py:148-153, and the torch branch _geometric_features_torch
// :236-338 + scatter_pca src/utils/scatter.py:41-125 the reference uses on GPU).
//
// Per point: population covariance of its neighbourhood (self + valid
// neighbours), symmetric 3x3 eigen-decomposition, the SPG eigenfeatures
//   [linearity, planarity, scattering, verticality, nx, ny, nz,
//    length, surface, volume, curvature]          (column order geometry.py:165-174)
// The reference materialises xyz[nn] ([N*(k+1),3]), six product columns, two
// scatters and a batched torch.linalg.eigh with host syncs.  Here one lane owns
// one point: the neighbourhood is accumulated in f64 relative to the point's
// own position (no cancellation), the eigenproblem is a cyclic Jacobi sweep in
// registers, nothing but the 11 outputs is written.
#include <math.h>

#include "common.hpp"
#include "geof_core.hpp"

namespace spt {

__device__ __forceinline__ void accumulate(const float* __restrict__ xyz, int64_t t, double px,
                                           double py, double pz, double s1[3], double s2[6],
                                           int& cnt) {
  const double dx = (double)xyz[t * 3] - px, dy = (double)xyz[t * 3 + 1] - py,
               dz = (double)xyz[t * 3 + 2] - pz;
  s1[0] += dx; s1[1] += dy; s1[2] += dz;
  s2[0] += dx * dx; s2[1] += dx * dy; s2[2] += dx * dz;
  s2[3] += dy * dy; s2[4] += dy * dz; s2[5] += dz * dz;
  ++cnt;
}

// CSR neighbourhoods val[ptr[i] .. ptr[i+1]): one lane per group
__global__ __launch_bounds__(256) void point_geof_csr_kernel(
    const float* __restrict__ xyz, int64_t n, const int64_t* __restrict__ nn,
    const int64_t* __restrict__ ptr, int add_self, int k_min, int post,
    float* __restrict__ feats) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int64_t lo = ptr[i], hi = ptr[i + 1];
    // origin of the moment sums: the point itself when it is part of its own
    // neighbourhood, else the first valid neighbour (row i of nn then describes a
    // GROUP of points, e.g. the samples of segment i - src/transforms/graph.py:239-242)
    int64_t o = i;
    if (!add_self) {
      o = -1;
      for (int64_t j = lo; j < hi && o < 0; ++j) o = nn[j] < 0 ? -1 : nn[j];
    }
    const double px = o < 0 ? 0.0 : (double)xyz[o * 3], py = o < 0 ? 0.0 : (double)xyz[o * 3 + 1],
                 pz = o < 0 ? 0.0 : (double)xyz[o * 3 + 2];
    double s1[3] = {0, 0, 0}, s2[6] = {0, 0, 0, 0, 0, 0};
    int cnt = add_self ? 1 : 0;          // the self contributes d = 0
    for (int64_t j = lo; j < hi; ++j) {
      const int64_t t = nn[j];
      if (t >= 0) accumulate(xyz, t, px, py, pz, s1, s2, cnt);
    }
    finish_features(s1, s2, cnt, k_min, post, feats + i * 11);
  }
}

// Dense neighbourhoods nn[i*k .. i*k+k), negative = missing: one lane per point, one wave
// per 64 points.  Half of the naive kernel's time was the index stream: lanes reading their
// own 8 k-byte rows touch 64 different cache lines per load and come back to each line 16
// times.  Here the wave loads the index columns [c0, c0+8) of its 64 rows cooperatively
// (8 lanes x 8 B = one 64-byte run per row) into LDS and each lane then reads its row back;
// the 8 position gathers of a chunk are independent and in flight together.
constexpr int GEOF_CH = 8;              // index columns per chunk
constexpr int GEOF_LD = GEOF_CH + 1;    // padded LDS row (int64 units)

__global__ __launch_bounds__(256) void point_geof_dense_kernel(
    const float* __restrict__ xyz, int64_t n, const int64_t* __restrict__ nn, int k, int64_t ld,
    int add_self, int k_min, int post, const int32_t* __restrict__ order,
    float* __restrict__ feats) {
  __shared__ int64_t idx_lds[4][64 * GEOF_LD];
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  int64_t* il = idx_lds[wid];
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int sub = lane >> 3, cc = lane & 7;
  for (int64_t base = wave * 64; base < n; base += nwaves * 64) {
    const int64_t ii = base + lane;
    // visiting order: spatially sorted when given, so that the neighbourhoods gathered by
    // the lanes of a wave overlap and stay in cache
    const int64_t i = (ii < n) ? (order ? (int64_t)order[ii] : ii) : -1;
    double px = 0.0, py = 0.0, pz = 0.0;
    bool have_origin = false;
    if (i >= 0 && add_self) {
      px = xyz[i * 3]; py = xyz[i * 3 + 1]; pz = xyz[i * 3 + 2];
      have_origin = true;
    }
    double s1[3] = {0, 0, 0}, s2[6] = {0, 0, 0, 0, 0, 0};
    int cnt = (i >= 0 && add_self) ? 1 : 0;
    for (int c0 = 0; c0 < k; c0 += GEOF_CH) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // unconditional loads from clamped addresses, all eight issued before the first is used
      // and masked afterwards: under a per-lane condition, or with the LDS store right behind
      // it, every load was followed by its own wait (eight dependent round trips)
      int64_t iv[8];
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int64_t ri = __shfl(i, 8 * p + sub, 64);
        const bool ok = ri >= 0 && c0 + cc < k;
        iv[p] = nn[ok ? ri * ld + c0 + cc : 0];
        iv[p] = ok ? iv[p] : -1;
      }
#pragma unroll
      for (int p = 0; p < 8; ++p) il[(8 * p + sub) * GEOF_LD + cc] = iv[p];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      int64_t t[GEOF_CH];
#pragma unroll
      for (int q = 0; q < GEOF_CH; ++q) t[q] = il[lane * GEOF_LD + q];
      if (!have_origin) {
        // no self: the first valid neighbour is the origin (see point_geof_csr_kernel)
#pragma unroll
        for (int q = 0; q < GEOF_CH; ++q)
          if (!have_origin && t[q] >= 0) {
            px = xyz[t[q] * 3]; py = xyz[t[q] * 3 + 1]; pz = xyz[t[q] * 3 + 2];
            have_origin = true;
          }
      }
      // the chunk's eight position gathers in flight together (missing neighbours read point 0
      // and are masked), then the moment sums in neighbour order
      float gx[GEOF_CH], gy[GEOF_CH], gz[GEOF_CH];
#pragma unroll
      for (int q = 0; q < GEOF_CH; ++q) {
        const int64_t tq = t[q] >= 0 ? t[q] : 0;
        gx[q] = xyz[tq * 3];
        gy[q] = xyz[tq * 3 + 1];
        gz[q] = xyz[tq * 3 + 2];
      }
#pragma unroll
      for (int q = 0; q < GEOF_CH; ++q)
        if (t[q] >= 0) {
          const double dx = (double)gx[q] - px, dy = (double)gy[q] - py, dz = (double)gz[q] - pz;
          s1[0] += dx; s1[1] += dy; s1[2] += dz;
          s2[0] += dx * dx; s2[1] += dx * dy; s2[2] += dx * dz;
          s2[3] += dy * dy; s2[4] += dy * dz; s2[5] += dz * dz;
          ++cnt;
        }
    }
    if (i >= 0) finish_features(s1, s2, cnt, k_min, post, feats + i * 11);
  }
}

}  // namespace spt

using namespace spt;

// ld: elements between the starts of consecutive rows of nn (>= k): the [N, k] table knn_1 hands
// out is columns 1 .. k of a [N, k + 1] search result - read in place, not copied (5.3 GB each way
// at 15 M points, k = 45)
extern "C" int spt_point_geof_dense_ld_f32(const float* xyz, int64_t n, const int64_t* nn,
                                           int k, int64_t ld, int add_self, int k_min, int post,
                                           const int32_t* order, float* feats,
                                           spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(n >= 0 && k >= 0 && ld >= k, "bad shape");
  if (n == 0) return 0;
  SPT_CHECK_ARG(xyz && feats && (nn || k == 0), "null pointer");
  point_geof_dense_kernel<<<stream_grid(n, 256), 256, 0, stream>>>(
      xyz, n, nn, k, ld, add_self, k_min, post, order, feats);
  SPT_CHECK_LAUNCH();
  return 0;
}

extern "C" int spt_point_geof_dense_f32(const float* xyz, int64_t n, const int64_t* nn,
                                        int k, int add_self, int k_min, int post,
                                        const int32_t* order, float* feats,
                                        spt_stream_t stream_) {
  return spt_point_geof_dense_ld_f32(xyz, n, nn, k, (int64_t)k, add_self, k_min, post, order, feats,
                                     stream_);
}

// ---- scatter_pca (src/utils/scatter.py:41-125) as an entry of its own ----------------------
// One lane per group of the CSR view (perm, rowptr): population covariance about the group's
// first row (f64 moments, no cancellation), cyclic Jacobi, eigenvalues ascending and clamped at
// 0 (scatter.py:123), eigenvectors in the columns of a row-major [3,3] block (the layout of
// torch.linalg.eigh), an empty group -> (1,1,1) / identity (scatter.py:113-118: 0/0 = NaN).
__global__ __launch_bounds__(256) void scatter_pca_kernel(
    const float* __restrict__ x, const int32_t* __restrict__ perm,
    const int32_t* __restrict__ rowptr, int64_t num_seg, float* __restrict__ eigenval,
    float* __restrict__ eigenvec) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < num_seg; s += stride) {
    const int lo = rowptr[s], hi = rowptr[s + 1];
    double w[3] = {1.0, 1.0, 1.0};
    double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    if (hi > lo) {
      const int64_t o = perm ? perm[lo] : lo;
      const double px = x[o * 3], py = x[o * 3 + 1], pz = x[o * 3 + 2];
      double s1[3] = {0, 0, 0}, s2[6] = {0, 0, 0, 0, 0, 0};
      int cnt = 0;
      for (int j = lo; j < hi; ++j) accumulate(x, perm ? perm[j] : j, px, py, pz, s1, s2, cnt);
      const double inv = 1.0 / (double)cnt;
      const double mx = s1[0] * inv, my = s1[1] * inv, mz = s1[2] * inv;
      double a[3][3];
      a[0][0] = s2[0] * inv - mx * mx;
      a[0][1] = a[1][0] = s2[1] * inv - mx * my;
      a[0][2] = a[2][0] = s2[2] * inv - mx * mz;
      a[1][1] = s2[3] * inv - my * my;
      a[1][2] = a[2][1] = s2[4] * inv - my * mz;
      a[2][2] = s2[5] * inv - mz * mz;
      eigh3(a, w, v);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) eigenval[s * 3 + q] = (float)(w[q] > 0.0 ? w[q] : 0.0);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q) eigenvec[s * 9 + r * 3 + q] = (float)v[r][q];
  }
}

extern "C" int spt_scatter_pca_f32(const float* x, const int32_t* perm, const int32_t* rowptr,
                                   int64_t num_seg, float* eigenval, float* eigenvec,
                                   spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(num_seg >= 0, "bad shape");
  if (num_seg == 0) return 0;
  SPT_CHECK_ARG(x && rowptr && eigenval && eigenvec, "null pointer");
  scatter_pca_kernel<<<stream_grid(num_seg, 256), 256, 0, stream>>>(x, perm, rowptr, num_seg,
                                                                   eigenval, eigenvec);
  SPT_CHECK_LAUNCH();
  return 0;
}

extern "C" int spt_point_geof_csr_f32(const float* xyz, int64_t n, const int64_t* nn_val,
                                      const int64_t* nn_ptr, int add_self, int k_min,
                                      int post, float* feats, spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(n >= 0, "bad shape");
  if (n == 0) return 0;
  SPT_CHECK_ARG(xyz && feats && nn_ptr, "null pointer");
  point_geof_csr_kernel<<<stream_grid(n, 256), 256, 0, stream>>>(
      xyz, n, nn_val, nn_ptr, add_self, k_min, post, feats);
  SPT_CHECK_LAUNCH();
  return 0;
}
