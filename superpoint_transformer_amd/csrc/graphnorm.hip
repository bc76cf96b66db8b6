// GraphNorm forward / backward (torch_geometric.nn.norm.GraphNorm as used by
// src/nn/mlp.py:85-94 and src/nn/transformer.py:258-265), optionally fused
// with the LeakyReLU that follows it in every MLP layer (src/nn/mlp.py:46-50).
//
//   mu_g  = mean_{batch=g} x            o = x - alpha * mu_g[batch]
//   var_g = mean_{batch=g} o^2          y = weight * o / sqrt(var_g + eps) + bias
//
// Regime: FEW (1..4) segments of up to 15 M rows each - the opposite of the
// superpoint pooling kernels.  The reference spends 2 scatter-mean launches
// with B-way atomic contention + 2 gathers + elementwise passes (~5 passes
// over x).  Here: ONE statistics pass (f64 register accumulators per lane,
// per-workgroup partial tables, deterministic fixed-order finalize) and ONE
// apply pass => 2 reads + 1 write of x.  Backward has the same two-pass shape.
#include <math.h>

#include "common.hpp"

namespace spt {

constexpr int GN_THREADS = 256;
constexpr int GN_MAX_BLOCKS = 1024;
#ifndef SPT_GN_UNR
#define SPT_GN_UNR 4
#endif
constexpr int GN_UNR = SPT_GN_UNR;

struct GnShape {
  int vec, lpr_log2, rpb;  // floats per lane, lanes per row (log2), rows per block-iteration
};

static bool gn_shape(int d, GnShape* s) {
  s->vec = (d % 4 == 0) ? 4 : (d % 2 == 0) ? 2 : 1;
  int lanes = d / s->vec;
  int l2 = 0;
  while ((1 << l2) < lanes) ++l2;
  if (l2 > 8) return false;  // d > 1024 floats per row: not a GraphNorm shape of this model family
  s->lpr_log2 = l2;
  s->rpb = GN_THREADS >> l2;
  return true;
}

template <int VEC>
__device__ __forceinline__ void ld(const float* p, float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else if constexpr (VEC == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    v[0] = t.x; v[1] = t.y;
  } else {
    v[0] = *p;
  }
}
template <int VEC>
__device__ __forceinline__ void st(float* p, const float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  } else if constexpr (VEC == 2) {
    *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
  } else {
    *p = v[0];
  }
}

// ---- statistics pass ---------------------------------------------------------
// Per (graph, channel): two f64 sums.  FWD: (sum x, sum x^2) + row count.
// BWD: (sum g, sum g*o) with g = gy * leaky'(y), o = x - alpha*mu.
// LDS table layout per block: [B][2*d + 1] doubles (last = row count).
template <int VEC, bool BWD>
__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(
    const float* __restrict__ x, const float* __restrict__ gy,
    const int64_t* __restrict__ batch, int64_t R, int d, int B, int lpr_log2,
    const float* __restrict__ am,     // [B,d] alpha*mu          (BWD)
    const float* __restrict__ scale,  // [B,d] weight*rstd       (BWD, act mask)
    const float* __restrict__ bias,   // [d]                     (BWD, act mask)
    float slope, double* __restrict__ partial,
    int b_lo, int Bc,                 // this launch owns graphs [b_lo, b_lo + Bc): its LDS table
    const float* __restrict__ mean = nullptr,     // BWD alternative to (am, scale): the saved
    const float* __restrict__ rstd = nullptr,     // statistics + parameters, tables formed here
    const float* __restrict__ weight = nullptr,   // exactly as gn_rebuild_tables_kernel does
    const float* __restrict__ mean_scale = nullptr) {
  extern __shared__ __attribute__((aligned(16))) double tab[];
  const int row_len = 2 * d + 1;
  for (int i = threadIdx.x; i < Bc * row_len; i += GN_THREADS) tab[i] = 0.0;
  __syncthreads();

  const int lpr = 1 << lpr_log2;
  const int rpb = GN_THREADS >> lpr_log2;
  const int rsub = threadIdx.x >> lpr_log2;
  const int lr = threadIdx.x & (lpr - 1);
  const int c0 = lr * VEC;
  const bool cv = c0 < d;

  double s1[VEC], s2[VEC];
  double cnt = 0.0;
  int cur = b_lo;
#pragma unroll
  for (int k = 0; k < VEC; ++k) s1[k] = s2[k] = 0.0;
  // per-(graph, channel) coefficients of the CURRENT graph live in registers:
  // the graph id changes a handful of times per lane, the rows are millions
  float t_am[VEC], t_sc[VEC], t_bs[VEC];
  auto load_tables = [&](int b) {
    if constexpr (BWD) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        if (mean) {
          t_am[k] = cv ? (float)((double)mean_scale[c0 + k] * (double)mean[b * d + c0 + k]) : 0.f;
          t_sc[k] = cv ? (float)((double)weight[c0 + k] * (double)rstd[b * d + c0 + k]) : 0.f;
        } else {
          t_am[k] = cv ? am[b * d + c0 + k] : 0.f;
          t_sc[k] = cv ? scale[b * d + c0 + k] : 0.f;
        }
        t_bs[k] = cv ? bias[c0 + k] : 0.f;
      }
    }
  };
  load_tables(b_lo);

  auto flush = [&]() {
    if (cv) {
      double* t = tab + (size_t)(cur - b_lo) * row_len;
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        if (s1[k] != 0.0) atomicAdd(&t[c0 + k], s1[k]);
        if (s2[k] != 0.0) atomicAdd(&t[d + c0 + k], s2[k]);
        s1[k] = s2[k] = 0.0;
      }
      if (lr == 0 && cnt != 0.0) atomicAdd(&t[2 * d], cnt);
    }
    cnt = 0.0;
  };

  const int64_t step = (int64_t)gridDim.x * rpb * GN_UNR;
  for (int64_t rb = (int64_t)blockIdx.x * rpb * GN_UNR; rb < R; rb += step) {
    int64_t row[GN_UNR];
    int b[GN_UNR];
    float v[GN_UNR][VEC], g[GN_UNR][VEC];
#pragma unroll
    for (int u = 0; u < GN_UNR; ++u) {
      row[u] = rb + (int64_t)u * rpb + rsub;
      const bool ok = row[u] < R;
      b[u] = ok ? (batch ? (int)batch[row[u]] : 0) : -1;
      if (b[u] < b_lo || b[u] >= b_lo + Bc) b[u] = -1;   // another launch's graph
      if (b[u] >= 0 && cv) {
        ld<VEC>(x + row[u] * d + c0, v[u]);
        if constexpr (BWD) ld<VEC>(gy + row[u] * d + c0, g[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < GN_UNR; ++u) {
      if (b[u] < 0) continue;
      if (b[u] != cur) {
        flush();
        cur = b[u];
        load_tables(cur);
      }
      cnt += 1.0;
      if (!cv) continue;
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        if constexpr (!BWD) {
          const double xv = (double)v[u][k];
          s1[k] += xv;
          s2[k] += xv * xv;
        } else {
          const float o = v[u][k] - t_am[k];
          float gg = g[u][k];
          if (slope != 1.f) {
            const float y = fmaf(o, t_sc[k], t_bs[k]);
            gg = (y > 0.f) ? gg : gg * slope;
          }
          s1[k] += (double)gg;
          s2[k] += (double)gg * (double)o;
        }
      }
    }
  }
  flush();
  __syncthreads();
  double* out = partial + ((size_t)blockIdx.x * B + b_lo) * row_len;
  for (int i = threadIdx.x; i < Bc * row_len; i += GN_THREADS) out[i] = tab[i];
}

// ---- finalize: fixed-order reduction of the per-block partial tables --------
// grid = (B, ceil(row_len / 16)); 256 threads = 16 columns x 16 slices: each thread
// sums <= 64 partials (the 64-column x 4-slice shape took 73 us per call, 40 calls
// per train step, purely on load latency).
__global__ __launch_bounds__(256) void gn_reduce_partials_kernel(
    const double* __restrict__ partial, int nblocks, int B, int row_len,
    double* __restrict__ total) {
  __shared__ double sl[16][17];
  const int b = blockIdx.x;
  const int cl = threadIdx.x & 15;
  const int col = blockIdx.y * 16 + cl;
  const int slice = threadIdx.x >> 4;
  double acc = 0.0;
  if (col < row_len) {
    const int per = (nblocks + 15) / 16;
    const int lo = slice * per, hi = (lo + per < nblocks) ? lo + per : nblocks;
    int k = lo;
    for (; k + 8 <= hi; k += 8) {                  // eight partials in flight, added in order
      double a[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = partial[((size_t)(k + j) * B + b) * row_len + col];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += a[j];
    }
    for (; k < hi; ++k) acc += partial[((size_t)k * B + b) * row_len + col];
  }
  sl[slice][cl] = acc;
  __syncthreads();
  if (slice == 0 && col < row_len) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sl[k][cl];   // fixed order: deterministic
    total[(size_t)b * row_len + col] = t;
  }
}

// The same fixed-order reduction with the table formulas applied by the thread that ends up with
// the totals: one launch instead of two between the statistics pass and the apply pass (at a
// train batch's row counts a GraphNorm is five to nine ~5-20 us launches; these merges take three
// of them away per forward + backward).  Arithmetic identical to the two-kernel sequences.
// (All the partial rows a thread needs - 3 columns x the graphs it covers - are loaded in the
// same loop iteration: one memory round trip per 16th of the blocks, as in the plain reduction.)
template <int NV>
__device__ __forceinline__ void gn_sliced_sums(const double* __restrict__ partial, int nblocks,
                                               size_t stride, const size_t (&off)[NV], int nv,
                                               int slice, int cl, double (*sl)[16][17],
                                               double (&out)[NV]) {
  double acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) acc[v] = 0.0;
  const int per = (nblocks + 15) / 16;
  const int lo = slice * per, hi = (lo + per < nblocks) ? lo + per : nblocks;
  int k = lo;
  constexpr int UNR = NV <= 3 ? 4 : 1;             // (twelve columns are twelve loads in flight already)
  if constexpr (UNR > 1)
  for (; k + UNR <= hi; k += UNR) {                // four blocks' rows in flight, added in order
    double t[NV][UNR];
#pragma unroll
    for (int j = 0; j < UNR; ++j)
#pragma unroll
      for (int v = 0; v < NV; ++v) t[v][j] = v < nv ? partial[(size_t)(k + j) * stride + off[v]] : 0.0;
#pragma unroll
    for (int j = 0; j < UNR; ++j)
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[v] += t[v][j];
  }
  for (; k < hi; ++k) {
    double t[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) t[v] = v < nv ? partial[(size_t)k * stride + off[v]] : 0.0;
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] += t[v];
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) sl[v][slice][cl] = acc[v];
  __syncthreads();
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sl[v][k][cl];     // fixed order: deterministic
    out[v] = t;
  }
}

__global__ __launch_bounds__(256) void gn_finalize_fwd_kernel(
    const double* __restrict__ partial, int nblocks, int B, int d,
    const float* __restrict__ weight, const float* __restrict__ mean_scale, float eps,
    float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ am,
    float* __restrict__ scale) {
  __shared__ double sl[3][16][17];
  const int b = blockIdx.x, cl = threadIdx.x & 15, slice = threadIdx.x >> 4;
  const int c = blockIdx.y * 16 + cl, row_len = 2 * d + 1;
  const int cc = c < d ? c : d - 1;
  const size_t off[3] = {(size_t)b * row_len + cc, (size_t)b * row_len + d + cc,
                         (size_t)b * row_len + 2 * d};
  double s[3];
  gn_sliced_sums<3>(partial, nblocks, (size_t)B * row_len, off, 3, slice, cl, sl, s);
  if (slice != 0 || c >= d) return;
  double n = s[2];
  if (n < 1.0) n = 1.0;                       // scatter_mean: clamp(count, 1)
  const double mu = s[0] / n;
  const double a = (double)mean_scale[c];
  double var = s[1] / n - (2.0 * a - a * a) * mu * mu;
  if (var < 0.0) var = 0.0;
  const double rs = 1.0 / sqrt(var + (double)eps);
  const float mu32 = (float)mu, rs32 = (float)rs;
  const int t = b * d + c;
  mean[t] = mu32;
  rstd[t] = rs32;
  am[t] = (float)(a * (double)mu32);
  scale[t] = (float)((double)weight[c] * (double)rs32);
}

constexpr int GN_FIN_B = 4;      // graphs one finalize_bwd block reduces side by side
__global__ __launch_bounds__(256) void gn_finalize_bwd_kernel(
    const double* __restrict__ partial, int nblocks, int B, int d,
    const float* __restrict__ weight, const float* __restrict__ mean_scale,
    const float* __restrict__ mean, const float* __restrict__ rstd,
    float* __restrict__ am, float* __restrict__ scale, float* __restrict__ c1,
    float* __restrict__ c2, float* __restrict__ c3, float* __restrict__ gweight,
    float* __restrict__ gbias, float* __restrict__ gms) {
  __shared__ double sl[3 * GN_FIN_B][16][17];
  const int cl = threadIdx.x & 15, slice = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl, row_len = 2 * d + 1;
  const int cc = c < d ? c : d - 1;
  const bool own = slice == 0 && c < d;
  const double w = (double)weight[cc], a = (double)mean_scale[cc];
  double gw = 0.0, gb = 0.0, ga = 0.0;
  for (int b0 = 0; b0 < B; b0 += GN_FIN_B) {
    const int nb = (B - b0 < GN_FIN_B) ? B - b0 : GN_FIN_B;
    size_t off[3 * GN_FIN_B];
#pragma unroll
    for (int j = 0; j < GN_FIN_B; ++j) {
      const size_t rb = (size_t)(b0 + (j < nb ? j : 0)) * row_len;
      off[3 * j] = rb + cc;
      off[3 * j + 1] = rb + d + cc;
      off[3 * j + 2] = rb + 2 * d;
    }
    double s3[3 * GN_FIN_B];
    if (b0) __syncthreads();
    gn_sliced_sums<3 * GN_FIN_B>(partial, nblocks, (size_t)B * row_len, off, 3 * nb, slice, cl, sl, s3);
    if (!own) continue;
    for (int j = 0; j < nb; ++j) {
      const int b = b0 + j;
      const double A = s3[3 * j], GO = s3[3 * j + 1];
      double n = s3[3 * j + 2];
      if (n < 1.0) n = 1.0;
      const double s = (double)rstd[b * d + c], mu = (double)mean[b * d + c];
      const double k2 = w * s * s * s * GO / n;
      const double sumdo = w * s * A - k2 * n * mu * (1.0 - a);
      c1[b * d + c] = (float)(w * s);
      c2[b * d + c] = (float)k2;
      c3[b * d + c] = (float)(a * sumdo / n);
      am[b * d + c] = (float)(a * mu);           // = gn_rebuild_tables_kernel, for the apply pass
      scale[b * d + c] = (float)(w * s);
      gw += s * GO;
      gb += A;
      ga += -mu * sumdo;
    }
  }
  if (own) {
    gweight[c] = (float)gw;
    gbias[c] = (float)gb;
    gms[c] = (float)ga;
  }
}

// forward tables from the totals
__global__ void gn_fwd_tables_kernel(const double* __restrict__ total, int B, int d,
                                     const float* __restrict__ weight,
                                     const float* __restrict__ mean_scale, float eps,
                                     float* __restrict__ mean, float* __restrict__ rstd,
                                     float* __restrict__ am, float* __restrict__ scale) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * d) return;
  const int b = t / d, c = t - b * d;
  const double* row = total + (size_t)b * (2 * d + 1);
  double n = row[2 * d];
  if (n < 1.0) n = 1.0;                       // scatter_mean: clamp(count, 1)
  const double mu = row[c] / n;
  const double a = (double)mean_scale[c];
  // E[(x - a mu)^2] = E[x^2] - (2a - a^2) mu^2
  double var = row[d + c] / n - (2.0 * a - a * a) * mu * mu;
  if (var < 0.0) var = 0.0;
  const double rs = 1.0 / sqrt(var + (double)eps);
  // (am, scale) derive from the f32-ROUNDED statistics so that the backward,
  // which only has (mean, rstd), rebuilds bit-identical tables.
  const float mu32 = (float)mu, rs32 = (float)rs;
  mean[t] = mu32;
  rstd[t] = rs32;
  am[t] = (float)(a * (double)mu32);
  scale[t] = (float)((double)weight[c] * (double)rs32);
}

__global__ void gn_rebuild_tables_kernel(const float* __restrict__ mean,
                                         const float* __restrict__ rstd,
                                         const float* __restrict__ weight,
                                         const float* __restrict__ mean_scale, int B,
                                         int d, float* __restrict__ am,
                                         float* __restrict__ scale) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * d) return;
  const int c = t % d;
  am[t] = (float)((double)mean_scale[c] * (double)mean[t]);
  scale[t] = (float)((double)weight[c] * (double)rstd[t]);
}

// y = (x - alpha*mu) * (weight*rstd) + bias ; optional LeakyReLU
template <int VEC>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_fwd_kernel(
    const float* __restrict__ x, const int64_t* __restrict__ batch, int64_t R, int d,
    int lpr_log2, const float* __restrict__ am, const float* __restrict__ scale,
    const float* __restrict__ bias, float slope, float* __restrict__ y) {
  const int lpr = 1 << lpr_log2;
  const int rpb = GN_THREADS >> lpr_log2;
  const int rsub = threadIdx.x >> lpr_log2;
  const int c0 = (threadIdx.x & (lpr - 1)) * VEC;
  if (c0 >= d) return;
  float bs[VEC], t_am[VEC], t_sc[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) bs[k] = bias[c0 + k];
  int cur = -1;
  const int64_t step = (int64_t)gridDim.x * rpb * GN_UNR;
  for (int64_t rb = (int64_t)blockIdx.x * rpb * GN_UNR; rb < R; rb += step) {
    float v[GN_UNR][VEC];
    int b[GN_UNR];
#pragma unroll
    for (int u = 0; u < GN_UNR; ++u) {
      const int64_t row = rb + (int64_t)u * rpb + rsub;
      b[u] = (row < R) ? (batch ? (int)batch[row] : 0) : -1;
      if (b[u] >= 0) ld<VEC>(x + row * d + c0, v[u]);
    }
#pragma unroll
    for (int u = 0; u < GN_UNR; ++u) {
      if (b[u] < 0) continue;
      if (b[u] != cur) {  // coefficient rows of the current graph stay in registers
        cur = b[u];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          t_am[k] = am[cur * d + c0 + k];
          t_sc[k] = scale[cur * d + c0 + k];
        }
      }
      const int64_t row = rb + (int64_t)u * rpb + rsub;
      float o[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        float yy = fmaf(v[u][k] - t_am[k], t_sc[k], bs[k]);
        if (slope != 1.f) yy = (yy > 0.f) ? yy : yy * slope;
        o[k] = yy;
      }
      st<VEC>(y + row * d + c0, o);
    }
  }
}

// backward tables:  gx = c1*g - c2*o - c3
__global__ void gn_bwd_tables_kernel(const double* __restrict__ total, int B, int d,
                                     const float* __restrict__ weight,
                                     const float* __restrict__ mean_scale,
                                     const float* __restrict__ mean,
                                     const float* __restrict__ rstd,
                                     float* __restrict__ c1, float* __restrict__ c2,
                                     float* __restrict__ c3, float* __restrict__ gweight,
                                     float* __restrict__ gbias, float* __restrict__ gms) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d) return;
  const double w = (double)weight[c], a = (double)mean_scale[c];
  double gw = 0.0, gb = 0.0, ga = 0.0;
  for (int b = 0; b < B; ++b) {
    const double* row = total + (size_t)b * (2 * d + 1);
    double n = row[2 * d];
    if (n < 1.0) n = 1.0;
    const double A = row[c], GO = row[d + c];
    const double s = (double)rstd[b * d + c], mu = (double)mean[b * d + c];
    const double k2 = w * s * s * s * GO / n;
    const double sumdo = w * s * A - k2 * n * mu * (1.0 - a);
    c1[b * d + c] = (float)(w * s);
    c2[b * d + c] = (float)k2;
    c3[b * d + c] = (float)(a * sumdo / n);
    gw += s * GO;
    gb += A;
    ga += -mu * sumdo;
  }
  gweight[c] = (float)gw;
  gbias[c] = (float)gb;
  gms[c] = (float)ga;
}

template <int VEC>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gy,
    const int64_t* __restrict__ batch, int64_t R, int d, int lpr_log2,
    const float* __restrict__ am, const float* __restrict__ scale,
    const float* __restrict__ bias, float slope, const float* __restrict__ c1,
    const float* __restrict__ c2, const float* __restrict__ c3,
    float* __restrict__ gx, const float* __restrict__ gadd) {
  // gadd (nullable): a second gradient of x - the residual branch around the norm - added in
  // this pass (gx = gadd + norm-backward) instead of by a separate elementwise launch
  const int lpr = 1 << lpr_log2;
  const int rpb = GN_THREADS >> lpr_log2;
  const int rsub = threadIdx.x >> lpr_log2;
  const int c0 = (threadIdx.x & (lpr - 1)) * VEC;
  if (c0 >= d) return;
  float bs[VEC], t_am[VEC], t_sc[VEC], t_c1[VEC], t_c2[VEC], t_c3[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) bs[k] = bias[c0 + k];
  int cur = -1;
  const int64_t step = (int64_t)gridDim.x * rpb * GN_UNR;
  for (int64_t rb = (int64_t)blockIdx.x * rpb * GN_UNR; rb < R; rb += step) {
    float v[GN_UNR][VEC], g[GN_UNR][VEC], ga[GN_UNR][VEC];
    int b[GN_UNR];
#pragma unroll
    for (int u = 0; u < GN_UNR; ++u) {
      const int64_t row = rb + (int64_t)u * rpb + rsub;
      b[u] = (row < R) ? (batch ? (int)batch[row] : 0) : -1;
      if (b[u] >= 0) {
        ld<VEC>(x + row * d + c0, v[u]);
        ld<VEC>(gy + row * d + c0, g[u]);
        if (gadd) ld<VEC>(gadd + row * d + c0, ga[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < GN_UNR; ++u) {
      if (b[u] < 0) continue;
      if (b[u] != cur) {
        cur = b[u];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const int t = cur * d + c0 + k;
          t_am[k] = am[t]; t_sc[k] = scale[t];
          t_c1[k] = c1[t]; t_c2[k] = c2[t]; t_c3[k] = c3[t];
        }
      }
      const int64_t row = rb + (int64_t)u * rpb + rsub;
      float o[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const float oo = v[u][k] - t_am[k];
        float gg = g[u][k];
        if (slope != 1.f) {
          const float yy = fmaf(oo, t_sc[k], bs[k]);
          gg = (yy > 0.f) ? gg : gg * slope;
        }
        o[k] = fmaf(t_c1[k], gg, -fmaf(t_c2[k], oo, t_c3[k]));
        if (gadd) o[k] += ga[u][k];
      }
      st<VEC>(gx + row * d + c0, o);
    }
  }
}

struct GnPlan {
  GnShape sh;
  int nblocks, row_len;
  size_t off_partial, off_total, off_am, off_scale, off_c1, off_c2, off_c3, total;
};

static bool gn_plan(int64_t R, int d, int B, GnPlan* p) {
  if (!gn_shape(d, &p->sh)) return false;
  p->row_len = 2 * d + 1;
  // one partial table per block: the finalize kernels walk them serially (16 slices), so a launch
  // over few rows gets fewer, fatter blocks - at least four unrolled iterations each unless that
  // leaves less than 128 blocks (train-batch sizes: 35 000 rows -> 137 tables instead of 547,
  // the table reduction 26 -> ~8 us); large inputs keep the full grid
  const int64_t per_iter = (int64_t)p->sh.rpb * GN_UNR;
  int64_t nb = ceil_div(R > 0 ? R : 1, per_iter * 4);
  if (nb < 128) {
    nb = ceil_div(R > 0 ? R : 1, per_iter);
    if (nb > 128) nb = 128;
  }
  if (nb > GN_MAX_BLOCKS) nb = GN_MAX_BLOCKS;
  p->nblocks = (int)nb;
  size_t o = 0;
  p->off_partial = o; o += align_up((size_t)p->nblocks * B * p->row_len * 8, 256);
  p->off_total = o;   o += align_up((size_t)B * p->row_len * 8, 256);
  const size_t tb = align_up((size_t)B * d * 4, 256);
  p->off_am = o; o += tb;
  p->off_scale = o; o += tb;
  p->off_c1 = o; o += tb;
  p->off_c2 = o; o += tb;
  p->off_c3 = o; o += tb;
  p->total = o;
  return true;
}

static int gn_graphs_per_launch(int row_len) {
  int cap = (int)((64 * 1024) / ((size_t)row_len * 8));
  return cap < 1 ? 1 : cap;          // d <= 1024: one graph row is at most 16 392 B
}

template <bool BWD>
static void launch_stats(const GnPlan& p, const float* x, const float* gy,
                         const int64_t* batch, int64_t R, int d, int B,
                         const float* am, const float* scale, const float* bias,
                         float slope, double* partial, hipStream_t stream,
                         const float* mean = nullptr, const float* rstd = nullptr,
                         const float* weight = nullptr, const float* mean_scale = nullptr) {
  // The per-graph table lives in LDS (64 KiB without opting into more).  More graphs than
  // fit (num_graphs > 31 at d = 128) are covered by several launches, each owning a window
  // of graph ids and skipping the other rows; the usual 1..8 graphs take one launch.
  const int cap = gn_graphs_per_launch(p.row_len);
  for (int b_lo = 0; b_lo < B; b_lo += cap) {
    const int Bc = (B - b_lo < cap) ? B - b_lo : cap;
    const size_t lds = (size_t)Bc * p.row_len * 8;
    if (p.sh.vec == 4)
      gn_stats_kernel<4, BWD><<<p.nblocks, GN_THREADS, lds, stream>>>(x, gy, batch, R, d, B, p.sh.lpr_log2, am, scale, bias, slope, partial, b_lo, Bc, mean, rstd, weight, mean_scale);
    else if (p.sh.vec == 2)
      gn_stats_kernel<2, BWD><<<p.nblocks, GN_THREADS, lds, stream>>>(x, gy, batch, R, d, B, p.sh.lpr_log2, am, scale, bias, slope, partial, b_lo, Bc, mean, rstd, weight, mean_scale);
    else
      gn_stats_kernel<1, BWD><<<p.nblocks, GN_THREADS, lds, stream>>>(x, gy, batch, R, d, B, p.sh.lpr_log2, am, scale, bias, slope, partial, b_lo, Bc, mean, rstd, weight, mean_scale);
  }
}

}  // namespace spt

using namespace spt;

extern "C" size_t spt_graphnorm_workspace_bytes(int64_t r, int d, int num_graphs) {
  GnPlan p;
  if (r < 0 || d < 1 || num_graphs < 1 || !gn_plan(r, d, num_graphs, &p)) return 0;
  return p.total;
}

static int gn_check(int64_t r, int d, int B, const GnPlan& p, const void* ws, size_t ws_bytes) {
  SPT_CHECK_ARG(ws && ws_bytes >= p.total, "workspace too small");
  return 0;
}

extern "C" int spt_graphnorm_fwd_f32(const float* x, const int64_t* batch, int64_t r,
                                     int d, int num_graphs, const float* weight,
                                     const float* bias, const float* mean_scale,
                                     float eps, float act_slope, float* y,
                                     float* mean, float* rstd, void* ws,
                                     size_t ws_bytes, spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int B = num_graphs;
  SPT_CHECK_ARG(r >= 0 && d >= 1 && B >= 1, "bad shape");
  GnPlan p;
  SPT_CHECK_ARG(gn_plan(r, d, B, &p), "dim > 1024 unsupported");
  if (int e = gn_check(r, d, B, p, ws, ws_bytes)) return e;
  SPT_CHECK_ARG(weight && bias && mean_scale && mean && rstd && (r == 0 || (x && y)), "null pointer");
  char* base = (char*)ws;
  double* partial = (double*)(base + p.off_partial);
  double* total = (double*)(base + p.off_total);
  float* am = (float*)(base + p.off_am);
  float* scale = (float*)(base + p.off_scale);
  launch_stats<false>(p, x, nullptr, batch, r, d, B, nullptr, nullptr, nullptr, 1.f, partial, stream);
  (void)total;
  gn_finalize_fwd_kernel<<<dim3(B, (d + 15) / 16), 256, 0, stream>>>(partial, p.nblocks, B, d, weight, mean_scale, eps, mean, rstd, am, scale);
  if (r > 0) {
    if (p.sh.vec == 4)
      gn_apply_fwd_kernel<4><<<p.nblocks * 2, GN_THREADS, 0, stream>>>(x, batch, r, d, p.sh.lpr_log2, am, scale, bias, act_slope, y);
    else if (p.sh.vec == 2)
      gn_apply_fwd_kernel<2><<<p.nblocks * 2, GN_THREADS, 0, stream>>>(x, batch, r, d, p.sh.lpr_log2, am, scale, bias, act_slope, y);
    else
      gn_apply_fwd_kernel<1><<<p.nblocks * 2, GN_THREADS, 0, stream>>>(x, batch, r, d, p.sh.lpr_log2, am, scale, bias, act_slope, y);
  }
  SPT_CHECK_LAUNCH();
  return 0;
}

// The backward recounts rows per graph in its own statistics pass, so no
// forward state besides (mean, rstd) is needed.
// Forward WITHOUT the apply pass: statistics of x and the coefficient tables of
// y = (x - am[g]) * scale[g] + bias into caller buffers [num_graphs, d] - for consumers that
// apply the norm on the fly while they read x (the pre-norm in front of the attention block's
// qkv Linear: spt_skinny_linear_pre_f32).
extern "C" int spt_graphnorm_stats_f32(const float* x, const int64_t* batch, int64_t r, int d,
                                       int num_graphs, const float* weight,
                                       const float* mean_scale, float eps, float* mean,
                                       float* rstd, float* am, float* scale, void* ws,
                                       size_t ws_bytes, spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int B = num_graphs;
  SPT_CHECK_ARG(r >= 0 && d >= 1 && B >= 1, "bad shape");
  GnPlan p;
  SPT_CHECK_ARG(gn_plan(r, d, B, &p), "dim > 1024 unsupported");
  if (int e = gn_check(r, d, B, p, ws, ws_bytes)) return e;
  SPT_CHECK_ARG(weight && mean_scale && mean && rstd && am && scale && (r == 0 || x), "null pointer");
  double* partial = (double*)((char*)ws + p.off_partial);
  launch_stats<false>(p, x, nullptr, batch, r, d, B, nullptr, nullptr, nullptr, 1.f, partial, stream);
  gn_finalize_fwd_kernel<<<dim3(B, (d + 15) / 16), 256, 0, stream>>>(partial, p.nblocks, B, d, weight, mean_scale, eps, mean, rstd, am, scale);
  SPT_CHECK_LAUNCH();
  return 0;
}

extern "C" int spt_graphnorm_bwd_f32(const float* x, const float* gy,
                                     const int64_t* batch, int64_t r, int d,
                                     int num_graphs, const float* weight,
                                     const float* bias, const float* mean_scale,
                                     const float* mean, const float* rstd,
                                     float act_slope, float* gx, float* gweight,
                                     float* gbias, float* gmean_scale, void* ws,
                                     size_t ws_bytes, spt_stream_t stream_) {
  return spt_graphnorm_bwd_acc_f32(x, gy, batch, r, d, num_graphs, weight, bias, mean_scale, mean,
                                   rstd, act_slope, nullptr, gx, gweight, gbias, gmean_scale, ws,
                                   ws_bytes, stream_);
}
// Same, plus `gx_add` [r, d] (nullable): gx = gx_add + backward of the norm - the gradient of the
// residual branch around a pre-norm (x = x + f(norm(x)), src/nn/transformer.py:231-234) joins in
// the apply pass instead of a separate elementwise launch.
extern "C" int spt_graphnorm_bwd_acc_f32(const float* x, const float* gy,
                                         const int64_t* batch, int64_t r, int d,
                                         int num_graphs, const float* weight,
                                         const float* bias, const float* mean_scale,
                                         const float* mean, const float* rstd,
                                         float act_slope, const float* gx_add, float* gx,
                                         float* gweight, float* gbias, float* gmean_scale,
                                         void* ws, size_t ws_bytes, spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int B = num_graphs;
  SPT_CHECK_ARG(r >= 0 && d >= 1 && B >= 1, "bad shape");
  GnPlan p;
  SPT_CHECK_ARG(gn_plan(r, d, B, &p), "dim > 1024 unsupported");
  if (int e = gn_check(r, d, B, p, ws, ws_bytes)) return e;
  SPT_CHECK_ARG(weight && bias && mean_scale && mean && rstd && gweight && gbias && gmean_scale && (r == 0 || (x && gy && gx)), "null pointer");
  char* base = (char*)ws;
  double* partial = (double*)(base + p.off_partial);
  double* total = (double*)(base + p.off_total);
  float* am = (float*)(base + p.off_am);
  float* scale = (float*)(base + p.off_scale);
  float* c1 = (float*)(base + p.off_c1);
  float* c2 = (float*)(base + p.off_c2);
  float* c3 = (float*)(base + p.off_c3);
  // (alpha*mu, weight*rstd) are rebuilt from the saved statistics inside the statistics pass
  // (per graph change of a lane) and written out for the apply pass by the finalize kernel
  launch_stats<true>(p, x, gy, batch, r, d, B, nullptr, nullptr, bias, act_slope, partial, stream,
                     mean, rstd, weight, mean_scale);
  (void)total;
  gn_finalize_bwd_kernel<<<(d + 15) / 16, 256, 0, stream>>>(partial, p.nblocks, B, d, weight, mean_scale, mean, rstd, am, scale, c1, c2, c3, gweight, gbias, gmean_scale);
  if (r > 0) {
    if (p.sh.vec == 4)
      gn_apply_bwd_kernel<4><<<p.nblocks * 2, GN_THREADS, 0, stream>>>(x, gy, batch, r, d, p.sh.lpr_log2, am, scale, bias, act_slope, c1, c2, c3, gx, gx_add);
    else if (p.sh.vec == 2)
      gn_apply_bwd_kernel<2><<<p.nblocks * 2, GN_THREADS, 0, stream>>>(x, gy, batch, r, d, p.sh.lpr_log2, am, scale, bias, act_slope, c1, c2, c3, gx, gx_add);
    else
      gn_apply_bwd_kernel<1><<<p.nblocks * 2, GN_THREADS, 0, stream>>>(x, gy, batch, r, d, p.sh.lpr_log2, am, scale, bias, act_slope, c1, c2, c3, gx, gx_add);
  }
  SPT_CHECK_LAUNCH();
  return 0;
}

// ---- pieces of the two-pass scheme, exposed for the fused MLP layers (fused_mlp.hip),
//      which produce / consume the per-graph totals themselves -------------------------

// totals [B][2d+1] (sum x, sum x^2, row count) -> mean, rstd, am = alpha*mean, scale = w*rstd
extern "C" int spt_graphnorm_tables_f32(const double* total, int num_graphs, int d,
                                        const float* weight, const float* mean_scale,
                                        float eps, float* mean, float* rstd, float* am,
                                        float* scale, spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(num_graphs >= 1 && d >= 1, "bad shape");
  SPT_CHECK_ARG(total && weight && mean_scale && mean && rstd && am && scale, "null pointer");
  gn_fwd_tables_kernel<<<(num_graphs * d + 255) / 256, 256, 0, stream>>>(
      total, num_graphs, d, weight, mean_scale, eps, mean, rstd, am, scale);
  SPT_CHECK_LAUNCH();
  return 0;
}

// y = leaky((x - am[batch]) * scale[batch] + bias)
extern "C" int spt_graphnorm_apply_f32(const float* x, const int64_t* batch, int64_t r, int d,
                                       int num_graphs, const float* am, const float* scale,
                                       const float* bias, float act_slope, float* y,
                                       spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(r >= 0 && d >= 1 && num_graphs >= 1, "bad shape");
  GnPlan p;
  SPT_CHECK_ARG(gn_plan(r, d, num_graphs, &p), "dim > 1024 unsupported");
  if (r == 0) return 0;
  SPT_CHECK_ARG(x && y && am && scale && bias, "null pointer");
  if (p.sh.vec == 4)
    gn_apply_fwd_kernel<4><<<p.nblocks * 2, GN_THREADS, 0, stream>>>(x, batch, r, d, p.sh.lpr_log2, am, scale, bias, act_slope, y);
  else if (p.sh.vec == 2)
    gn_apply_fwd_kernel<2><<<p.nblocks * 2, GN_THREADS, 0, stream>>>(x, batch, r, d, p.sh.lpr_log2, am, scale, bias, act_slope, y);
  else
    gn_apply_fwd_kernel<1><<<p.nblocks * 2, GN_THREADS, 0, stream>>>(x, batch, r, d, p.sh.lpr_log2, am, scale, bias, act_slope, y);
  SPT_CHECK_LAUNCH();
  return 0;
}

// totals [B][2d+1] = (sum g, sum g*o, row count), g = gy * leaky'(y), o = x - am
extern "C" int spt_graphnorm_bwd_stats_f32(const float* x, const float* gy, const int64_t* batch,
                                           int64_t r, int d, int num_graphs, const float* am,
                                           const float* scale, const float* bias,
                                           float act_slope, double* total, void* ws,
                                           size_t ws_bytes, spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int B = num_graphs;
  SPT_CHECK_ARG(r >= 0 && d >= 1 && B >= 1, "bad shape");
  GnPlan p;
  SPT_CHECK_ARG(gn_plan(r, d, B, &p), "dim > 1024 unsupported");
  if (int e = gn_check(r, d, B, p, ws, ws_bytes)) return e;
  SPT_CHECK_ARG(total && am && scale && bias && (r == 0 || (x && gy)), "null pointer");
  double* partial = (double*)((char*)ws + p.off_partial);
  launch_stats<true>(p, x, gy, batch, r, d, B, am, scale, bias, act_slope, partial, stream);
  gn_reduce_partials_kernel<<<dim3(B, (p.row_len + 15) / 16), 256, 0, stream>>>(partial, p.nblocks, B, p.row_len, total);
  SPT_CHECK_LAUNCH();
  return 0;
}

// ---- the same totals when gy is the backward of a segment max-pool -------------------------
// gy then has ONE non-zero per (segment, channel): gy[arg[s,c], c] = gout[s,c].  The sums
// sum g' and sum g' o over the rows of a graph only see those entries, so they are computed
// from (gout, arg) and a gather of the raw rows - num_seg*d elements instead of a pass over
// the [r, d] tensors (15.4 GB at scene S).  Layout and meaning of `total` as above; the row
// count of graph b is graph_rows[b].
// RAWV: x is not the [n, d] layer output but raw[num_seg, d] = its value at the arg row (the
// pool's third output, spt_segcsr_max_affine_raw_f32): a stream instead of a 4-byte gather per
// (segment, channel); the same values, the same sums.
template <bool X16, bool RAWV = false>
__global__ __launch_bounds__(256) void gn_bwd_stats_sparse_kernel(
    const float* __restrict__ x, const float* __restrict__ gout,
    const int32_t* __restrict__ arg, const int64_t* __restrict__ seg_graph, int64_t num_seg,
    int64_t n, int d, int B, const float* __restrict__ am, const float* __restrict__ scale,
    const float* __restrict__ bias, float slope, double* __restrict__ partial,
    int b_lo, int Bc) {
  extern __shared__ __attribute__((aligned(16))) double tab[];
  const int row_len = 2 * d + 1;
  for (int i = threadIdx.x; i < Bc * row_len; i += 256) tab[i] = 0.0;
  __syncthreads();
  const int spb = 256 / d;                 // segments per block iteration (d divides 256)
  const int c = threadIdx.x % d, sub = threadIdx.x / d;
  double s1 = 0.0, s2 = 0.0;
  int cur = b_lo;
  float t_am = am[(size_t)b_lo * d + c], t_sc = scale[(size_t)b_lo * d + c];
  const float t_bs = bias[c];
  auto flush = [&]() {
    if (s1 != 0.0) atomicAdd(&tab[(size_t)(cur - b_lo) * row_len + c], s1);
    if (s2 != 0.0) atomicAdd(&tab[(size_t)(cur - b_lo) * row_len + d + c], s2);
    s1 = s2 = 0.0;
  };
  // four segments per trip: their (graph, arg, gradient, value) loads are requested together (round
  // 6: the loop was one 4-byte load per array and iteration - pure latency at 55 M elements)
  const int64_t stride = (int64_t)gridDim.x * spb;
  for (int64_t s0 = (int64_t)blockIdx.x * spb + sub; s0 < num_seg; s0 += 4 * stride) {
    bool ok[4];
    int bv[4];
    int64_t iv[4];
    float gv[4], xr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t s = s0 + u * stride;
      ok[u] = s < num_seg;
      const int64_t sc_ = ok[u] ? s : s0;                 // (a valid segment: unconditional loads)
      bv[u] = seg_graph ? (int)seg_graph[sc_] : 0;
      iv[u] = arg[sc_ * d + c];
      gv[u] = gout[sc_ * d + c];
      xr[u] = RAWV ? x[sc_ * d + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!ok[u]) continue;
      const int b = bv[u];
      if (b < b_lo || b >= b_lo + Bc) continue;   // another launch's graph
      if (b != cur) {
        flush();
        cur = b;
        t_am = am[(size_t)b * d + c];
        t_sc = scale[(size_t)b * d + c];
      }
      const int64_t i = iv[u];
      if (i < 0 || i >= n) continue;         // empty segment: sentinel n
      const float xv = RAWV ? xr[u]
                       : (X16 ? __uint_as_float((unsigned)reinterpret_cast<const uint16_t*>(x)[i * d + c] << 16)
                              : x[i * d + c]);
      const float o = xv - t_am;
      float g = gv[u];
      if (slope != 1.f) {
        const float y = fmaf(o, t_sc, t_bs);
        g = (y > 0.f) ? g : g * slope;
      }
      s1 += (double)g;
      s2 += (double)g * (double)o;
    }
  }
  flush();
  __syncthreads();
  double* out = partial + ((size_t)blockIdx.x * B + b_lo) * row_len;
  for (int i = threadIdx.x; i < Bc * row_len; i += 256) out[i] = tab[i];
}

__global__ void gn_set_row_counts_kernel(double* __restrict__ total, int B, int row_len,
                                         const int64_t* __restrict__ graph_rows) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) total[(size_t)b * row_len + row_len - 1] = (double)graph_rows[b];
}

extern "C" size_t spt_graphnorm_bwd_stats_sparse_workspace_bytes(int64_t num_seg, int d,
                                                                 int num_graphs) {
  if (num_seg < 0 || d < 1 || num_graphs < 1) return 0;
  return (size_t)1024 * num_graphs * (2 * d + 1) * 8;
}

extern "C" int spt_graphnorm_bwd_stats_sparse_f32(
    const float* x, const float* gout, const int32_t* arg, const int64_t* seg_graph,
    const int64_t* graph_rows, int64_t num_seg, int64_t n, int d, int num_graphs,
    const float* am, const float* scale, const float* bias, float act_slope, double* total,
    void* ws, size_t ws_bytes, spt_stream_t stream_) {
  return spt_graphnorm_bwd_stats_sparse_ex_f32(x, 0, gout, arg, seg_graph, graph_rows, num_seg, n, d,
                                               num_graphs, am, scale, bias, act_slope, total, ws,
                                               ws_bytes, stream_);
}
// x_is_bf16 = 1: x holds bf16 values (the fused layers' activation storage option); 2: x is the
// pool's raw output [num_seg, d] (f32)
extern "C" int spt_graphnorm_bwd_stats_sparse_ex_f32(
    const void* x_, int x_is_bf16, const float* gout, const int32_t* arg, const int64_t* seg_graph,
    const int64_t* graph_rows, int64_t num_seg, int64_t n, int d, int num_graphs,
    const float* am, const float* scale, const float* bias, float act_slope, double* total,
    void* ws, size_t ws_bytes, spt_stream_t stream_) {
  const float* x = reinterpret_cast<const float*>(x_);
  hipStream_t stream = (hipStream_t)stream_;
  const int B = num_graphs;
  SPT_CHECK_ARG(num_seg >= 0 && n >= 0 && B >= 1, "bad shape");
  SPT_CHECK_ARG(d >= 1 && d <= 256 && 256 % d == 0, "dim must divide 256");
  SPT_CHECK_ARG(total && am && scale && bias && graph_rows && x && gout && arg, "null pointer");
  const int row_len = 2 * d + 1;
  SPT_CHECK_ARG(ws && ws_bytes >= spt_graphnorm_bwd_stats_sparse_workspace_bytes(num_seg, d, B),
                "workspace too small");
  const int spb = 256 / d;
  int64_t nb = ceil_div(num_seg > 0 ? num_seg : 1, (int64_t)spb * 8);
  if (nb > 1024) nb = 1024;
  double* partial = (double*)ws;
  const int cap = gn_graphs_per_launch(row_len);
  for (int b_lo = 0; b_lo < B; b_lo += cap) {
    const int Bc = (B - b_lo < cap) ? B - b_lo : cap;
    if (x_is_bf16 == 2)         // x = raw[num_seg, d] (spt_graphnorm_bwd_stats_sparse_raw_f32)
      gn_bwd_stats_sparse_kernel<false, true><<<(int)nb, 256, (size_t)Bc * row_len * 8, stream>>>(
          x, gout, arg, seg_graph, num_seg, n, d, B, am, scale, bias, act_slope, partial, b_lo, Bc);
    else if (x_is_bf16)
      gn_bwd_stats_sparse_kernel<true><<<(int)nb, 256, (size_t)Bc * row_len * 8, stream>>>(
          x, gout, arg, seg_graph, num_seg, n, d, B, am, scale, bias, act_slope, partial, b_lo, Bc);
    else
      gn_bwd_stats_sparse_kernel<false><<<(int)nb, 256, (size_t)Bc * row_len * 8, stream>>>(
          x, gout, arg, seg_graph, num_seg, n, d, B, am, scale, bias, act_slope, partial, b_lo, Bc);
  }
  gn_reduce_partials_kernel<<<dim3(B, (row_len + 15) / 16), 256, 0, stream>>>(partial, (int)nb, B,
                                                                           row_len, total);
  gn_set_row_counts_kernel<<<(B + 63) / 64, 64, 0, stream>>>(total, B, row_len, graph_rows);
  SPT_CHECK_LAUNCH();
  return 0;
}

// The same totals from raw[num_seg, d] = the layer output at the arg rows, as the pool wrote it
// (spt_segcsr_max_affine_raw_f32): no gather.  n = rows of the layer (arg's sentinel for an empty
// segment).
extern "C" int spt_graphnorm_bwd_stats_sparse_raw_f32(
    const float* raw, const float* gout, const int32_t* arg, const int64_t* seg_graph,
    const int64_t* graph_rows, int64_t num_seg, int64_t n, int d, int num_graphs,
    const float* am, const float* scale, const float* bias, float act_slope, double* total,
    void* ws, size_t ws_bytes, spt_stream_t stream_) {
  return spt_graphnorm_bwd_stats_sparse_ex_f32(raw, 2, gout, arg, seg_graph, graph_rows, num_seg, n, d,
                                               num_graphs, am, scale, bias, act_slope, total, ws,
                                               ws_bytes, stream_);
}

// backward coefficient rows (gx = c1*g - c2*o - c3) and the three parameter gradients
extern "C" int spt_graphnorm_bwd_tables_f32(const double* total, int num_graphs, int d,
                                            const float* weight, const float* mean_scale,
                                            const float* mean, const float* rstd, float* c1,
                                            float* c2, float* c3, float* gweight, float* gbias,
                                            float* gmean_scale, spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(num_graphs >= 1 && d >= 1, "bad shape");
  SPT_CHECK_ARG(total && weight && mean_scale && mean && rstd && c1 && c2 && c3 && gweight &&
                gbias && gmean_scale, "null pointer");
  gn_bwd_tables_kernel<<<(d + 127) / 128, 128, 0, stream>>>(total, num_graphs, d, weight, mean_scale,
                                                            mean, rstd, c1, c2, c3, gweight, gbias,
                                                            gmean_scale);
  SPT_CHECK_LAUNCH();
  return 0;
}
