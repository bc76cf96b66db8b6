// Cross-entropy of the classifier heads' logits (the criterion of the train step:
// configs/model/semantic/default.yaml:47-49 torch.nn.CrossEntropyLoss(ignore_index=num_classes),
// applied per output level in src/models/semantic.py; mean over the rows that are not ignored).
//
// [rows, C] logits with C = 13..32 classes: one lane per row keeps the row in registers, so
// log-sum-exp, the picked logit and (in the backward) the softmax cost one pass; the library's
// nll_loss forward / backward reduce with a single workgroup (0.26-0.42 ms per call at 428 571
// rows, profiles/r02z).  Sums are accumulated in f64 per workgroup and added up in a fixed order
// by a second kernel: deterministic.
#include <math.h>

#include "common.hpp"

namespace spt {

constexpr int CE_MAXC = 32;
constexpr int CE_BLOCK = 256;

template <int CMAX>
__global__ __launch_bounds__(CE_BLOCK) void ce_fwd_kernel(
    const float* __restrict__ logits, const int64_t* __restrict__ target, int64_t rows, int C,
    int64_t ignore_index, float* __restrict__ lse, double* __restrict__ partial) {
  __shared__ double s_sum[CE_BLOCK / 64], s_cnt[CE_BLOCK / 64];
  const int64_t row = (int64_t)blockIdx.x * CE_BLOCK + threadIdx.x;
  double li = 0.0, ci = 0.0;
  if (row < rows) {
    float v[CMAX];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      v[c] = (c < C) ? logits[row * C + c] : -INFINITY;
      m = fmaxf(m, v[c]);
    }
    float z = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) z += (c < C) ? expf(v[c] - m) : 0.f;
    const float l = m + logf(z);
    lse[row] = l;
    const int64_t t = target[row];
    if (t != ignore_index && t >= 0 && t < C) {
      float picked = 0.f;
#pragma unroll
      for (int c = 0; c < CMAX; ++c) picked = (c == (int)t) ? v[c] : picked;
      li = (double)(l - picked);
      ci = 1.0;
    } else if (t != ignore_index) {
      // a label outside [0, C) that is not ignore_index is a bug upstream (wrong num_classes,
      // unshifted void label): torch raises a device assert; here the loss comes out NaN instead
      // of silently averaging over fewer rows
      li = (double)NAN;
      ci = 1.0;
    }
  }
  // wave sums, then the block's four waves in order
  for (int o = 32; o > 0; o >>= 1) {
    li += __shfl_xor(li, o, 64);
    ci += __shfl_xor(ci, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    s_sum[threadIdx.x >> 6] = li;
    s_cnt[threadIdx.x >> 6] = ci;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < CE_BLOCK / 64; ++w) {
      a += s_sum[w];
      b += s_cnt[w];
    }
    partial[2 * (size_t)blockIdx.x] = a;
    partial[2 * (size_t)blockIdx.x + 1] = b;
  }
}

// loss = sum / max(count, 1) (0 when every row is ignored, like torch's nan-free convention is
// NOT: torch returns nan there; we return nan too by dividing 0 / 0 only when count == 0)
__global__ __launch_bounds__(256) void ce_finish_kernel(const double* __restrict__ partial, int nblocks,
                                                        float* __restrict__ loss,
                                                        float* __restrict__ count) {
  __shared__ double s_a[256], s_b[256];
  double a = 0.0, b = 0.0;
  const int per = (nblocks + 255) / 256;
  const int lo = threadIdx.x * per, hi = (lo + per < nblocks) ? lo + per : nblocks;
  for (int i = lo; i < hi; ++i) {
    a += partial[2 * (size_t)i];
    b += partial[2 * (size_t)i + 1];
  }
  s_a[threadIdx.x] = a;
  s_b[threadIdx.x] = b;
  __syncthreads();
  if (threadIdx.x == 0) {
    double ta = 0.0, tb = 0.0;
    for (int i = 0; i < 256; ++i) {
      ta += s_a[i];
      tb += s_b[i];
    }
    loss[0] = (float)(ta / tb);
    count[0] = (float)tb;
  }
}

template <int CMAX>
__global__ __launch_bounds__(CE_BLOCK) void ce_bwd_kernel(
    const float* __restrict__ logits, const int64_t* __restrict__ target,
    const float* __restrict__ lse, int64_t rows, int C, int64_t ignore_index,
    const float* __restrict__ gout, const float* __restrict__ count, float* __restrict__ glogits) {
  const int64_t row = (int64_t)blockIdx.x * CE_BLOCK + threadIdx.x;
  if (row >= rows) return;
  const float scale = gout[0] / count[0];
  const int64_t t = target[row];
  const bool valid = t != ignore_index && t >= 0 && t < C;
  const float l = lse[row];
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    if (c < C) {
      const float p = expf(logits[row * C + c] - l);
      glogits[row * C + c] = valid ? (p - ((int)t == c ? 1.f : 0.f)) * scale : 0.f;
    }
  }
}

}  // namespace spt

using namespace spt;

extern "C" size_t spt_cross_entropy_workspace_bytes(int64_t rows) {
  return (size_t)(ceil_div(rows > 0 ? rows : 1, (int64_t)CE_BLOCK)) * 2 * sizeof(double);
}

// loss[0] = mean over the rows with target != ignore_index of (logsumexp(logits[row]) -
// logits[row, target[row]]); lse[rows] is kept for the backward; count[0] = number of such rows.
extern "C" int spt_cross_entropy_fwd_f32(const float* logits, const int64_t* target, int64_t rows,
                                         int C, int64_t ignore_index, float* lse, float* loss,
                                         float* count, void* ws, size_t ws_bytes,
                                         spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(rows >= 1 && C >= 1 && C <= CE_MAXC, "rows >= 1, 1 <= C <= 32");
  SPT_CHECK_ARG(logits && target && lse && loss && count, "null pointer");
  SPT_CHECK_ARG(ws && ws_bytes >= spt_cross_entropy_workspace_bytes(rows), "workspace too small");
  const int nblocks = (int)ceil_div(rows, (int64_t)CE_BLOCK);
  double* partial = (double*)ws;
  if (C <= 16)
    ce_fwd_kernel<16><<<nblocks, CE_BLOCK, 0, stream>>>(logits, target, rows, C, ignore_index, lse, partial);
  else
    ce_fwd_kernel<32><<<nblocks, CE_BLOCK, 0, stream>>>(logits, target, rows, C, ignore_index, lse, partial);
  ce_finish_kernel<<<1, 256, 0, stream>>>(partial, nblocks, loss, count);
  SPT_CHECK_LAUNCH();
  return 0;
}

// glogits[row, c] = (softmax(logits[row])[c] - [c == target[row]]) * gout[0] / count[0], 0 for
// ignored rows.  gout and count are DEVICE scalars (no host round trip).
extern "C" int spt_cross_entropy_bwd_f32(const float* logits, const int64_t* target,
                                         const float* lse, int64_t rows, int C,
                                         int64_t ignore_index, const float* gout,
                                         const float* count, float* glogits, spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(rows >= 1 && C >= 1 && C <= CE_MAXC, "rows >= 1, 1 <= C <= 32");
  SPT_CHECK_ARG(logits && target && lse && gout && count && glogits, "null pointer");
  const int nblocks = (int)ceil_div(rows, (int64_t)CE_BLOCK);
  if (C <= 16)
    ce_bwd_kernel<16><<<nblocks, CE_BLOCK, 0, stream>>>(logits, target, lse, rows, C, ignore_index, gout, count, glogits);
  else
    ce_bwd_kernel<32><<<nblocks, CE_BLOCK, 0, stream>>>(logits, target, lse, rows, C, ignore_index, gout, count, glogits);
  SPT_CHECK_LAUNCH();
  return 0;
}
