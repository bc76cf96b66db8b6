// Operand layout of the head-group decomposition of wider attention layouts (SPT-128: 16 heads of
// value dim 8 = two passes of the built 16 x (4, 4) kernels; ops._EdgeAttentionSplit, reference
// src/nn/attention.py:202-315 with the KITTI-360 widths of configs/experiment/semantic/kitti360.yaml:22-27).
// Round 6: the per-pass operand slabs and the gradient's way back were torch ops - an index gather
// plus a transposing copy in front of every block, a sum over the value slices, three permuting
// copies and a cat behind it: 7 launches and 0.07 ms per block at the train batch, where the step
// is made of such launches.  One kernel each way, 16 bytes per thread, both sides coalesced.
//
// H = 16 G heads of qk_dim 4 and value dim 4 J.  qkv row = [q (64 G) | k (64 G) | v (H x 4 J)].
// Pass p = g J + j reads [q_g (64) | k_g (64) | v_g[:, 4 j .. 4 j + 3] (16 heads x 4)] = 192 floats.
#include "common.hpp"

namespace spt {

// qa[p][i][:] <- the pass's 48 chunks of qkv[i]
__global__ __launch_bounds__(256) void attn_split_pack_kernel(
    const float4* __restrict__ qkv, int64_t n, int G, int J, float4* __restrict__ qa) {
  const int64_t total = (int64_t)G * J * n * 48;
  const int ld4 = 32 * G + 16 * G * J;                 // chunks per qkv row
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
    const int c = (int)(q % 48);
    const int64_t r = q / 48;                          // p * n + i
    const int64_t p = r / n, i = r - p * n;
    const int g = (int)(p / J), j = (int)(p - (int64_t)g * J);
    int src;
    if (c < 16) src = 16 * g + c;                                        // q of head group g
    else if (c < 32) src = 16 * G + 16 * g + (c - 16);                   // k
    else src = 32 * G + (16 * g + (c - 32)) * J + j;                     // v slice j of head c - 32
    qa[q] = qkv[i * ld4 + src];
  }
}

// gqkv[i][:] <- the passes' gradients: q / k columns summed over the J value slices of their head
// group (ascending j), v columns copied from their pass
__global__ __launch_bounds__(256) void attn_split_grad_kernel(
    const float4* __restrict__ gqa, int64_t n, int G, int J, float4* __restrict__ gqkv) {
  const int ld4 = 32 * G + 16 * G * J;
  const int64_t total = n * ld4;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
    const int64_t i = q / ld4;
    const int c = (int)(q - i * ld4);
    float4 v;
    if (c < 32 * G) {
      const int kq = c >= 16 * G;                      // 0: q columns, 1: k columns
      const int cc = c - 16 * G * kq, g = cc >> 4, w = cc & 15;
      const float4* s = gqa + ((int64_t)g * J * n + i) * 48 + 16 * kq + w;
      v = s[0];
      for (int j = 1; j < J; ++j) {
        const float4 t = s[(int64_t)j * n * 48];
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
    } else {
      const int cc = c - 32 * G, hj = cc / J, j = cc - hj * J;   // hj = 16 g + head
      const int g = hj >> 4, h = hj & 15;
      v = gqa[((int64_t)(g * J + j) * n + i) * 48 + 32 + h];
    }
    gqkv[q] = v;
  }
}

}  // namespace spt

extern "C" int spt_attn_split_pack_f32(const float* qkv, int64_t n, int G, int J, float* qa,
                                       spt_stream_t stream_) {
  SPT_CHECK_ARG(n >= 0 && G >= 1 && J >= 1 && G * J <= 64, "bad shape");
  if (n == 0) return 0;
  SPT_CHECK_ARG(qkv && qa, "null pointer");
  SPT_CHECK_ARG(((uintptr_t)qkv | (uintptr_t)qa) % 16 == 0, "qkv / qa must be 16-byte aligned");
  spt::attn_split_pack_kernel<<<spt::stream_grid((int64_t)G * J * n * 48, 256), 256, 0,
                                (hipStream_t)stream_>>>(
      reinterpret_cast<const float4*>(qkv), n, G, J, reinterpret_cast<float4*>(qa));
  SPT_CHECK_LAUNCH();
  return 0;
}

extern "C" int spt_attn_split_grad_f32(const float* gqa, int64_t n, int G, int J, float* gqkv,
                                       spt_stream_t stream_) {
  SPT_CHECK_ARG(n >= 0 && G >= 1 && J >= 1 && G * J <= 64, "bad shape");
  if (n == 0) return 0;
  SPT_CHECK_ARG(gqa && gqkv, "null pointer");
  SPT_CHECK_ARG(((uintptr_t)gqa | (uintptr_t)gqkv) % 16 == 0, "gqa / gqkv must be 16-byte aligned");
  spt::attn_split_grad_kernel<<<spt::stream_grid(n * (32 * G + 16 * G * J), 256), 256, 0,
                                (hipStream_t)stream_>>>(
      reinterpret_cast<const float4*>(gqa), n, G, J, reinterpret_cast<float4*>(gqkv));
  SPT_CHECK_LAUNCH();
  return 0;
}
