// y = x W^T + b for tall-skinny operands: rows in the 10^5..10^7, K, N <= 192.
//
// The qkv / out_proj Linears of every SelfAttentionBlock (src/nn/attention.py:202-215,
// 311-313: [N_nodes, 64] -> 192 and 64 -> 64) and their dX products are 10 GFLOP
// GEMMs whose operands stream once: ~0.1 ms of HBM / f32-MFMA time at 428 571 rows.  The
// library's tile selection runs them at 0.6-0.75 ms each (profiles/r01u: the
// Cijk_..._MT32x64x128 / MT64x32x128 rows, 9 ms per step together).  This kernel is the
// forward half of the fused MLP layer without the GraphNorm plumbing: one wave owns a
// 16-row tile x 64-column slab, W's slab lives in B-operand registers for the whole
// launch, the A tile is staged through LDS with the next tile's global loads in flight
// during the MFMAs (v_mfma_f32_16x16x4_f32: f32 in, f32 accumulate).
//
// dX = G W is the same kernel on (G, W^T); dW stays a batched library GEMM (ops.py).
#include "common.hpp"

namespace spt {
namespace skinny {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TR = 16;      // rows per tile (MFMA M)
constexpr int WAVES = 4;
constexpr int SLAB = 64;    // output columns per wave (4 MFMA column blocks)

__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// K = 4 K4.  blockIdx.y = column slab; waves of blockIdx.x stride the row tiles.
template <int K4>
__global__ __launch_bounds__(WAVES * 64, (K4 <= 32) ? 2 : 1) void skinny_linear_kernel(
    const float* __restrict__ x, int64_t rows, const float* __restrict__ W,
    const float* __restrict__ bias, int N, float* __restrict__ y) {
  constexpr int K = 4 * K4, LDA = K + 4, V = K4 / 4;   // V float4 per lane per tile
  __shared__ __attribute__((aligned(16))) float a_lds[WAVES][TR * LDA];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int n0 = blockIdx.y * SLAB;
  float* al = a_lds[wid];

  float B[4][K4];                                       // lane (g, c): W[n0 + 16 nb + c][4 st + g]
  float bb[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const float* wr = W + (size_t)(n0 + 16 * nb + c) * K;
#pragma unroll
    for (int st = 0; st < K4; ++st) B[nb][st] = wr[4 * st + g];
    bb[nb] = bias ? bias[n0 + 16 * nb + c] : 0.f;
  }

  const int64_t ntiles = (rows + TR - 1) / TR;
  const int64_t wave = (int64_t)blockIdx.x * WAVES + wid;
  const int64_t nwaves = (int64_t)gridDim.x * WAVES;

  // the tile [16, K] is one contiguous run of 16 K floats: lane l takes float4 l, l+64, ...
  float4 nx[V];
  auto fetch = [&](int64_t t) {
    const int64_t base = t * TR * (int64_t)K;
    const int64_t lim = rows * (int64_t)K;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int64_t e = base + (int64_t)(v * 64 + lane) * 4;
      nx[v] = (e < lim) ? *reinterpret_cast<const float4*>(x + e)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  if (wave < ntiles) fetch(wave);
  for (int64_t t = wave; t < ntiles; t += nwaves) {
    wave_sync_lds();
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int q = (v * 64 + lane) * 4;                // element index inside the tile
      const int rr = q / K, k = q - rr * K;
      *reinterpret_cast<float4*>(al + rr * LDA + k) = nx[v];
    }
    wave_sync_lds();
    if (t + nwaves < ntiles) fetch(t + nwaves);         // in flight during the MFMAs
    float A[K4];
#pragma unroll
    for (int st = 0; st < K4; ++st) A[st] = al[c * LDA + 4 * st + g];
    f32x4 C[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) C[nb] = (f32x4){bb[nb], bb[nb], bb[nb], bb[nb]};
#pragma unroll
    for (int st = 0; st < K4; ++st)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
        C[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[st], B[nb][st], C[nb], 0, 0, 0);
    const int64_t row0 = t * TR;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = row0 + 4 * g + r;
      if (row < rows) {
        float* yr = y + row * N + n0 + c;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) __builtin_nontemporal_store(C[nb][r], yr + 16 * nb);
      }
    }
  }
}

}  // namespace skinny
}  // namespace spt

using namespace spt;
using namespace spt::skinny;

extern "C" int spt_skinny_linear_supported(int K, int N) {
  return (K == 32 || K == 64 || K == 128 || K == 192) && N >= SLAB && N % SLAB == 0 && N <= 1024;
}

extern "C" int spt_skinny_linear_f32(const float* x, int64_t rows, int K, const float* W,
                                     const float* bias, int N, float* y, spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(rows >= 0, "bad shape");
  SPT_CHECK_ARG(spt_skinny_linear_supported(K, N), "(K, N) not built");
  if (rows == 0) return 0;
  SPT_CHECK_ARG(x && W && y, "null pointer");
  const int slabs = N / SLAB;
  const int64_t tiles = ceil_div(rows, TR);
  int64_t bx = ceil_div(tiles, WAVES);
  const int64_t cap = (int64_t)256 * 8 / slabs > 1 ? (int64_t)256 * 8 / slabs : 1;
  if (bx > cap) bx = cap;
  const dim3 grid((unsigned)bx, (unsigned)slabs);
  switch (K) {
    case 32:  skinny_linear_kernel<8><<<grid, WAVES * 64, 0, stream>>>(x, rows, W, bias, N, y); break;
    case 64:  skinny_linear_kernel<16><<<grid, WAVES * 64, 0, stream>>>(x, rows, W, bias, N, y); break;
    case 128: skinny_linear_kernel<32><<<grid, WAVES * 64, 0, stream>>>(x, rows, W, bias, N, y); break;
    default:  skinny_linear_kernel<48><<<grid, WAVES * 64, 0, stream>>>(x, rows, W, bias, N, y); break;
  }
  SPT_CHECK_LAUNCH();
  return 0;
}
