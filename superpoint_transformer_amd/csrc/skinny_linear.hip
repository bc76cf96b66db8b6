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
// dX = G W is the same kernel on (G, W^T).  dW = G^T X (a reduction over 10^5..10^7 rows into a
// <= 192 x 64 block) has its own kernel below: the library ran it as a batched GEMM over row
// chunks at ~0.36 ms per 10-GFLOP product (profiles/r02z: the Cijk_..._MT16x32x512 rows).
#include "common.hpp"

namespace spt {
namespace skinny {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TR = 16;      // rows per tile (MFMA M)
constexpr int WAVES = 4;
constexpr int SLAB = 64;    // output columns per wave (4 MFMA column blocks)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// Round 6: the products on the bf16 matrix pipe, like every other GEMM of the default mode
// (precision.py: the reference ships float32_matmul_precision: high, configs/train.yaml:60-61).
// PR = planes-and-products scheme of a kernel instance:
//   0  f32 pipe (v_mfma_f32_16x16x4_f32; the f32-exact mode, and every shape without a bf16 instance)
//   1  operands rounded to bf16, one product                                  (the bf16 mode)
//   3  x = hi + lo, w = hi + lo: lo*hi + hi*lo + hi*hi (~2^-17 per product)   (the default mode's backward)
//   6  3-way splits, six products, smallest first: f32-exact                  (the default mode's forward)
// A 16 x K x 64 tile step on the f32 pipe costs 8 x 32 cycles per 32 columns of K; the bf16 pipe
// 16 cycles per product: 1 / 16, 3 / 16 or 6 / 16 of it.
__host__ __device__ constexpr int skinny_planes(int PR) { return PR == 6 ? 3 : (PR == 3 ? 2 : 1); }
template <int NPL>
__device__ __forceinline__ void split_planes(const float (&x)[8], bf16x8 (&p)[NPL]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 a = (__bf16)x[i];
    p[0][i] = a;
    if constexpr (NPL >= 2) {
      const float r1 = x[i] - (float)a;                 // exact
      const __bf16 b = (__bf16)r1;
      p[1][i] = b;
      if constexpr (NPL >= 3) p[2][i] = (__bf16)(r1 - (float)b);   // exact, fits bf16
    }
  }
}
// C += x w over one 32-column step, planes x[0] (hi) .. and w[0] (hi) ..
template <int PR>
__device__ __forceinline__ f32x4 mfma_planes(const bf16x8 (&x)[skinny_planes(PR)],
                                             const bf16x8 (&w)[skinny_planes(PR)], f32x4 c) {
  if constexpr (PR == 6) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[2], w[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[0], w[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[1], w[1], c, 0, 0, 0);
  }
  if constexpr (PR >= 3) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[1], w[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[0], w[1], c, 0, 0, 0);
  }
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[0], w[0], c, 0, 0, 0);
}

__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// K = 4 K4.  blockIdx.y = column slab of NBS 16-column blocks (4: N a multiple of 64; 1: the
// narrow heads, N <= 16 - the classifiers' 13 classes - with the missing columns read as zero
// and not stored); waves of blockIdx.x stride the row tiles.
// PRE: the rows of x go through a per-graph affine map on their way into the A tile,
//   x_n = (x - pam[g][k]) * psc[g][k] + pbs[k],  g = batch[row] (0 when batch is null)
// - the pre-norm GraphNorm in front of the attention block's qkv Linear (src/nn/transformer.py:
// 231-234) applied while x is read, bitwise the values gn_apply_fwd_kernel would have written
// (same fmaf), so the normalised [rows, K] tensor is never materialised.  Tables of PB graphs
// live in LDS.  RES: y = (x W^T + b) + res (the block's residual `shortcut + out_proj(.)`).
constexpr int PRE_MAX = 1024;  // floats per coefficient table in LDS (num_graphs x K)
// -DSPT_SKINNY_NO_MFMA (measurement build, tools/build_variant.sh: wrong results by design): the
// products replaced by one add per MFMA - what the kernels cost without the f32 matrix pipe
#ifdef SPT_SKINNY_NO_MFMA
__device__ __forceinline__ f32x4 skinny_fake_mfma(float a, float b, f32x4 c) {
  c[0] += a * b;
  return c;
}
#define SPT_SKINNY_MFMA(a, b, c) skinny_fake_mfma(a, b, c)
#else
#define SPT_SKINNY_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
#endif

// Round 6: the slab of W given as the TRANSPOSE of what the product needs (`Wt` [K][ldwt] row-major:
// slab row rr, column k <- Wt[k][n0 + rr]) - dX = G W of a Linear's backward reads the layer's own
// weight [N_out = K here][N_in = ldwt] instead of a transposed copy made by a torch launch per
// backward call (14 per SPT-64 step, 34 per SPT-128 step).  Coalesced float4 loads along a row of
// Wt, four LDS writes each; N % 4 == 0 and ldwt % 4 == 0 (16-byte aligned rows).
template <int NBS, int K, int LDA>
__device__ __forceinline__ void stage_slab_transposed(const float* __restrict__ Wt, int ldwt, int n0,
                                                      int N, float* wl, int tid, int nthreads) {
  constexpr int R4 = 4 * NBS;                           // float4 per slab row of Wt
  for (int q = tid; q < K * R4; q += nthreads) {
    const int k = q / R4, r4 = q - k * R4;
    const int n = n0 + 4 * r4;
    const float4 v = n < N ? *reinterpret_cast<const float4*>(Wt + (size_t)k * ldwt + n)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
    float* d = wl + (4 * r4) * LDA + k;
    d[0] = v.x;
    d[LDA] = v.y;
    d[2 * LDA] = v.z;
    d[3 * LDA] = v.w;
  }
}

template <int K4, int NBS = 4, bool PRE = false, bool RES = false, int PR = 0>
__global__ __launch_bounds__(WAVES * 64, (K4 < 32 || (K4 == 32 && !PRE)) ? 2 : 1) void skinny_linear_kernel(
    const float* __restrict__ x, int64_t rows, const float* __restrict__ W,
    const float* __restrict__ bias, int N, float* __restrict__ y,
    const float* __restrict__ pam = nullptr, const float* __restrict__ psc = nullptr,
    const float* __restrict__ pbs = nullptr, const int64_t* __restrict__ batch = nullptr, int PB = 1,
    const float* __restrict__ res = nullptr, int ldwt = 0) {
  constexpr int K = 4 * K4, LDA = K + 4, V = K4 / 4;   // V float4 per lane per tile
  __shared__ __attribute__((aligned(16))) float a_lds[WAVES][TR * LDA];
  __shared__ __attribute__((aligned(16))) float ptab[PRE ? 2 * PRE_MAX + K : 4];   // am | sc | bias
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int n0 = blockIdx.y * (16 * NBS);
  float* al = a_lds[wid];

  // W's slab [16 NBS x K] reaches the B-operand registers through LDS: the workgroup reads it
  // once with coalesced float4 loads (the waves' A-tile buffers double as the staging area: the
  // slab is 16 NBS x K <= WAVES x 16 x K floats), each lane then picks its K4 x NBS values with
  // LDS reads.  (Per-lane dword loads straight from global - 16 rows x 16 B per instruction, K4 x
  // NBS instructions per wave - cost more than the tile's MFMAs at train-batch row counts, where
  // a wave only sees one or two tiles.)
  constexpr int KS = K / 32, NPL = skinny_planes(PR);
  static_assert(PR == 0 || K % 32 == 0, "bf16 instances: whole 32-column steps");
  float B[PR == 0 ? NBS : 1][PR == 0 ? K4 : 1];         // lane (g, c): W[n0 + 16 nb + c][4 st + g]
  bf16x8 Bp[PR == 0 ? 1 : NBS][PR == 0 ? 1 : KS][NPL];  // lane (g, c): W[n0 + 16 nb + c][32 ks + 8 g ..]
  float bb[NBS];
  {
    float* wl = &a_lds[0][0];                           // [16 NBS][LDA]
    constexpr int WV = 16 * NBS * K4;                   // float4 of the slab
    if (ldwt > 0) {
      stage_slab_transposed<NBS, K, LDA>(W, ldwt, n0, N, wl, threadIdx.x, WAVES * 64);
    } else {
      for (int q = threadIdx.x; q < WV; q += WAVES * 64) {
        const int rr = q / K4, k4 = q - rr * K4;
        const bool nv = n0 + rr < N;
        const float4 wv = nv ? *reinterpret_cast<const float4*>(W + (size_t)(n0 + rr) * K + 4 * k4)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(wl + rr * LDA + 4 * k4) = wv;
      }
    }
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < NBS; ++nb) {
      if constexpr (PR == 0) {
#pragma unroll
        for (int st = 0; st < K4; ++st) B[nb][st] = wl[(16 * nb + c) * LDA + 4 * st + g];
      } else {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const float4 w0 = *reinterpret_cast<const float4*>(wl + (16 * nb + c) * LDA + 32 * ks + 8 * g);
          const float4 w1 = *reinterpret_cast<const float4*>(wl + (16 * nb + c) * LDA + 32 * ks + 8 * g + 4);
          const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
          split_planes<NPL>(wv, Bp[nb][ks]);
        }
      }
      bb[nb] = (bias && n0 + 16 * nb + c < N) ? bias[n0 + 16 * nb + c] : 0.f;
    }
    if constexpr (PRE) {
      for (int i = threadIdx.x; i < PB * K; i += WAVES * 64) {
        ptab[i] = pam[i];
        ptab[PRE_MAX + i] = psc[i];
      }
      for (int i = threadIdx.x; i < K; i += WAVES * 64) ptab[2 * PRE_MAX + i] = pbs[i];
    }
    __syncthreads();
  }

  const int64_t ntiles = (rows + TR - 1) / TR;
  const int64_t wave = (int64_t)blockIdx.x * WAVES + wid;
  const int64_t nwaves = (int64_t)gridDim.x * WAVES;

  // the tile [16, K] is one contiguous run of 16 K floats: lane l takes float4 l, l+64, ...
  float4 nx[V];
  int ng[V];                                            // graph of the row each float4 belongs to
  auto fetch = [&](int64_t t) {
    const int64_t base = t * TR * (int64_t)K;
    const int64_t lim = rows * (int64_t)K;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int64_t e = base + (int64_t)(v * 64 + lane) * 4;
      nx[v] = (e < lim) ? *reinterpret_cast<const float4*>(x + e)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (PRE) {
        const int64_t row = t * TR + ((v * 64 + lane) * 4) / K;
        ng[v] = (batch && row < rows) ? (int)batch[row] : 0;
      }
    }
  };
  if (wave < ntiles) fetch(wave);
  for (int64_t t = wave; t < ntiles; t += nwaves) {
    // the residual values of this tile's outputs (C layout), requested before the staging
    float rv[RES ? NBS : 1][4];
    if constexpr (RES) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = t * TR + 4 * g + r;
#pragma unroll
        for (int nb = 0; nb < NBS; ++nb)
          rv[nb][r] = (row < rows && n0 + 16 * nb + c < N) ? res[row * N + n0 + c + 16 * nb] : 0.f;
      }
    }
    wave_sync_lds();
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int q = (v * 64 + lane) * 4;                // element index inside the tile
      const int rr = q / K, k = q - rr * K;
      float4 w = nx[v];
      if constexpr (PRE) {
        const float4 a = *reinterpret_cast<const float4*>(ptab + ng[v] * K + k);
        const float4 sc4 = *reinterpret_cast<const float4*>(ptab + PRE_MAX + ng[v] * K + k);
        const float4 b4 = *reinterpret_cast<const float4*>(ptab + 2 * PRE_MAX + k);
        w.x = fmaf(w.x - a.x, sc4.x, b4.x); w.y = fmaf(w.y - a.y, sc4.y, b4.y);
        w.z = fmaf(w.z - a.z, sc4.z, b4.z); w.w = fmaf(w.w - a.w, sc4.w, b4.w);
      }
      *reinterpret_cast<float4*>(al + rr * LDA + k) = w;
    }
    wave_sync_lds();
    if (t + nwaves < ntiles) fetch(t + nwaves);         // in flight during the MFMAs
    f32x4 C[NBS];
#pragma unroll
    for (int nb = 0; nb < NBS; ++nb) C[nb] = (f32x4){bb[nb], bb[nb], bb[nb], bb[nb]};
    if constexpr (PR == 0) {
      float A[K4];
#pragma unroll
      for (int st = 0; st < K4; ++st) A[st] = al[c * LDA + 4 * st + g];
#pragma unroll
      for (int st = 0; st < K4; ++st)
#pragma unroll
        for (int nb = 0; nb < NBS; ++nb)
          C[nb] = SPT_SKINNY_MFMA(A[st], B[nb][st], C[nb]);
    } else {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float4 a0 = *reinterpret_cast<const float4*>(al + c * LDA + 32 * ks + 8 * g);
        const float4 a1 = *reinterpret_cast<const float4*>(al + c * LDA + 32 * ks + 8 * g + 4);
        const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        bf16x8 Ap[NPL];
        split_planes<NPL>(av, Ap);
#pragma unroll
        for (int nb = 0; nb < NBS; ++nb) C[nb] = mfma_planes<PR>(Ap, Bp[nb][ks], C[nb]);
      }
    }
    const int64_t row0 = t * TR;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = row0 + 4 * g + r;
      if (row < rows) {
        float* yr = y + row * N + n0 + c;
#pragma unroll
        for (int nb = 0; nb < NBS; ++nb)
          if (n0 + 16 * nb + c < N)                       // (the last slab of an N % 64 != 0 output)
            __builtin_nontemporal_store(RES ? C[nb][r] + rv[nb][r] : C[nb][r], yr + 16 * nb);
      }
    }
  }
}

// K = 192 (dX of the qkv Linear): 192 B-operand registers leave one wave per SIMD.  Here W's slab
// stays in LDS for the whole launch and every MFMA takes its B operand from there (one 4-byte
// LDS read per 32-cycle MFMA), 8-wave workgroups share the slab: two waves per SIMD.
constexpr int WAVES_L = 8;
// Also the kernel of the WIDE inputs (K = 132 / 260: [diameter | pos | x] and [.. | x_up | x_skip] of
// the KITTI-360 width's node MLPs, configs/experiment/semantic/kitti360.yaml:22-27 - 264 B-operand
// registers otherwise): any K a multiple of 4 (a tile is TR K / 4 16-byte chunks, the last wave-load
// of a tile partly idle), NWL waves per workgroup so that slab + tiles fit the LDS.
// PR > 0 (round 6, K % 32 == 0): the slab is kept as bf16 planes (hi | lo, rows of K + 8 values:
// conflict-free 16-byte reads) and every product runs on the bf16 pipe (see skinny_planes above).
template <int K4, int NWL = WAVES_L, int PR = 0>
__global__ __launch_bounds__(NWL * 64, (NWL >= 4 ? NWL / 4 : 1)) void skinny_linear_wlds_kernel(
    const float* __restrict__ x, int64_t rows, const float* __restrict__ W,
    const float* __restrict__ bias, int N, float* __restrict__ y, int ldwt = 0) {
  constexpr int K = 4 * K4, LDA = K + 4, NCH = TR * K4, V = (NCH + 63) / 64, NBS = 4;
  constexpr int NPL = skinny_planes(PR), LDW = K + 8, KS = K / 32;
  static_assert(PR == 0 || (K % 32 == 0 && NPL <= 2), "bf16 instances: whole 32-column steps, hi | lo");
  // (one buffer: the f32 slab while it is staged, the bf16 planes afterwards - 2 x 2 (K + 8) bytes
  // per row against 4 (K + 4))
  __shared__ __attribute__((aligned(16))) float w_lds[16 * NBS * LDA + (PR ? 16 * NBS * 4 : 0)];
  __shared__ __attribute__((aligned(16))) float a_lds[NWL][TR * LDA];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int n0 = blockIdx.y * (16 * NBS);
  float* al = a_lds[wid];
  // PR > 0: the f32 slab is staged in the (still unused) tile buffers and leaves as bf16 planes
  static_assert(PR == 0 || NWL * TR >= 16 * NBS, "the slab fits the tile buffers");
  float* slab = PR ? &a_lds[0][0] : w_lds;
  if (ldwt > 0) {
    stage_slab_transposed<NBS, K, LDA>(W, ldwt, n0, N, slab, threadIdx.x, NWL * 64);
  } else {
    for (int q = threadIdx.x; q < 16 * NBS * K4; q += NWL * 64) {
      const int rr = q / K4, k4 = q - rr * K4;
      *reinterpret_cast<float4*>(slab + rr * LDA + 4 * k4) =
          (n0 + rr < N) ? *reinterpret_cast<const float4*>(W + (size_t)(n0 + rr) * K + 4 * k4)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  float bb[NBS];
#pragma unroll
  for (int nb = 0; nb < NBS; ++nb) bb[nb] = (bias && n0 + 16 * nb + c < N) ? bias[n0 + 16 * nb + c] : 0.f;
  __syncthreads();
  __bf16* wpl = reinterpret_cast<__bf16*>(w_lds);       // [NPL][16 NBS][LDW]
  if constexpr (PR != 0) {
    for (int q = threadIdx.x; q < 16 * NBS * (K / 8); q += NWL * 64) {
      const int rr = q / (K / 8), k8 = q - rr * (K / 8);
      const float4 w0 = *reinterpret_cast<const float4*>(slab + rr * LDA + 8 * k8);
      const float4 w1 = *reinterpret_cast<const float4*>(slab + rr * LDA + 8 * k8 + 4);
      const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
      bf16x8 pl[NPL];
      split_planes<NPL>(wv, pl);
#pragma unroll
      for (int p = 0; p < NPL; ++p)
        *reinterpret_cast<bf16x8*>(wpl + (size_t)p * 16 * NBS * LDW + rr * LDW + 8 * k8) = pl[p];
    }
    __syncthreads();
  }
  const float* wl = w_lds + c * LDA + g;                // + 16 nb LDA + 4 st

  const int64_t ntiles = (rows + TR - 1) / TR;
  const int64_t wave = (int64_t)blockIdx.x * NWL + wid;
  const int64_t nwaves = (int64_t)gridDim.x * NWL;
  float4 nx[V];
  auto fetch = [&](int64_t t) {
    const int64_t base = t * TR * (int64_t)K;
    const int64_t lim = rows * (int64_t)K;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int64_t e = base + (int64_t)(v * 64 + lane) * 4;
      nx[v] = (v * 64 + lane < NCH && e < lim) ? *reinterpret_cast<const float4*>(x + e)
                                               : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  if (wave < ntiles) fetch(wave);
  for (int64_t t = wave; t < ntiles; t += nwaves) {
    wave_sync_lds();
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int q = (v * 64 + lane) * 4;
      const int rr = q / K, k = q - rr * K;
      if (v * 64 + lane < NCH) *reinterpret_cast<float4*>(al + rr * LDA + k) = nx[v];
    }
    wave_sync_lds();
    if (t + nwaves < ntiles) fetch(t + nwaves);         // in flight during the MFMAs
    f32x4 C[NBS];
#pragma unroll
    for (int nb = 0; nb < NBS; ++nb) C[nb] = (f32x4){bb[nb], bb[nb], bb[nb], bb[nb]};
    if constexpr (PR == 0) {
#pragma unroll
      for (int st = 0; st < K4; ++st) {
        const float a = al[c * LDA + 4 * st + g];
#pragma unroll
        for (int nb = 0; nb < NBS; ++nb)
          C[nb] = SPT_SKINNY_MFMA(a, wl[16 * nb * LDA + 4 * st], C[nb]);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float4 a0 = *reinterpret_cast<const float4*>(al + c * LDA + 32 * ks + 8 * g);
        const float4 a1 = *reinterpret_cast<const float4*>(al + c * LDA + 32 * ks + 8 * g + 4);
        const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        bf16x8 Ap[NPL];
        split_planes<NPL>(av, Ap);
#pragma unroll
        for (int nb = 0; nb < NBS; ++nb) {
          bf16x8 Bp[NPL];
#pragma unroll
          for (int p = 0; p < NPL; ++p)
            Bp[p] = *reinterpret_cast<const bf16x8*>(wpl + (size_t)p * 16 * NBS * LDW +
                                                     (16 * nb + c) * LDW + 32 * ks + 8 * g);
          C[nb] = mfma_planes<PR>(Ap, Bp, C[nb]);
        }
      }
    }
    const int64_t row0 = t * TR;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = row0 + 4 * g + r;
      if (row < rows) {
        float* yr = y + row * N + n0 + c;
#pragma unroll
        for (int nb = 0; nb < NBS; ++nb)
          if (n0 + 16 * nb + c < N) __builtin_nontemporal_store(C[nb][r], yr + 16 * nb);
      }
    }
  }
}

// dW[n, k] = sum_rows G[row, n] X[row, k].  blockIdx.y = 64-column slab of G; a wave strides the
// 16-row tiles, keeps its [64 x K] partial in MFMA accumulators (v_mfma_f32_16x16x4_f32: the
// contraction index is the tile's rows, four at a time; f32 in, f32 accumulate) and writes it
// once; a second kernel sums the per-wave partials in a fixed order (deterministic).
// PRE: x is normalised on the way in exactly like skinny_linear_kernel<.., PRE> does (the weight
// gradient of a Linear behind an on-the-fly pre-norm needs the NORMALISED rows).
// PR (round 6): 0 = f32 pipe (rows 4 st + g of the tile per v_mfma_f32_16x16x4_f32); 1 / 3 = the bf16
// pipe with the tile's 16 rows as ONE contraction step (v_mfma_f32_16x16x16_bf16: lane (g, c) holds
// rows 4 g .. 4 g + 3 of column c of both operands), operands rounded (1) or split hi + lo with the
// three products lo*hi + hi*lo + hi*hi (3).  The bias gradient sums the unrounded values.
template <int K4, bool PRE = false, int PR = 0>
__global__ __launch_bounds__(WAVES * 64, 2) void skinny_dw_kernel(
    const float* __restrict__ gy, const float* __restrict__ x, int64_t rows, int N, int KF,
    float* __restrict__ partial, const float* __restrict__ pam = nullptr,
    const float* __restrict__ psc = nullptr, const float* __restrict__ pbs = nullptr,
    const int64_t* __restrict__ batch = nullptr, int PB = 1) {
  // KF = the layer's full input width: blockIdx.z = K-column slab of x (KF = 128 runs as two
  // 64-column slabs, each re-reading its G slab - G is the small operand at these widths)
  constexpr int K = 4 * K4, KB = K / 16, LDG = SLAB + 4, LDX = K + 4;
  constexpr int VG = TR * SLAB / 4 / 64, VX = TR * K / 4 / 64;   // float4 per lane per tile
  static_assert(K % 16 == 0 && TR * K % 256 == 0, "K is a multiple of 16");
  __shared__ __attribute__((aligned(16))) float g_lds[WAVES][TR * LDG];
  __shared__ __attribute__((aligned(16))) float x_lds[WAVES][TR * LDX];
  __shared__ __attribute__((aligned(16))) float ptab[PRE ? 2 * PRE_MAX + 4 * K4 : 4];   // am | sc | bias (this k-slab)
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int n0 = blockIdx.y * SLAB;
  const int k0 = blockIdx.z * K;
  float* gl = g_lds[wid];
  float* xl = x_lds[wid];
  f32x4 C[4][KB];                                       // C[nb][kb][r] = dW[n0 + 16 nb + 4 g + r][16 kb + c]
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};                 // column sums of G (the bias gradient)
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) C[nb][kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int64_t ntiles = (rows + TR - 1) / TR;
  const int64_t wave = (int64_t)blockIdx.x * WAVES + wid;
  const int64_t nwaves = (int64_t)gridDim.x * WAVES;
  if constexpr (PRE) {                                  // columns k0 .. k0 + K of every graph's row
    for (int i = threadIdx.x; i < PB * K; i += WAVES * 64) {
      const int gph = i / K, k = i - gph * K;
      ptab[i] = pam[gph * KF + k0 + k];
      ptab[PRE_MAX + i] = psc[gph * KF + k0 + k];
    }
    for (int i = threadIdx.x; i < K; i += WAVES * 64) ptab[2 * PRE_MAX + i] = pbs[k0 + i];
    __syncthreads();
  }
  float4 ng[VG], nx[VX];
  int xg[VX];                                           // graph of the row each x chunk belongs to
  auto fetch = [&](int64_t t) {                         // rows past the end read as zero
#pragma unroll
    for (int v = 0; v < VG; ++v) {
      const int q = v * 64 + lane, rr = q / (SLAB / 4), ch = q - rr * (SLAB / 4);
      const int64_t row = t * TR + rr;
      ng[v] = (row < rows) ? *reinterpret_cast<const float4*>(gy + row * N + n0 + ch * 4)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int v = 0; v < VX; ++v) {
      const int q = v * 64 + lane, rr = q / K4, ch = q - rr * K4;
      const int64_t row = t * TR + rr;
      nx[v] = (row < rows && k0 + ch * 4 < KF)            // (columns past KF: the last, narrower slab)
                  ? *reinterpret_cast<const float4*>(x + row * KF + k0 + ch * 4)
                  : make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (PRE) xg[v] = (row < rows) ? (batch ? (int)batch[row] : 0) : -1;
    }
  };
  if (wave < ntiles) fetch(wave);
  for (int64_t t = wave; t < ntiles; t += nwaves) {
    wave_sync_lds();
#pragma unroll
    for (int v = 0; v < VG; ++v) {
      const int q = v * 64 + lane, rr = q / (SLAB / 4), ch = q - rr * (SLAB / 4);
      *reinterpret_cast<float4*>(gl + rr * LDG + ch * 4) = ng[v];
    }
#pragma unroll
    for (int v = 0; v < VX; ++v) {
      const int q = v * 64 + lane, rr = q / K4, ch = q - rr * K4;
      float4 w = nx[v];
      if constexpr (PRE) {
        if (xg[v] >= 0) {                               // rows past the end stay zero
          const float4 a = *reinterpret_cast<const float4*>(ptab + xg[v] * K + ch * 4);
          const float4 sc4 = *reinterpret_cast<const float4*>(ptab + PRE_MAX + xg[v] * K + ch * 4);
          const float4 b4 = *reinterpret_cast<const float4*>(ptab + 2 * PRE_MAX + ch * 4);
          w.x = fmaf(w.x - a.x, sc4.x, b4.x); w.y = fmaf(w.y - a.y, sc4.y, b4.y);
          w.z = fmaf(w.z - a.z, sc4.z, b4.z); w.w = fmaf(w.w - a.w, sc4.w, b4.w);
        }
      }
      *reinterpret_cast<float4*>(xl + rr * LDX + ch * 4) = w;
    }
    wave_sync_lds();
    if (t + nwaves < ntiles) fetch(t + nwaves);         // in flight during the MFMAs
    if constexpr (PR == 0) {
#pragma unroll
      for (int st = 0; st < TR / 4; ++st) {             // rows 4 st + g of the tile
        float A[4], B[KB];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) A[nb] = gl[(4 * st + g) * LDG + 16 * nb + c];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) B[kb] = xl[(4 * st + g) * LDX + 16 * kb + c];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          bsum[nb] += A[nb];
#pragma unroll
          for (int kb = 0; kb < KB; ++kb)
            C[nb][kb] = SPT_SKINNY_MFMA(A[nb], B[kb], C[nb][kb]);
        }
      }
    } else {
      s16x4 Ah[4], Al[4];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        float a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = gl[(4 * g + j) * LDG + 16 * nb + c];
        bsum[nb] += (a[0] + a[1]) + (a[2] + a[3]);
        bf16x4 h, l;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          h[j] = (__bf16)a[j];
          l[j] = (__bf16)(a[j] - (float)h[j]);
        }
        Ah[nb] = __builtin_bit_cast(s16x4, h);
        Al[nb] = __builtin_bit_cast(s16x4, l);
      }
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        bf16x4 h, l;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float b = xl[(4 * g + j) * LDX + 16 * kb + c];
          h[j] = (__bf16)b;
          l[j] = (__bf16)(b - (float)h[j]);
        }
        const s16x4 Bh = __builtin_bit_cast(s16x4, h), Bl = __builtin_bit_cast(s16x4, l);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          if constexpr (PR == 3) {
            C[nb][kb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Al[nb], Bh, C[nb][kb], 0, 0, 0);
            C[nb][kb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Ah[nb], Bl, C[nb][kb], 0, 0, 0);
          }
          C[nb][kb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Ah[nb], Bh, C[nb][kb], 0, 0, 0);
        }
      }
    }
  }
  // partial[wave][N x K | N]: this block fills rows n0 .. n0 + 63 of its wave's record
  float* pw = partial + (size_t)wave * ((size_t)N * KF + N);
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    float v = bsum[nb];
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (g == 0 && k0 == 0) pw[(size_t)N * KF + n0 + 16 * nb + c] = v;
  }
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (k0 + 16 * kb + c < KF)
          pw[(size_t)(n0 + 16 * nb + 4 * g + r) * KF + k0 + 16 * kb + c] = C[nb][kb][r];
}

// sums of per-wave records [ntab][len], fixed order: 16 columns x 64 slices per 1024-thread block
// (a thread sums <= ntab / 64 records, 4 loads in flight), then the 64 slice sums in order.
// Columns below `split` go to out0, the rest to out1 (weight gradient | bias gradient).
__global__ __launch_bounds__(1024) void sum_tables_kernel(const float* __restrict__ partial, int ntab,
                                                          int len, int split, float* __restrict__ out0,
                                                          float* __restrict__ out1) {
  __shared__ float sl[64][17];
  const int cl = threadIdx.x & 15;
  const int col = blockIdx.x * 16 + cl;
  const int slice = threadIdx.x >> 4;
  float acc = 0.f;
  if (col < len) {
    const int per = (ntab + 63) / 64;
    const int lo = slice * per, hi = (lo + per < ntab) ? lo + per : ntab;
    int k = lo;
    for (; k + 16 <= hi; k += 16) {                 // sixteen records in flight, added in order
      float a[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = partial[(size_t)(k + j) * len + col];
#pragma unroll
      for (int j = 0; j < 16; ++j) acc += a[j];
    }
    for (; k + 4 <= hi; k += 4) {
      const float a0 = partial[(size_t)k * len + col], a1 = partial[(size_t)(k + 1) * len + col];
      const float a2 = partial[(size_t)(k + 2) * len + col], a3 = partial[(size_t)(k + 3) * len + col];
      acc += a0; acc += a1; acc += a2; acc += a3;
    }
    for (; k < hi; ++k) acc += partial[(size_t)k * len + col];
  }
  sl[slice][cl] = acc;
  __syncthreads();
  if (slice == 0 && col < len) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 64; ++k) t += sl[k][cl];
    if (col < split) out0[col] = t;
    else if (out1) out1[col - split] = t;
  }
}

// Backward of a NARROW Linear (N <= 16 outputs, K = 64 inputs: the classifier heads, src/nn/mlp.py:
// 128-142) in one pass over x and gy: lane = input column k.  Per row the N gradient values are
// wave-uniform (read out of a register tile with v_readlane), so dX[row, k] = sum_n g_n W[n, k]
// and dW[n, k] += g_n x[row, k] are 2 N fused multiply-adds per lane, db[n] += g_n.  The library
// ran these as three GEMMs / reductions of 0.1-0.7 ms each at 428 571 rows (a [13 x 64] result).
// Per-wave partials of (dW, db), summed in a fixed order by sum_tables_kernel.
template <int NMAX>
__global__ __launch_bounds__(256) void narrow_linear_bwd_kernel(
    const float* __restrict__ gy, const float* __restrict__ x, const float* __restrict__ W,
    int64_t rows, int N, float* __restrict__ gx, float* __restrict__ partial, int K) {
  // K = the layer's input width: blockIdx.y = its 64-column slab (K = 128, the KITTI-360 width's
  // heads: two slabs, each re-reading the 13 gradient columns); x, W, gx, the partial records are
  // addressed with the full row stride K
  constexpr int RT = 16;                                // rows per step
  const int lane = (threadIdx.x & 63) + 64 * blockIdx.y;   // this lane's input column
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  float w[NMAX], aw[NMAX], ab[NMAX];
#pragma unroll
  for (int n = 0; n < NMAX; ++n) {
    w[n] = (n < N) ? W[(size_t)n * K + lane] : 0.f;
    aw[n] = ab[n] = 0.f;
  }
  const int64_t nsteps = (rows + RT - 1) / RT;
  for (int64_t t = wave; t < nsteps; t += nwaves) {
    const int64_t row0 = t * RT;
    const int cnt = (int)((rows - row0) < RT ? (rows - row0) : RT);
    // the step's gradient values: RT * N <= 256 floats, element e in lane e & 63 of gt[e >> 6]
    float gt[4];
    const int wl = threadIdx.x & 63;                    // lane inside the wave (lane = input column)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = q * 64 + wl;
      gt[q] = (e < cnt * N) ? gy[row0 * N + e] : 0.f;
    }
    float xr[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) xr[r] = (r < cnt) ? x[(row0 + r) * K + lane] : 0.f;
    if (N == NMAX) {                                    // compile-time element positions
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        float dx = 0.f;
#pragma unroll
        for (int n = 0; n < NMAX; ++n) {
          const int e = r * NMAX + n;
          const float gv = __builtin_bit_cast(
              float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gt[e >> 6]), e & 63));
          dx = fmaf(gv, w[n], dx);
          aw[n] = fmaf(gv, xr[r], aw[n]);
          ab[n] += gv;
        }
        if (gx && r < cnt) gx[(row0 + r) * K + lane] = dx;
      }
    } else {
      for (int r = 0; r < cnt; ++r) {
        float dx = 0.f;
        for (int n = 0; n < N; ++n) {
          const int e = r * N + n;
          const float src = (e >> 6) == 0 ? gt[0] : ((e >> 6) == 1 ? gt[1] : ((e >> 6) == 2 ? gt[2] : gt[3]));
          const float gv = __shfl(src, e & 63, 64);
          dx = fmaf(gv, w[n], dx);
#pragma unroll
          for (int m = 0; m < NMAX; ++m) {
            aw[m] = (m == n) ? fmaf(gv, xr[r < RT ? r : 0], aw[m]) : aw[m];
            ab[m] = (m == n) ? ab[m] + gv : ab[m];
          }
        }
        if (gx) gx[(row0 + r) * K + lane] = dx;
      }
    }
  }
  float* pw = partial + (size_t)wave * ((size_t)N * K + N);   // [N x K | N]
#pragma unroll
  for (int n = 0; n < NMAX; ++n)
    if (n < N) pw[n * K + lane] = aw[n];
  if (lane < N) {                                       // (slab 0's lanes: the bias gradient once)
    float v = 0.f;
#pragma unroll
    for (int n = 0; n < NMAX; ++n) v = (lane == n) ? ab[n] : v;
    pw[N * K + lane] = v;
  }
}

}  // namespace skinny
}  // namespace spt

using namespace spt;
using namespace spt::skinny;

constexpr int DW_BLOCKS = 256;                          // x 4 waves: partial tables per slab

extern "C" int spt_skinny_dw_supported(int K, int N) {
  // K > 64 runs as 64-column slabs of x (the last one narrower: 132 = 64 + 64 + 4, 260 = 4 x 64 + 4)
  return (K == 32 || K == 64 || K == 128 || K == 132 || K == 260) && N >= SLAB && N % SLAB == 0 && N <= 1024;
}
extern "C" size_t spt_skinny_dw_workspace_bytes(int K, int N) {
  return (size_t)DW_BLOCKS * WAVES * N * (K + 1) * sizeof(float);
}
// matrix mode of the skinny Linears: 0 f32 pipe, 1 (default) split bf16 - forward f32-exact (six
// products), input / weight gradients three products -, 3 bf16.  Process-wide default (< 0: query);
// the _m entries take the mode per call (< 0 there: this default).
static std::atomic<int> g_skinny_mode{[] { const char* e = getenv("SPT_SKINNY_MODE"); return e ? atoi(e) : 1; }()};
extern "C" int spt_skinny_use_split_bf16(int mode) {
  const int prev = g_skinny_mode;
  if (mode >= 0) g_skinny_mode = (mode == 3) ? 3 : (mode != 0 ? 1 : 0);
  return prev;
}
static inline int skinny_mode_of(int mode) { return mode < 0 ? (int)g_skinny_mode : ((mode & 3) == 3 ? 3 : ((mode & 3) ? 1 : 0)); }
static inline int skinny_pr_fwd(int mode) { const int m = skinny_mode_of(mode); return m == 0 ? 0 : (m == 3 ? 1 : 6); }
static inline int skinny_pr_bwd(int mode) { const int m = skinny_mode_of(mode); return m == 0 ? 0 : (m == 3 ? 1 : 3); }

// gw[N, K] = gy[rows, N]^T x[rows, K]  (the weight gradient of y = x W^T + b); gb[N] = column
// sums of gy (the bias gradient), or null
extern "C" int spt_skinny_dw_f32(const float* gy, const float* x, int64_t rows, int N, int K,
                                 float* gw, float* gb, void* ws, size_t ws_bytes,
                                 spt_stream_t stream_) {
  return spt_skinny_dw_pre_f32(gy, x, rows, N, K, gw, gb, nullptr, nullptr, nullptr, nullptr, 1, ws,
                               ws_bytes, stream_);
}
// Same with x normalised on the fly: x_n = (x - pre_am[g]) * pre_scale[g] + pre_bias, g =
// batch[row] (tables [num_graphs, K] / [K]; batch NULL = one graph) - the weight gradient of a
// Linear fed by spt_skinny_linear_pre_f32.  pre_am NULL = plain x.
extern "C" int spt_skinny_pre_supported(int K, int N, int num_graphs) {
  return (K == 32 || K == 64 || K == 128) && N >= SLAB && N % SLAB == 0 && N <= 1024 &&
         num_graphs >= 1 && num_graphs * K <= PRE_MAX;
}
extern "C" int spt_skinny_dw_pre_f32(const float* gy, const float* x, int64_t rows, int N, int K,
                                     float* gw, float* gb, const float* pre_am,
                                     const float* pre_scale, const float* pre_bias,
                                     const int64_t* batch, int num_graphs, void* ws,
                                     size_t ws_bytes, spt_stream_t stream_) {
  return spt_skinny_dw_pre_m_f32(gy, x, rows, N, K, gw, gb, pre_am, pre_scale, pre_bias, batch,
                                 num_graphs, -1, ws, ws_bytes, stream_);
}
// The same with the matrix mode per call (spt_skinny_use_split_bf16's values; < 0: the default)
extern "C" int spt_skinny_dw_pre_m_f32(const float* gy, const float* x, int64_t rows, int N, int K,
                                       float* gw, float* gb, const float* pre_am,
                                       const float* pre_scale, const float* pre_bias,
                                       const int64_t* batch, int num_graphs, int mode, void* ws,
                                       size_t ws_bytes, spt_stream_t stream_) {
  const int pr = skinny_pr_bwd(mode);
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(rows >= 0, "bad shape");
  SPT_CHECK_ARG(spt_skinny_dw_supported(K, N), "(K, N) not built");
  SPT_CHECK_ARG(gw && (rows == 0 || (gy && x)), "null pointer");
  SPT_CHECK_ARG(ws && ws_bytes >= spt_skinny_dw_workspace_bytes(K, N), "workspace too small");
  const bool pre = pre_am != nullptr;
  SPT_CHECK_ARG(!pre || (pre_scale && pre_bias && spt_skinny_pre_supported(K, N, num_graphs)),
                "pre-normalisation: incomplete tables or num_graphs * K too large");
  const int slabs = N / SLAB;
  if (rows == 0) {
    hipMemsetAsync(gw, 0, (size_t)N * K * sizeof(float), stream);
    if (gb) hipMemsetAsync(gb, 0, (size_t)N * sizeof(float), stream);
    return 0;
  }
  const int64_t tiles = ceil_div(rows, TR);
  int64_t bx = ceil_div(tiles, WAVES);
  const int kslabs = K > 64 ? (K + 63) / 64 : 1;            // the last slab may be narrower (132, 260)
  const int64_t cap = DW_BLOCKS / (slabs * kslabs) > 1 ? DW_BLOCKS / (slabs * kslabs) : 1;
  if (bx > cap) bx = cap;
  const dim3 grid((unsigned)bx, (unsigned)slabs, (unsigned)kslabs);
  float* partial = (float*)ws;
#define SPT_DW(K4, PRE_, PR_)                                                                       \
  skinny_dw_kernel<K4, PRE_, PR_><<<grid, WAVES * 64, 0, stream>>>(gy, x, rows, N, K, partial, pre_am, \
                                                                   pre_scale, pre_bias, batch, num_graphs);
#define SPT_DWS(K4, PRE_)                                                                           \
  if (pr == 3) { SPT_DW(K4, PRE_, 3) } else if (pr == 1) { SPT_DW(K4, PRE_, 1) } else { SPT_DW(K4, PRE_, 0) }
  if (pre) {
    if (K == 32) { SPT_DWS(8, true) } else { SPT_DWS(16, true) }
  } else if (K == 32) {
    SPT_DWS(8, false)
  } else {
    SPT_DWS(16, false)
  }
#undef SPT_DWS
#undef SPT_DW
  sum_tables_kernel<<<(N * (K + 1) + 15) / 16, 1024, 0, stream>>>(partial, (int)bx * WAVES, N * (K + 1),
                                                                  N * K, gw, gb);
  SPT_CHECK_LAUNCH();
  return 0;
}

extern "C" int spt_skinny_linear_supported(int K, int N) {
  const bool kok = K == 32 || K == 64 || K == 128 || K == 192 || K == 132 || K == 260 || K == 256;
  // N: whole 64-column slabs, or (the dX of a 132 / 260-wide input) any width from 64 up - the
  // last slab's missing columns are read as zero and not stored -, or the narrow heads (<= 16)
  return kok && ((N >= SLAB && N <= 1024) || (N >= 1 && N <= 16 && K <= 128));
}

extern "C" int spt_skinny_linear_f32(const float* x, int64_t rows, int K, const float* W,
                                     const float* bias, int N, float* y, spt_stream_t stream_) {
  return spt_skinny_linear_pre_f32(x, rows, K, W, bias, N, y, nullptr, nullptr, nullptr, nullptr, 1,
                                   nullptr, stream_);
}

// y = norm(x) W^T + b (+ residual): the pre-norm of a transformer block folded into its qkv Linear
// (x_n = (x - pre_am[g]) * pre_scale[g] + pre_bias applied while the rows are read; tables from
// spt_graphnorm_stats_f32; pre_am NULL = plain x) and the block's residual folded into its out_proj
// (`residual` [rows, N] or NULL: y = (x W^T + b) + residual) - src/nn/transformer.py:231-234.
// K in {32, 64, 128}, N a multiple of 64 for either option.
static int skinny_linear_impl(const float* x, int64_t rows, int K, const float* W, const float* bias,
                              int N, float* y, const float* pre_am, const float* pre_scale,
                              const float* pre_bias, const int64_t* batch, int num_graphs,
                              const float* residual, int ldwt, int pr, spt_stream_t stream_);
extern "C" int spt_skinny_linear_pre_f32(const float* x, int64_t rows, int K, const float* W,
                                         const float* bias, int N, float* y, const float* pre_am,
                                         const float* pre_scale, const float* pre_bias,
                                         const int64_t* batch, int num_graphs,
                                         const float* residual, spt_stream_t stream_) {
  return skinny_linear_impl(x, rows, K, W, bias, N, y, pre_am, pre_scale, pre_bias, batch, num_graphs,
                            residual, 0, skinny_pr_fwd(-1), stream_);
}
// The same with the matrix mode per call (precision.py: the mode word of the op's forward)
extern "C" int spt_skinny_linear_pre_m_f32(const float* x, int64_t rows, int K, const float* W,
                                           const float* bias, int N, float* y, const float* pre_am,
                                           const float* pre_scale, const float* pre_bias,
                                           const int64_t* batch, int num_graphs,
                                           const float* residual, int mode, spt_stream_t stream_) {
  return skinny_linear_impl(x, rows, K, W, bias, N, y, pre_am, pre_scale, pre_bias, batch, num_graphs,
                            residual, 0, skinny_pr_fwd(mode), stream_);
}
// y = x Wt (no transpose: Wt [K, N] row-major, e.g. dX = G W with the layer's own weight
// [N_out = K, N_in = N]); same shapes as spt_skinny_linear_f32 with N % 4 == 0, N >= 64.
extern "C" int spt_skinny_linear_wt_m_f32(const float* x, int64_t rows, int K, const float* Wt, int N,
                                          float* y, int mode, spt_stream_t stream_) {
  SPT_CHECK_ARG(N % 4 == 0 && N >= SLAB, "transposed weight: N % 4 == 0 and N >= 64");
  SPT_CHECK_ARG(((uintptr_t)Wt) % 16 == 0, "Wt must be 16-byte aligned");
  return skinny_linear_impl(x, rows, K, Wt, nullptr, N, y, nullptr, nullptr, nullptr, nullptr, 1, nullptr,
                            N, skinny_pr_bwd(mode), stream_);
}
extern "C" int spt_skinny_linear_wt_f32(const float* x, int64_t rows, int K, const float* Wt, int N,
                                        float* y, spt_stream_t stream_) {
  return spt_skinny_linear_wt_m_f32(x, rows, K, Wt, N, y, -1, stream_);
}
static int skinny_linear_impl(const float* x, int64_t rows, int K, const float* W, const float* bias,
                              int N, float* y, const float* pre_am, const float* pre_scale,
                              const float* pre_bias, const int64_t* batch, int num_graphs,
                              const float* residual, int ldwt, int pr, spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(rows >= 0, "bad shape");
  SPT_CHECK_ARG(spt_skinny_linear_supported(K, N), "(K, N) not built");
  if (rows == 0) return 0;
  SPT_CHECK_ARG(x && W && y, "null pointer");
  const bool pre = pre_am != nullptr, resid = residual != nullptr;
  SPT_CHECK_ARG(ldwt == 0 || (!pre && !resid && N > 16), "transposed weight: the plain product only");
  SPT_CHECK_ARG(!pre || (pre_scale && pre_bias && spt_skinny_pre_supported(K, N, num_graphs)),
                "pre-normalisation: incomplete tables, unbuilt shape or num_graphs * K too large");
  SPT_CHECK_ARG(!resid || spt_skinny_pre_supported(K, N, 1), "residual epilogue: unbuilt shape");
  const bool narrow = N <= 16;
  const int slabs = narrow ? 1 : (N + SLAB - 1) / SLAB;
  const int64_t tiles = ceil_div(rows, TR);
  int64_t bx = ceil_div(tiles, WAVES);
  const int64_t cap = (int64_t)256 * 8 / slabs > 1 ? (int64_t)256 * 8 / slabs : 1;
  if (bx > cap) bx = cap;
  const dim3 grid((unsigned)bx, (unsigned)slabs);
  if (pre || resid) {
#define SPT_SKINNY_PR(K4, PR_)                                                                     \
  if (pre && resid)                                                                                \
    skinny_linear_kernel<K4, 4, true, true, PR_><<<grid, WAVES * 64, 0, stream>>>(                 \
        x, rows, W, bias, N, y, pre_am, pre_scale, pre_bias, batch, num_graphs, residual);         \
  else if (pre)                                                                                    \
    skinny_linear_kernel<K4, 4, true, false, PR_><<<grid, WAVES * 64, 0, stream>>>(                \
        x, rows, W, bias, N, y, pre_am, pre_scale, pre_bias, batch, num_graphs, nullptr);          \
  else                                                                                             \
    skinny_linear_kernel<K4, 4, false, true, PR_><<<grid, WAVES * 64, 0, stream>>>(                \
        x, rows, W, bias, N, y, nullptr, nullptr, nullptr, nullptr, 1, residual);
#define SPT_SKINNY_PRS(K4)                                                                         \
  if (pr == 6) { SPT_SKINNY_PR(K4, 6) } else if (pr == 3) { SPT_SKINNY_PR(K4, 3) }                 \
  else if (pr == 1) { SPT_SKINNY_PR(K4, 1) } else { SPT_SKINNY_PR(K4, 0) }
    switch (K) {
      case 32:  SPT_SKINNY_PRS(8) break;
      case 64:  SPT_SKINNY_PRS(16) break;
      default:  SPT_SKINNY_PR(32, 0) break;            // (K = 128: the f32 pipe in every mode)
    }
#undef SPT_SKINNY_PRS
#undef SPT_SKINNY_PR
    SPT_CHECK_LAUNCH();
    return 0;
  }
  if (narrow) {
    switch (K) {
      case 32:  skinny_linear_kernel<8, 1><<<grid, WAVES * 64, 0, stream>>>(x, rows, W, bias, N, y); break;
      case 64:  skinny_linear_kernel<16, 1><<<grid, WAVES * 64, 0, stream>>>(x, rows, W, bias, N, y); break;
      default:  skinny_linear_kernel<32, 1><<<grid, WAVES * 64, 0, stream>>>(x, rows, W, bias, N, y); break;
    }
    SPT_CHECK_LAUNCH();
    return 0;
  }
  switch (K) {
#define SPT_SKINNY_PL(K4, PR_)                                                                     \
  skinny_linear_kernel<K4, 4, false, false, PR_><<<grid, WAVES * 64, 0, stream>>>(                 \
      x, rows, W, bias, N, y, nullptr, nullptr, nullptr, nullptr, 1, nullptr, ldwt);
#define SPT_SKINNY_PLS(K4)                                                                         \
  if (pr == 6) { SPT_SKINNY_PL(K4, 6) } else if (pr == 3) { SPT_SKINNY_PL(K4, 3) }                 \
  else if (pr == 1) { SPT_SKINNY_PL(K4, 1) } else { SPT_SKINNY_PL(K4, 0) }
    case 32:  SPT_SKINNY_PLS(8) break;
    case 64:  SPT_SKINNY_PLS(16) break;
#undef SPT_SKINNY_PLS
#undef SPT_SKINNY_PL
    case 128: skinny_linear_kernel<32><<<grid, WAVES * 64, 0, stream>>>(x, rows, W, bias, N, y, nullptr, nullptr, nullptr, nullptr, 1, nullptr, ldwt); break;
    case 256: {                                          // dX of the 128-wide blocks' qkv Linear (256 -> 128):
      int64_t b4 = ceil_div(tiles, (int64_t)4);          // 4-wave workgroups, slab + tiles = 133 KB of LDS
      const int64_t cap4 = (int64_t)256 / slabs > 1 ? (int64_t)256 / slabs : 1;
      if (b4 > cap4) b4 = cap4;
      skinny_linear_wlds_kernel<64, 4><<<dim3((unsigned)b4, (unsigned)slabs), 4 * 64, 0, stream>>>(
          x, rows, W, bias, N, y, ldwt);
      break;
    }
    case 260: {                                          // 4-wave workgroups: slab + tiles = 135 KB of LDS
      int64_t b4 = ceil_div(tiles, (int64_t)4);
      const int64_t cap4 = (int64_t)256 / slabs > 1 ? (int64_t)256 / slabs : 1;
      if (b4 > cap4) b4 = cap4;
      skinny_linear_wlds_kernel<65, 4><<<dim3((unsigned)b4, (unsigned)slabs), 4 * 64, 0, stream>>>(
          x, rows, W, bias, N, y, ldwt);
      break;
    }
    default: {
      int64_t b8 = ceil_div(tiles, WAVES_L);
      const int64_t cap8 = (int64_t)256 * (K == 192 ? 2 : 1) / slabs > 1 ? (int64_t)256 * (K == 192 ? 2 : 1) / slabs : 1;
      if (b8 > cap8) b8 = cap8;
      if (K == 132)
        skinny_linear_wlds_kernel<33><<<dim3((unsigned)b8, (unsigned)slabs), WAVES_L * 64, 0, stream>>>(
            x, rows, W, bias, N, y, ldwt);
      else
      {
        const dim3 g8((unsigned)b8, (unsigned)slabs);
        if (pr >= 3)                                    // (a forward call's six products: three here)
          skinny_linear_wlds_kernel<48, WAVES_L, 3><<<g8, WAVES_L * 64, 0, stream>>>(x, rows, W, bias, N, y, ldwt);
        else if (pr == 1)
          skinny_linear_wlds_kernel<48, WAVES_L, 1><<<g8, WAVES_L * 64, 0, stream>>>(x, rows, W, bias, N, y, ldwt);
        else
          skinny_linear_wlds_kernel<48><<<g8, WAVES_L * 64, 0, stream>>>(x, rows, W, bias, N, y, ldwt);
      }
      break;
    }
  }
  SPT_CHECK_LAUNCH();
  return 0;
}

constexpr int NARROW_BLOCKS = 256;                      // x 4 waves of partial (dW, db)

extern "C" int spt_narrow_linear_bwd_supported(int K, int N) { return (K == 64 || K == 128) && N >= 1 && N <= 16; }
extern "C" size_t spt_narrow_linear_bwd_workspace_bytes(int K, int N) {
  return (size_t)NARROW_BLOCKS * 4 * (16 * K + 16) * sizeof(float);
}
// Backward of y = x W^T + b with N <= 16, K = 64 in one pass: gx[rows,K] = gy W (nullable),
// gw[N,K] = gy^T x, gb[N] = column sums of gy (nullable).
extern "C" int spt_narrow_linear_bwd_f32(const float* gy, const float* x, const float* W,
                                         int64_t rows, int N, int K, float* gx, float* gw,
                                         float* gb, void* ws, size_t ws_bytes,
                                         spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(rows >= 0, "bad shape");
  SPT_CHECK_ARG(spt_narrow_linear_bwd_supported(K, N), "(K, N) not built");
  SPT_CHECK_ARG(W && gw && (rows == 0 || (gy && x)), "null pointer");
  SPT_CHECK_ARG(ws && ws_bytes >= spt_narrow_linear_bwd_workspace_bytes(K, N), "workspace too small");
  if (rows == 0) {
    hipMemsetAsync(gw, 0, (size_t)N * K * sizeof(float), stream);
    if (gb) hipMemsetAsync(gb, 0, (size_t)N * sizeof(float), stream);
    return 0;
  }
  int64_t bx = ceil_div(ceil_div(rows, (int64_t)16), (int64_t)4);
  if (bx > NARROW_BLOCKS) bx = NARROW_BLOCKS;
  float* partial = (float*)ws;
  int nmax;
  if (N == 13) {
    nmax = 13;
    narrow_linear_bwd_kernel<13><<<dim3((unsigned)bx, K / 64), 256, 0, stream>>>(gy, x, W, rows, N, gx, partial, K);
  } else {
    nmax = 16;
    narrow_linear_bwd_kernel<16><<<dim3((unsigned)bx, K / 64), 256, 0, stream>>>(gy, x, W, rows, N, gx, partial, K);
  }
  (void)nmax;
  sum_tables_kernel<<<(N * (K + 1) + 15) / 16, 1024, 0, stream>>>(partial, (int)bx * 4, N * (K + 1), N * K,
                                                                  gw, gb);
  SPT_CHECK_LAUNCH();
  return 0;
}
