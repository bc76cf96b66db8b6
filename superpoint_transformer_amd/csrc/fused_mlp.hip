// Fused layers of the point / node / edge MLPs (src/nn/mlp.py:8-94: bias-free
// nn.Linear -> GraphNorm -> LeakyReLU stacks; GraphNorm = PyG 2.3.0).
//
// Unfused, one layer over [rows, C] (rows up to 15 M) costs
//   fwd : GEMM (w h) | GraphNorm stats (r h) | GraphNorm apply (r h, w y) | next GEMM (r y)
//   bwd : GN stats (r h, r gy) | GN apply (r h, r gy, w gh) | dX GEMM (r gh, w gx) | dW GEMM (r gh, r y_prev)
// Here a layer is ONE kernel per direction:
//   fwd : y_prev = leaky(gn(h_prev)) is formed while the A tile is staged (h_prev
//         is read raw), the product runs as f32 MFMA 16x16x4 with the weight block in
//         B-operand registers, and the statistics of THIS layer's GraphNorm (column
//         sums / sums of squares, f64) come out of the epilogue;
//   bwd : gh = GraphNorm-backward(gy, h) and y_prev are formed while the tiles are
//         staged; gx = gh W and gW += gh^T y_prev are two MFMA GEMMs over the tile; the
//         statistics the PREVIOUS layer's GraphNorm backward needs (sum g', sum g' o')
//         come out of the gx epilogue.
// One launch covers the rows of one graph (statistics are per graph; a batch of B
// clouds = B launches over contiguous row ranges, described by a host `gptr`).
#include <math.h>
#include <stdlib.h>

#include "common.hpp"

namespace spt {
namespace fmlp {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int WAVES = 4;
constexpr int TR = 16;          // rows per MFMA tile
constexpr int MAX_BLOCKS = 1024;
constexpr int MAX_BWD_WAVES = 4096;   // per-wave partial tables of the backward

__device__ __forceinline__ double xg_sum_d(double v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- activations stored as bf16 (the `bf16` precision mode's storage option) -------------------
// four consecutive elements of a row as f32: a 16-byte load of f32 data, or an 8-byte load of
// bf16 data widened (bf16 -> f32 is a 16-bit shift: exact)
template <bool B16>
__device__ __forceinline__ float4 ld4(const float* __restrict__ base, int64_t elem) {
  if constexpr (B16) {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + elem);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                       __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
  } else {
    return *reinterpret_cast<const float4*>(base + elem);
  }
}
__device__ __forceinline__ unsigned bf16_bits(float v) {       // round to nearest even
  return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v);
}

// Stage a [cnt x K] tile of row-major x into LDS (row stride LD) with the widest
// aligned vector loads, optionally applying v <- leaky((v - am[k]) * sc[k] + bs[k])
// with the per-column tables read from LDS (tab = am | sc | bs, KT floats each).
// Rows >= cnt and columns >= K become 0.  RAW != nullptr also keeps the raw values.
// IND: the tile's rows are x[rid] with rid held by lane rr of `rid_l` (a gathered tile).
// X16: x holds bf16 values (whole-chunk rows only: K == KP).
template <int KP, int LD, int KT, bool IND = false, int GRP = 4, bool X16 = false>
__device__ __forceinline__ void stage_tile(const float* __restrict__ x, int64_t row0, int cnt,
                                           int K, bool pre, const float* tab, float slope,
                                           float* lds, float* raw, int lane, int rid_l = 0) {
  if (K == KP || X16) {
    // whole rows of 16-byte chunks, trip count known: the loads of up to four chunks per lane
    // are issued together and consumed afterwards - ONE memory round trip per group instead of
    // one per chunk (in-kernel cycle counts: this loop, load -> use -> ds_write per chunk, was
    // 58 % of the 64 -> 128 forward kernel and 18 % of its backward)
    constexpr int CH = KP / 4, NIT = (TR * CH + 63) / 64;
#pragma unroll 1
    for (int i0 = 0; i0 < NIT; i0 += GRP) {
      float4 v[GRP];
#pragma unroll
      for (int j = 0; j < GRP; ++j) {
        const int q = lane + 64 * (i0 + j), rr = q / CH, k = (q - rr * CH) << 2;
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i0 + j < NIT) {
          const int64_t xr = IND ? (int64_t)__shfl(rid_l, rr < TR ? rr : 0, 64) : row0 + rr;
          if (q < TR * CH && rr < cnt) v[j] = ld4<X16>(x, xr * K + k);
        }
      }
#pragma unroll
      for (int j = 0; j < GRP; ++j) {
        const int q = lane + 64 * (i0 + j), rr = q / CH, k = (q - rr * CH) << 2;
        if (i0 + j < NIT && q < TR * CH) {
          float4 w = v[j];
          if (raw) *reinterpret_cast<float4*>(raw + rr * LD + k) = w;
          if (pre && rr < cnt) {
            const float4 a = *reinterpret_cast<const float4*>(tab + k);
            const float4 s = *reinterpret_cast<const float4*>(tab + KT + k);
            const float4 b = *reinterpret_cast<const float4*>(tab + 2 * KT + k);
            w.x = fmaf(w.x - a.x, s.x, b.x); w.y = fmaf(w.y - a.y, s.y, b.y);
            w.z = fmaf(w.z - a.z, s.z, b.z); w.w = fmaf(w.w - a.w, s.w, b.w);
            w.x = w.x > 0.f ? w.x : w.x * slope; w.y = w.y > 0.f ? w.y : w.y * slope;
            w.z = w.z > 0.f ? w.z : w.z * slope; w.w = w.w > 0.f ? w.w : w.w * slope;
          }
          *reinterpret_cast<float4*>(lds + rr * LD + k) = w;
        }
      }
    }
  } else if ((K & 3) == 0) {
    const int CH = K >> 2;
    for (int q = lane; q < TR * CH; q += 64) {
      const int rr = q / CH, k = (q - rr * CH) << 2;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const int64_t xr = IND ? (int64_t)__shfl(rid_l, rr, 64) : row0 + rr;
      if (rr < cnt) v = *reinterpret_cast<const float4*>(x + xr * K + k);
      if (raw) *reinterpret_cast<float4*>(raw + rr * LD + k) = v;
      if (pre && rr < cnt) {
        const float4 a = *reinterpret_cast<const float4*>(tab + k);
        const float4 s = *reinterpret_cast<const float4*>(tab + KT + k);
        const float4 b = *reinterpret_cast<const float4*>(tab + 2 * KT + k);
        v.x = fmaf(v.x - a.x, s.x, b.x); v.y = fmaf(v.y - a.y, s.y, b.y);
        v.z = fmaf(v.z - a.z, s.z, b.z); v.w = fmaf(v.w - a.w, s.w, b.w);
        v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
        v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
      }
      *reinterpret_cast<float4*>(lds + rr * LD + k) = v;
    }
    // zero the padding columns [K, KP) once per tile (K % 4 == 0 => KP == K: nothing)
  } else {
    // rows that are not whole 16-byte chunks (K = 18: the raw edge features): element loads,
    // issued for the whole tile before the first one is used (they were load -> use -> ds_write
    // one at a time: five dependent round trips per tile)
    if constexpr (KP > 32) {                            // wide rows never come here unaligned
      for (int q = lane; q < TR * KP; q += 64) {
        const int rr = q / KP, k = q - rr * KP;
        float v = 0.f;
        const int64_t xr = IND ? (int64_t)__shfl(rid_l, rr, 64) : row0 + rr;
        if (rr < cnt && k < K) v = x[xr * K + k];
        if (raw) raw[rr * LD + k] = v;
        if (pre && rr < cnt && k < K) {
          v = fmaf(v - tab[k], tab[KT + k], tab[2 * KT + k]);
          v = (v > 0.f) ? v : v * slope;
        }
        lds[rr * LD + k] = v;
      }
      return;
    }
    constexpr int NE = KP > 32 ? 1 : (TR * KP + 63) / 64;
    float v[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int q = lane + 64 * i, rr = (q / KP) < TR ? q / KP : TR - 1, k = q - (q / KP) * KP;
      const int64_t xr = IND ? (int64_t)__shfl(rid_l, rr, 64) : row0 + rr;
      // clamped in-range address for the masked-out elements: an unconditional load keeps the
      // loads back to back (a load under a per-lane condition is followed by a wait + select)
      const bool ok = q < TR * KP && rr < cnt && k < K;
      const float t = x[ok ? xr * K + k : 0];
      v[i] = ok ? t : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int q = lane + 64 * i;
      if (q < TR * KP) {
        const int rr = q / KP, k = q - rr * KP;
        float w = v[i];
        if (raw) raw[rr * LD + k] = w;
        if (pre && rr < cnt && k < K) {
          w = fmaf(w - tab[k], tab[KT + k], tab[2 * KT + k]);
          w = (w > 0.f) ? w : w * slope;
        }
        lds[rr * LD + k] = w;
      }
    }
  }
}

// copy n floats (or zeros when src is null) into LDS, whole workgroup
__device__ __forceinline__ void load_table(float* dst, const float* __restrict__ src, int n,
                                           int cap) {
  for (int i = threadIdx.x; i < cap; i += WAVES * 64) dst[i] = (src && i < n) ? src[i] : 0.f;
}

// -DSPT_FMLP_PROFILE: per-section cycle counts (s_memtime) of wave 5 of the 64 -> 128 kernels,
// printed at its end - a measurement build only (gpurun_variants/), never shipped
#ifdef SPT_FMLP_PROFILE
#define FM_PROBE(i)                                               \
  {                                                               \
    const uint64_t now_ = __builtin_amdgcn_s_memtime();           \
    prof[i] += now_ - tlast;                                      \
    tlast = now_;                                                 \
  }
#define FM_PROBE_INIT uint64_t prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}; uint64_t tlast = __builtin_amdgcn_s_memtime();
#else
#define FM_PROBE(i)
#define FM_PROBE_INIT
#endif

// Multi-run launches (FmlpRuns, common.hpp): blockIdx.y = run; the kernel's range, its per-graph
// table rows and its slice of the partial tables are re-pointed here, the body is unchanged.
#define SPT_FMLP_RUN_FWD(N_)                                                  \
  if (rt.n > 0) {                                                             \
    const int run_ = blockIdx.y, gph_ = rt.g[run_];                           \
    r0 = rt.r0[run_];                                                         \
    r1 = rt.r1[run_];                                                         \
    if (am) { am += (size_t)gph_ * K; sc += (size_t)gph_ * K; }               \
    partial += (size_t)run_ * gridDim.x * (2 * (N_) + 1);                     \
  }
#define SPT_FMLP_RUN_BWD(N_, NW_) SPT_FMLP_RUN_BWD_T(N_, NW_, NW_)
/* TPB_: weight-gradient tables a workgroup writes (NW_ per-wave tables, or 1 when it sums them first) */
#define SPT_FMLP_RUN_BWD_T(N_, NW_, TPB_)                                     \
  if (rt.n > 0) {                                                             \
    const int run_ = blockIdx.y, gph_ = rt.g[run_];                           \
    r0 = rt.r0[run_];                                                         \
    r1 = rt.r1[run_];                                                         \
    am += (size_t)gph_ * (N_); sc += (size_t)gph_ * (N_);                     \
    c1 += (size_t)gph_ * (N_); c2 += (size_t)gph_ * (N_); c3 += (size_t)gph_ * (N_); \
    if (pam) { pam += (size_t)gph_ * K; psc += (size_t)gph_ * K; }            \
    gw_partial += (size_t)run_ * gridDim.x * (TPB_) * (N_) * K;               \
    if (pstat_partial) pstat_partial += (size_t)run_ * gridDim.x * (NW_) * (2 * K + 1); \
  }

// ---- forward -------------------------------------------------------------------
// K4 = ceil(K/4) k-steps, NBK = N/16 column blocks.
template <int K4, int NBK>
__global__ __launch_bounds__(WAVES * 64, 2) void fwd_kernel(
    const float* __restrict__ x, int64_t r0, int64_t r1, int K, const float* __restrict__ W,
    const float* __restrict__ am, const float* __restrict__ sc, const float* __restrict__ bs,
    float slope, float* __restrict__ h, double* __restrict__ partial, FmlpRuns rt) {
  constexpr int KP = K4 * 4, LDA = KP + 4, N = NBK * 16;
  SPT_FMLP_RUN_FWD(N)
  __shared__ __attribute__((aligned(16))) float a_lds[WAVES][TR * LDA];
  __shared__ __attribute__((aligned(16))) float tab[3 * KP];   // am | sc | bs of the previous norm
  __shared__ double red[WAVES][2 * N];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  float* al = a_lds[wid];
  const bool pre = am != nullptr;
  load_table(tab, am, K, KP);
  load_table(tab + KP, sc, K, KP);
  load_table(tab + 2 * KP, bs, K, KP);
  // padding columns of the A tile stay zero for the whole kernel
  for (int i = lane; i < TR * LDA; i += 64) al[i] = 0.f;
  __syncthreads();

  float B[NBK][K4];   // lane (g, c): W[16 nb + c][4 st + g]
#pragma unroll
  for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
    for (int st = 0; st < K4; ++st) {
      const int k = 4 * st + g;
      B[nb][st] = (k < K) ? W[(size_t)(16 * nb + c) * K + k] : 0.f;
    }
  double s1[NBK], s2[NBK];
#pragma unroll
  for (int nb = 0; nb < NBK; ++nb) s1[nb] = s2[nb] = 0.0;

  const int64_t ntiles = (r1 - r0 + TR - 1) / TR;
  const int64_t wave = (int64_t)blockIdx.x * WAVES + wid;
  const int64_t nwaves = (int64_t)gridDim.x * WAVES;
  FM_PROBE_INIT
  for (int64_t t = wave; t < ntiles; t += nwaves) {
    const int64_t row0 = r0 + t * TR;
    const int cnt = (int)((r1 - row0) < TR ? (r1 - row0) : TR);
    wave_sync_lds();
    FM_PROBE(3)
    stage_tile<KP, LDA, KP>(x, row0, cnt, K, pre, tab, slope, al, nullptr, lane);
    wave_sync_lds();
    FM_PROBE(0)
    float A[K4];
#pragma unroll
    for (int st = 0; st < K4; ++st) A[st] = al[c * LDA + 4 * st + g];
    f32x4 C[NBK];
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) C[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < K4; ++st)
#pragma unroll
      for (int nb = 0; nb < NBK; ++nb)
        C[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[st], B[nb][st], C[nb], 0, 0, 0);
    FM_PROBE(1)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = 4 * g + r;
      if (rr < cnt) {
        float* hr = h + (row0 + rr) * N + c;
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb) {
          const float v = C[nb][r];
          hr[16 * nb] = v;
#ifndef SPT_FMLP_NO_STATS   /* measurement builds only */
          s1[nb] += (double)v;
          s2[nb] += (double)v * (double)v;
#endif
        }
      }
    }
    FM_PROBE(2)
  }
#ifdef SPT_FMLP_PROFILE
  if (K4 == 16 && NBK == 8 && lane == 0 && wave == 5)
    printf("fmlp fwd<16,8> wave 5 cycles: stage %lu mfma %lu store+stats %lu looptop %lu\n",
           (unsigned long)prof[0], (unsigned long)prof[1], (unsigned long)prof[2],
           (unsigned long)prof[3]);
#endif
#pragma unroll
  for (int nb = 0; nb < NBK; ++nb) {
    const double a = xg_sum_d(s1[nb]), b = xg_sum_d(s2[nb]);
    if (g == 0) {
      red[wid][nb * 16 + c] = a;
      red[wid][N + nb * 16 + c] = b;
    }
  }
  __syncthreads();
  double* out = partial + (size_t)blockIdx.x * (2 * N + 1);
  for (int i = threadIdx.x; i < 2 * N; i += WAVES * 64)
    out[i] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
  if (threadIdx.x == 0) out[2 * N] = (blockIdx.x == 0) ? (double)(r1 - r0) : 0.0;  // row count once
}

// ---- backward ------------------------------------------------------------------
// gh[r, n] = c1[n] g - c2[n] o - c3[n],  g = gy * leaky'(y),  o = h - am,  y = o sc + bs
// gx = gh W  (optional), gW += gh^T y_prev, previous-layer statistics from gx.
// NW waves per workgroup share the LDS copies of the tables and (when the weight
// block is too big for B-operand registers next to the gW accumulators) of W itself:
// the 64 -> 128 layer then runs 2 waves per SIMD instead of 1.
template <int K4, int NBK, bool NEED_GX, int NW, bool W_IN_LDS>
__global__ __launch_bounds__(NW * 64, NW / 4) void bwd_kernel(
    const float* __restrict__ gy, const float* __restrict__ h, int64_t r0, int64_t r1,
    const float* __restrict__ am, const float* __restrict__ sc, const float* __restrict__ bs,
    float slope, const float* __restrict__ c1, const float* __restrict__ c2,
    const float* __restrict__ c3, const float* __restrict__ xprev, int K,
    const float* __restrict__ pam, const float* __restrict__ psc, const float* __restrict__ pbs,
    float pslope, const float* __restrict__ W, float* __restrict__ gx,
    float* __restrict__ gw_partial, double* __restrict__ pstat_partial, FmlpRuns rt) {
  constexpr int KP = K4 * 4, KB = (KP + 15) / 16, KPP = KB * 16, N = NBK * 16;
  constexpr int LDG = N + 4, LDX = KPP + 4;
  SPT_FMLP_RUN_BWD(N, NW)
  __shared__ __attribute__((aligned(16))) float g_lds[NW][TR * LDG];   // gh tile
  __shared__ __attribute__((aligned(16))) float x_lds[NW][TR * LDX];   // RAW h_prev tile
  // row stride KPP + 16: lane groups g = 0..3 read rows 4 st + g, whose 16-float column
  // windows then fall on alternating halves of the 32 LDS banks (2 lanes per bank = the
  // natural wave64 rate) instead of all four on the same 16 banks
  constexpr int LDW = KPP + 16;
  __shared__ __attribute__((aligned(16))) float w_lds[(NEED_GX && W_IN_LDS) ? N * LDW : 4];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  __shared__ __attribute__((aligned(16))) float gt[6 * N];      // am | sc | bs | c1 | c2 | c3
  __shared__ __attribute__((aligned(16))) float pt[3 * KPP];    // previous norm: am | sc | bs
  float* gl = g_lds[wid];
  float* xl = x_lds[wid];
  const bool pre = pam != nullptr;
  if constexpr (NEED_GX && W_IN_LDS) {
    for (int i = threadIdx.x; i < N * KPP; i += NW * 64) {
      const int n = i / KPP, k = i - n * KPP;
      w_lds[n * LDW + k] = (k < K) ? W[(size_t)n * K + k] : 0.f;
    }
  }
  load_table(gt, am, N, N);
  load_table(gt + N, sc, N, N);
  load_table(gt + 2 * N, bs, N, N);
  load_table(gt + 3 * N, c1, N, N);
  load_table(gt + 4 * N, c2, N, N);
  load_table(gt + 5 * N, c3, N, N);
  load_table(pt, pam, K, KPP);
  load_table(pt + KPP, psc, K, KPP);
  load_table(pt + 2 * KPP, pbs, K, KPP);
  for (int i = lane; i < TR * LDX; i += 64) xl[i] = 0.f;
  __syncthreads();

  // weight-gradient accumulators: C3[nb][kb][r] = gW[16 nb + 4 g + r][16 kb + c]
  f32x4 C3[NBK][KB];
#pragma unroll
  for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) C3[nb][kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // B operands of gx = gh W: lane (g, c) holds W[4 st + g][16 kb + c], st < N/4
  float BW[(NEED_GX && !W_IN_LDS) ? N / 4 : 1][(NEED_GX && !W_IN_LDS) ? KB : 1];
  if constexpr (NEED_GX && !W_IN_LDS) {
#pragma unroll
    for (int st = 0; st < N / 4; ++st)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int k = 16 * kb + c;
        BW[st][kb] = (k < K) ? W[(size_t)(4 * st + g) * K + k] : 0.f;
      }
  }
  double p1[KB], p2[KB];   // previous layer: sum g', sum g' o'   (columns 16 kb + c)
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) p1[kb] = p2[kb] = 0.0;

  const int64_t ntiles = (r1 - r0 + TR - 1) / TR;
  const int64_t wave = (int64_t)blockIdx.x * NW + wid;
  const int64_t nwaves = (int64_t)gridDim.x * NW;
  for (int64_t t = wave; t < ntiles; t += nwaves) {
    const int64_t row0 = r0 + t * TR;
    const int cnt = (int)((r1 - row0) < TR ? (r1 - row0) : TR);
    wave_sync_lds();
    // gh tile: each lane owns one 4-column group, 16-byte loads of gy and h
    {
      constexpr int CH = N / 4;
      for (int q = lane; q < TR * CH; q += 64) {
        const int rr = q / CH, n = (q - rr * CH) << 2;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rr < cnt) {
          const float4 hv = *reinterpret_cast<const float4*>(h + (row0 + rr) * N + n);
          const float4 gv = *reinterpret_cast<const float4*>(gy + (row0 + rr) * N + n);
          const float4 a = *reinterpret_cast<const float4*>(gt + n);
          const float4 sc4 = *reinterpret_cast<const float4*>(gt + N + n);
          const float4 b4 = *reinterpret_cast<const float4*>(gt + 2 * N + n);
          const float4 k1 = *reinterpret_cast<const float4*>(gt + 3 * N + n);
          const float4 k2 = *reinterpret_cast<const float4*>(gt + 4 * N + n);
          const float4 k3 = *reinterpret_cast<const float4*>(gt + 5 * N + n);
          const float hh[4] = {hv.x, hv.y, hv.z, hv.w}, gg4[4] = {gv.x, gv.y, gv.z, gv.w};
          const float aa[4] = {a.x, a.y, a.z, a.w}, ss[4] = {sc4.x, sc4.y, sc4.z, sc4.w};
          const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, q1[4] = {k1.x, k1.y, k1.z, k1.w};
          const float q2[4] = {k2.x, k2.y, k2.z, k2.w}, q3[4] = {k3.x, k3.y, k3.z, k3.w};
          float o4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float o = hh[e] - aa[e];
            float gg = gg4[e];
            if (slope != 1.f) {
              const float y = fmaf(o, ss[e], bb[e]);
              gg = (y > 0.f) ? gg : gg * slope;
            }
            o4[e] = fmaf(q1[e], gg, -fmaf(q2[e], o, q3[e]));
          }
          v = make_float4(o4[0], o4[1], o4[2], o4[3]);
        }
        *reinterpret_cast<float4*>(gl + rr * LDG + n) = v;
      }
    }
    // RAW h_prev tile; y_prev = leaky(gn(raw)) is formed where it is consumed
    stage_tile<KPP, LDX, KPP>(xprev, row0, cnt, K, false, pt, pslope, xl, nullptr, lane);
    wave_sync_lds();
    // ---- gW += gh^T y_prev : contraction index = row = 4 g + r  (k-slot = lane group)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float xb[KB];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int k = 16 * kb + c;
        float v = xl[(4 * g + r) * LDX + k];
        if (pre) {
          v = fmaf(v - pt[k], pt[KPP + k], pt[2 * KPP + k]);
          v = (v > 0.f) ? v : v * pslope;
          v = (4 * g + r < cnt && k < K) ? v : 0.f;   // padding rows / columns contribute nothing
        }
        xb[kb] = v;
      }
#pragma unroll
      for (int nb = 0; nb < NBK; ++nb) {
        const float ga = gl[(4 * g + r) * LDG + 16 * nb + c];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
          C3[nb][kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga, xb[kb], C3[nb][kb], 0, 0, 0);
      }
    }
    // ---- gx = gh W (+ statistics for the previous GraphNorm's backward) ------------
    if constexpr (NEED_GX) {
      f32x4 CX[KB];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) CX[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if constexpr (W_IN_LDS) {
        // operands of k-steps st+1, st+2 in flight under the MFMAs of st, st-1 (the
        // scheduler, left alone, waits for every LDS read right before its MFMA)
        constexpr int G = (KB > 4) ? 1 : 2;              // k-steps per group
        float ac[G], bc[G][KB], an[G], bn[G][KB];
#pragma unroll
        for (int u = 0; u < G; ++u) {
          ac[u] = gl[c * LDG + 4 * u + g];               // A[i = row c][k = 4 st + g]
#pragma unroll
          for (int kb = 0; kb < KB; ++kb) bc[u][kb] = w_lds[(4 * u + g) * LDW + 16 * kb + c];
        }
#pragma unroll
        for (int grp = 0; grp < N / 4 / G; ++grp) {
          if (grp + 1 < N / 4 / G) {
#pragma unroll
            for (int u = 0; u < G; ++u) {
              const int st = G * (grp + 1) + u;
              an[u] = gl[c * LDG + 4 * st + g];
#pragma unroll
              for (int kb = 0; kb < KB; ++kb) bn[u][kb] = w_lds[(4 * st + g) * LDW + 16 * kb + c];
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < G; ++u)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
              CX[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[u], bc[u][kb], CX[kb], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < G; ++u) {
            ac[u] = an[u];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) bc[u][kb] = bn[u][kb];
          }
        }
      } else {
#pragma unroll
        for (int st = 0; st < N / 4; ++st) {
          const float a = gl[c * LDG + 4 * st + g];     // A[i = row c][k = 4 st + g]
#pragma unroll
          for (int kb = 0; kb < KB; ++kb)
            CX[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, BW[st][kb], CX[kb], 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = 4 * g + r;
        if (rr < cnt) {
#pragma unroll
          for (int kb = 0; kb < KB; ++kb) {
            const int k = 16 * kb + c;
            if (k < K) {
              const float v = CX[kb][r];
              gx[(row0 + rr) * K + k] = v;
              if (pre) {
                const float o = xl[rr * LDX + k] - pt[k];
                float gg = v;
                if (pslope != 1.f) {
                  const float y = fmaf(o, pt[KPP + k], pt[2 * KPP + k]);
                  gg = (y > 0.f) ? gg : gg * pslope;
                }
                p1[kb] += (double)gg;
                p2[kb] += (double)gg * (double)o;
              }
            }
          }
        }
      }
    }
  }
  // per-wave partial weight gradients [N][K] and previous-layer statistics
  float* gwp = gw_partial + (size_t)wave * N * K;
#pragma unroll
  for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 16 * kb + c;
        if (k < K) gwp[(size_t)(16 * nb + 4 * g + r) * K + k] = C3[nb][kb][r];
      }
  if (pstat_partial) {
    double* pp = pstat_partial + (size_t)wave * (2 * K + 1);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const double a = xg_sum_d(p1[kb]), b = xg_sum_d(p2[kb]);
      const int k = 16 * kb + c;
      if (g == 0 && k < K) {
        pp[k] = a;
        pp[K + k] = b;
      }
    }
    if (lane == 0) pp[2 * K] = (wave == 0) ? (double)(r1 - r0) : 0.0;
  }
}

// =====================================================================================
// Split-bf16 variants.  x = hi + lo with hi = bf16(x), lo = bf16(x - hi)  (|x - hi - lo| <=
// 2^-18 |x|); an f32 product a*b is taken as a_hi b_hi + a_lo b_hi + a_hi b_lo on the bf16
// matrix pipe (16x the f32 pipe's rate; bf16 x bf16 is exact in f32, f32 accumulate), a
// relative error of <= ~3 * 2^-18 per product - the same class as a different summation
// order of an f32 GEMM (tests/test_fused_mlp_gpu.py bars unchanged).  The layers are
// HBM-bound once the matrix pipe time is gone; what remains per tile is the staging.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int NV, typename V>
__device__ __forceinline__ void split_bf16(const float (&x)[NV], V& hi, V& lo) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const __bf16 h = (__bf16)x[i];
    hi[i] = h;
    lo[i] = (__bf16)(x[i] - (float)h);
  }
}
// LO = false: plain bf16 operands (hi halves only): the bf16 precision mode
template <bool LO>
__device__ __forceinline__ f32x4 mfma3_32(const bf16x8& ah, const bf16x8& al, const bf16x8& bh,
                                          const bf16x8& bl, f32x4 c) {
  if constexpr (LO) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
  }
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
}
template <bool LO>
__device__ __forceinline__ f32x4 mfma3_16(const bf16x4& ah, const bf16x4& al, const bf16x4& bh,
                                          const bf16x4& bl, f32x4 c) {
  const s16x4 AH = __builtin_bit_cast(s16x4, ah), AL = __builtin_bit_cast(s16x4, al);
  const s16x4 BH = __builtin_bit_cast(s16x4, bh), BL = __builtin_bit_cast(s16x4, bl);
  if constexpr (LO) {
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(AL, BH, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(AH, BL, c, 0, 0, 0);
  }
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(AH, BH, c, 0, 0, 0);
}

// ---- forward: K padded to KS steps of 32, the weight block W[n][k] as split bf16 B operands
// (lane (g, c): W[16 nb + c][32 ks + 8 g .. + 7]) resident for the whole launch -----------
// IN16 / OUT16 (bf16 mode's storage option): x / h hold bf16 values.  The output tile then
// leaves through the wave's A-tile buffer: values rounded to bf16 (the statistics are taken from
// the ROUNDED values - what every consumer of h sees), lane pairs (c, c ^ 1) packed into 32-bit
// words, rows padded by 16 bytes (bank-conflict-free both ways), stored as whole 16-byte chunks of
// the tile's contiguous 16 x N region.
template <int K4, int NBK, bool LO, bool IN16 = false, bool OUT16 = false>
__global__ __launch_bounds__(WAVES * 64, 2) void fwd_kernel_bf(
    const float* __restrict__ x, int64_t r0, int64_t r1, int K, const float* __restrict__ W,
    const float* __restrict__ am, const float* __restrict__ sc, const float* __restrict__ bs,
    float slope, float* __restrict__ h, double* __restrict__ partial, FmlpRuns rt) {
  constexpr int KP = K4 * 4, KS = (KP + 31) / 32, KP32 = KS * 32, LDA = KP32 + 4, N = NBK * 16;
  SPT_FMLP_RUN_FWD(N)
  __shared__ __attribute__((aligned(16))) float a_lds[WAVES][TR * LDA];
  __shared__ __attribute__((aligned(16))) float tab[3 * KP32];
  __shared__ double red[WAVES][2 * N];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  float* al = a_lds[wid];
  const bool pre = am != nullptr;
  load_table(tab, am, K, KP32);
  load_table(tab + KP32, sc, K, KP32);
  load_table(tab + 2 * KP32, bs, K, KP32);
  for (int i = lane; i < TR * LDA; i += 64) al[i] = 0.f;   // padding columns stay zero
  __syncthreads();

  bf16x8 Bh[NBK][KS], Bl[NBK][KS];
#pragma unroll
  for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float w[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = 32 * ks + 8 * g + i;
        w[i] = (k < K) ? W[(size_t)(16 * nb + c) * K + k] : 0.f;
      }
      split_bf16<8>(w, Bh[nb][ks], Bl[nb][ks]);
    }
  double s1[NBK], s2[NBK];
#pragma unroll
  for (int nb = 0; nb < NBK; ++nb) s1[nb] = s2[nb] = 0.0;

  const int64_t ntiles = (r1 - r0 + TR - 1) / TR;
  const int64_t wave = (int64_t)blockIdx.x * WAVES + wid;
  const int64_t nwaves = (int64_t)gridDim.x * WAVES;
  for (int64_t t = wave; t < ntiles; t += nwaves) {
    const int64_t row0 = r0 + t * TR;
    const int cnt = (int)((r1 - row0) < TR ? (r1 - row0) : TR);
    wave_sync_lds();
    stage_tile<KP32, LDA, KP32, false, 4, IN16>(x, row0, cnt, K, pre, tab, slope, al, nullptr, lane);
    wave_sync_lds();
    f32x4 C[NBK];
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) C[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float4 a0 = *reinterpret_cast<const float4*>(al + c * LDA + 32 * ks + 8 * g);
      const float4 a1 = *reinterpret_cast<const float4*>(al + c * LDA + 32 * ks + 8 * g + 4);
      const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      bf16x8 ah, alo;
      split_bf16<8>(av, ah, alo);
#pragma unroll
      for (int nb = 0; nb < NBK; ++nb) C[nb] = mfma3_32<LO>(ah, alo, Bh[nb][ks], Bl[nb][ks], C[nb]);
    }
    if constexpr (OUT16) {
      constexpr int ROWB = 2 * N + 16;                    // bytes per row of the packed tile
      static_assert(TR * ROWB <= TR * LDA * 4, "the packed output tile fits the A-tile buffer");
      static_assert((TR * N / 8) % 64 == 0 || TR * N / 8 < 64, "whole 16-byte chunks");
      wave_sync_lds();                                    // every lane has read its A operands
      char* ob = reinterpret_cast<char*>(al);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = 4 * g + r;
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb) {
          const unsigned mine = bf16_bits(C[nb][r]);
          const float vr = __uint_as_float(mine << 16);
          if (rr < cnt) {
            s1[nb] += (double)vr;
            s2[nb] += (double)vr * (double)vr;
          }
          // neighbour lane c ^ 1 (DPP quad_perm [1, 0, 3, 2]); the lane whose parity matches the
          // row's packs the pair: even lanes rows 0 / 2, odd lanes rows 1 / 3
          const unsigned other = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0xB1, 0xF, 0xF, false);
          if ((c & 1) == (r & 1)) {
            const unsigned w = (c & 1) ? (other | (mine << 16)) : (mine | (other << 16));
            *reinterpret_cast<unsigned*>(ob + rr * ROWB + (16 * nb + (c & ~1)) * 2) = w;
          }
        }
      }
      wave_sync_lds();
      constexpr int CPR = N / 8;                          // 16-byte chunks per row
      uint16_t* h16 = reinterpret_cast<uint16_t*>(h);
#pragma unroll
      for (int q = lane; q < TR * CPR; q += 64) {
        const int rr = q / CPR, ch = q - rr * CPR;
        if (rr < cnt)
          *reinterpret_cast<uint4*>(h16 + (row0 + rr) * N + ch * 8) =
              *reinterpret_cast<const uint4*>(ob + rr * ROWB + ch * 16);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = 4 * g + r;
        if (rr < cnt) {
          float* hr = h + (row0 + rr) * N + c;
#pragma unroll
          for (int nb = 0; nb < NBK; ++nb) {
            const float v = C[nb][r];
            hr[16 * nb] = v;
            s1[nb] += (double)v;
            s2[nb] += (double)v * (double)v;
          }
        }
      }
    }
  }
#pragma unroll
  for (int nb = 0; nb < NBK; ++nb) {
    const double a = xg_sum_d(s1[nb]), b = xg_sum_d(s2[nb]);
    if (g == 0) {
      red[wid][nb * 16 + c] = a;
      red[wid][N + nb * 16 + c] = b;
    }
  }
  __syncthreads();
  double* out = partial + (size_t)blockIdx.x * (2 * N + 1);
  for (int i = threadIdx.x; i < 2 * N; i += WAVES * 64)
    out[i] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
  if (threadIdx.x == 0) out[2 * N] = (blockIdx.x == 0) ? (double)(r1 - r0) : 0.0;
}

// ---- forward, f32-EXACT on the bf16 matrix pipe: 3-way split, 6 products ------------------------
// An f32 value is exactly the sum of three bf16 values (8 + 8 + 8 mantissa bits):
//   x = x1 + x2 + x3,  w = w1 + w2 + w3,
//   x w = x1 w1 + (x1 w2 + x2 w1) + (x1 w3 + x3 w1 + x2 w2) + [x2 w3 + x3 w2 + x3 w3 <= 3 * 2^-24 |x w|]
// Every bf16 x bf16 product is exact in f32, so six v_mfma_f32_16x16x32_bf16 per (k-step, column
// block) give the f32 product to within one unit in the last place of the TERM (the dropped terms)
// plus the same f32 accumulation error class as an fmaf chain - the 2e-5 bar the 2-way split (3
// products, ~2^-16 per term) misses after three normalised layers holds.  Matrix-pipe time per
// 16-row tile at 64 -> 128: 96 instructions x 16 cycles against 128 x 32 on the f32 pipe.
// W lives in LDS as three bf16 planes (rows padded by 8 values: conflict-free 16-byte reads),
// shared by the 8 waves of the workgroup; smallest terms are accumulated first.
constexpr int WAVES_X3 = 8;
template <int NV, typename V>
__device__ __forceinline__ void split3_bf16(const float (&x)[NV], V& h, V& m, V& l) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const __bf16 a = (__bf16)x[i];
    const float r1 = x[i] - (float)a;              // exact
    const __bf16 b = (__bf16)r1;
    const float r2 = r1 - (float)b;                // exact, |r2| <= 2^-17 |x|: fits bf16 exactly
    h[i] = a;
    m[i] = b;
    l[i] = (__bf16)r2;
  }
}
template <int K4, int NBK>
__global__ __launch_bounds__(WAVES_X3 * 64, 2) void fwd_kernel_x3(
    const float* __restrict__ x, int64_t r0, int64_t r1, int K, const float* __restrict__ W,
    const float* __restrict__ am, const float* __restrict__ sc, const float* __restrict__ bs,
    float slope, float* __restrict__ h, double* __restrict__ partial, FmlpRuns rt) {
  constexpr int KP = K4 * 4, KS = (KP + 31) / 32, KP32 = KS * 32, LDA = KP32 + 4, N = NBK * 16;
  constexpr int LDW = KP32 + 8;                      // bf16 values per padded W row
  SPT_FMLP_RUN_FWD(N)
  __shared__ __attribute__((aligned(16))) float a_lds[WAVES_X3][TR * LDA];
  __shared__ __attribute__((aligned(16))) float tab[3 * KP32];
  __shared__ __attribute__((aligned(16))) __bf16 wpl[3][N * LDW];   // W = wpl[0] + wpl[1] + wpl[2]
  __shared__ double red[WAVES_X3][2 * N];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  float* al = a_lds[wid];
  const bool pre = am != nullptr;
  load_table(tab, am, K, KP32);
  load_table(tab + KP32, sc, K, KP32);
  load_table(tab + 2 * KP32, bs, K, KP32);
  for (int i = lane; i < TR * LDA; i += 64) al[i] = 0.f;   // padding columns stay zero
  for (int i = threadIdx.x; i < N * KP32; i += WAVES_X3 * 64) {
    const int n = i / KP32, k = i - n * KP32;
    const float w1[1] = {(k < K) ? W[(size_t)n * K + k] : 0.f};
    __bf16 a[1], b[1], cc[1];
    split3_bf16<1>(w1, a, b, cc);
    wpl[0][n * LDW + k] = a[0];
    wpl[1][n * LDW + k] = b[0];
    wpl[2][n * LDW + k] = cc[0];
  }
  __syncthreads();

  double s1[NBK], s2[NBK];
#pragma unroll
  for (int nb = 0; nb < NBK; ++nb) s1[nb] = s2[nb] = 0.0;

  const int64_t ntiles = (r1 - r0 + TR - 1) / TR;
  const int64_t wave = (int64_t)blockIdx.x * WAVES_X3 + wid;
  const int64_t nwaves = (int64_t)gridDim.x * WAVES_X3;
  // Round 6: rows of whole 16-byte chunks (K % 4 == 0) are PREFETCHED - the next tile's chunks are
  // requested into registers right after this tile has been staged and travel under its MFMAs and
  // stores (the staging loop was load -> transform -> ds_write per tile: at two waves per SIMD the
  // round trip showed, most where a launch has few tiles per wave).  Same values, same transform.
  constexpr int CHP = KP / 4, NITP = (TR * CHP + 63) / 64;
  const bool pref = (K & 3) == 0 && K == KP;
  float4 pv[NITP];
  auto prefetch = [&](int64_t t) {
    const int64_t row0 = r0 + t * TR;
    const int cnt = (t < ntiles) ? (int)((r1 - row0) < TR ? (r1 - row0) : TR) : 0;
#pragma unroll
    for (int j = 0; j < NITP; ++j) {
      const int q = lane + 64 * j, rr = q / CHP, k = (q - rr * CHP) << 2;
      // (clamped in-range address for the masked-out chunks: unconditional loads stay back to back)
      const bool ok = q < TR * CHP && rr < cnt;
      const float4 v = *reinterpret_cast<const float4*>(x + (ok ? (row0 + rr) * K + k : r0 * K));
      pv[j] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  if (pref && wave < ntiles) prefetch(wave);
  for (int64_t t = wave; t < ntiles; t += nwaves) {
    const int64_t row0 = r0 + t * TR;
    const int cnt = (int)((r1 - row0) < TR ? (r1 - row0) : TR);
    wave_sync_lds();
    if (pref) {
#pragma unroll
      for (int j = 0; j < NITP; ++j) {
        const int q = lane + 64 * j, rr = q / CHP, k = (q - rr * CHP) << 2;
        if (q < TR * CHP) {
          float4 w = pv[j];
          if (pre && rr < cnt) {
            const float4 a = *reinterpret_cast<const float4*>(tab + k);
            const float4 s4 = *reinterpret_cast<const float4*>(tab + KP32 + k);
            const float4 b = *reinterpret_cast<const float4*>(tab + 2 * KP32 + k);
            w.x = fmaf(w.x - a.x, s4.x, b.x); w.y = fmaf(w.y - a.y, s4.y, b.y);
            w.z = fmaf(w.z - a.z, s4.z, b.z); w.w = fmaf(w.w - a.w, s4.w, b.w);
            w.x = w.x > 0.f ? w.x : w.x * slope; w.y = w.y > 0.f ? w.y : w.y * slope;
            w.z = w.z > 0.f ? w.z : w.z * slope; w.w = w.w > 0.f ? w.w : w.w * slope;
          }
          *reinterpret_cast<float4*>(al + rr * LDA + k) = w;
        }
      }
    } else {
      stage_tile<KP32, LDA, KP32>(x, row0, cnt, K, pre, tab, slope, al, nullptr, lane);
    }
    wave_sync_lds();
    if (pref) prefetch(t + nwaves);
    f32x4 C[NBK];
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) C[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float4 a0 = *reinterpret_cast<const float4*>(al + c * LDA + 32 * ks + 8 * g);
      const float4 a1 = *reinterpret_cast<const float4*>(al + c * LDA + 32 * ks + 8 * g + 4);
      const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      bf16x8 x1, x2, x3;
      split3_bf16<8>(av, x1, x2, x3);
#pragma unroll
      for (int nb = 0; nb < NBK; ++nb) {
        const int wo = (16 * nb + c) * LDW + 32 * ks + 8 * g;
        const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(&wpl[0][wo]);
        const bf16x8 w2 = *reinterpret_cast<const bf16x8*>(&wpl[1][wo]);
        const bf16x8 w3 = *reinterpret_cast<const bf16x8*>(&wpl[2][wo]);
        f32x4 acc = C[nb];
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x3, w1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1, w3, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x2, w2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x2, w1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1, w2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1, w1, acc, 0, 0, 0);
        C[nb] = acc;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = 4 * g + r;
      if (rr < cnt) {
        float* hr = h + (row0 + rr) * N + c;
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb) {
          const float v = C[nb][r];
          hr[16 * nb] = v;
          s1[nb] += (double)v;
          s2[nb] += (double)v * (double)v;
        }
      }
    }
  }
#pragma unroll
  for (int nb = 0; nb < NBK; ++nb) {
    const double a = xg_sum_d(s1[nb]), b = xg_sum_d(s2[nb]);
    if (g == 0) {
      red[wid][nb * 16 + c] = a;
      red[wid][N + nb * 16 + c] = b;
    }
  }
  __syncthreads();
  double* out = partial + (size_t)blockIdx.x * (2 * N + 1);
  for (int i = threadIdx.x; i < 2 * N; i += WAVES_X3 * 64) {
    double t2 = 0.0;
#pragma unroll
    for (int w = 0; w < WAVES_X3; ++w) t2 += red[w][i];       // fixed order: deterministic
    out[i] = t2;
  }
  if (threadIdx.x == 0) out[2 * N] = (blockIdx.x == 0) ? (double)(r1 - r0) : 0.0;
}

// ---- backward: gW += gh^T y_prev as 16x16x16 products (contraction = the 16 rows of the
// tile: the 4 rows a lane group holds are one packed operand), gx = gh W as 16x16x32
// products with W^T as split bf16 rows in LDS (wt[k][n], n contiguous: lane (g, c) reads the 8
// values W[32 s + 8 g .. + 7][16 kb + c] with one 16-byte read each for hi and lo) ---------
// POOLED: gy is never materialised.  The layer's output went through a segment max-pool; its
// gradient is gy[i, c] = gout[s, c] if arg[s, c] == i else 0 with s the segment of row i.  The
// tiles then walk the rows in the CSR order of the pool (positions [r0, r1) of `perm`, segment
// of a position in `pos_seg`): h / x_prev rows are gathered, gx rows scattered (whole rows), and
// the (gout, arg) rows of a segment are read once per ~35 consecutive rows out of L1 / L2 -
// instead of the pool's backward writing a dense [rows, N] tensor that this kernel reads back.
// PIPE: the raw rows of the wave's NEXT tile (h, gy or gout/arg, x_prev) are requested into
// registers right after the current tile has been staged into LDS, so that they travel while
// the tile's GEMMs run; without it the 8-12 loads a lane issues per tile are consumed one
// round trip after the other (the staging loop is load -> transform -> ds_write) and the kernel
// is bound by memory LATENCY (937 k tiles x ~14 us / 2 048 waves = 6.4 ms at 64 -> 128), not by
// bandwidth.  Needs K % 4 == 0 (16-byte row chunks) and one wave per SIMD (the prefetch
// registers do not fit twice into 256).
// H16 / X16 (bf16 mode's storage option): h / xprev hold bf16 values (gy, gx stay f32).
// Round 6: the weight-gradient tables of a workgroup's waves are summed in LDS (wave 0, 1, ... in
// turn: a fixed order) and leave as ONE table per workgroup where the staging buffer of the x tiles
// holds N rows (every built shape but N = 128 on 4-wave workgroups).  The per-wave tables
// were 4 - 8 x the bytes: at the train batch a 35 000-row layer wrote 72 MB of partial tables for
// 18 MB of input and its post launch read them back (0.39 ms of a 6.2 ms step in 13 post launches).
__host__ __device__ constexpr bool fmlp_bf_wg_reduce(int K4, int NBK, int NW) {
  return NBK * 16 <= NW * TR && K4 > 0;          // N rows of the tiles' row stride fit the buffer
}
template <int K4, int NBK, bool NEED_GX, int NW, bool LO, bool POOLED = false, bool PIPE = false,
          bool H16 = false, bool X16 = false>
__global__ __launch_bounds__(NW * 64, NW / 4) void bwd_kernel_bf(
    const float* __restrict__ gy, const float* __restrict__ h, int64_t r0, int64_t r1,
    const float* __restrict__ am, const float* __restrict__ sc, const float* __restrict__ bs,
    float slope, const float* __restrict__ c1, const float* __restrict__ c2,
    const float* __restrict__ c3, const float* __restrict__ xprev, int K,
    const float* __restrict__ pam, const float* __restrict__ psc, const float* __restrict__ pbs,
    float pslope, const float* __restrict__ W, float* __restrict__ gx,
    float* __restrict__ gw_partial, double* __restrict__ pstat_partial, FmlpRuns rt,
    const int32_t* __restrict__ perm = nullptr, const int32_t* __restrict__ pos_seg = nullptr,
    const float* __restrict__ gout = nullptr, const int32_t* __restrict__ arg = nullptr) {
  constexpr int KP = K4 * 4, KB = (KP + 15) / 16, KPP = KB * 16, N = NBK * 16;
  constexpr int NS = (N + 31) / 32, NP32 = NS * 32;
  constexpr int LDG = NP32 + 4, LDX = KPP + 4, LDT = NP32 + 8;
  constexpr bool WGR = fmlp_bf_wg_reduce(K4, NBK, NW);
  SPT_FMLP_RUN_BWD_T(N, NW, (WGR ? 1 : NW))
  __shared__ __attribute__((aligned(16))) float g_lds[NW][TR * LDG];   // gh tile
  __shared__ __attribute__((aligned(16))) float x_lds[NW][TR * LDX];   // RAW h_prev tile
  __shared__ __attribute__((aligned(16))) __bf16 wt_hi[NEED_GX ? KPP * LDT : 8];
  __shared__ __attribute__((aligned(16))) __bf16 wt_lo[NEED_GX ? KPP * LDT : 8];
  __shared__ __attribute__((aligned(16))) float gt[6 * N];      // am | sc | bs | c1 | c2 | c3
  __shared__ __attribute__((aligned(16))) float pt[3 * KPP];    // previous norm: am | sc | bs
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  float* gl = g_lds[wid];
  float* xl = x_lds[wid];
  const bool pre = pam != nullptr;
  if constexpr (NEED_GX) {
    for (int i = threadIdx.x; i < KPP * NP32; i += NW * 64) {
      const int k = i / NP32, n = i - k * NP32;
      const float w = (k < K && n < N) ? W[(size_t)n * K + k] : 0.f;
      const __bf16 hh = (__bf16)w;
      wt_hi[k * LDT + n] = hh;
      wt_lo[k * LDT + n] = (__bf16)(w - (float)hh);
    }
  }
  load_table(gt, am, N, N);
  load_table(gt + N, sc, N, N);
  load_table(gt + 2 * N, bs, N, N);
  load_table(gt + 3 * N, c1, N, N);
  load_table(gt + 4 * N, c2, N, N);
  load_table(gt + 5 * N, c3, N, N);
  load_table(pt, pam, K, KPP);
  load_table(pt + KPP, psc, K, KPP);
  load_table(pt + 2 * KPP, pbs, K, KPP);
  for (int i = lane; i < TR * LDX; i += 64) xl[i] = 0.f;
  for (int i = lane; i < TR * LDG; i += 64) gl[i] = 0.f;     // columns [N, NP32) stay zero
  __syncthreads();

  f32x4 C3[NBK][KB];       // C3[nb][kb][r] = gW[16 nb + 4 g + r][16 kb + c]
#pragma unroll
  for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) C3[nb][kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  double p1[KB], p2[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) p1[kb] = p2[kb] = 0.0;

  const int64_t ntiles = (r1 - r0 + TR - 1) / TR;
  const int64_t wave = (int64_t)blockIdx.x * NW + wid;
  const int64_t nwaves = (int64_t)gridDim.x * NW;
  // gh[rr][n..n+3] from (h, g) of one 16-byte chunk; g = gy, or the pool's gradient routed to
  // the arg row
  auto gh_of = [&](const float4& hv, const float4& gv, int n) {
    const float4 a = *reinterpret_cast<const float4*>(gt + n);
    const float4 sc4 = *reinterpret_cast<const float4*>(gt + N + n);
    const float4 b4 = *reinterpret_cast<const float4*>(gt + 2 * N + n);
    const float4 k1 = *reinterpret_cast<const float4*>(gt + 3 * N + n);
    const float4 k2 = *reinterpret_cast<const float4*>(gt + 4 * N + n);
    const float4 k3 = *reinterpret_cast<const float4*>(gt + 5 * N + n);
    const float hh[4] = {hv.x, hv.y, hv.z, hv.w}, gg4[4] = {gv.x, gv.y, gv.z, gv.w};
    const float aa[4] = {a.x, a.y, a.z, a.w}, ss[4] = {sc4.x, sc4.y, sc4.z, sc4.w};
    const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, q1[4] = {k1.x, k1.y, k1.z, k1.w};
    const float q2[4] = {k2.x, k2.y, k2.z, k2.w}, q3[4] = {k3.x, k3.y, k3.z, k3.w};
    float o4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float o = hh[e] - aa[e];
      float gg = gg4[e];
      if (slope != 1.f) {
        const float y = fmaf(o, ss[e], bb[e]);
        gg = (y > 0.f) ? gg : gg * slope;
      }
      o4[e] = fmaf(q1[e], gg, -fmaf(q2[e], o, q3[e]));
    }
    return make_float4(o4[0], o4[1], o4[2], o4[3]);
  };
  static_assert(!(PIPE && (H16 || X16)), "the register-prefetch variant reads f32 rows");
  constexpr int CH = N / 4, CHX = KPP / 4;
  constexpr int HN = PIPE ? TR * CH / 64 : 1, XN = PIPE ? TR * CHX / 64 : 1;
  float4 p_h[HN], p_g[HN], p_x[XN];          // PIPE: raw chunks of the tile about to be staged
  int4 p_a[(PIPE && POOLED) ? HN : 1];
  int rid_n = 0, seg_n = 0;                   // POOLED: row id / segment of the NEXT tile's rows
  auto load_ids = [&](int64_t t) {
    rid_n = seg_n = 0;
    if constexpr (POOLED) {
      const int64_t rowf = r0 + t * TR;
      if (t < ntiles && rowf + lane < r1 && lane < TR) {
        rid_n = perm[rowf + lane];
        seg_n = pos_seg[rowf + lane];
      }
    }
  };
  auto load_raw = [&](int64_t t, int rid_l, int seg_l) {   // PIPE only
    const int64_t row0 = r0 + t * TR;
    const int cnt = (t < ntiles) ? (int)((r1 - row0) < TR ? (r1 - row0) : TR) : 0;
#pragma unroll
    for (int i = 0; i < HN; ++i) {
      const int q = lane + 64 * i, rr = q / CH, n = (q - rr * CH) << 2;
      const int rid = POOLED ? __shfl(rid_l, rr, 64) : 0;
      const int64_t sg = POOLED ? (int64_t)__shfl(seg_l, rr, 64) : 0;
      p_h[i] = p_g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rr < cnt) {
        const int64_t hr = POOLED ? (int64_t)rid : row0 + rr;
        p_h[i] = *reinterpret_cast<const float4*>(h + hr * N + n);
        if constexpr (POOLED) {
          p_a[i] = *reinterpret_cast<const int4*>(arg + sg * N + n);
          p_g[i] = *reinterpret_cast<const float4*>(gout + sg * N + n);
        } else {
          p_g[i] = *reinterpret_cast<const float4*>(gy + (row0 + rr) * N + n);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < XN; ++i) {
      const int q = lane + 64 * i, rr = q / CHX, k = (q - rr * CHX) << 2;
      const int64_t xr = POOLED ? (int64_t)__shfl(rid_l, rr, 64) : row0 + rr;
      p_x[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rr < cnt && k < K) p_x[i] = *reinterpret_cast<const float4*>(xprev + xr * K + k);
    }
  };
  load_ids(wave);
  if constexpr (PIPE) {
    load_raw(wave, rid_n, seg_n);
  }
  int rid_l = rid_n, seg_l = seg_n;
  load_ids(wave + nwaves);
  FM_PROBE_INIT
  for (int64_t t = wave; t < ntiles; t += nwaves) {
    const int64_t row0 = r0 + t * TR;
    const int cnt = (int)((r1 - row0) < TR ? (r1 - row0) : TR);
    wave_sync_lds();
    FM_PROBE(5)
    if constexpr (PIPE) {
      // stage the prefetched chunks, then request the next tile's while this one computes
#pragma unroll
      for (int i = 0; i < HN; ++i) {
        const int q = lane + 64 * i, rr = q / CH, n = (q - rr * CH) << 2;
        float4 gv = p_g[i];
        if constexpr (POOLED) {
          const int rid = __shfl(rid_l, rr, 64);
          gv = make_float4(p_a[i].x == rid ? gv.x : 0.f, p_a[i].y == rid ? gv.y : 0.f,
                           p_a[i].z == rid ? gv.z : 0.f, p_a[i].w == rid ? gv.w : 0.f);
        }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rr < cnt) v = gh_of(p_h[i], gv, n);
        *reinterpret_cast<float4*>(gl + rr * LDG + n) = v;
      }
#pragma unroll
      for (int i = 0; i < XN; ++i) {
        const int q = lane + 64 * i, rr = q / CHX, k = (q - rr * CHX) << 2;
        *reinterpret_cast<float4*>(xl + rr * LDX + k) = p_x[i];
      }
    } else {
      // chunks in groups of GS: all loads of a group first, then their transforms - one
      // memory round trip per group instead of one per chunk
      // (the pooled 64 -> 128 kernel has no registers to spare: 256 with the gW accumulators)
      constexpr int NITG = TR * CH / 64, GS = NITG < 2 ? 1 : 2;
      static_assert(TR * CH % 64 == 0, "whole waves of chunks");
#pragma unroll 1
      for (int i0 = 0; i0 < NITG; i0 += GS) {
        float4 hv[GS], gv[GS];
        int4 av[POOLED ? GS : 1];
        int ridv[GS];
#pragma unroll
        for (int j = 0; j < GS; ++j) {
          const int q = lane + 64 * (i0 + j), rr = q / CH, n = (q - rr * CH) << 2;
          ridv[j] = POOLED ? __shfl(rid_l, rr, 64) : 0;
          const int64_t sg = POOLED ? (int64_t)__shfl(seg_l, rr, 64) : 0;
          hv[j] = gv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
          if constexpr (POOLED) av[j] = make_int4(-1, -1, -1, -1);
          if (rr < cnt) {
            const int64_t hr = POOLED ? (int64_t)ridv[j] : row0 + rr;
            hv[j] = ld4<H16>(h, hr * N + n);
            if constexpr (POOLED) {
              av[j] = *reinterpret_cast<const int4*>(arg + sg * N + n);
              gv[j] = *reinterpret_cast<const float4*>(gout + sg * N + n);
            } else {
              gv[j] = *reinterpret_cast<const float4*>(gy + (row0 + rr) * N + n);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < GS; ++j) {
          const int q = lane + 64 * (i0 + j), rr = q / CH, n = (q - rr * CH) << 2;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (rr < cnt) {
            float4 g4 = gv[j];
            if constexpr (POOLED) {
              const int rid = ridv[j];
              g4 = make_float4(av[j].x == rid ? g4.x : 0.f, av[j].y == rid ? g4.y : 0.f,
                               av[j].z == rid ? g4.z : 0.f, av[j].w == rid ? g4.w : 0.f);
            }
            v = gh_of(hv[j], g4, n);
          }
          *reinterpret_cast<float4*>(gl + rr * LDG + n) = v;
        }
      }
      FM_PROBE(0)
      stage_tile<KPP, LDX, KPP, POOLED, (POOLED && NBK >= 8) ? 2 : 4, X16>(xprev, row0, cnt, K, false, pt,
                                                                             pslope, xl, nullptr, lane, rid_l);
      FM_PROBE(1)
    }
    const int rid_cur = rid_l;                 // the gx scatter below needs this tile's row ids
    if constexpr (PIPE) load_raw(t + nwaves, rid_n, seg_n);
    rid_l = rid_n;
    seg_l = seg_n;
    load_ids(t + 2 * nwaves);
    wave_sync_lds();
    FM_PROBE(2)
    // ---- gW += gh^T y_prev ------------------------------------------------------------
    {
      bf16x4 Xh[KB], Xl[KB];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int k = 16 * kb + c;
        float xv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = xl[(4 * g + r) * LDX + k];
          if (pre) {
            v = fmaf(v - pt[k], pt[KPP + k], pt[2 * KPP + k]);
            v = (v > 0.f) ? v : v * pslope;
            v = (4 * g + r < cnt && k < K) ? v : 0.f;
          }
          xv[r] = v;
        }
        split_bf16<4>(xv, Xh[kb], Xl[kb]);
      }
#pragma unroll
      for (int nb = 0; nb < NBK; ++nb) {
        float gv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) gv[r] = gl[(4 * g + r) * LDG + 16 * nb + c];
        bf16x4 gh4, gl4;
        split_bf16<4>(gv, gh4, gl4);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) C3[nb][kb] = mfma3_16<LO>(gh4, gl4, Xh[kb], Xl[kb], C3[nb][kb]);
      }
    }
    FM_PROBE(3)
    // ---- gx = gh W (+ statistics for the previous GraphNorm's backward) ------------------
    if constexpr (NEED_GX) {
      f32x4 CX[KB];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) CX[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // (round 6: the W^T fragments are requested LAB steps ahead of their MFMAs - one list of
      //  (32-column step, k block) pairs, rotating register sets; the compiler read each fragment
      //  right in front of its MFMAs and waited for it, which the K = 132 instance - one wave per
      //  SIMD - has nothing to hide behind.  Same products in the same order.)
      // (where the other instances sit at two or more waves per SIMD the fragment is read in front
      //  of its MFMAs as before: LAB = 0)
      constexpr int LAB = (NW == 4 && K4 > 16) ? 2 : 0, NSTG = NS * KB;
      bf16x8 bfh[LAB + 1], bfl[LAB + 1];
      auto ldb = [&](int step, bf16x8& bh, bf16x8& bl) {
        const int sg = step / KB, kb = step - sg * KB;
        bh = *reinterpret_cast<const bf16x8*>(wt_hi + (16 * kb + c) * LDT + 32 * sg + 8 * g);
        if constexpr (LO) bl = *reinterpret_cast<const bf16x8*>(wt_lo + (16 * kb + c) * LDT + 32 * sg + 8 * g);
        else bl = bh;
      };
#pragma unroll
      for (int i = 0; i < LAB && i < NSTG; ++i) ldb(i, bfh[i], bfl[i]);
      bf16x8 ah, alo;
#pragma unroll
      for (int step = 0; step < NSTG; ++step) {
        const int sg = step / KB, kb = step - sg * KB;
        if (kb == 0) {
          const float4 a0 = *reinterpret_cast<const float4*>(gl + c * LDG + 32 * sg + 8 * g);
          const float4 a1 = *reinterpret_cast<const float4*>(gl + c * LDG + 32 * sg + 8 * g + 4);
          const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
          split_bf16<8>(av, ah, alo);
        }
        if (step + LAB < NSTG) ldb(step + LAB, bfh[(step + LAB) % (LAB + 1)], bfl[(step + LAB) % (LAB + 1)]);
        if constexpr (LAB > 0) __builtin_amdgcn_sched_barrier(0);
        CX[kb] = mfma3_32<LO>(ah, alo, bfh[step % (LAB + 1)], bfl[step % (LAB + 1)], CX[kb]);
        if constexpr (LAB > 0) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = 4 * g + r;
        const int64_t orow = POOLED ? (int64_t)__shfl(rid_cur, rr, 64) : row0 + rr;
        if (rr < cnt) {
#pragma unroll
          for (int kb = 0; kb < KB; ++kb) {
            const int k = 16 * kb + c;
            if (k < K) {
              const float v = CX[kb][r];
              gx[orow * K + k] = v;
              if (pre) {
                const float o = xl[rr * LDX + k] - pt[k];
                float gg = v;
                if (pslope != 1.f) {
                  const float y = fmaf(o, pt[KPP + k], pt[2 * KPP + k]);
                  gg = (y > 0.f) ? gg : gg * pslope;
                }
                p1[kb] += (double)gg;
                p2[kb] += (double)gg * (double)o;
              }
            }
          }
        }
      }
    }
    FM_PROBE(4)
  }
#ifdef SPT_FMLP_PROFILE
  if (K4 == 16 && NBK == 8 && lane == 0 && wave == 5)
    printf("fmlp bwd<16,8,pooled=%d> wave 5 cycles: stage_gh %lu stage_x %lu ids+sync %lu dW %lu "
           "dX+store+stats %lu looptop %lu\n", (int)POOLED, (unsigned long)prof[0],
           (unsigned long)prof[1], (unsigned long)prof[2], (unsigned long)prof[3],
           (unsigned long)prof[4], (unsigned long)prof[5]);
#endif
  if constexpr (WGR) {
    // waves 1 .. NW - 1 hand their tables to wave 0 through the (now free) x tile buffer, one after
    // the other: plain stores by the giver, plain loads + adds by wave 0 (both pipeline; a
    // read-modify-write per element in LDS measured +50 us on a 100 us kernel), fixed order
    __syncthreads();                              // every wave is through its tiles
    float* red = &x_lds[0][0];                    // [N][LDX]: row stride = 4 (mod 16) floats, no bank conflicts
    for (int w = 1; w < NW; ++w) {
      if (wid == w) {
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
          for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(16 * nb + 4 * g + r) * LDX + 16 * kb + c] = C3[nb][kb][r];
      }
      __syncthreads();
      if (wid == 0) {
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
          for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) C3[nb][kb][r] += red[(16 * nb + 4 * g + r) * LDX + 16 * kb + c];
      }
      __syncthreads();
    }
  }
  if (!WGR || wid == 0) {
    float* gwp = gw_partial + (size_t)(WGR ? (int64_t)blockIdx.x : wave) * N * K;
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = 16 * kb + c;
          if (k < K) gwp[(size_t)(16 * nb + 4 * g + r) * K + k] = C3[nb][kb][r];
        }
  }
  if (pstat_partial) {
    double* pp = pstat_partial + (size_t)wave * (2 * K + 1);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const double a = xg_sum_d(p1[kb]), b = xg_sum_d(p2[kb]);
      const int k = 16 * kb + c;
      if (g == 0 && k < K) {
        pp[k] = a;
        pp[K + k] = b;
      }
    }
    if (lane == 0) pp[2 * K] = (wave == 0) ? (double)(r1 - r0) : 0.0;
  }
}

// sums of per-wave tables, fixed order: 16 columns x 64 slices per 1024-thread block (each
// thread sums <= ntab / 64 partials, 4-way unrolled: the loop is pure load latency; the
// 16-slice shape took 35-60 us per call, ~70 calls per train step at batch size)
template <typename T>
__device__ __forceinline__ void reduce_tables_body(const T* __restrict__ partial, int ntab, int len,
                                                   T* __restrict__ total, int accumulate,
                                                   int bx = -1) {
  __shared__ T sl[64][17];
  const int cl = threadIdx.x & 15;
  const int col = (bx < 0 ? (int)blockIdx.x : bx) * 16 + cl;
  const int slice = threadIdx.x >> 4;
  T acc = 0;
  if (col < len) {
    const int per = (ntab + 63) / 64;
    const int lo = slice * per, hi = (lo + per < ntab) ? lo + per : ntab;
    int k = lo;
    // eight records (four doubles) in flight, then four, then one (more spills under the 64-register cap): the loop is pure load latency (4 096 per-wave
    // tables of a train batch's level-0 layers = 64 records per slice: 16 round trips at four in
    // flight, 30 us per call); the adds keep the plain loop's order
    constexpr int UNR = sizeof(T) == 4 ? 8 : 4;
    for (; k + UNR <= hi; k += UNR) {
      T a[UNR];
#pragma unroll
      for (int j = 0; j < UNR; ++j) a[j] = partial[(size_t)(k + j) * len + col];
#pragma unroll
      for (int j = 0; j < UNR; ++j) acc += a[j];
    }
    for (; k + 4 <= hi; k += 4) {
      const T a0 = partial[(size_t)k * len + col], a1 = partial[(size_t)(k + 1) * len + col];
      const T a2 = partial[(size_t)(k + 2) * len + col], a3 = partial[(size_t)(k + 3) * len + col];
      acc += a0; acc += a1; acc += a2; acc += a3;           // same order as the plain loop
    }
    for (; k < hi; ++k) acc += partial[(size_t)k * len + col];
  }
  sl[slice][cl] = acc;
  __syncthreads();
  if (slice == 0 && col < len) {
    T t = 0;
#pragma unroll
    for (int k = 0; k < 64; ++k) t += sl[k][cl];             // fixed order: deterministic
    total[col] = accumulate ? total[col] + t : t;
  }
}
template <typename T>
__global__ __launch_bounds__(1024) void reduce_tables_kernel(const T* __restrict__ partial,
                                                             int ntab, int len,
                                                             T* __restrict__ total, int accumulate) {
  reduce_tables_body<T>(partial, ntab, len, total, accumulate);
}
// one table per graph: blockIdx.y = graph, its records = [start, start + count) of `partial`
template <typename T>
__global__ __launch_bounds__(1024) void reduce_tables_groups_kernel(const T* __restrict__ partial,
                                                                    FmlpGroups grp, int len,
                                                                    T* __restrict__ total) {
  const int b = blockIdx.y;
  reduce_tables_body<T>(partial + (size_t)grp.start[b] * len, grp.count[b], len,
                        total + (size_t)b * len, 0);
}

// ---- one "post" launch per fused layer call (round 6) --------------------------------------------
// The train-batch step is made of ~360 launches of a few microseconds: a fused layer used to be
// followed by its table sums and then - a separate C entry called by the host - by the GraphNorm
// table kernel that consumes them (forward: 3 launches per layer, backward: 4).  The sums of one
// column in a fixed order and the table formulas are unchanged (sliced_col_sum below is
// reduce_tables_body's loop, the formulas are gn_fwd_tables_kernel's / gn_bwd_tables_kernel's of
// graphnorm.hip): the same bits out of 2 launches per layer and direction.
// NV column sums side by side (one pass over the records: the loads of all columns are in flight
// together - a sum is pure load latency, NV sums one after the other cost NV round-trip chains).
// Per column the adds happen in reduce_tables_body's order: slices of ceil(ntab / 64) records,
// then the 64 slice sums in ascending order.  col[v] < 0: column v is not wanted (returns 0).
template <int NV>
__device__ __forceinline__ void sliced_col_sums(const double* const (&base)[NV], const int (&ntab)[NV],
                                                int len, const int (&col)[NV], double (*sl)[17],
                                                double (&out)[NV]) {
  const int cl = threadIdx.x & 15, slice = threadIdx.x >> 4;
  double acc[NV];
  int lo[NV], hi[NV], kmax = 0;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    acc[v] = 0;
    const int per = (ntab[v] + 63) / 64;
    lo[v] = slice * per;
    hi[v] = (lo[v] + per < ntab[v]) ? lo[v] + per : ntab[v];
    if (col[v] < 0) hi[v] = lo[v];
    const int n = hi[v] - lo[v];
    kmax = n > kmax ? n : kmax;
  }
  // UNR records of every column in flight (12 loads per round trip; one record at a time was
  // 16-32 round trips per slice at a train batch's table counts; more than 12 doubles in flight
  // spill under the 64-register cap of these 1 024-thread blocks)
  constexpr int UNR = NV <= 3 ? 4 : 1;
  int k = 0;
  if constexpr (UNR > 1)
  for (; k + UNR <= kmax; k += UNR) {
    double t[NV][UNR];
#pragma unroll
    for (int j = 0; j < UNR; ++j)
#pragma unroll
      for (int v = 0; v < NV; ++v)
        t[v][j] = (lo[v] + k + j < hi[v]) ? base[v][(size_t)(lo[v] + k + j) * len + col[v]] : 0.0;
#pragma unroll
    for (int j = 0; j < UNR; ++j)
#pragma unroll
      for (int v = 0; v < NV; ++v)
        if (lo[v] + k + j < hi[v]) acc[v] += t[v][j];        // (record order: the plain loop's)
  }
  for (; k < kmax; ++k) {
    double t[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v)
      t[v] = (lo[v] + k < hi[v]) ? base[v][(size_t)(lo[v] + k) * len + col[v]] : 0.0;
#pragma unroll
    for (int v = 0; v < NV; ++v)
      if (lo[v] + k < hi[v]) acc[v] += t[v];                 // (record order: the plain loop's)
  }
  // the slices meet through ONE [64][17] buffer, column after column (the loads above were the
  // slow part; a kernel-wide 52 KB of LDS for six buffers halved the residency of the
  // weight-gradient blocks of the same launch)
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    sl[slice][cl] = acc[v];
    __syncthreads();
    double t = 0;
    if (slice == 0) {                                        // (only these threads use the totals)
#pragma unroll 8
      for (int k = 0; k < 64; ++k) t += sl[k][cl];           // fixed order: deterministic
    }
    out[v] = t;
    __syncthreads();
  }
}

// forward: per-graph totals [2N + 1] of the layer's output AND the tables of its GraphNorm.
// grid (ceil(N / 16), num_graphs), 1024 threads = 16 channels x 64 slices.
__global__ __launch_bounds__(1024) void fwd_post_kernel(const double* __restrict__ partial,
                                                        FmlpGroups grp, int N,
                                                        double* __restrict__ total,
                                                        spt_gn_fwd_tables t) {
  __shared__ double sl[64][17];
  const int b = blockIdx.y, cl = threadIdx.x & 15, slice = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl, len = 2 * N + 1;
  const bool cv = c < N;
  const double* pb = partial + (size_t)grp.start[b] * len;
  const double* const base[3] = {pb, pb, pb};
  const int ntab[3] = {grp.count[b], grp.count[b], grp.count[b]};
  const int col[3] = {cv ? c : -1, cv ? N + c : -1, 2 * N};
  double s3[3];
  sliced_col_sums<3>(base, ntab, len, col, sl, s3);
  const double s1 = s3[0], s2 = s3[1], cnt = s3[2];
  if (slice != 0) return;
  if (total) {
    double* tr = total + (size_t)b * len;
    if (cv) {
      tr[c] = s1;
      tr[N + c] = s2;
    }
    if (blockIdx.x == 0 && cl == 0) tr[2 * N] = cnt;
  }
  if (!cv) return;
  double n = cnt;
  if (n < 1.0) n = 1.0;                       // scatter_mean: clamp(count, 1)
  const double mu = s1 / n;
  const double a = (double)t.mean_scale[c];
  double var = s2 / n - (2.0 * a - a * a) * mu * mu;
  if (var < 0.0) var = 0.0;
  const double rs = 1.0 / sqrt(var + (double)t.eps);
  const float mu32 = (float)mu, rs32 = (float)rs;
  const int o = b * N + c;
  t.mean[o] = mu32;
  t.rstd[o] = rs32;
  t.am[o] = (float)(a * (double)mu32);
  t.scale[o] = (float)((double)t.weight[c] * (double)rs32);
}

// backward: blocks [0, nA) sum the weight-gradient partials (reduce_tables_kernel<float>); the
// blocks behind them the statistics of the PREVIOUS layer's GraphNorm backward - plain totals
// (`pn.c1 == nullptr`: (2K + 1 + 15) / 16 blocks per graph, as reduce_tables_groups_kernel), or 16
// channels per block over all graphs (POST_GB of them side by side) with that norm's backward
// tables written on the spot.
constexpr int POST_GB = 2;
__global__ __launch_bounds__(1024, 8) void bwd_post_kernel(const float* __restrict__ gwp, int ntab_w,
                                                        int NK, float* __restrict__ gW, int accumulate,
                                                        int nA, const double* __restrict__ pst,
                                                        FmlpGroups grp, int K, int B,
                                                        double* __restrict__ prev_total,
                                                        spt_gn_bwd_tables pn) {
  if ((int)blockIdx.x < nA) {
    reduce_tables_body<float>(gwp, ntab_w, NK, gW, accumulate, (int)blockIdx.x);
    return;
  }
  const int bx = (int)blockIdx.x - nA, len = 2 * K + 1;
  if (!pn.c1) {
    const int per_graph = (len + 15) / 16;
    const int b = bx / per_graph;
    reduce_tables_body<double>(pst + (size_t)grp.start[b] * len, grp.count[b], len,
                               prev_total + (size_t)b * len, 0, bx - b * per_graph);
    return;
  }
  __shared__ double sl[64][17];
  const int cl = threadIdx.x & 15, slice = threadIdx.x >> 4;
  const int c = bx * 16 + cl;
  const bool cv = c < K;
  const double w = cv ? (double)pn.weight[c] : 0.0, a = cv ? (double)pn.mean_scale[c] : 0.0;
  double gw = 0.0, gb = 0.0, ga = 0.0;
  for (int b0 = 0; b0 < B; b0 += POST_GB) {
    const int nb = (B - b0 < POST_GB) ? B - b0 : POST_GB;
    const double* base[3 * POST_GB];
    int ntab[3 * POST_GB], col[3 * POST_GB];
#pragma unroll
    for (int j = 0; j < POST_GB; ++j) {
      const int b = b0 + (j < nb ? j : 0);
      const double* pb = pst + (size_t)grp.start[b] * len;
      base[3 * j] = base[3 * j + 1] = base[3 * j + 2] = pb;
      ntab[3 * j] = ntab[3 * j + 1] = ntab[3 * j + 2] = grp.count[b];
      col[3 * j] = (j < nb && cv) ? c : -1;
      col[3 * j + 1] = (j < nb && cv) ? K + c : -1;
      col[3 * j + 2] = j < nb ? 2 * K : -1;
    }
    double s[3 * POST_GB];
    sliced_col_sums<3 * POST_GB>(base, ntab, len, col, sl, s);
    if (slice != 0) continue;
#pragma unroll
    for (int j = 0; j < POST_GB; ++j) {                      // (compile-time indices: registers)
      if (j >= nb) break;
      const int b = b0 + j;
      const double A = s[3 * j], GO = s[3 * j + 1], cnt = s[3 * j + 2];
      if (prev_total) {
        double* tr = prev_total + (size_t)b * len;
        if (cv) {
          tr[c] = A;
          tr[K + c] = GO;
        }
        if (bx == 0 && cl == 0) tr[2 * K] = cnt;
      }
      if (!cv) continue;
      double n = cnt;
      if (n < 1.0) n = 1.0;
      const double sd = (double)pn.rstd[b * K + c], mu = (double)pn.mean[b * K + c];
      const double k2 = w * sd * sd * sd * GO / n;
      const double sumdo = w * sd * A - k2 * n * mu * (1.0 - a);
      pn.c1[b * K + c] = (float)(w * sd);
      pn.c2[b * K + c] = (float)k2;
      pn.c3[b * K + c] = (float)(a * sumdo / n);
      gw += sd * GO;
      gb += A;
      ga += -mu * sumdo;
    }
  }
  if (slice == 0 && cv) {
    pn.gweight[c] = (float)gw;
    pn.gbias[c] = (float)gb;
    pn.gmean_scale[c] = (float)ga;
  }
}

static void bwd_post_launch(const float* gwp, int ntab_w, int NK, float* gW, int accumulate,
                            const double* pst, const FmlpGroups& grp, int K, int B,
                            double* prev_total, const spt_gn_bwd_tables* pn, hipStream_t stream) {
  const int nA = (NK + 15) / 16;
  spt_gn_bwd_tables none = {};
  int nB = 0;
  if (pn && pn->c1) nB = (K + 15) / 16;
  else if (prev_total) nB = ((2 * K + 1 + 15) / 16) * B;
  bwd_post_kernel<<<nA + nB, 1024, 0, stream>>>(gwp, ntab_w, NK, gW, accumulate, nA, pst, grp, K, B,
                                                prev_total, (pn && pn->c1) ? *pn : none);
}

static int grid_for_nw(int64_t rows, int per_cu, int nwv) {
  const int64_t tiles = (rows + TR - 1) / TR;
  int64_t blocks = (tiles + nwv - 1) / nwv;
  const int64_t cap = 256 * per_cu;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

static int grid_for(int64_t rows, int per_cu) {
  const int64_t tiles = (rows + TR - 1) / TR;
  int64_t blocks = (tiles + WAVES - 1) / WAVES;
  const int64_t cap = 256 * per_cu;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

}  // namespace fmlp

// fused_mlp_dma.hip: the backward with LDS-DMA tile staging
bool fmlp_dma_supported(int K, int N);
int fmlp_dma_bwd_launch(bool pooled, bool lo, const float* gy, const float* h, FmlpRuns rt,
                        int64_t max_rows, int N, const float* am, const float* sc, const float* bs,
                        float slope, const float* c1, const float* c2, const float* c3,
                        const float* xprev, int K, const float* pam, const float* psc,
                        const float* pbs, float pslope, const float* W, float* gx,
                        float* gw_partial, double* pstat_partial, const int32_t* perm,
                        const int32_t* pos_seg, const float* gout, const int32_t* arg,
                        hipStream_t stream, bool s16 = false, int* gw_tabs = nullptr);
}  // namespace spt

using namespace spt;
using namespace spt::fmlp;

// shapes built: (K, N) of the SPT MLPs; K <= 64 (K4 <= 16), N in {32, 64, 128}
#define SPT_FMLP_SHAPES(X) X(3, 2) X(5, 2) X(8, 2) X(8, 4) X(16, 4) X(16, 8) X(17, 4) X(33, 4)

// 0: f32-in / f32-accumulate MFMA everywhere (bitwise an fmaf chain);
// 1 (default): the BACKWARD GEMMs (gW, gx) on the bf16 matrix pipe with split operands - the
//    gradients' parity bars are 1e-4 of the tensor's scale, ten times the split's error; the
//    forward's outputs feed GraphNorm statistics and are held to 2e-5 (measured with the 2-way
//    split forward: 1e-4 after three normalised layers): it runs the 3-way split (6 products,
//    f32-exact, fwd_kernel_x3) where that is enabled (spt_fused_linear_fwd_use_x3, default on),
//    else the f32 matrix pipe;
// 2: forward too;  3: plain bf16 operands (hi halves only) in both directions - the bf16
//    precision mode.  Process-wide, returns the previous setting.
static std::atomic<int> g_fmlp_mode{1};
// per-call mode word of the *_ex entries: < 0 = the process default above, else 0..3
static inline int fmlp_mode_of(int mode) { return mode < 0 ? (int)g_fmlp_mode : (mode & 3); }
// Backward formulation of the split-bf16 / bf16 modes: 1 (default) = tiles staged by LDS-DMA
// (fused_mlp_dma.hip) where the shape is built, 0 = register-staged everywhere (the process-wide
// switch holds under a per-call precision too, like the attention's formulation default).  Per
// call: bit 2 of the mode word (SPT_FMLP_BWD_REGISTER_STAGED) forces the register-staged kernels.
// forward of mode 1: 1 (default) = 3-way split on the bf16 pipe (fwd_kernel_x3), 0 = f32 matrix pipe
static std::atomic<int> g_fmlp_x3{[] { const char* e = getenv("SPT_FMLP_X3"); return e ? (atoi(e) != 0) : 1; }()};
extern "C" int spt_fused_linear_fwd_use_x3(int on) {
  const int prev = g_fmlp_x3;
  if (on >= 0) g_fmlp_x3 = on != 0;
  return prev;
}
static std::atomic<int> g_fmlp_dma{1};
static inline bool fmlp_dma_of(int mode) { return g_fmlp_dma != 0 && !(mode >= 0 && (mode & 4)); }
extern "C" int spt_fused_linear_bwd_use_dma(int on) {
  const int prev = g_fmlp_dma;
  if (on >= 0) g_fmlp_dma = on != 0;
  return prev;
}
extern "C" int spt_fused_linear_use_split_bf16(int mode) {
  const int prev = g_fmlp_mode;
  g_fmlp_mode = mode < 0 ? 0 : (mode > 3 ? 3 : mode);
  return prev;
}

// top layers that exist in front of a max-pool: the point MLP's last layer (64 -> 128 semantic,
// 64 -> 64 panoptic, 32 -> 64 for small MLPs)
#define SPT_FMLP_POOLED_SHAPES(X) X(16, 8) X(16, 4) X(8, 4)
extern "C" int spt_fused_linear_pooled_supported(int K, int N) {
  return spt_fused_linear_pooled_supported_ex(K, N, -1);
}
extern "C" int spt_fused_linear_pooled_supported_ex(int K, int N, int mode) {
  const int k4 = (K + 3) / 4, nbk = N / 16;
  if (N % 16 || fmlp_mode_of(mode) < 1) return 0;
#define X(a, b) if (k4 == a && nbk == b) return 1;
  SPT_FMLP_POOLED_SHAPES(X)
#undef X
  return 0;
}

// bf16 ACTIVATION STORAGE (precision mode 3 only): bit 3 of the mode word = h is read / written as
// bf16, bit 4 = x / xprev holds bf16 values.  Built for the point MLP's shapes (the level-0 chain
// carries ~90 % of a step's activation bytes); register-staged backward.
#define SPT_FMLP_ST_SHAPES(X) X(3, 2) X(8, 4) X(16, 8) X(16, 4)
#define SPT_FMLP_ST_POOLED_SHAPES(X) X(16, 8) X(16, 4) X(8, 4)
extern "C" int spt_fused_linear_storage_supported(int K, int N) {
  const int k4 = (K + 3) / 4, nbk = N / 16;
  if (N % 16) return 0;
#define X(a, b) if (k4 == a && nbk == b) return 1;
  SPT_FMLP_ST_SHAPES(X)
#undef X
  return 0;
}

extern "C" int spt_fused_linear_supported(int K, int N) {
  const int k4 = (K + 3) / 4, nbk = N / 16;
  if (N % 16) return 0;
#define X(a, b) if (k4 == a && nbk == b) return 1;
  SPT_FMLP_SHAPES(X)
#undef X
  return 0;
}

extern "C" size_t spt_fused_linear_workspace_bytes(int K, int N) {
  // fwd: MAX_BLOCKS x (2N+1) doubles; bwd: 1024 waves x (N*K floats + (2K+1) doubles)
  const size_t fwd = (size_t)MAX_BLOCKS * (2 * N + 1) * 8;
  const size_t bwd = (size_t)MAX_BWD_WAVES * ((size_t)N * K * 4 + (2 * K + 1) * 8);
  return align_up(fwd > bwd ? fwd : bwd, 256) + 4096;
}

// ---- run tables -------------------------------------------------------------------------------
// A launch covers `nruns` row ranges; run r = rows [r0[r], r1[r]) of graph g[r], runs sorted by
// graph (host arrays).  Per-graph coefficient tables are [num_graphs, width] arrays indexed by
// g[r]; `total` / `prev_total` are [num_graphs, 2 width + 1].
static const char* fmlp_make_runs(int nruns, const int64_t* r0, const int64_t* r1, const int32_t* g,
                                  int num_graphs, FmlpRuns* rt, int64_t* max_rows) {
  if (nruns < 1 || nruns > FMLP_MAX_RUNS) return "1 <= nruns <= 16";
  if (num_graphs < 1 || num_graphs > FMLP_MAX_RUNS) return "1 <= num_graphs <= 16";
  if (!r0 || !r1 || !g) return "null run table";
  rt->n = nruns;
  *max_rows = 0;
  for (int r = 0; r < nruns; ++r) {
    if (r1[r] < r0[r] || g[r] < 0 || g[r] >= num_graphs) return "bad run";
    if (r && g[r] < g[r - 1]) return "runs must be sorted by graph";
    rt->r0[r] = r0[r];
    rt->r1[r] = r1[r];
    rt->g[r] = g[r];
    if (r1[r] - r0[r] > *max_rows) *max_rows = r1[r] - r0[r];
  }
  return nullptr;
}
// records of graph b when every run owns `per_run` consecutive records
static FmlpGroups fmlp_groups(const FmlpRuns& rt, int num_graphs, int per_run) {
  FmlpGroups grp;
  for (int b = 0; b < FMLP_MAX_RUNS; ++b) grp.start[b] = grp.count[b] = 0;
  for (int r = 0; r < rt.n; ++r) {
    const int b = rt.g[r];
    if (grp.count[b] == 0) grp.start[b] = r * per_run;
    grp.count[b] += per_run;
  }
  (void)num_graphs;
  return grp;
}

static int fmlp_fwd_impl(const char* fn, const float* x, const FmlpRuns& rt, int64_t max_rows,
                         int num_graphs, int K, const float* W, int N, const float* pre_am,
                         const float* pre_scale, const float* pre_bias, float pre_slope, float* h,
                         double* total, int mode, void* ws, size_t ws_bytes, hipStream_t stream,
                         const spt_gn_fwd_tables* norm = nullptr) {
  const int g_fmlp_mode = fmlp_mode_of(mode);   // shadows the process default inside this call
  (void)fn;
  SPT_CHECK_ARG(K >= 1 && N >= 16, "bad shape");
  SPT_CHECK_ARG(spt_fused_linear_supported(K, N), "(K, N) not built");
  SPT_CHECK_ARG(x && W && h && (total || norm) && ws, "null pointer");
  SPT_CHECK_ARG(!norm || (norm->weight && norm->mean_scale && norm->mean && norm->rstd && norm->am &&
                          norm->scale), "incomplete norm tables");
  // the per-graph totals, or (norm) totals + the GraphNorm's forward tables, in one launch
  auto post = [&](const double* partial_, int gx__) {
    if (norm)
      fwd_post_kernel<<<dim3((N + 15) / 16, num_graphs), 1024, 0, stream>>>(
          partial_, fmlp_groups(rt, num_graphs, gx__), N, total, *norm);
    else
      reduce_tables_groups_kernel<double><<<dim3((2 * N + 1 + 15) / 16, num_graphs), 1024, 0, stream>>>(
          partial_, fmlp_groups(rt, num_graphs, gx__), 2 * N + 1, total);
  };
  SPT_CHECK_ARG(ws_bytes >= spt_fused_linear_workspace_bytes(K, N), "workspace too small");
  SPT_CHECK_ARG(!pre_am || (pre_scale && pre_bias), "incomplete pre-normalisation tables");
  const int k4 = (K + 3) / 4, nbk = N / 16;
  // small layers are latency-bound: give every SIMD 4 waves to overlap tiles (the
  // 64 -> 128 layer holds W in 128 B-operand registers and fits 2)
  const int per_cu = (k4 * nbk <= 32) ? 4 : 2;
  int gx_ = grid_for(max_rows, per_cu);
  const int cap = MAX_BLOCKS / rt.n;
  if (gx_ > cap) gx_ = cap;
  const bool x3 = g_fmlp_mode == 1 && g_fmlp_x3 && !(mode >= 0 && (mode & (SPT_FMLP_H_BF16 | SPT_FMLP_X_BF16)));
  if (x3) {
    // 8-wave workgroups (the W planes are shared through LDS): one or two per CU
    const int64_t tiles = (max_rows + TR - 1) / TR;
    int64_t blocks = (tiles + WAVES_X3 - 1) / WAVES_X3;
    const int64_t cap8 = 256 * ((k4 * nbk <= 32) ? 2 : 1);
    if (blocks > cap8) blocks = cap8;
    if (blocks > cap) blocks = cap;
    gx_ = (int)(blocks < 1 ? 1 : blocks);
  }
  const dim3 grid((unsigned)gx_, (unsigned)rt.n);
  const dim3 grid8 = grid;
  double* partial = (double*)ws;
  const bool h16 = mode >= 0 && (mode & SPT_FMLP_H_BF16), x16 = mode >= 0 && (mode & SPT_FMLP_X_BF16);
  if (h16 || x16) {
    SPT_CHECK_ARG(g_fmlp_mode == 3 && h16 && spt_fused_linear_storage_supported(K, N) &&
                  (!x16 || K % 32 == 0),
                  "bf16 activation storage: bf16 matrix mode, a built shape, h stored as bf16");
#define XS(a, b)                                                                                   \
  if (k4 == a && nbk == b) {                                                                       \
    if (x16)                                                                                       \
      fwd_kernel_bf<a, b, false, true, true><<<grid, WAVES * 64, 0, stream>>>(                     \
          x, 0, 0, K, W, pre_am, pre_scale, pre_bias, pre_slope, h, partial, rt);                  \
    else                                                                                           \
      fwd_kernel_bf<a, b, false, false, true><<<grid, WAVES * 64, 0, stream>>>(                    \
          x, 0, 0, K, W, pre_am, pre_scale, pre_bias, pre_slope, h, partial, rt);                  \
  }
    SPT_FMLP_ST_SHAPES(XS)
#undef XS
    post(partial, gx_);
    SPT_CHECK_LAUNCH();
    return 0;
  }
#define X(a, b)                                                                        \
  if (k4 == a && nbk == b) {                                                           \
    if (g_fmlp_mode == 3)                                                              \
      fwd_kernel_bf<a, b, false><<<grid, WAVES * 64, 0, stream>>>(x, 0, 0, K, W, pre_am, pre_scale, \
                                                                  pre_bias, pre_slope, h, partial, rt); \
    else if (g_fmlp_mode == 2)                                                         \
      fwd_kernel_bf<a, b, true><<<grid, WAVES * 64, 0, stream>>>(x, 0, 0, K, W, pre_am, pre_scale, \
                                                                 pre_bias, pre_slope, h, partial, rt);  \
    else if (g_fmlp_mode == 1 && g_fmlp_x3)                                            \
      fwd_kernel_x3<a, b><<<grid8, WAVES_X3 * 64, 0, stream>>>(x, 0, 0, K, W, pre_am, pre_scale, \
                                                               pre_bias, pre_slope, h, partial, rt); \
    else                                                                               \
      fwd_kernel<a, b><<<grid, WAVES * 64, 0, stream>>>(x, 0, 0, K, W, pre_am, pre_scale, \
                                                        pre_bias, pre_slope, h, partial, rt);  \
  }
  SPT_FMLP_SHAPES(X)
#undef X
  post(partial, gx_);
  SPT_CHECK_LAUNCH();
  return 0;
}

// h[r0:r1, :N] = act(gn_prev(x))[r0:r1, :K] W^T ; total[2N+1] = column sums, sums of
// squares and the row count of h over [r0, r1)  (one graph).
extern "C" int spt_fused_linear_fwd_f32(const float* x, int64_t r0, int64_t r1, int K,
                                        const float* W, int N, const float* pre_am,
                                        const float* pre_scale, const float* pre_bias,
                                        float pre_slope, float* h, double* total, void* ws,
                                        size_t ws_bytes, spt_stream_t stream_) {
  return spt_fused_linear_fwd_ex_f32(x, r0, r1, K, W, N, pre_am, pre_scale, pre_bias, pre_slope, h,
                                     total, -1, ws, ws_bytes, stream_);
}
extern "C" int spt_fused_linear_fwd_ex_f32(const float* x, int64_t r0, int64_t r1, int K,
                                           const float* W, int N, const float* pre_am,
                                           const float* pre_scale, const float* pre_bias,
                                           float pre_slope, float* h, double* total, int mode,
                                           void* ws, size_t ws_bytes, spt_stream_t stream_) {
  SPT_CHECK_ARG(r1 >= r0, "bad shape");
  FmlpRuns rt;
  rt.n = 1; rt.g[0] = 0; rt.r0[0] = r0; rt.r1[0] = r1;
  return fmlp_fwd_impl(__func__, x, rt, r1 - r0, 1, K, W, N, pre_am, pre_scale, pre_bias, pre_slope,
                       h, total, mode, ws, ws_bytes, (hipStream_t)stream_);
}
// The rows of SEVERAL graphs in one launch (see "run tables" above): pre_am / pre_scale
// [num_graphs, K] (or NULL), total [num_graphs, 2N+1].
extern "C" int spt_fused_linear_fwd_runs_f32(const float* x, int nruns, const int64_t* run_r0,
                                             const int64_t* run_r1, const int32_t* run_graph,
                                             int num_graphs, int K, const float* W, int N,
                                             const float* pre_am, const float* pre_scale,
                                             const float* pre_bias, float pre_slope, float* h,
                                             double* total, int mode, void* ws, size_t ws_bytes,
                                             spt_stream_t stream_) {
  FmlpRuns rt;
  int64_t max_rows;
  const char* err = fmlp_make_runs(nruns, run_r0, run_r1, run_graph, num_graphs, &rt, &max_rows);
  SPT_CHECK_ARG(!err, err ? err : "");
  return fmlp_fwd_impl(__func__, x, rt, max_rows, num_graphs, K, W, N, pre_am, pre_scale, pre_bias,
                       pre_slope, h, total, mode, ws, ws_bytes, (hipStream_t)stream_);
}
// ... and the forward tables of the layer's GraphNorm written by the same call (round 6: the
// totals' sum and spt_graphnorm_tables_f32 in one launch; `total` may be NULL then).
extern "C" int spt_fused_linear_fwd_runs_gn_f32(const float* x, int nruns, const int64_t* run_r0,
                                                const int64_t* run_r1, const int32_t* run_graph,
                                                int num_graphs, int K, const float* W, int N,
                                                const float* pre_am, const float* pre_scale,
                                                const float* pre_bias, float pre_slope, float* h,
                                                double* total, int mode, void* ws, size_t ws_bytes,
                                                const spt_gn_fwd_tables* norm, spt_stream_t stream_) {
  FmlpRuns rt;
  int64_t max_rows;
  const char* err = fmlp_make_runs(nruns, run_r0, run_r1, run_graph, num_graphs, &rt, &max_rows);
  SPT_CHECK_ARG(!err, err ? err : "");
  SPT_CHECK_ARG(norm, "null norm descriptor");
  return fmlp_fwd_impl(__func__, x, rt, max_rows, num_graphs, K, W, N, pre_am, pre_scale, pre_bias,
                       pre_slope, h, total, mode, ws, ws_bytes, (hipStream_t)stream_, norm);
}

static int fmlp_bwd_impl(bool pooled, const float* gy, const float* gout, const int32_t* arg,
                         const int32_t* perm, const int32_t* pos_seg, const float* h,
                         const FmlpRuns& rt, int64_t max_rows, int num_graphs, int N,
                         const float* am, const float* scale, const float* bias, float slope,
                         const float* c1, const float* c2, const float* c3, const float* xprev,
                         int K, const float* pre_am, const float* pre_scale, const float* pre_bias,
                         float pre_slope, const float* W, float* gx, float* gW, int accumulate,
                         double* prev_total, int mode, void* ws, size_t ws_bytes,
                         hipStream_t stream, const spt_gn_bwd_tables* prev_norm = nullptr) {
  const int g_fmlp_mode = fmlp_mode_of(mode);
  const bool g_fmlp_split_bf16 = g_fmlp_mode >= 1;
  // (prev_norm: the previous layer's backward tables are written by this call's post launch; its
  // statistics are then needed whether or not the caller wants the totals themselves)
  const bool want_prev = prev_total || prev_norm;
  SPT_CHECK_ARG(!prev_norm || (prev_norm->weight && prev_norm->mean_scale && prev_norm->mean &&
                               prev_norm->rstd && prev_norm->c1 && prev_norm->c2 && prev_norm->c3 &&
                               prev_norm->gweight && prev_norm->gbias && prev_norm->gmean_scale),
                "incomplete norm tables");
  SPT_CHECK_ARG(K >= 1 && N >= 16, "bad shape");
  SPT_CHECK_ARG(h && am && scale && bias && c1 && c2 && c3 && xprev && W && gW && ws, "null pointer");
  SPT_CHECK_ARG(ws_bytes >= spt_fused_linear_workspace_bytes(K, N), "workspace too small");
  if (pooled) {
    SPT_CHECK_ARG(spt_fused_linear_pooled_supported_ex(K, N, g_fmlp_mode),
                  "(K, N) has no pooled kernel in this matrix mode");
    SPT_CHECK_ARG(gout && arg && perm && pos_seg && gx, "null pointer");
    SPT_CHECK_ARG(!want_prev || pre_am, "previous-layer statistics need its tables");
  } else {
    SPT_CHECK_ARG(spt_fused_linear_supported(K, N), "(K, N) not built");
    SPT_CHECK_ARG(gy, "null pointer");
    SPT_CHECK_ARG(!want_prev || (gx && pre_am), "previous-layer statistics need gx and its tables");
  }
  const int k4 = (K + 3) / 4, nbk = N / 16;
  int gx_ = 1, nwv = 1;                        // blocks per run, waves per block
  float* gwp = (float*)ws;
  double* pst = (double*)((char*)ws + align_up((size_t)MAX_BWD_WAVES * N * K * 4, 256));
  double* pstp = want_prev ? pst : nullptr;
  const int nr = rt.n;
  auto cap_grid = [&](int g_, int nw_) {       // every run's waves own a record of the partial tables
    const int cap = MAX_BWD_WAVES / (nw_ * nr);
    return g_ > cap ? (cap < 1 ? 1 : cap) : g_;
  };
#define XP(a, b)                                                                                 \
  if (k4 == a && nbk == b) {                                                                     \
    /* (the register-prefetch variant PIPE = true measured slower: 7.3 vs 6.7 ms at 64 -> 128; */ \
    /* at one wave per SIMD nothing overlaps the ~1 450 VALU instructions per tile)            */ \
    constexpr bool big = (a * b >= 32);                                                          \
    constexpr int NWB = (a * b > 128) ? 4 : (big ? 8 : 4);                                       \
    gx_ = cap_grid(grid_for_nw(max_rows, (a * b > 128) ? 1 : (big ? ((a * b <= 32) ? 2 : 1) : 4), NWB), NWB); \
    nwv = NWB;                                                                                   \
    gw_tabs = gx_ * (fmlp_bf_wg_reduce(a, b, NWB) ? 1 : NWB);                                    \
    const dim3 grid((unsigned)gx_, (unsigned)nr);                                                \
    if (g_fmlp_mode == 3)                                                                        \
      bwd_kernel_bf<a, b, true, NWB, false, true><<<grid, NWB * 64, 0, stream>>>(                \
          nullptr, h, 0, 0, am, scale, bias, slope, c1, c2, c3, xprev, K, pre_am, pre_scale,     \
          pre_bias, pre_slope, W, gx, gwp, pstp, rt, perm, pos_seg, gout, arg);                  \
    else                                                                                         \
      bwd_kernel_bf<a, b, true, NWB, true, true><<<grid, NWB * 64, 0, stream>>>(                 \
          nullptr, h, 0, 0, am, scale, bias, slope, c1, c2, c3, xprev, K, pre_am, pre_scale,     \
          pre_bias, pre_slope, W, gx, gwp, pstp, rt, perm, pos_seg, gout, arg);                  \
  }
#define X(a, b)                                                                                  \
  if (k4 == a && nbk == b) {                                                                     \
    constexpr bool big = (a * b >= 32);   /* W as LDS B operands, 8-wave blocks */              \
    constexpr int NWV = big ? 8 : 4;                                                             \
    /* workgroups per CU the registers / LDS allow: 4 x 4 waves for the small layers, */         \
    /* 2 x 8 waves while the LDS tiles stay under 80 KB, else 1 x 8 */                            \
    /* split-bf16: the K = 132 layer keeps W^T, its accumulators and both split operands live: */ \
    /* 4-wave workgroups (1 wave per SIMD, 512 registers) instead of spilling at 256           */ \
    constexpr int NWB = (a * b > 128) ? 4 : NWV;                                                 \
    if (g_fmlp_split_bf16) {                                                                     \
      gx_ = cap_grid(grid_for_nw(max_rows, (a * b > 128) ? 1 : (big ? ((a * b <= 32) ? 2 : 1) : 4), NWB), NWB); \
      nwv = NWB;                                                                                 \
      gw_tabs = gx_ * (fmlp_bf_wg_reduce(a, b, NWB) ? 1 : NWB);                                  \
    } else {                                                                                     \
      gx_ = cap_grid(grid_for_nw(max_rows, big ? ((a * b <= 32) ? 2 : 1) : 4, NWV), NWV);        \
      nwv = NWV;                                                                                 \
      gw_tabs = gx_ * NWV;                                                                       \
    }                                                                                            \
    const dim3 grid((unsigned)gx_, (unsigned)nr);                                                \
    /* round 6: the one-wave-per-SIMD instances (K = 132) prefetch the next tile's raw rows into */ \
    /* registers (PIPE): nothing else hides their two memory round trips per tile                */ \
    constexpr bool PIPE_ = (a * b > 128);           /* (whole 16-byte row chunks: K % 4 == 0) */  \
    if (PIPE_ && K % 4 == 0 && g_fmlp_mode == 3 && gx)                                           \
      bwd_kernel_bf<a, b, true, NWB, false, false, PIPE_><<<grid, NWB * 64, 0, stream>>>(        \
          gy, h, 0, 0, am, scale, bias, slope, c1, c2, c3, xprev, K, pre_am, pre_scale,          \
          pre_bias, pre_slope, W, gx, gwp, pstp, rt);                                            \
    else if (PIPE_ && K % 4 == 0 && g_fmlp_split_bf16 && gx)                                     \
      bwd_kernel_bf<a, b, true, NWB, true, false, PIPE_><<<grid, NWB * 64, 0, stream>>>(         \
          gy, h, 0, 0, am, scale, bias, slope, c1, c2, c3, xprev, K, pre_am, pre_scale,          \
          pre_bias, pre_slope, W, gx, gwp, pstp, rt);                                            \
    else if (g_fmlp_mode == 3 && gx)                                                             \
      bwd_kernel_bf<a, b, true, NWB, false><<<grid, NWB * 64, 0, stream>>>(                      \
          gy, h, 0, 0, am, scale, bias, slope, c1, c2, c3, xprev, K, pre_am, pre_scale,          \
          pre_bias, pre_slope, W, gx, gwp, pstp, rt);                                            \
    else if (g_fmlp_mode == 3)                                                                   \
      bwd_kernel_bf<a, b, false, NWB, false><<<grid, NWB * 64, 0, stream>>>(                     \
          gy, h, 0, 0, am, scale, bias, slope, c1, c2, c3, xprev, K, pre_am, pre_scale,          \
          pre_bias, pre_slope, W, nullptr, gwp, nullptr, rt);                                    \
    else if (g_fmlp_split_bf16 && gx)                                                            \
      bwd_kernel_bf<a, b, true, NWB, true><<<grid, NWB * 64, 0, stream>>>(                       \
          gy, h, 0, 0, am, scale, bias, slope, c1, c2, c3, xprev, K, pre_am, pre_scale,          \
          pre_bias, pre_slope, W, gx, gwp, pstp, rt);                                            \
    else if (g_fmlp_split_bf16)                                                                  \
      bwd_kernel_bf<a, b, false, NWB, true><<<grid, NWB * 64, 0, stream>>>(                      \
          gy, h, 0, 0, am, scale, bias, slope, c1, c2, c3, xprev, K, pre_am, pre_scale,          \
          pre_bias, pre_slope, W, nullptr, gwp, nullptr, rt);                                    \
    else if (gx)                                                                                 \
      bwd_kernel<a, b, true, NWV, big><<<grid, NWV * 64, 0, stream>>>(                           \
          gy, h, 0, 0, am, scale, bias, slope, c1, c2, c3, xprev, K, pre_am, pre_scale,          \
          pre_bias, pre_slope, W, gx, gwp, pstp, rt);                                            \
    else                                                                                         \
      bwd_kernel<a, b, false, NWV, big><<<grid, NWV * 64, 0, stream>>>(                          \
          gy, h, 0, 0, am, scale, bias, slope, c1, c2, c3, xprev, K, pre_am, pre_scale,          \
          pre_bias, pre_slope, W, nullptr, gwp, nullptr, rt);                                    \
  }
  int per_run = 0;                              // wave records per run (statistics tables)
  int gw_tabs = 0;                              // weight-gradient tables per run (one per wave, or
                                                // one per workgroup: fmlp_bf_wg_reduce)
  const bool h16 = mode >= 0 && (mode & SPT_FMLP_H_BF16), x16 = mode >= 0 && (mode & SPT_FMLP_X_BF16);
  if (h16 || x16) {
    SPT_CHECK_ARG(g_fmlp_mode == 3 && h16 && spt_fused_linear_storage_supported(K, N) &&
                  (!x16 || K % 16 == 0),
                  "bf16 activation storage: bf16 matrix mode, a built shape, h stored as bf16");
#define XSP(a, b)                                                                                \
  if (pooled && k4 == a && nbk == b) {                                                           \
    constexpr bool big = (a * b >= 32);                                                          \
    constexpr int NWB = (a * b > 128) ? 4 : (big ? 8 : 4);                                       \
    gx_ = cap_grid(grid_for_nw(max_rows, (a * b > 128) ? 1 : (big ? ((a * b <= 32) ? 2 : 1) : 4), NWB), NWB); \
    nwv = NWB;                                                                                   \
    gw_tabs = gx_ * (fmlp_bf_wg_reduce(a, b, NWB) ? 1 : NWB);                                    \
    const dim3 grid((unsigned)gx_, (unsigned)nr);                                                \
    if (x16)                                                                                     \
      bwd_kernel_bf<a, b, true, NWB, false, true, false, true, true><<<grid, NWB * 64, 0, stream>>>( \
          nullptr, h, 0, 0, am, scale, bias, slope, c1, c2, c3, xprev, K, pre_am, pre_scale,     \
          pre_bias, pre_slope, W, gx, gwp, pstp, rt, perm, pos_seg, gout, arg);                  \
    else                                                                                         \
      bwd_kernel_bf<a, b, true, NWB, false, true, false, true, false><<<grid, NWB * 64, 0, stream>>>( \
          nullptr, h, 0, 0, am, scale, bias, slope, c1, c2, c3, xprev, K, pre_am, pre_scale,     \
          pre_bias, pre_slope, W, gx, gwp, pstp, rt, perm, pos_seg, gout, arg);                  \
  }
#define XSD(a, b)                                                                                \
  if (!pooled && k4 == a && nbk == b) {                                                          \
    constexpr bool big = (a * b >= 32);                                                          \
    constexpr int NWV = big ? 8 : 4;                                                             \
    constexpr int NWB = (a * b > 128) ? 4 : NWV;                                                 \
    gx_ = cap_grid(grid_for_nw(max_rows, (a * b > 128) ? 1 : (big ? ((a * b <= 32) ? 2 : 1) : 4), NWB), NWB); \
    nwv = NWB;                                                                                   \
    gw_tabs = gx_ * (fmlp_bf_wg_reduce(a, b, NWB) ? 1 : NWB);                                    \
    const dim3 grid((unsigned)gx_, (unsigned)nr);                                                \
    if (gx && x16)                                                                               \
      bwd_kernel_bf<a, b, true, NWB, false, false, false, true, true><<<grid, NWB * 64, 0, stream>>>( \
          gy, h, 0, 0, am, scale, bias, slope, c1, c2, c3, xprev, K, pre_am, pre_scale,          \
          pre_bias, pre_slope, W, gx, gwp, pstp, rt);                                            \
    else if (gx)                                                                                 \
      bwd_kernel_bf<a, b, true, NWB, false, false, false, true, false><<<grid, NWB * 64, 0, stream>>>( \
          gy, h, 0, 0, am, scale, bias, slope, c1, c2, c3, xprev, K, pre_am, pre_scale,          \
          pre_bias, pre_slope, W, gx, gwp, pstp, rt);                                            \
    else if (x16)                                                                                \
      bwd_kernel_bf<a, b, false, NWB, false, false, false, true, true><<<grid, NWB * 64, 0, stream>>>( \
          gy, h, 0, 0, am, scale, bias, slope, c1, c2, c3, xprev, K, pre_am, pre_scale,          \
          pre_bias, pre_slope, W, nullptr, gwp, nullptr, rt);                                    \
    else                                                                                         \
      bwd_kernel_bf<a, b, false, NWB, false, false, false, true, false><<<grid, NWB * 64, 0, stream>>>( \
          gy, h, 0, 0, am, scale, bias, slope, c1, c2, c3, xprev, K, pre_am, pre_scale,          \
          pre_bias, pre_slope, W, nullptr, gwp, nullptr, rt);                                    \
  }
    if (x16 && gx && fmlp_dma_of(mode) && fmlp_dma_supported(K, N)) {
      per_run = fmlp_dma_bwd_launch(pooled, false, gy, h, rt, max_rows, N, am, scale, bias, slope, c1,
                                    c2, c3, xprev, K, pre_am, pre_scale, pre_bias, pre_slope, W, gx,
                                    gwp, pstp, perm, pos_seg, gout, arg, stream, true, &gw_tabs);
    } else {
      SPT_FMLP_ST_POOLED_SHAPES(XSP)
      SPT_FMLP_ST_SHAPES(XSD)
      per_run = gx_ * nwv;
    }
#undef XSP
#undef XSD
  } else if (g_fmlp_split_bf16 && gx && fmlp_dma_of(mode) && fmlp_dma_supported(K, N)) {
    per_run = fmlp_dma_bwd_launch(pooled, g_fmlp_mode != 3, gy, h, rt, max_rows, N, am, scale, bias,
                                  slope, c1, c2, c3, xprev, K, pre_am, pre_scale, pre_bias, pre_slope,
                                  W, gx, gwp, pstp, perm, pos_seg, gout, arg, stream, false, &gw_tabs);
  } else if (pooled) {
    SPT_FMLP_POOLED_SHAPES(XP)
    per_run = gx_ * nwv;
  } else {
    SPT_FMLP_SHAPES(X)
    per_run = gx_ * nwv;
  }
#undef X
#undef XP
  SPT_CHECK_ARG(per_run > 0, "no kernel for this shape");
  // ONE post launch: the weight gradient's sum, the previous layer's statistics and - prev_norm -
  // that layer's backward tables (they were three launches: two here, gn_bwd_tables_kernel behind
  // a second C entry)
  bwd_post_launch(gwp, gw_tabs * nr, N * K, gW, accumulate, pst, fmlp_groups(rt, num_graphs, per_run), K,
                  num_graphs, prev_total, prev_norm, stream);
  SPT_CHECK_LAUNCH();
  return 0;
}

// Backward of the TOP layer when its output went through a segment max-pool (see POOLED above):
// positions [p0, p1) of the pool's CSR order instead of a row range, (gout, arg) instead of gy.
// Split-bf16 kernels only (mode >= 1); gx is required.
extern "C" int spt_fused_linear_bwd_pooled_f32(
    const float* gout, const int32_t* arg, const int32_t* perm, const int32_t* pos_seg,
    const float* h, int64_t p0, int64_t p1, int N, const float* am, const float* scale,
    const float* bias, float slope, const float* c1, const float* c2, const float* c3,
    const float* xprev, int K, const float* pre_am, const float* pre_scale, const float* pre_bias,
    float pre_slope, const float* W, float* gx, float* gW, int accumulate, double* prev_total,
    void* ws, size_t ws_bytes, spt_stream_t stream_) {
  return spt_fused_linear_bwd_pooled_ex_f32(gout, arg, perm, pos_seg, h, p0, p1, N, am, scale, bias,
                                            slope, c1, c2, c3, xprev, K, pre_am, pre_scale, pre_bias,
                                            pre_slope, W, gx, gW, accumulate, prev_total, -1, ws,
                                            ws_bytes, stream_);
}
extern "C" int spt_fused_linear_bwd_pooled_ex_f32(
    const float* gout, const int32_t* arg, const int32_t* perm, const int32_t* pos_seg,
    const float* h, int64_t p0, int64_t p1, int N, const float* am, const float* scale,
    const float* bias, float slope, const float* c1, const float* c2, const float* c3,
    const float* xprev, int K, const float* pre_am, const float* pre_scale, const float* pre_bias,
    float pre_slope, const float* W, float* gx, float* gW, int accumulate, double* prev_total,
    int mode, void* ws, size_t ws_bytes, spt_stream_t stream_) {
  SPT_CHECK_ARG(p1 >= p0, "bad shape");
  FmlpRuns rt;
  rt.n = 1; rt.g[0] = 0; rt.r0[0] = p0; rt.r1[0] = p1;
  return fmlp_bwd_impl(true, nullptr, gout, arg, perm, pos_seg, h, rt, p1 - p0, 1, N, am, scale, bias,
                       slope, c1, c2, c3, xprev, K, pre_am, pre_scale, pre_bias, pre_slope, W, gx, gW,
                       accumulate, prev_total, mode, ws, ws_bytes, (hipStream_t)stream_);
}
// Same over the CSR position ranges of several graphs in one launch: tables [num_graphs, N] /
// [num_graphs, K], prev_total [num_graphs, 2K+1]; gW receives the sum over all runs.
extern "C" int spt_fused_linear_bwd_pooled_runs_f32(
    const float* gout, const int32_t* arg, const int32_t* perm, const int32_t* pos_seg,
    const float* h, int nruns, const int64_t* run_p0, const int64_t* run_p1,
    const int32_t* run_graph, int num_graphs, int N, const float* am, const float* scale,
    const float* bias, float slope, const float* c1, const float* c2, const float* c3,
    const float* xprev, int K, const float* pre_am, const float* pre_scale, const float* pre_bias,
    float pre_slope, const float* W, float* gx, float* gW, double* prev_total, int mode, void* ws,
    size_t ws_bytes, spt_stream_t stream_) {
  FmlpRuns rt;
  int64_t max_rows;
  const char* err = fmlp_make_runs(nruns, run_p0, run_p1, run_graph, num_graphs, &rt, &max_rows);
  SPT_CHECK_ARG(!err, err ? err : "");
  return fmlp_bwd_impl(true, nullptr, gout, arg, perm, pos_seg, h, rt, max_rows, num_graphs, N, am,
                       scale, bias, slope, c1, c2, c3, xprev, K, pre_am, pre_scale, pre_bias,
                       pre_slope, W, gx, gW, 0, prev_total, mode, ws, ws_bytes, (hipStream_t)stream_);
}
// ... with the PREVIOUS layer's GraphNorm-backward tables written by the same call (round 6:
// spt_graphnorm_bwd_tables_f32 folded into the post launch; prev_total may be NULL then).
extern "C" int spt_fused_linear_bwd_pooled_runs_gn_f32(
    const float* gout, const int32_t* arg, const int32_t* perm, const int32_t* pos_seg,
    const float* h, int nruns, const int64_t* run_p0, const int64_t* run_p1,
    const int32_t* run_graph, int num_graphs, int N, const float* am, const float* scale,
    const float* bias, float slope, const float* c1, const float* c2, const float* c3,
    const float* xprev, int K, const float* pre_am, const float* pre_scale, const float* pre_bias,
    float pre_slope, const float* W, float* gx, float* gW, double* prev_total, int mode, void* ws,
    size_t ws_bytes, const spt_gn_bwd_tables* prev_norm, spt_stream_t stream_) {
  FmlpRuns rt;
  int64_t max_rows;
  const char* err = fmlp_make_runs(nruns, run_p0, run_p1, run_graph, num_graphs, &rt, &max_rows);
  SPT_CHECK_ARG(!err, err ? err : "");
  SPT_CHECK_ARG(prev_norm, "null norm descriptor");
  return fmlp_bwd_impl(true, nullptr, gout, arg, perm, pos_seg, h, rt, max_rows, num_graphs, N, am,
                       scale, bias, slope, c1, c2, c3, xprev, K, pre_am, pre_scale, pre_bias,
                       pre_slope, W, gx, gW, 0, prev_total, mode, ws, ws_bytes, (hipStream_t)stream_,
                       prev_norm);
}

// Backward of one layer over the rows [r0, r1) of one graph.
//   gy, h [rows, N]; (am, scale, bias, slope): this layer's GraphNorm forward tables;
//   (c1, c2, c3): its backward coefficient rows; xprev [rows, K] raw input of the layer
//   with its own (pre_*) tables or NULL; gx [rows, K] or NULL; gW [N, K] is ACCUMULATED
//   into when `accumulate` (further graphs of the batch); prev_total [2K+1] or NULL.
extern "C" int spt_fused_linear_bwd_f32(const float* gy, const float* h, int64_t r0, int64_t r1,
                                        int N, const float* am, const float* scale,
                                        const float* bias, float slope, const float* c1,
                                        const float* c2, const float* c3, const float* xprev,
                                        int K, const float* pre_am, const float* pre_scale,
                                        const float* pre_bias, float pre_slope, const float* W,
                                        float* gx, float* gW, int accumulate,
                                        double* prev_total, void* ws, size_t ws_bytes,
                                        spt_stream_t stream_) {
  return spt_fused_linear_bwd_ex_f32(gy, h, r0, r1, N, am, scale, bias, slope, c1, c2, c3, xprev, K,
                                     pre_am, pre_scale, pre_bias, pre_slope, W, gx, gW, accumulate,
                                     prev_total, -1, ws, ws_bytes, stream_);
}
extern "C" int spt_fused_linear_bwd_ex_f32(const float* gy, const float* h, int64_t r0, int64_t r1,
                                           int N, const float* am, const float* scale,
                                           const float* bias, float slope, const float* c1,
                                           const float* c2, const float* c3, const float* xprev,
                                           int K, const float* pre_am, const float* pre_scale,
                                           const float* pre_bias, float pre_slope, const float* W,
                                           float* gx, float* gW, int accumulate,
                                           double* prev_total, int mode, void* ws, size_t ws_bytes,
                                           spt_stream_t stream_) {
  SPT_CHECK_ARG(r1 >= r0, "bad shape");
  FmlpRuns rt;
  rt.n = 1; rt.g[0] = 0; rt.r0[0] = r0; rt.r1[0] = r1;
  return fmlp_bwd_impl(false, gy, nullptr, nullptr, nullptr, nullptr, h, rt, r1 - r0, 1, N, am, scale,
                       bias, slope, c1, c2, c3, xprev, K, pre_am, pre_scale, pre_bias, pre_slope, W,
                       gx, gW, accumulate, prev_total, mode, ws, ws_bytes, (hipStream_t)stream_);
}
// Same over the row ranges of several graphs in one launch (tables [num_graphs, .], prev_total
// [num_graphs, 2K+1], gW = the sum over all runs).
extern "C" int spt_fused_linear_bwd_runs_f32(
    const float* gy, const float* h, int nruns, const int64_t* run_r0, const int64_t* run_r1,
    const int32_t* run_graph, int num_graphs, int N, const float* am, const float* scale,
    const float* bias, float slope, const float* c1, const float* c2, const float* c3,
    const float* xprev, int K, const float* pre_am, const float* pre_scale, const float* pre_bias,
    float pre_slope, const float* W, float* gx, float* gW, double* prev_total, int mode, void* ws,
    size_t ws_bytes, spt_stream_t stream_) {
  FmlpRuns rt;
  int64_t max_rows;
  const char* err = fmlp_make_runs(nruns, run_r0, run_r1, run_graph, num_graphs, &rt, &max_rows);
  SPT_CHECK_ARG(!err, err ? err : "");
  return fmlp_bwd_impl(false, gy, nullptr, nullptr, nullptr, nullptr, h, rt, max_rows, num_graphs, N,
                       am, scale, bias, slope, c1, c2, c3, xprev, K, pre_am, pre_scale, pre_bias,
                       pre_slope, W, gx, gW, 0, prev_total, mode, ws, ws_bytes, (hipStream_t)stream_);
}
extern "C" int spt_fused_linear_bwd_runs_gn_f32(
    const float* gy, const float* h, int nruns, const int64_t* run_r0, const int64_t* run_r1,
    const int32_t* run_graph, int num_graphs, int N, const float* am, const float* scale,
    const float* bias, float slope, const float* c1, const float* c2, const float* c3,
    const float* xprev, int K, const float* pre_am, const float* pre_scale, const float* pre_bias,
    float pre_slope, const float* W, float* gx, float* gW, double* prev_total, int mode, void* ws,
    size_t ws_bytes, const spt_gn_bwd_tables* prev_norm, spt_stream_t stream_) {
  FmlpRuns rt;
  int64_t max_rows;
  const char* err = fmlp_make_runs(nruns, run_r0, run_r1, run_graph, num_graphs, &rt, &max_rows);
  SPT_CHECK_ARG(!err, err ? err : "");
  SPT_CHECK_ARG(prev_norm, "null norm descriptor");
  return fmlp_bwd_impl(false, gy, nullptr, nullptr, nullptr, nullptr, h, rt, max_rows, num_graphs, N,
                       am, scale, bias, slope, c1, c2, c3, xprev, K, pre_am, pre_scale, pre_bias,
                       pre_slope, W, gx, gW, 0, prev_total, mode, ws, ws_bytes, (hipStream_t)stream_,
                       prev_norm);
}

// ---- the top layer fused with the max-pool behind it (fused_pool.hip) -----------------------------
namespace spt {
bool fpool_supported(int K, int N);
int fpool_gram_len(int K);
int fpool_fwd_launch(int prec, bool in16, const float* x, const int32_t* perm, const int32_t* pos_seg,
                     const int32_t* rowptr, const FmlpRuns& rt, int64_t max_rows, int K, int N,
                     const float* W, const float* gnw, const float* pam, const float* psc,
                     const float* pbs, float pslope, float* raw, int32_t* argpos, double* partial,
                     hipStream_t stream);
void fpool_tables_launch(int K, const double* gram, int B, const float* W, int N, int bfw,
                         const float* weight, const float* mean_scale, float eps, double* total,
                         float* mean, float* rstd, float* am, float* scale, hipStream_t stream);
void fpool_apply_launch(int K, bool in16, const int32_t* rowptr, const int32_t* perm,
                        const int64_t* seg_graph, int64_t num_seg, int N, int64_t n_rows,
                        const float* am, const float* sc, const float* bs, float slope,
                        const float* gnw, const float* x, const float* W, const float* pam,
                        const float* psc, const float* pbs, float pslope, int bfw, float* raw,
                        int32_t* argpos, int32_t* arg, float* out, hipStream_t stream);
int fpool_bwd_launch(bool lo, bool x16, const float* gout, const float* raw, const int32_t* argpos,
                     const int32_t* perm, const int32_t* pos_seg, const int64_t* seg_graph,
                     int64_t num_seg, const FmlpRuns& rt, int64_t max_rows, int num_graphs, int K,
                     int N, const float* am, const float* sc, const float* bs, float slope,
                     const float* c1, const float* c2, const float* c3, const float* xprev,
                     const float* pam, const float* psc, const float* pbs, float pslope,
                     const float* W, float* gm, float* Mbuf, float* c0buf, float* gx,
                     float* gw_partial, double* pstat_partial, int max_waves, hipStream_t stream,
                     int* gw_tabs);
void fpool_gw_dense_launch(int K, const double* gram, int B, const float* W, int N, int bfw,
                           const float* am, const float* c2, const float* c3, float* gW,
                           hipStream_t stream);
}  // namespace spt

// matrix mode of the pool-fused top layer: forward 3 = f32-exact (3-way split), 2 = 2-way split,
// 1 = plain bf16; 0 = not built (the f32 matrix pipe keeps the materialised route)
static int fpool_prec_of(int mode) {
  const int m = fmlp_mode_of(mode);
  if (m == 1) return g_fmlp_x3 ? 3 : 0;
  return m == 2 ? 2 : (m == 3 ? 1 : 0);
}
extern "C" int spt_fused_linear_pool_supported(int K, int N, int mode) {
  return fpool_supported(K, N) && fpool_prec_of(mode) != 0;
}
extern "C" size_t spt_fused_linear_pool_gram_len(int K) { return (size_t)fpool_gram_len(K); }
extern "C" size_t spt_fused_linear_pool_workspace_bytes(int K, int N) {
  return align_up(spt_fused_linear_workspace_bytes(K, N), 256) +
         align_up((size_t)FMLP_MAX_RUNS * (K * K + K) * 4, 256) + 256;
}

extern "C" int spt_fused_linear_fwd_pool_runs_f32(
    const void* x, const int32_t* perm, const int32_t* pos_seg, const int32_t* rowptr,
    const int64_t* seg_graph, int64_t num_seg, int64_t n_rows, int nruns, const int64_t* run_p0,
    const int64_t* run_p1, const int32_t* run_graph, int num_graphs, int K, const float* W, int N,
    const float* gn_weight, const float* gn_bias, const float* gn_mean_scale, float eps, float slope,
    const float* pre_am, const float* pre_scale, const float* pre_bias, float pre_slope, float* out,
    int32_t* arg, int32_t* argpos, float* raw, double* gram, double* total, float* mean, float* rstd,
    float* am, float* scale, int mode, void* ws, size_t ws_bytes, spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  FmlpRuns rt;
  int64_t max_rows;
  const char* err = fmlp_make_runs(nruns, run_p0, run_p1, run_graph, num_graphs, &rt, &max_rows);
  SPT_CHECK_ARG(!err, err ? err : "");
  const int prec = fpool_prec_of(mode);
  const bool in16 = mode >= 0 && (mode & SPT_FMLP_X_BF16);
  SPT_CHECK_ARG(fpool_supported(K, N) && prec != 0 && (!in16 || prec == 1),
                "(K, N) has no pool-fused kernel in this matrix mode");
  SPT_CHECK_ARG(x && pos_seg && rowptr && W && gn_weight && gn_bias && gn_mean_scale && pre_am &&
                pre_scale && pre_bias && out && argpos && raw && gram && mean && rstd && am && scale && ws,
                "null pointer");
  SPT_CHECK_ARG(num_seg >= 0 && n_rows >= 0 && (num_graphs == 1 || seg_graph), "bad shape");
  SPT_CHECK_ARG(ws_bytes >= spt_fused_linear_pool_workspace_bytes(K, N), "workspace too small");
  // the runs are the graphs' CSR position ranges: contiguous, starting at 0, ending at n_rows
  for (int r = 0; r < rt.n; ++r)
    SPT_CHECK_ARG(rt.r0[r] == (r ? rt.r1[r - 1] : 0), "runs must tile the CSR positions");
  SPT_CHECK_ARG(rt.r1[rt.n - 1] == n_rows, "runs must tile the CSR positions");
  double* partial = (double*)ws;
  const int glen = fpool_gram_len(K);
  const int gx_ = fpool_fwd_launch(prec, in16, (const float*)x, perm, pos_seg, rowptr, rt, max_rows, K, N,
                                   W, gn_weight, pre_am, pre_scale, pre_bias, pre_slope, raw, argpos,
                                   partial, stream);
  SPT_CHECK_ARG(gx_ > 0, "no kernel for this variant");
  reduce_tables_groups_kernel<double><<<dim3((glen + 15) / 16, num_graphs), 1024, 0, stream>>>(
      partial, fmlp_groups(rt, num_graphs, gx_), glen, gram);
  fpool_tables_launch(K, gram, num_graphs, W, N, prec == 1, gn_weight, gn_mean_scale, eps, total, mean,
                      rstd, am, scale, stream);
  fpool_apply_launch(K, in16, rowptr, perm, seg_graph, num_seg, N, n_rows, am, scale, gn_bias, slope,
                     gn_weight, (const float*)x, W, pre_am, pre_scale, pre_bias, pre_slope, prec == 1,
                     raw, argpos, arg, out, stream);
  SPT_CHECK_LAUNCH();
  return 0;
}

static int fpool_bwd_entry(
    const float* gout, const float* raw, const int32_t* argpos, const int32_t* perm,
    const int32_t* pos_seg, const int64_t* seg_graph, int64_t num_seg, int nruns,
    const int64_t* run_p0, const int64_t* run_p1, const int32_t* run_graph, int num_graphs, int N,
    const float* am, const float* scale, const float* bias, float slope, const float* c1,
    const float* c2, const float* c3, const void* xprev, int K, const float* pre_am,
    const float* pre_scale, const float* pre_bias, float pre_slope, const float* W,
    const double* gram, float* gm, float* gx, float* gW, double* prev_total, int mode, void* ws,
    size_t ws_bytes, spt_stream_t stream_, const spt_gn_bwd_tables* prev_norm) {
  hipStream_t stream = (hipStream_t)stream_;
  FmlpRuns rt;
  int64_t max_rows;
  const char* err = fmlp_make_runs(nruns, run_p0, run_p1, run_graph, num_graphs, &rt, &max_rows);
  SPT_CHECK_ARG(!err, err ? err : "");
  const int prec = fpool_prec_of(mode);
  const bool x16 = mode >= 0 && (mode & SPT_FMLP_X_BF16);
  SPT_CHECK_ARG(fpool_supported(K, N) && prec != 0 && (!x16 || prec == 1),
                "(K, N) has no pool-fused kernel in this matrix mode");
  SPT_CHECK_ARG(gout && raw && argpos && pos_seg && am && scale && bias && c1 && c2 && c3 && xprev &&
                pre_am && pre_scale && pre_bias && W && gram && gm && gx && gW && (prev_total || prev_norm) && ws,
                "null pointer");
  SPT_CHECK_ARG(num_seg >= 0 && (num_graphs == 1 || seg_graph), "bad shape");
  SPT_CHECK_ARG(ws_bytes >= spt_fused_linear_pool_workspace_bytes(K, N), "workspace too small");
  float* gwp = (float*)ws;
  double* pst = (double*)((char*)ws + align_up((size_t)MAX_BWD_WAVES * N * K * 4, 256));
  float* Mbuf = (float*)((char*)ws + align_up(spt_fused_linear_workspace_bytes(K, N), 256));
  float* c0buf = Mbuf + (size_t)FMLP_MAX_RUNS * K * K;
  int gw_tabs = 0;
  const int per_run = fpool_bwd_launch(prec != 1, x16, gout, raw, argpos, perm, pos_seg, seg_graph, num_seg,
                                       rt, max_rows, num_graphs, K, N, am, scale, bias, slope, c1, c2, c3,
                                       (const float*)xprev, pre_am, pre_scale, pre_bias, pre_slope, W, gm,
                                       Mbuf, c0buf, gx, gwp, pst, MAX_BWD_WAVES, stream, &gw_tabs);
  SPT_CHECK_ARG(per_run > 0, "no kernel for this variant");
  bwd_post_launch(gwp, gw_tabs * rt.n, N * K, gW, 0, pst, fmlp_groups(rt, num_graphs, per_run), K,
                  num_graphs, prev_total, prev_norm, stream);
  fpool_gw_dense_launch(K, gram, num_graphs, W, N, prec == 1, am, c2, c3, gW, stream);
  SPT_CHECK_LAUNCH();
  return 0;
}
extern "C" int spt_fused_linear_bwd_pool_runs_f32(
    const float* gout, const float* raw, const int32_t* argpos, const int32_t* perm,
    const int32_t* pos_seg, const int64_t* seg_graph, int64_t num_seg, int nruns,
    const int64_t* run_p0, const int64_t* run_p1, const int32_t* run_graph, int num_graphs, int N,
    const float* am, const float* scale, const float* bias, float slope, const float* c1,
    const float* c2, const float* c3, const void* xprev, int K, const float* pre_am,
    const float* pre_scale, const float* pre_bias, float pre_slope, const float* W,
    const double* gram, float* gm, float* gx, float* gW, double* prev_total, int mode, void* ws,
    size_t ws_bytes, spt_stream_t stream_) {
  return fpool_bwd_entry(gout, raw, argpos, perm, pos_seg, seg_graph, num_seg, nruns, run_p0, run_p1, run_graph,
                         num_graphs, N, am, scale, bias, slope, c1, c2, c3, xprev, K, pre_am, pre_scale,
                         pre_bias, pre_slope, W, gram, gm, gx, gW, prev_total, mode, ws, ws_bytes, stream_, nullptr);
}
// ... with the PREVIOUS layer's GraphNorm-backward tables written by the post launch (round 6)
extern "C" int spt_fused_linear_bwd_pool_runs_gn_f32(
    const float* gout, const float* raw, const int32_t* argpos, const int32_t* perm,
    const int32_t* pos_seg, const int64_t* seg_graph, int64_t num_seg, int nruns,
    const int64_t* run_p0, const int64_t* run_p1, const int32_t* run_graph, int num_graphs, int N,
    const float* am, const float* scale, const float* bias, float slope, const float* c1,
    const float* c2, const float* c3, const void* xprev, int K, const float* pre_am,
    const float* pre_scale, const float* pre_bias, float pre_slope, const float* W,
    const double* gram, float* gm, float* gx, float* gW, double* prev_total, int mode, void* ws,
    size_t ws_bytes, const spt_gn_bwd_tables* prev_norm, spt_stream_t stream_) {
  SPT_CHECK_ARG(prev_norm, "null norm descriptor");
  return fpool_bwd_entry(gout, raw, argpos, perm, pos_seg, seg_graph, num_seg, nruns, run_p0, run_p1, run_graph,
                         num_graphs, N, am, scale, bias, slope, c1, c2, c3, xprev, K, pre_am, pre_scale,
                         pre_bias, pre_slope, W, gram, gm, gx, gW, prev_total, mode, ws, ws_bytes, stream_, prev_norm);
}
