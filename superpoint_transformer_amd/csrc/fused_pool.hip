// The point MLP's TOP layer and the level-0 -> level-1 max-pool as one algebraic unit
// (src/nn/mlp.py:43-56 bias-free Linear -> GraphNorm -> LeakyReLU, src/nn/stage.py:413-431,
// src/nn/pool.py:61-82 max over the children of a superpoint): the layer's [rows, N] output is
// consumed by nothing but the pool, so it is never written, read or recomputed.
//
// Three exact identities (h = y_prev W^T raw, y = leaky(sc (h - am) + bs), out[s] = max_i y_i):
//  (i)  y is a per-channel MONOTONE function of h - non-decreasing for GraphNorm weight w_c >= 0,
//       non-increasing for w_c < 0, and every rounding of its f32 evaluation is monotone - so
//       max_i y_ic = y(max_i h_ic) (w_c >= 0) or y(min_i h_ic) (w_c < 0), bit for bit.  The
//       forward pools the RAW tile out of the MFMA accumulators (the sign of w_c folded into W's
//       row, so it is always a max) while walking the rows in the pool's CSR order, and writes
//       one (raw value, arg row) pair per (segment, channel).  The arg is the first row
//       attaining the raw extremum: the reference's arg (first row attaining max y) except where
//       two rows with DIFFERENT h round to the same y - there both rows hold the maximum and the
//       reference's own choice hangs on the last bit of its norm (tests/test_fused_pool_gpu.py).
//       w_c == 0 (y constant: every row ties, the reference takes the first): arg = first row of
//       the segment and its true h, restored by pool_apply_kernel.
//  (ii) the norm's statistics come from the layer's INPUT: sum_i h_ic = w_c . sum_i y_prev_i,
//       sum_i h_ic^2 = w_c^T G w_c with G = sum_i y_prev_i y_prev_i^T (K x K, per graph),
//       accumulated on the matrix pipe (contraction = the tile's 16 rows) next to the product,
//       reduced and evaluated in f64.
//  (iii) backward: dL/dh_i = S_i + A + B o h_i with S the pool's sparse gradient (one non-zero per
//       (segment, channel), already times c1 and the activation's slope), A = c2 am - c3,
//       B = -c2 (the GraphNorm-backward coefficient rows), hence
//         g(y_prev)_i = S_i W + y_prev_i (W^T diag(B) W) + A W          (two GEMMs over the tile)
//         gW          = S^T y_prev + diag(B) W G + A (x) sum_i y_prev_i   (GEMM + closed form)
//       - h is neither read nor rebuilt; forming the dense GraphNorm backward of a 16 x N tile
//       becomes an `arg == row` select.
//
// Bytes at the S3DIS-scale scene (15 M rows, K = 64, N = 128): forward 3.84 GB (x) + 0.06 (ids) +
// 0.44 (raw, arg) instead of 11.5 GB + 8.2 GB for layer + pool; backward 3.84 (x) + 3.84 (gx) +
// 0.12 (ids) + 0.66 (gm, arg, raw) instead of 15.8 GB.
#include <math.h>

#include <type_traits>

#include "common.hpp"

namespace spt {
namespace fpool {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
// measurement builds (tools/build_variant.sh fp_skip<bits> fused_pool.hip -DSPT_FPOOL_SKIP=<bits>; wrong
// results by design, default 0 changes nothing): forward 1 no Gram statistics, 2 no segment max,
// 4 no product (the MFMAs of h' = y W'^T); backward 8 no weight-gradient product, 16 no input-gradient
// products, 32 no statistics of the previous norm
#ifndef SPT_FPOOL_SKIP
#define SPT_FPOOL_SKIP 0
#endif
#ifndef SPT_FPOOL_WPAD   // measurement builds: 8 = the row padding of the weight planes before round 6
#define SPT_FPOOL_WPAD 16
#endif
constexpr int TR = 16;           // rows per MFMA tile
constexpr int NWF = 8;           // waves per workgroup (the W planes are shared through LDS)
constexpr int ARG_NONE = 0x7fffffff;

__device__ __forceinline__ double xg_sum_d(double v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float xg_sum_f(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// four consecutive elements of a row as f32 (f32 storage: 16-byte load; bf16 storage: 8-byte
// load widened - exact)
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
template <int NV, typename V>
__device__ __forceinline__ void split2(const float (&x)[NV], V& hi, V& lo) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const __bf16 h = (__bf16)x[i];
    hi[i] = h;
    lo[i] = (__bf16)(x[i] - (float)h);
  }
}
template <int NV, typename V>
__device__ __forceinline__ void split3(const float (&x)[NV], V& h, V& m, V& l) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const __bf16 a = (__bf16)x[i];
    const float r1 = x[i] - (float)a;              // exact
    const __bf16 b = (__bf16)r1;
    const float r2 = r1 - (float)b;                // exact, fits bf16
    h[i] = a;
    m[i] = b;
    l[i] = (__bf16)r2;
  }
}
__device__ __forceinline__ f32x4 mfma16(const bf16x4& a, const bf16x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a),
                                                   __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}

// The 16 gathered rows x[rid] (rid of tile row rr in lane rr of `rid_l`) of K f32 / bf16 values:
// load_rows requests them into registers (all loads of the tile back to back: one round trip, and
// - issued a tile ahead - the trip runs under the previous tile's arithmetic), store_rows writes
// them to LDS (row stride LD floats), applying y = leaky((v - am) sc + bs) on the way when PRE
// (tab = am | sc | bs, K floats each; the expression of gn_apply_fwd_kernel).  Rows >= cnt are 0.
// (bf16 rows stay 8-byte register pairs while they travel and are widened when they are stored:
// widening at the load would put the wait for the load right behind it)
template <bool X16>
struct RowChunk { typedef float4 type; };
template <>
struct RowChunk<true> { typedef uint2 type; };
template <bool X16>
__device__ __forceinline__ float4 widen(const typename RowChunk<X16>::type& u) {
  if constexpr (X16) {
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                       __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
  } else {
    return u;
  }
}
template <int K, bool X16>
__device__ __forceinline__ void load_rows(const float* __restrict__ x, int rid_l,
                                          typename RowChunk<X16>::type (&v)[TR * (K / 4) / 64], int lane) {
  constexpr int CH = K / 4, NIT = TR * CH / 64;
  static_assert(TR * CH % 64 == 0, "whole waves of chunks");
  // UNCONDITIONAL loads (tile rows past the end carry the id of a valid row and are zeroed by
  // store_rows): a load under a per-lane condition is followed by a wait and a select, which
  // serialises the tile's loads and defeats issuing them a tile ahead
#pragma unroll
  for (int j = 0; j < NIT; ++j) {
    const int q = lane + 64 * j, rr = q / CH, k = (q - rr * CH) << 2;
    const int64_t xr = (int64_t)__shfl(rid_l, rr, 64);
    if constexpr (X16)
      v[j] = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(x) + xr * K + k);
    else
      v[j] = *reinterpret_cast<const float4*>(x + xr * K + k);
  }
}
template <int K, int LD, bool PRE, bool X16>
__device__ __forceinline__ void store_rows(const typename RowChunk<X16>::type (&v)[TR * (K / 4) / 64],
                                           int cnt, const float* tab, float slope, float* lds, int lane) {
  constexpr int CH = K / 4, NIT = TR * CH / 64;
#pragma unroll
  for (int j = 0; j < NIT; ++j) {
    const int q = lane + 64 * j, rr = q / CH, k = (q - rr * CH) << 2;
    float4 w = widen<X16>(v[j]);
    if constexpr (PRE) {
      const float4 a = *reinterpret_cast<const float4*>(tab + k);
      const float4 s = *reinterpret_cast<const float4*>(tab + K + k);
      const float4 b = *reinterpret_cast<const float4*>(tab + 2 * K + k);
      w.x = fmaf(w.x - a.x, s.x, b.x); w.y = fmaf(w.y - a.y, s.y, b.y);
      w.z = fmaf(w.z - a.z, s.z, b.z); w.w = fmaf(w.w - a.w, s.w, b.w);
      w.x = w.x > 0.f ? w.x : w.x * slope; w.y = w.y > 0.f ? w.y : w.y * slope;
      w.z = w.z > 0.f ? w.z : w.z * slope; w.w = w.w > 0.f ? w.w : w.w * slope;
    }
    if (rr >= cnt) w = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(lds + rr * LD + k) = w;
  }
}

// first position of a segment at or after p inside [r0, r1] (r0, r1 are segment boundaries)
__device__ __forceinline__ int64_t cut_at(int64_t p, int64_t r0, int64_t r1,
                                          const int32_t* __restrict__ pos_seg,
                                          const int32_t* __restrict__ rowptr) {
  if (p <= r0) return r0;
  if (p >= r1) return r1;
  const int s = __builtin_amdgcn_readfirstlane(pos_seg[p]);
  const int64_t b = (int64_t)__builtin_amdgcn_readfirstlane(rowptr[s]);
  if (b == p) return p;
  const int64_t e = (int64_t)__builtin_amdgcn_readfirstlane(rowptr[s + 1]);
  return e < r1 ? e : r1;
}

// ---- forward: product + Gram matrix + segment max of the raw tile ---------------------------------
// PREC 3: f32-exact product (3-way split operands, 6 bf16 products; the default `f32` mode);
//      2: 2-way split (3 products);  1: plain bf16 operands (the bf16 mode).
// IN16: x holds bf16 values (bf16 activation storage).
// One workgroup = 8 waves sharing W's bf16 planes in LDS; a wave owns a contiguous range of CSR
// positions cut at segment boundaries (no segment is shared between waves, nothing is merged).
// Multi-graph launches: blockIdx.y = run (a graph's contiguous CSR positions).
template <int K, int N, int PREC, bool IN16>
__global__ __launch_bounds__(NWF * 64, 2) void fwd_pool_kernel(
    const float* __restrict__ x, const int32_t* __restrict__ perm,
    const int32_t* __restrict__ pos_seg, const int32_t* __restrict__ rowptr,
    const float* __restrict__ W, const float* __restrict__ gnw, const float* __restrict__ pam,
    const float* __restrict__ psc, const float* __restrict__ pbs, float pslope,
    float* __restrict__ raw, int32_t* __restrict__ argpos, double* __restrict__ partial, FmlpRuns rt) {
  constexpr int KS = K / 32, KB = K / 16, NBK = N / 16, LDA = K + 4;
  // bf16 row stride of the W planes = 16 (mod 32): a ds_read_b128 is served in four groups of 16
  // lanes - {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) -
  // and the B fragment's address is (row c) stride + 16 bytes x (lane group g): with stride / 16 B
  // = 2 (mod 4) the 16 slots of every group are distinct; K + 8 (stride / 16 B odd) put 7 of a
  // group's lanes on a busy slot - every fragment read took 8 LDS cycles instead of 4
  // (profiles/r06y_pmc_fpool_sq.txt: SQ_LDS_BANK_CONFLICT 195.6 M -> 15.4 M cycles per call, LDS
  // cycles 582 M -> 401 M; the backward 283.9 M -> 103.9 M.  The kernels' TIME did not move - they
  // wait on dependent round trips, not on LDS throughput)
  constexpr int LDW = K + SPT_FPOOL_WPAD;
  static_assert(LDW % 32 == 16 || SPT_FPOOL_WPAD != 16, "conflict-free B fragments");
  constexpr int NPL = PREC == 3 ? 3 : (PREC == 2 ? 2 : 1);
  constexpr int NGB = KB * (KB + 1) / 2;                 // upper-triangular 16 x 16 blocks of G
  constexpr int GLEN = K * K + K + 1;
  static_assert(K % 32 == 0 && N % 16 == 0, "shape");
  __shared__ __attribute__((aligned(16))) float a_lds[NWF][TR * LDA];
  __shared__ __attribute__((aligned(16))) float tab[4 * K];             // am | sc | bs | Gram shift
  __shared__ __attribute__((aligned(16))) float sgn_l[N];
  __shared__ __attribute__((aligned(16))) __bf16 wpl[NPL][N * LDW];
  __shared__ double gred[K * K + K];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int run = blockIdx.y, gph = rt.g[run];
  const int64_t r0 = rt.r0[run], r1 = rt.r1[run];
  pam += (size_t)gph * K;
  psc += (size_t)gph * K;
  partial += ((size_t)run * gridDim.x + blockIdx.x) * GLEN;
  float* al = a_lds[wid];
  for (int i = threadIdx.x; i < K; i += NWF * 64) {
    tab[i] = pam[i];
    tab[K + i] = psc[i];
    tab[2 * K + i] = pbs[i];
    // Round 6 (advisor): the Gram matrix and the column sums are accumulated of y - shift, with
    // shift_k = leaky(bias_k) of the previous norm - the value a channel whose |mean| is large
    // against its spread sits at (the norm's output has mean ~ bias).  var = w^T G w / n - mu^2
    // cancels mu^2 / var of the f32 accumulation error over a wave's ~7 K rows; around the shift
    // the accumulated quantities are the spread itself.  The block's partial is un-shifted in f64
    // (G = G' + s sy'^T + sy' s^T + n s s^T, sy = sy' + n s), so nothing downstream changes.
    // The bf16 mode keeps shift 0: its statistics are those of the ROUNDED operands, one plane.
    const float bsv = pbs[i];
    tab[3 * K + i] = PREC == 1 ? 0.f : (bsv > 0.f ? bsv : pslope * bsv);
  }
  for (int i = threadIdx.x; i < N; i += NWF * 64) sgn_l[i] = gnw[i] < 0.f ? -1.f : 1.f;
  // W with the sign of the norm's weight folded into its rows (h' = sgn h: the pool is a max for
  // every channel; the Gram statistics do not see the sign), split into bf16 planes
  for (int i = threadIdx.x; i < N * K; i += NWF * 64) {
    const int n = i / K, k = i - n * K;
    const float w = gnw[n] < 0.f ? -W[i] : W[i];
    if constexpr (PREC == 3) {
      const float w1[1] = {w};
      __bf16 a[1], b[1], cc[1];
      split3<1>(w1, a, b, cc);
      wpl[0][n * LDW + k] = a[0];
      wpl[1][n * LDW + k] = b[0];
      wpl[2][n * LDW + k] = cc[0];
    } else if constexpr (PREC == 2) {
      const __bf16 hh = (__bf16)w;
      wpl[0][n * LDW + k] = hh;
      wpl[1][n * LDW + k] = (__bf16)(w - (float)hh);
    } else {
      wpl[0][n * LDW + k] = (__bf16)w;
    }
  }
  __syncthreads();

  f32x4 GA[NGB];
#pragma unroll
  for (int i = 0; i < NGB; ++i) GA[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float sy[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) sy[kb] = 0.f;
  // the wave's partial maximum of the open segment: its own rows (4 g + r of every tile) only,
  // with the CSR POSITION of the winner (inside a segment the positions ascend with the original
  // row ids - the view is a stable sort - so "first occurrence" is "smallest position"); the four
  // lane groups meet when the segment closes
  float pm[NBK];
  int pp[NBK];
#pragma unroll
  for (int nb = 0; nb < NBK; ++nb) {
    pm[nb] = -INFINITY;
    pp[nb] = ARG_NONE;
  }
  int cur_seg = -1;
  auto flush = [&](int seg) {
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) {
      float m = pm[nb];
      int a = pp[nb];
#pragma unroll
      for (int off = 16; off <= 32; off <<= 1) {
        const float om = __shfl_xor(m, off, 64);
        const int oa = __shfl_xor(a, off, 64);
        const bool t = (om > m) || (om == m && oa < a);      // segcsr's combine rule
        m = t ? om : m;
        a = t ? oa : a;
      }
      if ((nb & 3) == g) {                                   // the four groups share the stores
        const size_t o = (size_t)seg * N + 16 * nb + c;
        raw[o] = m * sgn_l[16 * nb + c];
        argpos[o] = a;
      }
      pm[nb] = -INFINITY;
      pp[nb] = ARG_NONE;
    }
  };

  const int64_t nw = (int64_t)gridDim.x * NWF, w = (int64_t)blockIdx.x * NWF + wid;
  const int64_t per = (((r1 - r0 + nw - 1) / nw) + TR - 1) / TR * TR;
  const int64_t pa0 = cut_at(r0 + w * per, r0, r1, pos_seg, rowptr);
  const int64_t pb0 = (w == nw - 1) ? r1 : cut_at(r0 + (w + 1) * per, r0, r1, pos_seg, rowptr);
  const bool has_rows = pa0 < pb0;            // (an empty range skips the loop; it still joins the
                                              //  workgroup's reduction below)
  // ids of the 16 positions from p (lanes past the wave's range read its last position: a valid
  // row that store_rows zeroes and the epilogue never looks at) - unconditional loads
  const int64_t plast = pb0 - 1;
  auto ids_of = [&](int64_t p, int& rid, int& sg) {
    int64_t q = p + (lane & (TR - 1));
    q = q < plast ? q : plast;
    rid = perm ? perm[q] : (int)q;
    sg = pos_seg[q];
  };
  if (has_rows) {
  // Software pipeline over the wave's tiles.  Memory operations are waited for by COUNT
  // (s_waitcnt vmcnt(n): at most n YOUNGER operations outstanding) and the results of a closing
  // segment are stored from data-dependent control flow, i.e. the compiler has to assume that
  // none of those stores was issued: any wait placed behind them drains them.  The order below
  // keeps every wait in front of the stores of its own iteration:
  //   multiply tile p | stage tile p+1 (its rows were requested before the multiplication) |
  //   take over the ids of tile p+2 (requested an iteration ago) | pool tile p, STORES |
  //   request the rows of tile p+2 and the ids of tile p+3.
  int seg_a, seg_b, rid_c, seg_c;               // segment ids of tile p, p+1; ids of tile p+2
  typename RowChunk<IN16>::type xv[TR * (K / 4) / 64];   // raw rows of the tile to be staged next
  {
    int rid0, rid1;
    ids_of(pa0, rid0, seg_a);
    load_rows<K, IN16>(x, rid0, xv, lane);
    ids_of(pa0 + TR, rid1, seg_b);
    ids_of(pa0 + 2 * TR, rid_c, seg_c);
    wave_sync_lds();
    store_rows<K, LDA, true, IN16>(xv, (int)((pb0 - pa0) < TR ? (pb0 - pa0) : TR), tab, pslope, al, lane);
    load_rows<K, IN16>(x, rid1, xv, lane);
  }
  for (int64_t p = pa0; p < pb0; p += TR) {
    const int cnt = (int)((pb0 - p) < TR ? (pb0 - p) : TR);
    wave_sync_lds();
    // ---- h' = y_prev W'^T ------------------------------------------------------------------
    f32x4 C[NBK];
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) C[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // (round 6: the W fragment of step i + 1 - a (k-step, column block) pair - is requested BEFORE
    //  the MFMAs of step i: left to the scheduler every fragment was read right in front of its own
    //  MFMAs and waited for, 16 exposed LDS round trips per tile - about as long as the MFMAs
    //  themselves at two waves per SIMD.  LA + 1 fragment register sets rotate.)
    if constexpr (!(SPT_FPOOL_SKIP & 4)) {
      constexpr int NSTEP = KS * NBK;
      constexpr int LA = NPL == 3 ? 2 : (NPL == 2 ? 3 : 6);  // steps of look-ahead (~100 MFMA cycles)
      bf16x8 wf[LA + 1][NPL];
      auto ldw = [&](int step, bf16x8 (&w)[NPL]) {
        const int ks = step / NBK, nb = step - ks * NBK;
        const int wo = (16 * nb + c) * LDW + 32 * ks + 8 * g;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) w[pl] = *reinterpret_cast<const bf16x8*>(&wpl[pl][wo]);
      };
#pragma unroll
      for (int i = 0; i < LA; ++i) ldw(i, wf[i]);
      bf16x8 x1, x2, x3;
#pragma unroll
      for (int step = 0; step < NSTEP; ++step) {
        const int ks = step / NBK, nb = step - ks * NBK;
        if (nb == 0) {
          const float4 a0 = *reinterpret_cast<const float4*>(al + c * LDA + 32 * ks + 8 * g);
          const float4 a1 = *reinterpret_cast<const float4*>(al + c * LDA + 32 * ks + 8 * g + 4);
          const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
          if constexpr (PREC == 3) {
            split3<8>(av, x1, x2, x3);
          } else if constexpr (PREC == 2) {
            split2<8>(av, x1, x2);
          } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) x1[i] = (__bf16)av[i];
          }
        }
        if (step + LA < NSTEP) ldw(step + LA, wf[(step + LA) % (LA + 1)]);
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 (&w)[NPL] = wf[step % (LA + 1)];
        f32x4 acc = C[nb];
        if constexpr (PREC == 3) {
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x3, w[0], acc, 0, 0, 0);   // smallest terms first
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1, w[NPL - 1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x2, w[NPL > 1 ? 1 : 0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x2, w[0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1, w[NPL > 1 ? 1 : 0], acc, 0, 0, 0);
        } else if constexpr (PREC == 2) {
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x2, w[0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1, w[NPL - 1], acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1, w[0], acc, 0, 0, 0);
        C[nb] = acc;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- G += y_prev^T y_prev (contraction = the tile's rows: the 4 rows a lane group holds of
    // one column are one packed operand), column sums of y_prev ---------------------------------
    if constexpr (!(SPT_FPOOL_SKIP & 1)) {
      bf16x4 Yh[KB], Yl[KB];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        float yv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) yv[r] = al[(4 * g + r) * LDA + 16 * kb + c];
        if constexpr (PREC != 1) {
          // around the shift (rows past the wave's range are zero rows of the tile: they stay zero)
          const float sh = tab[3 * K + 16 * kb + c];
#pragma unroll
          for (int r = 0; r < 4; ++r) yv[r] = (4 * g + r < cnt) ? yv[r] - sh : 0.f;
        }
        if constexpr (PREC == 1) {
          // the product above saw bf16(y): the statistics are those of what it computed
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            Yh[kb][r] = (__bf16)yv[r];
            yv[r] = (float)Yh[kb][r];
          }
        } else {
          split2<4>(yv, Yh[kb], Yl[kb]);
        }
        sy[kb] += (yv[0] + yv[1]) + (yv[2] + yv[3]);
      }
      int gi = 0;
#pragma unroll
      for (int mi = 0; mi < KB; ++mi)
#pragma unroll
        for (int ni = mi; ni < KB; ++ni) {
          f32x4 acc = GA[gi];
          if constexpr (PREC != 1) {
            // (lo lo too: on the diagonal it is a sum of squares - dropped, E[h^2] would sit a
            // systematic 1.3e-6 low; with it the split's error is zero-mean)
            acc = mfma16(Yl[mi], Yl[ni], acc);
            acc = mfma16(Yl[mi], Yh[ni], acc);
            acc = mfma16(Yh[mi], Yl[ni], acc);
          }
          GA[gi] = mfma16(Yh[mi], Yh[ni], acc);
          ++gi;
        }
    }
    // ---- tile p+1 into LDS (every lane has read its operands of tile p), ids of tile p+2 -------
    wave_sync_lds();
    {
      const int64_t left = pb0 - (p + TR);
      store_rows<K, LDA, true, IN16>(xv, (int)(left < 0 ? 0 : (left < TR ? left : TR)), tab, pslope, al, lane);
      asm volatile("" : "+v"(rid_c), "+v"(seg_c));                 // (the wait for them belongs HERE)
    }
    const int seg_cur = seg_a;
    // ---- segment max of the raw tile ---------------------------------------------------------
    if constexpr (!(SPT_FPOOL_SKIP & 2)) {
      const int pos0 = (int)p + 4 * g;                            // position of the lane's row r = 0
      int row = 0;
      while (row < cnt) {
        const int s = __builtin_amdgcn_readlane(seg_cur, row);
        const uint64_t diff = __ballot(lane < cnt && lane > row && seg_cur != s);
        const int e = diff ? (int)__builtin_ctzll(diff) : cnt;   // rows [row, e) belong to s
        if (s != cur_seg) {
          if (cur_seg >= 0) flush(cur_seg);
          cur_seg = s;
        }
        if (row == 0 && e == TR) {                                // the whole tile: no masks
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb) {
              const float v = C[nb][r];
              const bool win = v > pm[nb];
              pm[nb] = win ? v : pm[nb];
              pp[nb] = win ? pos0 + r : pp[nb];
            }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rr = 4 * g + r;
            const bool in = rr >= row && rr < e;
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb) {
              const float v = in ? C[nb][r] : -INFINITY;
              const bool win = v > pm[nb];
              pm[nb] = win ? v : pm[nb];
              pp[nb] = win ? pos0 + r : pp[nb];
            }
          }
        }
        row = e;
      }
    }
    // ---- requests for the tiles ahead (behind this tile's stores) -------------------------------
    load_rows<K, IN16>(x, rid_c, xv, lane);                       // rows of tile p+2
    seg_a = seg_b;
    seg_b = seg_c;
    ids_of(p + 3 * TR, rid_c, seg_c);
  }
  }
  if (cur_seg >= 0) flush(cur_seg);

  // ---- the workgroup's Gram partial: waves in a fixed order, f64 -----------------------------
  for (int i = threadIdx.x; i < K * K + K; i += NWF * 64) gred[i] = 0.0;
  __syncthreads();
  for (int ww = 0; ww < NWF; ++ww) {
    if (wid == ww) {
      int gi = 0;
#pragma unroll
      for (int mi = 0; mi < KB; ++mi)
#pragma unroll
        for (int ni = mi; ni < KB; ++ni) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            gred[(16 * mi + 4 * g + r) * K + 16 * ni + c] += (double)GA[gi][r];
          ++gi;
        }
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const double t = xg_sum_d((double)sy[kb]);
        if (g == 0) gred[K * K + 16 * kb + c] += t;
      }
    }
    __syncthreads();
  }
  // rows of this workgroup (its waves' ranges are consecutive): the un-shift needs the count
  const int64_t wfirst = (int64_t)blockIdx.x * NWF, wlast = wfirst + NWF - 1;
  const int64_t ba = cut_at(r0 + wfirst * per, r0, r1, pos_seg, rowptr);
  const int64_t bb = (wlast >= nw - 1) ? r1 : cut_at(r0 + (wlast + 1) * per, r0, r1, pos_seg, rowptr);
  const double nblk = (double)(bb > ba ? bb - ba : 0);
  for (int i = threadIdx.x; i < K * K; i += NWF * 64) {
    const int rr = i / K, cc = i - rr * K;
    const double gp = (rr / 16 <= cc / 16) ? gred[i] : gred[cc * K + rr];   // lower blocks: the mirror
    const double sr = (double)tab[3 * K + rr], sc_ = (double)tab[3 * K + cc];
    partial[i] = gp + sr * gred[K * K + cc] + gred[K * K + rr] * sc_ + nblk * sr * sc_;
  }
  for (int i = threadIdx.x; i < K; i += NWF * 64)
    partial[K * K + i] = gred[K * K + i] + nblk * (double)tab[3 * K + i];
  if (threadIdx.x == 0) partial[K * K + K] = (blockIdx.x == 0) ? (double)(r1 - r0) : 0.0;
}

// ---- statistics and coefficient rows of the norm from the Gram totals ----------------------------
// gram [B][K K + K + 1] (G | column sums | row count).  grid = (ceil(N / CPB), B); a block of
// CPB x K threads: thread (c, j) forms w_j (G w)_j and w_j sy_j of channel c, a tree over j sums
// them (f64, fixed order).  Same formulas and roundings as gn_fwd_tables_kernel (graphnorm.hip).
// BFW: the product ran on bf16(W) (the bf16 mode) - the statistics are those of what it computed.
template <int K>
__global__ __launch_bounds__(256) void gram_tables_kernel(
    const double* __restrict__ gram, const float* __restrict__ W, int N, int bfw,
    const float* __restrict__ weight, const float* __restrict__ mean_scale, float eps,
    double* __restrict__ total, float* __restrict__ mean, float* __restrict__ rstd,
    float* __restrict__ am, float* __restrict__ scale) {
  constexpr int GLEN = K * K + K + 1, CPB = 256 / K;
  __shared__ double gl[K * K + K];
  __shared__ double wl[CPB][K];
  __shared__ double r1[CPB][K], r2[CPB][K];
  const int b = blockIdx.y;
  const double* gr = gram + (size_t)b * GLEN;
  for (int i = threadIdx.x; i < K * K + K; i += 256) gl[i] = gr[i];
  const int cl = threadIdx.x / K, j = threadIdx.x - cl * K;
  const int c = blockIdx.x * CPB + cl;
  {
    float w = c < N ? W[(size_t)c * K + j] : 0.f;
    if (bfw) w = (float)(__bf16)w;
    wl[cl][j] = (double)w;
  }
  __syncthreads();
  double t = 0.0;
#pragma unroll 8
  for (int k = 0; k < K; ++k) t += gl[j * K + k] * wl[cl][k];
  r2[cl][j] = wl[cl][j] * t;
  r1[cl][j] = wl[cl][j] * gl[K * K + j];
  __syncthreads();
  for (int st = K / 2; st > 0; st >>= 1) {
    if (j < st) {
      r1[cl][j] += r1[cl][j + st];
      r2[cl][j] += r2[cl][j + st];
    }
    __syncthreads();
  }
  if (j != 0 || c >= N) return;
  double n = gr[K * K + K];
  const double nraw = n;
  if (n < 1.0) n = 1.0;
  const double s1 = r1[cl][0];
  double s2 = r2[cl][0];
  if (s2 < 0.0) s2 = 0.0;
  if (total) {
    double* tr = total + (size_t)b * (2 * N + 1);
    tr[c] = s1;
    tr[N + c] = s2;
    if (c == 0) tr[2 * N] = nraw;
  }
  const double mu = s1 / n;
  const double a = (double)mean_scale[c];
  double var = s2 / n - (2.0 * a - a * a) * mu * mu;
  if (var < 0.0) var = 0.0;
  const double rs = 1.0 / sqrt(var + (double)eps);
  const float mu32 = (float)mu, rs32 = (float)rs;
  const int o = b * N + c;
  mean[o] = mu32;
  rstd[o] = rs32;
  am[o] = (float)(a * (double)mu32);
  scale[o] = (float)((double)weight[c] * (double)rs32);
}

// ---- out = y(raw); empty segments; channels whose norm weight is exactly 0 ------------------------
// one thread per (segment, channel): arg = the original row of the winner's CSR position.
// w_c == 0: y is the constant leaky(bias_c) - every row ties and the reference's arg is the
// segment's first row; its raw value (the norm's weight gradient needs the true h of the arg row)
// is rebuilt from that row of x.
template <int K, bool IN16>
__global__ __launch_bounds__(256) void pool_apply_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ perm,
    const int64_t* __restrict__ seg_graph, int64_t num_seg, int N, int64_t n_rows,
    const float* __restrict__ am, const float* __restrict__ sc, const float* __restrict__ bs,
    float slope, const float* __restrict__ gnw, const float* __restrict__ x,
    const float* __restrict__ W, const float* __restrict__ pam, const float* __restrict__ psc,
    const float* __restrict__ pbs, float pslope, int bfw, float* __restrict__ raw,
    int32_t* __restrict__ argpos, int32_t* __restrict__ arg, float* __restrict__ out) {
  // (round 6: four channels per thread - 16-byte reads of raw / argpos, 16-byte writes of out / arg;
  //  N % 4 == 0.  The per-element arithmetic is unchanged.)
  const int64_t t4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (t4 >= num_seg * N) return;
  const int64_t s = t4 / N;
  const int c0 = (int)(t4 - s * N);
  const int a0 = rowptr[s], a1 = rowptr[s + 1];
  if (a1 <= a0) {                                        // empty: segcsr's convention
    const int32_t nr = (int32_t)n_rows;
    *reinterpret_cast<float4*>(out + t4) = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(raw + t4) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (arg) *reinterpret_cast<int4*>(arg + t4) = make_int4(nr, nr, nr, nr);
    *reinterpret_cast<int4*>(argpos + t4) = make_int4(nr, nr, nr, nr);
    return;
  }
  const int64_t gph = seg_graph ? seg_graph[s] : 0;
  const float4 h4 = *reinterpret_cast<const float4*>(raw + t4);
  const int4 p4 = *reinterpret_cast<const int4*>(argpos + t4);
  const float4 w4 = *reinterpret_cast<const float4*>(gnw + c0);
  const float4 am4 = *reinterpret_cast<const float4*>(am + gph * N + c0);
  const float4 sc4 = *reinterpret_cast<const float4*>(sc + gph * N + c0);
  const float4 bs4 = *reinterpret_cast<const float4*>(bs + c0);
  float hh[4] = {h4.x, h4.y, h4.z, h4.w};
  int pp[4] = {p4.x, p4.y, p4.z, p4.w};
  const float gw[4] = {w4.x, w4.y, w4.z, w4.w};
  const float aa[4] = {am4.x, am4.y, am4.z, am4.w}, ss[4] = {sc4.x, sc4.y, sc4.z, sc4.w};
  const float bb[4] = {bs4.x, bs4.y, bs4.z, bs4.w};
  float yy[4];
  int ar[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (gw[e] == 0.f) {
      const int c = c0 + e;
      pp[e] = a0;
      const int row = perm ? perm[a0] : a0;
      double acc = 0.0;
      for (int k = 0; k < K; ++k) {
        float v = IN16 ? __uint_as_float((unsigned)reinterpret_cast<const uint16_t*>(x)[(int64_t)row * K + k] << 16)
                       : x[(int64_t)row * K + k];
        v = fmaf(v - pam[gph * K + k], psc[gph * K + k], pbs[k]);
        v = v > 0.f ? v : v * pslope;
        float w = W[(size_t)c * K + k];
        if (bfw) {
          w = (float)(__bf16)w;
          v = (float)(__bf16)v;
        }
        acc += (double)v * (double)w;
      }
      hh[e] = (float)acc;
      raw[t4 + e] = hh[e];
      argpos[t4 + e] = pp[e];
    }
    // (no row won - every value NaN: the sentinel of an empty segment)
    // arg (the winner's ORIGINAL row: one scattered 4-byte read of perm per (segment, channel)) is
    // optional - the fused backward works on argpos (round 6: a training step passes NULL)
    if (arg) ar[e] = (pp[e] >= a0 && pp[e] < a1) ? (perm ? perm[pp[e]] : pp[e]) : (int32_t)n_rows;
    float y = fmaf(hh[e] - aa[e], ss[e], bb[e]);         // gn_apply_fwd_kernel's expression
    yy[e] = y > 0.f ? y : y * slope;
  }
  if (arg) *reinterpret_cast<int4*>(arg + t4) = make_int4(ar[0], ar[1], ar[2], ar[3]);
  *reinterpret_cast<float4*>(out + t4) = make_float4(yy[0], yy[1], yy[2], yy[3]);
}

// ---- backward, preparation ------------------------------------------------------------------------
// gm[s, c] = c1[g, c] gout[s, c] leaky'(y(raw[s, c])): the pool's gradient as the GraphNorm
// backward's c1 g term, one non-zero per (segment, channel) at row arg[s, c].
__global__ __launch_bounds__(256) void pool_bwd_gm_kernel(
    const float* __restrict__ gout, const float* __restrict__ raw,
    const int64_t* __restrict__ seg_graph, int64_t num_seg, int N, const float* __restrict__ am,
    const float* __restrict__ sc, const float* __restrict__ bs, float slope,
    const float* __restrict__ c1, float* __restrict__ gm) {
  const int64_t t4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (t4 >= num_seg * N) return;
  const int64_t s = t4 / N;
  const int c = (int)(t4 - s * N);
  const int64_t gph = seg_graph ? seg_graph[s] : 0;
  const float4 go = *reinterpret_cast<const float4*>(gout + t4);
  const float4 hv = *reinterpret_cast<const float4*>(raw + t4);
  const float4 a = *reinterpret_cast<const float4*>(am + gph * N + c);
  const float4 sv = *reinterpret_cast<const float4*>(sc + gph * N + c);
  const float4 b = *reinterpret_cast<const float4*>(bs + c);
  const float4 k1 = *reinterpret_cast<const float4*>(c1 + gph * N + c);
  const float gg[4] = {go.x, go.y, go.z, go.w}, hh[4] = {hv.x, hv.y, hv.z, hv.w};
  const float aa[4] = {a.x, a.y, a.z, a.w}, ss[4] = {sv.x, sv.y, sv.z, sv.w};
  const float bb[4] = {b.x, b.y, b.z, b.w}, q1[4] = {k1.x, k1.y, k1.z, k1.w};
  float o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float g_ = gg[e];
    if (slope != 1.f) {
      const float y = fmaf(hh[e] - aa[e], ss[e], bb[e]);
      g_ = (y > 0.f) ? g_ : g_ * slope;
    }
    o[e] = q1[e] * g_;
  }
  *reinterpret_cast<float4*>(gm + t4) = make_float4(o[0], o[1], o[2], o[3]);
}

// per graph: Bc = -c2, Ac = c2 am - c3;  M = W^T diag(Bc) W  [K, K] (symmetric),  c0 = Ac W  [K]
template <int K>
__global__ __launch_bounds__(256) void pool_bwd_coef_kernel(
    const float* __restrict__ W, int N, const float* __restrict__ am, const float* __restrict__ c2,
    const float* __restrict__ c3, float* __restrict__ M, float* __restrict__ c0) {
  const int b = blockIdx.x;
  extern __shared__ float sm[];                          // Bc [N] | Ac [N]
  float* Bc = sm;
  float* Ac = sm + N;
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    const float k2 = c2[b * N + i];
    Bc[i] = -k2;
    Ac[i] = fmaf(k2, am[b * N + i], -c3[b * N + i]);
  }
  __syncthreads();
  // (round 6: blockIdx.y slices the K K + K outputs - one workgroup per graph took 65 us for the
  //  4 160 128-term sums of the 64 -> 128 layer; same sums per output)
  for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < K * K + K; i += gridDim.y * blockDim.x) {
    double acc = 0.0;
    if (i < K * K) {
      const int j = i / K, k = i - j * K;
      for (int c = 0; c < N; ++c)
        acc += (double)W[(size_t)c * K + j] * (double)Bc[c] * (double)W[(size_t)c * K + k];
      M[(size_t)b * K * K + i] = (float)acc;
    } else {
      const int k = i - K * K;
      for (int c = 0; c < N; ++c) acc += (double)Ac[c] * (double)W[(size_t)c * K + k];
      c0[(size_t)b * K + k] = (float)acc;
    }
  }
}

// gW[c, k] += sum_b ( Bc[b, c] sum_j W[c, j] G[b][j, k] + Ac[b, c] sy[b][k] ): the dense part of
// the GraphNorm backward in closed form (f64), added to the sparse part the main kernel summed.
// BFW: the operands the forward saw (bf16 mode).
template <int K>
__global__ __launch_bounds__(256) void pool_bwd_gw_dense_kernel(
    const double* __restrict__ gram, int B, const float* __restrict__ W, int N, int bfw,
    const float* __restrict__ am, const float* __restrict__ c2, const float* __restrict__ c3,
    float* __restrict__ gW) {
  constexpr int GLEN = K * K + K + 1;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * K) return;
  const int c = t / K, k = t - c * K;
  double acc = 0.0;
  for (int b = 0; b < B; ++b) {
    const double* gr = gram + (size_t)b * GLEN;
    const double k2 = (double)c2[b * N + c];
    const double Ac = (double)fmaf(c2[b * N + c], am[b * N + c], -c3[b * N + c]);
    double t1 = 0.0;
    for (int j = 0; j < K; ++j) {
      float w = W[(size_t)c * K + j];
      if (bfw) w = (float)(__bf16)w;
      t1 += (double)w * gr[j * K + k];
    }
    acc += -k2 * t1 + Ac * gr[K * K + k];
  }
  gW[t] = (float)((double)gW[t] + acc);
}

// ---- backward, main kernel ------------------------------------------------------------------------
// Tiles walk the rows in the pool's CSR order.  S tile (f32, LDS) = gm where the winner's CSR
// position is this row, else 0: the tile is zero-filled and the (gm, argpos) rows of the segments
// it touches - one or two at 35 rows per segment - are SCATTERED into it (a lane owns N / 64
// channels; no search, no compare per (row, channel)).  gW += S^T y_prev (16x16x16, contraction =
// the tile's rows), gy = S W + y_prev M + c0 (16x16x32, W^T and M as split-bf16 rows in LDS), then
// - as every fused layer's backward - the rows leave as the gradient of the previous layer's
// normalised output and the two sums its GraphNorm backward needs (sum g', sum g' o') fall out of
// the same registers.  The gathered x rows of the NEXT tile are requested before this tile's
// GEMMs (one HBM round trip per tile, under the arithmetic).
// WAVE PAIRS: the 128 x 64 weight-gradient accumulators are 128 registers per lane - with them a
// wave has no room for the next tile's rows in flight and spills (measured: 78 % of the wave
// cycles parked on memory).  Two waves of a pair therefore walk the SAME tiles: each stages the
// tile into its own LDS buffers (the second read of a row is an L2 hit; no hand-shake between
// the two is needed) and owns HALF of the output - the gW rows of N / 2 channels and K / 2 columns
// of gy.  Per wave: 64 + 8 accumulators, ~150 registers, nothing spilled.
// LO: split operands (hi + lo, 3 products);  X16: xprev holds bf16 values.
template <int K, int N, bool LO, bool X16, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void bwd_pool_kernel(
    const float* __restrict__ gm, const int32_t* __restrict__ argpos,
    const int32_t* __restrict__ perm, const int32_t* __restrict__ pos_seg,
    const float* __restrict__ xprev, const float* __restrict__ pam,
    const float* __restrict__ psc, const float* __restrict__ pbs, float pslope,
    const float* __restrict__ W, const float* __restrict__ Mg, const float* __restrict__ c0g,
    float* __restrict__ gx, float* __restrict__ gw_partial, double* __restrict__ pstat_partial,
    FmlpRuns rt) {
  constexpr int KB = K / 16, NBK = N / 16, NS = N / 32, KS = K / 32;
  constexpr int NBH = NBK / 2, KBH = KB / 2;             // a wave's half of the outputs
  constexpr int LDG = N + 4, LDX = K + 4;
  constexpr int LDT = N + SPT_FPOOL_WPAD, LDM = K + SPT_FPOOL_WPAD;   // = 16 (mod 32): see fwd_pool_kernel's LDW
  static_assert((LDT % 32 == 16 && LDM % 32 == 16) || SPT_FPOOL_WPAD != 16, "conflict-free B fragments");
  constexpr int CPL = N / 64;                            // channels a lane scatters per segment
  static_assert(K % 32 == 0 && N % 64 == 0 && NW % 2 == 0, "shape");
  __shared__ __attribute__((aligned(16))) float g_lds[NW][TR * LDG];   // S tile
  __shared__ __attribute__((aligned(16))) float x_lds[NW][TR * LDX];   // RAW xprev tile
  __shared__ __attribute__((aligned(16))) __bf16 wt_hi[K * LDT];       // wt[k][n] = W[n][k]
  __shared__ __attribute__((aligned(16))) __bf16 wt_lo[LO ? K * LDT : 8];
  __shared__ __attribute__((aligned(16))) __bf16 mt_hi[K * LDM];       // mt[k][j] = M[j][k] (= M[k][j])
  __shared__ __attribute__((aligned(16))) __bf16 mt_lo[LO ? K * LDM : 8];
  __shared__ __attribute__((aligned(16))) float pt[3 * K];             // previous norm: am | sc | bs
  __shared__ __attribute__((aligned(16))) float c0l[K];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hf = wid & 1;                                // which half of the outputs
  const int nb0 = hf * NBH, kb0 = hf * KBH;
  const int g = lane >> 4, c = lane & 15;
  const int run = blockIdx.y, gph = rt.g[run];
  const int64_t r0 = rt.r0[run], r1 = rt.r1[run];
  pam += (size_t)gph * K;
  psc += (size_t)gph * K;
  Mg += (size_t)gph * K * K;
  c0g += (size_t)gph * K;
  // weight-gradient tables: ONE per workgroup (round 6, summed over the workgroup's wave pairs in
  // the epilogue) in front, one scratch record per wave pair behind them (the loop's dummy stores)
  constexpr int SCR = 4 * (K / 32) * 64;                 // floats a wave's dummy stores touch
  float* gw_scratch = gw_partial + (size_t)gridDim.y * gridDim.x * N * K +
                      (size_t)run * gridDim.x * (NW / 2) * 2 * SCR;
  gw_partial += (size_t)run * gridDim.x * N * K;
  pstat_partial += (size_t)run * gridDim.x * (NW / 2) * (2 * K + 1);
  float* gl = g_lds[wid];
  float* xl = x_lds[wid];
  for (int i = threadIdx.x; i < K * N; i += NW * 64) {
    const int k = i / N, n = i - k * N;
    const float w = W[(size_t)n * K + k];
    const __bf16 hh = (__bf16)w;
    wt_hi[k * LDT + n] = hh;
    if constexpr (LO) wt_lo[k * LDT + n] = (__bf16)(w - (float)hh);
  }
  for (int i = threadIdx.x; i < K * K; i += NW * 64) {
    const int k = i / K, j = i - k * K;
    const float m = Mg[(size_t)j * K + k];
    const __bf16 hh = (__bf16)m;
    mt_hi[k * LDM + j] = hh;
    if constexpr (LO) mt_lo[k * LDM + j] = (__bf16)(m - (float)hh);
  }
  for (int i = threadIdx.x; i < K; i += NW * 64) {
    pt[i] = pam[i];
    pt[K + i] = psc[i];
    pt[2 * K + i] = pbs[i];
    c0l[i] = c0g[i];
  }
  __syncthreads();

  f32x4 C3[NBH][KB];       // C3[j][kb][r] = gW[16 (nb0 + j) + 4 g + r][16 kb + c]
#pragma unroll
  for (int nb = 0; nb < NBH; ++nb)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) C3[nb][kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  double p1[KBH], p2[KBH];
#pragma unroll
  for (int kb = 0; kb < KBH; ++kb) p1[kb] = p2[kb] = 0.0;

  const int64_t ntiles = (r1 - r0 + TR - 1) / TR;
  const int64_t pair = (int64_t)blockIdx.x * (NW / 2) + (wid >> 1);
  const int64_t npairs = (int64_t)gridDim.x * (NW / 2);
  auto cnt_of = [&](int64_t t) {
    const int64_t row0 = r0 + t * TR;
    return (t < ntiles) ? (int)((r1 - row0) < TR ? (r1 - row0) : TR) : 0;
  };
  // ids of tile t's 16 positions - UNCONDITIONAL loads (see load_rows): a tile past the end reads
  // the run's last tile again, positions past r1 its last position; nothing of it is used
  auto load_ids = [&](int64_t t, int& rid, int& sg) {
    const int64_t tt = t < ntiles ? t : ntiles - 1;
    int64_t q = r0 + tt * TR + (lane & (TR - 1));
    q = q < r1 - 1 ? q : r1 - 1;
    rid = perm ? perm[q] : (int)q;
    sg = pos_seg[q];
  };
  // (gm, argpos) of the lane's CPL channels of segment sgm
  auto load_seg = [&](int sgm, int (&ap)[CPL], float (&gv)[CPL]) {
    const size_t o = (size_t)sgm * N + CPL * lane;
    if constexpr (CPL == 2) {
      const int2 a2 = *reinterpret_cast<const int2*>(argpos + o);
      const float2 g2 = *reinterpret_cast<const float2*>(gm + o);
      ap[0] = a2.x; ap[1] = a2.y;
      gv[0] = g2.x; gv[1] = g2.y;
    } else {
      ap[0] = argpos[o];
      gv[0] = gm[o];
    }
  };
  // winners of one segment whose tile rows are [lo, hi): S[row][channel] = gm.  Branch-free: a
  // lane whose winner lies outside writes a pad column of row 0 instead (never read)
  auto scatter = [&](const int (&ap)[CPL], const float (&gv)[CPL], int row0i, int lo, int hi) {
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      const int rr = ap[q] - row0i;
      const bool ok = rr >= lo && rr < hi;
      gl[ok ? rr * LDG + CPL * lane + q : N + q] = gv[q];
    }
  };
  if (ntiles > 0) {                             // (an empty run still writes its zero records below)
  int rid_l, seg_l, rid_n, seg_n;
  load_ids(pair, rid_l, seg_l);
  typename RowChunk<X16>::type xv[TR * (K / 4) / 64];   // raw rows of the tile about to be staged
  load_rows<K, X16>(xprev, rid_l, xv, lane);
  // the tile's FIRST and LAST segment (at 35 rows per segment a tile touches one or two): their
  // (gm, argpos) rows travel a tile ahead like the x rows; further segments are fetched in place
  int apA[CPL], apB[CPL];
  float gvA[CPL], gvB[CPL];
  {
    const int c0 = cnt_of(pair < ntiles ? pair : ntiles - 1);
    load_seg(__builtin_amdgcn_readlane(seg_l, 0), apA, gvA);
    load_seg(__builtin_amdgcn_readlane(seg_l, c0 - 1), apB, gvB);
  }
  load_ids(pair + npairs, rid_n, seg_n);
  // The loop's memory operations are waited for by COUNT (s_waitcnt vmcnt(n): at most n younger
  // operations outstanding), and the compiler merges the counts of the loop's entry with those of
  // its back edge: entering with the same sequence in flight as an iteration leaves behind - the
  // requests above, then as many stores as a tile's gx - keeps every count in the loop exact
  // (otherwise each wait for a prefetched row also drains the previous tile's 4 KBH stores).
  // The stores go to this pair's own scratch record behind the weight-gradient tables.
  {
    float* scratch = gw_scratch + ((size_t)pair * 2 + hf) * SCR;
#pragma unroll
    for (int i = 0; i < 4 * KBH; ++i) scratch[lane + 64 * i] = 0.f;
  }
  // One tile.  FULL (every tile but a run's last): 16 rows, so no per-row condition anywhere -
  // in particular the gx stores are unconditional, and the compiler's count of memory operations
  // in flight stays EXACT across the loop: the waits for the rows requested a tile ahead then do
  // not drain the stores issued after them (vmcnt counts in order; a store under a per-lane
  // condition is assumed not to have been issued, and every later wait over-waits by that much).
  auto tile = [&](auto full_c, int64_t t) {
    constexpr bool FULL = decltype(full_c)::value;
    const int64_t row0 = r0 + t * TR;
    const int cnt = FULL ? TR : cnt_of(t);
    wave_sync_lds();
    // ---- RAW xprev tile (gathered rows; rows >= cnt zero).  FIRST: the rows were requested
    // before everything else of this tile, and the rare 3+-segment loop below contains loads of
    // its own - whatever is waited for behind it is waited for with vmcnt(0)
    store_rows<K, LDX, false, X16>(xv, cnt, nullptr, 1.f, xl, lane);
    // ---- S tile: zero, then the winners of the tile's segments scattered into it ---------------
    {
      constexpr int NZ = TR * (N / 4) / 64;
#pragma unroll
      for (int j = 0; j < NZ; ++j) {
        const int q = lane + 64 * j, rr = q / (N / 4), n = (q - rr * (N / 4)) << 2;
        *reinterpret_cast<float4*>(gl + rr * LDG + n) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      const int sA = __builtin_amdgcn_readlane(seg_l, 0);
      const int sB = __builtin_amdgcn_readlane(seg_l, cnt - 1);
      const uint64_t notA = __ballot(lane < cnt && seg_l != sA);
      const int eA = notA ? (int)__builtin_ctzll(notA) : cnt;       // rows [0, eA): segment sA
      const uint64_t isB = __ballot(lane < cnt && seg_l == sB);
      const int bB = (int)__builtin_ctzll(isB);                     // rows [bB, cnt): segment sB
      scatter(apA, gvA, (int)row0, 0, eA);
      if (sB != sA) scatter(apB, gvB, (int)row0, bB, cnt);
      int row = eA;
      while (row < bB) {                                           // 3+ segments in the tile (rare)
        const int sgm = __builtin_amdgcn_readlane(seg_l, row);
        const uint64_t diff = __ballot(lane < cnt && lane > row && seg_l != sgm);
        const int e = diff ? (int)__builtin_ctzll(diff) : cnt;
        int ap[CPL];
        float gv[CPL];
        load_seg(sgm, ap, gv);
        scatter(ap, gv, (int)row0, row, e);
        row = e;
      }
    }
    const int rid_cur = rid_l;                 // the gx scatter below needs this tile's row ids
    // everything the NEXT tile needs from memory is requested here, before this tile's GEMMs;
    // the id registers rotate HERE, on values that have arrived (a rotation at the loop's end
    // would have to wait for the ids just requested)
    load_rows<K, X16>(xprev, rid_n, xv, lane);
    {
      const int64_t tn = t + npairs < ntiles ? t + npairs : ntiles - 1;
      load_seg(__builtin_amdgcn_readlane(seg_n, 0), apA, gvA);
      load_seg(__builtin_amdgcn_readlane(seg_n, cnt_of(tn) - 1), apB, gvB);
    }
    rid_l = rid_n;
    seg_l = seg_n;
    load_ids(t + 2 * npairs, rid_n, seg_n);
    wave_sync_lds();
    // y_prev of one raw value (rows >= cnt: 0)
    auto ynorm = [&](float v, int k, bool ok) {
      v = fmaf(v - pt[k], pt[K + k], pt[2 * K + k]);
      v = (v > 0.f) ? v : v * pslope;
      return ok ? v : 0.f;
    };
    // ---- gW[this half's rows] += S^T y_prev -------------------------------------------------------
    if constexpr (!(SPT_FPOOL_SKIP & 8)) {
      bf16x4 Xh[KB], Xl[KB];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int k = 16 * kb + c;
        float xr4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) xr4[r] = ynorm(xl[(4 * g + r) * LDX + k], k, 4 * g + r < cnt);
        if constexpr (LO) {
          split2<4>(xr4, Xh[kb], Xl[kb]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) Xh[kb][r] = (__bf16)xr4[r];
        }
      }
#pragma unroll
      for (int nb = 0; nb < NBH; ++nb) {
        float sv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) sv[r] = gl[(4 * g + r) * LDG + 16 * (nb0 + nb) + c];
        bf16x4 sh, sl;
        if constexpr (LO) {
          split2<4>(sv, sh, sl);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) sh[r] = (__bf16)sv[r];
        }
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          f32x4 acc = C3[nb][kb];
          if constexpr (LO) {
            acc = mfma16(sl, Xh[kb], acc);
            acc = mfma16(sh, Xl[kb], acc);
          }
          C3[nb][kb] = mfma16(sh, Xh[kb], acc);
        }
      }
    }
    // ---- gy[this half's columns] = c0 + S W + y_prev M (+ statistics for the previous norm) ------
    {
      f32x4 CX[KBH];
#pragma unroll
      for (int kb = 0; kb < KBH; ++kb) {
        const float z = c0l[16 * (kb0 + kb) + c];
        CX[kb] = (f32x4){z, z, z, z};
      }
      // (round 6: the B fragments - W^T planes, then M planes - are requested LA steps ahead of
      //  their MFMAs, as in the forward: one list of NS KBH + KS KBH (step, column block) pairs)
      constexpr int NPLB = LO ? 2 : 1, LAB = 2;
      constexpr int NST1 = NS * KBH, NST = NST1 + KS * KBH;
      bf16x8 bf[LAB + 1][NPLB];
      auto ldb = [&](int step, bf16x8 (&b)[NPLB]) {
        if (step < NST1) {
          const int sg = step / KBH, kb = step - sg * KBH;
          const int wo = (16 * (kb0 + kb) + c) * LDT + 32 * sg + 8 * g;
          b[0] = *reinterpret_cast<const bf16x8*>(wt_hi + wo);
          if constexpr (LO) b[NPLB - 1] = *reinterpret_cast<const bf16x8*>(wt_lo + wo);
        } else {
          const int st = step - NST1, ks = st / KBH, kb = st - ks * KBH;
          const int mo = (16 * (kb0 + kb) + c) * LDM + 32 * ks + 8 * g;
          b[0] = *reinterpret_cast<const bf16x8*>(mt_hi + mo);
          if constexpr (LO) b[NPLB - 1] = *reinterpret_cast<const bf16x8*>(mt_lo + mo);
        }
      };
#pragma unroll
      for (int i = 0; i < LAB; ++i) ldb(i, bf[i]);
      bf16x8 ah, alo;
#pragma unroll
      for (int step = 0; step < ((SPT_FPOOL_SKIP & 16) ? 0 : NST); ++step) {
        const bool first = step < NST1;
        const int sub = first ? step : step - NST1;
        const int blk = sub / KBH, kb = sub - blk * KBH;    // blk = sg (S W) or ks (y_prev M)
        if (kb == 0) {
          float av[8];
          if (first) {
            const float4 a0 = *reinterpret_cast<const float4*>(gl + c * LDG + 32 * blk + 8 * g);
            const float4 a1 = *reinterpret_cast<const float4*>(gl + c * LDG + 32 * blk + 8 * g + 4);
            av[0] = a0.x; av[1] = a0.y; av[2] = a0.z; av[3] = a0.w;
            av[4] = a1.x; av[5] = a1.y; av[6] = a1.z; av[7] = a1.w;
          } else {
            const float4 a0 = *reinterpret_cast<const float4*>(xl + c * LDX + 32 * blk + 8 * g);
            const float4 a1 = *reinterpret_cast<const float4*>(xl + c * LDX + 32 * blk + 8 * g + 4);
            const float rv[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) av[i] = ynorm(rv[i], 32 * blk + 8 * g + i, c < cnt);
          }
          if constexpr (LO) {
            split2<8>(av, ah, alo);
          } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) ah[i] = (__bf16)av[i];
          }
        }
        if (step + LAB < NST) ldb(step + LAB, bf[(step + LAB) % (LAB + 1)]);
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 (&b)[NPLB] = bf[step % (LAB + 1)];
        f32x4 acc = CX[kb];
        if constexpr (LO) {
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo, b[0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, b[NPLB - 1], acc, 0, 0, 0);
        }
        CX[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, b[0], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = 4 * g + r;
        const int64_t orow = (int64_t)__shfl(rid_cur, rr, 64);
        if (FULL || rr < cnt) {
#pragma unroll
          for (int kb = 0; kb < KBH; ++kb) {
            const int k = 16 * (kb0 + kb) + c;
            const float v = CX[kb][r];
            gx[orow * K + k] = v;
            const float o = xl[rr * LDX + k] - pt[k];
            float gg = v;
            if (pslope != 1.f) {
              const float y = fmaf(o, pt[K + k], pt[2 * K + k]);
              gg = (y > 0.f) ? gg : gg * pslope;
            }
            if constexpr (!(SPT_FPOOL_SKIP & 32)) {
              p1[kb] += (double)gg;
              p2[kb] += (double)gg * (double)o;
            }
          }
        }
      }
    }
  };
  const int64_t nfull = (r1 - r0) / TR;         // complete tiles; at most one partial tile follows
  int64_t t = pair;
  for (; t < nfull; t += npairs) tile(std::true_type{}, t);
  if (t < ntiles) tile(std::false_type{}, t);
  }
  double* pp = pstat_partial + (size_t)pair * (2 * K + 1);
#pragma unroll
  for (int kb = 0; kb < KBH; ++kb) {
    const double a = xg_sum_d(p1[kb]), b = xg_sum_d(p2[kb]);
    const int k = 16 * (kb0 + kb) + c;
    if (g == 0) {
      pp[k] = a;
      pp[K + k] = b;
    }
  }
  if (lane == 0 && hf == 0) pp[2 * K] = (pair == 0) ? (double)(r1 - r0) : 0.0;
  // gW: one table per WORKGROUP - the pairs 1 .. NW / 2 - 1 hand their tables (each wave its rows)
  // to pair 0 through the free S-tile buffer, one pair after the other (plain stores, plain loads +
  // adds, fixed order); statistics: one record per pair, each wave its columns
  {
    constexpr int LDR = K + 4;                    // row stride = 4 (mod 16) floats: no bank conflicts
    static_assert(N * LDR <= NW * TR * LDG, "the table fits the S-tile buffers");
    const int pr = wid >> 1;
    __syncthreads();                              // every wave is through its tiles
    float* red = &g_lds[0][0];
#pragma unroll 1
    for (int p = 1; p < NW / 2; ++p) {
      if (pr == p) {
#pragma unroll
        for (int nb = 0; nb < NBH; ++nb)
#pragma unroll
          for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              red[(16 * (nb0 + nb) + 4 * g + r) * LDR + 16 * kb + c] = C3[nb][kb][r];
      }
      __syncthreads();
      if (pr == 0) {
#pragma unroll
        for (int nb = 0; nb < NBH; ++nb)
#pragma unroll
          for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              C3[nb][kb][r] += red[(16 * (nb0 + nb) + 4 * g + r) * LDR + 16 * kb + c];
      }
      __syncthreads();
    }
    if (pr == 0) {
      float* gwp = gw_partial + (size_t)blockIdx.x * N * K;
#pragma unroll
      for (int nb = 0; nb < NBH; ++nb)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            gwp[(size_t)(16 * (nb0 + nb) + 4 * g + r) * K + 16 * kb + c] = C3[nb][kb][r];
    }
  }
}

}  // namespace fpool

// ---- launchers (called from fused_mlp.hip's C entry points) -------------------------------------
// shapes built: the top layers that sit in front of a max-pool (64 -> 128 semantic, 64 -> 64
// panoptic, 32 -> 64 small MLPs)
#define SPT_FPOOL_SHAPES(X) X(64, 128) X(64, 64) X(32, 64)

bool fpool_supported(int K, int N) {
#define X(k, n) if (K == k && N == n) return true;
  SPT_FPOOL_SHAPES(X)
#undef X
  return false;
}
int fpool_gram_len(int K) { return K * K + K + 1; }

// workgroups of the forward per run (its partial Gram records)
int fpool_fwd_blocks(int64_t max_rows, int nruns) {
  int64_t blocks = (max_rows + fpool::TR * fpool::NWF * 4 - 1) / (fpool::TR * fpool::NWF * 4);
  int64_t cap = 256 / nruns;
  if (cap < 1) cap = 1;
  if (blocks > cap) blocks = cap;
  return (int)(blocks < 1 ? 1 : blocks);
}

// prec: 3 = f32-exact (3-way split), 2 = 2-way split, 1 = plain bf16.  Returns blocks per run.
int fpool_fwd_launch(int prec, bool in16, const float* x, const int32_t* perm, const int32_t* pos_seg,
                     const int32_t* rowptr, const FmlpRuns& rt, int64_t max_rows, int K, int N,
                     const float* W, const float* gnw, const float* pam, const float* psc,
                     const float* pbs, float pslope, float* raw, int32_t* argpos, double* partial,
                     hipStream_t stream) {
  const int gx_ = fpool_fwd_blocks(max_rows, rt.n);
  const dim3 grid((unsigned)gx_, (unsigned)rt.n);
#define X(k, n)                                                                                   \
  if (K == k && N == n) {                                                                         \
    if (prec == 3 && !in16)                                                                       \
      fpool::fwd_pool_kernel<k, n, 3, false><<<grid, fpool::NWF * 64, 0, stream>>>(               \
          x, perm, pos_seg, rowptr, W, gnw, pam, psc, pbs, pslope, raw, argpos, partial, rt);        \
    else if (prec == 2 && !in16)                                                                  \
      fpool::fwd_pool_kernel<k, n, 2, false><<<grid, fpool::NWF * 64, 0, stream>>>(               \
          x, perm, pos_seg, rowptr, W, gnw, pam, psc, pbs, pslope, raw, argpos, partial, rt);        \
    else if (prec == 1 && !in16)                                                                  \
      fpool::fwd_pool_kernel<k, n, 1, false><<<grid, fpool::NWF * 64, 0, stream>>>(               \
          x, perm, pos_seg, rowptr, W, gnw, pam, psc, pbs, pslope, raw, argpos, partial, rt);        \
    else if (prec == 1 && in16)                                                                   \
      fpool::fwd_pool_kernel<k, n, 1, true><<<grid, fpool::NWF * 64, 0, stream>>>(                \
          x, perm, pos_seg, rowptr, W, gnw, pam, psc, pbs, pslope, raw, argpos, partial, rt);        \
    else                                                                                          \
      return -1;                                                                                  \
  }
  SPT_FPOOL_SHAPES(X)
#undef X
  return gx_;
}

void fpool_tables_launch(int K, const double* gram, int B, const float* W, int N, int bfw,
                         const float* weight, const float* mean_scale, float eps, double* total,
                         float* mean, float* rstd, float* am, float* scale, hipStream_t stream) {
  if (K == 64)
    fpool::gram_tables_kernel<64><<<dim3((N + 3) / 4, B), 256, 0, stream>>>(gram, W, N, bfw, weight, mean_scale, eps,
                                                         total, mean, rstd, am, scale);
  else
    fpool::gram_tables_kernel<32><<<dim3((N + 7) / 8, B), 256, 0, stream>>>(gram, W, N, bfw, weight, mean_scale, eps,
                                                         total, mean, rstd, am, scale);
}

void fpool_apply_launch(int K, bool in16, const int32_t* rowptr, const int32_t* perm,
                        const int64_t* seg_graph, int64_t num_seg, int N, int64_t n_rows,
                        const float* am, const float* sc, const float* bs, float slope,
                        const float* gnw, const float* x, const float* W, const float* pam,
                        const float* psc, const float* pbs, float pslope, int bfw, float* raw,
                        int32_t* argpos, int32_t* arg, float* out, hipStream_t stream) {
  const int64_t total = num_seg * N;
  if (total <= 0) return;
  const int grid = (int)ceil_div(total, 1024);            // four channels per thread
#define XA(k, i16)                                                                                 \
  fpool::pool_apply_kernel<k, i16><<<grid, 256, 0, stream>>>(rowptr, perm, seg_graph, num_seg, N,  \
                                                             n_rows, am, sc, bs, slope, gnw, x, W, \
                                                             pam, psc, pbs, pslope, bfw, raw, argpos, arg, out)
  if (K == 64) { if (in16) XA(64, true); else XA(64, false); }
  else { if (in16) XA(32, true); else XA(32, false); }
#undef XA
}

// backward: gm, coefficient matrices, main kernel.  Returns wave records per run (gW / statistics
// partial tables), or -1 for an unbuilt variant.
int fpool_bwd_launch(bool lo, bool x16, const float* gout, const float* raw, const int32_t* argpos,
                     const int32_t* perm, const int32_t* pos_seg, const int64_t* seg_graph,
                     int64_t num_seg, const FmlpRuns& rt, int64_t max_rows, int num_graphs, int K,
                     int N, const float* am, const float* sc, const float* bs, float slope,
                     const float* c1, const float* c2, const float* c3, const float* xprev,
                     const float* pam, const float* psc, const float* pbs, float pslope,
                     const float* W, float* gm, float* Mbuf, float* c0buf, float* gx,
                     float* gw_partial, double* pstat_partial, int max_waves, hipStream_t stream,
                     int* gw_tabs) {
  if (num_seg > 0)
    fpool::pool_bwd_gm_kernel<<<(int)ceil_div(num_seg * N / 4, 256), 256, 0, stream>>>(
        gout, raw, seg_graph, num_seg, N, am, sc, bs, slope, c1, gm);
  if (K == 64)
    fpool::pool_bwd_coef_kernel<64><<<dim3(num_graphs, (64 * 64 + 64 + 255) / 256), 256, 2 * N * sizeof(float), stream>>>(W, N, am, c2, c3, Mbuf, c0buf);
  else
    fpool::pool_bwd_coef_kernel<32><<<dim3(num_graphs, (32 * 32 + 32 + 255) / 256), 256, 2 * N * sizeof(float), stream>>>(W, N, am, c2, c3, Mbuf, c0buf);
  constexpr int NW = 8;
  const int64_t tiles = (max_rows + fpool::TR - 1) / fpool::TR;
  int64_t blocks = (tiles + NW / 2 - 1) / (NW / 2);         // a tile is walked by a pair of waves
  int64_t cap = (K * N > 4096) ? 256 : 512;               // 64 -> 128: one 8-wave workgroup per CU
  // the workspace (max_waves tables) holds one table per workgroup and a 4 KB scratch record per
  // wave pair behind them: the old bound of one table per pair still covers both
  const int64_t cap_ws = max_waves / ((NW / 2) * rt.n);
  if (cap > cap_ws) cap = cap_ws;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  const dim3 grid((unsigned)blocks, (unsigned)rt.n);
#define X(k, n)                                                                                   \
  if (K == k && N == n) {                                                                         \
    if (lo && !x16)                                                                               \
      fpool::bwd_pool_kernel<k, n, true, false, NW><<<grid, NW * 64, 0, stream>>>(                \
          gm, argpos, perm, pos_seg, xprev, pam, psc, pbs, pslope, W, Mbuf, c0buf, gx, gw_partial,   \
          pstat_partial, rt);                                                                     \
    else if (!lo && !x16)                                                                         \
      fpool::bwd_pool_kernel<k, n, false, false, NW><<<grid, NW * 64, 0, stream>>>(               \
          gm, argpos, perm, pos_seg, xprev, pam, psc, pbs, pslope, W, Mbuf, c0buf, gx, gw_partial,   \
          pstat_partial, rt);                                                                     \
    else if (!lo && x16)                                                                          \
      fpool::bwd_pool_kernel<k, n, false, true, NW><<<grid, NW * 64, 0, stream>>>(                \
          gm, argpos, perm, pos_seg, xprev, pam, psc, pbs, pslope, W, Mbuf, c0buf, gx, gw_partial,   \
          pstat_partial, rt);                                                                     \
    else                                                                                          \
      return -1;                                                                                  \
  }
  SPT_FPOOL_SHAPES(X)
#undef X
  if (gw_tabs) *gw_tabs = (int)blocks;                     // weight-gradient tables per run
  return (int)blocks * (NW / 2);                           // statistics records per run
}

void fpool_gw_dense_launch(int K, const double* gram, int B, const float* W, int N, int bfw,
                           const float* am, const float* c2, const float* c3, float* gW,
                           hipStream_t stream) {
  const int grid = (N * K + 255) / 256;
  if (K == 64)
    fpool::pool_bwd_gw_dense_kernel<64><<<grid, 256, 0, stream>>>(gram, B, W, N, bfw, am, c2, c3, gW);
  else
    fpool::pool_bwd_gw_dense_kernel<32><<<grid, 256, 0, stream>>>(gram, B, W, N, bfw, am, c2, c3, gW);
}

}  // namespace spt
