// Segment reduce over a CSR view (sum / mean / min / max, optional arg) and its
// backward, plus the row gather that is both IndexUnpool's forward and the
// backward of a segment sum.  gfx950: wave64, 16-byte lane accesses, one
// sub-wave lane group per segment, shuffle tree across the group, no atomics.
//
// Reference semantics restated from torch_scatter (un-vendored; call sites
// src/nn/pool.py:61-82, src/nn/norm.py:118-126, src/nn/unpool.py:12-13):
// empty segment -> 0 (arg = n), mean divides by max(count,1), min/max gradient
// goes to the arg element only, ties -> first occurrence (CPU kernel rule).
#include <math.h>
#include <stdlib.h>

#include "common.hpp"

namespace spt {

template <int VEC>
struct Vec;
template <>
struct Vec<1> {
  float v[1];
};
template <>
struct Vec<2> {
  float v[2];
};
template <>
struct Vec<4> {
  float v[4];
};

// measured on scene S (profiles/r01q_segcsr_variants.txt): 8 rows in flight per lane slot
// and non-temporal row loads (each child row is read exactly once) = +9 % over 4 / plain
#ifndef SPT_SEG_UNR
#define SPT_SEG_UNR 8
#endif
#ifndef SPT_SEG_NT
#define SPT_SEG_NT 1
#endif
#ifndef SPT_SEG_PREFETCH
#define SPT_SEG_PREFETCH 0
#endif
// grid cap 256 CUs x 64 workgroups (measured: x16 per CU = 61 %, x64 = 69 %, x256 = 66 % of 8 TB/s)
#ifndef SPT_SEG_GRIDCAP_MUL
#define SPT_SEG_GRIDCAP_MUL 4
#endif
#ifndef SPT_SEG_RPG_BIAS
#define SPT_SEG_RPG_BIAS 0
#endif
// row ids of the next chunk in flight under the rows of the current one (see the kernel)
#ifndef SPT_SEG_PIPE
#define SPT_SEG_PIPE 0
#endif

template <int VEC>
__device__ __forceinline__ Vec<VEC> ldv(const float* __restrict__ p) {
  Vec<VEC> r;
  if constexpr (VEC == 4) {
#if SPT_SEG_NT
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f t4 = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    const float4 t = make_float4(t4[0], t4[1], t4[2], t4[3]);
#else
    const float4 t = *reinterpret_cast<const float4*>(p);
#endif
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
  } else if constexpr (VEC == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    r.v[0] = t.x; r.v[1] = t.y;
  } else {
    r.v[0] = *p;
  }
  return r;
}

template <int VEC>
__device__ __forceinline__ void stv(float* __restrict__ p, const Vec<VEC>& r) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
  } else if constexpr (VEC == 2) {
    *reinterpret_cast<float2*>(p) = make_float2(r.v[0], r.v[1]);
  } else {
    *p = r.v[0];
  }
}

// streaming store (rows written once, consumed by a later kernel)
template <int VEC>
__device__ __forceinline__ void stv_nt(float* __restrict__ p, const Vec<VEC>& r) {
  if constexpr (VEC == 4) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f t = {r.v[0], r.v[1], r.v[2], r.v[3]};
    __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p));
  } else {
    stv<VEC>(p, r);
  }
}

template <int VEC>
__device__ __forceinline__ void sti(int32_t* __restrict__ p, const int32_t (&a)[VEC]) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<int4*>(p) = make_int4(a[0], a[1], a[2], a[3]);
  } else if constexpr (VEC == 2) {
    *reinterpret_cast<int2*>(p) = make_int2(a[0], a[1]);
  } else {
    *p = a[0];
  }
}

// SUM / MEAN accumulate in f64: a segment of any length then sums to the
// correctly rounded f32 result whatever the lane-group shape (deterministic,
// and never worse than the f32 atomics of the reference).  The kernel stays
// HBM-bound: one v_cvt + one v_add_f64 per loaded float.
template <int OP>
struct AccT {
  using type = float;
};
template <>
struct AccT<SPT_SUM> {
  using type = double;
};
template <>
struct AccT<SPT_MEAN> {
  using type = double;
};

template <int OP>
__device__ __forceinline__ typename AccT<OP>::type op_identity() {
  if constexpr (OP == SPT_MIN) return INFINITY;
  if constexpr (OP == SPT_MAX) return -INFINITY;
  return 0;
}

// combine (v, r) into (acc, accr); r = original row index of v
template <int OP, bool ARG, typename A>
__device__ __forceinline__ void combine(A& acc, int32_t& accr, A v, int32_t r) {
  if constexpr (OP == SPT_SUM || OP == SPT_MEAN) {
    acc += v;
  } else if constexpr (OP == SPT_MAX) {
    if constexpr (ARG) {
      const bool take = (v > acc) || (v == acc && r < accr);
      acc = take ? v : acc;
      accr = take ? r : accr;
    } else {
      acc = (v > acc) ? v : acc;
    }
  } else {
    if constexpr (ARG) {
      const bool take = (v < acc) || (v == acc && r < accr);
      acc = take ? v : acc;
      accr = take ? r : accr;
    } else {
      acc = (v < acc) ? v : acc;
    }
  }
}

// Optional per-element map applied to every loaded value BEFORE the reduction:
// y = leaky(sc[g][c] (x - am[g][c]) + bs[c]), g = graph of the segment.  This is the
// GraphNorm-apply + LeakyReLU that ends the point MLP: pooling its RAW output this way
// removes the [15 M, 128] apply pass (read + write of 7.7 GB each) in front of the pool.
struct Affine {
  const float* am;           // [B, c] alpha * mean
  const float* sc;           // [B, c] weight * rstd
  const float* bs;           // [c]
  const int64_t* seg_graph;  // [num_seg] graph of each segment, or null (one graph)
  float slope;
};

// One lane group of G = LPR*RPG lanes per segment: LPR lanes span the channels
// of a row (VEC floats each), RPG rows are in flight side by side, UNR deep.
template <int OP, int VEC, bool ARG, bool AFF = false>
__global__ __launch_bounds__(256) void segcsr_reduce_kernel(
    const float* __restrict__ x, const int32_t* __restrict__ perm,
    const int32_t* __restrict__ rowptr, int64_t n, int64_t num_seg, int c,
    int lpr_log2, int rpg_log2, float* __restrict__ out,
    int32_t* __restrict__ arg, Affine af = Affine{}) {
  constexpr int UNR = SPT_SEG_UNR;
  const int lane = threadIdx.x & 63;
  const int g_log2 = lpr_log2 + rpg_log2;
  const int lpr = 1 << lpr_log2;
  const int rpg = 1 << rpg_log2;
  const int spw = 64 >> g_log2;  // segments per wave
  const int slot = lane >> g_log2;
  const int lg = lane & ((1 << g_log2) - 1);
  const int rsub = lg >> lpr_log2;
  const int lr = lg & (lpr - 1);
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int ctile = lpr * VEC;

#if SPT_SEG_PREFETCH
  int nstart = 0, nend = 0;
  {
    const int64_t s0 = wave * spw + slot;
    if (s0 < num_seg) {
      nstart = rowptr[s0];
      nend = rowptr[s0 + 1];
    }
  }
#endif
  for (int64_t sbase = wave * spw; sbase < num_seg; sbase += nwaves * spw) {
    const int64_t s = sbase + slot;
    const bool sv = s < num_seg;
#if SPT_SEG_PREFETCH
    const int start = nstart, end = nend;
    {  // the next segment's range is requested a whole segment ahead
      const int64_t sn = s + nwaves * spw;
      nstart = nend = 0;
      if (sn < num_seg) {
        nstart = rowptr[sn];
        nend = rowptr[sn + 1];
      }
    }
#else
    const int start = sv ? rowptr[s] : 0;
    const int end = sv ? rowptr[s + 1] : 0;
#endif
    for (int cb = 0; cb < c; cb += ctile) {
      const int c0 = cb + lr * VEC;
      const bool cv = c0 < c;
      using A = typename AccT<OP>::type;
      A acc[VEC];
      int32_t accr[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        acc[k] = op_identity<OP>();
        accr[k] = 0x7fffffff;
      }
      float t_am[VEC], t_sc[VEC], t_bs[VEC];
      const bool leaky01 = AFF && af.slope >= 0.f && af.slope <= 1.f;
      if constexpr (AFF) {
        const int64_t gph = (sv && af.seg_graph) ? af.seg_graph[s] : 0;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          t_am[k] = cv ? af.am[gph * c + c0 + k] : 0.f;
          t_sc[k] = cv ? af.sc[gph * c + c0 + k] : 0.f;
          t_bs[k] = cv ? af.bs[c0 + k] : 0.f;
        }
      }
#if SPT_SEG_PIPE
      // Row ids of the NEXT chunk are requested before the rows of the current one are waited
      // for: the perm -> row dependency (two memory latencies per chunk of rpg * UNR rows) is
      // taken off the critical path inside a segment.
      {
        int32_t rc[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int jj = start + rsub + u * rpg;
          rc[u] = (jj < end) ? (perm ? perm[jj] : jj) : -1;
        }
        for (int j = start + rsub; j < end; j += rpg * UNR) {
          Vec<VEC> v[UNR];
#pragma unroll
          for (int u = 0; u < UNR; ++u)
            if (rc[u] >= 0 && cv) v[u] = ldv<VEC>(x + (int64_t)rc[u] * c + c0);
          int32_t rn[UNR];
#pragma unroll
          for (int u = 0; u < UNR; ++u) {
            const int jj = j + rpg * UNR + u * rpg;
            rn[u] = (jj < end) ? (perm ? perm[jj] : jj) : -1;
          }
#pragma unroll
          for (int u = 0; u < UNR; ++u)
            if (rc[u] >= 0 && cv) {
#pragma unroll
              for (int k = 0; k < VEC; ++k) {
                float val = v[u].v[k];
                if constexpr (AFF) {
                  val = fmaf(val - t_am[k], t_sc[k], t_bs[k]);
                  val = leaky01 ? fmaxf(val, val * af.slope) : (val > 0.f ? val : val * af.slope);
                }
                combine<OP, ARG, A>(acc[k], accr[k], (A)val, rc[u]);
              }
            }
#pragma unroll
          for (int u = 0; u < UNR; ++u) rc[u] = rn[u];
        }
      }
#else
      for (int j = start + rsub; j < end; j += rpg * UNR) {
        int32_t r[UNR];
        Vec<VEC> v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int jj = j + u * rpg;
          r[u] = (jj < end) ? (perm ? perm[jj] : jj) : -1;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
          if (r[u] >= 0 && cv) v[u] = ldv<VEC>(x + (int64_t)r[u] * c + c0);
#pragma unroll
        for (int u = 0; u < UNR; ++u)
          if (r[u] >= 0 && cv) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
              float val = v[u].v[k];
              if constexpr (AFF) {
                val = fmaf(val - t_am[k], t_sc[k], t_bs[k]);     // same expression as gn_apply
                // leaky(y) == max(y, slope y) for 0 <= slope <= 1 (bitwise, incl. -0 / NaN)
                val = leaky01 ? fmaxf(val, val * af.slope) : (val > 0.f ? val : val * af.slope);
              }
              combine<OP, ARG, A>(acc[k], accr[k], (A)val, r[u]);
            }
          }
      }
#endif
      // tree across the RPG row slots of the group
      for (int o = lpr; o < (1 << g_log2); o <<= 1) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const A ov = __shfl_xor(acc[k], o, 64);
          if constexpr (ARG) {
            const int32_t orr = __shfl_xor(accr[k], o, 64);
            combine<OP, ARG, A>(acc[k], accr[k], ov, orr);
          } else {
            int32_t dummy = 0;
            combine<OP, false, A>(acc[k], dummy, ov, 0);
          }
        }
      }
      if (sv && cv && rsub == 0) {
        const int cnt = end - start;
        Vec<VEC> o;
        int32_t oa[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          A val = acc[k];
          if constexpr (OP == SPT_MEAN) val = val / (A)(cnt > 0 ? cnt : 1);
          if constexpr (OP == SPT_MIN || OP == SPT_MAX) val = (cnt > 0) ? val : 0.f;
          o.v[k] = (float)val;
          oa[k] = (cnt > 0 && accr[k] != 0x7fffffff) ? accr[k] : (int32_t)n;
        }
        stv<VEC>(out + s * c + c0, o);
        if constexpr (ARG) sti<VEC>(arg + s * c + c0, oa);
      }
    }
  }
}

// ---- segment max of 128-channel rows, row-streaming formulation ------------------------------
// The kernel above gives every segment a lane group: a segment of 35 rows (the mean of the
// level-0 -> level-1 pool) is 16 + 16 + 3 rows in flight, the group waits for its own perm ->
// row chain twice, and a wave is as slow as its longest segment.  Here a wave owns a contiguous
// range of CSR POSITIONS (cut at segment boundaries, so no segment is shared between waves and
// nothing is merged afterwards) and streams its rows sixteen at a time, whatever segments they
// belong to: one 512-byte row per load instruction (64 lanes x 2 channels), the row ids of the
// next chunk prefetched as one coalesced load, a segment's result written the moment its last
// row has been combined.  Control flow is wave-uniform (a row belongs to one segment for all
// lanes).  Same combine rule, same affine expression, bit-identical results.
// segments [sa, sb) of one graph (one set of coefficient rows): their rows are the contiguous
// CSR positions [rowptr[sa], rowptr[sb])
// X16: the rows hold bf16 values (the bf16 mode's activation storage): 8-byte loads widened
// to f32 on the way in (exact), everything else unchanged.
// RAW (with AFF): raw[s, c] = the value BEFORE the map of the element that won (0 for an empty
// segment) - what the sparse GraphNorm-backward statistics of the pooled layer need, read there as
// a stream instead of one 4-byte gather per (segment, channel).
template <bool AFF, bool X16 = false, bool RAW = false>
__device__ __forceinline__ void segmax_stream_range(
    const float* __restrict__ x, const int32_t* __restrict__ perm,
    const int32_t* __restrict__ rowptr, int64_t n, int64_t num_seg, float* __restrict__ out,
    int32_t* __restrict__ arg, float* __restrict__ raw, int64_t sa, int64_t sb,
    const float (&t_am)[4], const float (&t_sc)[4], const float (&t_bs)[4], float slope, int lane) {
  // a row = 32 lanes x 4 channels; the two half-waves take the even / odd positions of the
  // stream (2 rows per load instruction, 4 channels per combine: half the instructions per byte
  // of a 64-lane x 2-channel row).  Both halves work on the same segment unless a boundary falls
  // between an even position and the odd one after it (1 pair in ~35): that pair is taken one
  // position at a time.
  // (12 rows per chunk with the affine map: its 12 coefficient registers then still leave room
  // for 5 waves per SIMD)
  constexpr int C = 128, CHK = AFF ? (RAW && !X16 ? 10 : 12) : 16, V = 4;   // (RAW: 4 more registers)
  const int hf = lane >> 5;                               // 0: even positions, 1: odd positions
  const int c0 = (lane & 31) * V;
  const bool leaky01 = AFF && slope >= 0.f && slope <= 1.f;
  float acc[V];
  int32_t ar[V];
  float hb[RAW ? V : 1];                                  // RAW: the winner's value before the map
#pragma unroll
  for (int k = 0; k < V; ++k) {
    acc[k] = op_identity<SPT_MAX>();
    ar[k] = 0x7fffffff;
    if constexpr (RAW) hb[k] = 0.f;
  }
  // result of segment s: the odd half's partial joins the even half's, which writes the row
  auto emit = [&](int64_t s, bool empty) {
    Vec<V> o, oh;
    int32_t oa[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const float ov = __shfl_xor(acc[k], 32, 64);
      const int32_t orr = __shfl_xor(ar[k], 32, 64);
      if constexpr (RAW) {
        const float ohv = __shfl_xor(hb[k], 32, 64);
        const bool other = (ov > acc[k]) || (ov == acc[k] && orr < ar[k]);   // combine's rule
        hb[k] = other ? ohv : hb[k];
      }
      combine<SPT_MAX, true, float>(acc[k], ar[k], ov, orr);
      o.v[k] = empty ? 0.f : acc[k];
      oa[k] = (!empty && ar[k] != 0x7fffffff) ? ar[k] : (int32_t)n;
      if constexpr (RAW) {
        oh.v[k] = (!empty && ar[k] != 0x7fffffff) ? hb[k] : 0.f;
        hb[k] = 0.f;
      }
      acc[k] = op_identity<SPT_MAX>();
      ar[k] = 0x7fffffff;
    }
    if (hf == 0) {
      stv<V>(out + s * C + c0, o);
      sti<V>(arg + s * C + c0, oa);
      if constexpr (RAW) stv<V>(raw + s * C + c0, oh);
    }
  };
  int64_t s = sa;
  int64_t j = __builtin_amdgcn_readfirstlane(rowptr[sa]);
  const int64_t jend = __builtin_amdgcn_readfirstlane(rowptr[sb]);
  int64_t cur_end = __builtin_amdgcn_readfirstlane(rowptr[s + 1]);
  // the end of the segment AFTER the current one is requested a segment ahead and only made
  // uniform when it becomes current: a segment boundary then costs no memory round trip
  int32_t nxt_v = (s + 2 <= num_seg) ? rowptr[s + 2] : 0;
  auto advance = [&]() {                                  // s -> s + 1
    ++s;
    cur_end = __builtin_amdgcn_readfirstlane(nxt_v);
    nxt_v = (s + 2 <= num_seg) ? rowptr[s + 2] : 0;
  };
  // position p was the last one combined: close every segment that ends there
  auto close_at = [&](int64_t p) {
    if (p + 1 == cur_end) {
      emit(s, false);
      advance();
      while (s < sb && cur_end == p + 1) {                // empty segments behind it
        emit(s, true);
        advance();
      }
    }
  };
  while (s < sb && cur_end == j) {                       // leading empty segments
    emit(s, true);
    advance();
  }
  if (s >= sb) return;
  auto ids_of = [&](int64_t jj) {                        // lane u < 16: row id of position jj + u
    const int64_t p = jj + (lane & 15);
    return (p < jend) ? (perm ? perm[p] : (int32_t)p) : 0;
  };
  auto take = [&](const Vec<V>& v, int32_t r) {           // one row into the lane's partial
#pragma unroll
    for (int k = 0; k < V; ++k) {
      float a = v.v[k];
      if constexpr (AFF) {
        a = fmaf(a - t_am[k], t_sc[k], t_bs[k]);           // same expression as gn_apply
        a = leaky01 ? fmaxf(a, a * slope) : (a > 0.f ? a : a * slope);
      }
      if constexpr (RAW) {
        const bool win = (a > acc[k]) || (a == acc[k] && r < ar[k]);         // combine's rule
        hb[k] = win ? v.v[k] : hb[k];
      }
      combine<SPT_MAX, true, float>(acc[k], ar[k], a, r);
    }
  };
  int32_t idn = ids_of(j);
  for (; j < jend; j += CHK) {
    const int32_t idc = idn;
    idn = ids_of(j + CHK);
    Vec<V> v[CHK / 2];
    auto row_of = [&](int q) {                            // the lane's row of pair q
      const int32_t r0 = __builtin_amdgcn_readlane(idc, 2 * q);
      const int32_t r1 = __builtin_amdgcn_readlane(idc, 2 * q + 1);
      return hf ? r1 : r0;
    };
    // unconditional loads (positions past the end read row ids_of() = 0 and are never combined):
    // a load under a per-lane condition is followed by a wait and a select, which serialises the
    // eight loads of the chunk
#pragma unroll
    for (int q = 0; q < CHK / 2; ++q) {
      if constexpr (X16) {
        const uint2 u = *reinterpret_cast<const uint2*>(
            reinterpret_cast<const uint16_t*>(x) + (int64_t)row_of(q) * C + c0);
        v[q].v[0] = __uint_as_float(u.x << 16);
        v[q].v[1] = __uint_as_float(u.x & 0xffff0000u);
        v[q].v[2] = __uint_as_float(u.y << 16);
        v[q].v[3] = __uint_as_float(u.y & 0xffff0000u);
      } else {
        v[q] = ldv<V>(x + (int64_t)row_of(q) * C + c0);
      }
    }
#pragma unroll
    for (int q = 0; q < CHK / 2; ++q) {
      const int64_t p0 = j + 2 * q;
      if (p0 < jend) {
        const int32_t rq = row_of(q);
        if (p0 + 1 < cur_end && p0 + 1 < jend) {             // both rows in the current segment
          take(v[q], rq);
          close_at(p0 + 1);
        } else {                                             // a boundary between them (or the end)
          if (hf == 0) take(v[q], rq);
          close_at(p0);
          if (p0 + 1 < jend) {
            if (hf == 1) take(v[q], rq);
            close_at(p0 + 1);
          }
        }
      }
    }
  }
}

template <bool AFF, bool X16 = false, bool RAW = false>
__global__ __launch_bounds__(256, 5) void segmax_stream_kernel(
    const float* __restrict__ x, const int32_t* __restrict__ perm,
    const int32_t* __restrict__ rowptr, int64_t n, int64_t num_seg, float* __restrict__ out,
    int32_t* __restrict__ arg, float* __restrict__ raw, Affine af, int64_t rows_per_wave) {
  constexpr int C = 128;
  const int lane = threadIdx.x & 63;
  const int c0 = (lane & 31) * 4;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  // first segment whose start is >= p (num_seg when there is none)
  auto lower = [&](int64_t p) {
    int64_t lo = 0, hi = num_seg;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)__builtin_amdgcn_readfirstlane(rowptr[mid]) >= p) hi = mid; else lo = mid + 1;
    }
    return lo;
  };
  const int64_t sa = lower(wave * rows_per_wave);
  const int64_t sb = (wave == nwaves - 1) ? num_seg : lower((wave + 1) * rows_per_wave);
  float t_am[4] = {0.f, 0.f, 0.f, 0.f}, t_sc[4] = {0.f, 0.f, 0.f, 0.f}, t_bs[4] = {0.f, 0.f, 0.f, 0.f};
  // the coefficient rows depend on the segment's graph: the wave's segments are taken graph by
  // graph (graphs are contiguous runs of segments; a range spans more than one at B - 1 places)
  int64_t s_lo = sa;
  while (s_lo < sb) {
    int64_t s_hi = sb;
    if constexpr (AFF) {
      int64_t g = 0;
      if (af.seg_graph) {
        g = __builtin_amdgcn_readfirstlane((int)af.seg_graph[s_lo]);
        if ((int64_t)__builtin_amdgcn_readfirstlane((int)af.seg_graph[sb - 1]) != g) {
          int64_t lo = s_lo + 1, hi = sb;                   // first segment of another graph
          while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)__builtin_amdgcn_readfirstlane((int)af.seg_graph[mid]) != g) hi = mid;
            else lo = mid + 1;
          }
          s_hi = lo;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        t_am[k] = af.am[g * C + c0 + k];
        t_sc[k] = af.sc[g * C + c0 + k];
        t_bs[k] = af.bs[c0 + k];
      }
    }
    segmax_stream_range<AFF, X16, RAW>(x, perm, rowptr, n, num_seg, out, arg, raw, s_lo, s_hi, t_am,
                                       t_sc, t_bs, af.slope, lane);
    s_lo = s_hi;
  }
}

static std::atomic<int> g_seg_stream{-1};   // -1: from the environment (SPT_SEG_STREAM=0 turns it off)
// per-call choice of the *_ex entries (thread-local, set for the duration of one call)
static thread_local int tl_seg_stream = -1;
struct SegStreamScope {
  int prev;
  explicit SegStreamScope(int f) : prev(tl_seg_stream) { tl_seg_stream = f < 0 ? -1 : (f != 0); }
  ~SegStreamScope() { tl_seg_stream = prev; }
};
static bool seg_stream_on() {
  if (tl_seg_stream >= 0) return tl_seg_stream != 0;
  if (g_seg_stream < 0) {
    const char* e = getenv("SPT_SEG_STREAM");
    g_seg_stream = (e && e[0] == '0') ? 0 : 1;
  }
  return g_seg_stream != 0;
}
// max + arg of 128-channel rows over many rows: the row-streaming kernel
static bool seg_stream_shape(int c, int64_t n, bool want_arg) {
  return seg_stream_on() && c == 128 && want_arg && n >= (1 << 16);
}
template <bool AFF, bool X16 = false>
static void launch_stream(const float* x, const int32_t* perm, const int32_t* rowptr, int64_t n,
                          int64_t num_seg, float* out, int32_t* arg, const Affine& af,
                          hipStream_t stream, float* raw = nullptr) {
  // 8 waves per SIMD over the whole chip, at least 256 rows per wave
  // one resident round: 5 waves per SIMD (96 registers) over the whole chip - with more waves
  // than fit at once the last partial round costs more than finer ranges balance (measured:
  // 5 120 waves 1.51 ms, 8 192: 1.59, 32 768: 1.60 at 15 M rows)
  static const int64_t kWaves = [] {
    const char* e = getenv("SPT_SEG_STREAM_WAVES");
    return e ? (int64_t)atoll(e) : (int64_t)0;
  }();
  int64_t waves = kWaves > 0 ? kWaves : (int64_t)256 * 4 * 5;
  if (waves * 256 > n) waves = n / 256 > 4 ? n / 256 : 4;
  const int grid = (int)ceil_div(waves, (int64_t)4);
  const int64_t rows_per_wave = n / ((int64_t)grid * 4) + 1;
  if constexpr (AFF) {
    if (raw) {
      segmax_stream_kernel<AFF, X16, true><<<grid, 256, 0, stream>>>(x, perm, rowptr, n, num_seg, out,
                                                                     arg, raw, af, rows_per_wave);
      return;
    }
  }
  segmax_stream_kernel<AFF, X16><<<grid, 256, 0, stream>>>(x, perm, rowptr, n, num_seg, out, arg,
                                                           nullptr, af, rows_per_wave);
}

// Row-parallel "gather with modifier":
//   MODE 0: out[i,:] = src[idx[i],:]                      (gather / sum bwd)
//   MODE 1: out[i,:] = src[idx[i],:] / max(count[idx[i]],1)   (mean bwd)
//   MODE 2: out[i,c] = arg[idx[i],c]==i ? src[idx[i],c] : 0   (min/max bwd)
template <int MODE, int VEC>
__global__ __launch_bounds__(256) void gather_mod_kernel(
    const float* __restrict__ src, const int32_t* __restrict__ arg,
    const int64_t* __restrict__ idx, const int32_t* __restrict__ rowptr,
    int64_t n, int c, int lpr_log2, float* __restrict__ out) {
  constexpr int UNR = 4;
  const int lane = threadIdx.x & 63;
  const int lpr = 1 << lpr_log2;
  const int rpw = 64 >> lpr_log2;  // rows per wave-instruction
  const int rsub = lane >> lpr_log2;
  const int lr = lane & (lpr - 1);
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int ctile = lpr * VEC;
  for (int64_t rb = wave * rpw * UNR; rb < n; rb += nwaves * rpw * UNR) {
    int64_t row[UNR];
    int64_t p[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      row[u] = rb + u * rpw + rsub;
      p[u] = (row[u] < n) ? idx[row[u]] : -1;
    }
    for (int cb = 0; cb < c; cb += ctile) {
      const int c0 = cb + lr * VEC;
      if (c0 >= c) continue;
      Vec<VEC> v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u)
        if (p[u] >= 0) v[u] = ldv<VEC>(src + p[u] * c + c0);
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (p[u] < 0) continue;
        if constexpr (MODE == 1) {
          const int cnt = rowptr[p[u] + 1] - rowptr[p[u]];
          const float d = (float)(cnt > 0 ? cnt : 1);
#pragma unroll
          for (int k = 0; k < VEC; ++k) v[u].v[k] = v[u].v[k] / d;
        }
        if constexpr (MODE == 2) {
#pragma unroll
          for (int k = 0; k < VEC; ++k) {
            const int32_t a = arg[p[u] * c + c0 + k];
            v[u].v[k] = (a == (int32_t)row[u]) ? v[u].v[k] : 0.f;
          }
        }
        stv<VEC>(out + row[u] * c + c0, v[u]);
      }
    }
  }
}

// Backward in CSR order (wide rows): the parent's gout (and arg) row is loaded
// ONCE per segment and streamed to its children through perm - no re-gather of
// the parent row per child as in the idx-ordered kernel above.
//   MODE 0: gx[r,:] = gout[s,:]          MODE 1: gout[s,:]/max(cnt,1)
//   MODE 2: gx[r,c] = arg[s,c]==r ? gout[s,c] : 0
template <int MODE, int VEC>
__global__ __launch_bounds__(256) void segcsr_bwd_kernel(
    const float* __restrict__ gout, const int32_t* __restrict__ arg,
    const int32_t* __restrict__ perm, const int32_t* __restrict__ rowptr,
    int64_t num_seg, int c, int lpr_log2, int rpg_log2, float* __restrict__ gx) {
  constexpr int UNR = 4;
  const int lane = threadIdx.x & 63;
  const int g_log2 = lpr_log2 + rpg_log2;
  const int lpr = 1 << lpr_log2;
  const int rpg = 1 << rpg_log2;
  const int spw = 64 >> g_log2;
  const int slot = lane >> g_log2;
  const int lg = lane & ((1 << g_log2) - 1);
  const int rsub = lg >> lpr_log2;
  const int lr = lg & (lpr - 1);
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int ctile = lpr * VEC;
  for (int64_t sbase = wave * spw; sbase < num_seg; sbase += nwaves * spw) {
    const int64_t s = sbase + slot;
    const bool sv = s < num_seg;
    const int start = sv ? rowptr[s] : 0;
    const int end = sv ? rowptr[s + 1] : 0;
    if (end == start) continue;
    for (int cb = 0; cb < c; cb += ctile) {
      const int c0 = cb + lr * VEC;
      if (c0 >= c) continue;
      Vec<VEC> g = ldv<VEC>(gout + s * c + c0);
      int32_t a[VEC];
      if constexpr (MODE == 1) {
        const float d = (float)(end - start);
#pragma unroll
        for (int k = 0; k < VEC; ++k) g.v[k] = g.v[k] / d;
      }
      if constexpr (MODE == 2) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) a[k] = arg[s * c + c0 + k];
      }
      for (int j = start + rsub; j < end; j += rpg * UNR) {
        int32_t r[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int jj = j + u * rpg;
          r[u] = (jj < end) ? (perm ? perm[jj] : jj) : -1;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          if (r[u] < 0) continue;
          Vec<VEC> o = g;
          if constexpr (MODE == 2) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) o.v[k] = (a[k] == r[u]) ? g.v[k] : 0.f;
          }
          stv_nt<VEC>(gx + (int64_t)r[u] * c + c0, o);
        }
      }
    }
  }
}

__global__ void segcsr_sum_i64_kernel(const int64_t* __restrict__ x,
                                      const int32_t* __restrict__ perm,
                                      const int32_t* __restrict__ rowptr,
                                      int64_t num_seg, int c,
                                      int64_t* __restrict__ out) {
  const int64_t total = num_seg * c;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t s = t / c;
    const int ch = (int)(t - s * c);
    int64_t acc = 0;
    for (int j = rowptr[s]; j < rowptr[s + 1]; ++j) {
      const int64_t r = perm ? perm[j] : j;
      acc += x[r * c + ch];
    }
    out[t] = acc;
  }
}

static inline int log2_floor(int64_t v) {
  int b = 0;
  while ((((int64_t)1) << (b + 1)) <= v) ++b;
  return b;
}
static inline int log2_ceil(int64_t v) {
  int b = 0;
  while ((((int64_t)1) << b) < v) ++b;
  return b;
}

static int rows_in_flight_log2(int64_t n, int64_t num_seg, int lpr_log2) {
  // rows in flight per segment: aim at >= 4 rows per lane slot
  const int64_t avg = (num_seg > 0) ? n / num_seg : 0;
  int rpg_log2 = log2_floor(avg / 4 > 1 ? avg / 4 : 1);
  if (rpg_log2 > 6 - lpr_log2) rpg_log2 = 6 - lpr_log2;
  rpg_log2 -= SPT_SEG_RPG_BIAS;
  if (rpg_log2 < 0) rpg_log2 = 0;
  return rpg_log2;
}

struct RowShape {
  int vec, lpr_log2;
};
static RowShape row_shape(int c) {
  RowShape r;
  r.vec = (c % 4 == 0) ? 4 : (c % 2 == 0) ? 2 : 1;
  int lanes = c / r.vec;
  if (lanes > 64) lanes = 64;
  r.lpr_log2 = log2_ceil(lanes);
  return r;
}

template <int OP, int VEC>
static void launch_reduce(bool want_arg, const float* x, const int32_t* perm,
                          const int32_t* rowptr, int64_t n, int64_t num_seg, int c,
                          int lpr_log2, int rpg_log2, float* out, int32_t* arg,
                          hipStream_t stream) {
  const int spw = 64 >> (lpr_log2 + rpg_log2);
  int grid;
  {
    const int64_t want = ceil_div(ceil_div(num_seg, spw), 4);
    const int64_t cap = (int64_t)256 * 16 * SPT_SEG_GRIDCAP_MUL;
    grid = (int)(want < cap ? (want > 0 ? want : 1) : cap);
  }
  if constexpr (OP == SPT_MIN || OP == SPT_MAX) {
    if (want_arg) {
      segcsr_reduce_kernel<OP, VEC, true><<<grid, 256, 0, stream>>>(
          x, perm, rowptr, n, num_seg, c, lpr_log2, rpg_log2, out, arg);
      return;
    }
  }
  segcsr_reduce_kernel<OP, VEC, false><<<grid, 256, 0, stream>>>(
      x, perm, rowptr, n, num_seg, c, lpr_log2, rpg_log2, out, arg);
}

template <int OP>
static void launch_reduce_vec(int vec, bool want_arg, const float* x,
                              const int32_t* perm, const int32_t* rowptr, int64_t n,
                              int64_t num_seg, int c, int lpr_log2, int rpg_log2,
                              float* out, int32_t* arg, hipStream_t stream) {
  if (vec == 4)
    launch_reduce<OP, 4>(want_arg, x, perm, rowptr, n, num_seg, c, lpr_log2, rpg_log2, out, arg, stream);
  else if (vec == 2)
    launch_reduce<OP, 2>(want_arg, x, perm, rowptr, n, num_seg, c, lpr_log2, rpg_log2, out, arg, stream);
  else
    launch_reduce<OP, 1>(want_arg, x, perm, rowptr, n, num_seg, c, lpr_log2, rpg_log2, out, arg, stream);
}

template <int MODE>
static void launch_bwd_csr(int vec, const float* gout, const int32_t* arg,
                           const int32_t* perm, const int32_t* rowptr,
                           int64_t num_seg, int c, int lpr_log2, int rpg_log2,
                           float* gx, hipStream_t stream) {
  const int spw = 64 >> (lpr_log2 + rpg_log2);
  int grid;
  {
    const int64_t want = ceil_div(ceil_div(num_seg, spw), 4);
    const int64_t cap = (int64_t)256 * 16 * SPT_SEG_GRIDCAP_MUL;
    grid = (int)(want < cap ? (want > 0 ? want : 1) : cap);
  }
  if (vec == 4)
    segcsr_bwd_kernel<MODE, 4><<<grid, 256, 0, stream>>>(gout, arg, perm, rowptr, num_seg, c, lpr_log2, rpg_log2, gx);
  else if (vec == 2)
    segcsr_bwd_kernel<MODE, 2><<<grid, 256, 0, stream>>>(gout, arg, perm, rowptr, num_seg, c, lpr_log2, rpg_log2, gx);
  else
    segcsr_bwd_kernel<MODE, 1><<<grid, 256, 0, stream>>>(gout, arg, perm, rowptr, num_seg, c, lpr_log2, rpg_log2, gx);
}

template <int MODE>
static void launch_gather(int vec, const float* src, const int32_t* arg,
                          const int64_t* idx, const int32_t* rowptr, int64_t n, int c,
                          int lpr_log2, float* out, hipStream_t stream) {
  const int rpw = 64 >> lpr_log2;
  const int grid = stream_grid(ceil_div(n, (int64_t)rpw * 4), 4);
  if (vec == 4)
    gather_mod_kernel<MODE, 4><<<grid, 256, 0, stream>>>(src, arg, idx, rowptr, n, c, lpr_log2, out);
  else if (vec == 2)
    gather_mod_kernel<MODE, 2><<<grid, 256, 0, stream>>>(src, arg, idx, rowptr, n, c, lpr_log2, out);
  else
    gather_mod_kernel<MODE, 1><<<grid, 256, 0, stream>>>(src, arg, idx, rowptr, n, c, lpr_log2, out);
}

}  // namespace spt

using namespace spt;

extern "C" int spt_segcsr_use_stream(int on) {
  const int prev = seg_stream_on() ? 1 : 0;
  g_seg_stream = on != 0;
  return prev;
}

extern "C" int spt_segcsr_reduce_f32(int op, const float* x, const int32_t* perm,
                                     const int32_t* rowptr, int64_t n,
                                     int64_t num_seg, int c, float* out,
                                     int32_t* arg, spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(op >= SPT_SUM && op <= SPT_MAX, "unknown op");
  SPT_CHECK_ARG(n >= 0 && num_seg >= 0 && c >= 1, "bad shape");
  SPT_CHECK_ARG(rowptr && out && (x || n == 0), "null pointer");
  if (num_seg == 0) return 0;
  const RowShape rs = row_shape(c);
  const int rpg_log2 = rows_in_flight_log2(n, num_seg, rs.lpr_log2);
  const bool want_arg = arg != nullptr;
  switch (op) {
    case SPT_SUM:
      launch_reduce_vec<SPT_SUM>(rs.vec, false, x, perm, rowptr, n, num_seg, c, rs.lpr_log2, rpg_log2, out, arg, stream);
      break;
    case SPT_MEAN:
      launch_reduce_vec<SPT_MEAN>(rs.vec, false, x, perm, rowptr, n, num_seg, c, rs.lpr_log2, rpg_log2, out, arg, stream);
      break;
    case SPT_MIN:
      launch_reduce_vec<SPT_MIN>(rs.vec, want_arg, x, perm, rowptr, n, num_seg, c, rs.lpr_log2, rpg_log2, out, arg, stream);
      break;
    default:
      if (seg_stream_shape(c, n, want_arg))
        launch_stream<false>(x, perm, rowptr, n, num_seg, out, arg, Affine{}, stream);
      else
        launch_reduce_vec<SPT_MAX>(rs.vec, want_arg, x, perm, rowptr, n, num_seg, c, rs.lpr_log2, rpg_log2, out, arg, stream);
      break;
  }
  SPT_CHECK_LAUNCH();
  return 0;
}

extern "C" int spt_segcsr_reduce_ex_f32(int op, const float* x, const int32_t* perm,
                                        const int32_t* rowptr, int64_t n, int64_t num_seg, int c,
                                        float* out, int32_t* arg, int formulation,
                                        spt_stream_t stream_) {
  SegStreamScope scope(formulation);
  return spt_segcsr_reduce_f32(op, x, perm, rowptr, n, num_seg, c, out, arg, stream_);
}

extern "C" int spt_segcsr_max_affine_ex_f32(const float* x, const int32_t* perm,
                                            const int32_t* rowptr, int64_t n, int64_t num_seg,
                                            int c, const float* am, const float* scale,
                                            const float* bias, float act_slope,
                                            const int64_t* seg_graph, float* out, int32_t* arg,
                                            int formulation, spt_stream_t stream_) {
  SegStreamScope scope(formulation);
  return spt_segcsr_max_affine_f32(x, perm, rowptr, n, num_seg, c, am, scale, bias, act_slope,
                                   seg_graph, out, arg, stream_);
}

extern "C" int spt_segcsr_max_affine_f32(const float* x, const int32_t* perm,
                                         const int32_t* rowptr, int64_t n, int64_t num_seg,
                                         int c, const float* am, const float* scale,
                                         const float* bias, float act_slope,
                                         const int64_t* seg_graph, float* out, int32_t* arg,
                                         spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(n >= 0 && num_seg >= 0 && c >= 4 && (c & 3) == 0, "c must be a multiple of 4");
  SPT_CHECK_ARG(rowptr && out && arg && am && scale && bias && (x || n == 0), "null pointer");
  if (num_seg == 0) return 0;
  const RowShape rs = row_shape(c);
  SPT_CHECK_ARG(rs.vec == 4, "unsupported row shape");
  const int rpg_log2 = rows_in_flight_log2(n, num_seg, rs.lpr_log2);
  const int spw = 64 >> (rs.lpr_log2 + rpg_log2);
  const int64_t want = ceil_div(ceil_div(num_seg, spw), 4);
  const int64_t cap = (int64_t)256 * 16 * SPT_SEG_GRIDCAP_MUL;
  const int grid = (int)(want < cap ? (want > 0 ? want : 1) : cap);
  Affine af;
  af.am = am; af.sc = scale; af.bs = bias; af.seg_graph = seg_graph; af.slope = act_slope;
  if (seg_stream_shape(c, n, true))
    launch_stream<true>(x, perm, rowptr, n, num_seg, out, arg, af, stream);
  else
    segcsr_reduce_kernel<SPT_MAX, 4, true, true><<<grid, 256, 0, stream>>>(
        x, perm, rowptr, n, num_seg, c, rs.lpr_log2, rpg_log2, out, arg, af);
  SPT_CHECK_LAUNCH();
  return 0;
}

// Same with x stored as bf16 [n, c] (the fused layers' activation storage option): the streaming
// kernel only - c = 128, n >= 65 536, as spt_segcsr_max_affine_bf16_supported reports.
extern "C" int spt_segcsr_max_affine_bf16_supported(int c, int64_t n) {
  return seg_stream_shape(c, n, true) ? 1 : 0;
}
extern "C" int spt_segcsr_max_affine_bf16(const void* x_bf16, const int32_t* perm,
                                          const int32_t* rowptr, int64_t n, int64_t num_seg,
                                          int c, const float* am, const float* scale,
                                          const float* bias, float act_slope,
                                          const int64_t* seg_graph, float* out, int32_t* arg,
                                          spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(n >= 0 && num_seg >= 0, "bad shape");
  SPT_CHECK_ARG(spt_segcsr_max_affine_bf16_supported(c, n), "bf16 rows: c = 128 and n >= 65536 only");
  SPT_CHECK_ARG(rowptr && out && arg && am && scale && bias && x_bf16, "null pointer");
  if (num_seg == 0) return 0;
  Affine af;
  af.am = am; af.sc = scale; af.bs = bias; af.seg_graph = seg_graph; af.slope = act_slope;
  launch_stream<true, true>(reinterpret_cast<const float*>(x_bf16), perm, rowptr, n, num_seg, out, arg,
                            af, stream);
  SPT_CHECK_LAUNCH();
  return 0;
}

// The same pool (x f32 or - x_is_bf16 - bf16 rows) with a third output, raw[s, c] = the value
// BEFORE the map of the element that won (0 for an empty segment): the sparse GraphNorm-backward
// statistics of the pooled layer (spt_graphnorm_bwd_stats_sparse_raw_f32) then read a stream
// instead of gathering x[arg[s, c], c].  The streaming kernel only (c = 128, n >= 65 536, as
// spt_segcsr_max_affine_raw_supported reports).
extern "C" int spt_segcsr_max_affine_raw_supported(int c, int64_t n) {
  return seg_stream_shape(c, n, true) ? 1 : 0;
}
extern "C" int spt_segcsr_max_affine_raw_f32(const void* x, int x_is_bf16, const int32_t* perm,
                                             const int32_t* rowptr, int64_t n, int64_t num_seg,
                                             int c, const float* am, const float* scale,
                                             const float* bias, float act_slope,
                                             const int64_t* seg_graph, float* out, int32_t* arg,
                                             float* raw, spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(n >= 0 && num_seg >= 0, "bad shape");
  SPT_CHECK_ARG(spt_segcsr_max_affine_raw_supported(c, n), "raw output: c = 128 and n >= 65536 only");
  SPT_CHECK_ARG(rowptr && out && arg && raw && am && scale && bias && x, "null pointer");
  if (num_seg == 0) return 0;
  Affine af;
  af.am = am; af.sc = scale; af.bs = bias; af.seg_graph = seg_graph; af.slope = act_slope;
  if (x_is_bf16)
    launch_stream<true, true>(reinterpret_cast<const float*>(x), perm, rowptr, n, num_seg, out, arg, af,
                              stream, raw);
  else
    launch_stream<true, false>(reinterpret_cast<const float*>(x), perm, rowptr, n, num_seg, out, arg,
                               af, stream, raw);
  SPT_CHECK_LAUNCH();
  return 0;
}

extern "C" int spt_segcsr_reduce_bwd_f32(int op, const float* gout,
                                         const int32_t* arg, const int64_t* idx,
                                         const int32_t* perm,
                                         const int32_t* rowptr, int64_t n,
                                         int64_t num_seg, int c, float* gx,
                                         spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(op >= SPT_SUM && op <= SPT_MAX, "unknown op");
  SPT_CHECK_ARG(n >= 0 && num_seg >= 0 && c >= 1, "bad shape");
  if (n == 0) return 0;
  SPT_CHECK_ARG(gout && idx && gx, "null pointer");
  const RowShape rs = row_shape(c);
  // wide rows (>= one 128-B line) and every row owned by a segment: stream in
  // CSR order; narrow rows keep coalesced idx-ordered writes.
  if (perm && rowptr && c * 4 >= 128 && n / (num_seg > 0 ? num_seg : 1) >= 2) {
    SPT_CHECK_ARG(op == SPT_SUM || op == SPT_MEAN || arg, "min/max backward needs arg");
    const int rpg_log2 = rows_in_flight_log2(n, num_seg, rs.lpr_log2);
    if (op == SPT_SUM)
      launch_bwd_csr<0>(rs.vec, gout, nullptr, perm, rowptr, num_seg, c, rs.lpr_log2, rpg_log2, gx, stream);
    else if (op == SPT_MEAN)
      launch_bwd_csr<1>(rs.vec, gout, nullptr, perm, rowptr, num_seg, c, rs.lpr_log2, rpg_log2, gx, stream);
    else
      launch_bwd_csr<2>(rs.vec, gout, arg, perm, rowptr, num_seg, c, rs.lpr_log2, rpg_log2, gx, stream);
    SPT_CHECK_LAUNCH();
    return 0;
  }
  if (op == SPT_SUM) {
    launch_gather<0>(rs.vec, gout, nullptr, idx, nullptr, n, c, rs.lpr_log2, gx, stream);
  } else if (op == SPT_MEAN) {
    SPT_CHECK_ARG(rowptr, "mean backward needs rowptr");
    launch_gather<1>(rs.vec, gout, nullptr, idx, rowptr, n, c, rs.lpr_log2, gx, stream);
  } else {
    SPT_CHECK_ARG(arg, "min/max backward needs arg");
    launch_gather<2>(rs.vec, gout, arg, idx, nullptr, n, c, rs.lpr_log2, gx, stream);
  }
  SPT_CHECK_LAUNCH();
  return 0;
}

extern "C" int spt_gather_rows_f32(const float* x, const int64_t* idx, int64_t n,
                                   int64_t num_src, int c, float* out,
                                   spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(n >= 0 && num_src >= 0 && c >= 1, "bad shape");
  if (n == 0) return 0;
  SPT_CHECK_ARG(x && idx && out, "null pointer");
  const RowShape rs = row_shape(c);
  launch_gather<0>(rs.vec, x, nullptr, idx, nullptr, n, c, rs.lpr_log2, out, stream);
  SPT_CHECK_LAUNCH();
  return 0;
}

extern "C" int spt_segcsr_sum_i64(const int64_t* x, const int32_t* perm,
                                  const int32_t* rowptr, int64_t n, int64_t num_seg,
                                  int c, int64_t* out, spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(n >= 0 && num_seg >= 0 && c >= 1, "bad shape");
  if (num_seg == 0) return 0;
  SPT_CHECK_ARG(rowptr && out && (x || n == 0), "null pointer");
  segcsr_sum_i64_kernel<<<stream_grid(num_seg * c, 256), 256, 0, stream>>>(
      x, perm, rowptr, num_seg, c, out);
  SPT_CHECK_LAUNCH();
  return 0;
}
