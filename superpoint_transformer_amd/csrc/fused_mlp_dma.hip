// Backward of a fused MLP layer (see fused_mlp.hip for the layer's algebra) with the tile
// staging done by the memory system instead of by the wave: `global_load_lds_dwordx4` moves the
// raw rows of a 16-row tile (h: 16 x N, x_prev: 16 x K, and - pooled - the (gout, arg) rows of
// the segments the tile touches) from global memory straight into the wave's LDS buffers, ONE
// round trip per tile for all of them, no VGPRs in between.
//
// Why: the register-staged kernel (fused_mlp.hip, bwd_kernel_bf) reads a tile as
// load -> transform -> ds_write groups, 6 dependent memory round trips per tile at 64 -> 128
// (4 for h / gout / arg, 2 for x_prev); with the 128 gW accumulators a wave holds there are only
// 2 waves per SIMD to hide them: 12.4 us per wave-tile, of which ~2 us are instructions
// (profiles/r02q: 41 % of the cycles waiting on memory).  A deeper register prefetch does not fit
// (PIPE variant: 1 wave per SIMD, slower).  LDS-DMA needs no registers for data in flight.
//
// Layout.  A DMA instruction writes 64 lanes x 16 B contiguously (1 KB at M0's base) but every
// lane picks its own GLOBAL address, so the 16-byte chunks of a row can land in any order.  Rows
// stay unpadded (N floats) and chunk j of row i is stored at chunk position j ^ (i & 7)
// (x_prev: j ^ 4 ((i >> 2) & 1)): with that swizzle both matrix-operand read patterns
// are bank-conflict free -
//   gW A operand: lane (g, c) reads gh[4 g + r][16 nb + c]: the two lane groups of a 32-lane
//                 LDS cycle read rows r and r + 4, 16 floats apart after the swizzle;
//   gx A operand: lane (g, c) reads 8 floats of row c at column 32 s + 8 g: the 8 lanes of a
//                 16-byte LDS cycle read 8 different rows = 8 consecutive chunk positions.
// gh = GraphNorm-backward(gy, h) overwrites h in place (same slot, same lane); gx leaves through
// the x_prev buffer (in place, after the statistics read x) as whole 16-byte row chunks.
//
// LDS per wave at 64 -> 128: 8 KB (h / gh) + 4 KB (x_prev / gx) + 2 KB (gout, arg of 2
// segments) = 14 KB; 8 waves + W^T as split bf16 (35 KB) + tables = 151 KB of the CU's 160.
#include <math.h>

#include "common.hpp"

// measurement builds only (tools/build_variant.sh ... -DSPT_FDMA_SKIP=<bits>): leave out
// 1 the GraphNorm-backward transform, 2 the gW GEMM, 4 the gx GEMM, 8 the statistics, 16 the gx store,
// 32 the DMA (instruction costs alone)
#ifndef SPT_FDMA_SKIP
#define SPT_FDMA_SKIP 0
#endif

namespace spt {
namespace fdma {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int TR = 16;

__device__ __forceinline__ double xg_sum_d(double v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ void lds_order() {   // this wave's LDS writes before its later reads
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void lds_dma16(const void* g, float* lds) {
#if SPT_FDMA_SKIP & 32
  return;                                         // measurement: no tile traffic at all
#endif
  const unsigned a = __builtin_amdgcn_readfirstlane(
      (unsigned)(size_t)((__attribute__((address_space(3))) void*)lds));
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
               :: "s"(a), "v"(g) : "memory");
}
__device__ __forceinline__ void wait_vm0() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void wait_lds() {      // LDS reads returned: the buffer may be overwritten
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
// the lane id as a value the compiler cannot see through: index arithmetic derived from it is
// redone where it is used instead of being hoisted out of the tile loop into (scarce) registers
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
template <int NV, typename V>
__device__ __forceinline__ void split_bf16(const float (&x)[NV], V& hi, V& lo) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const __bf16 h = (__bf16)x[i];
    hi[i] = h;
    lo[i] = (__bf16)(x[i] - (float)h);
  }
}
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

// value of lane (base + sel) of v, sel in [0, R), base wave-uniform: readlane + select
template <int R>
__device__ __forceinline__ int pick_lane(int v, int base, int sel) {
  int r = __builtin_amdgcn_readlane(v, base);
#pragma unroll
  for (int j = 1; j < R; ++j) r = (sel == j) ? __builtin_amdgcn_readlane(v, base + j) : r;
  return r;
}

// Arguments as fused_mlp.hip's bwd_kernel_bf (K, N compile-time: whole 16-byte row chunks and
// whole 32-column MFMA steps).  POOLED: the rows walk positions [r0, r1) of the pool's CSR order.
// S16 (bf16 mode's activation storage): h and xprev hold bf16 rows.  The DMA lands the raw bf16
// rows (half the bytes, half the DMA instructions) in the UPPER HALF of the wave's h / x buffers;
// the GraphNorm-backward transform reads its four h values from there (8 bytes) and writes gh as
// f32 into the usual swizzled slot, a short pass widens the x rows the same way - row blocks are
// visited in ascending order, so an f32 block never overwrites bf16 rows that are still to be read
// (block i ends at byte 1024 (i + 1) <= the first unread bf16 row).  Everything downstream (the two
// GEMMs, the statistics, gx as f32 rows) is unchanged.
template <int K, int N, int NW, int OCC, bool LO, bool POOLED, bool S16 = false>
__global__ __launch_bounds__(NW * 64, OCC) void bwd_dma_kernel(
    const float* __restrict__ gy, const float* __restrict__ h, int64_t r0, int64_t r1,
    const float* __restrict__ am, const float* __restrict__ sc, const float* __restrict__ bs,
    float slope, const float* __restrict__ c1, const float* __restrict__ c2,
    const float* __restrict__ c3, const float* __restrict__ xprev,
    const float* __restrict__ pam, const float* __restrict__ psc, const float* __restrict__ pbs,
    float pslope, const float* __restrict__ W, float* __restrict__ gx,
    float* __restrict__ gw_partial, double* __restrict__ pstat_partial,
    const int32_t* __restrict__ perm, const int32_t* __restrict__ pos_seg,
    const float* __restrict__ gout, const int32_t* __restrict__ arg, FmlpRuns rt) {
  constexpr int NC = N / 4, KC = K / 4, NBK = N / 16, KB = K / 16, NS = N / 32;
  if (rt.n > 0) {                           // multi-run launch (common.hpp): blockIdx.y = run
    const int run_ = blockIdx.y, gph_ = rt.g[run_];
    r0 = rt.r0[run_];
    r1 = rt.r1[run_];
    am += (size_t)gph_ * N; sc += (size_t)gph_ * N;
    c1 += (size_t)gph_ * N; c2 += (size_t)gph_ * N; c3 += (size_t)gph_ * N;
    if (pam) { pam += (size_t)gph_ * K; psc += (size_t)gph_ * K; }
    gw_partial += (size_t)run_ * gridDim.x * N * K;          // one table per workgroup (epilogue)
    if (pstat_partial) pstat_partial += (size_t)run_ * gridDim.x * NW * (2 * K + 1);
  }
  constexpr int HI = TR * NC / 64, XI = TR * KC / 64;   // DMA instructions per tile (h, x)
  constexpr int RH = 64 / NC;                            // rows of h per DMA instruction
  constexpr int NSEG = RH;                               // segments one (gout | arg) DMA covers
  constexpr int LDT = N + 8;
  constexpr int W_H = TR * N, W_X = TR * K, W_G = POOLED ? NSEG * N : 0;
  static_assert(K % 16 == 0 && N % 32 == 0 && (TR * NC) % 64 == 0 && (TR * KC) % 64 == 0, "shape");
  __shared__ __attribute__((aligned(16))) float lw[NW][W_H + W_X + 2 * W_G];
  __shared__ __attribute__((aligned(16))) __bf16 wt_hi[K * LDT];
  __shared__ __attribute__((aligned(16))) __bf16 wt_lo[LO ? K * LDT : 8];
  __shared__ __attribute__((aligned(16))) float pt[3 * K];      // previous norm: am | sc | bs
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  float* HB = lw[wid];
  float* XB = HB + W_H;
  float* GO = XB + W_X;
  int* AR = reinterpret_cast<int*>(GO + W_G);
  const bool pre = pam != nullptr;
  for (int i = threadIdx.x; i < K * N; i += NW * 64) {
    const int k = i / N, n = i - k * N;
    const float w = W[(size_t)n * K + k];
    const __bf16 hh = (__bf16)w;
    wt_hi[k * LDT + n] = hh;
    if constexpr (LO) wt_lo[k * LDT + n] = (__bf16)(w - (float)hh);
  }
  for (int i = threadIdx.x; i < K; i += NW * 64) {
    pt[i] = pre ? pam[i] : 0.f;
    pt[K + i] = pre ? psc[i] : 1.f;
    pt[2 * K + i] = pre ? pbs[i] : 0.f;
  }
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

  // ---- per-lane constants ---------------------------------------------------------------
  // DMA slot (instruction i, lane L) <-> row hi + RH i, chunk position p = L % NC; the transform
  // visits the same LDS slots but each lane always owns COLUMN chunk n4 = L % NC (it sits at
  // position n4 ^ (row & 7)): its GraphNorm tables live in 24 registers for the whole launch.
  const float4 t_am = *reinterpret_cast<const float4*>(am + 4 * (lane % NC));
  const float4 t_sc = *reinterpret_cast<const float4*>(sc + 4 * (lane % NC));
  const float4 t_bs = *reinterpret_cast<const float4*>(bs + 4 * (lane % NC));
  const float4 t_c1 = *reinterpret_cast<const float4*>(c1 + 4 * (lane % NC));
  const float4 t_c2 = *reinterpret_cast<const float4*>(c2 + 4 * (lane % NC));
  const float4 t_c3 = *reinterpret_cast<const float4*>(c3 + 4 * (lane % NC));
  // index arithmetic below derives from `ln`, re-made opaque once per tile: the ~40 LDS / global
  // offsets it leads to are recomputed where used (one or two integer ops each) instead of being
  // hoisted out of the tile loop into registers the 128 gW accumulators leave no room for
  int ln = opaque(lane);
  auto gh1 = [&](float hh, float gg, float aa, float ss, float bb, float q1, float q2, float q3) {
    const float o = hh - aa;
    if (slope != 1.f) {
      const float y = fmaf(o, ss, bb);
      gg = (y > 0.f) ? gg : gg * slope;
    }
    return fmaf(q1, gg, -fmaf(q2, o, q3));
  };
  auto gh_of = [&](const float4& hv, const float4& gv) {
    return make_float4(gh1(hv.x, gv.x, t_am.x, t_sc.x, t_bs.x, t_c1.x, t_c2.x, t_c3.x),
                       gh1(hv.y, gv.y, t_am.y, t_sc.y, t_bs.y, t_c1.y, t_c2.y, t_c3.y),
                       gh1(hv.z, gv.z, t_am.z, t_sc.z, t_bs.z, t_c1.z, t_c2.z, t_c3.z),
                       gh1(hv.w, gv.w, t_am.w, t_sc.w, t_bs.w, t_c1.w, t_c2.w, t_c3.w));
  };
  // value of lane (base + hi) of v, base a compile-time multiple of RH: readlane + select
  auto pick = [&](int v, int base) {
    const int hi = ln / NC;
    int r = __builtin_amdgcn_readlane(v, base);
#pragma unroll
    for (int j = 1; j < RH; ++j) r = (hi == j) ? __builtin_amdgcn_readlane(v, base + j) : r;
    return r;
  };
  constexpr int RX = 64 / KC;                    // rows of x per DMA instruction
  auto pickx = [&](int v, int base) {
    const int hx = ln / KC;
    int r = __builtin_amdgcn_readlane(v, base);
#pragma unroll
    for (int j = 1; j < RX; ++j) r = (hx == j) ? __builtin_amdgcn_readlane(v, base + j) : r;
    return r;
  };

  // POOLED: lane rr < 16 holds the row id / segment of row rr of a tile (0 past the end: the DMA
  // of a short last tile re-reads row perm[.] = 0 / row0, masked after the transform)
  int rid_n = 0, seg_n = 0;
  auto load_ids = [&](int64_t t) {
    rid_n = seg_n = 0;
    const int64_t rowf = r0 + t * TR;
    if constexpr (POOLED) {
      if (t < ntiles && rowf + lane < r1 && lane < TR) {
        rid_n = perm[rowf + lane];
        seg_n = pos_seg[rowf + lane];
      }
    } else {
      if (t < ntiles && lane < TR) rid_n = (rowf + lane < r1) ? lane : 0;   // row offset in the tile
    }
  };
  constexpr int NC16 = N / 8, RH16 = 64 / NC16, HI16 = TR / RH16;     // bf16 rows: 8 values per chunk
  constexpr int KC16 = K / 8, RX16 = 64 / KC16, XI16 = (TR + RX16 - 1) / RX16;
  static_assert(!S16 || (TR % RH16 == 0 && RX16 <= TR), "bf16 rows: whole DMA instructions per tile");
  auto issue_h = [&](int64_t t, int rid_l, int seg_l) {
    if (t >= ntiles) return;
    const int64_t row0 = r0 + t * TR;
    const int hi = ln / NC, n4 = ln % NC;
    if constexpr (S16) {
      const uint16_t* hb = reinterpret_cast<const uint16_t*>(h) + (POOLED ? 0 : row0 * N);
      const int h16 = ln / NC16, p16 = ln % NC16;
#pragma unroll
      for (int i = 0; i < HI16; ++i)
        lds_dma16(hb + (int64_t)pick_lane<RH16>(rid_l, RH16 * i, h16) * N + 8 * p16,
                  HB + W_H / 2 + 256 * i);
    } else {
      const float* hb = POOLED ? h : h + row0 * N;
#pragma unroll
      for (int i = 0; i < HI; ++i) {
        const int rr7 = (hi + RH * i) & 7;
        lds_dma16(hb + (int64_t)pick(rid_l, RH * i) * N + 4 * (n4 ^ rr7), HB + 256 * i);
      }
    }
    if constexpr (POOLED) {
      const int cnt = (int)((r1 - row0) < TR ? (r1 - row0) : TR);
      const int s0 = __builtin_amdgcn_readlane(seg_l, 0);
      const int s1 = __builtin_amdgcn_readfirstlane(__shfl(seg_l, cnt - 1, 64));
      const int64_t sgm = (s0 + hi < s1) ? s0 + hi : s1;
      lds_dma16(gout + sgm * N + 4 * n4, GO);
      lds_dma16(arg + sgm * N + 4 * n4, reinterpret_cast<float*>(AR));
    }
  };
  auto issue_x = [&](int64_t t, int rid_l) {
    if (t >= ntiles) return;
    const int64_t row0 = r0 + t * TR;
    if constexpr (S16) {
      const uint16_t* xb = reinterpret_cast<const uint16_t*>(xprev) + (POOLED ? 0 : row0 * K);
      const int x16 = ln / KC16, q16 = ln % KC16;
#pragma unroll
      for (int i = 0; i < XI16; ++i)
        lds_dma16(xb + (int64_t)pick_lane<RX16>(rid_l, RX16 * i, x16) * K + 8 * q16,
                  XB + W_X / 2 + 256 * i);
    } else {
      const float* xb = POOLED ? xprev : xprev + row0 * K;
      const int hx = ln / KC, px = ln % KC;
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        const int rr = hx + RX * i;
        lds_dma16(xb + (int64_t)pickx(rid_l, RX * i) * K + 4 * (px ^ (4 * ((rr >> 2) & 1))), XB + 256 * i);
      }
    }
  };

  load_ids(wave);
  int rid_l = rid_n, seg_l = seg_n;
  issue_h(wave, rid_l, seg_l);
  issue_x(wave, rid_l);
  load_ids(wave + nwaves);
  for (int64_t t = wave; t < ntiles; t += nwaves) {
    const int64_t row0 = r0 + t * TR;
    const int cnt = (int)((r1 - row0) < TR ? (r1 - row0) : TR);
    ln = opaque(ln);
    const int hi = ln / NC, n4 = ln % NC, hx = ln / KC, px = ln % KC, g = ln >> 4, c = ln & 15;
    // dense gy: this tile's rows into registers, in flight together with the DMA
    float4 gyv[POOLED ? 1 : HI];
    if constexpr (!POOLED) {
#pragma unroll
      for (int i = 0; i < HI; ++i) {
        const int rr = hi + RH * i;
        gyv[i] = (rr < cnt) ? *reinterpret_cast<const float4*>(gy + (row0 + rr) * N + 4 * n4)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    wait_vm0();
    lds_order();
    if constexpr (S16) {
      // widen the x rows: lane (rr, px) owns the f32 slot at position px of row rr, which holds
      // column chunk px ^ swizzle(rr) (the layout the f32 DMA produces)
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        const int rr = hx + RX * i;
        const int cx = px ^ (4 * ((rr >> 2) & 1));
        const uint2 u = *reinterpret_cast<const uint2*>(
            reinterpret_cast<const uint16_t*>(XB + W_X / 2) + rr * K + 4 * cx);
        *reinterpret_cast<float4*>(XB + rr * K + 4 * px) =
            make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                        __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
      }
    }
    // ---- gh = GraphNorm-backward(gy, h), in place --------------------------------------------
    if constexpr (!(SPT_FDMA_SKIP & 1)) {
      const int s0 = POOLED ? __builtin_amdgcn_readlane(seg_l, 0) : 0;
      const int s1 = POOLED ? __builtin_amdgcn_readfirstlane(__shfl(seg_l, cnt - 1, 64)) : 0;
      const bool in_lds = s1 - s0 < NSEG;          // all of the tile's segments were DMA'd
#pragma unroll
      for (int i = 0; i < HI; ++i) {
        const int rr = hi + RH * i;
        float* slot = HB + rr * N + 4 * (n4 ^ (rr & 7));
        float4 hv;
        if constexpr (S16) {                       // the row's bf16 values, upper half of the buffer
          const uint2 u = *reinterpret_cast<const uint2*>(
              reinterpret_cast<const uint16_t*>(HB + W_H / 2) + rr * N + 4 * n4);
          hv = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                           __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
        } else {
          hv = *reinterpret_cast<const float4*>(slot);
        }
        float4 gv;
        if constexpr (POOLED) {
          const int rid = pick(rid_l, RH * i);
          const int sg = pick(seg_l, RH * i);
          int4 av;
          if (in_lds) {
            gv = *reinterpret_cast<const float4*>(GO + (sg - s0) * N + 4 * n4);
            av = *reinterpret_cast<const int4*>(AR + (sg - s0) * N + 4 * n4);
          } else {                                   // a tile over more than NSEG segments
            gv = *reinterpret_cast<const float4*>(gout + (int64_t)sg * N + 4 * n4);
            av = *reinterpret_cast<const int4*>(arg + (int64_t)sg * N + 4 * n4);
          }
          gv = make_float4(av.x == rid ? gv.x : 0.f, av.y == rid ? gv.y : 0.f,
                           av.z == rid ? gv.z : 0.f, av.w == rid ? gv.w : 0.f);
        } else {
          gv = gyv[i];
        }
        *reinterpret_cast<float4*>(slot) = gh_of(hv, gv);
      }
      if (cnt < TR) {                               // the last tile: rows past the end count as zero
#pragma unroll
        for (int i = 0; i < HI; ++i) {
          const int rr = hi + RH * i;
          if (rr >= cnt)
            *reinterpret_cast<float4*>(HB + rr * N + 4 * (n4 ^ (rr & 7))) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
    lds_order();
    // ---- gW += gh^T y_prev --------------------------------------------------------------------
    if constexpr (!(SPT_FDMA_SKIP & 2)) {
      bf16x4 Xh[KB], Xl[KB];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int k = 16 * kb + c;
        float xv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = XB[(4 * g + r) * K + 4 * ((4 * kb + (c >> 2)) ^ (4 * (g & 1))) + (c & 3)];
          if (pre) {
            v = fmaf(v - pt[k], pt[K + k], pt[2 * K + k]);
            v = (v > 0.f) ? v : v * pslope;
          }
          xv[r] = v;                                  // (rows past the end meet gh = 0)
        }
        split_bf16<4>(xv, Xh[kb], Xl[kb]);
      }
#pragma unroll
      for (int nb = 0; nb < NBK; ++nb) {
        float gv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
          gv[r] = HB[(4 * g + r) * N + 4 * ((4 * nb + (c >> 2)) ^ (4 * (g & 1) + r)) + (c & 3)];
        bf16x4 gh4, gl4;
        split_bf16<4>(gv, gh4, gl4);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) C3[nb][kb] = mfma3_16<LO>(gh4, gl4, Xh[kb], Xl[kb], C3[nb][kb]);
      }
    }
    // ---- gx = gh W -----------------------------------------------------------------------------
    f32x4 CX[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) CX[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (!(SPT_FDMA_SKIP & 4)) {
#pragma unroll
      for (int sg = 0; sg < NS; ++sg) {
        const float4 a0 = *reinterpret_cast<const float4*>(HB + c * N + 4 * ((8 * sg + 2 * g) ^ (c & 7)));
        const float4 a1 = *reinterpret_cast<const float4*>(HB + c * N + 4 * ((8 * sg + 2 * g + 1) ^ (c & 7)));
        const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        bf16x8 ah, alo;
        split_bf16<8>(av, ah, alo);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          const bf16x8 bh = *reinterpret_cast<const bf16x8*>(wt_hi + (16 * kb + c) * LDT + 32 * sg + 8 * g);
          bf16x8 bl = bh;
          if constexpr (LO) bl = *reinterpret_cast<const bf16x8*>(wt_lo + (16 * kb + c) * LDT + 32 * sg + 8 * g);
          CX[kb] = mfma3_32<LO>(ah, alo, bh, bl, CX[kb]);
        }
      }
    }
    // the gh buffer is free: the next tile's h (and gout / arg) rows start travelling now
    const int rid_cur = rid_l;
    wait_lds();
    issue_h(t + nwaves, rid_n, seg_n);
    // ---- statistics of the previous GraphNorm's backward; gx through the x buffer ------------
    // (f64 per element, like the register-staged kernel: sum g' and sum g' o' cancel to a small
    //  fraction of their terms - f32 partial sums over the 4 rows of a lane moved the input
    //  gradient three layers down by more than 1e-5 of its entries)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const int k = 16 * kb + c;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* xs = XB + (4 * g + r) * K + 4 * ((4 * kb + (c >> 2)) ^ (4 * (g & 1))) + (c & 3);
        const float v = CX[kb][r];
        if (!(SPT_FDMA_SKIP & 8) && pre) {          // (rows past the end: gh = 0 => v = 0)
          const float o = *xs - pt[k];
          float gg = v;
          if (pslope != 1.f) {
            const float y = fmaf(o, pt[K + k], pt[2 * K + k]);
            gg = (y > 0.f) ? gg : gg * pslope;
          }
          p1[kb] += (double)gg;
          p2[kb] += (double)gg * (double)o;
        }
        *xs = v;
      }
    }
    lds_order();
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int rr = hx + RX * i;
      const float4 v = *reinterpret_cast<const float4*>(XB + rr * K + 4 * px);
      const int64_t orow = POOLED ? (int64_t)pickx(rid_cur, RX * i) : row0 + rr;
      if (!(SPT_FDMA_SKIP & 16) && rr < cnt)
        *reinterpret_cast<float4*>(gx + orow * K + 4 * (px ^ (4 * ((rr >> 2) & 1)))) = v;
    }
    wait_lds();
    issue_x(t + nwaves, rid_n);
    rid_l = rid_n;
    seg_l = seg_n;
    load_ids(t + 2 * nwaves);
  }
  wait_vm0();
  // Round 6: ONE weight-gradient table per workgroup - waves 1 .. NW - 1 hand theirs to wave 0
  // through the (now free) tile buffers, one after the other (plain stores, plain loads + adds; the
  // order of fused_mlp.hip's bwd_kernel_bf, whose tables this kernel's equal bit for bit)
  {
    constexpr int LDR = K + 4;                    // row stride = 4 (mod 16) floats: no bank conflicts
    static_assert(N * LDR <= NW * (W_H + W_X + 2 * W_G), "the table fits the tile buffers");
    __syncthreads();                              // every wave is through its tiles
    float* red = &lw[0][0];
    for (int w = 1; w < NW; ++w) {
      if (wid == w) {
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
          for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(16 * nb + 4 * g + r) * LDR + 16 * kb + c] = C3[nb][kb][r];
      }
      __syncthreads();
      if (wid == 0) {
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
          for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) C3[nb][kb][r] += red[(16 * nb + 4 * g + r) * LDR + 16 * kb + c];
      }
      __syncthreads();
    }
  }
  if (wid == 0) {
    float* gwp = gw_partial + (size_t)blockIdx.x * N * K;
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          gwp[(size_t)(16 * nb + 4 * g + r) * K + 16 * kb + c] = C3[nb][kb][r];
  }
  if (pstat_partial) {
    double* pp = pstat_partial + (size_t)wave * (2 * K + 1);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const double a = xg_sum_d(p1[kb]), b = xg_sum_d(p2[kb]);
      if (g == 0) {
        pp[16 * kb + c] = a;
        pp[K + 16 * kb + c] = b;
      }
    }
    if (lane == 0) pp[2 * K] = (wave == 0) ? (double)(r1 - r0) : 0.0;
  }
}

}  // namespace fdma

// (K, N) with a DMA-staged backward: the point MLP's 64 -> 128, 32 -> 64 and (panoptic) 64 -> 64
bool fmlp_dma_supported(int K, int N) {
  return (K == 64 && N == 128) || (K == 32 && N == 64) || (K == 64 && N == 64);
}

// Launches the layer's backward; returns the number of per-wave partial tables written
// (gw_partial: [workgroups][N x K], pstat_partial: [waves][2 K + 1] or null), 0 if (K, N) is not built.
// returns the number of wave records PER RUN written to the statistics tables (0: shape not built);
// *gw_tabs = the weight-gradient tables per run (one per workgroup)
int fmlp_dma_bwd_launch(bool pooled, bool lo, const float* gy, const float* h, FmlpRuns rt,
                        int64_t max_rows, int N, const float* am, const float* sc, const float* bs,
                        float slope, const float* c1, const float* c2, const float* c3,
                        const float* xprev, int K, const float* pam, const float* psc,
                        const float* pbs, float pslope, const float* W, float* gx,
                        float* gw_partial, double* pstat_partial, const int32_t* perm,
                        const int32_t* pos_seg, const float* gout, const int32_t* arg,
                        hipStream_t stream, bool s16, int* gw_tabs) {
  using namespace fdma;
  const int64_t tiles = (max_rows + TR - 1) / TR;
  const int nr = rt.n < 1 ? 1 : rt.n;
#define SPT_DMA_CASE(KK, NN, NWV, PER_CU, OCC)                                                        \
  if (K == KK && N == NN) {                                                                        \
    int64_t blocks = (tiles + NWV - 1) / NWV;                                                      \
    /* one resident round of workgroups over all runs together */                                  \
    const int64_t cap = (256 * PER_CU) / nr > 1 ? (256 * PER_CU) / nr : 1;                         \
    if (blocks > cap) blocks = cap;                                                                \
    if (blocks < 1) blocks = 1;                                                                    \
    const dim3 grid((unsigned)blocks, (unsigned)nr);                                               \
    if (s16 && pooled)                                                                             \
      bwd_dma_kernel<KK, NN, NWV, OCC, false, true, true><<<grid, NWV * 64, 0, stream>>>(          \
          gy, h, 0, 0, am, sc, bs, slope, c1, c2, c3, xprev, pam, psc, pbs, pslope, W, gx,         \
          gw_partial, pstat_partial, perm, pos_seg, gout, arg, rt);                                \
    else if (s16)                                                                                  \
      bwd_dma_kernel<KK, NN, NWV, OCC, false, false, true><<<grid, NWV * 64, 0, stream>>>(         \
          gy, h, 0, 0, am, sc, bs, slope, c1, c2, c3, xprev, pam, psc, pbs, pslope, W, gx,         \
          gw_partial, pstat_partial, perm, pos_seg, gout, arg, rt);                                \
    else if (pooled && lo)                                                                              \
      bwd_dma_kernel<KK, NN, NWV, OCC, true, true><<<grid, NWV * 64, 0, stream>>>(                      \
          gy, h, 0, 0, am, sc, bs, slope, c1, c2, c3, xprev, pam, psc, pbs, pslope, W, gx,         \
          gw_partial, pstat_partial, perm, pos_seg, gout, arg, rt);                                \
    else if (pooled)                                                                               \
      bwd_dma_kernel<KK, NN, NWV, OCC, false, true><<<grid, NWV * 64, 0, stream>>>(                     \
          gy, h, 0, 0, am, sc, bs, slope, c1, c2, c3, xprev, pam, psc, pbs, pslope, W, gx,         \
          gw_partial, pstat_partial, perm, pos_seg, gout, arg, rt);                                \
    else if (lo)                                                                                   \
      bwd_dma_kernel<KK, NN, NWV, OCC, true, false><<<grid, NWV * 64, 0, stream>>>(                     \
          gy, h, 0, 0, am, sc, bs, slope, c1, c2, c3, xprev, pam, psc, pbs, pslope, W, gx,         \
          gw_partial, pstat_partial, perm, pos_seg, gout, arg, rt);                                \
    else                                                                                           \
      bwd_dma_kernel<KK, NN, NWV, OCC, false, false><<<grid, NWV * 64, 0, stream>>>(                    \
          gy, h, 0, 0, am, sc, bs, slope, c1, c2, c3, xprev, pam, psc, pbs, pslope, W, gx,         \
          gw_partial, pstat_partial, perm, pos_seg, gout, arg, rt);                                \
    if (gw_tabs) *gw_tabs = (int)blocks;                                                           \
    return (int)blocks * NWV;                                                                      \
  }
  SPT_DMA_CASE(64, 128, 8, 1, 2)
  SPT_DMA_CASE(64, 64, 8, 1, 2)
  SPT_DMA_CASE(32, 64, 8, 2, 4)
#undef SPT_DMA_CASE
  return 0;
}

}  // namespace spt
