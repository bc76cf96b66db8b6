// Edge-lane backward of the fused edge attention in TARGET order ("TO"), SPT-64 head layout
// (H = 16, qk_dim = 4, value dim = 4, in_rpe_dim = 32).  Same math and the same machinery as
// edge_attn_el.hip (autograd of src/nn/attention.py:202-315: 16-edge tiles, a pair of waves per
// tile stream with half of the heads each, transposed recompute GEMM, matrix-pipe segmented
// reductions, LDS-DMA for everything that is gathered, counted waits) - with the edge stream sorted
// by TARGET node instead of by source.
//
// Why.  The source-ordered kernel is HBM-bound on its own traffic (14.65 GB per level-1 call at
// ~5 TB/s); half of that is the per-edge [dk | dv] rows (512 B) it streams out and
// attn_kv_reduce_kernel reads back, because dk / dv are sums over the edges INTO a node while the
// tiles walk the edges OUT OF a node.  Walking the edges in target order swaps the roles:
//   * k / v of the edge's target are the same row for a run of consecutive edges: plain register
//     loads that hit the L1 (what the source's q / gout / softmax state were before);
//   * dk / dv are reduced per target INSIDE the tile on the matrix pipe (the S-matrix product the
//     source-ordered kernel uses for dq) and reach gqkv as a few atomic rows per tile;
//   * dq (64 floats per edge, half of [dk | dv]) becomes the streamed quantity: each edge's row is
//     written at the edge's SOURCE-order position (a whole 128-byte line per wave and edge), so
//     attn_q_reduce_kernel sums contiguous runs of rows per source - a pure stream, no index;
//   * the source's record (q * scale | gout | (delta, ml): 640 B per node, packed by the prep
//     kernel) becomes the gathered operand, 1.25x the bytes of the k / v gather.
// Per edge: 1 024 B of [dk | dv] round trip -> 512 B of dq round trip, 512 B -> 640 B of gathers:
// -2.7 GB of 14.65 per level-1 call.  Results agree with the source-ordered kernel to summation
// order (dq is now summed in a fixed order, dk / dv by atomics - the mirror image of before).
#include <math.h>
#include <stdlib.h>

#include "common.hpp"

namespace spt {
namespace to {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int TE = 16;            // edges per tile
constexpr int F = 32;             // in_rpe_dim
constexpr int NBW = 2;            // 16-column blocks of each projection owned by a wave
constexpr int WAVES = 8;          // per workgroup: 4 tile streams x 2 head halves, one workgroup per CU
constexpr int LD = 192;           // qkv row length
constexpr int REC = 160;          // floats of a node's record: 2 head halves x [qs 32 | gout 32 | dm 16]
constexpr int REC16 = 96;         // the bf16 mode's record, in floats: 2 head halves x [qs 32 bf16 | gout 32 bf16 | dm 16 f32]
constexpr int IDS = 64;           // ints of a tile record: edge rows | targets | sources | source-order positions

// LDS map, in floats.  Per wave:
constexpr int L_G = 0;                         // [16 edges][16 chunks]: (qs | gout) of the edge's source, chunk ^ edge
constexpr int L_GD = L_G + TE * 64;            // [16 edges][4 chunks]: (delta, ml) of the source, chunk + edge / 4
constexpr int DT_LD = 36;                      // 32 words + 4: conflict-free both ways
constexpr int L_DT = L_GD + TE * 16;           // [16 edges][DT_LD] words (hi << 16 | lo), one projection
constexpr int L_TBL = L_DT + TE * DT_LD;       // rank[16], node[16]
constexpr int L_END = L_TBL + 32;
// per pair of waves (the two head halves of one tile stream):
constexpr int P_EA = 0;                        // 2 x [16][32] edge_attr rows (16-B chunks XOR-swizzled)
constexpr int P_IDS = P_EA + 2 * TE * F;       // 4 slots of tile records
constexpr int P_MB = P_IDS + 4 * IDS;          // [64 lanes][8]: the leader's half of d edge_attr
constexpr int P_GEA = P_MB + 512;              // [2][64 lanes][4]: what gedge_attr holds for the tile (accumulate)
constexpr int P_FLAG = P_GEA + 512;            // hand-shake counters (F_*)
constexpr int P_END = P_FLAG + 8;
constexpr int F_EA = 0;       // leader -> follower: edge_attr rows of tile k and ids of tile k + 1 landed (k + 1)
constexpr int F_TOP = 1;      // follower -> leader: operands of tile k read (k + 1)
constexpr int F_MB = 2;       // leader -> follower: mailbox holds tile k (k + 1)
constexpr int F_MBFREE = 3;   // follower -> leader: mailbox of tile k consumed (k + 1)
constexpr int F_DONE = 4;     // follower -> leader: tile k finished, its id slot is free (k + 1)
constexpr int WB_LD = F + 8;
constexpr int WB_ELEMS = 192 * WB_LD;

// Outstanding VMEM operations per iteration, in issue order:
//   leader  : [top] ids of tile k+2 (1), k / v rows of tile k's targets (4), edge_attr rows of tile k+1 (2)
//             [core] dq rows (2)  [mid] source records of tile k+1 (5)  [tail] dk / dv atomics (0..16)
//   follower: [top] k / v rows (4), old d edge_attr rows of tile k (2; accumulating calls only)
//             [core] dq rows (2)  [mid] source records of tile k+1 (5)
//             [tail] d edge_attr rows (2), dk / dv atomics (0..16)
// measurement builds (tools/build_variant.sh to_skip<bits> edge_attn_to.hip ... -DSPT_TO_SKIP=<bits>,
// tools/to_variants.sh; the results are wrong by design, the default 0 changes nothing):
// 1 no dk / dv atomics, 2 no target rows, 4 no dq stores, 8 no record gathers, 16 no d edge_attr stores
#ifndef SPT_TO_SKIP
#define SPT_TO_SKIP 0
#endif
constexpr int N_DQ = (SPT_TO_SKIP & 4) ? 0 : 2;   // dq stores per tile and wave (f32 rows; bf16 rows: 1)
constexpr int N_GATHER = (SPT_TO_SKIP & 8) ? 0 : 5;

__device__ __forceinline__ void lds_dma16(const float* g, float* lds) {
  const unsigned a = __builtin_amdgcn_readfirstlane(
      (unsigned)(size_t)((__attribute__((address_space(3))) void*)lds));
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
               :: "s"(a), "v"(g) : "memory");
}
__device__ __forceinline__ void lds_dma4(const void* g, float* lds) {
  const unsigned a = __builtin_amdgcn_readfirstlane(
      (unsigned)(size_t)((__attribute__((address_space(3))) void*)lds));
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off"
               :: "s"(a), "v"(g) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void wait_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
typedef __attribute__((address_space(3))) volatile int lds_flag_t;   // ds_read / ds_write, never flat
__device__ __forceinline__ void flag_set(lds_flag_t* f, int v) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  *f = v;
}
__device__ __forceinline__ void flag_wait(lds_flag_t* f, int v) {
  int spins = 0;
  while (__builtin_amdgcn_readfirstlane(*f) < v) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1 << 22)) __builtin_trap();   // a lost partner: fail loudly, never hang the GPU
  }
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void lds_order() {   // this wave's LDS writes before its later reads
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ float xor16(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float((threadIdx.x & 16) ? r[0] : r[1]);
}
__device__ __forceinline__ float xor32(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}
__device__ __forceinline__ float xg_sum(float v) {  // sum over the 4 lane groups g
  v += xor16(v);
  v += xor32(v);
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
__device__ __forceinline__ float qk_scale_of(int mode, float a, int deg) {
  const float g = __builtin_amdgcn_rsqf((float)deg);   // v_rsq_f32 (1 ulp; the same in every attention kernel)
  if (mode == 0) return a * g;
  if (mode == 1) return a + g;
  return a;
}

// ---- per-node record the backward gathers per edge (the edge's SOURCE) --------------------------
//   rec[node][hh][0..31]  = q * qk-scale, heads 8 hh .. 8 hh + 7 (head 4 b + g at 16 bl + 4 g, b = 2 hh + bl)
//   rec[node][hh][32..63] = gout, same columns
//   rec[node][hh][64 + 4 g + {0, 1, 2, 3}] = delta(bl = 0), delta(bl = 1), ml(bl = 0), ml(bl = 1)
//     delta = <gout, out> of the head, ml = m + log(z + 1e-16) (softmax weight = exp(p - ml))
//   scl[node] = the node's qk scale (0 for a node without edges)
// One thread per (node, head).  Also zero-fills the k / v columns of gqkv (dk / dv arrive by atomics).
//   B16 (the bf16 mode, round 6): q * scale and gout as bf16 - 384 bytes per node instead of 640,
//   gathered once per edge: rec[node][hh] = [qs 32 bf16 | gout 32 bf16 | dm 16 f32] (192 bytes);
//   (delta, ml) stay f32 (the softmax weight is exp(p - ml)).
template <bool B16>
__global__ __launch_bounds__(256) void attn_bwd_to_prep_kernel(
    const float* __restrict__ qkv, const float* __restrict__ gout, const float* __restrict__ out,
    const float* __restrict__ m, const float* __restrict__ z, const int32_t* __restrict__ erowptr,
    int64_t N, int scale_mode, float scale_a, float* __restrict__ rec, float* __restrict__ scl,
    float* __restrict__ gqkv) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t node = idx >> 4;
  const int h = (int)(idx & 15);
  if (node >= N) return;
  const float4 g4 = *reinterpret_cast<const float4*>(gout + node * 64 + 4 * h);
  const float4 o4 = *reinterpret_cast<const float4*>(out + node * 64 + 4 * h);
  const float delta = (g4.x * o4.x + g4.y * o4.y) + (g4.z * o4.z + g4.w * o4.w);
  const float ml = m[node * 16 + h] + __logf(z[node * 16 + h] + 1e-16f);
  const int b = h >> 2, g = h & 3, hh = b >> 1, bl = b & 1;
  const int deg = erowptr[node + 1] - erowptr[node];
  const float scale = deg > 0 ? qk_scale_of(scale_mode, scale_a, deg) : 0.f;
  const float4 q4 = *reinterpret_cast<const float4*>(qkv + node * LD + 4 * h);
  if constexpr (B16) {
    float* r = rec + node * REC16 + hh * 48;
    auto pk = [](float a, float b2) {
      const __bf16 ha = (__bf16)a, hb = (__bf16)b2;
      return (unsigned)__builtin_bit_cast(unsigned short, ha) |
             ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
    };
    unsigned* r16 = reinterpret_cast<unsigned*>(r);          // 2 bf16 per word
    *reinterpret_cast<u32x2*>(r16 + (16 * bl + 4 * g) / 2) =
        (u32x2){pk(q4.x * scale, q4.y * scale), pk(q4.z * scale, q4.w * scale)};
    *reinterpret_cast<u32x2*>(r16 + 16 + (16 * bl + 4 * g) / 2) = (u32x2){pk(g4.x, g4.y), pk(g4.z, g4.w)};
    r[32 + 4 * g + bl] = delta;
    r[32 + 4 * g + 2 + bl] = ml;
  } else {
    float* r = rec + node * REC + hh * 80;
    *reinterpret_cast<float4*>(r + 16 * bl + 4 * g) =
        make_float4(q4.x * scale, q4.y * scale, q4.z * scale, q4.w * scale);
    *reinterpret_cast<float4*>(r + 32 + 16 * bl + 4 * g) = g4;
    r[64 + 4 * g + bl] = delta;
    r[64 + 4 * g + 2 + bl] = ml;
  }
  if (h == 0) scl[node] = scale;
  *reinterpret_cast<float4*>(gqkv + node * LD + 64 + 4 * h) = make_float4(0.f, 0.f, 0.f, 0.f);
  *reinterpret_cast<float4*>(gqkv + node * LD + 128 + 4 * h) = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ids of a tile of 16 consecutive TARGET-order positions in one 256-byte record:
// [16 edge rows | 16 targets | 16 sources | 16 source-order positions]; target-order position j is
// source-order position tperm[j] (tperm = stable argsort of the targets over the source-order
// positions); positions beyond the edge list repeat the last edge (they run with softmax weight 0).
__global__ __launch_bounds__(256) void pack_tile_ids_to_kernel(
    const int32_t* __restrict__ eperm, const int32_t* __restrict__ tgt,
    const int32_t* __restrict__ src, const int32_t* __restrict__ tperm, int64_t E, int64_t ntiles,
    int32_t* __restrict__ ids4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= ntiles * IDS) return;
  const int64_t tile = i / IDS;
  const int l = (int)(i - tile * IDS);
  int64_t j = tile * TE + (l & 15);
  j = j < E ? j : E - 1;
  const int32_t sp = tperm[j];
  const int f = l >> 4;
  ids4[i] = f == 0 ? (eperm ? eperm[sp] : sp) : (f == 1 ? tgt[sp] : (f == 2 ? src[sp] : sp));
}

// ---- the by-target stream WITHOUT a second sort (round 6) ---------------------------------------
// The reference's final edge list is [i<j | j>i | loops] (src/transforms/graph.py:1268, 1442-1446:
// OnTheFlyHorizontalEdgeFeatures appends the flipped copy of the trimmed list, NAGAddSelfLoops the
// loops; the S3DIS / DALES / ScanNet configs ship `sample_edge_n_min: -1`, so nothing thins it
// afterwards): edge e < M has its mirror at e + M, a loop is its own mirror.  The edges INTO node t
// are then the mirrors of the edges OUT OF t, which the by-source view already holds as one
// contiguous run: walking the by-source positions p = 0 .. E - 1 and taking mirror(eperm[p]) IS a
// stream grouped by target (ascending), with
//     edge row = mirror(eperm[p]),  target = src_sorted[p],  source = tgt_sorted[p],
//     source-order position = inv[mirror(eperm[p])]        (inv = inverse of eperm).
// mirror_prepare_kernel writes inv and CHECKS the structure the caller declared (a flag word:
// bit 0 = a pair (e, e + M) that is not (s, t) / (t, s), bit 1 = a "loop" with s != t);
// pack_tile_ids_mirror_kernel writes the 64-int tile records from it.  No radix sort of the
// targets, no [E] int64 cast, no row-pointer pass.
__device__ __forceinline__ int64_t mirror_edge(int64_t e, int64_t M) {
  return e < M ? e + M : (e < 2 * M ? e - M : e);
}

__global__ __launch_bounds__(256) void mirror_prepare_kernel(
    const int64_t* __restrict__ ei, const int32_t* __restrict__ eperm, int64_t E, int64_t M,
    int32_t* __restrict__ inv, int32_t* __restrict__ flag) {
  int bad = 0;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < E; e += (int64_t)gridDim.x * 256) {
    inv[eperm[e]] = (int32_t)e;
    const int64_t s = ei[e], t = ei[E + e];
    if (e < M) {
      if (ei[e + M] != t || ei[E + e + M] != s) bad |= 1;
    } else if (e >= 2 * M) {
      if (s != t) bad |= 2;
    }
  }
  if (bad) atomicOr(flag, bad);
}

__global__ __launch_bounds__(256) void pack_tile_ids_mirror_kernel(
    const int32_t* __restrict__ eperm, const int32_t* __restrict__ tgt,
    const int32_t* __restrict__ src, const int32_t* __restrict__ inv, int64_t E, int64_t M,
    int64_t ntiles, int32_t* __restrict__ ids4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= ntiles * IDS) return;
  const int64_t tile = i / IDS;
  const int l = (int)(i - tile * IDS);
  int64_t p = tile * TE + (l & 15);
  p = p < E ? p : E - 1;
  const int f = l >> 4;
  if (f == 1) {
    ids4[i] = src[p];                        // the mirror edge points INTO the source of position p
  } else if (f == 2) {
    ids4[i] = tgt[p];
  } else {
    const int64_t em = mirror_edge(eperm[p], M);
    ids4[i] = f == 0 ? (int32_t)em : inv[em];
  }
}

// gqkv[s][0 .. 63] = qk-scale(s) * sum over the edges OUT OF s of their dq rows: the rows of a
// source are the contiguous run [erowptr[s], erowptr[s + 1]) of the temporary (the main kernel
// writes every edge's row at its source-order position), summed in ascending order (deterministic).
// A quarter wave per node: 16 lanes x 16 bytes = one 256-byte row per load, eight rows in flight.
// B16 (the bf16 mode): the rows hold bf16 values (128 bytes per edge), 8 bytes per lane.
template <bool B16>
__device__ __forceinline__ f32x4 dq_row_chunk(const float* __restrict__ dqt, int64_t row, int l16) {
  if constexpr (B16) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 u = __builtin_nontemporal_load(
        reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(dqt) + row * 64 + 4 * l16));
    return (f32x4){__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                   __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
  } else {
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dqt + row * 64 + 4 * l16));
  }
}
template <bool B16>
__global__ __launch_bounds__(256) void attn_q_reduce_kernel(
    const float* __restrict__ dqt, const int32_t* __restrict__ erowptr,
    const float* __restrict__ scl, int64_t N, float* __restrict__ gqkv) {
  const int l16 = threadIdx.x & 15;
  const int64_t grp = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int64_t ngrp = ((int64_t)gridDim.x * 256) >> 4;
  for (int64_t s = grp; s < N; s += ngrp) {
    const int a = erowptr[s], b = erowptr[s + 1];
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    int u = a;
    for (; u + 8 <= b; u += 8) {
      f32x4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = dq_row_chunk<B16>(dqt, (int64_t)(u + i), l16);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += v[i];
    }
    if (u < b) {
      f32x4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int64_t j = u + i < b ? u + i : b - 1;
        v[i] = dq_row_chunk<B16>(dqt, j, l16);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (u + i < b) acc += v[i];
      }
    }
    *reinterpret_cast<f32x4*>(gqkv + s * LD + 4 * l16) = acc * scl[s];
  }
}

template <int PREC>
__global__ __launch_bounds__(WAVES * 64, 2) void attn_bwd_to_kernel(
    const float* __restrict__ qkv, int64_t E, const int32_t* __restrict__ ids4, int64_t ntiles,
    int64_t tpw, const float* __restrict__ ea, const float* __restrict__ Wk,
    const float* __restrict__ bk, const float* __restrict__ Wq, const float* __restrict__ bq,
    const float* __restrict__ Wv, const float* __restrict__ bv, const float* __restrict__ rec,
    float* __restrict__ gqkv, float* __restrict__ gea, int gea_acc, float* __restrict__ dqt,
    float* __restrict__ partial, int bands) {
  static_assert(PREC == 1 || PREC == 3, "bf16 matrix pipe only");
  constexpr bool LO = PREC == 3;
  // the bf16 mode streams its dq rows as bf16: 64 bytes per edge and wave, ONE store per tile
  constexpr bool DQ16 = PREC == 1;
  constexpr int NDQ = (SPT_TO_SKIP & 4) ? 0 : (DQ16 ? 1 : N_DQ);
  // ... and gathers its source records as bf16 (q * scale | gout: 128 bytes per edge and wave
  // instead of 256; (delta, ml) stay f32): three requests per tile instead of five
  constexpr bool R16 = PREC == 1;
  constexpr int NG = (SPT_TO_SKIP & 8) ? 0 : (R16 ? 3 : N_GATHER);
  __shared__ __attribute__((aligned(16))) float lds_wave[WAVES][L_END];
  __shared__ __attribute__((aligned(16))) float lds_pair[WAVES / 2][P_END];
  __shared__ __attribute__((aligned(16))) __bf16 wb_hi[WB_ELEMS];
  __shared__ __attribute__((aligned(16))) __bf16 wb_lo[LO ? WB_ELEMS : 8];
  __shared__ __attribute__((aligned(16))) float bias_lds[192];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  float* L = lds_wave[wid];
  float* P = lds_pair[wid >> 1];
  lds_flag_t* flg = (lds_flag_t*)(__attribute__((address_space(3))) void*)(P + P_FLAG);
  const int64_t wave = (int64_t)blockIdx.x * WAVES + wid;
  const int64_t pair = wave >> 1;
  const int hh = wid & 1;                   // head half: blocks b = 2 hh + bl of every projection
  const bool leader = hh == 0;              // leader: fetches the pair's edge_attr rows and tile ids;
                                            // follower: combines and writes the pair's d edge_attr
  const bool acc = (gea_acc & 1) != 0;
  // bit 1 (SPT_TO_READ_OLD=1, measurement switch): fetch what gedge_attr holds even when the call
  // does not accumulate (the behaviour before round 6)
  const bool rd_old = acc || (gea_acc & 2) != 0;

  // ---- operands ---------------------------------------------------------------------------
  auto Wof = [&](int p) { return p == 0 ? Wk : (p == 1 ? Wq : Wv); };
  for (int i = threadIdx.x; i < 64 * F; i += WAVES * 64) {
    const int n = i / F, f = i - n * F;
    const float w3[3] = {Wk[i], Wq[i], Wv[i]};
#pragma unroll
    for (int p3 = 0; p3 < 3; ++p3) {
      const __bf16 h = (__bf16)w3[p3];
      wb_hi[(64 * p3 + n) * WB_LD + f] = h;
      if constexpr (LO) wb_lo[(64 * p3 + n) * WB_LD + f] = (__bf16)(w3[p3] - (float)h);
    }
  }
  if (threadIdx.x < 192) {
    const int p3 = threadIdx.x >> 6, n = threadIdx.x & 63;
    const float* bp = p3 == 0 ? bk : (p3 == 1 ? bq : bv);
    bias_lds[threadIdx.x] = bp ? bp[n] : 0.f;
  }
  if (lane < 8) flg[lane] = 0;
  __syncthreads();
  bf16x8 Wbh[3][2], Wbl[3][2];
#pragma unroll
  for (int p = 0; p < 3; ++p) {
#pragma unroll
    for (int fb = 0; fb < 2; ++fb) {
      float w[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int o = 16 * (2 * hh + (i >> 2)) + 4 * g + (i & 3);
        w[i] = Wof(p)[(size_t)o * F + 16 * fb + c];
      }
      split_bf16<8>(w, Wbh[p][fb], Wbl[p][fb]);
    }
  }
  f32x4 C3[3 * NBW][2];
  float gb[3 * NBW];
#pragma unroll
  for (int q6 = 0; q6 < 3 * NBW; ++q6) {
    C3[q6][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    C3[q6][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    gb[q6] = 0.f;
  }

  // The pair's tiles: a contiguous range of the target-order stream (bands == 0), or - round 6,
  // XCD bands - every (pairs per XCD)-th tile of the contiguous EIGHTH of the stream that belongs
  // to this workgroup's XCD (workgroup b runs on XCD b % 8: observed placement, only speed depends
  // on it).  The resident pairs of an XCD then sweep ONE window of consecutive targets together,
  // and the source records those targets' edges gather (the targets' spatial neighbours in the
  // MortonOrder layout) are fetched into that XCD's L2 once for all of them; with contiguous
  // ranges every pair sat in its own far-away part of the graph.  Same tiles, same per-tile
  // arithmetic; the sums over tiles (weight gradients, atomics) meet in another order.
  int64_t t_begin = pair * tpw, t_step = 1;
  int64_t t_end = (t_begin + tpw < ntiles) ? t_begin + tpw : ntiles;
  if (bands) {
    const int x = blockIdx.x & 7;
    const int64_t tpx = (ntiles + 7) >> 3;
    t_step = (int64_t)(gridDim.x >> 3) * (WAVES / 2);
    t_begin = x * tpx + (int64_t)(blockIdx.x >> 3) * (WAVES / 2) + (wid >> 1);
    t_end = (x + 1) * tpx < ntiles ? (x + 1) * tpx : ntiles;
  }
  if (t_begin < t_end) {
    int* ids_ring = reinterpret_cast<int*>(P + P_IDS);
    const int64_t t_last = t_begin + (t_end - 1 - t_begin) / t_step * t_step;
    auto issue_ids = [&](int64_t t, int slot) {        // one 256-byte record: 64 lanes x 4 bytes
      t = t < t_last ? t : t_last;
      lds_dma4(ids4 + t * IDS + lane, P + P_IDS + slot * IDS);
    };
    auto issue_ea = [&](int slot, int buf) {
      const int* ids = ids_ring + slot * IDS;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int u = p * 8 + (lane >> 3), ch = lane & 7;
        const int64_t e = ids[u];
        lds_dma16(ea + e * F + ((ch ^ (u & 7)) << 2), P + P_EA + buf * TE * F + p * 256);
      }
    };
    auto issue_old = [&](int slot) {                   // what gedge_attr holds for the tile's edges
      const int* ids = ids_ring + slot * IDS;
      const float* row = gea + (int64_t)ids[c] * F + 4 * g;
      lds_dma16(row, P + P_GEA);
      lds_dma16(row + 16, P + P_GEA + 256);
    };
    // the SOURCE records of the tile's edges, whole 128-byte lines per request: (qs | gout) as 16
    // chunks per edge at position chunk ^ edge (4 edges per instruction), (delta, ml) as 4 chunks
    // per edge at position (chunk + edge / 4) % 4 (16 edges in one instruction)
    auto issue_gather = [&](int slot) {
      if constexpr ((SPT_TO_SKIP & 8) != 0) return;
      const int* ids = ids_ring + slot * IDS;
      float* G = L + L_G;
      if constexpr (R16) {
        // (qs | gout) bf16: 8 chunks of 16 bytes per edge at position chunk ^ (edge % 8), 8 edges per
        // instruction; the second instruction's block starts 128 bytes further (edges e and e + 8
        // would otherwise sit on the same banks)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int e = 8 * j + (lane >> 3);
          const int ch = (lane & 7) ^ (e & 7);
          const int64_t sc = ids[32 + e];
          lds_dma16(rec + sc * REC16 + hh * 48 + 4 * ch, G + j * 288);
        }
        const int e = lane >> 2, x = ((lane & 3) - (e >> 2)) & 3;
        const int64_t sc = ids[32 + e];
        lds_dma16(rec + sc * REC16 + hh * 48 + 32 + 4 * x, L + L_GD);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int e = 4 * j + g;
          const int ch = c ^ e;
          const int64_t sc = ids[32 + e];
          lds_dma16(rec + sc * REC + hh * 80 + 4 * ch, G + j * 256);
        }
        {
          const int e = lane >> 2, x = ((lane & 3) - (e >> 2)) & 3;
          const int64_t sc = ids[32 + e];
          lds_dma16(rec + sc * REC + hh * 80 + 64 + 4 * x, L + L_GD);
        }
      }
    };
    // k / v rows of the tile's TARGETS, straight into registers (consecutive edges share their
    // target: L1 hits; asm: invisible to the compiler's wait-count pass, waited for by hand)
    f32x4 nk[NBW], nv[NBW];
    auto issue_node = [&](int slot) {
      if constexpr ((SPT_TO_SKIP & 2) != 0) return;
      const int* ids = ids_ring + slot * IDS;
      const int64_t tc = ids[16 + c];
      const float* kk = qkv + tc * LD + 64 + 32 * hh + 4 * g;
      const float* vv = qkv + tc * LD + 128 + 32 * hh + 4 * g;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(nk[0]) : "v"(kk) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(nk[1]) : "v"(kk) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(nv[0]) : "v"(vv) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(nv[1]) : "v"(vv) : "memory");
    };
#define SPT_TO_NODE_WAIT_SEL(SEL)                                                        \
  asm volatile("v_readfirstlane_b32 vcc_lo, %4\n\t"                                       \
               "s_cmp_lg_u32 vcc_lo, 0\n\t"                                              \
               "s_cbranch_scc1 1f\n\t"                                                   \
               "s_waitcnt vmcnt(0)\n\t"                                                  \
               "s_branch 2f\n"                                                           \
               "1:\n\t"                                                                  \
               "s_waitcnt vmcnt(2)\n"                                                    \
               "2:"                                                                      \
               : "+v"(nk[0]), "+v"(nk[1]), "+v"(nv[0]), "+v"(nv[1])                      \
               : "v"(SEL)                                                                \
               : "memory", "scc", "vcc")
    const int wait2 = (leader || rd_old) ? 1 : 0;

    // ---- prologue ------------------------------------------------------------------------------
    if (leader) {
      issue_ids(t_begin, 0);
      issue_ids(t_begin + t_step, 1);
      wait_vm<0>();
      issue_ea(0, 0);
      wait_vm<0>();
      flag_set(flg + F_EA, 1);
    } else {
      flag_wait(flg + F_EA, 1);
    }
    issue_gather(0);
    wait_vm<0>();

    int k = 0;                                    // tile index within the pair's range
    for (int64_t t = t_begin; t < t_end; t += t_step, ++k) {
      const int s0 = k & 3, s1 = (k + 1) & 3, s2 = (k + 2) & 3;   // id ring slots of tiles k, k+1, k+2
      if (leader) {
        // edge_attr rows of tile k and the ids of tile k + 1 have landed; behind them in the queue:
        // the dq rows of tile k - 1, the gathers of tile k, dk / dv atomics
        wait_vm<NDQ + NG>();
        flag_set(flg + F_EA, k + 1);
      } else {
        flag_wait(flg + F_EA, k + 1);
      }
      const float* slab = P + P_EA + (k & 1) * TE * F;
      bf16x8 Ah, Al;
      {
        const float4 a0 = *reinterpret_cast<const float4*>(slab + c * F + (((2 * g) ^ (c & 7)) << 2));
        const float4 a1 = *reinterpret_cast<const float4*>(slab + c * F + (((2 * g + 1) ^ (c & 7)) << 2));
        const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        split_bf16<8>(a, Ah, Al);
      }
      s16x4 Eh[2], El[2];
#pragma unroll
      for (int fb = 0; fb < 2; ++fb) {
        float ev[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int e = 4 * g + i, f = 16 * fb + c;
          ev[i] = slab[e * F + (((f >> 2) ^ (e & 7)) << 2) + (f & 3)];
        }
        bf16x4 eh, el;
        split_bf16<4>(ev, eh, el);
        Eh[fb] = __builtin_bit_cast(s16x4, eh);
        El[fb] = __builtin_bit_cast(s16x4, el);
      }
      const int* ids = ids_ring + s0 * IDS;
      const int t_c = ids[16 + c];                                    // target node of edge c
      wait_lds();                                 // the edge_attr slab is consumed
      // Both roles put exactly TWO requests behind the target rows (leader: the next tile's
      // edge_attr rows; follower of an accumulating call: what gedge_attr holds for this tile), so
      // that one counted wait serves both
      if (leader) {
        flag_wait(flg + F_DONE, k - 1);
        issue_ids(t + 2 * t_step, s2);
        issue_node(s0);
        flag_wait(flg + F_TOP, k);
        issue_ea(s1, (k + 1) & 1);
      } else {
        flag_set(flg + F_TOP, k + 1);
        issue_node(s0);
        // (round 6: only an accumulating call reads what gedge_attr holds - 128 bytes per edge
        // that the first block of a stage's backward used to fetch for nothing)
        if (rd_old) issue_old(s0);
      }

      // ---- recompute GEMM, transposed: C[o = 16 b + 4 g + r][e = c] ---------------------------
      f32x4 Ck[NBW], Cq[NBW], Cv[NBW];
      {
        const __bf16* wh = wb_hi + (32 * hh + c) * WB_LD + 8 * g;
        const __bf16* wl = wb_lo + (32 * hh + c) * WB_LD + 8 * g;
        const float* bi = bias_lds + 32 * hh + 4 * g;
#pragma unroll
        for (int bl = 0; bl < NBW; ++bl) {
          Ck[bl] = *reinterpret_cast<const f32x4*>(bi + 16 * bl);
          Cq[bl] = *reinterpret_cast<const f32x4*>(bi + 64 + 16 * bl);
          Cv[bl] = *reinterpret_cast<const f32x4*>(bi + 128 + 16 * bl);
          const bf16x8 kh = *reinterpret_cast<const bf16x8*>(wh + (16 * bl) * WB_LD);
          const bf16x8 qh = *reinterpret_cast<const bf16x8*>(wh + (64 + 16 * bl) * WB_LD);
          const bf16x8 vh = *reinterpret_cast<const bf16x8*>(wh + (128 + 16 * bl) * WB_LD);
          if constexpr (LO) {
            const bf16x8 kl = *reinterpret_cast<const bf16x8*>(wl + (16 * bl) * WB_LD);
            const bf16x8 ql = *reinterpret_cast<const bf16x8*>(wl + (64 + 16 * bl) * WB_LD);
            const bf16x8 vl = *reinterpret_cast<const bf16x8*>(wl + (128 + 16 * bl) * WB_LD);
            Ck[bl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl, Ah, Ck[bl], 0, 0, 0);
            Cq[bl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ql, Ah, Cq[bl], 0, 0, 0);
            Cv[bl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl, Ah, Cv[bl], 0, 0, 0);
            Ck[bl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, Al, Ck[bl], 0, 0, 0);
            Cq[bl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qh, Al, Cq[bl], 0, 0, 0);
            Cv[bl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, Al, Cv[bl], 0, 0, 0);
          }
          Ck[bl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, Ah, Ck[bl], 0, 0, 0);
          Cq[bl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qh, Ah, Cq[bl], 0, 0, 0);
          Cv[bl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, Ah, Cv[bl], 0, 0, 0);
        }
      }

      // ---- node ranks of the tile's edges (for the dk / dv reduction): rank = number of TARGET
      //      changes up to the edge; 16-lane scan on DPP row shifts --------------------------------
      int rank;
      {
        const int prev = __builtin_amdgcn_update_dpp(t_c, t_c, 0x111, 0xF, 0xF, false);
        rank = (t_c != prev) ? 1 : 0;
        rank += __builtin_amdgcn_update_dpp(0, rank, 0x111, 0xF, 0xF, false);
        rank += __builtin_amdgcn_update_dpp(0, rank, 0x112, 0xF, 0xF, false);
        rank += __builtin_amdgcn_update_dpp(0, rank, 0x114, 0xF, 0xF, false);
        rank += __builtin_amdgcn_update_dpp(0, rank, 0x118, 0xF, 0xF, false);
      }
      const int nn = __builtin_amdgcn_readlane(rank, 15) + 1;        // target nodes in the tile

      // ---- per-edge gradients, in place: Ck <- dk, Cq <- dq, Cv <- dv -------------------------
      // the gathered source records and the target rows of tile k have landed; behind them in the
      // queue: two requests of this tile's top (see there)
      // ONE asm statement for both counts (a follower that does not accumulate has nothing behind
      // its target rows): with the two waits in two branches the compiler copied the row registers
      // into the statement's operands BEFORE the wait of one branch - registers with loads pending
      SPT_TO_NODE_WAIT_SEL(wait2);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      {
        const float* G = L + L_G + c * 64;
        const f32x4 dmv = *reinterpret_cast<const f32x4*>(L + L_GD + (c * 4 + ((g + (c >> 2)) & 3)) * 4);
        const bool valid = t * TE + c < E;
        {
          int* tb = reinterpret_cast<int*>(L + L_TBL);
          tb[c] = rank;                           // every lane group writes the same values
          tb[16 + rank] = t_c;
        }
#pragma unroll
        for (int bl = 0; bl < NBW; ++bl) {
          const f32x4 kt = nk[bl];
          const f32x4 vt = nv[bl];
          f32x4 qr, gs;
          if constexpr (R16) {
            // edge c: block 8 (c / 8) x 128 B (+ 128 B for the second block), 16-byte chunk
            // 2 bl + g / 2 (qs) / 4 + 2 bl + g / 2 (gout) at position chunk ^ (c % 8), half g % 2
            const unsigned* Gb = reinterpret_cast<const unsigned*>(L + L_G) + (c >> 3) * 288 + (c & 7) * 32;
            const u32x2 qw = *reinterpret_cast<const u32x2*>(Gb + 4 * ((2 * bl + (g >> 1)) ^ (c & 7)) + 2 * (g & 1));
            const u32x2 gw = *reinterpret_cast<const u32x2*>(Gb + 4 * ((4 + 2 * bl + (g >> 1)) ^ (c & 7)) + 2 * (g & 1));
            qr = (f32x4){__uint_as_float(qw.x << 16), __uint_as_float(qw.x & 0xffff0000u),
                         __uint_as_float(qw.y << 16), __uint_as_float(qw.y & 0xffff0000u)};
            gs = (f32x4){__uint_as_float(gw.x << 16), __uint_as_float(gw.x & 0xffff0000u),
                         __uint_as_float(gw.y << 16), __uint_as_float(gw.y & 0xffff0000u)};
          } else {
            qr = *reinterpret_cast<const f32x4*>(G + 4 * ((4 * bl + g) ^ c));
            gs = *reinterpret_cast<const f32x4*>(G + 4 * ((8 + 4 * bl + g) ^ c));
          }
          float kk[4], q[4], v[4];
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            kk[d] = Ck[bl][d] + kt[d];
            q[d] = Cq[bl][d] + qr[d];
            v[d] = Cv[bl][d] + vt[d];
          }
          const float p = fmaf(q[3], kk[3], fmaf(q[2], kk[2], fmaf(q[1], kk[1], q[0] * kk[0])));
          const float a = valid ? __expf(p - dmv[2 + bl]) : 0.f;
          const float da = fmaf(gs[3], v[3], fmaf(gs[2], v[2], fmaf(gs[1], v[1], gs[0] * v[0])));
          const float dc = a * (da - dmv[bl]);
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            Ck[bl][d] = dc * q[d];
            Cq[bl][d] = dc * kk[d];
            Cv[bl][d] = a * gs[d];
          }
        }
        // dq rows of the tile's edges: each edge's row goes to its SOURCE-order position (a whole
        // 128-byte line per wave and edge).  Through the (consumed) gather buffer: [edge c][position
        // chunk ^ (c / 2) % 8] <- the lane's two chunks, read back as [edge 8 j + l / 8][position l % 8]
        wait_lds();                               // the gathered records are consumed
        float* Gq = L + L_G;
#pragma unroll
        for (int bl = 0; bl < NBW; ++bl)
          *reinterpret_cast<f32x4*>(Gq + c * 32 + 4 * ((4 * bl + g) ^ ((c >> 1) & 7))) = Cq[bl];
        lds_order();
        if constexpr (DQ16) {
          // lane -> edge lane / 4, columns 8 (lane % 4) .. + 7 of the wave's half row, rounded to bf16
          typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
          const int e = lane >> 2, cp = lane & 3, sw = (e >> 1) & 7;
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(Gq + e * 32 + 4 * ((2 * cp) ^ sw));
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(Gq + e * 32 + 4 * ((2 * cp + 1) ^ sw));
          auto pk = [](float a, float b) {
            const __bf16 ha = (__bf16)a, hb = (__bf16)b;
            return (unsigned int)__builtin_bit_cast(unsigned short, ha) |
                   ((unsigned int)__builtin_bit_cast(unsigned short, hb) << 16);
          };
          const u32x4 w4 = {pk(v0[0], v0[1]), pk(v0[2], v0[3]), pk(v1[0], v1[1]), pk(v1[2], v1[3])};
          const int64_t sp = ids[48 + e];
          if (t * TE + e < E && !(SPT_TO_SKIP & 4))
            __builtin_nontemporal_store(
                w4, reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(dqt) + sp * 64 + 32 * hh + 8 * cp));
        } else {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int e = 8 * j + (lane >> 3);
            const int x = (lane & 7) ^ ((e >> 1) & 7);
            const f32x4 v4 = *reinterpret_cast<const f32x4*>(Gq + j * 256 + lane * 4);
            const int64_t sp = ids[48 + e];
            // rows beyond the edge list own no row (masked per lane: the instruction still issues)
            if (t * TE + e < E && !(SPT_TO_SKIP & 4))
              __builtin_nontemporal_store(v4, reinterpret_cast<f32x4*>(dqt + sp * 64 + 32 * hh + 4 * x));
          }
        }
      }
      wait_lds();                                 // the transposition buffer is free again
      issue_gather(s1);                           // records of tile k + 1 (its ids landed at the top)

      // ---- per projection p (k, q, v): split D_p (B operand of d edge_attr^T += W_p^T D_p^T; one word
      //      per value -> LDS), read back transposed (A operand of dW_p += D_p^T EA, and B operand of
      //      the reduction per TARGET node of the tile on the matrix pipe: dk (p == 0) and dv (p == 2)
      //      of the nodes and, summed over the node rows, the bias gradient of every block) ---------
      f32x4 C2[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
      f32x4 CnK[NBW], CnV[NBW];                   // per node: Cn[n = 4 g + r][o = 16 b + c]
      i32x4 nd4;
      s16x4 S;                                    // S[n = c][e = 4 g + i] = 1 if edge e belongs to node n
      {
        const int* tb = reinterpret_cast<const int*>(L + L_TBL);
        const i32x4 rk4 = *reinterpret_cast<const i32x4*>(tb + 4 * g);
        nd4 = *reinterpret_cast<const i32x4*>(tb + 16 + 4 * g);
#pragma unroll
        for (int i = 0; i < 4; ++i) S[i] = rk4[i] == c ? (short)0x3F80 : (short)0;
      }
      unsigned* dtw = reinterpret_cast<unsigned*>(L + L_DT) + c * DT_LD + 4 * g;
      const unsigned* dtr = reinterpret_cast<const unsigned*>(L + L_DT) + (4 * g) * DT_LD + c;
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        {
          bf16x8 Dh, Dl;                          // slots 0-3 block bl = 0, 4-7 block bl = 1
#pragma unroll
          for (int bl = 0; bl < NBW; ++bl) {
            const f32x4& X = p == 0 ? Ck[bl] : (p == 1 ? Cq[bl] : Cv[bl]);
            u32x4 w;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              const __bf16 h = (__bf16)X[d];
              const __bf16 l = (__bf16)(X[d] - (float)h);
              Dh[4 * bl + d] = h;
              Dl[4 * bl + d] = l;
              w[d] = ((unsigned)__builtin_bit_cast(unsigned short, h) << 16) |
                     (unsigned)__builtin_bit_cast(unsigned short, l);
            }
            *reinterpret_cast<u32x4*>(dtw + 16 * bl) = w;
          }
#pragma unroll
          for (int fb = 0; fb < 2; ++fb) {
            if constexpr (LO) {
              C2[fb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wbh[p][fb], Dl, C2[fb], 0, 0, 0);
              C2[fb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wbl[p][fb], Dh, C2[fb], 0, 0, 0);
            }
            C2[fb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wbh[p][fb], Dh, C2[fb], 0, 0, 0);
          }
        }
        lds_order();
#pragma unroll
        for (int bl = 0; bl < NBW; ++bl) {
          const int q6 = NBW * p + bl;
          unsigned w[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) w[i] = dtr[i * DT_LD + 16 * bl];
          const s16x4 Th = __builtin_bit_cast(s16x4, (u32x2){__builtin_amdgcn_perm(w[1], w[0], 0x07060302u),
                                                             __builtin_amdgcn_perm(w[3], w[2], 0x07060302u)});
          const s16x4 Tl = __builtin_bit_cast(s16x4, (u32x2){__builtin_amdgcn_perm(w[1], w[0], 0x05040100u),
                                                             __builtin_amdgcn_perm(w[3], w[2], 0x05040100u)});
#pragma unroll
          for (int fb = 0; fb < 2; ++fb) {
            if constexpr (LO) {
              C3[q6][fb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Tl, Eh[fb], C3[q6][fb], 0, 0, 0);
              C3[q6][fb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Th, El[fb], C3[q6][fb], 0, 0, 0);
            }
            C3[q6][fb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Th, Eh[fb], C3[q6][fb], 0, 0, 0);
          }
          f32x4 Cs = (f32x4){0.f, 0.f, 0.f, 0.f};
          if constexpr (LO) Cs = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(S, Tl, Cs, 0, 0, 0);
          Cs = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(S, Th, Cs, 0, 0, 0);
          gb[q6] += (Cs[0] + Cs[1]) + (Cs[2] + Cs[3]);        // rows >= nn are 0
          if (p == 0) CnK[bl] = Cs;
          if (p == 2) CnV[bl] = Cs;
        }
        lds_order();
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- d edge_attr rows: the leader hands its half over, the follower adds (to what the
      //      buffer holds, in accumulate mode) and stores 16 bytes per lane ----------------------
      if (leader) {
        flag_wait(flg + F_MBFREE, k);             // the follower has consumed tile k - 1
        *reinterpret_cast<f32x4*>(P + P_MB + lane * 8) = C2[0];
        *reinterpret_cast<f32x4*>(P + P_MB + lane * 8 + 4) = C2[1];
        flag_set(flg + F_MB, k + 1);
      } else {
        flag_wait(flg + F_MB, k + 1);
        C2[0] += *reinterpret_cast<const f32x4*>(P + P_MB + lane * 8);
        C2[1] += *reinterpret_cast<const f32x4*>(P + P_MB + lane * 8 + 4);
        flag_set(flg + F_MBFREE, k + 1);
        // the old rows (issued at the top) have landed.  Behind them in the queue: this tile's dq
        // stores and the next tile's gathers - but the compiler branches around a dq store whose
        // lanes are all masked (f32 rows: the second store of a last tile with <= 8 edges), so
        // only NDQ - 1 of them are counted on there (the bf16 row store always has live lanes)
        if (rd_old) wait_vm<NG + (NDQ == 0 ? 0 : (DQ16 ? NDQ : NDQ - 1))>();
        if (acc) {
          C2[0] += *reinterpret_cast<const f32x4*>(P + P_GEA + lane * 4);
          C2[1] += *reinterpret_cast<const f32x4*>(P + P_GEA + 256 + lane * 4);
        }
        float* row = gea + (int64_t)ids[c] * F + 4 * g;
        if (t * TE + c < E && !(SPT_TO_SKIP & 16)) {
          *reinterpret_cast<f32x4*>(row) = C2[0];
          *reinterpret_cast<f32x4*>(row + 16) = C2[1];
        }
      }
      // dk / dv of the tile's TARGET nodes; the only data-dependent memory instructions: issued last
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (4 * g + r < nn && !(SPT_TO_SKIP & 1)) {
          float* row = gqkv + (int64_t)nd4[r] * LD + 32 * hh + c;
#pragma unroll
          for (int bl = 0; bl < NBW; ++bl) {
            unsafeAtomicAdd(row + 64 + 16 * bl, CnK[bl][r]);
            unsafeAtomicAdd(row + 128 + 16 * bl, CnV[bl][r]);
          }
        }
      }
      if (!leader) flag_set(flg + F_DONE, k + 1);
      lds_order();
    }
#undef SPT_TO_NODE_WAIT_SEL
  }
  // per-pair partial tables [192 rows][F + 1]: each wave of the pair writes its 96 rows
  if (partial) {
    float* pw = partial + (size_t)pair * 192 * (F + 1);
#pragma unroll
    for (int q6 = 0; q6 < 3 * NBW; ++q6) {
      const int p = q6 / NBW, bl = q6 % NBW;
      const int o0 = 64 * p + 16 * (2 * hh + bl);
#pragma unroll
      for (int fb = 0; fb < 2; ++fb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          pw[(size_t)(o0 + 4 * g + r) * (F + 1) + 16 * fb + c] = C3[q6][fb][r];
      const float bsum = xg_sum(gb[q6]);
      if (g == 0) pw[(size_t)(o0 + c) * (F + 1) + F] = bsum;
    }
  }
}

}  // namespace to

// ---- launcher called from edge_attn.hip's C entry points ------------------------------------
constexpr int ATTN_TO_MAX_PAIRS = 1024;     // 256 workgroups of 4 pairs: one workgroup per CU

// edge order of the edge-lane backward: 1 = by TARGET (this file; default), 0 = by source
// (edge_attn_el.hip).  Process-wide measurement / fallback switch.
static std::atomic<int> g_attn_el_target_order{[] { const char* e = getenv("SPT_EL_TARGET_ORDER"); return e ? (atoi(e) != 0) : 1; }()};
extern "C" int spt_attn_bwd_el_target_order(int on) {
  const int prev = g_attn_el_target_order;
  if (on >= 0) g_attn_el_target_order = on != 0;
  return prev;
}
bool attn_bwd_to_enabled() { return g_attn_el_target_order != 0; }

size_t attn_bwd_to_workspace_bytes(int64_t n, int64_t e) {
  const size_t ee = (size_t)(e > 0 ? e : 1);
  const size_t ntiles = (ee + to::TE - 1) / to::TE;
  return align_up((size_t)n * to::REC * 4, 256) + align_up((size_t)n * 4, 256) +
         align_up(ntiles * to::IDS * 4, 256) + align_up(ee * 64 * 4, 256);
}

void attn_pack_tile_ids_to_launch(const int32_t* eperm, const int32_t* tgt, const int32_t* src,
                                  const int32_t* tperm, int64_t e, int32_t* ids4, hipStream_t stream) {
  const int64_t ntiles = ceil_div(e, (int64_t)to::TE);
  if (ntiles > 0)
    to::pack_tile_ids_to_kernel<<<(int)ceil_div(ntiles * to::IDS, 256), 256, 0, stream>>>(
        eperm, tgt, src, tperm, e, ntiles, ids4);
}

bool attn_xcd_bands();      // edge_attn.hip: SPT_ATTN_XCD_BANDS (default on)
int attn_mirror_prepare_launch(const int64_t* ei, const int32_t* eperm, int64_t e, int64_t pairs,
                               int32_t* inv, int32_t* flag, hipStream_t stream) {
  if (e > 0)
    to::mirror_prepare_kernel<<<(int)(ceil_div(e, 256) < 256 * 16 ? ceil_div(e, 256) : 256 * 16), 256, 0,
                                stream>>>(ei, eperm, e, pairs, inv, flag);
  return 0;
}
void attn_pack_tile_ids_mirror_launch(const int32_t* eperm, const int32_t* tgt, const int32_t* src,
                                      const int32_t* inv, int64_t e, int64_t pairs, int32_t* ids4,
                                      hipStream_t stream) {
  const int64_t ntiles = ceil_div(e, (int64_t)to::TE);
  if (ntiles > 0)
    to::pack_tile_ids_mirror_kernel<<<(int)ceil_div(ntiles * to::IDS, 256), 256, 0, stream>>>(
        eperm, tgt, src, inv, e, pairs, ntiles, ids4);
}

// returns the number of partial tables written (<= ATTN_TO_MAX_PAIRS).  gqkv needs no
// initialisation: the k / v columns are zero-filled by the prep kernel (dk / dv are added per tile
// and node), the q columns are written by the reduction.
int attn_bwd_to_launch(const float* qkv, int64_t n, const int32_t* erowptr, const int32_t* eperm,
                       const int32_t* tgt, const int32_t* src, const int32_t* tile_ids,
                       const int32_t* tperm, int64_t e, const float* ea, const float* Wk,
                       const float* bk, const float* Wq, const float* bq, const float* Wv,
                       const float* bv, int scale_mode, float scale_a, const float* out,
                       const float* m, const float* z, const float* gout, float* gqkv, float* gea,
                       int gea_acc, float* partial, void* ws, int prec, hipStream_t stream) {
  const int64_t ntiles = ceil_div(e, (int64_t)to::TE);
  char* w = (char*)ws;
  float* rec = (float*)w;
  w += align_up((size_t)n * to::REC * 4, 256);
  float* scl = (float*)w;
  w += align_up((size_t)n * 4, 256);
  int32_t* ids4 = (int32_t*)w;
  w += align_up((size_t)ntiles * to::IDS * 4, 256);
  float* dqt = (float*)w;
  if (prec == 3)
    to::attn_bwd_to_prep_kernel<false><<<(int)ceil_div(n * 16, 256), 256, 0, stream>>>(
        qkv, gout, out, m, z, erowptr, n, scale_mode, scale_a, rec, scl, gqkv);
  else
    to::attn_bwd_to_prep_kernel<true><<<(int)ceil_div(n * 16, 256), 256, 0, stream>>>(
        qkv, gout, out, m, z, erowptr, n, scale_mode, scale_a, rec, scl, gqkv);
  if (!tile_ids) {
    attn_pack_tile_ids_to_launch(eperm, tgt, src, tperm, e, ids4, stream);
    tile_ids = ids4;
  }
  int64_t pairs = ntiles < ATTN_TO_MAX_PAIRS ? ntiles : ATTN_TO_MAX_PAIRS;
  const int64_t tpw = ceil_div(ntiles, pairs);
  pairs = ceil_div(ntiles, tpw);
  const int grid = (int)ceil_div(pairs, to::WAVES / 2);
  // XCD bands where the grid is the full one (a multiple of 8) and a band outlasts a few sweeps
  const int bands = attn_xcd_bands() && grid % 8 == 0 && ntiles >= 8 * pairs;
  static const int read_old = [] { const char* e = getenv("SPT_TO_READ_OLD"); return e ? atoi(e) != 0 : 0; }();
  gea_acc = (gea_acc != 0 ? 1 : 0) | (read_old ? 2 : 0);
  if (prec == 3)
    to::attn_bwd_to_kernel<3><<<grid, to::WAVES * 64, 0, stream>>>(
        qkv, e, tile_ids, ntiles, tpw, ea, Wk, bk, Wq, bq, Wv, bv, rec, gqkv, gea, gea_acc, dqt, partial, bands);
  else
    to::attn_bwd_to_kernel<1><<<grid, to::WAVES * 64, 0, stream>>>(
        qkv, e, tile_ids, ntiles, tpw, ea, Wk, bk, Wq, bq, Wv, bv, rec, gqkv, gea, gea_acc, dqt, partial, bands);
  const int64_t rblocks = ceil_div(n, (int64_t)16);
  if (prec == 3)
    to::attn_q_reduce_kernel<false><<<(int)(rblocks < 256 * 16 ? rblocks : 256 * 16), 256, 0, stream>>>(
        dqt, erowptr, scl, n, gqkv);
  else
    to::attn_q_reduce_kernel<true><<<(int)(rblocks < 256 * 16 ? rblocks : 256 * 16), 256, 0, stream>>>(
        dqt, erowptr, scl, n, gqkv);
  return grid * (to::WAVES / 2);
}

}  // namespace spt
