// Edge-lane ("EL") backward of the fused edge attention, SPT-64 head layout
// (H = 16, qk_dim = 4, value dim = 4, in_rpe_dim = 32).  Same math as edge_attn.hip /
// edge_attn_mfma.hip (autograd of src/nn/attention.py:202-315), different mapping - built for
// TWO waves per SIMD (<= 256 registers, ~18 KB of LDS per wave) where the packed kernel of
// edge_attn_mfma.hip needs the whole register file and LDS of a CU for four waves:
//
//   * 16-edge tiles over the CSR-by-source edge stream; a wave owns HALF of the heads (two of
//     the four 16-column blocks of each of the k / q / v projections) of a contiguous range of
//     tiles, its partner wave the other half.  The two never synchronise.
//   * the recompute GEMM runs TRANSPOSED: C[o][e] = W[o][:] . ea[e][:] (A operand = the weight
//     block, resident in registers; B operand = the tile's edge_attr rows).  In that C layout a
//     lane (g, c) holds, for edge c, the four dims of head 4 b + g in its four registers: the
//     per-head dot products are in-lane FMAs, nothing is computed redundantly across a DPP quad
//     (one exp per (edge, head) instead of four), and every per-edge quantity - the source's q /
//     gout / softmax state, the target's k / v - is simply the lane's own data, fetched by
//     LDS-DMA with the lane's own address.  Any number of source nodes per tile, no node
//     contexts, no passes.
//   * the gradient block D[e][o] in that layout IS the A operand of  d edge_attr = D W
//     (contraction over o, 16x16x32, W as B operands resident in registers); the result comes
//     out with lanes along the 32 feature columns -> 64-byte coalesced atomic rows.
//   * D goes through LDS once, as one 32-bit word (hi | lo bf16) per value, to the transposed
//     layout (lane = output column, registers = edges) that  dW += D^T EA  needs; the same
//     registers feed a segmented reduction over the tile's source nodes done ON THE MATRIX PIPE:
//     dq_node[n][o] = sum_e S[n][e] dq[e][o] with S the 0/1 node-membership matrix of the tile
//     (the bias gradients fall out of the same products).
//   * NO atomics for dk / dv.  The f32 atomic units of the chip retire ~320 G lane-adds/s whatever
//     the locality (tools/ubench/atomic_rate.hip: 900 M adds = one level-1 call = 2.8 ms), which
//     was the floor of every formulation that scattered dk / dv to the target rows.  Here the
//     per-edge rows [dk | dv] (512 B) are streamed out in CSR order (plain 16-byte stores, 8 KB
//     contiguous per tile) and summed per target by attn_kv_reduce_kernel through a CSR view of
//     the TARGETS (built once per batch and level like the source view): deterministic, f32
//     exact per edge, and ~3x cheaper than the atomics.
//   * nothing in the loop is a register-returning global load: tile data, per-edge node rows and
//     tile ids all arrive by LDS-DMA (ids two tiles ahead, edge_attr one tile ahead, the gathered
//     rows refilled as soon as the tile's per-edge math has consumed them), waited for with
//     counted vmcnt so the tile's own atomics never stall the next tile.
//
// Split-bf16 arithmetic as in edge_attn_mfma.hip: x = hi + lo, products hi*hi + lo*hi + hi*lo on
// the bf16 matrix pipe, f32 accumulate (~2^-17 relative per product: 17 of f32's 24 bits).  PREC = 1: hi only (bf16
// matrix-precision mode).
#include <math.h>
#include <stdlib.h>

#include "common.hpp"

namespace spt {
namespace el {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int TE = 16;            // edges per tile
constexpr int F = 32;             // in_rpe_dim
constexpr int NBW = 2;            // 16-column blocks of each projection owned by a wave
constexpr int WAVES = 8;          // per workgroup: 4 tile streams x 2 head halves, one workgroup per CU
constexpr int LD = 192;           // qkv row length

// LDS map, in floats.  Per wave:
constexpr int L_G = 0;                         // 4 x [64 lanes][4]: (k, v) rows of the targets x 2 blocks
constexpr int DT_LD = 36;                      // 32 words + 4: conflict-free both ways
constexpr int L_DT = L_G + 4 * 256;            // [16 edges][DT_LD] words (hi << 16 | lo), one projection
constexpr int L_TBL = L_DT + TE * DT_LD;       // rank[16], node[16]
constexpr int L_END = L_TBL + 32;
// per pair of waves (the two head halves of one tile stream):
constexpr int P_EA = 0;                        // 2 x [16][32] edge_attr rows (16-B chunks XOR-swizzled)
constexpr int P_IDS = P_EA + 2 * TE * F;       // 4 slots x (edge row, target, source) x 16
constexpr int P_MB = P_IDS + 4 * 48;           // [64 lanes][8]: the leader's half of d edge_attr
constexpr int P_GEA = P_MB + 512;              // [2][64 lanes][4]: what gedge_attr holds for the tile (accumulate)
constexpr int P_FLAG = P_GEA + 512;            // hand-shake counters (F_*)
constexpr int P_END = P_FLAG + 8;
constexpr int F_EA = 0;       // leader -> follower: edge_attr rows of tile k and ids of tile k + 1 landed (k + 1)
constexpr int F_TOP = 1;      // follower -> leader: operands of tile k read (k + 1)
constexpr int F_MB = 2;       // leader -> follower: mailbox holds tile k (k + 1)
constexpr int F_MBFREE = 3;   // follower -> leader: mailbox of tile k consumed (k + 1)
constexpr int F_DONE = 4;     // follower -> leader: tile k finished, its id slot is free (k + 1)
// shared by the workgroup: padded bf16 copies of [Wk; Wq; Wv] (row = output column, 32 + 8 bf16:
// 80-byte rows put the 16 lanes of a ds_read_b128 group on distinct bank windows) and the biases
constexpr int WB_LD = F + 8;
constexpr int WB_ELEMS = 192 * WB_LD;

// Outstanding VMEM operations per iteration, in issue order (every count is static: no memory
// instruction in the loop sits under a data-dependent branch except the dq atomics, issued last):
//   leader  : [top] ids of tile k+2 (1), node rows of tile k (5), edge_attr rows of tile k+1 (2)
//             [core] dk/dv rows (4)  [mid] k / v gathers of tile k+1 (4)  [tail] dq atomics (0..8)
//   follower: [top] node rows of tile k (5), old d edge_attr rows of tile k (2)
//             [core] dk/dv rows (4)  [mid] k / v gathers of tile k+1 (4)
//             [tail] d edge_attr rows (2), dq atomics (0..8)
// The node rows (q * scale, gout, (delta, ml) of the edge's SOURCE: the 16 edges of a tile share
// one to three sources, so these hit the L1) are register loads: an LDS-DMA instruction costs
// the CU ~84 cycles whatever it hits (~12 B / clk / CU), an L1-hitting register load a fraction.
constexpr int N_TOP = 3;
constexpr int N_NODE = 5;         // (q*scale, gout) x 2 blocks + (delta, ml)
constexpr int N_GATHER = 4;       // (k, v) x 2 blocks
constexpr int N_KV = 4;
constexpr int N_GEA = 2;
constexpr int N_OLD = 2;

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
// pair hand-shake through LDS counters (monotonic per launch).  LDS operations of a wave are
// performed in order, so data written before flag_set() is visible to whoever saw the flag.
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

// ---- per-node rows the backward reads per edge --------------------------------------------
//   dm[node][hh][g][{delta(bl=0), delta(bl=1), ml(bl=0), ml(bl=1)}],  head = 4 (2 hh + bl) + g
//     delta = <gout, out> of the head, ml = m + log(z + 1e-16) (softmax weight = exp(p - ml))
//   qs[node][64] = q * qk-scale of the node,  scl[node] = that scale (0 for a node without edges)
// One thread per (node, head).  Also zero-fills the q columns of gqkv (dq arrives by atomics).
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(
    const float* __restrict__ qkv, const float* __restrict__ gout, const float* __restrict__ out,
    const float* __restrict__ m, const float* __restrict__ z, const int32_t* __restrict__ erowptr,
    int64_t N, int scale_mode, float scale_a, float* __restrict__ dm, float* __restrict__ qs,
    float* __restrict__ scl, float* __restrict__ gqkv) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t node = idx >> 4;
  const int h = (int)(idx & 15);
  if (node >= N) return;
  const float4 g4 = *reinterpret_cast<const float4*>(gout + node * 64 + 4 * h);
  const float4 o4 = *reinterpret_cast<const float4*>(out + node * 64 + 4 * h);
  const float delta = (g4.x * o4.x + g4.y * o4.y) + (g4.z * o4.z + g4.w * o4.w);
  const float ml = m[node * 16 + h] + __logf(z[node * 16 + h] + 1e-16f);
  const int b = h >> 2, g = h & 3, hh = b >> 1, bl = b & 1;
  float* rec = dm + node * 32 + hh * 16 + g * 4;
  rec[bl] = delta;
  rec[2 + bl] = ml;
  const int deg = erowptr[node + 1] - erowptr[node];
  const float scale = deg > 0 ? qk_scale_of(scale_mode, scale_a, deg) : 0.f;
  const float4 q4 = *reinterpret_cast<const float4*>(qkv + node * LD + 4 * h);
  *reinterpret_cast<float4*>(qs + node * 64 + 4 * h) =
      make_float4(q4.x * scale, q4.y * scale, q4.z * scale, q4.w * scale);
  if (h == 0) scl[node] = scale;
  *reinterpret_cast<float4*>(gqkv + node * LD + 4 * h) = make_float4(0.f, 0.f, 0.f, 0.f);
}

// source node of every CSR position (when the caller does not hand it over)
__global__ __launch_bounds__(256) void expand_rowptr_kernel(const int32_t* __restrict__ rp,
                                                            int64_t N, int32_t* __restrict__ src) {
  const int64_t node = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (node >= N) return;
  const int a = rp[node], b = rp[node + 1];
  for (int j = a; j < b; ++j) src[j] = (int32_t)node;
}

// ids of a tile in one 192-byte record: [16 edge rows | 16 targets | 16 sources]; positions beyond
// the edge list repeat the last edge (they run with softmax weight 0)
__global__ __launch_bounds__(256) void pack_tile_ids_kernel(
    const int32_t* __restrict__ eperm, const int32_t* __restrict__ tgt,
    const int32_t* __restrict__ src, int64_t E, int64_t ntiles, int32_t* __restrict__ ids3) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= ntiles * 48) return;
  const int64_t tile = i / 48;
  const int l = (int)(i - tile * 48);
  int64_t j = tile * TE + (l & 15);
  j = j < E ? j : E - 1;
  ids3[i] = l < 16 ? (eperm ? eperm[j] : (int32_t)j) : (l < 32 ? tgt[j] : src[j]);
}

// gqkv[t][64 .. 191] = sum over the edges INTO t of their [dk | dv] rows, in ascending CSR position
// (tperm / trowptr = CSR view of the targets over the CSR-by-source positions; stable sort ->
// a fixed summation order -> deterministic).  Half a wave per target node: 32 lanes x 16 bytes =
// one 512-byte row per load, eight rows in flight.  The same pass applies the node's qk scale to
// the q columns (the main kernel adds the unscaled dq of the node's edges).
__global__ __launch_bounds__(256) void attn_kv_reduce_kernel(
    const float* __restrict__ dkv, const int32_t* __restrict__ tperm,
    const int32_t* __restrict__ trowptr, const float* __restrict__ scl, int64_t N,
    float* __restrict__ gqkv) {
  const int lane = threadIdx.x & 31;
  const int64_t half = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 5;
  const int64_t nhalf = ((int64_t)gridDim.x * 256) >> 5;
  for (int64_t t = half; t < N; t += nhalf) {
    const int a = trowptr[t], b = trowptr[t + 1];
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    int u = a;
    for (; u + 8 <= b; u += 8) {
      f32x4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int64_t j = tperm[u + i];
        v[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dkv + j * 128 + 4 * lane));
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += v[i];
    }
    if (u < b) {
      f32x4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int64_t j = tperm[u + i < b ? u + i : b - 1];
        v[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dkv + j * 128 + 4 * lane));
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (u + i < b) acc += v[i];
      }
    }
    *reinterpret_cast<f32x4*>(gqkv + t * LD + 64 + 4 * lane) = acc;
    if (lane < 16) {          // the q columns hold the unscaled sum of dq over the node's edges
      f32x4* qp = reinterpret_cast<f32x4*>(gqkv + t * LD + 4 * lane);
      *qp = *qp * scl[t];
    }
  }
}

// measurement switches (variant builds only): where does the time go
#if defined(SPT_EL_STORE)
#define SPT_EL_ADD(p, v) (*(p) = (v))
#else
#define SPT_EL_ADD(p, v) unsafeAtomicAdd((p), (v))
#endif
// -DSPT_ATTN_PROFILE: per-section cycle counts (s_memtime) of one wave, printed at its end - a
// measurement build only (tools/attn_microbench.py against gpurun_variants/), never shipped
#ifdef SPT_ATTN_PROFILE
#define SPT_PROBE(i)                                              \
  {                                                               \
    const uint64_t now_ = __builtin_amdgcn_s_memtime();           \
    prof[i] += now_ - tlast;                                      \
    tlast = now_;                                                 \
  }
#else
#define SPT_PROBE(i)
#endif

// FL ("full line"): the gathered k / v rows and the streamed [dk | dv] rows move as WHOLE 128-byte
// lines.  The MFMA layout gives lane (g, c) 16 bytes of edge c per request, so an instruction of
// the plain mapping touches 16 rows x 64 bytes - every 128-byte line of the traffic is split over
// two instructions.  With FL an instruction covers 4 edges x 256 bytes (lane l: edge 4 j + l / 16,
// 16-byte chunk l % 16 of the wave's [k (128 B) | v (128 B)] half of the row): for the gathers the
// DMA lands that as [edge][chunk] with the chunk position XOR-swizzled by the edge (the lane's own
// (k, v) chunks are then conflict-free 16-byte LDS reads); for the stores the per-edge results go
// through the same buffer (free once the rows are consumed) the other way round.  Same values, same
// instruction counts - only the shape of each request changes.
template <int PREC, bool FL>
__global__ __launch_bounds__(WAVES * 64, 2) void attn_bwd_el_kernel(
    const float* __restrict__ qkv, int64_t E, const int32_t* __restrict__ ids3, int64_t ntiles,
    int64_t tpw, const float* __restrict__ ea, const float* __restrict__ Wk,
    const float* __restrict__ bk, const float* __restrict__ Wq, const float* __restrict__ bq,
    const float* __restrict__ Wv, const float* __restrict__ bv, const float* __restrict__ dm,
    const float* __restrict__ qs, const float* __restrict__ gout, float* __restrict__ gqkv,
    float* __restrict__ gea, int gea_acc, float* __restrict__ dkv, float* __restrict__ partial) {
  static_assert(PREC == 1 || PREC == 3, "bf16 matrix pipe only");
  constexpr bool LO = PREC == 3;
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
  const bool acc = gea_acc != 0;

  // ---- operands ---------------------------------------------------------------------------
  auto Wof = [&](int p) { return p == 0 ? Wk : (p == 1 ? Wq : Wv); };
  // recompute GEMM, A operand (16x16x32): lane (g, c) reads W[16 b + c][8 g .. 8 g + 7] out of the
  // workgroup's LDS copy every tile
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
  // d edge_attr GEMM, resident operand (16x16x32 over the wave's 32 output columns of a
  // projection): lane (g, c), slot i holds W[16 (2 hh + (i >> 2)) + 4 g + (i & 3)][16 fb + c] -
  // as the A operand it yields the transposed product  dEA^T[f][e] = W^T D^T
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
  // weight-gradient accumulators: C3[q6][fb][r] = dW[64 p + 16 b + 4 g + r][16 fb + c], q6 = 2 p + bl
  f32x4 C3[3 * NBW][2];
  float gb[3 * NBW];                         // bias gradients of column 64 p + 16 b + c, per-g partials
#pragma unroll
  for (int q6 = 0; q6 < 3 * NBW; ++q6) {
    C3[q6][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    C3[q6][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    gb[q6] = 0.f;
  }

  const int64_t t_begin = pair * tpw;
  const int64_t t_end = (t_begin + tpw < ntiles) ? t_begin + tpw : ntiles;
  if (t_begin < t_end) {
    int* ids_ring = reinterpret_cast<int*>(P + P_IDS);
    const int64_t t_last = t_end - 1;
    auto issue_ids = [&](int64_t t, int slot) {        // one 192-byte record: 48 lanes x 4 bytes
      t = t < t_last ? t : t_last;
      if (lane < 48) lds_dma4(ids3 + t * 48 + lane, P + P_IDS + slot * 48);
    };
    auto issue_ea = [&](int slot, int buf) {
      const int* ids = ids_ring + slot * 48;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int u = p * 8 + (lane >> 3), ch = lane & 7;
        const int64_t e = ids[u];
        lds_dma16(ea + e * F + ((ch ^ (u & 7)) << 2), P + P_EA + buf * TE * F + p * 256);
      }
    };
    auto issue_old = [&](int slot) {                   // what gedge_attr holds for the tile's edges:
      const int* ids = ids_ring + slot * 48;           // lane (g, c): row of edge c, columns 16 fb + 4 g ..
      const float* row = gea + (int64_t)ids[c] * F + 4 * g;
      lds_dma16(row, P + P_GEA);
      lds_dma16(row + 16, P + P_GEA + 256);
    };
    auto issue_gather = [&](int slot) {
      const int* ids = ids_ring + slot * 48;
      float* G = L + L_G;
      if constexpr (FL) {
        // chunk ch of the wave's half row: ch < 8 -> k columns 64 + 32 hh + 4 ch, else v columns
        // 128 + 32 hh + 4 (ch - 8); lane l fetches chunk (l % 16) ^ (edge % 16) of edge 4 j + l / 16
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int e = 4 * j + g;
          const int ch = c ^ e;
          const int64_t tc = ids[16 + e];
          lds_dma16(qkv + tc * LD + 64 + 32 * hh + 4 * ch + ((ch & 8) ? 32 : 0), G + j * 256);
        }
      } else {
        const int64_t tc = ids[16 + c];
        const float* kv = qkv + tc * LD + 32 * hh + 4 * g;
#pragma unroll
        for (int bl = 0; bl < NBW; ++bl) {
          lds_dma16(kv + 64 + 16 * bl, G + (2 * bl + 0) * 256);
          lds_dma16(kv + 128 + 16 * bl, G + (2 * bl + 1) * 256);
        }
      }
    };
    // node rows of the tile's edges, straight into registers (asm: invisible to the compiler's
    // wait-count pass, waited for by hand before the per-edge math)
    f32x4 nq[NBW], ng[NBW], ndm;
    auto issue_node = [&](int slot) {
      const int* ids = ids_ring + slot * 48;
      const int64_t sc = ids[32 + c];
      const float* qq = qs + sc * 64 + 32 * hh + 4 * g;
      const float* gg = gout + sc * 64 + 32 * hh + 4 * g;
      const float* dd = dm + sc * 32 + hh * 16 + g * 4;
#ifndef SPT_EL_NO_NODE
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(nq[0]) : "v"(qq) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(nq[1]) : "v"(qq) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ng[0]) : "v"(gg) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(ng[1]) : "v"(gg) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ndm) : "v"(dd) : "memory");
#else   // measurement variant: no node rows (garbage results)
      asm volatile("" : "=v"(nq[0]), "=v"(nq[1]), "=v"(ng[0]), "=v"(ng[1]), "=v"(ndm) : "v"(qq), "v"(gg), "v"(dd));
#endif
    };
#define SPT_NODE_WAIT(N)                                                              \
  asm volatile("s_waitcnt vmcnt(" #N ")"                                             \
               : "+v"(nq[0]), "+v"(nq[1]), "+v"(ng[0]), "+v"(ng[1]), "+v"(ndm)::"memory")

    // ---- prologue: ids of the first two tiles, the first tile's edge_attr rows (leader), then
    //      every wave's own gathered rows --------------------------------------------------------
    if (leader) {
      issue_ids(t_begin, 0);
      issue_ids(t_begin + 1, 1);
      wait_vm<0>();
      issue_ea(0, 0);
      wait_vm<0>();
      flag_set(flg + F_EA, 1);
    } else {
      flag_wait(flg + F_EA, 1);
    }
    issue_gather(0);
    wait_vm<0>();

#ifdef SPT_ATTN_PROFILE
    uint64_t prof[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t tlast = __builtin_amdgcn_s_memtime();
#endif
    int k = 0;                                    // tile index within the pair's range
    for (int64_t t = t_begin; t < t_end; ++t, ++k) {
      const int s0 = k & 3, s1 = (k + 1) & 3, s2 = (k + 2) & 3;   // id ring slots of tiles k, k+1, k+2
      if (leader) {
        // edge_attr rows of tile k and the ids of tile k + 1 have landed; behind them in the queue:
        // the dk / dv rows of tile k - 1, the gathers of tile k, dq atomics
        wait_vm<N_KV + N_GATHER>();
        flag_set(flg + F_EA, k + 1);
      } else {
        flag_wait(flg + F_EA, k + 1);
      }
      SPT_PROBE(0)
      const float* slab = P + P_EA + (k & 1) * TE * F;
      // B operand of the recompute GEMM: lane (g, c) holds ea[edge c][8 g .. 8 g + 7]
      bf16x8 Ah, Al;
      {
        const float4 a0 = *reinterpret_cast<const float4*>(slab + c * F + (((2 * g) ^ (c & 7)) << 2));
        const float4 a1 = *reinterpret_cast<const float4*>(slab + c * F + (((2 * g + 1) ^ (c & 7)) << 2));
        const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        split_bf16<8>(a, Ah, Al);
      }
      // B operand of the weight-gradient GEMM: ea[edge 4 g + i][16 fb + c]
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
      const int* ids = ids_ring + s0 * 48;
      const int s_c = ids[32 + c];                                    // source node of edge c
      wait_lds();                                 // the edge_attr slab is consumed
      SPT_PROBE(1)
      // Both roles put exactly TWO requests behind the node rows (leader: the next tile's
      // edge_attr rows; follower: what gedge_attr holds for this tile - fetched in store mode too,
      // unused there), so that ONE unbranched counted wait serves both: the wait names the loads'
      // destination registers, and behind a branch the compiler copies them BEFORE the wait.
      if (leader) {
        // slot s2 held tile k - 2: the follower must have finished that tile (it reads the edge
        // rows last); buffer (k + 1) & 1 held tile k - 1: the follower must have read its operands
        flag_wait(flg + F_DONE, k - 1);
        issue_ids(t + 2, s2);
        issue_node(s0);
        flag_wait(flg + F_TOP, k);
        issue_ea(s1, (k + 1) & 1);
      } else {
        flag_set(flg + F_TOP, k + 1);
        issue_node(s0);
        issue_old(s0);
      }
      SPT_PROBE(2)

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

      // ---- node ranks of the tile's edges (for the dq reduction): rank = number of source
      //      changes up to the edge; 16-lane scan on DPP row shifts ---------------------------
      int rank;
      {
        const int prev = __builtin_amdgcn_update_dpp(s_c, s_c, 0x111, 0xF, 0xF, false);
        rank = (s_c != prev) ? 1 : 0;
        rank += __builtin_amdgcn_update_dpp(0, rank, 0x111, 0xF, 0xF, false);
        rank += __builtin_amdgcn_update_dpp(0, rank, 0x112, 0xF, 0xF, false);
        rank += __builtin_amdgcn_update_dpp(0, rank, 0x114, 0xF, 0xF, false);
        rank += __builtin_amdgcn_update_dpp(0, rank, 0x118, 0xF, 0xF, false);
      }
      const int nn = __builtin_amdgcn_readlane(rank, 15) + 1;        // source nodes in the tile

      // ---- per-edge gradients, in place: Ck <- dk, Cq <- dq, Cv <- dv -------------------------
      SPT_PROBE(3)
      // the gathered k / v rows and the node rows of tile k have landed; behind them in the queue:
      // two requests of this tile's top (see there)
      SPT_NODE_WAIT(2);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      SPT_PROBE(4)
      {
        const float* Gb = L + L_G + lane * 4;
        const f32x4 dmv = ndm;
        const bool valid = t * TE + c < E;
        {
          int* tb = reinterpret_cast<int*>(L + L_TBL);
          tb[c] = rank;                           // every lane group writes the same values
          tb[16 + rank] = s_c;
        }
#pragma unroll
        for (int bl = 0; bl < NBW; ++bl) {
          // FL: chunk 4 bl + g (k) / 8 + 4 bl + g (v) of edge c sits at position chunk ^ c
          const f32x4 kt = FL ? *reinterpret_cast<const f32x4*>(L + L_G + c * 64 + 4 * ((4 * bl + g) ^ c))
                              : *reinterpret_cast<const f32x4*>(Gb + (2 * bl + 0) * 256);
          const f32x4 vt = FL ? *reinterpret_cast<const f32x4*>(L + L_G + c * 64 + 4 * ((8 + 4 * bl + g) ^ c))
                              : *reinterpret_cast<const f32x4*>(Gb + (2 * bl + 1) * 256);
          const f32x4 qr = nq[bl];
          const f32x4 gs = ng[bl];
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
        // dk / dv rows of the tile's edges, CSR order: row j = [dk (64) | dv (64)], this lane's four
        // dims of head 4 b + g of edge c - 16-byte non-temporal stores, summed per target later
        if constexpr (FL) {
          // through the (consumed) gather buffer: [edge c][position chunk ^ c] <- the lane's four
          // chunks, read back as [edge 4 j + l / 16][position l % 16] = whole 128-byte lines
          wait_lds();                             // the gathered rows are consumed
          float* G = L + L_G;
#pragma unroll
          for (int bl = 0; bl < NBW; ++bl) {
            *reinterpret_cast<f32x4*>(G + c * 64 + 4 * ((4 * bl + g) ^ c)) = Ck[bl];
            *reinterpret_cast<f32x4*>(G + c * 64 + 4 * ((8 + 4 * bl + g) ^ c)) = Cv[bl];
          }
          lds_order();
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int e = 4 * j + g;
            const int ch = c ^ e;
            const f32x4 v4 = *reinterpret_cast<const f32x4*>(G + j * 256 + lane * 4);
            // rows beyond the edge list own no row (masked per lane: the instruction still issues
            // - unless a whole instruction's edges are past the end, which only the LAST tile of the
            // edge list can have: the one counted wait behind it in that tile allows for it)
            if (t * TE + e < E)
              __builtin_nontemporal_store(
                  v4, reinterpret_cast<f32x4*>(dkv + (t * TE + e) * 128 + 32 * hh + 4 * (ch & 7) + ((ch & 8) ? 64 : 0)));
          }
        } else {
          float* drow = dkv + (t * TE + c) * 128 + 32 * hh + 4 * g;
          if (valid) {                              // rows beyond the edge list own no row
#pragma unroll
            for (int bl = 0; bl < NBW; ++bl) {
              __builtin_nontemporal_store(Ck[bl], reinterpret_cast<f32x4*>(drow + 16 * bl));
              __builtin_nontemporal_store(Cv[bl], reinterpret_cast<f32x4*>(drow + 64 + 16 * bl));
            }
          }
        }
      }
      wait_lds();                                 // the gathered rows are consumed
      SPT_PROBE(5)
      issue_gather(s1);                           // rows of tile k + 1 (its ids landed at the top)
      SPT_PROBE(6)

      // ---- per projection p (k, q, v), fenced for the scheduler so that one projection's
      //      operands die before the next one's are built:
      //   (1) split D_p once: packed halves = B operand of  d edge_attr^T += W_p^T D_p^T ; one word
      //       (hi << 16 | lo) per value -> LDS
      //   (2) read the words back transposed (lane = output column 16 bl + c, registers = edges
      //       4 g + i): A operand of  dW_p += D_p^T EA , and B operand of the reduction per source
      //       node of the tile on the matrix pipe, sum_e S[n][e] D[e][o]: dq of the nodes (p == 1)
      //       and, summed over the node rows, the bias gradient of every block
      // C2[fb][r] = d edge_attr[edge c][16 fb + 4 g + r], the wave's 96 columns' share
      f32x4 C2[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
      f32x4 Cn[NBW];                              // dq per node: Cn[n = 4 g + r][o = 16 b + c]
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
          if (p == 1) Cn[bl] = Cs;
        }
        lds_order();
        __builtin_amdgcn_sched_barrier(0);
      }
      SPT_PROBE(7)
      // ---- d edge_attr rows: the leader hands its half over, the follower adds (to what the
      //      buffer holds, in accumulate mode: each row is owned by exactly one tile, so this is a
      //      plain read-modify-write) and stores 16 bytes per lane --------------------------------
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
        // the old rows (issued at the top) have landed.  Behind them in the queue: the next tile's
        // gathers and this tile's dk / dv row stores - of which the full-line form may lose all but
        // one (the compiler branches around a store whose edges are all past the end of the edge
        // list: the last tile), so it counts on one only
        wait_vm<N_GATHER + (FL ? 1 : N_KV)>();
        if (acc) {
          C2[0] += *reinterpret_cast<const f32x4*>(P + P_GEA + lane * 4);
          C2[1] += *reinterpret_cast<const f32x4*>(P + P_GEA + 256 + lane * 4);
        }
        float* row = gea + (int64_t)ids[c] * F + 4 * g;
        if (t * TE + c < E) {                     // rows beyond the edge list own no row (only the
          *reinterpret_cast<f32x4*>(row) = C2[0];        // last tile of the edge list has any: no later
          *reinterpret_cast<f32x4*>(row + 16) = C2[1];   // iteration counts on these two stores)
        }
      }
      // dq of the tile's nodes, unscaled (the reduction pass applies the node's qk scale); the
      // only data-dependent memory instructions: issued last
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (4 * g + r < nn) {
          float* row = gqkv + (int64_t)nd4[r] * LD + 32 * hh + c;
#pragma unroll
          for (int bl = 0; bl < NBW; ++bl) unsafeAtomicAdd(row + 16 * bl, Cn[bl][r]);
        }
      }
      if (!leader) flag_set(flg + F_DONE, k + 1);
      SPT_PROBE(8)
      lds_order();
    }
#ifdef SPT_ATTN_PROFILE
    if (lane == 0 && (wave == 4 || wave == 5 || wave == 1030))
      printf("attn_bwd_el wave %ld tiles %ld cycles: topwait %lu operands %lu issue_top %lu regemm+rank %lu "
             "gwait %lu core %lu issue_g %lu rounds %lu tail %lu\n", (long)wave,
             (long)(t_end - t_begin), (unsigned long)prof[0], (unsigned long)prof[1],
             (unsigned long)prof[2], (unsigned long)prof[3], (unsigned long)prof[4],
             (unsigned long)prof[5], (unsigned long)prof[6], (unsigned long)prof[7],
             (unsigned long)prof[8]);
#endif
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

}  // namespace el

// request shape of the edge-lane backward's k / v gathers and [dk | dv] stores: 1 = whole 128-byte
// lines per instruction (FL), 0 = the MFMA layout's own 64-byte pieces.  Same results bit for bit.
static std::atomic<int> g_attn_el_full_line{[] { const char* e = getenv("SPT_EL_FULL_LINE"); return e ? (atoi(e) != 0) : 1; }()};
extern "C" int spt_attn_bwd_el_full_line(int on) {
  const int prev = g_attn_el_full_line;
  if (on >= 0) g_attn_el_full_line = on != 0;
  return prev;
}

// ---- launcher called from edge_attn.hip's C entry points ------------------------------------
constexpr int ATTN_EL_MAX_PAIRS = 1024;     // 256 workgroups of 4 pairs: one workgroup per CU

size_t attn_bwd_el_workspace_bytes(int64_t n, int64_t e) {
  const size_t ee = (size_t)(e > 0 ? e : 1);
  const size_t ntiles = (ee + el::TE - 1) / el::TE;
  return align_up((size_t)n * 32 * 4, 256) + align_up((size_t)n * 64 * 4, 256) +
         align_up((size_t)n * 4, 256) + align_up(ee * 4, 256) + align_up(ntiles * 48 * 4, 256) +
         align_up(ee * 128 * 4, 256);
}

// returns the number of partial tables written (<= ATTN_EL_MAX_PAIRS).  gqkv needs no
// initialisation: the q columns are zero-filled by the prep kernel (dq is added per tile and node)
// and scaled by the reduction, which writes the k / v columns.
void attn_pack_tile_ids_launch(const int32_t* eperm, const int32_t* tgt, const int32_t* src,
                               int64_t e, int32_t* ids3, hipStream_t stream) {
  const int64_t ntiles = ceil_div(e, (int64_t)el::TE);
  if (ntiles > 0)
    el::pack_tile_ids_kernel<<<(int)ceil_div(ntiles * 48, 256), 256, 0, stream>>>(eperm, tgt, src, e,
                                                                                  ntiles, ids3);
}

int attn_bwd_el_launch(const float* qkv, int64_t n, const int32_t* erowptr, const int32_t* eperm,
                       const int32_t* tgt, const int32_t* src, const int32_t* tile_ids,
                       const int32_t* tperm, const int32_t* trowptr, int64_t e, const float* ea,
                       const float* Wk, const float* bk, const float* Wq, const float* bq,
                       const float* Wv, const float* bv, int scale_mode, float scale_a,
                       const float* out, const float* m, const float* z, const float* gout,
                       float* gqkv, float* gea, int gea_acc, float* partial, void* ws, int prec,
                       hipStream_t stream) {
  const int64_t ntiles = ceil_div(e, (int64_t)el::TE);
  char* w = (char*)ws;
  float* dm = (float*)w;
  w += align_up((size_t)n * 32 * 4, 256);
  float* qs = (float*)w;
  w += align_up((size_t)n * 64 * 4, 256);
  float* scl = (float*)w;
  w += align_up((size_t)n * 4, 256);
  int32_t* srcbuf = (int32_t*)w;
  w += align_up((size_t)(e > 0 ? e : 1) * 4, 256);
  int32_t* ids3 = (int32_t*)w;
  w += align_up((size_t)ntiles * 48 * 4, 256);
  float* dkv = (float*)w;
  el::attn_bwd_prep_kernel<<<(int)ceil_div(n * 16, 256), 256, 0, stream>>>(
      qkv, gout, out, m, z, erowptr, n, scale_mode, scale_a, dm, qs, scl, gqkv);
  if (!tile_ids) {
    if (!src) {
      el::expand_rowptr_kernel<<<(int)ceil_div(n, 256), 256, 0, stream>>>(erowptr, n, srcbuf);
      src = srcbuf;
    }
    attn_pack_tile_ids_launch(eperm, tgt, src, e, ids3, stream);
    tile_ids = ids3;
  }
  int64_t pairs = ntiles < ATTN_EL_MAX_PAIRS ? ntiles : ATTN_EL_MAX_PAIRS;
  const int64_t tpw = ceil_div(ntiles, pairs);
  pairs = ceil_div(ntiles, tpw);
  const int grid = (int)ceil_div(pairs, el::WAVES / 2);
#define SPT_EL_LAUNCH(P, F)                                                                      \
  el::attn_bwd_el_kernel<P, F><<<grid, el::WAVES * 64, 0, stream>>>(                             \
      qkv, e, tile_ids, ntiles, tpw, ea, Wk, bk, Wq, bq, Wv, bv, dm, qs, gout, gqkv, gea, gea_acc, \
      dkv, partial)
  if (prec == 3) {
    if (g_attn_el_full_line) SPT_EL_LAUNCH(3, true); else SPT_EL_LAUNCH(3, false);
  } else {
    if (g_attn_el_full_line) SPT_EL_LAUNCH(1, true); else SPT_EL_LAUNCH(1, false);
  }
#undef SPT_EL_LAUNCH
  const int64_t rblocks = ceil_div(n, (int64_t)8);
  el::attn_kv_reduce_kernel<<<(int)(rblocks < 256 * 16 ? rblocks : 256 * 16), 256, 0, stream>>>(
      dkv, tperm, trowptr, scl, n, gqkv);
  return grid * (el::WAVES / 2);
}

}  // namespace spt
