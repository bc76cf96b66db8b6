// MFMA formulation of the fused edge attention for the SPT-64 head layout
// (H = 16 heads, qk_dim = 4, value dim = 4, in_rpe_dim = 32; the S3DIS / DALES
// configs).  Same math and same results as edge_attn.hip, different mapping:
//
//   * a wave owns one source node and walks its edges in tiles of 16;
//   * the three RPE projections of a tile are ONE [16 edges x 32] x [32 x 192]
//     product on the matrix pipe: 96 x v_mfma_f32_16x16x4_f32 (f32 in, f32
//     accumulate - bitwise an fmaf chain, so the f32 parity bar is unchanged),
//     with the [32 x 192] weight block resident in 96 VGPRs as B operands;
//   * everything after the GEMM happens in the MFMA C layout - lane (g, c),
//     register r  <->  edge 4g + r, output column 16b + c - where a head is one
//     DPP quad, so the per-head dot product is two v_add_f32_dpp, and the
//     softmax / value accumulation costs ~12 VALU instructions per edge
//     instead of ~150 in the lane-per-output VALU kernel;
//   * tiles stream global -> LDS asynchronously (global_load_lds), edge_attr
//     rows XOR-swizzled so that the A-operand reads are <= 2-way bank conflicts.
//
// The backward runs three GEMMs per tile on the same pipe: the recompute, the
// RPE weight gradients  dW[192x32] += D^T[192x16] EA[16x32]  (A operand = the
// C-layout registers as they are) and  d edge_attr[16x32] = D[16x192] W[192x32]
// (D transposed through LDS).
#include <math.h>

#include "common.hpp"

namespace spt {
namespace mfma {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TE = 16;            // edges per tile = MFMA M
constexpr int F = 32;             // in_rpe_dim
constexpr int NB = 4;             // 16-column blocks per 64-wide projection
constexpr int WAVES = 4;
constexpr int EA_FLOATS = TE * F;         // 512
constexpr int ROW = 64;                   // k / v row length
constexpr int SLAB = EA_FLOATS + 2 * TE * ROW;  // 2560 floats = 10 KB

#define SPT_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define SPT_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ float quad_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));
  return v;
}
// Four quad sums at once with the DPP operand folded into the add (round 6): the builtin's
// v_mov_b32_dpp + v_add pairs get SLP-packed by the compiler into v_mov 0 / v_mov_dpp / v_pk_add
// groups - 2.5 instructions per value and step; this is 1.  The s_nop covers the two wait states
// between a VALU write and a DPP read of the same register (the hazard recogniser does not look
// inside asm); the second step's sources are three instructions old by construction.
__device__ __forceinline__ void quad_sum4(const float (&x)[4], float (&o)[4]) {
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]));
}
// lane ^ 16 / lane ^ 32 through gfx950's row / half swaps (VALU, no LDS permute)
__device__ __forceinline__ float xor16(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);   // {[r0 r0 r2 r2], [r1 r1 r3 r3]}
  return __uint_as_float((threadIdx.x & 16) ? r[0] : r[1]);
}
__device__ __forceinline__ float xor32(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // {[lo lo], [hi hi]}
  return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}
__device__ __forceinline__ float xg_sum(float v) {  // sum over the 4 lane groups g
  v += xor16(v);
  v += xor32(v);
  return v;
}
// The same for the forward kernel (the backward kernels of this file sit at their register limit and
// spill with another schedule).  permlane16_swap(u, u) returns {[r0 r0 r2 r2], [r1 r1 r3 r3]}
// (rows of 16 lanes), permlane32_swap(u, u) {[lo lo], [hi hi]}: the two halves of the result ARE the
// two operands of the step in every lane - no per-lane select (round 6; v + xor16(v) picked one of
// them with a v_cndmask first; same sums bit for bit, the operands only swap sides in odd rows).
__device__ __forceinline__ float xg_sum_sw(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const unsigned w = __float_as_uint(__uint_as_float(r[0]) + __uint_as_float(r[1]));
  const auto t = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}
// v_max_f32 as it is (fmaxf costs a canonicalising v_max x, x per operand)
__device__ __forceinline__ float vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float xg_max_sw(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const unsigned w = __float_as_uint(vmax(__uint_as_float(r[0]), __uint_as_float(r[1])));
  const auto t = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return vmax(__uint_as_float(t[0]), __uint_as_float(t[1]));
}

__device__ __forceinline__ float qk_scale_of(int mode, float a, int deg) {
  const float g = __builtin_amdgcn_rsqf((float)deg);   // v_rsq_f32 (1 ulp; the same in every attention kernel)
  if (mode == 0) return a * g;
  if (mode == 1) return a + g;
  return a;
}

struct Tile {
  int64_t s;
  int start, end, t0;
  bool valid;
};
__device__ __forceinline__ Tile tile_of_node(int64_t s, int64_t N, const int32_t* __restrict__ rp) {
  Tile d;
  d.s = s;
  d.valid = s < N;
  d.start = d.valid ? __builtin_amdgcn_readfirstlane(rp[s]) : 0;
  d.end = d.valid ? __builtin_amdgcn_readfirstlane(rp[s + 1]) : 0;
  d.t0 = d.start;
  return d;
}
__device__ __forceinline__ Tile tile_advance(const Tile& c, int64_t N, int64_t nw,
                                             const int32_t* __restrict__ rp) {
  if (c.valid && c.t0 + TE < c.end) {
    Tile d = c;
    d.t0 += TE;
    return d;
  }
  return tile_of_node(c.s + nw, N, rp);
}
__device__ __forceinline__ int tile_count(const Tile& d) {
  const int r = d.end - d.t0;
  return r < TE ? (r > 0 ? r : 0) : TE;
}
__device__ __forceinline__ void wait_vmem_all() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// LDS-DMA through inline asm: the compiler's waitcnt pass puts a vmcnt(0) in front
// of EVERY later LDS read once it has seen a global_load_lds builtin (it cannot
// tell the two ping-pong buffers apart), which turns the prefetch synchronous.
// Issued as asm, the DMA is invisible to that pass; this file waits explicitly
// (wait_vmem_all) before the first read of a freshly filled buffer.
__device__ __forceinline__ void lds_dma16(const float* g, float* lds) {
  const unsigned a = __builtin_amdgcn_readfirstlane(
      (unsigned)(size_t)((__attribute__((address_space(3))) void*)lds));
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
               :: "s"(a), "v"(g) : "memory");
}
__device__ __forceinline__ void lds_dma4(const float* g, float* lds) {
  const unsigned a = __builtin_amdgcn_readfirstlane(
      (unsigned)(size_t)((__attribute__((address_space(3))) void*)lds));
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off"
               :: "s"(a), "v"(g) : "memory");
}

// Asynchronous loads of one tile: edge_attr rows (16-B chunks XOR-swizzled by the
// row index), gathered k rows and v rows of the edge targets (4 rows / instruction).
__device__ __forceinline__ void tile_issue(const float* __restrict__ qkv, int ld,
                                           const float* __restrict__ ea, int e_lane,
                                           int t_lane, int cnt, float* buf, int lane) {
  // all six cross-lane reads first: one LDS round trip (interleaved with the conditional issues
  // below, each ds_bpermute was waited for on its own - six round trips at the top of every tile)
  int e2[2];
  int64_t t4[4];
#pragma unroll
  for (int p = 0; p < 2; ++p) e2[p] = __shfl(e_lane, p * 8 + (lane >> 3), 64);
#pragma unroll
  for (int i = 0; i < 4; ++i) t4[i] = __shfl(t_lane, i * 4 + (lane >> 4), 64);
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int u = p * 8 + (lane >> 3), ch = lane & 7;
    if (u < cnt)
      lds_dma16(ea + (size_t)e2[p] * F + ((ch ^ (u & 7)) << 2), buf + p * 256);
  }
  float* kbuf = buf + EA_FLOATS;
  float* vbuf = kbuf + TE * ROW;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int u = i * 4 + (lane >> 4), ch = lane & 15;
    const int64_t t = t4[i];
    if (u < cnt) {
      lds_dma16(qkv + t * ld + 64 + ch * 4, kbuf + i * 256);
      lds_dma16(qkv + t * ld + 128 + ch * 4, vbuf + i * 256);
    }
  }
}

// Same, with the tile's ids in LDS (ids[0..15] = edge rows, ids[16..31] = targets; `e0` >= 0:
// no edge permutation, rows are e0 + u): every lane reads the id of the row its chunk belongs
// to - no cross-lane shuffles, one LDS round trip for the six reads.
__device__ __forceinline__ void tile_issue_ids(const float* __restrict__ qkv, int ld,
                                               const float* __restrict__ ea, const int* ids,
                                               int64_t e0, int cnt, float* buf, int lane) {
  int e2[2], t4[4];
#pragma unroll
  for (int p = 0; p < 2; ++p) e2[p] = ids[p * 8 + (lane >> 3)];
#pragma unroll
  for (int i = 0; i < 4; ++i) t4[i] = ids[TE + i * 4 + (lane >> 4)];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int u = p * 8 + (lane >> 3), ch = lane & 7;
    const int64_t e = e0 >= 0 ? e0 + u : (int64_t)e2[p];
    if (u < cnt)
      lds_dma16(ea + (size_t)e * F + ((ch ^ (u & 7)) << 2), buf + p * 256);
  }
  float* kbuf = buf + EA_FLOATS;
  float* vbuf = kbuf + TE * ROW;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int u = i * 4 + (lane >> 4), ch = lane & 15;
    const int64_t t = t4[i];
    if (u < cnt) {
      lds_dma16(qkv + t * ld + 64 + ch * 4, kbuf + i * 256);
      lds_dma16(qkv + t * ld + 128 + ch * 4, vbuf + i * 256);
    }
  }
}

// B operands of the RPE GEMM: lane (g, c) holds W[16 b + c][4 step + g]
__device__ __forceinline__ void load_b(const float* __restrict__ W, int g, int c,
                                       float (&B)[NB][8]) {
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int st = 0; st < 8; ++st) B[b][st] = W[(size_t)(16 * b + c) * F + 4 * st + g];
}

// A operands of a tile: lane (g, c) holds ea[edge c][4 step + g] (un-swizzle on read)
__device__ __forceinline__ void load_a(const float* slab, int g, int c, float (&A)[8]) {
#pragma unroll
  for (int st = 0; st < 8; ++st) A[st] = slab[c * F + ((st ^ (c & 7)) << 2) + g];
}

// ---- split-bf16 operands: x = hi + lo (+ 2^-18 |x|), both bf16 (RNE).  A product of two
// f32 numbers is taken as hi*hi + lo*hi + hi*lo on the bf16 matrix pipe (16x the f32 pipe's
// rate, exact bf16 x bf16 products, f32 accumulate): relative error <= ~3 * 2^-18 per product,
// i.e. ~2^-17 relative per product: 17 of f32's 24 bits - inside the attention parity bar (1e-5 + 1e-4 |ref|).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <int NV, typename V>
__device__ __forceinline__ void split_bf16(const float (&x)[NV], V& hi, V& lo) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const __bf16 h = (__bf16)x[i];
    hi[i] = h;
    lo[i] = (__bf16)(x[i] - (float)h);
  }
}

// B operands of the RPE GEMM on the bf16 pipe (16x16x32): lane (g, c) holds
// W[16 b + c][8 g .. 8 g + 7], split
__device__ __forceinline__ void load_b_bf(const float* __restrict__ W, int g, int c,
                                          bf16x8 (&Bh)[NB], bf16x8 (&Bl)[NB]) {
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    float w[8];
    const float4 w0 = *reinterpret_cast<const float4*>(W + (size_t)(16 * b + c) * F + 8 * g);
    const float4 w1 = *reinterpret_cast<const float4*>(W + (size_t)(16 * b + c) * F + 8 * g + 4);
    w[0] = w0.x; w[1] = w0.y; w[2] = w0.z; w[3] = w0.w;
    w[4] = w1.x; w[5] = w1.y; w[6] = w1.z; w[7] = w1.w;
    split_bf16<8>(w, Bh[b], Bl[b]);
  }
}

// A operands of a tile on the bf16 pipe: lane (g, c) holds ea[edge c][8 g .. 8 g + 7]
// (two 16-byte chunks, un-swizzled on read), split
__device__ __forceinline__ void load_a_bf(const float* slab, int g, int c, bf16x8& Ah, bf16x8& Al) {
  const float4 a0 = *reinterpret_cast<const float4*>(slab + c * F + (((2 * g) ^ (c & 7)) << 2));
  const float4 a1 = *reinterpret_cast<const float4*>(slab + c * F + (((2 * g + 1) ^ (c & 7)) << 2));
  const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
  split_bf16<8>(a, Ah, Al);
}

// LO = false: plain bf16 operands (the hi halves only) - the "bf16 matrix" precision mode
template <bool LO>
__device__ __forceinline__ void rpe_gemm_bf(const bf16x8& Ah, const bf16x8& Al,
                                            const bf16x8 (&Bh)[NB], const bf16x8 (&Bl)[NB],
                                            const f32x4 (&init)[NB], f32x4 (&C)[NB]) {
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    // the start value is a ready register quad (splat once per launch / per node, read as srcC
    // by the first MFMA) - not four moves per accumulator and tile
    C[b] = init[b];
    if constexpr (LO) {
      C[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, Bh[b], C[b], 0, 0, 0);
      C[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bl[b], C[b], 0, 0, 0);
    }
    C[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bh[b], C[b], 0, 0, 0);
  }
}

// C[b][r] starts at init[b] (bias / node term folded into the accumulator for free)
__device__ __forceinline__ void rpe_gemm(const float (&A)[8], const float (&B)[NB][8],
                                         const float (&init)[NB], f32x4 (&C)[NB]) {
#pragma unroll
  for (int b = 0; b < NB; ++b) C[b] = (f32x4){init[b], init[b], init[b], init[b]};
#pragma unroll
  for (int st = 0; st < 8; ++st)
#pragma unroll
    for (int b = 0; b < NB; ++b)
      C[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[st], B[b][st], C[b], 0, 0, 0);
}

// The three RPE products of a tile with the B operands read from the padded LDS copy of
// [Wk; Wq; Wv] (row n = output column, stride W_LD): lane (g, c) reads
// w[(64 p + 16 b + c) W_LD + 4 st + g], bank = c + g (+ const) - conflict-free up to the
// natural 2 lanes per bank.  Software-pipelined by hand: the 12 operands of k-step st + 1
// are in flight while the 12 MFMAs of step st run, so only 24 of them are ever live (left
// to itself the scheduler hoists all 96 loads and spills into AGPRs).
constexpr int W_LD = F + 1;
__device__ __forceinline__ void rpe_gemm3_lds(const float (&A)[8], const float* w, int g, int c,
                                              const float (&ik)[NB], const float (&iq)[NB],
                                              const float (&iv)[NB], f32x4 (&Ck)[NB],
                                              f32x4 (&Cq)[NB], f32x4 (&Cv)[NB]) {
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    Ck[b] = (f32x4){ik[b], ik[b], ik[b], ik[b]};
    Cq[b] = (f32x4){iq[b], iq[b], iq[b], iq[b]};
    Cv[b] = (f32x4){iv[b], iv[b], iv[b], iv[b]};
  }
  const float* wr = w + c * W_LD + g;
  float cur[3][NB], nxt[3][NB];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int b = 0; b < NB; ++b) cur[p][b] = wr[(64 * p + 16 * b) * W_LD];
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    if (st < 7) {
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int b = 0; b < NB; ++b) nxt[p][b] = wr[(64 * p + 16 * b) * W_LD + 4 * (st + 1)];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      Ck[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[st], cur[0][b], Ck[b], 0, 0, 0);
      Cq[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[st], cur[1][b], Cq[b], 0, 0, 0);
      Cv[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[st], cur[2][b], Cv[b], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int b = 0; b < NB; ++b) cur[p][b] = nxt[p][b];
  }
}

// Three-deep software pipeline over the tiles of a wave's nodes (s, s+nw, ...):
//   cur : being computed out of LDS buffer b
//   nxt : its edge_attr / k / v rows are streaming into buffer b^1
//   nn  : its edge ids / targets and the CSR range of the node after it are in
//         flight in VGPRs
// so no load is ever waited for in the iteration that issued it.  Node ranges
// travel through VMEM (lanes 0/1) rather than SMEM: an outstanding s_load would
// be waited for by every LDS lgkmcnt(0) in between.
struct Pipe {
  Tile cur, nxt, nn;
  int e_cur, t_cur, e_nxt, t_nxt, e_nn, t_nn;  // edge id / target held by lane u < cnt
  float q_nxt[NB];                    // raw q row (columns 16 b + c) of nxt's node
  int rp_ahead;                       // lane 0/1: CSR range of node(nn)+nw (or beyond)
  int64_t s_ahead;                    // the node rp_ahead describes
  int bsel;
  int64_t nw, N;
  const int32_t *rp, *eperm, *tgt;
  const float *qkv, *ea;
  int ld;
  float* base;

  __device__ __forceinline__ float* buf_at(int b) const { return base + b * SLAB; }
  __device__ __forceinline__ const float* cur_buf() const { return buf_at(bsel); }
  __device__ __forceinline__ void load_q(const Tile& d, int lane) {
    if (d.valid && d.t0 == d.start) {
#pragma unroll
      for (int b = 0; b < NB; ++b) q_nxt[b] = qkv[d.s * ld + 16 * b + (lane & 15)];
    }
  }

  __device__ __forceinline__ void load_idx(const Tile& d, int& e_l, int& t_l, int lane) {
    e_l = 0;
    t_l = 0;
    if (d.valid && lane < tile_count(d)) {
      e_l = eperm ? eperm[d.t0 + lane] : d.t0 + lane;
      t_l = tgt[d.t0 + lane];
    }
  }
  __device__ __forceinline__ void load_range(int64_t s, int lane) {
    s_ahead = s;
    rp_ahead = (s < N && lane < 2) ? rp[s + lane] : 0;
  }
  // next tile after d; crossing into node d.s+nw uses the prefetched range
  __device__ __forceinline__ Tile advance(const Tile& d) {
    if (d.valid && d.t0 + TE < d.end) {
      Tile t = d;
      t.t0 += TE;
      return t;
    }
    Tile t;
    t.s = d.s + nw;
    t.valid = d.valid && t.s < N;
    // (s_ahead == t.s by construction)
    t.start = __builtin_amdgcn_readlane(rp_ahead, 0);   // SGPRs: the tile walk is scalar code
    t.end = __builtin_amdgcn_readlane(rp_ahead, 1);
    t.t0 = t.start;
    return t;
  }

  __device__ __forceinline__ void init(int64_t wave, int64_t nwaves, int64_t n,
                                       const int32_t* erowptr, const int32_t* ep,
                                       const int32_t* tg, const float* qkv_, int ld_,
                                       const float* ea_, float* b0, float* b1, int lane) {
    nw = nwaves; N = n; rp = erowptr; eperm = ep; tgt = tg; qkv = qkv_; ld = ld_; ea = ea_;
    base = b0; (void)b1; bsel = 0;
    cur = tile_of_node(wave, N, rp);
    load_idx(cur, e_cur, t_cur, lane);
    load_range(wave + nw, lane);
#pragma unroll
    for (int b = 0; b < NB; ++b) q_nxt[b] = 0.f;
    load_q(cur, lane);   // the kernel reads q_nxt for the very first node
    tile_issue(qkv, ld, ea, e_cur, t_cur, tile_count(cur), buf_at(0), lane);
    nxt = advance(cur);
    load_idx(nxt, e_nxt, t_nxt, lane);
    if (nxt.valid && nxt.t0 == nxt.start) load_range(nxt.s + nw, lane);
    // nn is produced by the first top()
    nn = nxt;
    e_nn = t_nn = 0;
  }

  // called when `cur` is about to be computed
  __device__ __forceinline__ void top(int lane) {
    wait_vmem_all();
    if (nxt.valid) tile_issue(qkv, ld, ea, e_nxt, t_nxt, tile_count(nxt), buf_at(bsel ^ 1), lane);
    load_q(nxt, lane);
    nn = advance(nxt);
    load_idx(nn, e_nn, t_nn, lane);
    if (nn.valid && nn.t0 == nn.start) load_range(nn.s + nw, lane);
  }
  __device__ __forceinline__ void rotate() {
    cur = nxt;
    nxt = nn;
    e_cur = e_nxt;
    t_cur = t_nxt;
    e_nxt = e_nn;
    t_nxt = t_nn;
    bsel ^= 1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  }
};

// PREC: 0 = f32 matrix pipe, 3 = split-bf16 (3 products), 1 = plain bf16 operands
template <int PREC>
__global__ __launch_bounds__(WAVES * 64, 2) void attn_fwd_mfma_kernel(
    const float* __restrict__ qkv, int ld, int64_t N, const int32_t* __restrict__ erowptr,
    const int32_t* __restrict__ eperm, const int32_t* __restrict__ tgt,
    const float* __restrict__ ea, const float* __restrict__ Wk, const float* __restrict__ bk,
    const float* __restrict__ Wq, const float* __restrict__ bq, const float* __restrict__ Wv,
    const float* __restrict__ bv, int scale_mode, float scale_a, float* __restrict__ out,
    float* __restrict__ mbuf, float* __restrict__ zbuf, int bands) {
  constexpr bool BF3 = PREC != 0, LO = PREC == 3;
  __shared__ __attribute__((aligned(16))) float slab_all[WAVES][2][SLAB];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  float Bk[BF3 ? 1 : NB][8], Bq[BF3 ? 1 : NB][8], Bv[BF3 ? 1 : NB][8], bk4[NB], bq4[NB], bv4[NB];
  bf16x8 Bkh[NB], Bkl[NB], Bqh[NB], Bql[NB], Bvh[NB], Bvl[NB];
  if constexpr (BF3) {
    load_b_bf(Wk, g, c, Bkh, Bkl);
    load_b_bf(Wq, g, c, Bqh, Bql);
    load_b_bf(Wv, g, c, Bvh, Bvl);
  } else {
    load_b(Wk, g, c, Bk);
    load_b(Wq, g, c, Bq);
    load_b(Wv, g, c, Bv);
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    bk4[b] = bk ? bk[16 * b + c] : 0.f;
    bq4[b] = bq ? bq[16 * b + c] : 0.f;
    bv4[b] = bv ? bv[16 * b + c] : 0.f;
  }
  for (int i = lane; i < 2 * SLAB; i += 64) (&slab_all[wid][0][0])[i] = 0.f;  // stale rows stay finite
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  int64_t wave = (int64_t)blockIdx.x * WAVES + wid;
  int64_t nwaves = (int64_t)gridDim.x * WAVES;
  if (bands) {
    // Round 6: XCD bands.  Workgroup b runs on XCD b % 8 (observed placement; only speed depends
    // on it) and every XCD has its own L2: XCD x takes the contiguous EIGHTH [x N / 8, (x + 1) N / 8)
    // of the nodes and its resident waves walk it node by node, so the k | v rows the band's
    // nodes gather - their spatial neighbours, a few hundred rows around them in the MortonOrder
    // layout - are fetched into ONE L2 once instead of into all eight (the plain grid stride
    // deals consecutive nodes to consecutive XCDs).  Per-node results do not depend on the walk.
    const int x = blockIdx.x & 7;
    const int64_t npx = (N + 7) >> 3, lim = (x + 1) * npx < N ? (x + 1) * npx : N;
    wave = x * npx + (int64_t)(blockIdx.x >> 3) * WAVES + wid;
    nwaves = (int64_t)(gridDim.x >> 3) * WAVES;
    N = lim;
  }
  if (wave >= N) return;
  Pipe P;
  P.init(wave, nwaves, N, erowptr, eperm, tgt, qkv, ld, ea, slab_all[wid][0], slab_all[wid][1], lane);

  // Round 6 (the kernel was VALU-bound: ~590 VALU instructions per tile, 1.45 tiles per node):
  //   * the accumulators' start values as ready register quads (bias splats once per launch, the
  //     node's q * scale + bq once per node);
  //   * the head dot products' quad sums as v_add_f32_dpp (quad_sum4);
  //   * the softmax in the base-2 domain: the running maximum is kept as ms = max(p) * log2(e) and
  //     a weight is exp2(fma(p, log2(e), -ms)) - one instruction less per exponential; every
  //     weight, z and the rescaling of a node share the same ms, so its rounding cancels in
  //     acc / z, and m = ms * ln(2) goes out for the backward (m + log z is the same logsumexp to
  //     |m| 2^-24);
  //   * out = acc * rcp(z + 1e-16) (1 ulp) instead of the IEEE division sequence.
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  //   * the value bias leaves the per-edge path: sum_e a_e (v_e + bv) = sum_e a_e v_e + bv z, so
  //     out = acc / z + bv (the f32-pipe variant keeps it in the accumulator's start value);
  f32x4 bk4v[NB], qs4v[NB], zero4v[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    bk4v[b] = (f32x4){bk4[b], bk4[b], bk4[b], bk4[b]};
    qs4v[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    zero4v[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const float bvg = g == 0 ? bv4[0] : (g == 1 ? bv4[1] : (g == 2 ? bv4[2] : bv4[3]));
  float qs4[NB], m[NB], z[NB], acc[NB];       // m: the running maximum times log2(e)
  while (P.cur.valid) {
    wait_vmem_all();  // tile `cur`, nxt's indices and cur's q row have landed
    float qraw[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) qraw[b] = P.q_nxt[b];
    P.top(lane);      // issues nxt's data, nn's indices, nxt's q row, range prefetch
    const Tile cur = P.cur;
    const float* slab = P.cur_buf();
    const float* kslab = slab + EA_FLOATS;
    const float* vslab = kslab + TE * ROW;
    const int cnt = tile_count(cur);
    if (cur.t0 == cur.start) {
      const int deg = cur.end - cur.start;
      const float scale = deg > 0 ? qk_scale_of(scale_mode, scale_a, deg) : 0.f;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        qs4[b] = fmaf(qraw[b], scale, bq4[b]);  // q_s*scale + bq
        qs4v[b] = (f32x4){qs4[b], qs4[b], qs4[b], qs4[b]};
        m[b] = -INFINITY;
        z[b] = 0.f;
        acc[b] = 0.f;
      }
    }
    if (cnt > 0) {
      f32x4 Ck[NB], Cq[NB], Cv[NB];
      if constexpr (BF3) {
        bf16x8 Ah, Al;
        load_a_bf(slab, g, c, Ah, Al);
        rpe_gemm_bf<LO>(Ah, Al, Bkh, Bkl, bk4v, Ck);
        rpe_gemm_bf<LO>(Ah, Al, Bqh, Bql, qs4v, Cq);
        rpe_gemm_bf<LO>(Ah, Al, Bvh, Bvl, zero4v, Cv);      // (the compiler folds the zeros into the MFMA's inline 0)
      } else {
        float A[8];
        load_a(slab, g, c, A);
        rpe_gemm(A, Bk, bk4, Ck);        // k_e - k_t   = Wk ea + bk
        rpe_gemm(A, Bq, qs4, Cq);        // q_e         = Wq ea + bq + q_s scale
        rpe_gemm(A, Bv, bv4, Cv);        // v_e - v_t   = Wv ea + bv
      }
      const bool partial = cnt < TE;   // wave-uniform
      const float* kp = kslab + 4 * g * ROW + c;
      const float* vp = vslab + 4 * g * ROW + c;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        // pairs of edges as packed f32 operations (v_pk_add / v_pk_mul / v_pk_fma: two lanes' worth
        // per issue slot)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 k01 = {kp[16 * b], kp[ROW + 16 * b]}, k23 = {kp[2 * ROW + 16 * b], kp[3 * ROW + 16 * b]};
        const f32x2 t01 = (f32x2){Cq[b][0], Cq[b][1]} * ((f32x2){Ck[b][0], Ck[b][1]} + k01);
        const f32x2 t23 = (f32x2){Cq[b][2], Cq[b][3]} * ((f32x2){Ck[b][2], Ck[b][3]} + k23);
        const float qk[4] = {t01.x, t01.y, t23.x, t23.y};
        float p[4];
        quad_sum4(qk, p);
        if (partial) {
#pragma unroll
          for (int r = 0; r < 4; ++r) p[r] = (4 * g + r < cnt) ? p[r] : -INFINITY;
        }
        const float mt = xg_max_sw(vmax3(p[0], p[1], vmax(p[2], p[3])));  // tile max of this head
        const float mn = vmax(m[b], mt * LOG2E);
        const float corr = __builtin_amdgcn_exp2f(m[b] - mn);  // m = -inf on the first tile -> 0
        f32x2 pe01, pe23;                                      // rows beyond cnt: 2^-inf = 0
        pe01.x = __builtin_amdgcn_exp2f(fmaf(p[0], LOG2E, -mn));
        pe01.y = __builtin_amdgcn_exp2f(fmaf(p[1], LOG2E, -mn));
        pe23.x = __builtin_amdgcn_exp2f(fmaf(p[2], LOG2E, -mn));
        pe23.y = __builtin_amdgcn_exp2f(fmaf(p[3], LOG2E, -mn));
        const f32x2 v01 = (f32x2){Cv[b][0], Cv[b][1]} + (f32x2){vp[16 * b], vp[ROW + 16 * b]};
        const f32x2 v23 = (f32x2){Cv[b][2], Cv[b][3]} + (f32x2){vp[2 * ROW + 16 * b], vp[3 * ROW + 16 * b]};
        const f32x2 w2 = __builtin_elementwise_fma(pe23, v23, pe01 * v01);
        const f32x2 z2 = pe01 + pe23;
        // z / acc stay per-lane-group partial sums until the node ends
        z[b] = fmaf(z[b], corr, z2.x + z2.y);
        acc[b] = fmaf(acc[b], corr, w2.x + w2.y);
        m[b] = mn;
      }
    }
    if (cur.t0 + TE >= cur.end) {  // last tile of the node
      // Reduce-scatter over the four lane groups (round 6): block b's totals are only needed where
      // they are stored - lane group b writes out[s][16 b + c], i.e. the wave writes the node's
      // 256-byte row with ONE instruction.  permlane16_swap(X0, X1) -> {[X0.r0 X1.r0 X0.r2 X1.r2],
      // [X0.r1 X1.r1 X0.r3 X1.r3]}: their sum holds X0's (rows 0, 2) and X1's (rows 1, 3) partial
      // sums of two groups; the same between the two pair sums with permlane32_swap leaves block
      // g's total in lane group g - 3 swaps + 3 adds for four values and no register copies
      // (8 instructions PER VALUE with the all-to-all sums, 4 x 16-lane stores).
      auto rscat = [](float x0, float x1, float x2, float x3) {
        const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x0), __float_as_uint(x1), false, false);
        const auto bq = __builtin_amdgcn_permlane16_swap(__float_as_uint(x2), __float_as_uint(x3), false, false);
        const float p01 = __uint_as_float(a[0]) + __uint_as_float(a[1]);
        const float p23 = __uint_as_float(bq[0]) + __uint_as_float(bq[1]);
        const auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(p01), __float_as_uint(p23), false, false);
        return __uint_as_float(t[0]) + __uint_as_float(t[1]);
      };
      const float zt = rscat(z[0], z[1], z[2], z[3]);          // lane group g: block g
      const float at = rscat(acc[0], acc[1], acc[2], acc[3]);
      const float mg = g == 0 ? m[0] : (g == 1 ? m[1] : (g == 2 ? m[2] : m[3]));
      // (a node without edges keeps out = 0: no weight mass, no bias)
      const float bvn = (BF3 && cur.end > cur.start) ? bvg : 0.f;
      out[cur.s * 64 + lane] = fmaf(at, __builtin_amdgcn_rcpf(zt + 1e-16f), bvn);
      if ((c & 3) == 0 && mbuf) {
        mbuf[cur.s * 16 + 4 * g + (c >> 2)] = mg * LN2;
        zbuf[cur.s * 16 + 4 * g + (c >> 2)] = zt;
      }
    }
    P.rotate();
  }
}

// ---- backward ------------------------------------------------------------------
constexpr int DT_STRIDE = 196;                 // transposed D tile: [16 edges][192 (+4 pad)]
constexpr int DT_FLOATS = TE * DT_STRIDE;      // 3136 floats = 12.25 KB

constexpr int W_FLOATS = 192 * (F + 1);   // [Wk; Wq; Wv] rows (padded), shared by the 4 waves:
                                          // B operands of the recompute GEMM and of D W

// split-bf16 variant: padded bf16 copies of [Wk; Wq; Wv] (row n = output column, 32 + 8 bf16
// per row: 80-byte rows put the 16 lanes of a group on 16 distinct 4-bank windows)
constexpr int WB_LD = F + 8;
constexpr int WB_ELEMS = 192 * WB_LD;
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int PREC>
__global__ __launch_bounds__(WAVES * 64, 1) void attn_bwd_mfma_kernel(
    const float* __restrict__ qkv, int ld, int64_t N, const int32_t* __restrict__ erowptr,
    const int32_t* __restrict__ eperm, const int32_t* __restrict__ tgt,
    const float* __restrict__ ea, const float* __restrict__ Wk, const float* __restrict__ bk,
    const float* __restrict__ Wq, const float* __restrict__ bq, const float* __restrict__ Wv,
    const float* __restrict__ bv, int scale_mode, float scale_a, const float* __restrict__ out,
    const float* __restrict__ mbuf, const float* __restrict__ zbuf,
    const float* __restrict__ gout, float* __restrict__ gqkv, float* __restrict__ gea,
    float* __restrict__ partial, int gea_acc) {
  constexpr bool BF3 = PREC != 0, LO = PREC == 3;
  __shared__ __attribute__((aligned(16))) float slab_all[WAVES][2][SLAB];
  __shared__ __attribute__((aligned(16))) float dt_all[WAVES][DT_FLOATS];
  __shared__ __attribute__((aligned(16))) float w_lds[BF3 ? 4 : W_FLOATS];
  __shared__ __attribute__((aligned(16))) __bf16 wb_hi[BF3 ? WB_ELEMS : 8];
  __shared__ __attribute__((aligned(16))) __bf16 wb_lo[BF3 ? WB_ELEMS : 8];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  float* dt = dt_all[wid];
  for (int i = threadIdx.x; i < 64 * F; i += WAVES * 64) {
    const int n = i / F, f = i - n * F;
    if constexpr (BF3) {
      const float w3[3] = {Wk[i], Wq[i], Wv[i]};
#pragma unroll
      for (int p3 = 0; p3 < 3; ++p3) {
        const __bf16 h = (__bf16)w3[p3];
        wb_hi[(64 * p3 + n) * WB_LD + f] = h;
        wb_lo[(64 * p3 + n) * WB_LD + f] = (__bf16)(w3[p3] - (float)h);
      }
    } else {
      w_lds[n * W_LD + f] = Wk[i];
      w_lds[(64 + n) * W_LD + f] = Wq[i];
      w_lds[(128 + n) * W_LD + f] = Wv[i];
    }
  }
  __syncthreads();
  // split-bf16: B operands of d edge_attr = D W live in registers for the whole launch:
  // lane (g, c) holds W[o = 32 s + 8 g .. + 7][f = 16 fb + c]  (o runs over [Wk; Wq; Wv] rows)
  bf16x8 W2h[BF3 ? 6 : 1][2], W2l[BF3 ? 6 : 1][2];
  if constexpr (BF3) {
#pragma unroll
    for (int sg = 0; sg < 6; ++sg)
#pragma unroll
      for (int fb = 0; fb < 2; ++fb) {
        float w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int o = 32 * sg + 8 * g + i;
          const float* Wp = o < 64 ? Wk : (o < 128 ? Wq : Wv);
          w[i] = Wp[(size_t)(o & 63) * F + 16 * fb + c];
        }
        split_bf16<8>(w, W2h[sg][fb], W2l[sg][fb]);
      }
  }

  float bk4[NB], bq4[NB], bv4[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    bk4[b] = bk ? bk[16 * b + c] : 0.f;
    bq4[b] = bq ? bq[16 * b + c] : 0.f;
    bv4[b] = bv ? bv[16 * b + c] : 0.f;
  }
  // weight-gradient accumulators: C3[ob][fb][r] = dW[16 ob + 4 g + r][16 fb + c]
  f32x4 C3[3 * NB][2];
  float gb[3 * NB];   // bias gradients, per-lane-group partials for o = 16 ob + c
#pragma unroll
  for (int ob = 0; ob < 3 * NB; ++ob) {
    C3[ob][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    C3[ob][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    gb[ob] = 0.f;
  }
  for (int i = lane; i < 2 * SLAB; i += 64) (&slab_all[wid][0][0])[i] = 0.f;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  const int64_t wave = (int64_t)blockIdx.x * WAVES + wid;
  const int64_t nwaves = (int64_t)gridDim.x * WAVES;
  if (wave < N) {
    Pipe P;
    P.init(wave, nwaves, N, erowptr, eperm, tgt, qkv, ld, ea, slab_all[wid][0], slab_all[wid][1],
           lane);
    float qs4[NB], m[NB], zi[NB], delta[NB], g4[NB], dqa[NB];
    // rows of the NEXT node (gout, out, m, z), fetched one tile ahead: read at the node's
    // first tile they cost a full memory latency per node with nothing to hide it behind
    float g4n[NB], on[NB], mn[NB], zn[NB];
    auto fetch_node_rows = [&](int64_t sn) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        g4n[b] = gout[sn * 64 + 16 * b + c];
        on[b] = out[sn * 64 + 16 * b + c];
        mn[b] = mbuf[sn * 16 + 4 * b + (c >> 2)];
        zn[b] = zbuf[sn * 16 + 4 * b + (c >> 2)];
      }
    };
    fetch_node_rows(wave);
    float scale = 0.f;
    while (P.cur.valid) {
      wait_vmem_all();
      float qraw[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) qraw[b] = P.q_nxt[b];
      const int e_cur = P.e_cur, t_cur = P.t_cur;
      P.top(lane);
      const Tile cur = P.cur;
      const float* slab = P.cur_buf();
      const float* kslab = slab + EA_FLOATS;
      const float* vslab = kslab + TE * ROW;
      const int cnt = tile_count(cur);
      const int64_t s = cur.s;
      if (cur.t0 == cur.start) {  // first tile of the node
        const int deg = cur.end - cur.start;
        scale = deg > 0 ? qk_scale_of(scale_mode, scale_a, deg) : 0.f;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          qs4[b] = fmaf(qraw[b], scale, bq4[b]);
          dqa[b] = 0.f;
          g4[b] = g4n[b];
          delta[b] = quad_sum(g4[b] * on[b]);                       // <g, out> per head
          m[b] = mn[b];
          zi[b] = 1.0f / (zn[b] + 1e-16f);
        }
      }
      if (P.nxt.valid && P.nxt.t0 == P.nxt.start) fetch_node_rows(P.nxt.s);
      if (cnt > 0) {
        f32x4 Ck[NB], Cq[NB], Cv[NB];
        if constexpr (BF3) {
          bf16x8 Ah, Al;
          load_a_bf(slab, g, c, Ah, Al);
          const __bf16* wh = wb_hi + c * WB_LD + 8 * g;
          const __bf16* wl = wb_lo + c * WB_LD + 8 * g;
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            Ck[b] = (f32x4){bk4[b], bk4[b], bk4[b], bk4[b]};
            Cq[b] = (f32x4){qs4[b], qs4[b], qs4[b], qs4[b]};
            Cv[b] = (f32x4){bv4[b], bv4[b], bv4[b], bv4[b]};
          }
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const bf16x8 kh = *reinterpret_cast<const bf16x8*>(wh + (16 * b) * WB_LD);
            const bf16x8 kl = *reinterpret_cast<const bf16x8*>(wl + (16 * b) * WB_LD);
            const bf16x8 qh = *reinterpret_cast<const bf16x8*>(wh + (64 + 16 * b) * WB_LD);
            const bf16x8 ql = *reinterpret_cast<const bf16x8*>(wl + (64 + 16 * b) * WB_LD);
            const bf16x8 vh = *reinterpret_cast<const bf16x8*>(wh + (128 + 16 * b) * WB_LD);
            const bf16x8 vl = *reinterpret_cast<const bf16x8*>(wl + (128 + 16 * b) * WB_LD);
            if constexpr (LO) {
              Ck[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, kh, Ck[b], 0, 0, 0);
              Cq[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, qh, Cq[b], 0, 0, 0);
              Cv[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, vh, Cv[b], 0, 0, 0);
              Ck[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, kl, Ck[b], 0, 0, 0);
              Cq[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, ql, Cq[b], 0, 0, 0);
              Cv[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, vl, Cv[b], 0, 0, 0);
            }
            Ck[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, kh, Ck[b], 0, 0, 0);
            Cq[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, qh, Cq[b], 0, 0, 0);
            Cv[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, vh, Cv[b], 0, 0, 0);
          }
        } else {
          float A[8];
          load_a(slab, g, c, A);
          rpe_gemm3_lds(A, w_lds, g, c, bk4, qs4, bv4, Ck, Cq, Cv);
        }
        const float* kp = kslab + 4 * g * ROW + c;
        const float* vp = vslab + 4 * g * ROW + c;
        int64_t trow[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) trow[r] = (int64_t)__shfl(t_cur, 4 * g + r, 64) * ld;
        // ---- per-edge gradients, in place: Ck <- dk, Cq <- dq, Cv <- dv ------------
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool valid = 4 * g + r < cnt;
            const float k = Ck[b][r] + kp[r * ROW + 16 * b];
            const float q = Cq[b][r];
            const float v = Cv[b][r] + vp[r * ROW + 16 * b];
            const float p = quad_sum(q * k);
            const float a = valid ? __expf(p - m[b]) * zi[b] : 0.f;
            const float da = quad_sum(g4[b] * v);
            const float dc = a * (da - delta[b]);
            const float dk = dc * q, dq = dc * k, dv = a * g4[b];
            Ck[b][r] = dk;
            Cq[b][r] = dq;
            Cv[b][r] = dv;
            dqa[b] += dq;
            gb[b] += dk;
            gb[NB + b] += dq;
            gb[2 * NB + b] += dv;
          }
        }
        // dk / dv go to the target rows.  One branch per edge row r (validity depends on
        // (g, r) only) instead of one per (b, r): the 16 iterations above stay one basic
        // block the scheduler can interleave.
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (4 * g + r < cnt) {
            float* row = gqkv + trow[r] + c;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
              unsafeAtomicAdd(row + 64 + 16 * b, Ck[b][r]);
              unsafeAtomicAdd(row + 128 + 16 * b, Cv[b][r]);
            }
          }
        }
        // ---- d edge_attr = D W : D transposed through LDS (before dW: its stores and the
        //      atomics above then drain under the 96 MFMAs of the weight-gradient GEMM) ----
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* row = dt + (4 * g + r) * DT_STRIDE + 16 * b + c;
            row[0] = Ck[b][r];
            row[64] = Cq[b][r];
            row[128] = Cv[b][r];
          }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        f32x4 C2[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
        if constexpr (BF3) {
          // A[i = edge c][k = o = 32 s + 8 g .. + 7] out of the transposed tile, split on the fly
          const float* arow = dt + c * DT_STRIDE + 8 * g;
#pragma unroll
          for (int sg = 0; sg < 6; ++sg) {
            const float4 d0 = *reinterpret_cast<const float4*>(arow + 32 * sg);
            const float4 d1 = *reinterpret_cast<const float4*>(arow + 32 * sg + 4);
            const float dd[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
            bf16x8 dh, dl;
            split_bf16<8>(dd, dh, dl);
#pragma unroll
            for (int fb = 0; fb < 2; ++fb) {
              if constexpr (LO) {
                C2[fb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dl, W2h[sg][fb], C2[fb], 0, 0, 0);
                C2[fb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dh, W2l[sg][fb], C2[fb], 0, 0, 0);
              }
              C2[fb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dh, W2h[sg][fb], C2[fb], 0, 0, 0);
            }
          }
        } else {
        const float* arow = dt + c * DT_STRIDE + g;       // A[i = edge c][k = 4 st + g]
        const float* brow = w_lds + g * W_LD + c;         // B[k = 4 st + g][j = 16 fb + c]
        // operands of the next 4 k-steps in flight under the 8 MFMAs of the current 4
        {
          float ac[4], b0c[4], b1c[4], an[4], b0n[4], b1n[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            ac[u] = arow[4 * u];
            b0c[u] = brow[4 * u * W_LD];
            b1c[u] = brow[4 * u * W_LD + 16];
          }
#pragma unroll
          for (int grp = 0; grp < 12; ++grp) {
            if (grp < 11) {
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const int st = 4 * (grp + 1) + u;
                an[u] = arow[4 * st];
                b0n[u] = brow[4 * st * W_LD];
                b1n[u] = brow[4 * st * W_LD + 16];
              }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              C2[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[u], b0c[u], C2[0], 0, 0, 0);
              C2[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[u], b1c[u], C2[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) { ac[u] = an[u]; b0c[u] = b0n[u]; b1c[u] = b1n[u]; }
          }
        }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t e = __shfl(e_cur, 4 * g + r, 64);
          if (4 * g + r < cnt) {
            if (gea_acc) {          // shared gradient buffer of the stage's blocks
              unsafeAtomicAdd(gea + e * F + c, C2[0][r]);
              unsafeAtomicAdd(gea + e * F + 16 + c, C2[1][r]);
            } else {
              gea[e * F + c] = C2[0][r];
              gea[e * F + 16 + c] = C2[1][r];
            }
          }
        }
        // ---- dW += D^T EA : A operand = the C-layout registers as they are ----------
        if constexpr (BF3) {
          // contraction index = edge 4 g + r: the 4 registers of a block, packed, are the A
          // operand of a 16x16x16 bf16 product; B = ea[edge 4 g + r][16 fb + c]
          s16x4 Eh[2], El[2];
#pragma unroll
          for (int fb = 0; fb < 2; ++fb) {
            float ev[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int e = 4 * g + r, f = 16 * fb + c;
              ev[r] = slab[e * F + (((f >> 2) ^ (e & 7)) << 2) + (f & 3)];
            }
            bf16x4 eh, el;
            split_bf16<4>(ev, eh, el);
            Eh[fb] = __builtin_bit_cast(s16x4, eh);
            El[fb] = __builtin_bit_cast(s16x4, el);
          }
#pragma unroll
          for (int ob = 0; ob < 3 * NB; ++ob) {
            const f32x4& Dsrc = ob < NB ? Ck[ob % NB] : (ob < 2 * NB ? Cq[ob % NB] : Cv[ob % NB]);
            const float dv4[4] = {Dsrc[0], Dsrc[1], Dsrc[2], Dsrc[3]};
            bf16x4 dh, dl;
            split_bf16<4>(dv4, dh, dl);
            const s16x4 Dh = __builtin_bit_cast(s16x4, dh), Dl = __builtin_bit_cast(s16x4, dl);
#pragma unroll
            for (int fb = 0; fb < 2; ++fb) {
              if constexpr (LO) {
                C3[ob][fb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Dl, Eh[fb], C3[ob][fb], 0, 0, 0);
                C3[ob][fb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Dh, El[fb], C3[ob][fb], 0, 0, 0);
              }
              C3[ob][fb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Dh, Eh[fb], C3[ob][fb], 0, 0, 0);
            }
          }
        } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float eb[2];
#pragma unroll
          for (int fb = 0; fb < 2; ++fb) {
            const int e = 4 * g + r, f = 16 * fb + c;
            eb[fb] = slab[e * F + (((f >> 2) ^ (e & 7)) << 2) + (f & 3)];
          }
#pragma unroll
          for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int fb = 0; fb < 2; ++fb) {
              C3[b][fb] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ck[b][r], eb[fb], C3[b][fb], 0, 0, 0);
              C3[NB + b][fb] = __builtin_amdgcn_mfma_f32_16x16x4f32(Cq[b][r], eb[fb], C3[NB + b][fb], 0, 0, 0);
              C3[2 * NB + b][fb] = __builtin_amdgcn_mfma_f32_16x16x4f32(Cv[b][r], eb[fb], C3[2 * NB + b][fb], 0, 0, 0);
            }
          }
        }
        }
      }
      if (cur.t0 + TE >= cur.end) {  // last tile of the node
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const float dq = xg_sum(dqa[b]);
          if (g == 0) gqkv[s * ld + 16 * b + c] = dq * scale;
        }
      }
      P.rotate();
    }
  }
  // per-wave partial tables [192 rows][F + 1]: weight block + bias column
  if (partial) {
    float* pw = partial + (size_t)wave * 192 * (F + 1);
#pragma unroll
    for (int ob = 0; ob < 3 * NB; ++ob) {
#pragma unroll
      for (int fb = 0; fb < 2; ++fb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          pw[(size_t)(16 * ob + 4 * g + r) * (F + 1) + 16 * fb + c] = C3[ob][fb][r];
      const float bsum = xg_sum(gb[ob]);
      if (g == 0) pw[(size_t)(16 * ob + c) * (F + 1) + F] = bsum;
    }
  }
}


// =====================================================================================
// Packed backward: 16-edge tiles over the EDGE stream, not per node.
//
// The kernels above give every source node its own tiles: at a mean degree of 16.4 a tile is
// 62-69 % full (a node of degree 17 costs two tiles), and since the backward became VALU bound
// (split-bf16) the empty rows are pure loss.  The backward does not need the running softmax
// state of the forward - m and z of every node are saved - so nothing ties a tile to one node:
// here tile t holds edges [16 t, 16 t + 16) of the CSR-by-source order, whatever nodes they
// belong to.  A tile is processed in passes over at most TWO consecutive node contexts
// (A = rows below `split`, B = rows from `split` on; per-row select of the node quantities);
// tiles holding more than two nodes (rare: both of degree < 16 and a third one starting) run
// another pass with the rows of the finished nodes masked.  A wave owns a contiguous range of
// tiles: a node cut by a range boundary gets its dq from two waves, hence dq by atomicAdd
// (gqkv is zero-filled by the entry point); dk / dv were atomic already.  The node owning a
// wave's first edge comes from one binary search of the CSR pointers at the wave's start; from
// there the nodes are walked in order.
// -DSPT_ATTN_PROFILE: per-section cycle counts (s_memtime) of one wave, printed at its end -
// a measurement build only (tools/attn_microbench.py against gpurun_variants/), never shipped
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

struct NodeCtx {
  // ml = m + log(z + 1e-16): the softmax weight of an edge is exp(p - ml) (one value and one
  // select per head instead of m and 1/z)
  float qs[NB], g4[NB], delta[NB], ml[NB], dqa[NB];
  float scale;
  int64_t node;
  int end;                                 // rp[node + 1]
};

template <int PREC>
__global__ __launch_bounds__(WAVES * 64, 1) void attn_bwd_packed_kernel(
    const float* __restrict__ qkv, int ld, int64_t N, int64_t E,
    const int32_t* __restrict__ erowptr, const int32_t* __restrict__ eperm,
    const int32_t* __restrict__ tgt, int64_t ntiles, const float* __restrict__ ea, const float* __restrict__ Wk, const float* __restrict__ bk,
    const float* __restrict__ Wq, const float* __restrict__ bq, const float* __restrict__ Wv,
    const float* __restrict__ bv, int scale_mode, float scale_a, const float* __restrict__ out,
    const float* __restrict__ mbuf, const float* __restrict__ zbuf,
    const float* __restrict__ gout, float* __restrict__ gqkv, float* __restrict__ gea,
    float* __restrict__ partial, int gea_acc) {
  static_assert(PREC == 1 || PREC == 3, "bf16 matrix pipe only");
  constexpr bool LO = PREC == 3;
  __shared__ __attribute__((aligned(16))) float slab_all[WAVES][2][SLAB];
  __shared__ __attribute__((aligned(16))) float dt_all[WAVES][DT_FLOATS];
  __shared__ __attribute__((aligned(16))) int ids_all[WAVES][2][2 * TE];   // the last free KB
  __shared__ __attribute__((aligned(16))) __bf16 wb_hi[WB_ELEMS];
  __shared__ __attribute__((aligned(16))) __bf16 wb_lo[WB_ELEMS];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  float* dt = dt_all[wid];
  for (int i = threadIdx.x; i < 64 * F; i += WAVES * 64) {
    const int n = i / F, f = i - n * F;
    const float w3[3] = {Wk[i], Wq[i], Wv[i]};
#pragma unroll
    for (int p3 = 0; p3 < 3; ++p3) {
      const __bf16 h = (__bf16)w3[p3];
      wb_hi[(64 * p3 + n) * WB_LD + f] = h;
      wb_lo[(64 * p3 + n) * WB_LD + f] = (__bf16)(w3[p3] - (float)h);
    }
  }
  __syncthreads();
  bf16x8 W2h[6][2], W2l[6][2];
#pragma unroll
  for (int sg = 0; sg < 6; ++sg)
#pragma unroll
    for (int fb = 0; fb < 2; ++fb) {
      float w[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int o = 32 * sg + 8 * g + i;
        const float* Wp = o < 64 ? Wk : (o < 128 ? Wq : Wv);
        w[i] = Wp[(size_t)(o & 63) * F + 16 * fb + c];
      }
      split_bf16<8>(w, W2h[sg][fb], W2l[sg][fb]);
    }
  float bk4[NB], bq4[NB], bv4[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    bk4[b] = bk ? bk[16 * b + c] : 0.f;
    bq4[b] = bq ? bq[16 * b + c] : 0.f;
    bv4[b] = bv ? bv[16 * b + c] : 0.f;
  }
  f32x4 C3[3 * NB][2];
  float gb[3 * NB];
#pragma unroll
  for (int ob = 0; ob < 3 * NB; ++ob) {
    C3[ob][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    C3[ob][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    gb[ob] = 0.f;
  }
  for (int i = lane; i < 2 * SLAB; i += 64) (&slab_all[wid][0][0])[i] = 0.f;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  const int64_t wave = (int64_t)blockIdx.x * WAVES + wid;
  const int64_t nwaves = (int64_t)gridDim.x * WAVES;
  const int64_t tpw = (ntiles + nwaves - 1) / nwaves;
  const int64_t t_begin = wave * tpw;
  const int64_t t_end = (t_begin + tpw < ntiles) ? t_begin + tpw : ntiles;
  if (t_begin < t_end) {
    // ---- node contexts ------------------------------------------------------------------
    // A = the node owning the next unprocessed row (ready).  B = the node after it: its rows
    // are requested as soon as A is known and land IN B's own fields (qs <- raw q, delta <- out
    // row, ml <- m, dqa <- z), converted in place once they are there.  Converting right after
    // the top-of-iteration wait is free; anywhere else the compiler's own vmcnt for those loads
    // also waits for the tile DMA issued in between (it cannot see the asm loads): a stall, paid
    // only when a tile holds a second node boundary.
    NodeCtx A, B;
    int b_rp0 = 0;                                      // rowptr[B.node] while B is raw
    int b_state = 0;                                    // 0 none, 1 requested this iteration,
                                                        // 2 requested earlier (landed), 3 ready
    bool a_ready = false;
    auto fetch_b = [&](int64_t nd) {
      B.node = nd;
      B.end = 0;
      b_rp0 = 0;
      if (nd < N) {
        b_rp0 = erowptr[nd];
        B.end = erowptr[nd + 1];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          B.qs[b] = qkv[nd * ld + 16 * b + c];
          B.g4[b] = gout[nd * 64 + 16 * b + c];
          B.delta[b] = out[nd * 64 + 16 * b + c];
          B.ml[b] = mbuf[nd * 16 + 4 * b + (c >> 2)];
          B.dqa[b] = zbuf[nd * 16 + 4 * b + (c >> 2)];
        }
      }
      b_state = 1;
    };
    auto convert_b = [&]() {                            // raw rows -> context, in place
      int s0 = __builtin_amdgcn_readfirstlane(b_rp0), s1 = __builtin_amdgcn_readfirstlane(B.end);
      while (B.node < N && s1 == s0) {                  // nodes without edges own no row: skip
        fetch_b(B.node + 1);
        wait_vmem_all();
        s0 = __builtin_amdgcn_readfirstlane(b_rp0);
        s1 = __builtin_amdgcn_readfirstlane(B.end);
      }
      B.end = s1;
      const int deg = s1 - s0;
      B.scale = deg > 0 ? qk_scale_of(scale_mode, scale_a, deg) : 0.f;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        B.qs[b] = fmaf(B.qs[b], B.scale, bq4[b]);
        B.delta[b] = quad_sum(B.g4[b] * B.delta[b]);
        B.ml[b] = B.ml[b] + __logf(B.dqa[b] + 1e-16f);
        B.dqa[b] = 0.f;
      }
      b_state = 3;
    };
    auto finish_node = [&](NodeCtx& X) {                // dq of a node whose last row is done
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float dq = xg_sum(X.dqa[b]);
        if (g == 0) unsafeAtomicAdd(gqkv + X.node * ld + 16 * b + c, dq * X.scale);
      }
    };
    auto pop_node = [&]() {                              // A is finished: B takes over
      if (b_state != 3) convert_b();
      A = B;                                             // carries B's partial dq when it started
      fetch_b(A.node + 1);
    };
    {
      // node owning the wave's first edge: largest a with rowptr[a] <= e (then rowptr[a+1] > e)
      const int e_first = (int)(t_begin * TE);
      int64_t blo = 0, bhi = N;
      while (bhi - blo > 1) {
        const int64_t mid = (blo + bhi) >> 1;
        if (__builtin_amdgcn_readfirstlane(erowptr[mid]) <= e_first) blo = mid; else bhi = mid;
      }
      fetch_b(blo);
    }
    // ---- tile pipeline: cur in LDS buffer bsel, nxt streaming into bsel ^ 1, indices of the
    //      tile after that in VGPRs -----------------------------------------------------------
    // The ids of a tile (edge rows, targets) travel by LDS-DMA too, two tiles ahead, into a
    // two-slot ring (slot = tile & 1).  As register loads they cost a stall per tile: the
    // values lived in AGPRs across the iteration, the move there needs the data, so the compiler
    // put a vmcnt(0) right behind the load - with the next tile's DMA in flight (in-kernel cycle
    // counts: 17 % of the loop in "issue the next tile").
    int bsel = 0;
    int e_cur = 0, t_cur = 0;
    auto cnt_of = [&](int64_t t) {
      const int64_t r = E - t * TE;
      return (int)(t < t_end ? (r < TE ? r : TE) : 0);
    };
    int* ids_ring = ids_all[wid][0];
    auto ids_issue = [&](int64_t t) {                   // lanes 0-15: edge rows, 16-31: targets
      const int u = lane & 15;
      if (lane < 2 * TE && u < cnt_of(t) && (lane >= TE || eperm)) {
        const int32_t* src = (lane < TE ? eperm : tgt) + t * TE + u;
        lds_dma4(reinterpret_cast<const float*>(src),
                 reinterpret_cast<float*>(ids_ring + (t & 1) * 2 * TE));
      }
    };
    float* base = slab_all[wid][0];
    ids_issue(t_begin);
    ids_issue(t_begin + 1);
    wait_vmem_all();
    tile_issue_ids(qkv, ld, ea, ids_ring + (t_begin & 1) * 2 * TE,
                   eperm ? (int64_t)-1 : t_begin * TE, cnt_of(t_begin), base, lane);

#ifdef SPT_ATTN_PROFILE
    uint64_t prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t tlast = __builtin_amdgcn_s_memtime();
#endif
    for (int64_t t = t_begin; t < t_end; ++t) {
      wait_vmem_all();                                   // tile t landed, indices of t + 1 too
      SPT_PROBE(0)
      if (!a_ready) {                                    // first iteration: the wave's first node
        pop_node();
        a_ready = true;
      } else if (b_state == 1 || b_state == 2) {
        convert_b();                                     // requested before this wait: landed
      }
      SPT_PROBE(1)
      {
        // this tile's ids for the scatters below, then the slot is free for tile t + 2
        const int* mine = ids_ring + (t & 1) * 2 * TE;
        e_cur = eperm ? mine[lane & 15] : (int)(t * TE) + (lane & 15);
        t_cur = mine[TE + (lane & 15)];
      }
      if (t + 1 < t_end)
        tile_issue_ids(qkv, ld, ea, ids_ring + ((t + 1) & 1) * 2 * TE,
                       eperm ? (int64_t)-1 : (t + 1) * TE, cnt_of(t + 1),
                       base + (bsel ^ 1) * SLAB, lane);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // e_cur / t_cur are out of the slot
      ids_issue(t + 2);
      SPT_PROBE(2)
      const float* slab = base + bsel * SLAB;
      const float* kslab = slab + EA_FLOATS;
      const float* vslab = kslab + TE * ROW;
      const int cnt = cnt_of(t);
      const int e0 = (int)(t * TE);

      // ---- recompute GEMM (context independent: q's node term is added per row below) ------
      f32x4 Ck[NB], Cq[NB], Cv[NB];
      {
        bf16x8 Ah, Al;
        load_a_bf(slab, g, c, Ah, Al);
        const __bf16* wh = wb_hi + c * WB_LD + 8 * g;
        const __bf16* wl = wb_lo + c * WB_LD + 8 * g;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          Ck[b] = (f32x4){bk4[b], bk4[b], bk4[b], bk4[b]};
          Cq[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
          Cv[b] = (f32x4){bv4[b], bv4[b], bv4[b], bv4[b]};
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const bf16x8 kh = *reinterpret_cast<const bf16x8*>(wh + (16 * b) * WB_LD);
          const bf16x8 kl = *reinterpret_cast<const bf16x8*>(wl + (16 * b) * WB_LD);
          const bf16x8 qh = *reinterpret_cast<const bf16x8*>(wh + (64 + 16 * b) * WB_LD);
          const bf16x8 ql = *reinterpret_cast<const bf16x8*>(wl + (64 + 16 * b) * WB_LD);
          const bf16x8 vh = *reinterpret_cast<const bf16x8*>(wh + (128 + 16 * b) * WB_LD);
          const bf16x8 vl = *reinterpret_cast<const bf16x8*>(wl + (128 + 16 * b) * WB_LD);
          if constexpr (LO) {
            Ck[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, kh, Ck[b], 0, 0, 0);
            Cq[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, qh, Cq[b], 0, 0, 0);
            Cv[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, vh, Cv[b], 0, 0, 0);
            Ck[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, kl, Ck[b], 0, 0, 0);
            Cq[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, ql, Cq[b], 0, 0, 0);
            Cv[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, vl, Cv[b], 0, 0, 0);
          }
          Ck[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, kh, Ck[b], 0, 0, 0);
          Cq[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, qh, Cq[b], 0, 0, 0);
          Cv[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, vh, Cv[b], 0, 0, 0);
        }
      }
      SPT_PROBE(3)
      // ---- per-edge gradients: rows below `split` belong to A, the rest (up to `hi`) to B;
      //      more than two nodes in the tile -> further passes over the remaining rows ---------
      const float* kp = kslab + 4 * g * ROW + c;
      const float* vp = vslab + 4 * g * ROW + c;
      // gradients replace (k, q, v) in place, row by row as the passes reach them: Ck <- dk,
      // Cq <- dq, Cv <- dv (rows of a later pass keep their k / q / v until then)
      int lo = 0;
      while (lo < cnt) {
        // A owns the row `lo`; B is needed when A ends inside the tile
        int split = A.end - e0;                            // first row after A
        split = split > cnt ? cnt : split;
        int hi = cnt;
        if (split < cnt) {
          if (b_state != 3) convert_b();
          hi = B.node < N ? B.end - e0 : cnt;              // first row after B
          hi = hi > cnt ? cnt : hi;
        }
        if (hi <= lo) hi = cnt;                            // cannot happen: guarantees progress
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 4 * g + r;
            const bool valid = row >= lo && row < hi;
            const bool isB = row >= split;
            const float qsr = isB ? B.qs[b] : A.qs[b];
            const float mr = isB ? B.ml[b] : A.ml[b];
            const float gr = isB ? B.g4[b] : A.g4[b];
            const float dr = isB ? B.delta[b] : A.delta[b];
            const float k = Ck[b][r] + kp[r * ROW + 16 * b];
            const float q = Cq[b][r] + qsr;
            const float v = Cv[b][r] + vp[r * ROW + 16 * b];
            const float p = quad_sum(q * k);
            const float a = valid ? __expf(p - mr) : 0.f;
            const float da = quad_sum(gr * v);
            const float dc = a * (da - dr);
            const float dk = dc * q, dq = dc * k, dv = a * gr;
            Ck[b][r] = valid ? dk : Ck[b][r];
            Cq[b][r] = valid ? dq : Cq[b][r];
            Cv[b][r] = valid ? dv : Cv[b][r];
            A.dqa[b] += isB ? 0.f : dq;                    // a == 0 outside [lo, hi): adds 0
            B.dqa[b] += isB ? dq : 0.f;
          }
        }
        // node bookkeeping (wave-uniform)
        if (A.end - e0 <= hi) {                            // A's last row is inside this pass
          const bool b_started = hi > split;
          finish_node(A);
          pop_node();
          if (b_started && A.end - e0 <= hi) {             // the next node ended in the pass too
            finish_node(A);
            pop_node();
          }
        }
        lo = hi;
      }
      SPT_PROBE(4)
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (4 * g + r >= cnt) {                          // rows beyond the edge list (last tile)
            Ck[b][r] = 0.f;
            Cq[b][r] = 0.f;
            Cv[b][r] = 0.f;
          }
          gb[b] += Ck[b][r];
          gb[NB + b] += Cq[b][r];
          gb[2 * NB + b] += Cv[b][r];
        }
      // dk / dv to the target rows
      int64_t trow[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) trow[r] = (int64_t)__shfl(t_cur, 4 * g + r, 64) * ld;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (4 * g + r < cnt) {
          float* row = gqkv + trow[r] + c;
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            unsafeAtomicAdd(row + 64 + 16 * b, Ck[b][r]);
            unsafeAtomicAdd(row + 128 + 16 * b, Cv[b][r]);
          }
        }
      }
      SPT_PROBE(5)
      // ---- d edge_attr = D W ---------------------------------------------------------------
      // (the rows' edge ids first, all four shuffles in flight under the D-tile round trip:
      // taken one by one next to their stores they cost an LDS round trip each)
      int e4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) e4[r] = __shfl(e_cur, 4 * g + r, 64);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* row = dt + (4 * g + r) * DT_STRIDE + 16 * b + c;
          row[0] = Ck[b][r];
          row[64] = Cq[b][r];
          row[128] = Cv[b][r];
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      asm volatile("" : "+v"(e4[0]), "+v"(e4[1]), "+v"(e4[2]), "+v"(e4[3]));
      f32x4 C2[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
      {
        const float* arow = dt + c * DT_STRIDE + 8 * g;
#pragma unroll
        for (int sg = 0; sg < 6; ++sg) {
          const float4 d0 = *reinterpret_cast<const float4*>(arow + 32 * sg);
          const float4 d1 = *reinterpret_cast<const float4*>(arow + 32 * sg + 4);
          const float dd[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
          bf16x8 dh, dl;
          split_bf16<8>(dd, dh, dl);
#pragma unroll
          for (int fb = 0; fb < 2; ++fb) {
            if constexpr (LO) {
              C2[fb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dl, W2h[sg][fb], C2[fb], 0, 0, 0);
              C2[fb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dh, W2l[sg][fb], C2[fb], 0, 0, 0);
            }
            C2[fb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dh, W2h[sg][fb], C2[fb], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t e = e4[r];
        if (4 * g + r < cnt) {
          if (gea_acc) {            // shared gradient buffer of the stage's blocks
            unsafeAtomicAdd(gea + e * F + c, C2[0][r]);
            unsafeAtomicAdd(gea + e * F + 16 + c, C2[1][r]);
          } else {
            gea[e * F + c] = C2[0][r];
            gea[e * F + 16 + c] = C2[1][r];
          }
        }
      }
SPT_PROBE(6)
            // ---- dW += D^T EA ------------------------------------------------------------------------
      {
        s16x4 Eh[2], El[2];
#pragma unroll
        for (int fb = 0; fb < 2; ++fb) {
          float ev[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int e = 4 * g + r, f = 16 * fb + c;
            ev[r] = slab[e * F + (((f >> 2) ^ (e & 7)) << 2) + (f & 3)];
          }
          bf16x4 eh, el;
          split_bf16<4>(ev, eh, el);
          Eh[fb] = __builtin_bit_cast(s16x4, eh);
          El[fb] = __builtin_bit_cast(s16x4, el);
        }
#pragma unroll
        for (int ob = 0; ob < 3 * NB; ++ob) {
          const f32x4& Dsrc = ob < NB ? Ck[ob % NB] : (ob < 2 * NB ? Cq[ob % NB] : Cv[ob % NB]);
          const float dv4[4] = {Dsrc[0], Dsrc[1], Dsrc[2], Dsrc[3]};
          bf16x4 dh, dl;
          split_bf16<4>(dv4, dh, dl);
          const s16x4 Dh = __builtin_bit_cast(s16x4, dh), Dl = __builtin_bit_cast(s16x4, dl);
#pragma unroll
          for (int fb = 0; fb < 2; ++fb) {
            if constexpr (LO) {
              C3[ob][fb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Dl, Eh[fb], C3[ob][fb], 0, 0, 0);
              C3[ob][fb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Dh, El[fb], C3[ob][fb], 0, 0, 0);
            }
            C3[ob][fb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Dh, Eh[fb], C3[ob][fb], 0, 0, 0);
          }
        }
      }
      SPT_PROBE(7)
      // rotate the pipeline
      bsel ^= 1;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
#ifdef SPT_ATTN_PROFILE
    if (lane == 0 && (wave == 5 || wave == 517))
      printf("attn_bwd_packed wave %ld tiles %ld cycles: wait %lu node %lu issue %lu regemm %lu "
             "pass %lu atomics %lu dea %lu dw %lu\n", (long)wave, (long)(t_end - t_begin),
             (unsigned long)prof[0], (unsigned long)prof[1], (unsigned long)prof[2],
             (unsigned long)prof[3], (unsigned long)prof[4], (unsigned long)prof[5],
             (unsigned long)prof[6], (unsigned long)prof[7]);
#endif
    // nodes left open at the end of the wave's range: their remaining rows belong to the next
    // wave; what this wave accumulated goes out now
    if (a_ready && A.node < N) finish_node(A);
  }
  if (partial) {
    float* pw = partial + (size_t)wave * 192 * (F + 1);
#pragma unroll
    for (int ob = 0; ob < 3 * NB; ++ob) {
#pragma unroll
      for (int fb = 0; fb < 2; ++fb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          pw[(size_t)(16 * ob + 4 * g + r) * (F + 1) + 16 * fb + c] = C3[ob][fb][r];
      const float bsum = xg_sum(gb[ob]);
      if (g == 0) pw[(size_t)(16 * ob + c) * (F + 1) + F] = bsum;
    }
  }
}

}  // namespace mfma

// ---- launchers called from edge_attn.hip's C entry points --------------------
bool attn_xcd_bands();      // edge_attn.hip: SPT_ATTN_XCD_BANDS (default on)
bool attn_mfma_shape_ok(int H, int D, int Dv, int F, const void* ea, const void* Wk,
                        const void* Wq, const void* Wv) {
  return H == 16 && D == 4 && Dv == 4 && F == 32 && ea && Wk && Wq && Wv;
}

void attn_fwd_mfma_launch(const float* qkv, int64_t n, const int32_t* erowptr,
                          const int32_t* eperm, const int32_t* tgt, const float* ea,
                          const float* Wk, const float* bk, const float* Wq, const float* bq,
                          const float* Wv, const float* bv, int scale_mode, float scale_a,
                          float* out, float* m, float* z, int split_bf16, hipStream_t stream) {
  const int64_t blocks = ceil_div(n, mfma::WAVES);
  // 2 048 workgroups (four resident rounds of two per CU) at scene sizes; ONE round (512) up to
  // 131 072 nodes (<= 64 nodes per wave): every workgroup pays the prologue - the weight fragments
  // of three projections, ~100 registers of loads and splits per wave - and a second round that
  // is not full leaves most CUs idle behind it (a train batch's 14 500-node level: 72 -> 51 us
  // with 512 workgroups instead of 2 048, `profiles/r06*_sceneT_captured_kernel_stats.csv`)
  const int64_t want = n <= 131072 ? 512 : 256 * 2 * 4;
  const int grid = (int)(blocks < want ? blocks : want);
  // XCD bands only where the grid is the capped one (a multiple of 8) and a band outlasts a round
  const int bands = attn_xcd_bands() && grid == 256 * 2 * 4 && n >= 8 * (int64_t)grid * mfma::WAVES;
  if (split_bf16 == 3)
    mfma::attn_fwd_mfma_kernel<3><<<grid, mfma::WAVES * 64, 0, stream>>>(
        qkv, 192, n, erowptr, eperm, tgt, ea, Wk, bk, Wq, bq, Wv, bv, scale_mode, scale_a, out, m, z, bands);
  else if (split_bf16 == 1)
    mfma::attn_fwd_mfma_kernel<1><<<grid, mfma::WAVES * 64, 0, stream>>>(
        qkv, 192, n, erowptr, eperm, tgt, ea, Wk, bk, Wq, bq, Wv, bv, scale_mode, scale_a, out, m, z, bands);
  else
    mfma::attn_fwd_mfma_kernel<0><<<grid, mfma::WAVES * 64, 0, stream>>>(
        qkv, 192, n, erowptr, eperm, tgt, ea, Wk, bk, Wq, bq, Wv, bv, scale_mode, scale_a, out, m, z, bands);
}

constexpr int ATTN_BWD_MFMA_BLOCKS = 256;  // one 4-wave workgroup per CU (1 wave / SIMD)

int attn_bwd_mfma_launch(const float* qkv, int64_t n, const int32_t* erowptr,
                         const int32_t* eperm, const int32_t* tgt, const float* ea,
                         const float* Wk, const float* bk, const float* Wq, const float* bq,
                         const float* Wv, const float* bv, int scale_mode, float scale_a,
                         const float* out, const float* m, const float* z, const float* gout,
                         float* gqkv, float* gea, int gea_acc, float* partial, int split_bf16,
                         int64_t e, int packed, hipStream_t stream) {
  const int64_t blocks = ceil_div(n, mfma::WAVES);
  const int grid = (int)(blocks < ATTN_BWD_MFMA_BLOCKS ? blocks : ATTN_BWD_MFMA_BLOCKS);
  if (packed && split_bf16 != 0 && e > 0) {
    // tiles over the edge stream: every wave owns a contiguous range of 16-edge tiles
    const int64_t ntiles = ceil_div(e, (int64_t)mfma::TE);
    const int64_t pb = ceil_div(ntiles, mfma::WAVES);
    const int pgrid = (int)(pb < ATTN_BWD_MFMA_BLOCKS ? pb : ATTN_BWD_MFMA_BLOCKS);
    if (split_bf16 == 3)
      mfma::attn_bwd_packed_kernel<3><<<pgrid, mfma::WAVES * 64, 0, stream>>>(
          qkv, 192, n, e, erowptr, eperm, tgt, ntiles, ea, Wk, bk, Wq, bq, Wv, bv, scale_mode,
          scale_a, out, m, z, gout, gqkv, gea, partial, gea_acc);
    else
      mfma::attn_bwd_packed_kernel<1><<<pgrid, mfma::WAVES * 64, 0, stream>>>(
          qkv, 192, n, e, erowptr, eperm, tgt, ntiles, ea, Wk, bk, Wq, bq, Wv, bv, scale_mode,
          scale_a, out, m, z, gout, gqkv, gea, partial, gea_acc);
    return pgrid * mfma::WAVES;
  }
  if (split_bf16 == 3)
    mfma::attn_bwd_mfma_kernel<3><<<grid, mfma::WAVES * 64, 0, stream>>>(
        qkv, 192, n, erowptr, eperm, tgt, ea, Wk, bk, Wq, bq, Wv, bv, scale_mode, scale_a, out, m,
        z, gout, gqkv, gea, partial, gea_acc);
  else if (split_bf16 == 1)
    mfma::attn_bwd_mfma_kernel<1><<<grid, mfma::WAVES * 64, 0, stream>>>(
        qkv, 192, n, erowptr, eperm, tgt, ea, Wk, bk, Wq, bq, Wv, bv, scale_mode, scale_a, out, m,
        z, gout, gqkv, gea, partial, gea_acc);
  else
    mfma::attn_bwd_mfma_kernel<0><<<grid, mfma::WAVES * 64, 0, stream>>>(
        qkv, 192, n, erowptr, eperm, tgt, ea, Wk, bk, Wq, bq, Wv, bv, scale_mode, scale_a, out, m,
        z, gout, gqkv, gea, partial, gea_acc);
  return grid * mfma::WAVES;  // number of partial tables written
}

}  // namespace spt
