// Fused sparse superpoint-graph self-attention with relative-pose encodings
// (src/nn/attention.py:202-315): per source node s, over its outgoing edges e
//
//   q_e = q[s]*scale(s) + Wq ea_e + bq     k_e = k[t_e] + Wk ea_e + bk
//   v_e = v[t_e] + Wv ea_e + bv            c_eh = <q_e[h,:], k_e[h,:]>
//   a_eh = softmax_{e in out(s)}(c_eh)     out[s,h,:] = sum_e a_eh v_e[h,:]
//
// The reference materialises ~10 [E, C] f32 temporaries over ~25 launches
// (gathers, 3 RPE Linears, einsum, PyG softmax = scatter-max/exp/scatter-sum,
// scatter-sum with float atomics).  Here one wave owns one source node: its
// lanes are the (head, dim) slots of q/k/v, the three RPE weight blocks live
// in that lane's registers for the whole kernel, edge_attr tiles are staged
// through LDS and broadcast, the softmax is online (running max / sum per
// head) and nothing per-edge is ever written to HBM.  Edges are visited
// through the CSR-by-source view (erowptr / eperm), so the result is
// deterministic (no atomics in the forward).
#include <math.h>
#include <stdlib.h>

#include "common.hpp"

namespace spt {

constexpr int EA_TE = 8;           // edges per LDS tile
constexpr int EA_WAVES = 4;        // waves per workgroup

struct AttnShape {
  int H, D, Dv, QK, C;             // heads, qk dim, value dim, H*D, H*Dv
  int lph_log2, lphv_log2;         // lanes per head (qk / v)
};

// lane-owned weight rows --------------------------------------------------------
template <int PL, int F>
__device__ __forceinline__ void load_rows(const float* __restrict__ W,
                                          const float* __restrict__ b, int first,
                                          int total, float (&w)[PL][F], float (&bb)[PL]) {
#pragma unroll
  for (int i = 0; i < PL; ++i) {
    bb[i] = 0.f;
#pragma unroll
    for (int f = 0; f < F; ++f) w[i][f] = 0.f;
  }
  if (W == nullptr) return;  // wave-uniform
#pragma unroll
  for (int i = 0; i < PL; ++i) {
    const int r = first + i;
    const int rc = r < total ? r : total - 1;  // clamped: loads stay in bounds, no branches
    const float keep = r < total ? 1.f : 0.f;
    if (b) bb[i] = b[rc] * keep;
#pragma unroll
    for (int f = 0; f < F; ++f) w[i][f] = W[(size_t)rc * F + f] * keep;
  }
}

__device__ __forceinline__ void lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// stage the edge_attr rows of one tile into this wave's LDS slab
template <int F>
__device__ __forceinline__ void stage_tile(const float* __restrict__ ea,
                                           int e_lane /* edge id held by lane u<TE */,
                                           int cnt, float* slab) {
  const int lane = threadIdx.x & 63;
  if constexpr (F % 4 == 0) {
    constexpr int CH = F / 4;                 // float4 chunks per row
    constexpr int TOT = EA_TE * CH;
#pragma unroll
    for (int p = 0; p < (TOT + 63) / 64; ++p) {
      const int q = p * 64 + lane;
      const int u = q / CH, ch = q - u * CH;
      const int e = __shfl(e_lane, u < EA_TE ? u : 0, 64);
      if (q < TOT && u < cnt) {
        const float4 t = *reinterpret_cast<const float4*>(ea + (size_t)e * F + ch * 4);
        *reinterpret_cast<float4*>(slab + u * F + ch * 4) = t;
      }
    }
  } else {
    constexpr int TOT = EA_TE * F;
#pragma unroll
    for (int p = 0; p < (TOT + 63) / 64; ++p) {
      const int q = p * 64 + lane;
      const int u = q / F, f = q - u * F;
      const int e = __shfl(e_lane, u < EA_TE ? u : 0, 64);
      if (q < TOT && u < cnt) slab[u * F + f] = ea[(size_t)e * F + f];
    }
  }
}

template <int PL, int F>
__device__ __forceinline__ void rpe(const float (&w)[PL][F], const float (&b)[PL],
                                    const float* row, float (&o)[PL]) {
#pragma unroll
  for (int i = 0; i < PL; ++i) o[i] = b[i];
  if constexpr (F % 4 == 0) {
#pragma unroll
    for (int f = 0; f < F; f += 4) {
      const float4 a = *reinterpret_cast<const float4*>(row + f);  // LDS broadcast
#pragma unroll
      for (int i = 0; i < PL; ++i) {
        o[i] = fmaf(w[i][f + 0], a.x, o[i]);
        o[i] = fmaf(w[i][f + 1], a.y, o[i]);
        o[i] = fmaf(w[i][f + 2], a.z, o[i]);
        o[i] = fmaf(w[i][f + 3], a.w, o[i]);
      }
    }
  } else {
#pragma unroll
    for (int f = 0; f < F; ++f) {
      const float a = row[f];
#pragma unroll
      for (int i = 0; i < PL; ++i) o[i] = fmaf(w[i][f], a, o[i]);
    }
  }
}

__device__ __forceinline__ float qk_scale_of(int mode, float a, int deg) {
  // src/utils/nn.py:83-127: D = (dim // num_heads)^-0.5, G = deg(s)^-0.5
  const float g = __builtin_amdgcn_rsqf((float)deg);   // v_rsq_f32 (1 ulp; the same in every attention kernel)
  if (mode == 0) return a * g;   // 'd.g' (default), 'g' with a = 1
  if (mode == 1) return a + g;   // 'd+g'
  return a;                      // 'd' or a user constant
}

// waves per SIMD the register-resident weight block leaves room for
#ifndef SPT_EA_FWD_MAXW
#define SPT_EA_FWD_MAXW 3
#endif
constexpr int ea_min_waves(int qpl, int vpl, int f) {
  const int w = (2 * qpl + vpl) * f;
  const int r = w <= 96 ? 3 : (w <= 136 ? 2 : 1);
  return r < SPT_EA_FWD_MAXW ? r : SPT_EA_FWD_MAXW;
}

// ---- asynchronous tile pipeline (global -> LDS without a VGPR round trip) -------
// A wave walks the flat sequence of 8-edge tiles of its nodes.  While it
// computes tile i out of LDS buffer b, the edge_attr rows and the gathered k / v
// rows of tile i+1 stream into buffer b^1 with `global_load_lds` (the data never
// touches VGPRs, so the prefetch costs no registers), and the edge / target
// indices of tile i+2 are already in flight.
struct TileDesc {
  int64_t s;        // source node
  int start, end;   // CSR range of the node
  int t0;           // first edge of this tile
  bool valid;
};

__device__ __forceinline__ TileDesc tile_of_node(int64_t s, int64_t N,
                                                 const int32_t* __restrict__ erowptr) {
  TileDesc d;
  d.s = s;
  d.valid = s < N;
  d.start = d.valid ? erowptr[s] : 0;
  d.end = d.valid ? erowptr[s + 1] : 0;
  d.t0 = d.start;
  return d;
}

__device__ __forceinline__ TileDesc tile_advance(const TileDesc& c, int64_t N, int64_t nwaves,
                                                 const int32_t* __restrict__ erowptr) {
  if (c.valid && c.t0 + EA_TE < c.end) {
    TileDesc d = c;
    d.t0 += EA_TE;
    return d;
  }
  return tile_of_node(c.s + nwaves, N, erowptr);
}

__device__ __forceinline__ int tile_count(const TileDesc& d) {
  const int r = d.end - d.t0;
  return r < EA_TE ? (r > 0 ? r : 0) : EA_TE;
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

#define SPT_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define SPT_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// issue the asynchronous loads of one tile into `buf` (layout: ea | k rows | v rows)
template <int QPL, int VPL, int F>
__device__ __forceinline__ void tile_issue(const float* __restrict__ qkv, int ld,
                                           const AttnShape& sh, const float* __restrict__ ea,
                                           bool has_rpe, int e_lane, int t_lane, int cnt,
                                           float* buf, int lane, int j0, int c0, bool qv,
                                           bool vv) {
  constexpr int KROW = 64 * QPL, VROW = 64 * VPL;
  float* kbuf = buf + EA_TE * F;
  float* vbuf = kbuf + EA_TE * KROW;
  if (has_rpe) {
    if constexpr (F % 4 == 0) {
      constexpr int CH = F / 4, TOT = EA_TE * CH;
#pragma unroll
      for (int p = 0; p < (TOT + 63) / 64; ++p) {
        const int q = p * 64 + lane;
        const int u = q / CH, ch = q - u * CH;
        const int e = __shfl(e_lane, u < EA_TE ? u : 0, 64);
        if (q < TOT && u < cnt)
          lds_dma16(ea + (size_t)e * F + ch * 4, buf + p * 256);
      }
    } else {
      constexpr int TOT = EA_TE * F;
#pragma unroll
      for (int p = 0; p < (TOT + 63) / 64; ++p) {
        const int q = p * 64 + lane;
        const int u = q / F, f = q - u * F;
        const int e = __shfl(e_lane, u < EA_TE ? u : 0, 64);
        if (q < TOT && u < cnt)
          lds_dma4(ea + (size_t)e * F + f, buf + p * 64);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < EA_TE; ++u) {
    const int64_t t = __shfl(t_lane, u, 64);
    if (u < cnt) {
#pragma unroll
      for (int i = 0; i < QPL; ++i)
        if (qv)
          lds_dma4(qkv + t * ld + sh.QK + j0 + i, kbuf + u * KROW + i * 64);
#pragma unroll
      for (int i = 0; i < VPL; ++i)
        if (vv)
          lds_dma4(qkv + t * ld + 2 * sh.QK + c0 + i, vbuf + u * VROW + i * 64);
    }
  }
}

template <int QPL, int VPL, int F>
__global__ __launch_bounds__(EA_WAVES * 64, ea_min_waves(QPL, VPL, F)) void edge_attn_fwd_kernel(
    const float* __restrict__ qkv, int ld, int64_t N, AttnShape sh,
    const int32_t* __restrict__ erowptr, const int32_t* __restrict__ eperm,
    const int32_t* __restrict__ tgt, const float* __restrict__ ea,
    const float* __restrict__ Wk, const float* __restrict__ bk,
    const float* __restrict__ Wq, const float* __restrict__ bq,
    const float* __restrict__ Wv, const float* __restrict__ bv, int scale_mode,
    float scale_a, float* __restrict__ out, float* __restrict__ mbuf,
    float* __restrict__ zbuf) {
  // two per-wave LDS buffers: edge_attr tile | gathered k rows | gathered v rows
  constexpr int KROW = 64 * QPL, VROW = 64 * VPL;
  constexpr int SLAB = EA_TE * (F + KROW + VROW);
  __shared__ __attribute__((aligned(16))) float slab_all[EA_WAVES][2][SLAB];
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;

  const int j0 = lane * QPL, c0 = lane * VPL;
  const bool qv = j0 < sh.QK, vv = c0 < sh.C;
  const bool has_rpe = ea != nullptr && (Wk || Wq || Wv);
  float wk[QPL][F], wq[QPL][F], wv[VPL][F], bbk[QPL], bbq[QPL], bbv[VPL];
  load_rows<QPL, F>(Wk, bk, j0, sh.QK, wk, bbk);
  load_rows<QPL, F>(Wq, bq, j0, sh.QK, wq, bbq);
  load_rows<VPL, F>(Wv, bv, c0, sh.C, wv, bbv);

  const int lph = 1 << sh.lph_log2;
  const bool aligned = sh.lph_log2 == sh.lphv_log2;
  const int hv = vv ? c0 / sh.Dv : 0;
  const int src_lane = (hv * sh.D) / QPL;   // qk lane holding my value head's softmax state
  const int my_head = qv ? j0 / sh.D : 0;
  const bool head_leader = qv && ((lane & (lph - 1)) == 0);

  const int64_t wave = (int64_t)blockIdx.x * EA_WAVES + wid;
  const int64_t nwaves = (int64_t)gridDim.x * EA_WAVES;

  auto load_idx = [&](const TileDesc& d, int& e_lane, int& t_lane) {
    e_lane = 0;
    t_lane = 0;
    const int cnt = tile_count(d);
    if (d.valid && lane < cnt) {
      e_lane = eperm ? eperm[d.t0 + lane] : d.t0 + lane;
      t_lane = tgt[d.t0 + lane];
    }
  };

  for (int i = lane; i < 2 * SLAB; i += 64) (&slab_all[wid][0][0])[i] = 0.f;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  TileDesc cur = tile_of_node(wave, N, erowptr);
  if (!cur.valid) return;
  int e_cur, t_cur, e_nxt, t_nxt;
  load_idx(cur, e_cur, t_cur);
  int b = 0;
  tile_issue<QPL, VPL, F>(qkv, ld, sh, ea, has_rpe, e_cur, t_cur, tile_count(cur),
                          slab_all[wid][b], lane, j0, c0, qv, vv);
  TileDesc nxt = tile_advance(cur, N, nwaves, erowptr);
  load_idx(nxt, e_nxt, t_nxt);

  float qs[QPL], acc[VPL];
  float m = -INFINITY, z = 0.f;
  while (cur.valid) {
    wait_vmem_all();  // tile `cur` has landed in buffer b; the indices of `nxt` are here
    if (nxt.valid)
      tile_issue<QPL, VPL, F>(qkv, ld, sh, ea, has_rpe, e_nxt, t_nxt, tile_count(nxt),
                              slab_all[wid][b ^ 1], lane, j0, c0, qv, vv);
    const float* slab = slab_all[wid][b];
    const float* kslab = slab + EA_TE * F;
    const float* vslab = kslab + EA_TE * KROW;
    const int cnt = tile_count(cur);
    if (cur.t0 == cur.start) {  // first tile of the node: reset the state
      const int deg = cur.end - cur.start;
      const float scale = deg > 0 ? qk_scale_of(scale_mode, scale_a, deg) : 0.f;
#pragma unroll
      for (int i = 0; i < QPL; ++i) qs[i] = qv ? qkv[cur.s * ld + j0 + i] * scale : 0.f;
#pragma unroll
      for (int i = 0; i < VPL; ++i) acc[i] = 0.f;
      m = -INFINITY;
      z = 0.f;
    }
#pragma unroll 1
    for (int u = 0; u < cnt; ++u) {
      float ke[QPL], qe[QPL], ve[VPL];
      if (has_rpe) {
        rpe<QPL, F>(wk, bbk, slab + u * F, ke);
        rpe<QPL, F>(wq, bbq, slab + u * F, qe);
        rpe<VPL, F>(wv, bbv, slab + u * F, ve);
      } else {
#pragma unroll
        for (int i = 0; i < QPL; ++i) ke[i] = qe[i] = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) ve[i] = 0.f;
      }
      float p = 0.f;
#pragma unroll
      for (int i = 0; i < QPL; ++i) {
        ke[i] += kslab[u * KROW + i * 64 + lane];
        qe[i] += qs[i];
        p = fmaf(qe[i], ke[i], p);
      }
      for (int o = 1; o < lph; o <<= 1) p += __shfl_xor(p, o, 64);
      // online softmax per head (state replicated on the head's lanes);
      // v_exp_f32 path: relative error ~1e-6 at |x| <= 20, far inside the 1e-4 bar
      const float mn = fmaxf(m, p);
      const float corr = __expf(m - mn);
      const float pe = __expf(p - mn);
      z = fmaf(z, corr, pe);
      m = mn;
      float corr_v = corr, pe_v = pe;
      if (!aligned) {
        corr_v = __shfl(corr, src_lane, 64);
        pe_v = __shfl(pe, src_lane, 64);
      }
#pragma unroll
      for (int i = 0; i < VPL; ++i)
        acc[i] = fmaf(acc[i], corr_v, pe_v * (ve[i] + vslab[u * VROW + i * 64 + lane]));
    }
    if (cur.t0 + EA_TE >= cur.end) {  // last tile of the node: write its row
      float zz = z;
      if (!aligned) zz = __shfl(z, src_lane, 64);
      if (vv) {
#pragma unroll
        for (int i = 0; i < VPL; ++i) out[cur.s * sh.C + c0 + i] = acc[i] / (zz + 1e-16f);
      }
      if (head_leader && mbuf) {
        mbuf[cur.s * sh.H + my_head] = m;
        zbuf[cur.s * sh.H + my_head] = z;
      }
    }
    // rotate: the indices of the tile after `nxt` are requested now and land
    // while `nxt` is being computed
    cur = nxt;
    e_cur = e_nxt;
    t_cur = t_nxt;
    nxt = tile_advance(cur, N, nwaves, erowptr);
    load_idx(nxt, e_nxt, t_nxt);
    b ^= 1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  }
}

// ---- backward ------------------------------------------------------------------
// One pass over the same CSR view.  Per edge the forward quantities are
// recomputed from (m, z) saved per (node, head); then
//   dv_e = a g_s            da = <g_s, v_e>_head      dc = a (da - delta_s)
//   dq_e = dc k_e           dk_e = dc q_e             delta_s = <g_s, out_s>_head
// dq accumulates in registers (one writer per node); dk / dv go to the target
// row with hardware f32 atomics (the only non-deterministic sums of the
// library, like the reference's scatter backward); d edge_attr is a wave
// reduce-scatter of the lane partials; the RPE weight gradients accumulate in
// the owner lane's registers for the whole kernel and leave as per-wave partial
// tables that a fixed-order kernel reduces.
// Reduce-scatter of N per-lane partials across the 64 lanes of a wave: each
// step halves the live values (compile-time indices only: a runtime index into
// a register array would turn into a 32-way select chain).
template <int HALF, int O, int N>
__device__ __forceinline__ void rs_step(float (&p)[N], int lane) {
  const bool upper = (lane & O) != 0;
#pragma unroll
  for (int i = 0; i < HALF; ++i) {
    const float send = upper ? p[i] : p[i + HALF];
    const float keep = upper ? p[i + HALF] : p[i];
    p[i] = keep + __shfl_xor(send, O, 64);
  }
  if constexpr (HALF > 1) rs_step<HALF / 2, O / 2, N>(p, lane);
}

template <int N>
__device__ __forceinline__ void wave_reduce_scatter(float (&p)[N], int lane) {
  // N = 32: after the call p[0] of lane l holds the sum over the 32 lanes that
  // share bit 0 with l of column  b5*16 + b4*8 + b3*4 + b2*2 + b1  (bits of l).
  rs_step<N / 2, 32, N>(p, lane);
}

template <int QPL, int VPL, int F>
__global__ __launch_bounds__(EA_WAVES * 64, 1) void edge_attn_bwd_kernel(
    const float* __restrict__ qkv, int ld, int64_t N, AttnShape sh,
    const int32_t* __restrict__ erowptr, const int32_t* __restrict__ eperm,
    const int32_t* __restrict__ tgt, const float* __restrict__ ea,
    const float* __restrict__ Wk, const float* __restrict__ bk,
    const float* __restrict__ Wq, const float* __restrict__ bq,
    const float* __restrict__ Wv, const float* __restrict__ bv, int scale_mode,
    float scale_a, const float* __restrict__ out, const float* __restrict__ mbuf,
    const float* __restrict__ zbuf, const float* __restrict__ gout,
    float* __restrict__ gqkv, float* __restrict__ gea, float* __restrict__ partial,
    int gea_acc) {
  constexpr int KROW = 64 * QPL, VROW = 64 * VPL;
  constexpr int SLAB = EA_TE * (F + KROW + VROW);
  __shared__ __attribute__((aligned(16))) float slab_all[EA_WAVES][2][SLAB];
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;

  const int j0 = lane * QPL, c0 = lane * VPL;
  const bool qv = j0 < sh.QK, vv = c0 < sh.C;
  const bool has_rpe = ea != nullptr && (Wk || Wq || Wv);
  float wk[QPL][F], wq[QPL][F], wv[VPL][F], bbk[QPL], bbq[QPL], bbv[VPL];
  load_rows<QPL, F>(Wk, bk, j0, sh.QK, wk, bbk);
  load_rows<QPL, F>(Wq, bq, j0, sh.QK, wq, bbq);
  load_rows<VPL, F>(Wv, bv, c0, sh.C, wv, bbv);
  float gwk[QPL][F], gwq[QPL][F], gwv[VPL][F], gbk[QPL], gbq[QPL], gbv[VPL];
#pragma unroll
  for (int i = 0; i < QPL; ++i) {
    gbk[i] = gbq[i] = 0.f;
#pragma unroll
    for (int f = 0; f < F; ++f) gwk[i][f] = gwq[i][f] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    gbv[i] = 0.f;
#pragma unroll
    for (int f = 0; f < F; ++f) gwv[i][f] = 0.f;
  }

  const int lph = 1 << sh.lph_log2, lphv = 1 << sh.lphv_log2;
  const bool aligned = sh.lph_log2 == sh.lphv_log2;
  const int hv = vv ? c0 / sh.Dv : 0;
  const int src_lane = (hv * sh.D) / QPL;            // qk lane holding my value head's state
  const int my_head = qv ? j0 / sh.D : 0;
  const int vsrc_lane = (my_head * sh.Dv) / VPL;     // v lane holding my qk head's value sums

  const int64_t wave = (int64_t)blockIdx.x * EA_WAVES + wid;
  const int64_t nwaves = (int64_t)gridDim.x * EA_WAVES;
  auto load_idx = [&](const TileDesc& d, int& e_lane, int& t_lane) {
    e_lane = 0;
    t_lane = 0;
    const int cnt = tile_count(d);
    if (d.valid && lane < cnt) {
      e_lane = eperm ? eperm[d.t0 + lane] : d.t0 + lane;
      t_lane = tgt[d.t0 + lane];
    }
  };
  TileDesc cur = tile_of_node(wave, N, erowptr);
  int e_lane = 0, t_lane = 0, e_nxt = 0, t_nxt = 0;
  int bsel = 0;
  TileDesc nxt = cur;
  if (cur.valid) {
    load_idx(cur, e_lane, t_lane);
    tile_issue<QPL, VPL, F>(qkv, ld, sh, ea, has_rpe, e_lane, t_lane, tile_count(cur),
                            slab_all[wid][bsel], lane, j0, c0, qv, vv);
    nxt = tile_advance(cur, N, nwaves, erowptr);
    load_idx(nxt, e_nxt, t_nxt);
  }
  float qs[QPL], dqa[QPL], g[VPL];
  float scale = 0.f, delta = 0.f, m = 0.f, zi = 0.f;
  while (cur.valid) {
    wait_vmem_all();
    if (nxt.valid)
      tile_issue<QPL, VPL, F>(qkv, ld, sh, ea, has_rpe, e_nxt, t_nxt, tile_count(nxt),
                              slab_all[wid][bsel ^ 1], lane, j0, c0, qv, vv);
    const float* slab = slab_all[wid][bsel];
    const float* kslab = slab + EA_TE * F;
    const float* vslab = kslab + EA_TE * KROW;
    const int64_t s = cur.s;
    const int cnt = tile_count(cur);
    if (cur.t0 == cur.start) {  // first tile of the node
      const int deg = cur.end - cur.start;
      scale = deg > 0 ? qk_scale_of(scale_mode, scale_a, deg) : 0.f;
#pragma unroll
      for (int i = 0; i < QPL; ++i) {
        qs[i] = qv ? qkv[s * ld + j0 + i] * scale : 0.f;
        dqa[i] = 0.f;
      }
      float dl = 0.f;
#pragma unroll
      for (int i = 0; i < VPL; ++i) {
        g[i] = vv ? gout[s * sh.C + c0 + i] : 0.f;
        dl = fmaf(g[i], vv ? out[s * sh.C + c0 + i] : 0.f, dl);
      }
      for (int o = 1; o < lphv; o <<= 1) dl += __shfl_xor(dl, o, 64);
      delta = aligned ? dl : __shfl(dl, vsrc_lane, 64);
      m = qv ? mbuf[s * sh.H + my_head] : 0.f;
      zi = qv ? 1.0f / (zbuf[s * sh.H + my_head] + 1e-16f) : 0.f;
    }
    {
#pragma unroll 1
      for (int u = 0; u < cnt; ++u) {
        const float* row = slab + u * F;
        float ke[QPL], qe[QPL], ve[VPL];
        if (has_rpe) {
          rpe<QPL, F>(wk, bbk, row, ke);
          rpe<QPL, F>(wq, bbq, row, qe);
          rpe<VPL, F>(wv, bbv, row, ve);
        } else {
#pragma unroll
          for (int i = 0; i < QPL; ++i) ke[i] = qe[i] = 0.f;
#pragma unroll
          for (int i = 0; i < VPL; ++i) ve[i] = 0.f;
        }
        float p = 0.f;
#pragma unroll
        for (int i = 0; i < QPL; ++i) {
          ke[i] += kslab[u * KROW + i * 64 + lane];
          qe[i] += qs[i];
          p = fmaf(qe[i], ke[i], p);
        }
        for (int o = 1; o < lph; o <<= 1) p += __shfl_xor(p, o, 64);
        const float a = __expf(p - m) * zi;                   // attention weight (qk lanes)
        const float a_v = aligned ? a : __shfl(a, src_lane, 64);
        float dv[VPL], da = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
          ve[i] += vslab[u * VROW + i * 64 + lane];
          dv[i] = a_v * g[i];
          da = fmaf(g[i], ve[i], da);
        }
        for (int o = 1; o < lphv; o <<= 1) da += __shfl_xor(da, o, 64);
        const float da_q = aligned ? da : __shfl(da, vsrc_lane, 64);
        const float dc = a * (da_q - delta);
        float dq[QPL], dk[QPL];
#pragma unroll
        for (int i = 0; i < QPL; ++i) {
          dq[i] = qv ? dc * ke[i] : 0.f;
          dk[i] = qv ? dc * qe[i] : 0.f;
          dqa[i] += dq[i];
        }
        // scatter to the target row
        const int64_t t = __shfl(t_lane, u, 64);
        if (qv) {
#pragma unroll
          for (int i = 0; i < QPL; ++i) unsafeAtomicAdd(gqkv + t * ld + sh.QK + j0 + i, dk[i]);
        }
        if (vv) {
#pragma unroll
          for (int i = 0; i < VPL; ++i) unsafeAtomicAdd(gqkv + t * ld + 2 * sh.QK + c0 + i, dv[i]);
        }
        if (has_rpe) {
          // weight / bias gradients: owner-lane register accumulators
#pragma unroll
          for (int i = 0; i < QPL; ++i) {
            gbk[i] += dk[i];
            gbq[i] += dq[i];
          }
#pragma unroll
          for (int i = 0; i < VPL; ++i) gbv[i] += dv[i];
          float pf[F];
          if constexpr (F % 4 == 0) {
#pragma unroll
            for (int f = 0; f < F; f += 4) {
              const float4 av = *reinterpret_cast<const float4*>(row + f);
              const float a4[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < QPL; ++i) {
                  gwk[i][f + r] = fmaf(dk[i], a4[r], gwk[i][f + r]);
                  gwq[i][f + r] = fmaf(dq[i], a4[r], gwq[i][f + r]);
                  acc = fmaf(wk[i][f + r], dk[i], acc);
                  acc = fmaf(wq[i][f + r], dq[i], acc);
                }
#pragma unroll
                for (int i = 0; i < VPL; ++i) {
                  gwv[i][f + r] = fmaf(dv[i], a4[r], gwv[i][f + r]);
                  acc = fmaf(wv[i][f + r], dv[i], acc);
                }
                pf[f + r] = acc;
              }
            }
          } else {
#pragma unroll
            for (int f = 0; f < F; ++f) {
              const float av = row[f];
              float acc = 0.f;
#pragma unroll
              for (int i = 0; i < QPL; ++i) {
                gwk[i][f] = fmaf(dk[i], av, gwk[i][f]);
                gwq[i][f] = fmaf(dq[i], av, gwq[i][f]);
                acc = fmaf(wk[i][f], dk[i], acc);
                acc = fmaf(wq[i][f], dq[i], acc);
              }
#pragma unroll
              for (int i = 0; i < VPL; ++i) {
                gwv[i][f] = fmaf(dv[i], av, gwv[i][f]);
                acc = fmaf(wv[i][f], dv[i], acc);
              }
              pf[f] = acc;
            }
          }
          // d edge_attr[e, f] = sum over lanes of pf[f]
          const int64_t e = __shfl(e_lane, u, 64);
          if constexpr (F == 32) {
            wave_reduce_scatter<32>(pf, lane);
            // column owned by this lane pair: bits 5..1 of the lane, MSB first
            const int col = ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 +
                            ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
            const float tot = pf[0] + __shfl_xor(pf[0], 1, 64);
            if ((lane & 1) == 0) {
              if (gea_acc) unsafeAtomicAdd(gea + e * F + col, tot);
              else gea[e * F + col] = tot;
            }
          } else {
#pragma unroll
            for (int f = 0; f < F; ++f) {
              const float tot = wave_reduce_sum(pf[f]);
              if (lane == 0) {
                if (gea_acc) unsafeAtomicAdd(gea + e * F + f, tot);
                else gea[e * F + f] = tot;
              }
            }
          }
        }
      }
    }
    if (cur.t0 + EA_TE >= cur.end && qv) {  // last tile of the node
#pragma unroll
      for (int i = 0; i < QPL; ++i) gqkv[s * ld + j0 + i] = dqa[i] * scale;
    }
    cur = nxt;
    e_lane = e_nxt;
    t_lane = t_nxt;
    nxt = tile_advance(cur, N, nwaves, erowptr);
    load_idx(nxt, e_nxt, t_nxt);
    bsel ^= 1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  }
  // per-wave partial tables [rows = 2*QK + C][F + 1]
  if (has_rpe && partial) {
    float* pw = partial + (size_t)wave * (2 * sh.QK + sh.C) * (F + 1);
    if (qv) {
#pragma unroll
      for (int i = 0; i < QPL; ++i) {
        float* rk = pw + (size_t)(j0 + i) * (F + 1);
        float* rq = pw + (size_t)(sh.QK + j0 + i) * (F + 1);
#pragma unroll
        for (int f = 0; f < F; ++f) {
          rk[f] = gwk[i][f];
          rq[f] = gwq[i][f];
        }
        rk[F] = gbk[i];
        rq[F] = gbq[i];
      }
    }
    if (vv) {
#pragma unroll
      for (int i = 0; i < VPL; ++i) {
        float* rv = pw + (size_t)(2 * sh.QK + c0 + i) * (F + 1);
#pragma unroll
        for (int f = 0; f < F; ++f) rv[f] = gwv[i][f];
        rv[F] = gbv[i];
      }
    }
  }
}

// fixed-order sum of the per-wave partial tables: 16 columns x 16 slices per block
// Fixed-order sum of the per-wave partial tables [rows][F + 1] and the split of the result into
// the six gradient tensors, in ONE launch (round 6: the sum and the split were two kernels, 14
// launches of ~5 us per train-batch step).  Same slices, same order of adds as before.
__global__ __launch_bounds__(256) void attn_reduce_partials_kernel(
    const float* __restrict__ partial, int nwaves, int len, int QK, int C, int F,
    float* __restrict__ gWk, float* __restrict__ gbk, float* __restrict__ gWq,
    float* __restrict__ gbq, float* __restrict__ gWv, float* __restrict__ gbv) {
  __shared__ float sl[16][17];
  const int cl = threadIdx.x & 15;
  const int col = blockIdx.x * 16 + cl;
  const int slice = threadIdx.x >> 4;
  float acc = 0.f;
  if (col < len) {
    const int per = (nwaves + 15) / 16;
    const int lo = slice * per, hi = (lo + per < nwaves) ? lo + per : nwaves;
    // sixteen tables in flight, added in order (1 024 pair tables = 64 per slice: one at a time
    // was 64 dependent round trips, 20-28 us per call)
    int k = lo;
    for (; k + 16 <= hi; k += 16) {
      float a[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = partial[(size_t)(k + j) * len + col];
#pragma unroll
      for (int j = 0; j < 16; ++j) acc += a[j];
    }
    for (; k < hi; ++k) acc += partial[(size_t)k * len + col];
  }
  sl[slice][cl] = acc;
  __syncthreads();
  if (slice == 0 && col < len) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sl[k][cl];
    const int r = col / (F + 1), f = col - r * (F + 1);
    float* W;
    float* b;
    int rr;
    if (r < QK) { W = gWk; b = gbk; rr = r; }
    else if (r < 2 * QK) { W = gWq; b = gbq; rr = r - QK; }
    else { W = gWv; b = gbv; rr = r - 2 * QK; }
    if (f < F) { if (W) W[(size_t)rr * F + f] = t; }
    else if (b) b[rr] = t;
  }
}

static bool attn_shape(int H, int D, int Dv, int qpl, int vpl, AttnShape* sh) {
  sh->H = H; sh->D = D; sh->Dv = Dv; sh->QK = H * D; sh->C = H * Dv;
  if (sh->QK > 64 * qpl || sh->C > 64 * vpl) return false;
  if (D % qpl || Dv % vpl) return false;
  const int lph = D / qpl, lphv = Dv / vpl;
  if ((lph & (lph - 1)) || (lphv & (lphv - 1))) return false;
  sh->lph_log2 = 0; while ((1 << sh->lph_log2) < lph) ++sh->lph_log2;
  sh->lphv_log2 = 0; while ((1 << sh->lphv_log2) < lphv) ++sh->lphv_log2;
  return true;
}

static inline int per_lane(int total) { return total <= 64 ? 1 : (total <= 128 ? 2 : 0); }

}  // namespace spt

namespace spt {
// edge_attn_mfma.hip: matrix-pipe formulation for the SPT-64 head layout
bool attn_mfma_shape_ok(int H, int D, int Dv, int F, const void* ea, const void* Wk,
                        const void* Wq, const void* Wv);
void attn_fwd_mfma_launch(const float* qkv, int64_t n, const int32_t* erowptr,
                          const int32_t* eperm, const int32_t* tgt, const float* ea,
                          const float* Wk, const float* bk, const float* Wq, const float* bq,
                          const float* Wv, const float* bv, int scale_mode, float scale_a,
                          float* out, float* m, float* z, int split_bf16, hipStream_t stream);
int attn_bwd_mfma_launch(const float* qkv, int64_t n, const int32_t* erowptr,
                         const int32_t* eperm, const int32_t* tgt, const float* ea,
                         const float* Wk, const float* bk, const float* Wq, const float* bq,
                         const float* Wv, const float* bv, int scale_mode, float scale_a,
                         const float* out, const float* m, const float* z, const float* gout,
                         float* gqkv, float* gea, int gea_acc, float* partial, int split_bf16,
                         int64_t e, int packed, hipStream_t stream);
// edge_attn_el.hip: edge-lane backward (two waves per SIMD)
size_t attn_bwd_el_workspace_bytes(int64_t n, int64_t e);
void attn_pack_tile_ids_launch(const int32_t* eperm, const int32_t* tgt, const int32_t* src,
                               int64_t e, int32_t* ids3, hipStream_t stream);
// edge_attn_to.hip: the same backward over the edge stream in TARGET order
bool attn_bwd_to_enabled();
size_t attn_bwd_to_workspace_bytes(int64_t n, int64_t e);
void attn_pack_tile_ids_to_launch(const int32_t* eperm, const int32_t* tgt, const int32_t* src,
                                  const int32_t* tperm, int64_t e, int32_t* ids4, hipStream_t stream);
// XCD bands of the attention kernels' work distribution (edge_attn_mfma.hip forward, edge_attn_to.hip
// backward): a SPEED choice between two walks of the same work - read once from the environment
// (SPT_ATTN_XCD_BANDS=0 restores the plain grid stride / contiguous tile ranges for A/B runs).
bool attn_xcd_bands() {
  static const bool on = [] { const char* e = getenv("SPT_ATTN_XCD_BANDS"); return e ? atoi(e) != 0 : true; }();
  return on;
}
int attn_mirror_prepare_launch(const int64_t* ei, const int32_t* eperm, int64_t e, int64_t pairs,
                               int32_t* inv, int32_t* flag, hipStream_t stream);
void attn_pack_tile_ids_mirror_launch(const int32_t* eperm, const int32_t* tgt, const int32_t* src,
                                      const int32_t* inv, int64_t e, int64_t pairs, int32_t* ids4,
                                      hipStream_t stream);
int attn_bwd_to_launch(const float* qkv, int64_t n, const int32_t* erowptr, const int32_t* eperm,
                       const int32_t* tgt, const int32_t* src, const int32_t* tile_ids,
                       const int32_t* tperm, int64_t e, const float* ea, const float* Wk,
                       const float* bk, const float* Wq, const float* bq, const float* Wv,
                       const float* bv, int scale_mode, float scale_a, const float* out,
                       const float* m, const float* z, const float* gout, float* gqkv, float* gea,
                       int gea_acc, float* partial, void* ws, int prec, hipStream_t stream);
int attn_bwd_el_launch(const float* qkv, int64_t n, const int32_t* erowptr, const int32_t* eperm,
                       const int32_t* tgt, const int32_t* src, const int32_t* tile_ids,
                       const int32_t* tperm, const int32_t* trowptr, int64_t e, const float* ea,
                       const float* Wk, const float* bk, const float* Wq, const float* bq,
                       const float* Wv, const float* bv, int scale_mode, float scale_a,
                       const float* out, const float* m, const float* z, const float* gout,
                       float* gqkv, float* gea, int gea_acc, float* partial, void* ws, int prec,
                       hipStream_t stream);
// 0: lane-per-output VALU kernels, 1: f32 matrix pipe (bitwise an fmaf chain), 2 (default):
// split-bf16 on the bf16 matrix pipe (3 products per f32 product, ~2^-17 relative per product: 17 of f32's 24 bits), 3: plain
// bf16 operands, f32 accumulate (the bf16 precision mode)
static std::atomic<int> g_attn_mfma{-1};  // -1: decide from the environment on first use
static int mfma_mode() {
  if (g_attn_mfma < 0) g_attn_mfma = getenv("SPT_ATTN_VALU_ONLY") == nullptr ? 2 : 0;
  return g_attn_mfma;
}
static bool use_mfma() { return mfma_mode() != 0; }
// backward of the bf16-pipe modes: 2 (default) edge-lane kernel (edge_attn_el.hip; needs the
// larger workspace of spt_edge_attn_bwd_ex_workspace_bytes, else 1 is used), 1 packed tiles over
// the edge stream, 0 one tile set per source node
static std::atomic<int> g_attn_bwd_packed{2};
// per-call mode word (include/spt_hip.h): < 0 = the process defaults above
static int mode_precision(int mode) { return mode < 0 ? mfma_mode() : (mode & 3); }
static int mode_bwd_form(int mode) {
  const int f = mode < 0 ? 0 : ((mode >> 4) & 3);
  return f == 0 ? (int)g_attn_bwd_packed : f - 1;
}
// bits 6-7 of the mode word: edge order of the edge-lane backward (0 = the process default of
// spt_attn_bwd_el_target_order, SPT_ATTN_BWD_TARGET_ORDER, SPT_ATTN_BWD_SOURCE_ORDER)
static bool mode_target_order(int mode) {
  const int o = mode < 0 ? 0 : ((mode >> 6) & 3);
  return o == 0 ? attn_bwd_to_enabled() : o == 1;
}
}  // namespace spt

using namespace spt;

#define SPT_ATTN_DISPATCH(FN, ...)                                             \
  do {                                                                         \
    bool done__ = false;                                                       \
    SPT_ATTN_CASE(FN, 1, 1, 32, __VA_ARGS__)                                   \
    SPT_ATTN_CASE(FN, 1, 2, 32, __VA_ARGS__)                                   \
    SPT_ATTN_CASE(FN, 2, 2, 32, __VA_ARGS__)                                   \
    SPT_ATTN_CASE(FN, 1, 1, 16, __VA_ARGS__)  /* nano: 16-D edge encodings */  \
    SPT_ATTN_CASE(FN, 1, 1, 18, __VA_ARGS__)                                   \
    SPT_ATTN_CASE(FN, 1, 2, 18, __VA_ARGS__)                                   \
    SPT_ATTN_CASE(FN, 2, 2, 18, __VA_ARGS__)                                   \
    if (!done__)                                                               \
      return ::spt::fail(-4, "%s: unsupported attention shape H=%d D=%d Dv=%d F=%d " \
                         "(built: H*D<=128, H*Dv<=128, F in {18,32}; F=16 with H*D, H*Dv <= 64)", __func__, H, D, Dv, F); \
  } while (0)

extern "C" int spt_attn_bwd_packed(int on) {
  const int prev = g_attn_bwd_packed;
  g_attn_bwd_packed = on < 0 ? 0 : (on > 2 ? 2 : on);
  return prev;
}

extern "C" int spt_attn_use_mfma(int mode) {
  const int prev = mfma_mode();
  if (mode < -1) return prev;            // query only
  g_attn_mfma = mode < 0 ? 0 : (mode > 3 ? 3 : mode);
  return prev;
}

extern "C" int spt_edge_attn_fwd_f32(const float* qkv, int64_t n, int H, int D, int Dv,
                                     const int32_t* erowptr, const int32_t* eperm,
                                     const int32_t* tgt_sorted, int64_t e,
                                     const float* edge_attr, int F, const float* Wk,
                                     const float* bk, const float* Wq, const float* bq,
                                     const float* Wv, const float* bv, int scale_mode,
                                     float scale_a, float* out, float* m, float* z,
                                     spt_stream_t stream_) {
  return spt_edge_attn_fwd_ex_f32(qkv, n, H, D, Dv, erowptr, eperm, tgt_sorted, e, edge_attr, F, Wk,
                                  bk, Wq, bq, Wv, bv, scale_mode, scale_a, out, m, z, -1, stream_);
}

extern "C" int spt_edge_attn_fwd_ex_f32(const float* qkv, int64_t n, int H, int D, int Dv,
                                        const int32_t* erowptr, const int32_t* eperm,
                                        const int32_t* tgt_sorted, int64_t e,
                                        const float* edge_attr, int F, const float* Wk,
                                        const float* bk, const float* Wq, const float* bq,
                                        const float* Wv, const float* bv, int scale_mode,
                                        float scale_a, float* out, float* m, float* z, int mode,
                                        spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int prec = mode_precision(mode);
  SPT_CHECK_ARG(n >= 0 && e >= 0 && H >= 1 && D >= 1 && Dv >= 1, "bad shape");
  if (n == 0) return 0;
  SPT_CHECK_ARG(qkv && erowptr && out && (tgt_sorted || e == 0), "null pointer");
  SPT_CHECK_ARG((m == nullptr) == (z == nullptr), "pass both m and z or neither");
  if (prec != 0 && attn_mfma_shape_ok(H, D, Dv, F, edge_attr, Wk, Wq, Wv)) {
    attn_fwd_mfma_launch(qkv, n, erowptr, eperm, tgt_sorted, edge_attr, Wk, bk, Wq, bq, Wv, bv,
                         scale_mode, scale_a, out, m, z,
                         prec == 2 ? 3 : (prec == 3 ? 1 : 0), stream);
    SPT_CHECK_LAUNCH();
    return 0;
  }
  if (!edge_attr) F = 32;  // no RPE: any compiled F will do
  const int qpl = per_lane(H * D), vpl = per_lane(H * Dv);
  const int ld = 2 * H * D + H * Dv;
  const int grid = (int)(ceil_div(n, EA_WAVES) < 256 * 8 ? ceil_div(n, EA_WAVES) : 256 * 8);
#define SPT_ATTN_CASE(FN, Q, V, FF, ...)                                        \
  if (!done__ && qpl == Q && vpl == V && F == FF) {                            \
    AttnShape sh;                                                              \
    if (attn_shape(H, D, Dv, Q, V, &sh)) {                                     \
      FN<Q, V, FF><<<grid, EA_WAVES * 64, 0, stream>>>(__VA_ARGS__);           \
      done__ = true;                                                           \
    }                                                                          \
  }
  // `sh` is rebuilt inside each case; the kernel argument list refers to it
  SPT_ATTN_DISPATCH(edge_attn_fwd_kernel, qkv, ld, n, sh, erowptr, eperm, tgt_sorted,
                    edge_attr, Wk, bk, Wq, bq, Wv, bv, scale_mode, scale_a, out, m, z);
#undef SPT_ATTN_CASE
  SPT_CHECK_LAUNCH();
  return 0;
}

constexpr int EA_BWD_BLOCKS = 256;  // one 4-wave workgroup per CU (1 wave per SIMD)

static size_t attn_tables_bytes(int H, int D, int Dv, int F) {
  const size_t len = (size_t)(2 * H * D + H * Dv) * (F + 1);
  return align_up((size_t)EA_BWD_BLOCKS * EA_WAVES * len * 4, 256) + align_up(len * 4, 256);
}

extern "C" size_t spt_edge_attn_bwd_workspace_bytes(int H, int D, int Dv, int F) {
  return attn_tables_bytes(H, D, Dv, F);
}

extern "C" int spt_edge_attn_bwd_el_supported(int H, int D, int Dv, int F, int mode) {
  return H == 16 && D == 4 && Dv == 4 && F == 32 && mode_bwd_form(mode) == 2 &&
         mode_precision(mode) >= 2;
}

extern "C" size_t spt_edge_attn_bwd_ex_workspace_bytes(int64_t n, int64_t e, int H, int D, int Dv,
                                                       int F) {
  // enough for EITHER edge order of the edge-lane backward: the target-order layout (640 B of
  // records per node + 256 B of dq rows per edge) is the larger one on graphs with e < ~1.4 n
  const size_t el = attn_bwd_el_workspace_bytes(n, e), to = attn_bwd_to_workspace_bytes(n, e);
  return attn_tables_bytes(H, D, Dv, F) + (el > to ? el : to);
}

extern "C" int spt_edge_attn_bwd_f32(const float* qkv, int64_t n, int H, int D, int Dv,
                                     const int32_t* erowptr, const int32_t* eperm,
                                     const int32_t* tgt_sorted, int64_t e,
                                     const float* edge_attr, int F, const float* Wk,
                                     const float* bk, const float* Wq, const float* bq,
                                     const float* Wv, const float* bv, int scale_mode,
                                     float scale_a, const float* out, const float* m,
                                     const float* z, const float* gout, float* gqkv,
                                     float* gedge_attr, float* gWk, float* gbk,
                                     float* gWq, float* gbq, float* gWv, float* gbv,
                                     void* ws, size_t ws_bytes, spt_stream_t stream_) {
  return spt_edge_attn_bwd_ex_f32(qkv, n, H, D, Dv, erowptr, eperm, tgt_sorted, nullptr, nullptr,
                                  nullptr, nullptr, e, edge_attr, F, Wk, bk, Wq, bq, Wv, bv,
                                  scale_mode, scale_a, out, m, z, gout, gqkv, gedge_attr, 0, gWk, gbk,
                                  gWq, gbq, gWv, gbv, -1, ws, ws_bytes, stream_);
}

// Same, with `gedge_attr_accumulate` != 0: d edge_attr is ADDED to what gedge_attr holds (f32
// hardware atomics, fire-and-forget) instead of stored - the blocks of one stage all read the same
// edge_attr, so their backward passes can share one gradient buffer instead of autograd summing
// 3-4 [E, F] tensors afterwards.
extern "C" int spt_edge_attn_bwd_acc_f32(const float* qkv, int64_t n, int H, int D, int Dv,
                                         const int32_t* erowptr, const int32_t* eperm,
                                         const int32_t* tgt_sorted, int64_t e,
                                         const float* edge_attr, int F, const float* Wk,
                                         const float* bk, const float* Wq, const float* bq,
                                         const float* Wv, const float* bv, int scale_mode,
                                         float scale_a, const float* out, const float* m,
                                         const float* z, const float* gout, float* gqkv,
                                         float* gedge_attr, int gedge_attr_accumulate,
                                         float* gWk, float* gbk, float* gWq, float* gbq,
                                         float* gWv, float* gbv, void* ws, size_t ws_bytes,
                                         spt_stream_t stream_) {
  return spt_edge_attn_bwd_ex_f32(qkv, n, H, D, Dv, erowptr, eperm, tgt_sorted, nullptr, nullptr,
                                  nullptr, nullptr, e, edge_attr, F, Wk, bk, Wq, bq, Wv, bv,
                                  scale_mode, scale_a, out, m, z, gout, gqkv, gedge_attr,
                                  gedge_attr_accumulate, gWk, gbk, gWq, gbq, gWv, gbv, -1, ws,
                                  ws_bytes, stream_);
}

// [ceil(e / 16)][48] int32 tile records of the edge-lane backward: per 16 CSR positions the edge
// rows (eperm, or the positions when NULL), targets and sources (positions beyond e repeat the last
// edge).  Depends on the graph only: a caller builds it once per batch and level and hands it to
// spt_edge_attn_bwd_ex_f32 (which otherwise rebuilds it in its workspace on every call).
extern "C" int spt_attn_pack_tile_ids(const int32_t* eperm, const int32_t* tgt_sorted,
                                      const int32_t* src_sorted, int64_t e, int32_t* tile_ids,
                                      spt_stream_t stream_) {
  SPT_CHECK_ARG(e >= 0 && (e == 0 || (tgt_sorted && src_sorted && tile_ids)), "null pointer");
  // the 48-int source-order records: the target-order backward (the default) reads another format
  // and would silently walk garbage - refuse instead of writing records nobody can use
  SPT_CHECK_ARG(!attn_bwd_to_enabled(),
                "the target-order backward is on: build the tile records with spt_attn_pack_tile_ids_ex / "
                "_m (or switch it off with spt_attn_bwd_el_target_order(0))");
  attn_pack_tile_ids_launch(eperm, tgt_sorted, src_sorted, e, tile_ids, (hipStream_t)stream_);
  SPT_CHECK_LAUNCH();
  return 0;
}

// Tile records in the format the edge-lane backward of the current process setting reads
// (spt_attn_bwd_el_target_order): 64 ints per tile in TARGET order when it is on (needs `tperm`,
// the CSR view of the targets over the CSR positions, and `src_sorted`), else the 48-int
// source-order records of spt_attn_pack_tile_ids.  spt_attn_tile_record_ints() = ints per tile.
extern "C" int spt_attn_tile_record_ints(void) { return attn_bwd_to_enabled() ? 64 : 48; }
extern "C" int spt_attn_tile_record_ints_m(int mode) { return mode_target_order(mode) ? 64 : 48; }
extern "C" int spt_attn_pack_tile_ids_ex(const int32_t* eperm, const int32_t* tgt_sorted,
                                         const int32_t* src_sorted, const int32_t* tperm, int64_t e,
                                         int32_t* tile_ids, spt_stream_t stream_) {
  return spt_attn_pack_tile_ids_m(eperm, tgt_sorted, src_sorted, tperm, e, -1, tile_ids, stream_);
}
// Same with the edge order taken from bits 6-7 of a per-call `mode` word (the word handed to
// spt_edge_attn_bwd_ex_f32 afterwards): the records' format follows the call, not the process.
extern "C" int spt_attn_pack_tile_ids_m(const int32_t* eperm, const int32_t* tgt_sorted,
                                        const int32_t* src_sorted, const int32_t* tperm, int64_t e,
                                        int mode, int32_t* tile_ids, spt_stream_t stream_) {
  SPT_CHECK_ARG(e >= 0 && (e == 0 || (tgt_sorted && src_sorted && tile_ids)), "null pointer");
  if (mode_target_order(mode)) {
    SPT_CHECK_ARG(e == 0 || tperm, "target-order tile records need the target view (tperm)");
    attn_pack_tile_ids_to_launch(eperm, tgt_sorted, src_sorted, tperm, e, tile_ids, (hipStream_t)stream_);
  } else {
    attn_pack_tile_ids_launch(eperm, tgt_sorted, src_sorted, e, tile_ids, (hipStream_t)stream_);
  }
  SPT_CHECK_LAUNCH();
  return 0;
}

// Target-order tile records from the MIRROR structure of the edge list instead of a sorted target
// view (edge_attn_to.hip): `edge_index` = the [2, e] int64 list as the reference hands it over,
// `pairs` = M with edge i < M mirrored at i + M and every edge from 2 M on a self loop.
// spt_attn_mirror_prepare writes inv [e] (the inverse of eperm) and ORs into *flag (int32, device;
// the caller clears it): bit 0 = a pair that is not (s, t) / (t, s), bit 1 = a loop with s != t.
// spt_attn_pack_tile_ids_mirror then writes the same 64-int records spt_attn_pack_tile_ids_m
// writes in target order (the edges into a node in another order: sums of the same terms).
extern "C" int spt_attn_mirror_prepare(const int64_t* edge_index, const int32_t* eperm, int64_t e,
                                       int64_t pairs, int32_t* inv, int32_t* flag,
                                       spt_stream_t stream_) {
  SPT_CHECK_ARG(e >= 0 && pairs >= 0 && 2 * pairs <= e, "bad shape");
  SPT_CHECK_ARG(e == 0 || (edge_index && eperm && inv && flag), "null pointer");
  attn_mirror_prepare_launch(edge_index, eperm, e, pairs, inv, flag, (hipStream_t)stream_);
  SPT_CHECK_LAUNCH();
  return 0;
}
extern "C" int spt_attn_pack_tile_ids_mirror(const int32_t* eperm, const int32_t* tgt_sorted,
                                             const int32_t* src_sorted, const int32_t* inv,
                                             int64_t e, int64_t pairs, int32_t* tile_ids,
                                             spt_stream_t stream_) {
  SPT_CHECK_ARG(e >= 0 && pairs >= 0 && 2 * pairs <= e, "bad shape");
  SPT_CHECK_ARG(e == 0 || (eperm && tgt_sorted && src_sorted && inv && tile_ids), "null pointer");
  attn_pack_tile_ids_mirror_launch(eperm, tgt_sorted, src_sorted, inv, e, pairs, tile_ids,
                                   (hipStream_t)stream_);
  SPT_CHECK_LAUNCH();
  return 0;
}

// The general entry: `src_sorted` (nullable) = source node of every CSR position (edge_index[0]
// in CSR order); `tile_ids` (nullable) = spt_attn_pack_tile_ids of the graph; `tperm` / `trowptr`
// (nullable, both or none) = CSR view of tgt_sorted over the
// CSR positions (spt_csr_build on tgt_sorted): the edge-lane backward sums dk / dv per target
// through it instead of scattering them with atomics; `mode` = per-call formulation word (< 0:
// process defaults).
extern "C" int spt_edge_attn_bwd_ex_f32(const float* qkv, int64_t n, int H, int D, int Dv,
                                        const int32_t* erowptr, const int32_t* eperm,
                                        const int32_t* tgt_sorted, const int32_t* src_sorted,
                                        const int32_t* tile_ids, const int32_t* tperm,
                                        const int32_t* trowptr, int64_t e, const float* edge_attr,
                                        int F, const float* Wk,
                                        const float* bk, const float* Wq, const float* bq,
                                        const float* Wv, const float* bv, int scale_mode,
                                        float scale_a, const float* out, const float* m,
                                        const float* z, const float* gout, float* gqkv,
                                        float* gedge_attr, int gedge_attr_accumulate,
                                        float* gWk, float* gbk, float* gWq, float* gbq,
                                        float* gWv, float* gbv, int mode, void* ws,
                                        size_t ws_bytes, spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int gea_acc = gedge_attr_accumulate != 0;
  const int prec = mode_precision(mode);
  int form = mode_bwd_form(mode);
  SPT_CHECK_ARG(n >= 0 && e >= 0 && H >= 1 && D >= 1 && Dv >= 1, "bad shape");
  if (n == 0) return 0;
  SPT_CHECK_ARG(qkv && erowptr && out && m && z && gout && gqkv && (tgt_sorted || e == 0), "null pointer");
  const bool has_rpe = edge_attr && (Wk || Wq || Wv);
  SPT_CHECK_ARG(!has_rpe || gedge_attr, "gedge_attr is required with RPE");
  if (!edge_attr) F = 32;
  const int qpl = per_lane(H * D), vpl = per_lane(H * Dv);
  const int ld = 2 * H * D + H * Dv;
  const size_t len = (size_t)ld * (F + 1);
  const size_t need = attn_tables_bytes(H, D, Dv, F);
  SPT_CHECK_ARG(!has_rpe || (ws && ws_bytes >= need), "workspace too small");
  float* partial = has_rpe ? (float*)ws : nullptr;
  const size_t partial_bytes = align_up((size_t)EA_BWD_BLOCKS * EA_WAVES * len * 4, 256);
  (void)partial_bytes;   // (the reduced table used to live behind the partials: written in place now)
  const int grid = (int)(ceil_div(n, EA_WAVES) < EA_BWD_BLOCKS ? ceil_div(n, EA_WAVES) : EA_BWD_BLOCKS);
  // (the target-order route needs either the sorted target view or ready-made tile records -
  // spt_attn_pack_tile_ids_mirror builds them without a view; the source-order route needs the view)
  const bool el_ok = form == 2 && prec >= 2 && e > 0 &&
                     (tperm || (mode_target_order(mode) && tile_ids && src_sorted)) &&
                     attn_mfma_shape_ok(H, D, Dv, F, edge_attr, Wk, Wq, Wv);
  // The edge-lane route is decided ONCE, from the mode word and from what the caller handed over;
  // each edge order has its own scratch layout, and a route only runs inside the bytes it needs
  // (spt_edge_attn_bwd_ex_workspace_bytes covers both).
  const bool to_sel = mode_target_order(mode);
  const size_t need_el = need + attn_bwd_el_workspace_bytes(n, e);
  const size_t need_to = need + attn_bwd_to_workspace_bytes(n, e);
  const bool run_to = el_ok && to_sel && src_sorted && ws_bytes >= need_to;
  const bool run_el = el_ok && !run_to && tperm && ws_bytes >= need_el;
  // records built for the target order (64 ints) must not reach the source-order kernel (48 ints):
  // when the call falls back from the selected target order, the ids are rebuilt in the workspace
  if (run_el && to_sel) tile_ids = nullptr;
  // gqkv receives atomics: start from zero (the edge-lane paths initialise it themselves)
  if (!(run_to || run_el)) hipMemsetAsync(gqkv, 0, (size_t)n * ld * 4, stream);
  if (prec != 0 && attn_mfma_shape_ok(H, D, Dv, F, edge_attr, Wk, Wq, Wv)) {
    int ntab;
    if (form == 2 && mode >= 0 && ((mode >> 4) & 3) == 3)
      SPT_CHECK_ARG((run_to && (trowptr || tile_ids)) || (run_el && trowptr),
                    "edge-lane backward: workspace of spt_edge_attn_bwd_ex_workspace_bytes, the "
                    "target CSR view (or target-order tile records) and a bf16-pipe precision are required");
    if (mode >= 0 && ((mode >> 6) & 3) == 1 && el_ok)
      SPT_CHECK_ARG(run_to, "target-order backward: src_sorted and the workspace of "
                            "spt_edge_attn_bwd_ex_workspace_bytes are required");
    SPT_CHECK_ARG((tperm == nullptr) == (trowptr == nullptr), "pass both tperm and trowptr or neither");
    if (run_to) {
      // edge stream in target order (edge_attn_to.hip); `tile_ids`, when given, are the 64-int
      // records of spt_attn_pack_tile_ids_ex
      ntab = attn_bwd_to_launch(qkv, n, erowptr, eperm, tgt_sorted, src_sorted, tile_ids, tperm, e,
                                edge_attr, Wk, bk, Wq, bq, Wv, bv, scale_mode, scale_a, out, m, z,
                                gout, gqkv, gedge_attr, gea_acc, partial, (char*)ws + need,
                                prec == 2 ? 3 : 1, stream);
    } else if (run_el) {
      ntab = attn_bwd_el_launch(qkv, n, erowptr, eperm, tgt_sorted, src_sorted, tile_ids, tperm,
                                trowptr, e, edge_attr, Wk,
                                bk, Wq, bq, Wv, bv, scale_mode, scale_a, out, m, z, gout, gqkv,
                                gedge_attr, gea_acc, partial, (char*)ws + need, prec == 2 ? 3 : 1,
                                stream);
    } else {
      ntab = attn_bwd_mfma_launch(qkv, n, erowptr, eperm, tgt_sorted, edge_attr, Wk, bk,
                                  Wq, bq, Wv, bv, scale_mode, scale_a, out, m, z, gout,
                                  gqkv, gedge_attr, gea_acc, partial,
                                  prec == 2 ? 3 : (prec == 3 ? 1 : 0), e, form != 0, stream);
    }
    attn_reduce_partials_kernel<<<(int)ceil_div((int64_t)len, 16), 256, 0, stream>>>(
        partial, ntab, (int)len, H * D, H * Dv, F, gWk, gbk, gWq, gbq, gWv, gbv);
    SPT_CHECK_LAUNCH();
    return 0;
  }
#define SPT_ATTN_CASE(FN, Q, V, FF, ...)                                        \
  if (!done__ && qpl == Q && vpl == V && F == FF) {                            \
    AttnShape sh;                                                              \
    if (attn_shape(H, D, Dv, Q, V, &sh)) {                                     \
      FN<Q, V, FF><<<grid, EA_WAVES * 64, 0, stream>>>(__VA_ARGS__);           \
      done__ = true;                                                           \
    }                                                                          \
  }
  SPT_ATTN_DISPATCH(edge_attn_bwd_kernel, qkv, ld, n, sh, erowptr, eperm, tgt_sorted,
                    edge_attr, Wk, bk, Wq, bq, Wv, bv, scale_mode, scale_a, out, m, z,
                    gout, gqkv, gedge_attr, partial, gea_acc);
#undef SPT_ATTN_CASE
  if (has_rpe) {
    attn_reduce_partials_kernel<<<(int)ceil_div((int64_t)len, 16), 256, 0, stream>>>(
        partial, grid * EA_WAVES, (int)len, H * D, H * Dv, F, gWk, gbk, gWq, gbq, gWv, gbv);
  }
  SPT_CHECK_LAUNCH();
  return 0;
}
