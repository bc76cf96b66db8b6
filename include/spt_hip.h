/*
 * spt_hip.h — flat C ABI of libspt_hip.so, the MI355X (gfx950) hot path of
 * Superpoint Transformer.
 *
 * Every entry point replaces one call the reference makes into an un-vendored
 * native dependency (torch_scatter / torch_geometric / FRNN / pgeof) or one
 * multi-launch Python composite of those calls.  Reference citations are
 * file:line relative to drprojects/superpoint_transformer v3.0.0.
 *
 * Conventions
 *   - all pointers are DEVICE pointers owned by the caller (PyTorch's caching
 *     allocator in practice); the library never allocates device memory.
 *     Scratch comes from a caller-provided workspace (ws, ws_bytes) whose size
 *     is returned by the paired *_workspace_bytes() query;
 *   - every function is asynchronous on `stream` (a hipStream_t passed as
 *     void*) and re-entrant.  State held by the library: the thread-local
 *     last-error string, and a handful of process-wide FORMULATION DEFAULTS
 *     (std::atomic<int>, set by the spt_*_use_* / spt_attn_bwd_* setters below or
 *     read once from SPT_* environment variables: matrix-pipe precision, backward
 *     tiling, edge order, LDS-DMA staging, streaming segment-max, cell-centric
 *     kNN).  They select between implementations of the SAME result and apply
 *     only where a call passes no choice of its own: every op they govern has a
 *     *_ex / *_m entry whose `mode` word carries the choice per call, so two
 *     callers (threads, streams, models) never change each other's arithmetic;
 *   - return value: 0 = ok, <0 = error (see spt_last_error()).  Nothing throws
 *     across the boundary;
 *   - row-major, contiguous tensors; f32 features; int64 indices as the
 *     reference hands them over (NAGCast, src/transforms/data.py:54-150);
 *     int32 perm/rowptr inside CSR views (N0 < 2^31 rows);
 *   - segment ops take a CSR view (perm, rowptr) of an UNSORTED index built by
 *     spt_csr_build(): rows of segment s are perm[rowptr[s] .. rowptr[s+1]) in
 *     ascending original order (stable), so reductions are deterministic and
 *     the arg-tie rule is "first occurrence" like torch_scatter's CPU kernel.
 */
#ifndef SPT_HIP_H
#define SPT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* spt_stream_t; /* hipStream_t */

enum spt_reduce_op {
  SPT_SUM = 0,
  SPT_MEAN = 1,
  SPT_MIN = 2,
  SPT_MAX = 3
};

/* Round 6: a GraphNorm whose tables a fused layer call writes from the statistics it has just
 * summed, instead of the caller making a second call for them (spt_graphnorm_tables_f32 /
 * spt_graphnorm_bwd_tables_f32: same formulas, same bits).  HOST structs of DEVICE pointers, passed
 * by address and copied into the launch: weight / mean_scale [d] are the norm's parameters
 * (PyG GraphNorm, src/nn/mlp.py:85-94); the forward tables mean / rstd / am / scale are
 * [num_graphs, d]; the backward tables c1 / c2 / c3 [num_graphs, d] and the parameter gradients
 * gweight / gbias / gmean_scale [d]. */
typedef struct spt_gn_fwd_tables {
  const float* weight;
  const float* mean_scale;
  float eps;
  float* mean;
  float* rstd;
  float* am;
  float* scale;
} spt_gn_fwd_tables;
typedef struct spt_gn_bwd_tables {
  const float* weight;
  const float* mean_scale;
  const float* mean;
  const float* rstd;
  float* c1;
  float* c2;
  float* c3;
  float* gweight;
  float* gbias;
  float* gmean_scale;
} spt_gn_bwd_tables;

/* Library ABI version (major*1000 + minor). */
int spt_version(void);
/* Thread-local description of the last error returned by this library. */
const char* spt_last_error(void);

/* ------------------------------------------------------------------------
 * CSR view of an unsorted segment index.
 * Replaces the implicit "group by index" that every torch_scatter COO kernel
 * redoes with atomics (call sites: src/nn/pool.py:61-62, src/nn/norm.py:118-126,
 * src/nn/attention.py:307,315, src/data/nag.py:97,108).
 *   idx     [n]          int64, values in [0, num_seg)
 *   perm    [n]          int32 out: stable argsort of idx
 *   rowptr  [num_seg+1]  int32 out: segment s owns perm[rowptr[s]..rowptr[s+1])
 * ---------------------------------------------------------------------- */
size_t spt_csr_build_workspace_bytes(int64_t n, int64_t num_seg);
int spt_csr_build(const int64_t* idx, int64_t n, int64_t num_seg,
                  int32_t* perm, int32_t* rowptr,
                  void* ws, size_t ws_bytes, spt_stream_t stream);
/* Companions of the view: pos_seg[j] = segment of CSR position j (= idx[perm[j]], expanded from
 * rowptr: no gather), and out[j] = (int32) src[perm[j]] (e.g. edge_index[1] in CSR order) - one
 * kernel each where the host side used three torch launches (cast, index_select, cast). */
int spt_csr_pos_seg(const int32_t* rowptr, int64_t num_seg, int64_t n, int32_t* pos_seg,
                    spt_stream_t stream);
int spt_csr_gather_i64_i32(const int64_t* src, const int32_t* perm, int64_t n, int32_t* out,
                           spt_stream_t stream);

/* ------------------------------------------------------------------------
 * Segment reduce  out[s,:] = reduce_{i in seg s} x[i,:]      (a1, a4, a8)
 * Replaces torch_scatter.scatter / scatter_{sum,mean,min,max} and the PyG
 * {Sum,Mean,Min,Max}Aggregation the pools dispatch to
 * (src/nn/pool.py:61-82 <- src/nn/stage.py:429-431; src/nn/norm.py:118-126).
 *   x    [n,c] f32      out [num_seg,c] f32
 *   arg  [num_seg,c] int32 or NULL: for MIN/MAX the row index of the selected
 *        element, n for an empty segment (torch_scatter's sentinel).
 * Empty segments reduce to 0.  MEAN divides by max(count,1).
 * ---------------------------------------------------------------------- */
int spt_segcsr_reduce_f32(int op, const float* x, const int32_t* perm,
                          const int32_t* rowptr, int64_t n, int64_t num_seg,
                          int c, float* out, int32_t* arg, spt_stream_t stream);
/* Max + arg of 128-channel rows over >= 65 536 rows (the level-0 -> level-1 pool) runs a
 * row-streaming kernel: a wave owns a contiguous range of CSR positions cut at segment boundaries
 * and streams sixteen 512-byte rows at a time whatever segments they belong to (1, default;
 * SPT_SEG_STREAM=0 in the environment or 0 here = the lane-group-per-segment kernel for every
 * shape).  Bit-identical results.  Returns the previous setting. */
int spt_segcsr_use_stream(int on);
/* Same two entries with the formulation chosen PER CALL (-1: process default, 0: lane group per
 * segment, 1: row-streaming where it applies); they touch no process-wide state. */
int spt_segcsr_reduce_ex_f32(int op, const float* x, const int32_t* perm,
                             const int32_t* rowptr, int64_t n, int64_t num_seg,
                             int c, float* out, int32_t* arg, int formulation,
                             spt_stream_t stream);
int spt_segcsr_max_affine_ex_f32(const float* x, const int32_t* perm, const int32_t* rowptr,
                                 int64_t n, int64_t num_seg, int c, const float* am,
                                 const float* scale, const float* bias, float act_slope,
                                 const int64_t* seg_graph, float* out, int32_t* arg,
                                 int formulation, spt_stream_t stream);

/* Backward of the above w.r.t. x (a2):
 *   SUM : gx[i,:] = gout[idx[i],:]
 *   MEAN: gx[i,:] = gout[idx[i],:] / max(count[idx[i]],1)
 *   MIN/MAX: gx[i,c] = gout[idx[i],c] if arg[idx[i],c]==i else 0
 * (torch_scatter routes the gradient to the single arg element.)           */
/* Max-pool of y = leaky(scale[g] (x - am[g]) + bias) computed on the fly from the RAW x:
 * the GraphNorm-apply + LeakyReLU that ends the point MLP (src/nn/mlp.py:46-50) folded into
 * the L0 -> L1 pool read (src/nn/pool.py:61-82).  am / scale [B, c] per graph, bias [c],
 * seg_graph [num_seg] int64 graph of each segment (NULL = one graph); out / arg as above.
 * Same values and arg as applying the norm first, without writing the normalised rows. */
int spt_segcsr_max_affine_f32(const float* x, const int32_t* perm, const int32_t* rowptr,
                              int64_t n, int64_t num_seg, int c, const float* am,
                              const float* scale, const float* bias, float act_slope,
                              const int64_t* seg_graph, float* out, int32_t* arg,
                              spt_stream_t stream);
/* Same with x stored as bf16 [n, c] (SPT_FMLP_H_BF16 below); built for c = 128, n >= 65 536. */
int spt_segcsr_max_affine_bf16_supported(int c, int64_t n);
int spt_segcsr_max_affine_bf16(const void* x_bf16, const int32_t* perm, const int32_t* rowptr,
                               int64_t n, int64_t num_seg, int c, const float* am,
                               const float* scale, const float* bias, float act_slope,
                               const int64_t* seg_graph, float* out, int32_t* arg,
                               spt_stream_t stream);
/* Same pool (x f32, or bf16 rows with x_is_bf16 != 0) with a third output raw [num_seg, c] f32 =
 * x[arg[s, c], c], the winner's value BEFORE the map (0 for an empty segment): what the top
 * GraphNorm's backward statistics of a max-pooled layer need (spt_graphnorm_bwd_stats_sparse_raw_f32
 * reads it as a stream instead of gathering).  Built for c = 128, n >= 65 536. */
int spt_segcsr_max_affine_raw_supported(int c, int64_t n);
int spt_segcsr_max_affine_raw_f32(const void* x, int x_is_bf16, const int32_t* perm,
                                  const int32_t* rowptr, int64_t n, int64_t num_seg, int c,
                                  const float* am, const float* scale, const float* bias,
                                  float act_slope, const int64_t* seg_graph, float* out,
                                  int32_t* arg, float* raw, spt_stream_t stream);
int spt_segcsr_reduce_bwd_f32(int op, const float* gout, const int32_t* arg,
                              const int64_t* idx, const int32_t* perm,
                              const int32_t* rowptr, int64_t n, int64_t num_seg,
                              int c, float* gx, spt_stream_t stream);

/* Wide rows (>= 128 B) stream in CSR order when perm/rowptr are given (the
 * parent row is read once per segment); narrow rows run in idx order. */

/* Bit-exact integer segment sum (a8: NAG.get_sub_size, src/data/nag.py:59-110). */
int spt_segcsr_sum_i64(const int64_t* x, const int32_t* perm,
                       const int32_t* rowptr, int64_t n, int64_t num_seg,
                       int c, int64_t* out, spt_stream_t stream);

/* ------------------------------------------------------------------------
 * Row gather  out[i,:] = x[idx[i],:]                               (a3)
 * Replaces IndexUnpool (src/nn/unpool.py:12-13) and the parent->child
 * broadcasts of src/nn/norm.py:132-133, src/nn/stage.py:269.  Its backward
 * is spt_segcsr_reduce_f32(SPT_SUM) on the CSR view of idx.
 * ---------------------------------------------------------------------- */
int spt_gather_rows_f32(const float* x, const int64_t* idx, int64_t n,
                        int64_t num_src, int c, float* out, spt_stream_t stream);

/* ------------------------------------------------------------------------
 * Fused UnitSphereNorm                                              (a4)
 * Replaces UnitSphereNorm._forward_scatter / _forward
 * (src/nn/norm.py:86-138 <- src/nn/stage.py:250) and its helper
 * scatter_mean_weighted (src/utils/scatter.py:17-38): per segment bounding
 * box -> diameter = max axis span; centre = (weighted) mean;
 * pos_out = (pos - centre[idx]) / (diameter[idx] + 1e-2).
 *   pos [n,3] f32; idx [n] int64 (NULL = one segment, norm.py:86-110);
 *   (perm, rowptr) = CSR view of idx; w_f32 / w_i64: optional weights (at
 *   most one non-NULL; int64 node_size is cast to f32 like the reference);
 *   pos_out [n,3], diam [num_seg], center [num_seg,3] f32 out.
 * Empty segment -> diameter 0, centre 0.  No gradient (positions are data).
 * ---------------------------------------------------------------------- */
int spt_unit_sphere_norm_f32(const float* pos, const int64_t* idx,
                             const int32_t* perm, const int32_t* rowptr,
                             const float* w_f32, const int64_t* w_i64,
                             int64_t n, int64_t num_seg, float* pos_out,
                             float* diam, float* center, spt_stream_t stream);
/* The same statistics with the stage's input assembly (src/nn/stage.py:249-271, the three
 * fusion concatenations for use_pos / use_diameter_parent) done by the writing pass:
 * xcat [n, 4 + cx] = [diam[idx] | pos_normalised | x], x [n, cx] with cx % 4 == 0 - the
 * normalised positions, the gathered parent diameter and the copy of x are never materialised. */
int spt_unit_sphere_assemble_f32(const float* pos, const int64_t* idx,
                                 const int32_t* perm, const int32_t* rowptr,
                                 const float* w_f32, const int64_t* w_i64,
                                 int64_t n, int64_t num_seg, const float* x, int cx,
                                 float* xcat, float* diam, float* center, spt_stream_t stream);
/* Both entries with a caller workspace of spt_unit_sphere_workspace_bytes(n, num_seg) bytes:
 * segments of thousands of rows (idx = NULL: the whole top level is one segment, norm.py:86-110)
 * are reduced by several workgroups each and merged in a fixed order instead of by one. */
size_t spt_unit_sphere_workspace_bytes(int64_t n, int64_t num_seg);
int spt_unit_sphere_norm_ws_f32(const float* pos, const int64_t* idx, const int32_t* perm,
                                const int32_t* rowptr, const float* w_f32, const int64_t* w_i64,
                                int64_t n, int64_t num_seg, float* pos_out, float* diam,
                                float* center, void* ws, size_t ws_bytes, spt_stream_t stream);
int spt_unit_sphere_assemble_ws_f32(const float* pos, const int64_t* idx, const int32_t* perm,
                                    const int32_t* rowptr, const float* w_f32, const int64_t* w_i64,
                                    int64_t n, int64_t num_seg, const float* x, int cx, float* xcat,
                                    float* diam, float* center, void* ws, size_t ws_bytes,
                                    spt_stream_t stream);

/* ------------------------------------------------------------------------
 * GraphNorm forward / backward, optionally fused with LeakyReLU      (a5)
 * Replaces torch_geometric.nn.norm.GraphNorm (PyG 2.3.0, install.sh:100) as
 * called from src/nn/mlp.py:85-94 (every MLP layer, followed by the
 * LeakyReLU of mlp.py:46-50) and src/nn/transformer.py:258-265:
 *   mu = mean_g x; o = x - mean_scale*mu[batch]; var = mean_g o^2;
 *   y = weight * o / sqrt(var + eps) + bias;  y = leaky_relu(y, act_slope)
 *   x [r,d] f32; batch [r] int64 in [0,num_graphs) or NULL (one graph);
 *   act_slope = 1 disables the activation; mean, rstd [num_graphs,d] out
 *   (saved for the backward).  Statistics accumulate in f64 with a
 *   fixed-order reduction: results are run-to-run deterministic.
 * ws: caller workspace of spt_graphnorm_workspace_bytes(r,d,num_graphs).
 * ---------------------------------------------------------------------- */
size_t spt_graphnorm_workspace_bytes(int64_t r, int d, int num_graphs);
int spt_graphnorm_fwd_f32(const float* x, const int64_t* batch, int64_t r, int d,
                          int num_graphs, const float* weight, const float* bias,
                          const float* mean_scale, float eps, float act_slope,
                          float* y, float* mean, float* rstd, void* ws,
                          size_t ws_bytes, spt_stream_t stream);
int spt_graphnorm_bwd_f32(const float* x, const float* gy, const int64_t* batch,
                          int64_t r, int d, int num_graphs, const float* weight,
                          const float* bias, const float* mean_scale,
                          const float* mean, const float* rstd, float act_slope,
                          float* gx, float* gweight, float* gbias,
                          float* gmean_scale, void* ws, size_t ws_bytes,
                          spt_stream_t stream);
/* Statistics + coefficient tables only (no apply pass): mean, rstd and the rows of
 * y = (x - am[g]) * scale[g] + bias, all [num_graphs, d], for consumers that normalise on the fly
 * (spt_skinny_linear_pre_f32).  ws: spt_graphnorm_workspace_bytes(). */
int spt_graphnorm_stats_f32(const float* x, const int64_t* batch, int64_t r, int d, int num_graphs,
                            const float* weight, const float* mean_scale, float eps, float* mean,
                            float* rstd, float* am, float* scale, void* ws, size_t ws_bytes,
                            spt_stream_t stream);
/* spt_graphnorm_bwd_f32 with gx = gx_add + (backward of the norm); gx_add [r, d] or NULL: the
 * gradient of the residual branch around a pre-norm joins in the apply pass. */
int spt_graphnorm_bwd_acc_f32(const float* x, const float* gy, const int64_t* batch,
                              int64_t r, int d, int num_graphs, const float* weight,
                              const float* bias, const float* mean_scale,
                              const float* mean, const float* rstd, float act_slope,
                              const float* gx_add, float* gx, float* gweight, float* gbias,
                              float* gmean_scale, void* ws, size_t ws_bytes,
                              spt_stream_t stream);

/* ------------------------------------------------------------------------
 * Fused sparse graph self-attention with relative-pose encodings   (a6.2-a6.7)
 * Replaces the body of SelfAttentionBlock.forward between the qkv and the
 * out_proj Linear (src/nn/attention.py:202-315): edge gathers q[s], k[t], v[t],
 * qk scaling (src/utils/nn.py:75-127), the k/q/v RPE Linears on edge_attr,
 * the per-head dot product, torch_geometric.utils.softmax over the edges of
 * each source node and the scatter_sum of the weighted values.
 *   qkv        [n, 2*H*D + H*Dv] f32: output of the block's own qkv Linear,
 *              columns [q | k | v], each head-major (attention.py:202-204)
 *   erowptr    [n+1], eperm [e] (NULL = edges already grouped by source):
 *              CSR view of edge_index[0] from spt_csr_build
 *   tgt_sorted [e] int32: edge_index[1] in CSR order (tgt[eperm[j]])
 *   edge_attr  [e, F] f32 in ORIGINAL edge order, or NULL (no RPE)
 *   Wk,bk / Wq,bq / Wv,bv: k_rpe / q_rpe / v_rpe Linear parameters
 *              ([H*D,F],[H*D],[H*Dv,F],[H*Dv]); any may be NULL (encoder absent)
 *   scale_mode/scale_a: 0: a*deg^-0.5 ('d.g', 'g'), 1: a + deg^-0.5 ('d+g'),
 *              2: a ('d' or a constant); deg(s) = erowptr[s+1]-erowptr[s]
 *   out [n, H*Dv]; m, z [n, H]: running max / sum of the softmax, saved for
 *              the backward (both NULL to skip).
 * Built shapes: H*D <= 128, H*Dv <= 128, F in {18, 32}; anything else returns
 * an error (never a silent fallback).  Forward is deterministic.
 * Backward: gout [n, H*Dv] -> gqkv [n, ld] (zero-filled here; k/v columns use
 * f32 atomics), gedge_attr [e, F], and the six RPE parameter gradients
 * (NULL to skip one).  ws: spt_edge_attn_bwd_workspace_bytes().
 * ---------------------------------------------------------------------- */
int spt_edge_attn_fwd_f32(const float* qkv, int64_t n, int H, int D, int Dv,
                          const int32_t* erowptr, const int32_t* eperm,
                          const int32_t* tgt_sorted, int64_t e,
                          const float* edge_attr, int F, const float* Wk,
                          const float* bk, const float* Wq, const float* bq,
                          const float* Wv, const float* bv, int scale_mode,
                          float scale_a, float* out, float* m, float* z,
                          spt_stream_t stream);
size_t spt_edge_attn_bwd_workspace_bytes(int H, int D, int Dv, int F);
/* Three formulations exist for the SPT-64 head layout (H=16, D=Dv=4, F=32):
 *   2 (default)  matrix pipe, split-bf16: every f32 product of the three RPE GEMMs (and of
 *                the two gradient GEMMs) is hi*hi + lo*hi + hi*lo of bf16 halves on the bf16
 *                MFMA (16x the f32 pipe's rate), f32 accumulate: the dropped lo*lo term and the
 *                halves' rounding leave ~2^-17 relative per product (17 of f32's 24 bits);
 *   1            matrix pipe, f32 in / f32 accumulate (bitwise an fmaf chain);
 *   0            generic lane-per-output VALU kernels (every other shape uses these).
 * spt_attn_use_mfma(mode) selects process-wide and returns the previous mode (mode < -1: query
 * only, nothing changes); tests cross-check all three at full scene size. */
int spt_attn_use_mfma(int mode);
/* Backward tiling of the bf16-pipe modes (2, 3): 2 (default) = the edge-lane kernel described
 * below when the caller's workspace allows it (else 1); 1 = 16-edge tiles over the edge
 * stream, at most two consecutive source nodes per pass, tiles 100 % full (dq by atomicAdd: a
 * node may be cut between two waves); 0 = one tile set per source node (62-69 % full at mean
 * degree 16).  Same results up to f32 summation order.  Returns the previous setting. */
int spt_attn_bwd_packed(int on);
/* Request shape of the edge-lane backward's k / v gathers and [dk | dv] stores: 1 (default) = whole
 * 128-byte lines per instruction, 0 = the 64-byte pieces of the MFMA layout.  Results are bit-identical;
 * process-wide measurement switch (< 0: query), returns the previous setting. */
int spt_attn_bwd_el_full_line(int on);
/* Edge order of the edge-lane backward: 1 (default) = the edge stream sorted by TARGET node
 * (csrc/edge_attn_to.hip: k / v of the target are L1 hits, dk / dv are reduced inside the tile, the
 * streamed per-edge quantity is dq - half the bytes of [dk | dv]), 0 = by source
 * (csrc/edge_attn_el.hip).  Process-wide (< 0: query), returns the previous setting.  The tile
 * records a caller builds once per batch and level differ between the two:
 * spt_attn_pack_tile_ids_ex writes the format of the current setting, spt_attn_tile_record_ints()
 * ints per 16-edge tile (64 / 48). */
int spt_attn_bwd_el_target_order(int on);
int spt_attn_tile_record_ints(void);
int spt_attn_pack_tile_ids_ex(const int32_t* eperm, const int32_t* tgt_sorted,
                              const int32_t* src_sorted, const int32_t* tperm, int64_t e,
                              int32_t* tile_ids, spt_stream_t stream);
/* The same two with the edge order taken from bits 6-7 of a per-call `mode` word (< 0 or 0 in
 * those bits: the process default) - the word later handed to spt_edge_attn_bwd_ex_f32.
 * ABI note (round 4 -> 5): spt_attn_pack_tile_ids (48-int records, declared further down) REFUSES
 * while the process default is the target order, because records of the wrong width would be read
 * as garbage; callers of the older pair (spt_attn_pack_tile_ids + spt_edge_attn_bwd_ex_f32) either
 * move to the _m entries or pin SPT_ATTN_BWD_SOURCE_ORDER in their mode word and call
 * spt_attn_pack_tile_ids_m with it. */
int spt_attn_tile_record_ints_m(int mode);
int spt_attn_pack_tile_ids_m(const int32_t* eperm, const int32_t* tgt_sorted,
                             const int32_t* src_sorted, const int32_t* tperm, int64_t e, int mode,
                             int32_t* tile_ids, spt_stream_t stream);
/* Round 6: TARGET-order tile records without a sorted target view, from the MIRROR structure of
 * the reference's final edge list [i<j | j>i | loops] (src/transforms/graph.py:1268, 1442-1446:
 * OnTheFlyHorizontalEdgeFeatures appends the flipped copy of the trimmed list, NAGAddSelfLoops the
 * loops; the shipped S3DIS / DALES / ScanNet configs skip SampleEdges - `sample_edge_n_min: -1`).
 * `edge_index` = the [2, e] int64 list, `pairs` = M: edge i < M is mirrored at i + M, every edge
 * from 2 M on is a self loop.  The edges INTO a node are the mirrors of the edges OUT OF it, which
 * the by-source view (eperm) holds as one run: no second radix sort per level.
 * spt_attn_mirror_prepare writes inv [e] (int32: the inverse of eperm) and ORs into *flag (int32,
 * device; the caller clears it) bit 0 = a pair (i, i + M) that is not (s, t) / (t, s), bit 1 = a
 * loop with s != t - the structure is CHECKED, not trusted.  spt_attn_pack_tile_ids_mirror writes
 * the 64-int records of spt_attn_pack_tile_ids_m's target order (same fields; the edges into a
 * node in by-source order of their mirrors instead of ascending source: the same sums).
 * spt_edge_attn_bwd_ex_f32 accepts them with tperm = trowptr = NULL. */
int spt_attn_mirror_prepare(const int64_t* edge_index, const int32_t* eperm, int64_t e, int64_t pairs,
                            int32_t* inv, int32_t* flag, spt_stream_t stream);
int spt_attn_pack_tile_ids_mirror(const int32_t* eperm, const int32_t* tgt_sorted,
                                  const int32_t* src_sorted, const int32_t* inv, int64_t e,
                                  int64_t pairs, int32_t* tile_ids, spt_stream_t stream);
/* Round 6: operand layout of the head-group decomposition that runs wider head layouts on the
 * built 16 x (qk 4, value 4) kernels (H = 16 G heads of value dim 4 J; SPT-128 of
 * configs/experiment/semantic/kitti360.yaml:22-27: G = 1, J = 2 - src/nn/attention.py:202-315 is
 * linear in the value dims, so the passes are exact).  qkv rows are [q 64 G | k 64 G | v H x 4 J].
 * spt_attn_split_pack_f32: qa [G J, n, 192] <- per pass p = g J + j the columns
 *   [q_g | k_g | v_g[:, 4 j .. 4 j + 3]] of every row (what one index gather + one transposing copy did);
 * spt_attn_split_grad_f32: gqkv [n, 128 G + 64 G J] <- the passes' gradients gqa [G J, n, 192], the
 *   q / k columns summed over the J slices of their head group in ascending j, the v columns copied
 *   (what a sum, three permuting copies and a cat did).  16-byte aligned pointers. */
int spt_attn_split_pack_f32(const float* qkv, int64_t n, int G, int J, float* qa, spt_stream_t stream);
int spt_attn_split_grad_f32(const float* gqa, int64_t n, int G, int J, float* gqkv, spt_stream_t stream);
int spt_edge_attn_bwd_f32(const float* qkv, int64_t n, int H, int D, int Dv,
                          const int32_t* erowptr, const int32_t* eperm,
                          const int32_t* tgt_sorted, int64_t e,
                          const float* edge_attr, int F, const float* Wk,
                          const float* bk, const float* Wq, const float* bq,
                          const float* Wv, const float* bv, int scale_mode,
                          float scale_a, const float* out, const float* m,
                          const float* z, const float* gout, float* gqkv,
                          float* gedge_attr, float* gWk, float* gbk, float* gWq,
                          float* gbq, float* gWv, float* gbv, void* ws,
                          size_t ws_bytes, spt_stream_t stream);
/* Same with gedge_attr_accumulate != 0: d edge_attr is ADDED (f32 hardware atomics) to what
 * gedge_attr already holds instead of stored.  The transformer blocks of a stage all read the same
 * edge_attr (src/nn/stage.py:137-141), so their backward passes share one gradient buffer instead
 * of leaving autograd to sum 3-4 [E,F] tensors. */
int spt_edge_attn_bwd_acc_f32(const float* qkv, int64_t n, int H, int D, int Dv,
                              const int32_t* erowptr, const int32_t* eperm,
                              const int32_t* tgt_sorted, int64_t e,
                              const float* edge_attr, int F, const float* Wk,
                              const float* bk, const float* Wq, const float* bq,
                              const float* Wv, const float* bv, int scale_mode,
                              float scale_a, const float* out, const float* m,
                              const float* z, const float* gout, float* gqkv,
                              float* gedge_attr, int gedge_attr_accumulate, float* gWk,
                              float* gbk, float* gWq, float* gbq, float* gWv, float* gbv,
                              void* ws, size_t ws_bytes, spt_stream_t stream);

/* Per-call formulation: the *_ex entries take a `mode` word instead of reading the process-wide
 * switches above, so two callers (two models at different precisions, two streams) never change
 * each other's arithmetic.  mode < 0 (SPT_ATTN_DEFAULT): the process defaults.  Otherwise
 *   bits 0-1  precision: SPT_ATTN_VALU 0, SPT_ATTN_F32 1 (f32 matrix pipe), SPT_ATTN_SPLIT_BF16 2,
 *             SPT_ATTN_BF16 3 (operands rounded to bf16);
 *   bits 4-5  backward tiling of the bf16-pipe precisions: 0 auto (edge-lane when the workspace
 *             allows), SPT_ATTN_BWD_PER_NODE, SPT_ATTN_BWD_PACKED, SPT_ATTN_BWD_EDGE_LANE;
 *   bits 6-7  edge order of the edge-lane backward: 0 = the process default
 *             (spt_attn_bwd_el_target_order), SPT_ATTN_BWD_TARGET_ORDER (csrc/edge_attn_to.hip:
 *             dk / dv reduced per target inside the tile, a few float atomics per node - the
 *             sums' order, not their value, varies run to run), SPT_ATTN_BWD_SOURCE_ORDER
 *             (csrc/edge_attn_el.hip: no atomics, every gradient bitwise reproducible).  The
 *             tile records a caller pre-builds must come from spt_attn_pack_tile_ids_m with the
 *             SAME word (64 ints per tile in target order, 48 in source order); when a call
 *             cannot run the selected target order (src_sorted NULL or a workspace below
 *             spt_edge_attn_bwd_ex_workspace_bytes) it falls back to the source order and
 *             rebuilds the records itself - it never reads records of the other format.
 * Edge-lane backward (csrc/edge_attn_el.hip; H=16, D=Dv=4, F=32): 16-edge tiles over the CSR edge
 * stream, a wave owns half of the heads; the recompute GEMM runs transposed so that a lane holds
 * whole heads of ONE edge (no redundant softmax math, per-edge node rows fetched by LDS-DMA, any
 * number of source nodes per tile), dq is reduced per source node on the matrix pipe, two waves
 * per SIMD.  dk / dv are NOT scattered with atomics (the chip retires ~320 G f32 atomic adds / s:
 * 2.8 ms for one level-1 call): the per-edge [dk | dv] rows are streamed to the workspace in CSR
 * order and summed per target through (tperm, trowptr) = spt_csr_build(tgt_sorted, e, n) - a CSR
 * view of the TARGETS over the CSR positions, built once per batch and level - in a fixed order
 * (deterministic).  It needs the scratch of spt_edge_attn_bwd_ex_workspace_bytes(n, e, ...)
 * (per-node rows + 512 B per edge), that view, and optionally src_sorted [e] int32 =
 * edge_index[0] in CSR order (NULL: rebuilt from erowptr).  Without the view or the workspace the
 * packed kernel runs instead.
 * With gedge_attr_accumulate == 0 the edge-lane kernel zero-fills gedge_attr first (both waves of
 * a pair add their halves). */
#define SPT_ATTN_DEFAULT (-1)
#define SPT_ATTN_VALU 0
#define SPT_ATTN_F32 1
#define SPT_ATTN_SPLIT_BF16 2
#define SPT_ATTN_BF16 3
#define SPT_ATTN_BWD_PER_NODE (1 << 4)
#define SPT_ATTN_BWD_PACKED (2 << 4)
#define SPT_ATTN_BWD_EDGE_LANE (3 << 4)
#define SPT_ATTN_BWD_TARGET_ORDER (1 << 6)
#define SPT_ATTN_BWD_SOURCE_ORDER (2 << 6)
int spt_edge_attn_fwd_ex_f32(const float* qkv, int64_t n, int H, int D, int Dv,
                             const int32_t* erowptr, const int32_t* eperm,
                             const int32_t* tgt_sorted, int64_t e,
                             const float* edge_attr, int F, const float* Wk,
                             const float* bk, const float* Wq, const float* bq,
                             const float* Wv, const float* bv, int scale_mode,
                             float scale_a, float* out, float* m, float* z, int mode,
                             spt_stream_t stream);
/* Scratch for either edge order of the edge-lane backward (the larger of the two layouts: the
 * target order needs 644 B per node + 256 B per edge + 16 B per tile id, the source order
 * 388 B per node + 516 B per edge + 12 B per tile id) on top of the weight-gradient tables. */
size_t spt_edge_attn_bwd_ex_workspace_bytes(int64_t n, int64_t e, int H, int D, int Dv, int F);
/* 1 when the edge-lane backward is built for this head layout AND `mode` (< 0: the process
 * defaults) selects it (a caller uses it to decide whether to build the target view). */
int spt_edge_attn_bwd_el_supported(int H, int D, int Dv, int F, int mode);
/* [ceil(e / 16)][48] int32 tile records of the edge-lane backward (edge rows | targets | sources of
 * 16 consecutive CSR positions): depends on the graph only, built once per batch and level by the
 * caller (tile_ids = NULL: rebuilt in the workspace on every call). */
int spt_attn_pack_tile_ids(const int32_t* eperm, const int32_t* tgt_sorted,
                           const int32_t* src_sorted, int64_t e, int32_t* tile_ids,
                           spt_stream_t stream);
int spt_edge_attn_bwd_ex_f32(const float* qkv, int64_t n, int H, int D, int Dv,
                             const int32_t* erowptr, const int32_t* eperm,
                             const int32_t* tgt_sorted, const int32_t* src_sorted,
                             const int32_t* tile_ids, const int32_t* tperm,
                             const int32_t* trowptr, int64_t e, const float* edge_attr, int F,
                             const float* Wk,
                             const float* bk, const float* Wq, const float* bq,
                             const float* Wv, const float* bv, int scale_mode,
                             float scale_a, const float* out, const float* m,
                             const float* z, const float* gout, float* gqkv,
                             float* gedge_attr, int gedge_attr_accumulate, float* gWk,
                             float* gbk, float* gWq, float* gbq, float* gWv, float* gbv,
                             int mode, void* ws, size_t ws_bytes, spt_stream_t stream);

/* ------------------------------------------------------------------------
 * Radius-bounded exact kNN on a uniform grid                        (a9)
 * Replaces frnn.frnn_grid_points of the un-vendored FRNN CUDA extension
 * (src/utils/neighbors.py:48 <- knn_1 :90, knn_2 :231): for each query the K
 * nearest search points with squared distance < r^2 (<= when inclusive),
 * ascending by (distance, index) - ties broken by ascending search index -,
 * missing entries idx = -1 / dist = -1.  d2 = (dx*dx + dy*dy) + dz*dz in f32.
 *   query [nq,3], search [ns,3] f32; K in [1,64]; idx [nq,K] int64,
 *   dist [nq,K] f32 (squared when `squared`, FRNN's convention, else Euclidean);
 *   cell_size/origin[3]/dims[3]: HOST description of the uniform grid that
 *   covers the search points (any cell size gives the same result; ~K/4 points
 *   per non-empty cell is fastest); dims[0]*dims[1]*dims[2] < 2^31;
 *   order_queries_by_cell: when query == search, visit queries in cell order.
 *   cell_order (nullable) [ns] int32: receives the search points' cell order (the
 *   permutation spt_spatial_order computes), a by-product of the grid build.
 * ws: spt_grid_knn_workspace_bytes(ns, ncells).
 * ---------------------------------------------------------------------- */
size_t spt_grid_knn_workspace_bytes(int64_t ns, int64_t ncells);
/* A self-search visited in cell order (query == search, order_queries_by_cell) has two
 * equivalent implementations: waves that own 64 consecutive cell-sorted points and share
 * one candidate stream (default), and one wave per query (also the path of every other
 * call).  spt_knn_use_cell_path(0|1) selects process-wide and returns the previous
 * setting; tests cross-check the two bit for bit. */
int spt_knn_use_cell_path(int on);
/* spt_grid_knn_f32 with the formulation chosen per call (-1: process default, 0: one wave per
 * query, 1: cell-centric self-search where it applies). */
int spt_grid_knn_ex_f32(const float* query, int64_t nq, const float* search, int64_t ns,
                        int K, float r, float cell_size, const float* origin,
                        const int32_t* dims, int order_queries_by_cell, int inclusive,
                        int squared, int64_t* idx, float* dist, int32_t* cell_order,
                        int formulation, void* ws, size_t ws_bytes, spt_stream_t stream);
/* KNN + PointFeatures of the reference's preprocessing chain in ONE call (src/transforms/
 * neighbors.py:11-95 -> src/transforms/point.py:160-180 -> src/utils/geometry.py:80-126): a
 * self-search of `xyz` [n, 3] for the K nearest of every point (itself included, column 0:
 * K = k + 1 of knn_1, K <= 64) that also writes feats [n, 11] = the eigenfeatures (pgeof's
 * column order) of each point's neighbourhood {found points} - what spt_point_geof_dense_f32
 * returns for (xyz, idx[:, 1:], add_self = 1, k_min, post), without the index rows and the
 * neighbours' positions coming back from HBM: the kNN kernel sums the f64 moments of the winners
 * while their rows are still in cache.  The summation order differs from the stand-alone entry:
 * results are identical whenever the sums are exact in f64 (coordinates of similar magnitude,
 * the usual case) and within an ulp of the f32 outputs otherwise.  idx / dist / cell_order /
 * formulation / ws as in spt_grid_knn_ex_f32 (the one-wave-per-query formulation computes the
 * features from the index rows it wrote). */
int spt_grid_knn_geof_f32(const float* xyz, int64_t n, int K, float r, float cell_size,
                          const float* origin, const int32_t* dims, int inclusive, int squared,
                          int k_min, int post, int64_t* idx, float* dist, float* feats,
                          int32_t* cell_order, int formulation, void* ws, size_t ws_bytes,
                          spt_stream_t stream);
/* Probes of the host-side grid description (which cell size makes the search fastest; the
 * RESULT of a search never depends on it).  spt_knn_subsample_f32: the points of the coarse cells
 * (edge `coarse`, origin lo[3] - HOST floats) whose coordinate hash & 0xFFFF is below `thresh`,
 * compacted into out [<= n, 3] in no particular order, their number in the DEVICE int32 *count.
 * spt_grid_cell_ids_f32: the linear cell id ((z * dims[1] + y) * dims[0] + x, clamped) of every
 * point for a grid description as in spt_grid_knn_f32. */
int spt_knn_subsample_f32(const float* xyz, int64_t n, const float* lo, float coarse, int thresh,
                          float* out, int32_t* count, spt_stream_t stream);
int spt_grid_cell_ids_f32(const float* xyz, int64_t n, float cell_size, const float* origin,
                          const int32_t* dims, int64_t* cell, spt_stream_t stream);
/* The number of non-empty cells of that grid in the DEVICE int64 *count (exact: one bit per cell,
 * set by the points, then counted - no sort).  ws: spt_grid_count_cells_workspace_bytes(ncells). */
size_t spt_grid_count_cells_workspace_bytes(int64_t ncells);
int spt_grid_count_cells_f32(const float* xyz, int64_t n, float cell_size, const float* origin,
                             const int32_t* dims, int64_t* count, void* ws, size_t ws_bytes,
                             spt_stream_t stream);
/* Bounding box of a cloud, the input of the host-side grid description above
 * (src/utils/neighbors.py has no counterpart: FRNN derives its grid internally).
 * lo_hi: DEVICE float[12]; [0..3) = min, [3..6) = max, [6..12) scratch. */
int spt_bbox_f32(const float* xyz, int64_t n, float* lo_hi, spt_stream_t stream);
int spt_grid_knn_f32(const float* query, int64_t nq, const float* search, int64_t ns,
                     int K, float r, float cell_size, const float* origin,
                     const int32_t* dims, int order_queries_by_cell, int inclusive,
                     int squared, int64_t* idx, float* dist, int32_t* cell_order, void* ws,
                     size_t ws_bytes, spt_stream_t stream);
/* Continuation of a search: for every query the K search points ranked strictly AFTER
 * (after_d2[q], after_idx[q]) in the contract's (squared distance, index) order - chained after a
 * K = 64 call it returns neighbours 65..128, which lifts the 64-per-call limit of the kernels
 * (cluster_radius_nn_graph's k_max = 100, src/utils/neighbors.py:491).  after_idx[q] < 0 = the
 * previous list was not full: nothing left (all -1).  after_d2 is SQUARED whatever `squared`. */
int spt_grid_knn_after_f32(const float* query, int64_t nq, const float* search, int64_t ns, int K,
                           float r, float cell_size, const float* origin, const int32_t* dims,
                           int inclusive, int squared, const int64_t* after_idx,
                           const float* after_d2, int64_t* idx, float* dist, void* ws,
                           size_t ws_bytes, spt_stream_t stream);

/* ------------------------------------------------------------------------
 * Point geometric features                                          (a11-a14)
 * Replaces pgeof.compute_features (src/utils/geometry.py:148-153) and the torch
 * branch the reference takes on GPU (_geometric_features_torch :236-338 +
 * scatter_pca, src/utils/scatter.py:41-125): per point the population
 * covariance of {self} + valid neighbours, its 3x3 eigen-decomposition and
 *   feats [n,11] = [linearity, planarity, scattering, verticality, nx, ny, nz,
 *                   length, surface, volume, curvature]   (geometry.py:165-174)
 * dense: nn [n,k] int64, negative = missing (FRNN padding); csr: nn_val / nn_ptr
 * [n+1] int64 (pgeof layout).  add_self: count the point itself (geometry.py:95-96);
 * features are zeroed where the neighbourhood has < k_min points; post != 0
 * applies geometric_features' tail (verticality * 2, normal flipped to z >= 0,
 * geometry.py:121,124).  order (nullable): a permutation of the points, e.g.
 * spt_spatial_order's; points are visited in that order so that the neighbourhoods a
 * wave gathers overlap (results do not depend on it).
 * spt_spatial_order: order[j] = j-th point when grouped by the cells of a uniform grid
 * (HOST cell_size / origin[3] / dims[3] as for spt_grid_knn_f32).
 * ---------------------------------------------------------------------- */
size_t spt_spatial_order_workspace_bytes(int64_t n, int64_t ncells);
int spt_spatial_order(const float* xyz, int64_t n, float cell_size, const float* origin,
                      const int32_t* dims, int32_t* order, void* ws, size_t ws_bytes,
                      spt_stream_t stream);
int spt_point_geof_dense_f32(const float* xyz, int64_t n, const int64_t* nn, int k,
                             int add_self, int k_min, int post, const int32_t* order,
                             float* feats, spt_stream_t stream);
/* The same with a row pitch: nn[i * ld + c], ld >= k - a column slice of a wider table (knn_1's
 * [N, k] result is columns 1 .. k of a [N, k + 1] search) is read in place. */
int spt_point_geof_dense_ld_f32(const float* xyz, int64_t n, const int64_t* nn, int k, int64_t ld,
                                int add_self, int k_min, int post, const int32_t* order,
                                float* feats, spt_stream_t stream);
int spt_point_geof_csr_f32(const float* xyz, int64_t n, const int64_t* nn_val,
                           const int64_t* nn_ptr, int add_self, int k_min, int post,
                           float* feats, spt_stream_t stream);
/* scatter_pca (src/utils/scatter.py:41-125, algorithm='eigh') as an entry of its own (a12):
 * for every group of the CSR view (perm nullable = rows already grouped) the population
 * covariance of x [rows, 3], its eigenvalues ascending and clamped at 0 -> eigenval [S,3],
 * eigenvectors in the COLUMNS of eigenvec [S,3,3] (torch.linalg.eigh's layout; the sign of a
 * column is arbitrary there as here); an empty group yields (1,1,1) / identity
 * (scatter.py:113-118). */
int spt_scatter_pca_f32(const float* x, const int32_t* perm, const int32_t* rowptr,
                        int64_t num_seg, float* eigenval, float* eigenvec, spt_stream_t stream);
/* neighbors_dense_to_csr (src/utils/neighbors.py:668-684)                       (a10)
 * nn [n,k] int64 with negative = missing -> ptr [n+1], val [capacity n*k, first ptr[n]
 * entries valid, row order kept], sizes [n] (all int64). */
size_t spt_neighbors_dense_to_csr_workspace_bytes(int64_t n);
int spt_neighbors_dense_to_csr(const int64_t* nn, int64_t n, int k, int64_t* ptr, int64_t* val,
                               int64_t* sizes, void* ws, size_t ws_bytes, spt_stream_t stream);

/* ------------------------------------------------------------------------
 * Per-segment random sampling without replacement                      (f2/f3)
 * Replaces sparse_sample (src/utils/sparse.py:142-243) behind NAG.get_sampling
 * (src/data/nag.py:672-711): shuffle + stable sort by segment + take the first
 * n_samples[s] of every segment, with
 *   n_samples = clamp(floor(n_max * tanh(size / n_max)), n_min, size)   (n_max > 0)
 *             = clamp(round(sqrt(size)), n_min, size)                   (n_max <= 0)
 * mask (nullable, 1 = keep): heuristic on the unmasked sizes, then clamped to the
 * number of kept elements (sparse.py:180-205).  seed: counter-based RNG, same seed
 * => same draw.  out_ptr [num_seg+1] int64 (out_ptr[num_seg] = number of samples,
 * never more than n), out_idx [n] int64 (only the first out_ptr[num_seg] are written).
 * ---------------------------------------------------------------------- */
size_t spt_sparse_sample_workspace_bytes(int64_t n, int64_t num_seg);
int spt_sparse_sample(const int64_t* idx, int64_t n, int64_t num_seg, const uint8_t* mask,
                      int n_max, int n_min, uint64_t seed, int64_t* out_ptr,
                      int64_t* out_idx, void* ws, size_t ws_bytes, spt_stream_t stream);

/* ------------------------------------------------------------------------
 * Segment statistics of SegmentFeatures                                   (f2)
 * spt_segment_std_f32: torch_scatter.scatter_std(x, idx, dim=0) (unbiased, denominator
 *   max(cnt-1, 1) + 1e-6) as called at src/transforms/graph.py:285; x [n,c], out [num_seg,c].
 * spt_segment_mean_orientation_f32: scatter_mean_orientation
 *   (src/utils/scatter.py:249-300); orientation [n,3] -> out [num_seg,3] (unit, z >= 0).
 * Rows are visited through the CSR view (perm nullable = already grouped).
 * ---------------------------------------------------------------------- */
int spt_segment_std_f32(const float* x, const int32_t* perm, const int32_t* rowptr,
                        int64_t num_seg, int c, float* out, spt_stream_t stream);
int spt_segment_mean_orientation_f32(const float* orientation, const int32_t* perm,
                                     const int32_t* rowptr, int64_t num_seg, float* out,
                                     spt_stream_t stream);

/* ------------------------------------------------------------------------
 * Radius graph between point clusters                                     (f2)
 * The two device-heavy stages of cluster_radius_nn_graph
 * (src/utils/neighbors.py:491-665).
 *
 * spt_cluster_graph_edges (:563-613): neighbours [S,k] int64 (-1 = missing) and
 *   distances [S,k] f32 of the cluster centres (knn_1's output), r_cluster [S] = diam/2;
 *   keeps `dist <= r_s + r_t + 1.732 * gap`, then to_trimmed (trim != 0: s < t,
 *   src/utils/graph.py:466-502) or coalesce, both with reduce = 'min', no self loops.
 *   edges [2, S*k] int64 (row stride S*k), edge_dist [S*k]: the first *count columns are
 *   written, sorted by (s, t); count: device int64.
 *
 * spt_cluster_pair_anchors_f32: scatter_nearest_neighbor (src/utils/scatter.py:128-238)
 *   + anchor distance (neighbors.py:631-632) for every edge (s, t): `cycles` rounds of
 *   "closest point of t to s's candidate, closest point of s to t's candidate" starting
 *   from the cluster centroids [S,3]; clusters are walked through the CSR view
 *   (perm, rowptr) of the point -> cluster index; ties -> first point in CSR order.
 *   edges: row 0 at edges[0..E), row 1 at edges[edge_stride..]; anchors [2,E] int64 point
 *   indices, d_nn [E].  group_lanes in {8,16,64}: lanes cooperating on one edge.
 * ---------------------------------------------------------------------- */
size_t spt_cluster_graph_edges_workspace_bytes(int64_t num_clusters, int k);
int spt_cluster_graph_edges(const int64_t* neighbors, const float* distances,
                            const float* r_cluster, int64_t num_clusters, int k, float gap,
                            int trim, int64_t* edges, float* edge_dist, int64_t* count,
                            void* ws, size_t ws_bytes, spt_stream_t stream);
int spt_cluster_pair_anchors_f32(const float* points, const int32_t* perm, const int32_t* rowptr,
                                 const float* centroid, const int64_t* edges, int64_t num_edges,
                                 int64_t edge_stride, int cycles, int group_lanes,
                                 int64_t* anchors, float* d_nn, spt_stream_t stream);

/* ------------------------------------------------------------------------
 * NAG selection / re-indexing                                             (f3)
 * The integer work of NAG.select (src/data/nag.py:306-399), Data.select
 * (src/data/data.py:286-470) and Cluster.select (src/data/cluster.py:79-140).  All ids
 * are bounded by a level size, so every consecutive_cluster (torch.unique) of the
 * reference is a presence bitmap + scan here.  Counts are device int64 scalars.
 *
 * spt_index_inverse: inv [n] = -1 except inv[idx[j]] = j          (data.py:365-368)
 * spt_select_edges: edges whose two ends survive, relabelled through inv, original
 *   order kept; idx_edge = their positions (data.py:369-373).  edge_index rows at
 *   [0..E) and [edge_stride..); out rows at [0..) and [out_stride..).
 * spt_cluster_select: clusters idx [k] (no duplicates) of the CSR (pointers [S+1],
 *   points [num_points], point ids < n_sub) -> new_pointers [k+1], new_points (dense
 *   new ids, first new_pointers[k] entries), idx_sub (ascending surviving point ids,
 *   first *count_sub entries) and sub_super (new cluster of each surviving point)
 *   (src/data/csr.py:328-408, cluster.py:127-138).
 * spt_relabel_consecutive: consecutive_cluster of values[gather[i]] (gather nullable)
 *   for labels in [0, n_range): new_values [k] dense labels in sorted order, uniques =
 *   the labels present, ascending (data.py:404-406: new super_index / idx_super).
 * spt_radius_ball_f32: ascending indices of the nodes within r of `center` (HOST float[3];
 *   z ignored when cylindrical) and of batch item `batch_id` (batch nullable) - the seed
 *   neighbourhood of SampleRadiusSubgraphs (src/transforms/sampling.py:1196-1230, via
 *   knn_brute_force, src/utils/neighbors.py:245-295: Euclidean norm, d <= r kept).
 * ---------------------------------------------------------------------- */
int spt_index_inverse(const int64_t* idx, int64_t k, int64_t n, int64_t* inv,
                      spt_stream_t stream);
size_t spt_select_edges_workspace_bytes(int64_t num_edges);
int spt_select_edges(const int64_t* edge_index, int64_t num_edges, int64_t edge_stride,
                     const int64_t* inv, int64_t n, int64_t* out_edges, int64_t out_stride,
                     int64_t* idx_edge, int64_t* count, void* ws, size_t ws_bytes,
                     spt_stream_t stream);
size_t spt_cluster_select_workspace_bytes(int64_t k, int64_t num_points, int64_t n_sub);
int spt_cluster_select(const int64_t* pointers, const int64_t* points, int64_t num_points,
                       const int64_t* idx, int64_t k, int64_t n_sub, int64_t* new_pointers,
                       int64_t* new_points, int64_t* idx_sub, int64_t* sub_super,
                       int64_t* count_sub, void* ws, size_t ws_bytes, spt_stream_t stream);
size_t spt_radius_ball_workspace_bytes(int64_t n);
int spt_radius_ball_f32(const float* pos, int64_t n, const float* center, float r,
                        int cylindrical, const int64_t* batch, int64_t batch_id,
                        int64_t* out_idx, int64_t* count, void* ws, size_t ws_bytes,
                        spt_stream_t stream);
size_t spt_relabel_consecutive_workspace_bytes(int64_t n_range);
int spt_relabel_consecutive(const int64_t* values, const int64_t* gather, int64_t k,
                            int64_t n_range, int64_t* new_values, int64_t* uniques,
                            int64_t* count, void* ws, size_t ws_bytes, spt_stream_t stream);

/* ------------------------------------------------------------------------
 * On-the-fly horizontal edge features + symmetrisation + self loops   (f1)
 * Replaces _on_the_fly_horizontal_edge_features (src/transforms/graph.py:1135-1277,
 * all default keys) followed by NAGAddSelfLoops (:1419-1452).
 *   se [2,e] int64 trimmed edges (row-major: sources then targets);
 *   edge_attr7 [e,7] f32 = [mean_off(3), std_off(3), mean_dist];
 *   pos, normal [n,3]; log_length / log_surface / log_volume / log_size [n];
 *   edge_index_out [2, 2e (+n)] int64 = [(s,t) | (t,s) | (i,i)];
 *   edge_attr_out [2e (+n), 18] f32 in the reference's f_list column order;
 *   self-loop rows are zero.
 * ---------------------------------------------------------------------- */
int spt_horizontal_edge_features_f32(
    const int64_t* se, int64_t e, int64_t n, const float* edge_attr7, const float* pos,
    const float* normal, const float* log_length, const float* log_surface,
    const float* log_volume, const float* log_size, int add_self_loops,
    int64_t* edge_index_out, float* edge_attr_out, spt_stream_t stream);
/* On-the-fly VERTICAL edge features (src/transforms/graph.py:1335-1416, all default keys), one
 * row per child node with parent p = super_index[i]:
 *   [centroid_dir (3, 0/0 -> 0, clipped to [-1,1]), sqrt(centroid_dist), |<n_i, n_p>|,
 *    log_length_p - log_length_i, log_surface.., log_volume.., log_size..]      out [n, 9]. */
int spt_vertical_edge_features_f32(
    const int64_t* super_index, int64_t n, const float* child_pos, const float* child_normal,
    const float* child_log_length, const float* child_log_surface, const float* child_log_volume,
    const float* child_log_size, const float* parent_pos, const float* parent_normal,
    const float* parent_log_length, const float* parent_log_surface,
    const float* parent_log_volume, const float* parent_log_size, float* v_edge_attr,
    spt_stream_t stream);
/* Symmetric edge features of the panoptic edge-affinity head (src/models/panoptic.py:477-480):
 *   out[e] = cat(|x[a_e] - x[b_e]|, (x[a_e] + x[b_e]) / 2)   x [n,C], C % 4 == 0, out [e,2C].
 * Backward: gend [2e, C], row e = gradient reaching x[a_e] through edge e, row e + E the one
 * reaching x[b_e]; the caller sums them per node with spt_segcsr_reduce_f32 over cat(a, b). */
int spt_edge_affinity_features_f32(const float* x, int64_t n, int C, const int64_t* edge_a,
                                   const int64_t* edge_b, int64_t e, float* out,
                                   spt_stream_t stream);
int spt_edge_affinity_features_bwd_f32(const float* x, const float* gout, int64_t n, int C,
                                       const int64_t* edge_a, const int64_t* edge_b, int64_t e,
                                       float* gend, spt_stream_t stream);

/* ------------------------------------------------------------------------
 * Tall-skinny Linear: y [rows, N] = x [rows, K] W[N, K]^T (+ bias [N] or NULL), f32 in /
 * f32 accumulate.  The qkv / out_proj nn.Linear of SelfAttentionBlock
 * (src/nn/attention.py:202-215, 311-313) and, on (gy, W^T), their input gradients.
 * K in {32, 64, 128, 192} and - round 5 - 256 (the dX of the 128-wide blocks' qkv Linear) and
 * 132 / 260: the first Linear of the KITTI-360 width's
 * node MLPs ([4 + 128, 128, 128] / [4 + 256, 128, 128], configs/experiment/semantic/kitti360.yaml:
 * 22-27, src/nn/mlp.py:43-56), which ran on the vendor GEMM before.  N any width from 64 up (a last
 * slab of fewer than 64 columns - the dX of a 132 / 260-wide input - reads the missing weight rows
 * as zero and stores nothing for them), or N <= 16 (the heads); x, W, y contiguous.
 * ---------------------------------------------------------------------- */
int spt_skinny_linear_supported(int K, int N);
int spt_skinny_linear_f32(const float* x, int64_t rows, int K, const float* W, const float* bias,
                          int N, float* y, spt_stream_t stream);
/* Round 6: y [rows, N] = x [rows, K] Wt [K, N] - the same product with the weight given as the
 * TRANSPOSE of what spt_skinny_linear_f32 takes: the input gradient dX = G W of a Linear reads the
 * layer's own weight W [N_out = K, N_in = N] (what autograd of src/nn/attention.py:202-215 /
 * src/nn/mlp.py:43-56 computes) instead of a transposed copy made per backward call.  Same K as
 * above, N % 4 == 0 and N >= 64, Wt 16-byte aligned; no bias. */
int spt_skinny_linear_wt_f32(const float* x, int64_t rows, int K, const float* Wt, int N, float* y,
                             spt_stream_t stream);
/* Weight gradient of the same Linear: gw[N,K] = gy[rows,N]^T x[rows,K] (src/nn/attention.py's qkv /
 * out_proj under autograd), a reduction over 10^5..10^7 rows.  K in {32, 64, 128, 132, 260} (above
 * 64: 64-column slabs of x, the last one narrower), N a multiple of 64;
 * f32 in / f32 accumulate on the matrix pipe, per-wave partials summed in a fixed order
 * (deterministic).  gb[N] (nullable) receives the column sums of gy - the bias gradient - from the
 * same pass.  ws: spt_skinny_dw_workspace_bytes(K, N) bytes of device scratch. */
/* The transformer block's pre-norm and residual folded into its two Linears
 * (x = x + out_proj(attention(qkv(norm(x)))), src/nn/transformer.py:231-234):
 *   pre_am / pre_scale [num_graphs, K], pre_bias [K]: x_n = (x - pre_am[g]) * pre_scale[g] +
 *     pre_bias with g = batch[row] (batch NULL = one graph) applied while the rows are read - the
 *     tables of spt_graphnorm_stats_f32; pre_am NULL = plain x;
 *   residual [rows, N] or NULL: y = (x W^T + b) + residual.
 * K in {32, 64, 128}, N a multiple of 64, num_graphs * K <= 1024 (spt_skinny_pre_supported). */
int spt_skinny_pre_supported(int K, int N, int num_graphs);
int spt_skinny_linear_pre_f32(const float* x, int64_t rows, int K, const float* W, const float* bias,
                              int N, float* y, const float* pre_am, const float* pre_scale,
                              const float* pre_bias, const int64_t* batch, int num_graphs,
                              const float* residual, spt_stream_t stream);
/* gw = gy^T norm(x) (+ gb): the weight gradient of the Linear above (same tables). */
int spt_skinny_dw_pre_f32(const float* gy, const float* x, int64_t rows, int N, int K, float* gw,
                          float* gb, const float* pre_am, const float* pre_scale,
                          const float* pre_bias, const int64_t* batch, int num_graphs, void* ws,
                          size_t ws_bytes, spt_stream_t stream);
/* Round 6: matrix mode of the tall-skinny Linears (K = 32 / 64 and the K = 192 input gradient; the
 * other widths stay on the f32 pipe in every mode).  They ran on v_mfma_f32_16x16x4_f32 in every
 * mode; like the attention and the fused layers they now follow the mode the reference ships
 * (configs/train.yaml:60-61 float32_matmul_precision: high):
 *   0  f32 pipe (the f32-exact mode)
 *   1  (default) split bf16: the forward as the f32-EXACT six-product 3-way split, the input and
 *      weight gradients as hi*hi + lo*hi + hi*lo (three products)
 *   3  operands rounded to bf16 (the bf16 mode)
 * spt_skinny_use_split_bf16 sets the process-wide default and returns the previous one (< 0: query);
 * the _m entries take the mode per call (< 0: that default).  _pre_m = spt_skinny_linear_pre_f32 (a
 * forward: pre-norm / residual options), _wt_m = spt_skinny_linear_wt_f32 (an input gradient),
 * _dw_pre_m = spt_skinny_dw_pre_f32 (a weight gradient). */
int spt_skinny_use_split_bf16(int mode);
int spt_skinny_linear_pre_m_f32(const float* x, int64_t rows, int K, const float* W, const float* bias,
                                int N, float* y, const float* pre_am, const float* pre_scale,
                                const float* pre_bias, const int64_t* batch, int num_graphs,
                                const float* residual, int mode, spt_stream_t stream);
int spt_skinny_linear_wt_m_f32(const float* x, int64_t rows, int K, const float* Wt, int N, float* y,
                               int mode, spt_stream_t stream);
int spt_skinny_dw_pre_m_f32(const float* gy, const float* x, int64_t rows, int N, int K, float* gw,
                            float* gb, const float* pre_am, const float* pre_scale,
                            const float* pre_bias, const int64_t* batch, int num_graphs, int mode,
                            void* ws, size_t ws_bytes, spt_stream_t stream);
int spt_skinny_dw_supported(int K, int N);
size_t spt_skinny_dw_workspace_bytes(int K, int N);
int spt_skinny_dw_f32(const float* gy, const float* x, int64_t rows, int N, int K, float* gw,
                      float* gb, void* ws, size_t ws_bytes, spt_stream_t stream);
/* Narrow Linears (N <= 16 outputs: the classifier heads, src/nn/mlp.py:128-142).  Forward: the
 * skinny kernel above accepts N <= 16 with K in {32, 64, 128} (missing columns read as zero, not
 * stored).  Backward in ONE pass over x and gy for K = 64 and (round 5: the KITTI-360 width's heads,
 * two 64-column slabs) K = 128: gx[rows,K] = gy W (nullable),
 * gw[N,K] = gy^T x, gb[N] = column sums of gy (nullable); per-wave partials, fixed-order sum. */
int spt_narrow_linear_bwd_supported(int K, int N);
size_t spt_narrow_linear_bwd_workspace_bytes(int K, int N);
int spt_narrow_linear_bwd_f32(const float* gy, const float* x, const float* W, int64_t rows, int N,
                              int K, float* gx, float* gw, float* gb, void* ws, size_t ws_bytes,
                              spt_stream_t stream);

/* Pieces of the GraphNorm two-pass scheme for callers that produce / consume the
 * per-graph totals themselves (the fused MLP layers below).  totals layout:
 * [num_graphs][2d+1] f64 = (column sums, column sums of the second quantity, row count). */
int spt_graphnorm_tables_f32(const double* total, int num_graphs, int d, const float* weight,
                             const float* mean_scale, float eps, float* mean, float* rstd,
                             float* am, float* scale, spt_stream_t stream);
int spt_graphnorm_apply_f32(const float* x, const int64_t* batch, int64_t r, int d,
                            int num_graphs, const float* am, const float* scale,
                            const float* bias, float act_slope, float* y, spt_stream_t stream);
int spt_graphnorm_bwd_stats_f32(const float* x, const float* gy, const int64_t* batch, int64_t r,
                                int d, int num_graphs, const float* am, const float* scale,
                                const float* bias, float act_slope, double* total, void* ws,
                                size_t ws_bytes, spt_stream_t stream);
/* The totals of spt_graphnorm_bwd_stats_f32 when gy is the backward of a segment max-pool
 * (one non-zero per (segment, channel): gy[arg[s,c], c] = gout[s,c]), computed from
 * (gout, arg [num_seg, d], the raw rows x gathered at arg) without a pass over [r, d] tensors.
 * seg_graph [num_seg] (NULL = one graph), graph_rows [B] int64 = rows of every graph. */
size_t spt_graphnorm_bwd_stats_sparse_workspace_bytes(int64_t num_seg, int d, int num_graphs);
int spt_graphnorm_bwd_stats_sparse_f32(const float* x, const float* gout, const int32_t* arg,
                                       const int64_t* seg_graph, const int64_t* graph_rows,
                                       int64_t num_seg, int64_t n, int d, int num_graphs,
                                       const float* am, const float* scale, const float* bias,
                                       float act_slope, double* total, void* ws,
                                       size_t ws_bytes, spt_stream_t stream);
/* x_is_bf16 = 1: x holds bf16 values (SPT_FMLP_H_BF16 below); 2: x is raw [num_seg, d] f32. */
int spt_graphnorm_bwd_stats_sparse_ex_f32(
    const void* x, int x_is_bf16, const float* gout, const int32_t* arg, const int64_t* seg_graph,
    const int64_t* graph_rows, int64_t num_seg, int64_t n, int d, int num_graphs,
    const float* am, const float* scale, const float* bias, float act_slope, double* total,
    void* ws, size_t ws_bytes, spt_stream_t stream);
/* The same totals from raw [num_seg, d] = the layer output at the arg rows as the pool wrote it
 * (spt_segcsr_max_affine_raw_f32): a stream, no gather.  n = rows of the layer (arg's sentinel). */
int spt_graphnorm_bwd_stats_sparse_raw_f32(
    const float* raw, const float* gout, const int32_t* arg, const int64_t* seg_graph,
    const int64_t* graph_rows, int64_t num_seg, int64_t n, int d, int num_graphs,
    const float* am, const float* scale, const float* bias, float act_slope, double* total,
    void* ws, size_t ws_bytes, spt_stream_t stream);
int spt_graphnorm_bwd_tables_f32(const double* total, int num_graphs, int d, const float* weight,
                                 const float* mean_scale, const float* mean, const float* rstd,
                                 float* c1, float* c2, float* c3, float* gweight, float* gbias,
                                 float* gmean_scale, spt_stream_t stream);

/* ------------------------------------------------------------------------
 * Fused MLP layer: bias-free Linear whose input is the previous layer's RAW output
 * (GraphNorm-apply + LeakyReLU happen while the tile is staged) and whose epilogue
 * yields the statistics of its own GraphNorm.  Replaces, per layer of
 * src/nn/mlp.py:8-94, one library GEMM + GraphNorm statistics pass + GraphNorm apply
 * pass (forward), and GraphNorm backward apply + the dX / dW GEMMs (backward).
 * One call covers the rows [r0, r1) of ONE graph of the batch.
 *   x / xprev [rows, K] raw, W [N, K], h [rows, N] raw output; pre_* = (am, scale
 *   [K], bias [K], slope) of the previous GraphNorm or NULL (layer input used as is);
 *   total [2N+1] f64 out; backward: gy, h [rows, N], this layer's (am, scale, bias,
 *   slope) and backward rows (c1, c2, c3); gx [rows, K] or NULL; gW [N, K] (+= when
 *   accumulate); prev_total [2K+1] = statistics for the previous GraphNorm's backward.
 * ---------------------------------------------------------------------- */
int spt_fused_linear_supported(int K, int N);
/* Matrix pipe of the fused layers' GEMMs.  0: f32 in / f32 accumulate everywhere (bitwise an
 * fmaf chain).  1 (default): the backward GEMMs (gW, gx) on the bf16 pipe with split operands
 * (each f32 product = hi*hi + lo*hi + hi*lo of bf16 halves, f32 accumulate, ~2^-17 relative per product: 17 of f32's 24 bits -
 * a tenth of the gradients' parity bar); the forward stays exact f32 (its outputs feed
 * GraphNorm statistics, bar 2e-5).  2: forward too.  Returns the previous setting. */
int spt_fused_linear_use_split_bf16(int mode);
size_t spt_fused_linear_workspace_bytes(int K, int N);
int spt_fused_linear_fwd_f32(const float* x, int64_t r0, int64_t r1, int K, const float* W,
                             int N, const float* pre_am, const float* pre_scale,
                             const float* pre_bias, float pre_slope, float* h, double* total,
                             void* ws, size_t ws_bytes, spt_stream_t stream);
int spt_fused_linear_bwd_f32(const float* gy, const float* h, int64_t r0, int64_t r1, int N,
                             const float* am, const float* scale, const float* bias,
                             float slope, const float* c1, const float* c2, const float* c3,
                             const float* xprev, int K, const float* pre_am,
                             const float* pre_scale, const float* pre_bias, float pre_slope,
                             const float* W, float* gx, float* gW, int accumulate,
                             double* prev_total, void* ws, size_t ws_bytes, spt_stream_t stream);
/* Backward of the TOP layer of a fused MLP whose output went through a segment max-pool
 * (MaxPool(MLP(x)), nn/stage.py:429-431): the pool's gradient (gout [S,N], arg [S,N]) is consumed
 * directly - rows are visited in the pool's CSR order (positions [p0, p1) of perm; pos_seg[j] =
 * segment of position j), h / xprev rows gathered, gx rows scattered - so the dense [rows, N]
 * gradient the pool's backward would write (and this layer read back) never exists.  Other
 * arguments as spt_fused_linear_bwd_f32.  spt_fused_linear_pooled_supported: built shapes, under
 * the current matrix mode (split-bf16 kernels only). */
int spt_fused_linear_pooled_supported(int K, int N);
/* Per-call matrix mode (no process-wide state): the *_ex entries take `mode` = 0..3 as
 * spt_fused_linear_use_split_bf16 describes them (< 0: the process default), so that two models
 * at different precisions, or two streams, never change each other's arithmetic. */
int spt_fused_linear_pooled_supported_ex(int K, int N, int mode);
/* Tile staging of the split-bf16 / bf16 backward kernels: LDS-DMA (global_load_lds: one memory
 * round trip per 16-row tile, csrc/fused_mlp_dma.hip; default, where the (K, N) is built: 64->128,
 * 64->64, 32->64) or register-staged (csrc/fused_mlp.hip).  Same arithmetic, same summation order
 * inside a tile.  Process-wide: spt_fused_linear_bwd_use_dma(0 | 1) (< 0: query), returns the
 * previous setting; per call: OR SPT_FMLP_BWD_REGISTER_STAGED into the `mode` of the *_ex entries. */
#define SPT_FMLP_BWD_REGISTER_STAGED 4
/* bf16 ACTIVATION STORAGE (the reference's `precision: bf16`, configs/trainer/gpu.yaml:7-10: under
 * autocast the Linear outputs ARE bf16 tensors).  With matrix mode 3 in the mode word of the *_ex /
 * *_runs entries: SPT_FMLP_H_BF16 = the layer's raw output h is written (forward) / read (backward)
 * as bf16 [rows, N]; SPT_FMLP_X_BF16 = the layer's input x / xprev holds bf16 values (the previous
 * layer's h).  Statistics, gradients and tables stay f32 / f64.  spt_fused_linear_storage_supported:
 * shapes built with this option (the point MLP's). */
#define SPT_FMLP_H_BF16 8
#define SPT_FMLP_X_BF16 16
int spt_fused_linear_storage_supported(int K, int N);
int spt_fused_linear_bwd_use_dma(int on);
/* Forward of matrix mode 1 (the default "f32"): 1 (default) = the f32 product as SIX bf16 products
 * of 3-way split operands on the bf16 matrix pipe (f32-exact: every dropped term is below 2^-24 of
 * its product), 0 = the f32 matrix pipe.  Process-wide; < 0 queries; returns the previous setting. */
int spt_fused_linear_fwd_use_x3(int on);
int spt_fused_linear_fwd_ex_f32(const float* x, int64_t r0, int64_t r1, int K, const float* W,
                                int N, const float* pre_am, const float* pre_scale,
                                const float* pre_bias, float pre_slope, float* h, double* total,
                                int mode, void* ws, size_t ws_bytes, spt_stream_t stream);
int spt_fused_linear_bwd_ex_f32(const float* gy, const float* h, int64_t r0, int64_t r1, int N,
                                const float* am, const float* scale, const float* bias,
                                float slope, const float* c1, const float* c2, const float* c3,
                                const float* xprev, int K, const float* pre_am,
                                const float* pre_scale, const float* pre_bias, float pre_slope,
                                const float* W, float* gx, float* gW, int accumulate,
                                double* prev_total, int mode, void* ws, size_t ws_bytes,
                                spt_stream_t stream);
int spt_fused_linear_bwd_pooled_ex_f32(
    const float* gout, const int32_t* arg, const int32_t* perm, const int32_t* pos_seg,
    const float* h, int64_t p0, int64_t p1, int N, const float* am, const float* scale,
    const float* bias, float slope, const float* c1, const float* c2, const float* c3,
    const float* xprev, int K, const float* pre_am, const float* pre_scale, const float* pre_bias,
    float pre_slope, const float* W, float* gx, float* gW, int accumulate, double* prev_total,
    int mode, void* ws, size_t ws_bytes, spt_stream_t stream);
int spt_fused_linear_bwd_pooled_f32(
    const float* gout, const int32_t* arg, const int32_t* perm, const int32_t* pos_seg,
    const float* h, int64_t p0, int64_t p1, int N, const float* am, const float* scale,
    const float* bias, float slope, const float* c1, const float* c2, const float* c3,
    const float* xprev, int K, const float* pre_am, const float* pre_scale, const float* pre_bias,
    float pre_slope, const float* W, float* gx, float* gW, int accumulate, double* prev_total,
    void* ws, size_t ws_bytes, spt_stream_t stream);

/* The rows of SEVERAL graphs in ONE launch of the fused layer kernels (a batch of B clouds, or an
 * index that is sorted only piecewise - the edge MLP's norm index norm_index[edge_index[0]],
 * src/models/components/spt.py:829-836, is sorted inside each third of [i<j | j>i | loops]).
 * Run r covers rows [run_r0[r], run_r1[r]) of graph run_graph[r]; HOST arrays, 1 <= nruns <= 16,
 * sorted by graph, num_graphs <= 16.  Per-graph tables are [num_graphs, width] arrays, `total` /
 * `prev_total` [num_graphs, 2 width + 1] (a graph without rows reads as zeros), gW = the sum over
 * all runs.  Otherwise as the one-range entries above. */
int spt_fused_linear_fwd_runs_f32(const float* x, int nruns, const int64_t* run_r0,
                                  const int64_t* run_r1, const int32_t* run_graph, int num_graphs,
                                  int K, const float* W, int N, const float* pre_am,
                                  const float* pre_scale, const float* pre_bias, float pre_slope,
                                  float* h, double* total, int mode, void* ws, size_t ws_bytes,
                                  spt_stream_t stream);
int spt_fused_linear_bwd_runs_f32(
    const float* gy, const float* h, int nruns, const int64_t* run_r0, const int64_t* run_r1,
    const int32_t* run_graph, int num_graphs, int N, const float* am, const float* scale,
    const float* bias, float slope, const float* c1, const float* c2, const float* c3,
    const float* xprev, int K, const float* pre_am, const float* pre_scale, const float* pre_bias,
    float pre_slope, const float* W, float* gx, float* gW, double* prev_total, int mode, void* ws,
    size_t ws_bytes, spt_stream_t stream);
int spt_fused_linear_bwd_pooled_runs_f32(
    const float* gout, const int32_t* arg, const int32_t* perm, const int32_t* pos_seg,
    const float* h, int nruns, const int64_t* run_p0, const int64_t* run_p1,
    const int32_t* run_graph, int num_graphs, int N, const float* am, const float* scale,
    const float* bias, float slope, const float* c1, const float* c2, const float* c3,
    const float* xprev, int K, const float* pre_am, const float* pre_scale, const float* pre_bias,
    float pre_slope, const float* W, float* gx, float* gW, double* prev_total, int mode, void* ws,
    size_t ws_bytes, spt_stream_t stream);
/* Round 6: the same three calls with a GraphNorm descriptor - the layer's own norm (forward: its
 * tables mean / rstd / am / scale come out of the call, spt_graphnorm_tables_f32 is not needed;
 * `total` may be NULL) or the PREVIOUS layer's norm (backward: its c1 / c2 / c3 and parameter
 * gradients come out of the call, spt_graphnorm_bwd_tables_f32 is not needed; `prev_total` may be
 * NULL).  One "post" launch per call sums the partial tables and writes the norm's tables: a fused
 * layer costs 2 launches per direction instead of 3 (forward) / 4 (backward) - what a train-batch
 * step, ~360 launches of a few microseconds, is made of. */
int spt_fused_linear_fwd_runs_gn_f32(const float* x, int nruns, const int64_t* run_r0,
                                     const int64_t* run_r1, const int32_t* run_graph, int num_graphs,
                                     int K, const float* W, int N, const float* pre_am,
                                     const float* pre_scale, const float* pre_bias, float pre_slope,
                                     float* h, double* total, int mode, void* ws, size_t ws_bytes,
                                     const spt_gn_fwd_tables* norm, spt_stream_t stream);
int spt_fused_linear_bwd_runs_gn_f32(
    const float* gy, const float* h, int nruns, const int64_t* run_r0, const int64_t* run_r1,
    const int32_t* run_graph, int num_graphs, int N, const float* am, const float* scale,
    const float* bias, float slope, const float* c1, const float* c2, const float* c3,
    const float* xprev, int K, const float* pre_am, const float* pre_scale, const float* pre_bias,
    float pre_slope, const float* W, float* gx, float* gW, double* prev_total, int mode, void* ws,
    size_t ws_bytes, const spt_gn_bwd_tables* prev_norm, spt_stream_t stream);
int spt_fused_linear_bwd_pooled_runs_gn_f32(
    const float* gout, const int32_t* arg, const int32_t* perm, const int32_t* pos_seg,
    const float* h, int nruns, const int64_t* run_p0, const int64_t* run_p1,
    const int32_t* run_graph, int num_graphs, int N, const float* am, const float* scale,
    const float* bias, float slope, const float* c1, const float* c2, const float* c3,
    const float* xprev, int K, const float* pre_am, const float* pre_scale, const float* pre_bias,
    float pre_slope, const float* W, float* gx, float* gW, double* prev_total, int mode, void* ws,
    size_t ws_bytes, const spt_gn_bwd_tables* prev_norm, spt_stream_t stream);

/* Consistency check of a stored segment CSR (pointers [num_seg + 1], points [n], int64 as the
 * reference's Cluster keeps them, src/data/cluster.py:19-77) against the index it is to be adopted
 * as the view of (idx [n] int64): ORs into *flag (int32, device; the caller clears it) bit 0 =
 * pointers not 0 .. n / decreasing, bit 1 = a point outside [0, n), bit 2 = idx[points[j]] is not
 * the segment holding position j, bit 3 (check_ascending) = a segment's points not ascending.
 * One launch, no host round trip. */
int spt_csr_check_i64(const int64_t* idx, const int64_t* points, const int64_t* pointers, int64_t n,
                      int64_t num_seg, int check_ascending, int32_t* flag, spt_stream_t stream);
/* Round 6: the same check AND the int32 view the segment kernels read, in one launch - what
 * csr.adopt_csr runs when it installs nag[i+1].sub as the view of nag[i].super_index
 * (src/data/cluster.py:19-77, src/data/nag.py:306-399).  perm32 [n] = points clamped into [0, n),
 * rowptr32 [num_seg + 1] = pointers clamped into [0, n]: a stored CSR that fails the check cannot
 * lead a segment kernel outside its buffers before the (deferred) verdict has been read.  The
 * ascending comparison (bit 3) always runs.  *flag as above (the caller clears it). */
int spt_csr_adopt_i64(const int64_t* idx, const int64_t* points, const int64_t* pointers, int64_t n,
                      int64_t num_seg, int32_t* perm32, int32_t* rowptr32, int32_t* flag,
                      spt_stream_t stream);

/* ------------------------------------------------------------------------
 * The point MLP's TOP layer and the max-pool behind it as one unit     (a1 + a2 + a5, round 5)
 * Replaces, for the layer whose output feeds only the level-0 -> level-1 max-pool
 * (src/nn/stage.py:413-431: PointStage MLP -> pool; layer = bias-free Linear -> GraphNorm ->
 * LeakyReLU, src/nn/mlp.py:43-56; pool = MaxPool, src/nn/pool.py:61-82), the sequence
 * spt_fused_linear_fwd_* -> spt_graphnorm_tables_f32 -> spt_segcsr_max_affine_* (forward) and
 * spt_fused_linear_bwd_pooled_* (backward): the layer's [rows, N] output is never written, read
 * or recomputed (csrc/fused_pool.hip explains the three identities used).
 *   forward : out [num_seg, N] = max over the segment's rows of leaky(GraphNorm(x_act W^T)), bitwise
 *             the value of norm-then-pool evaluated with THESE statistics; arg [num_seg, N] int32 =
 *             first row attaining the raw extremum (n_rows for an empty segment, like
 *             spt_segcsr_reduce_f32; the reference's arg except on ties of y between rows of
 *             different h); argpos [num_seg, N] int32 = the CSR position of that row (arg =
 *             perm[argpos]: what the backward scatters by; arg may be NULL (round 6): a caller that only
 *             runs the fused backward needs the positions alone, and the row ids cost one scattered
 *             4-byte read per (segment, channel)); raw [num_seg, N] = h of the arg row; gram [num_graphs, K K + K + 1] f64
 *             = (sum_i y_i y_i^T | sum_i y_i | rows) of the layer's activated INPUT per graph, from
 *             which the norm's statistics are evaluated (total [num_graphs, 2N+1], nullable, receives
 *             them in spt_fused_linear_fwd_*'s layout) and its tables mean / rstd / am / scale
 *             [num_graphs, N] are written (spt_graphnorm_tables_f32's formulas).
 *             x [n_rows, K]: RAW output of the previous layer (f32, or bf16 with SPT_FMLP_X_BF16 in
 *             `mode`), pre_* its norm tables ([num_graphs, K], [num_graphs, K], [K]) and slope;
 *             (perm, pos_seg, rowptr): the pool's CSR view and the segment of every CSR position;
 *             runs: the graphs' CSR position ranges (contiguous, tiling [0, n_rows)); seg_graph
 *             [num_seg] int64 (NULL with one graph).
 *   backward: from gout, raw, argpos [num_seg, N] and the GraphNorm-backward coefficient rows c1 / c2 / c3
 *             (spt_graphnorm_bwd_stats_sparse_raw_f32 -> spt_graphnorm_bwd_tables_f32) to gx
 *             [n_rows, K] (gradient of the previous layer's normalised output), gW [N, K] and
 *             prev_total [num_graphs, 2K+1] (statistics of the previous norm's backward);
 *             gm [num_seg, N] f32: scratch of the call.
 *   mode    : the fused layers' mode word; built for matrix modes 1 (f32-exact 3-way split forward),
 *             2 and 3, K in {32, 64}, N in {64, 128} (spt_fused_linear_pool_supported).
 *   ws      : spt_fused_linear_pool_workspace_bytes(K, N). */
int spt_fused_linear_pool_supported(int K, int N, int mode);
size_t spt_fused_linear_pool_gram_len(int K);
size_t spt_fused_linear_pool_workspace_bytes(int K, int N);
int spt_fused_linear_fwd_pool_runs_f32(
    const void* x, const int32_t* perm, const int32_t* pos_seg, const int32_t* rowptr,
    const int64_t* seg_graph, int64_t num_seg, int64_t n_rows, int nruns, const int64_t* run_p0,
    const int64_t* run_p1, const int32_t* run_graph, int num_graphs, int K, const float* W, int N,
    const float* gn_weight, const float* gn_bias, const float* gn_mean_scale, float eps, float slope,
    const float* pre_am, const float* pre_scale, const float* pre_bias, float pre_slope, float* out,
    int32_t* arg, int32_t* argpos, float* raw, double* gram, double* total, float* mean, float* rstd,
    float* am, float* scale, int mode, void* ws, size_t ws_bytes, spt_stream_t stream);
int spt_fused_linear_bwd_pool_runs_f32(
    const float* gout, const float* raw, const int32_t* argpos, const int32_t* perm,
    const int32_t* pos_seg, const int64_t* seg_graph, int64_t num_seg, int nruns,
    const int64_t* run_p0, const int64_t* run_p1, const int32_t* run_graph, int num_graphs, int N,
    const float* am, const float* scale, const float* bias, float slope, const float* c1,
    const float* c2, const float* c3, const void* xprev, int K, const float* pre_am,
    const float* pre_scale, const float* pre_bias, float pre_slope, const float* W,
    const double* gram, float* gm, float* gx, float* gW, double* prev_total, int mode, void* ws,
    size_t ws_bytes, spt_stream_t stream);
/* ... with the previous layer's GraphNorm-backward tables written by the call (round 6, see
 * spt_fused_linear_bwd_runs_gn_f32; prev_total may be NULL). */
int spt_fused_linear_bwd_pool_runs_gn_f32(
    const float* gout, const float* raw, const int32_t* argpos, const int32_t* perm,
    const int32_t* pos_seg, const int64_t* seg_graph, int64_t num_seg, int nruns,
    const int64_t* run_p0, const int64_t* run_p1, const int32_t* run_graph, int num_graphs, int N,
    const float* am, const float* scale, const float* bias, float slope, const float* c1,
    const float* c2, const float* c3, const void* xprev, int K, const float* pre_am,
    const float* pre_scale, const float* pre_bias, float pre_slope, const float* W,
    const double* gram, float* gm, float* gx, float* gW, double* prev_total, int mode, void* ws,
    size_t ws_bytes, const spt_gn_bwd_tables* prev_norm, spt_stream_t stream);

/* ------------------------------------------------------------------------
 * Cross-entropy of the classifier heads' logits                (train step)
 * torch.nn.CrossEntropyLoss(ignore_index=...) of configs/model/semantic/default.yaml:47-49 as
 * src/models/semantic.py applies it per output level: mean over the rows whose target is not
 * ignore_index of logsumexp(logits[row]) - logits[row, target[row]].  C <= 32 classes.
 *   fwd: lse[rows] (kept for the backward), loss[1], count[1] (rows that counted, as f32);
 *        ws: spt_cross_entropy_workspace_bytes(rows).  Deterministic (f64 partial sums, fixed order).
 *        A target outside [0, C) other than ignore_index makes the loss NaN (torch: device assert).
 *   bwd: glogits = (softmax - onehot) * gout[0] / count[0]; gout / count are device scalars. */
size_t spt_cross_entropy_workspace_bytes(int64_t rows);
int spt_cross_entropy_fwd_f32(const float* logits, const int64_t* target, int64_t rows, int C,
                              int64_t ignore_index, float* lse, float* loss, float* count,
                              void* ws, size_t ws_bytes, spt_stream_t stream);
int spt_cross_entropy_bwd_f32(const float* logits, const int64_t* target, const float* lse,
                              int64_t rows, int C, int64_t ignore_index, const float* gout,
                              const float* count, float* glogits, spt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SPT_HIP_H */
