// Radius graph between point clusters.
//
// Replaces the two device-heavy stages of cluster_radius_nn_graph
// (src/utils/neighbors.py:491-665):
//
//  * spt_cluster_graph_edges  - lines 563-613: the [S, k] neighbour table of the cluster
//    centres -> candidate edges, radius filter `dist <= r_s + r_t + 1.732 gap`, -1
//    removal, to_trimmed (src/utils/graph.py:466-502: flip to s < t, coalesce with
//    reduce='min', drop self loops) or plain coalesce.  The reference runs ~10 boolean
//    indexing / vstack passes and a torch_geometric coalesce (sort + scatter); here one
//    key-generation kernel, two stable radix sorts (by t, then by s), a run-head flag
//    kernel, a scan and one compaction.
//
//  * spt_cluster_pair_anchors_f32 - scatter_nearest_neighbor (src/utils/scatter.py:128-238)
//    + the anchor distance (neighbors.py:631-632).  The reference expands EVERY edge into
//    the concatenation of all points of both clusters (edge_wise_points, src/utils/edge.py:
//    22-77: E x mean cluster size rows, chunked to survive) and runs 2 x cycles scatter_min
//    over it.  Here a lane group owns an edge and walks the two clusters through the CSR
//    view of the point->cluster index; nothing is materialised.
#include "radix_sort.hpp"

namespace spt {
namespace cgraph {

__global__ void candidate_keys_kernel(const int64_t* __restrict__ nn,
                                      const float* __restrict__ dist,
                                      const float* __restrict__ r_seg, int64_t S, int k,
                                      float gap_term, int trim, uint32_t* __restrict__ lo,
                                      uint32_t* __restrict__ hi) {
  const int64_t m = S * k;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < m; p += stride) {
    const int64_t i = p / k;
    const int64_t j = nn[p];
    bool ok = j >= 0 && j < S && j != i;
    if (ok) ok = dist[p] <= (r_seg[i] + r_seg[j]) + gap_term;        // neighbors.py:591-593
    uint32_t a = (uint32_t)i, b = (uint32_t)(ok ? j : 0);
    if (trim && a > b) { const uint32_t t = a; a = b; b = t; }        // graph.py:486-487
    lo[p] = ok ? a : (uint32_t)S;                                     // rejected -> tail bucket
    hi[p] = ok ? b : 0u;
  }
}

__global__ void gather_u32_kernel(const uint32_t* __restrict__ src,
                                  const uint32_t* __restrict__ order, int64_t m,
                                  uint32_t* __restrict__ dst) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < m; p += stride)
    dst[p] = src[order[p]];
}

// flag[p] = 1 where sorted position p starts a run of equal (lo, hi) and is not rejected
__global__ void run_heads_kernel(const uint32_t* __restrict__ lo, const uint32_t* __restrict__ hi,
                                 const uint32_t* __restrict__ order, int64_t m, int64_t S,
                                 uint32_t* __restrict__ flag) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p <= m; p += stride) {
    if (p == m) { flag[p] = 0; continue; }                            // slot for the total
    const uint32_t v = order[p];
    const uint32_t a = lo[v], b = hi[v];
    bool head = (int64_t)a < S;
    if (head && p > 0) {
      const uint32_t u = order[p - 1];
      head = lo[u] != a || hi[u] != b;
    }
    flag[p] = head ? 1u : 0u;
  }
}

__global__ void emit_edges_kernel(const uint32_t* __restrict__ lo, const uint32_t* __restrict__ hi,
                                  const uint32_t* __restrict__ order,
                                  const uint32_t* __restrict__ pos, const float* __restrict__ dist,
                                  int64_t m, int64_t S, int64_t cap, int64_t* __restrict__ edges,
                                  float* __restrict__ edge_dist, int64_t* __restrict__ count) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < m; p += stride) {
    if (p == 0) *count = (int64_t)pos[m];
    if (pos[p + 1] == pos[p]) continue;                               // not a run head
    const uint32_t v = order[p];
    const uint32_t a = lo[v], b = hi[v];
    float d = dist[v];
    for (int64_t q = p + 1; q < m; ++q) {                             // reduce='min' over the run
      const uint32_t u = order[q];
      if (lo[u] != a || hi[u] != b) break;
      d = fminf(d, dist[u]);
    }
    const int64_t e = pos[p];
    edges[e] = (int64_t)a;
    edges[cap + e] = (int64_t)b;
    edge_dist[e] = d;
  }
}

struct Plan {
  size_t off_lo, off_hi, off_key, off_flag, off_sort, total;
};

static Plan make_plan(int64_t m) {
  Plan p;
  size_t o = 0;
  const int64_t mm = m > 0 ? m : 1;
  p.off_lo = o;   o += align_up((size_t)mm * 4, 256);
  p.off_hi = o;   o += align_up((size_t)mm * 4, 256);
  p.off_key = o;  o += align_up((size_t)mm * 4, 256);
  p.off_flag = o; o += align_up((size_t)(mm + 1) * 4, 256);
  p.off_sort = o; o += RadixScratch::bytes(m + 1);
  p.total = o;
  return p;
}

// ---- nearest anchors ---------------------------------------------------------------
__device__ __forceinline__ uint32_t ordered_bits(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// closest point of cluster `c` to `y`; ties -> first in CSR (= ascending original) order.
// LG lanes of a group cooperate; every lane returns the winner's point index.
template <int LG>
__device__ __forceinline__ int64_t closest_in_cluster(const float* __restrict__ points,
                                                      const int32_t* __restrict__ perm,
                                                      const int32_t* __restrict__ rowptr,
                                                      int64_t c, const float y[3], int gl) {
  const int lo = rowptr[c], hi = rowptr[c + 1];
  uint64_t best = ~0ull;
  for (int j = lo + gl; j < hi; j += LG) {
    const float* p = points + (int64_t)perm[j] * 3;
    const float dx = p[0] - y[0], dy = p[1] - y[1], dz = p[2] - y[2];
    const float d = sqrtf(dx * dx + dy * dy + dz * dz);               // (X - Y).norm(dim=1)
    const uint64_t key = ((uint64_t)ordered_bits(d) << 32) | (uint32_t)(j - lo);
    best = key < best ? key : best;
  }
#pragma unroll
  for (int o = LG / 2; o > 0; o >>= 1) {
    const uint64_t t = __shfl_xor((unsigned long long)best, o, 64);
    best = t < best ? t : best;
  }
  return (hi > lo) ? (int64_t)perm[lo + (int)(uint32_t)best] : -1;
}

template <int LG>
__global__ __launch_bounds__(256) void pair_anchors_kernel(
    const float* __restrict__ points, const int32_t* __restrict__ perm,
    const int32_t* __restrict__ rowptr, const float* __restrict__ centroid,
    const int64_t* __restrict__ edges, int64_t E, int64_t edge_stride, int cycles,
    int64_t* __restrict__ anchors, float* __restrict__ d_nn) {
  const int gl = threadIdx.x & (LG - 1);
  const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LG;
  const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) / LG;
  for (int64_t e = group; e < E; e += ngroups) {
    const int64_t s = edges[e], t = edges[edge_stride + e];
    float sc[3], tc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) { sc[q] = centroid[s * 3 + q]; tc[q] = centroid[t * 3 + q]; }
    int64_t si = -1, ti = -1;
    for (int it = 0; it < cycles; ++it) {                             // scatter.py:229-231
      ti = closest_in_cluster<LG>(points, perm, rowptr, t, sc, gl);
      if (ti >= 0) {
#pragma unroll
        for (int q = 0; q < 3; ++q) tc[q] = points[ti * 3 + q];
      }
      si = closest_in_cluster<LG>(points, perm, rowptr, s, tc, gl);
      if (si >= 0) {
#pragma unroll
        for (int q = 0; q < 3; ++q) sc[q] = points[si * 3 + q];
      }
    }
    if (gl == 0) {
      anchors[e] = si;
      anchors[E + e] = ti;
      const float dx = sc[0] - tc[0], dy = sc[1] - tc[1], dz = sc[2] - tc[2];
      d_nn[e] = sqrtf(dx * dx + dy * dy + dz * dz);                   // neighbors.py:632
    }
  }
}

}  // namespace cgraph
}  // namespace spt

using namespace spt;
using namespace spt::cgraph;

extern "C" size_t spt_cluster_graph_edges_workspace_bytes(int64_t num_clusters, int k) {
  if (num_clusters < 0 || k < 1) return 0;
  return make_plan(num_clusters * k).total;
}

extern "C" int spt_cluster_graph_edges(const int64_t* neighbors, const float* distances,
                                       const float* r_cluster, int64_t num_clusters, int k,
                                       float gap, int trim, int64_t* edges, float* edge_dist,
                                       int64_t* count, void* ws, size_t ws_bytes,
                                       spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t S = num_clusters, m = S * k;
  SPT_CHECK_ARG(S >= 0 && k >= 1 && m < ((int64_t)1 << 31) - SORT_TILE - 1, "shape out of range");
  SPT_CHECK_ARG(count != nullptr, "count is null");
  if (m == 0) {
    (void)hipMemsetAsync(count, 0, 8, stream);
    return 0;
  }
  SPT_CHECK_ARG(neighbors && distances && r_cluster && edges && edge_dist, "null pointer");
  const Plan p = make_plan(m);
  SPT_CHECK_ARG(ws_bytes >= p.total && ws, "workspace too small");
  char* base = (char*)ws;
  uint32_t* lo = (uint32_t*)(base + p.off_lo);
  uint32_t* hi = (uint32_t*)(base + p.off_hi);
  uint32_t* key = (uint32_t*)(base + p.off_key);
  uint32_t* flag = (uint32_t*)(base + p.off_flag);
  RadixScratch s;
  s.carve(base + p.off_sort, m + 1);

  const float gap_term = (float)(1.732 * (double)gap);                // neighbors.py:592
  const int g = stream_grid(m, 256);
  candidate_keys_kernel<<<g, 256, 0, stream>>>(neighbors, distances, r_cluster, S, k, gap_term,
                                               trim, lo, hi);
  // lexicographic (lo, hi): LSD = stable sort by hi, then stable sort by lo
  const int nb = bits_for(S + 1);
  const uint32_t *ks, *vs;
  radix_sort_pairs<2>(nullptr, hi, nullptr, m, nb, s, nullptr, &ks, &vs, stream);
  gather_u32_kernel<<<g, 256, 0, stream>>>(lo, vs, m, key);
  RadixScratch s2 = s;
  if (vs == s.v0) { s2.v0 = s.v1; s2.v1 = s.v0; }
  radix_sort_pairs<0>(nullptr, key, vs, m, nb, s2, nullptr, &ks, &vs, stream);
  run_heads_kernel<<<stream_grid(m + 1, 256), 256, 0, stream>>>(lo, hi, vs, m, S, flag);
  device_exclusive_scan(flag, m + 1, s.part, stream);
  emit_edges_kernel<<<g, 256, 0, stream>>>(lo, hi, vs, flag, distances, m, S, m, edges,
                                           edge_dist, count);
  SPT_CHECK_LAUNCH();
  return 0;
}

extern "C" int spt_cluster_pair_anchors_f32(const float* points, const int32_t* perm,
                                            const int32_t* rowptr, const float* centroid,
                                            const int64_t* edges, int64_t num_edges,
                                            int64_t edge_stride, int cycles, int group_lanes,
                                            int64_t* anchors, float* d_nn, spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(num_edges >= 0 && cycles >= 1, "bad shape");
  SPT_CHECK_ARG(group_lanes == 8 || group_lanes == 16 || group_lanes == 64,
                "group_lanes must be 8, 16 or 64");
  if (num_edges == 0) return 0;
  SPT_CHECK_ARG(points && perm && rowptr && centroid && edges && anchors && d_nn, "null pointer");
  const int64_t groups_per_block = 256 / group_lanes;
  int64_t blocks = ceil_div(num_edges, groups_per_block);
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (group_lanes == 8)
    pair_anchors_kernel<8><<<(int)blocks, 256, 0, stream>>>(
        points, perm, rowptr, centroid, edges, num_edges, edge_stride, cycles, anchors, d_nn);
  else if (group_lanes == 16)
    pair_anchors_kernel<16><<<(int)blocks, 256, 0, stream>>>(
        points, perm, rowptr, centroid, edges, num_edges, edge_stride, cycles, anchors, d_nn);
  else
    pair_anchors_kernel<64><<<(int)blocks, 256, 0, stream>>>(
        points, perm, rowptr, centroid, edges, num_edges, edge_stride, cycles, anchors, d_nn);
  SPT_CHECK_LAUNCH();
  return 0;
}
