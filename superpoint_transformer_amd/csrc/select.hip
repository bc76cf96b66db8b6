// Per-batch NAG selection / re-indexing on the device.
//
// The integer work behind NAG.select (src/data/nag.py:306-399), Data.select
// (src/data/data.py:286-470) and Cluster.select (src/data/cluster.py:79-140):
// relabel what survives to dense ids, drop what does not, keep every level of the
// hierarchy consistent.  The reference does this with torch.unique-based
// consecutive_cluster calls (flagged "bottleneck" at cluster.py:128 and data.py:403),
// scatter_ re-index tables and boolean indexing.  All labels here are bounded
// (point / cluster ids of a known level size), so every "unique + inverse" is a
// presence bitmap + device scan + gather/compaction: O(n) streaming passes, no sort,
// deterministic.
//
//   spt_index_inverse        reindex table of data.py:365-368
//   spt_select_edges         data.py:369-373   (relabel, drop, idx_edge; order kept)
//   spt_cluster_select       CSRData.select + cluster.py:127-138 (idx_sub, sub_super)
//   spt_relabel_consecutive  consecutive_cluster on bounded labels (data.py:404-406)
#include "radix_sort.hpp"

namespace spt {
namespace sel {

__global__ void fill_i64_kernel(int64_t* __restrict__ a, int64_t n, int64_t v) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) a[i] = v;
}

__global__ void scatter_iota_kernel(const int64_t* __restrict__ idx, int64_t k, int64_t n,
                                    int64_t* __restrict__ inv) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < k; j += stride) {
    const int64_t v = idx[j];
    if (v >= 0 && v < n) inv[v] = j;
  }
}

__global__ void edge_flags_kernel(const int64_t* __restrict__ ei, int64_t E, int64_t stride_e,
                                  const int64_t* __restrict__ inv, int64_t n,
                                  uint32_t* __restrict__ flag) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e <= E; e += stride) {
    if (e == E) { flag[e] = 0; continue; }
    const int64_t s = ei[e], t = ei[stride_e + e];
    const bool ok = s >= 0 && s < n && t >= 0 && t < n && inv[s] >= 0 && inv[t] >= 0;
    flag[e] = ok ? 1u : 0u;
  }
}

__global__ void edge_emit_kernel(const int64_t* __restrict__ ei, int64_t E, int64_t stride_e,
                                 const int64_t* __restrict__ inv, const uint32_t* __restrict__ pos,
                                 int64_t out_stride, int64_t* __restrict__ out,
                                 int64_t* __restrict__ idx_edge, int64_t* __restrict__ count) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += stride) {
    if (e == 0) *count = (int64_t)pos[E];
    if (pos[e + 1] == pos[e]) continue;
    const int64_t o = pos[e];
    out[o] = inv[ei[e]];
    out[out_stride + o] = inv[ei[stride_e + e]];
    idx_edge[o] = e;
  }
}

// ---- cluster select ------------------------------------------------------------------
__global__ void selected_sizes_kernel(const int64_t* __restrict__ ptr,
                                      const int64_t* __restrict__ idx, int64_t k,
                                      uint32_t* __restrict__ sizes) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j <= k; j += stride)
    sizes[j] = (j < k) ? (uint32_t)(ptr[idx[j] + 1] - ptr[idx[j]]) : 0u;
}

// cluster of output position o: largest j with newptr[j] <= o (empty clusters skipped)
__device__ __forceinline__ int64_t owner_of(const uint32_t* __restrict__ newptr, int64_t k,
                                            uint32_t o) {
  int64_t lo = 0, hi = k;                 // invariant: newptr[lo] <= o < newptr[hi]
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (newptr[mid] <= o) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ void cluster_copy_kernel(const int64_t* __restrict__ ptr,
                                    const int64_t* __restrict__ points,
                                    const int64_t* __restrict__ idx,
                                    const uint32_t* __restrict__ newptr, int64_t k,
                                    int64_t n_sub, int64_t* __restrict__ tmp_points,
                                    uint32_t* __restrict__ owner, uint32_t* __restrict__ present) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t m_new = newptr[k];
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < m_new; o += stride) {
    const int64_t j = owner_of(newptr, k, (uint32_t)o);
    const int64_t p = points[ptr[idx[j]] + (o - newptr[j])];
    tmp_points[o] = p;
    owner[o] = (uint32_t)j;
    if (p >= 0 && p < n_sub) present[p] = 1u;         // same value from every writer
  }
}

__global__ void cluster_emit_kernel(const int64_t* __restrict__ tmp_points,
                                    const uint32_t* __restrict__ owner,
                                    const uint32_t* __restrict__ rank,
                                    const uint32_t* __restrict__ newptr, int64_t k,
                                    int64_t n_sub, int64_t* __restrict__ new_points,
                                    int64_t* __restrict__ sub_super) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t m_new = newptr[k];
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < m_new; o += stride) {
    const int64_t p = tmp_points[o];
    if (p < 0 || p >= n_sub) { new_points[o] = -1; continue; }
    const int64_t r = rank[p];
    new_points[o] = r;
    sub_super[r] = (int64_t)owner[o];                 // cluster.to_super_index (cluster.py:67-77)
  }
}

// values present in [0, n): uniques[rank[v]] = v
__global__ void compact_present_kernel(const uint32_t* __restrict__ rank, int64_t n,
                                       int64_t* __restrict__ uniques,
                                       int64_t* __restrict__ count) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v <= n; v += stride) {
    if (v == n) { *count = (int64_t)rank[n]; continue; }
    if (rank[v + 1] != rank[v]) uniques[rank[v]] = v;
  }
}

__global__ void widen_kernel(const uint32_t* __restrict__ a, int64_t n, int64_t* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = (int64_t)a[i];
}

// ---- relabel -------------------------------------------------------------------------
__global__ void mark_values_kernel(const int64_t* __restrict__ values,
                                   const int64_t* __restrict__ gather, int64_t k, int64_t n,
                                   uint32_t* __restrict__ present) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < k; i += stride) {
    const int64_t v = values[gather ? gather[i] : i];
    if (v >= 0 && v < n) present[v] = 1u;
  }
}

__global__ void relabel_kernel(const int64_t* __restrict__ values,
                               const int64_t* __restrict__ gather, int64_t k, int64_t n,
                               const uint32_t* __restrict__ rank, int64_t* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < k; i += stride) {
    const int64_t v = values[gather ? gather[i] : i];
    out[i] = (v >= 0 && v < n) ? (int64_t)rank[v] : -1;
  }
}

// ---- radius ball around a seed (SampleRadiusSubgraphs) -------------------------------
__global__ void ball_flags_kernel(const float* __restrict__ pos, int64_t n, float cx, float cy,
                                  float cz, float wz, float r, const int64_t* __restrict__ batch,
                                  int64_t batch_id, uint32_t* __restrict__ flag) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += stride) {
    if (i == n) { flag[i] = 0; continue; }
    const float dx = pos[i * 3] - cx, dy = pos[i * 3 + 1] - cy, dz = (pos[i * 3 + 2] - cz) * wz;
    const float d = sqrtf(dx * dx + dy * dy + dz * dz);      // (xyz_search - xyz_query).norm()
    const bool same = !batch || batch[i] == batch_id;
    flag[i] = (same && d <= r) ? 1u : 0u;                    // neighbors.py:283-285 keeps d <= r
  }
}

static size_t scan_part_bytes(int64_t m) {
  return align_up((size_t)ceil_div(m > 0 ? m : 1, SCAN_TILE) * 4, 256);
}

}  // namespace sel
}  // namespace spt

using namespace spt;
using namespace spt::sel;

extern "C" int spt_index_inverse(const int64_t* idx, int64_t k, int64_t n, int64_t* inv,
                                 spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(k >= 0 && n >= 0, "bad shape");
  if (n == 0) return 0;
  SPT_CHECK_ARG(inv && (k == 0 || idx), "null pointer");
  fill_i64_kernel<<<stream_grid(n, 256), 256, 0, stream>>>(inv, n, -1);
  if (k > 0) scatter_iota_kernel<<<stream_grid(k, 256), 256, 0, stream>>>(idx, k, n, inv);
  SPT_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t spt_select_edges_workspace_bytes(int64_t num_edges) {
  if (num_edges < 0) return 0;
  return align_up((size_t)(num_edges + 1) * 4, 256) + scan_part_bytes(num_edges + 1);
}

extern "C" int spt_select_edges(const int64_t* edge_index, int64_t num_edges, int64_t edge_stride,
                                const int64_t* inv, int64_t n, int64_t* out_edges,
                                int64_t out_stride, int64_t* idx_edge, int64_t* count, void* ws,
                                size_t ws_bytes, spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t E = num_edges;
  SPT_CHECK_ARG(E >= 0 && E < ((int64_t)1 << 32) - 2 && n >= 0, "bad shape");
  SPT_CHECK_ARG(count != nullptr, "count is null");
  if (E == 0) {
    (void)hipMemsetAsync(count, 0, 8, stream);
    return 0;
  }
  SPT_CHECK_ARG(edge_index && inv && out_edges && idx_edge, "null pointer");
  SPT_CHECK_ARG(ws && ws_bytes >= spt_select_edges_workspace_bytes(E), "workspace too small");
  uint32_t* flag = (uint32_t*)ws;
  uint32_t* part = (uint32_t*)((char*)ws + align_up((size_t)(E + 1) * 4, 256));
  edge_flags_kernel<<<stream_grid(E + 1, 256), 256, 0, stream>>>(edge_index, E, edge_stride, inv,
                                                                n, flag);
  device_exclusive_scan(flag, E + 1, part, stream);
  edge_emit_kernel<<<stream_grid(E, 256), 256, 0, stream>>>(edge_index, E, edge_stride, inv, flag,
                                                           out_stride, out_edges, idx_edge, count);
  SPT_CHECK_LAUNCH();
  return 0;
}

// ---- neighbors_dense_to_csr (src/utils/neighbors.py:668-684) ------------------------------
// [n, k] neighbour table with negative = missing -> (ptr [n+1], val [nnz], sizes [n]):
// per-row counts, one device scan, one emit pass that keeps each row's order.  The reference
// does this with a boolean-mask gather (`nn[~mask]`: a scan + host sync inside torch).
namespace {
__global__ __launch_bounds__(256) void dense_count_kernel(const int64_t* __restrict__ nn, int64_t n,
                                                          int k, uint32_t* __restrict__ cnt,
                                                          int64_t* __restrict__ sizes) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += stride) {
    uint32_t c = 0;
    if (i < n) {
      for (int j = 0; j < k; ++j) c += nn[i * k + j] >= 0 ? 1u : 0u;
      sizes[i] = (int64_t)c;
    }
    cnt[i] = c;                                     // cnt[n] = 0: the scan leaves the total there
  }
}
__global__ __launch_bounds__(256) void dense_emit_kernel(const int64_t* __restrict__ nn, int64_t n,
                                                         int k, const uint32_t* __restrict__ off,
                                                         int64_t* __restrict__ ptr,
                                                         int64_t* __restrict__ val) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += stride) {
    int64_t o = (int64_t)off[i];
    ptr[i] = o;
    if (i < n)
      for (int j = 0; j < k; ++j) {
        const int64_t v = nn[i * k + j];
        if (v >= 0) val[o++] = v;
      }
  }
}
}  // namespace

extern "C" size_t spt_neighbors_dense_to_csr_workspace_bytes(int64_t n) {
  if (n < 0) return 0;
  return align_up((size_t)(n + 1) * 4, 256) + scan_part_bytes(n + 1);
}

extern "C" int spt_neighbors_dense_to_csr(const int64_t* nn, int64_t n, int k, int64_t* ptr,
                                          int64_t* val, int64_t* sizes, void* ws,
                                          size_t ws_bytes, spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(n >= 0 && k >= 0, "bad shape");
  SPT_CHECK_ARG((int64_t)n * k < ((int64_t)1 << 32) - 2, "more than 2^32 entries");
  SPT_CHECK_ARG(ptr && ws && ws_bytes >= spt_neighbors_dense_to_csr_workspace_bytes(n),
                "null pointer / workspace too small");
  SPT_CHECK_ARG(n == 0 || (sizes && (k == 0 || (nn && val))), "null pointer");
  uint32_t* cnt = (uint32_t*)ws;
  uint32_t* part = (uint32_t*)((char*)ws + align_up((size_t)(n + 1) * 4, 256));
  dense_count_kernel<<<stream_grid(n + 1, 256), 256, 0, stream>>>(nn, n, k, cnt, sizes);
  device_exclusive_scan(cnt, n + 1, part, stream);
  dense_emit_kernel<<<stream_grid(n + 1, 256), 256, 0, stream>>>(nn, n, k, cnt, ptr, val);
  SPT_CHECK_LAUNCH();
  return 0;
}

namespace {
struct ClusterPlan {
  size_t off_sizes, off_owner, off_present, off_tmp, off_part, total;
};
ClusterPlan cluster_plan(int64_t k, int64_t m, int64_t n_sub) {
  ClusterPlan p;
  size_t o = 0;
  p.off_sizes = o;   o += align_up((size_t)(k + 1) * 4, 256);
  p.off_owner = o;   o += align_up((size_t)(m > 0 ? m : 1) * 4, 256);
  p.off_present = o; o += align_up((size_t)(n_sub + 1) * 4, 256);
  p.off_tmp = o;     o += align_up((size_t)(m > 0 ? m : 1) * 8, 256);
  const int64_t big = (k + 1 > n_sub + 1) ? k + 1 : n_sub + 1;
  p.off_part = o;    o += scan_part_bytes(big);
  p.total = o;
  return p;
}
}  // namespace

extern "C" size_t spt_cluster_select_workspace_bytes(int64_t k, int64_t num_points,
                                                     int64_t n_sub) {
  if (k < 0 || num_points < 0 || n_sub < 0) return 0;
  return cluster_plan(k, num_points, n_sub).total;
}

extern "C" int spt_cluster_select(const int64_t* pointers, const int64_t* points,
                                  int64_t num_points, const int64_t* idx, int64_t k,
                                  int64_t n_sub, int64_t* new_pointers, int64_t* new_points,
                                  int64_t* idx_sub, int64_t* sub_super, int64_t* count_sub,
                                  void* ws, size_t ws_bytes, spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t M = num_points;
  SPT_CHECK_ARG(k >= 0 && M >= 0 && n_sub >= 0 && M < ((int64_t)1 << 32) - 2 &&
                    n_sub < ((int64_t)1 << 32) - 2, "bad shape");
  SPT_CHECK_ARG(new_pointers && count_sub, "null output");
  const ClusterPlan p = cluster_plan(k, M, n_sub);
  SPT_CHECK_ARG(ws && ws_bytes >= p.total, "workspace too small");
  char* base = (char*)ws;
  uint32_t* sizes = (uint32_t*)(base + p.off_sizes);
  uint32_t* owner = (uint32_t*)(base + p.off_owner);
  uint32_t* present = (uint32_t*)(base + p.off_present);
  int64_t* tmp = (int64_t*)(base + p.off_tmp);
  uint32_t* part = (uint32_t*)(base + p.off_part);
  if (k == 0) {
    (void)hipMemsetAsync(new_pointers, 0, 8, stream);
    (void)hipMemsetAsync(count_sub, 0, 8, stream);
    return 0;
  }
  SPT_CHECK_ARG(pointers && idx && (M == 0 || (points && new_points && idx_sub && sub_super)),
                "null pointer");
  // new pointers = exclusive scan of the selected sizes        (csr.py:343-346)
  selected_sizes_kernel<<<stream_grid(k + 1, 256), 256, 0, stream>>>(pointers, idx, k, sizes);
  device_exclusive_scan(sizes, k + 1, part, stream);
  widen_kernel<<<stream_grid(k + 1, 256), 256, 0, stream>>>(sizes, k + 1, new_pointers);
  (void)hipMemsetAsync(present, 0, (size_t)(n_sub + 1) * 4, stream);
  // the number of copied points (newptr[k]) is only known on the device: launch over
  // the upper bound M, the kernels stop at newptr[k]
  if (M > 0) {
    const int g = stream_grid(M, 256);
    cluster_copy_kernel<<<g, 256, 0, stream>>>(pointers, points, idx, sizes, k, n_sub, tmp, owner,
                                               present);
    device_exclusive_scan(present, n_sub + 1, part, stream);
    cluster_emit_kernel<<<g, 256, 0, stream>>>(tmp, owner, present, sizes, k, n_sub, new_points,
                                               sub_super);
  } else {
    device_exclusive_scan(present, n_sub + 1, part, stream);
  }
  // idx_sub = the surviving sub points in ascending order      (cluster.py:130-131)
  compact_present_kernel<<<stream_grid(n_sub + 1, 256), 256, 0, stream>>>(present, n_sub, idx_sub,
                                                                        count_sub);
  SPT_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t spt_relabel_consecutive_workspace_bytes(int64_t n_range) {
  if (n_range < 0) return 0;
  return align_up((size_t)(n_range + 1) * 4, 256) + scan_part_bytes(n_range + 1);
}

extern "C" int spt_relabel_consecutive(const int64_t* values, const int64_t* gather, int64_t k,
                                       int64_t n_range, int64_t* new_values, int64_t* uniques,
                                       int64_t* count, void* ws, size_t ws_bytes,
                                       spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(k >= 0 && n_range >= 0 && n_range < ((int64_t)1 << 32) - 2, "bad shape");
  SPT_CHECK_ARG(count != nullptr, "count is null");
  SPT_CHECK_ARG(ws && ws_bytes >= spt_relabel_consecutive_workspace_bytes(n_range),
                "workspace too small");
  SPT_CHECK_ARG(k == 0 || (values && new_values), "null pointer");
  SPT_CHECK_ARG(n_range == 0 || uniques, "uniques is null");
  uint32_t* present = (uint32_t*)ws;
  uint32_t* part = (uint32_t*)((char*)ws + align_up((size_t)(n_range + 1) * 4, 256));
  (void)hipMemsetAsync(present, 0, (size_t)(n_range + 1) * 4, stream);
  if (k > 0)
    mark_values_kernel<<<stream_grid(k, 256), 256, 0, stream>>>(values, gather, k, n_range, present);
  device_exclusive_scan(present, n_range + 1, part, stream);
  if (k > 0)
    relabel_kernel<<<stream_grid(k, 256), 256, 0, stream>>>(values, gather, k, n_range, present,
                                                            new_values);
  compact_present_kernel<<<stream_grid(n_range + 1, 256), 256, 0, stream>>>(present, n_range,
                                                                          uniques, count);
  SPT_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t spt_radius_ball_workspace_bytes(int64_t n) {
  if (n < 0) return 0;
  return align_up((size_t)(n + 1) * 4, 256) + scan_part_bytes(n + 1);
}

extern "C" int spt_radius_ball_f32(const float* pos, int64_t n, const float* center, float r,
                                   int cylindrical, const int64_t* batch, int64_t batch_id,
                                   int64_t* out_idx, int64_t* count, void* ws, size_t ws_bytes,
                                   spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(n >= 0 && n < ((int64_t)1 << 32) - 2 && r >= 0.f, "bad shape");
  SPT_CHECK_ARG(count && center, "null pointer");
  SPT_CHECK_ARG(ws && ws_bytes >= spt_radius_ball_workspace_bytes(n), "workspace too small");
  SPT_CHECK_ARG(n == 0 || (pos && out_idx), "null pointer");
  uint32_t* flag = (uint32_t*)ws;
  uint32_t* part = (uint32_t*)((char*)ws + align_up((size_t)(n + 1) * 4, 256));
  ball_flags_kernel<<<stream_grid(n + 1, 256), 256, 0, stream>>>(
      pos, n, center[0], center[1], center[2], cylindrical ? 0.f : 1.f, r, batch, batch_id, flag);
  device_exclusive_scan(flag, n + 1, part, stream);
  compact_present_kernel<<<stream_grid(n + 1, 256), 256, 0, stream>>>(flag, n, out_idx, count);
  SPT_CHECK_LAUNCH();
  return 0;
}
