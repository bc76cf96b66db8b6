// CSR view of an unsorted segment index: stable LSD radix sort of (idx, iota)
// followed by boundary detection.  Hand-written for gfx950 (wave64 ballots for
// the stable in-wave ranking; 4-wave workgroups; LDS digit counters).
//
// Why a sort at all: the reference groups rows by index with float atomics in
// torch_scatter's COO kernels on every call (src/nn/pool.py:61-62,
// src/nn/norm.py:118-126, src/nn/attention.py:307,315).  We group ONCE per
// batch and level, deterministically, and every later segment kernel streams
// rows through (perm, rowptr) without atomics.
#include "radix_sort.hpp"

using namespace spt;

extern "C" size_t spt_csr_build_workspace_bytes(int64_t n, int64_t num_seg) {
  if (n < 0 || num_seg < 1) return 0;
  return RadixScratch::bytes(n);
}

extern "C" int spt_csr_build(const int64_t* idx, int64_t n, int64_t num_seg,
                             int32_t* perm, int32_t* rowptr, void* ws,
                             size_t ws_bytes, spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(n >= 0 && n < ((int64_t)1 << 31) - SORT_TILE, "n out of range");
  SPT_CHECK_ARG(num_seg >= 1 && num_seg < ((int64_t)1 << 31), "num_seg out of range");
  SPT_CHECK_ARG(rowptr != nullptr, "rowptr is null");
  SPT_CHECK_ARG(n == 0 || (idx && perm), "null idx/perm");
  const size_t need = RadixScratch::bytes(n);
  SPT_CHECK_ARG(ws_bytes >= need && (ws || need == 0), "workspace too small");

  if (n == 0) {
    rowptr_from_sorted_kernel<<<stream_grid(num_seg + 1, 256), 256, 0, stream>>>(
        nullptr, 0, num_seg, rowptr);
    SPT_CHECK_LAUNCH();
    return 0;
  }
  if (num_seg <= 1) {
    iota_kernel<<<stream_grid(n, 256), 256, 0, stream>>>(perm, n);
    rowptr_single_kernel<<<1, 1, 0, stream>>>(rowptr, n);
    SPT_CHECK_LAUNCH();
    return 0;
  }
  RadixScratch s;
  s.carve((char*)ws, n);
  const uint32_t *ks, *vs;
  radix_sort_pairs<1>(idx, nullptr, nullptr, n, bits_for(num_seg), s, (uint32_t*)perm, &ks,
                      &vs, stream);
  rowptr_from_sorted_kernel<<<stream_grid(n + 1, 256), 256, 0, stream>>>(
      ks, n, num_seg, rowptr);
  SPT_CHECK_LAUNCH();
  return 0;
}


// ---- companions of the CSR view the host side otherwise builds with torch gathers ------------
namespace spt {
// pos_seg[j] = the segment of CSR position j (= idx[perm[j]], without touching idx: segment s
// owns positions rowptr[s] .. rowptr[s + 1]).  One lane group of 16 per segment (segments of the
// L0 -> L1 pool hold ~35 rows), coalesced stores.
__global__ __launch_bounds__(256) void csr_pos_seg_kernel(const int32_t* __restrict__ rowptr,
                                                          int64_t num_seg,
                                                          int32_t* __restrict__ pos_seg) {
  const int sub = threadIdx.x & 15;
  const int64_t stride = (int64_t)gridDim.x * 16;
  for (int64_t s = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); s < num_seg; s += stride) {
    const int a = rowptr[s], b = rowptr[s + 1];
    for (int j = a + sub; j < b; j += 16) pos_seg[j] = (int32_t)s;
  }
}
// out[j] = (int32) src[perm[j]]: e.g. the targets of the edges in CSR order
__global__ __launch_bounds__(256) void csr_gather_i64_i32_kernel(const int64_t* __restrict__ src,
                                                                 const int32_t* __restrict__ perm,
                                                                 int64_t n, int32_t* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride)
    out[j] = (int32_t)src[perm[j]];
}
}  // namespace spt

extern "C" int spt_csr_pos_seg(const int32_t* rowptr, int64_t num_seg, int64_t n, int32_t* pos_seg,
                               spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(num_seg >= 0 && n >= 0, "bad shape");
  if (n == 0 || num_seg == 0) return 0;
  SPT_CHECK_ARG(rowptr && pos_seg, "null pointer");
  spt::csr_pos_seg_kernel<<<spt::stream_grid(num_seg, 16), 256, 0, stream>>>(rowptr, num_seg, pos_seg);
  SPT_CHECK_LAUNCH();
  return 0;
}

extern "C" int spt_csr_gather_i64_i32(const int64_t* src, const int32_t* perm, int64_t n,
                                      int32_t* out, spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(n >= 0, "bad shape");
  if (n == 0) return 0;
  SPT_CHECK_ARG(src && perm && out, "null pointer");
  spt::csr_gather_i64_i32_kernel<<<spt::stream_grid(n, 256), 256, 0, stream>>>(src, perm, n, out);
  SPT_CHECK_LAUNCH();
  return 0;
}
