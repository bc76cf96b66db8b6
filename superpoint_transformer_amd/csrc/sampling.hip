// Per-segment random sampling without replacement.
//
// Replaces sparse_sample (src/utils/sparse.py:142-243), the sampler behind
// NAG.get_sampling (src/data/nag.py:672-711), SegmentFeatures
// (src/transforms/graph.py:215-220) and SampleSubNodes.  The reference shuffles all
// elements (fast_randperm), stable-sorts them by segment and takes the first
// n_samples[s] of every segment; its authors flag the randperm + sort as the
// bottleneck (sparse.py:213-215).  The same construction here is ONE lexicographic
// sort by (segment, 32-bit counter-based random key) on the wave64 radix sorter
// (4 + ceil(bits(num_seg)/8) passes over n elements, no host round trip), a
// per-segment count, a device scan and a compaction.
//
//   n_samples[s] = clamp(floor(n_max * tanh(size[s] / n_max)), n_min, size[s])   (n_max > 0)
//                = clamp(round(sqrt(size[s])),                n_min, size[s])   (n_max <= 0)
//   with a mask: sizes for the heuristic are the UNMASKED ones, then clamped to the
//   number of kept elements (sparse.py:180-205).
//
// Which elements are drawn depends on the RNG, so parity with the reference is on
// the counts / pointers (exact) and on the sampling contract (members of the right
// segment, distinct, kept by the mask, uniform) - see tests/test_sampling_gpu.py.
#include "radix_sort.hpp"

namespace spt {
namespace sampling {

// splitmix64 finaliser over (seed, counter): counter-based, no state, reproducible
__device__ __forceinline__ uint32_t rand32(uint64_t seed, uint64_t i) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (i + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}

__global__ void random_keys_kernel(uint64_t seed, int64_t n, uint32_t* __restrict__ keys) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    keys[i] = rand32(seed, (uint64_t)i);
}

// segment key of the element now at sorted position p (masked-out -> tail bucket)
__global__ void segment_keys_kernel(const int64_t* __restrict__ idx,
                                    const uint8_t* __restrict__ mask,
                                    const uint32_t* __restrict__ vals, int64_t n,
                                    int64_t num_seg, uint32_t* __restrict__ keys,
                                    uint32_t* __restrict__ size_all) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
    const uint32_t v = vals[p];
    const int64_t s = idx[v];
    const bool ok = s >= 0 && s < num_seg;
    if (size_all && ok) atomicAdd(&size_all[s], 1u);   // integer atomics: deterministic
    const bool keep = ok && (!mask || mask[v]);
    keys[p] = keep ? (uint32_t)s : (uint32_t)num_seg;
  }
}

__global__ void sample_counts_kernel(const int32_t* __restrict__ rowptr,
                                     const uint32_t* __restrict__ size_all,
                                     int64_t num_seg, int n_max, int n_min,
                                     uint32_t* __restrict__ cnt) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s <= num_seg; s += stride) {
    if (s == num_seg) { cnt[s] = 0; continue; }        // slot for the scan total
    const int64_t kept = rowptr[s + 1] - rowptr[s];
    const int64_t size = size_all ? (int64_t)size_all[s] : kept;
    int64_t ns;
    if (n_max > 0) {
      // f32 like the reference: size / n_max (true divide), tanh, * n_max, floor
      const float t = tanhf((float)size / (float)n_max);
      ns = (int64_t)floorf((float)n_max * t);
    } else {
      ns = (int64_t)rintf(sqrtf((float)size));          // torch.round = half to even
    }
    if (ns < n_min) ns = n_min;
    if (ns > size) ns = size;
    if (ns > kept) ns = kept;
    cnt[s] = (uint32_t)ns;
  }
}

__global__ void widen_ptr_kernel(const uint32_t* __restrict__ ptr32, int64_t m,
                                 int64_t* __restrict__ ptr64) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride)
    ptr64[i] = (int64_t)ptr32[i];
}

// sorted position p -> (segment s, rank j inside s); keep the first cnt[s] of each
__global__ void take_samples_kernel(const uint32_t* __restrict__ skeys,
                                    const uint32_t* __restrict__ svals,
                                    const int32_t* __restrict__ rowptr,
                                    const uint32_t* __restrict__ ptr32, int64_t n,
                                    int64_t num_seg, int64_t* __restrict__ out_idx) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
    const uint32_t s = skeys[p];
    if ((int64_t)s >= num_seg) continue;
    const uint32_t j = (uint32_t)(p - rowptr[s]);
    const uint32_t lo = ptr32[s], hi = ptr32[s + 1];
    if (j < hi - lo) out_idx[lo + j] = (int64_t)svals[p];
  }
}

struct Plan {
  size_t off_keys, off_rowptr, off_cnt, off_size, off_sort, total;
};

static Plan make_plan(int64_t n, int64_t num_seg) {
  Plan p;
  size_t o = 0;
  const int64_t m = n > 0 ? n : 1;
  p.off_keys = o;   o += align_up((size_t)m * 4, 256);
  p.off_rowptr = o; o += align_up((size_t)(num_seg + 2) * 4, 256);
  p.off_cnt = o;    o += align_up((size_t)(num_seg + 1) * 4, 256);
  p.off_size = o;   o += align_up((size_t)(num_seg + 1) * 4, 256);
  p.off_sort = o;   o += RadixScratch::bytes(n);
  p.total = o;
  return p;
}

}  // namespace sampling
}  // namespace spt

using namespace spt;
using namespace spt::sampling;

extern "C" size_t spt_sparse_sample_workspace_bytes(int64_t n, int64_t num_seg) {
  if (n < 0 || num_seg < 1) return 0;
  return make_plan(n, num_seg).total;
}

extern "C" int spt_sparse_sample(const int64_t* idx, int64_t n, int64_t num_seg,
                                 const uint8_t* mask, int n_max, int n_min, uint64_t seed,
                                 int64_t* out_ptr, int64_t* out_idx, void* ws,
                                 size_t ws_bytes, spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(n >= 0 && n < ((int64_t)1 << 31) - SORT_TILE, "n out of range");
  SPT_CHECK_ARG(num_seg >= 1 && num_seg < ((int64_t)1 << 31) - 2, "num_seg out of range");
  SPT_CHECK_ARG(n_min >= 0 && n_min <= n_max, "need 0 <= n_min <= n_max");
  SPT_CHECK_ARG(out_ptr != nullptr, "out_ptr is null");
  SPT_CHECK_ARG(n == 0 || (idx && out_idx), "null idx/out_idx");
  const Plan p = make_plan(n, num_seg);
  SPT_CHECK_ARG(ws_bytes >= p.total && ws, "workspace too small");
  char* base = (char*)ws;
  uint32_t* keys = (uint32_t*)(base + p.off_keys);
  int32_t* rowptr = (int32_t*)(base + p.off_rowptr);
  uint32_t* cnt = (uint32_t*)(base + p.off_cnt);
  uint32_t* size_all = mask ? (uint32_t*)(base + p.off_size) : nullptr;
  RadixScratch s;
  s.carve(base + p.off_sort, n);

  const uint32_t *ks = nullptr, *vs = nullptr;
  if (n > 0) {
    // 1. shuffle: sort positions by a 32-bit random key
    random_keys_kernel<<<stream_grid(n, 256), 256, 0, stream>>>(seed, n, keys);
    radix_sort_pairs<2>(nullptr, keys, nullptr, n, 32, s, nullptr, &ks, &vs, stream);
    // 2. stable sort of the shuffled elements by segment (masked-out -> bucket num_seg)
    if (size_all) (void)hipMemsetAsync(size_all, 0, (size_t)(num_seg + 1) * 4, stream);
    segment_keys_kernel<<<stream_grid(n, 256), 256, 0, stream>>>(idx, mask, vs, n, num_seg,
                                                                keys, size_all);
    // the shuffled values sit in one of the scratch value buffers; sorting with
    // MODE 0 ping-pongs between them, starting from the OTHER buffer
    RadixScratch s2 = s;
    if (vs == s.v0) { s2.v0 = s.v1; s2.v1 = s.v0; }
    radix_sort_pairs<0>(nullptr, keys, vs, n, bits_for(num_seg + 1), s2, nullptr, &ks, &vs,
                        stream);
  }
  rowptr_from_sorted_kernel<<<stream_grid(n + 1, 256), 256, 0, stream>>>(ks, n, num_seg + 1,
                                                                        rowptr);
  // 3. how many to take per segment, exclusive scan -> pointers
  sample_counts_kernel<<<stream_grid(num_seg + 1, 256), 256, 0, stream>>>(
      rowptr, size_all, num_seg, n_max, n_min, cnt);
  device_exclusive_scan(cnt, num_seg + 1, s.part, stream);
  widen_ptr_kernel<<<stream_grid(num_seg + 1, 256), 256, 0, stream>>>(cnt, num_seg + 1, out_ptr);
  // 4. first cnt[s] shuffled members of every segment
  if (n > 0)
    take_samples_kernel<<<stream_grid(n, 256), 256, 0, stream>>>(ks, vs, rowptr, cnt, n,
                                                                num_seg, out_idx);
  SPT_CHECK_LAUNCH();
  return 0;
}
