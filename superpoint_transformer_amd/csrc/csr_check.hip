// Consistency check of a STORED segment CSR against the index it claims to be the view of
// (round 5, advisor: `csr.adopt_csr` installs nag[i+1].sub - pointers, points - as the memoised
// view of nag[i].super_index; src/data/cluster.py:19-77 keeps the two consistent, a hand-built or
// stale `sub` does not).  One launch, no host round trip: every violated condition ORs a bit into
// a device flag word that the caller reads whenever it likes.
//   bit 0: pointers does not start at 0 / end at n / is decreasing somewhere
//   bit 1: a point id outside [0, n)
//   bit 2: membership - idx[points[j]] is not the segment whose range holds position j
//   bit 3: the points of a segment are not strictly ascending (the stable-sort order)
#include "common.hpp"

namespace spt {

__global__ __launch_bounds__(256) void csr_check_kernel(
    const int64_t* __restrict__ idx, const int64_t* __restrict__ points,
    const int64_t* __restrict__ pointers, int64_t n, int64_t num_seg, int check_ascending,
    int32_t* __restrict__ flag) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int bad = 0;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n || j <= num_seg; j += stride) {
    if (j < num_seg && pointers[j] > pointers[j + 1]) bad |= 1;
    if (j == 0 && (pointers[0] != 0 || pointers[num_seg] != n)) bad |= 1;
    if (j >= n) continue;
    const int64_t p = points[j];
    if (p < 0 || p >= n) {
      bad |= 2;
      continue;
    }
    // segment of position j: the last s with pointers[s] <= j (bounded search: garbage pointers
    // cannot lead it outside [0, num_seg))
    int64_t lo = 0, hi = num_seg - 1;
    while (lo < hi) {
      const int64_t mid = (lo + hi + 1) >> 1;
      if (pointers[mid] <= j) lo = mid; else hi = mid - 1;
    }
    if (!(pointers[lo] <= j && j < pointers[lo + 1]) || idx[p] != lo) bad |= 4;
    if (check_ascending && j > 0 && j > pointers[lo] && points[j - 1] >= p) bad |= 8;
  }
  if (bad) atomicOr(flag, bad);
}

// Round 6 (advisor, medium): adopt = check + the int32 casts the segment kernels read, in ONE launch.
// The casts are CLAMPED (perm into [0, n), rowptr into [0, n]), so a stored CSR that fails the
// check cannot send a segment kernel outside its buffers while the verdict is still on its way to
// the host (csr.adopt_csr(verify="deferred")).  No search: position j holds point p, which CLAIMS
// segment s = idx[p]; the claim is true iff j lies in [pointers[s], pointers[s + 1]) - with
// monotone pointers (bit 0) the ranges are disjoint, so this is "idx[p] is the segment holding
// position j" in three dependent loads per position instead of a 19-step binary search.  The
// ascending comparison always runs (one neighbouring load).
__global__ __launch_bounds__(256) void csr_adopt_kernel(
    const int64_t* __restrict__ idx, const int64_t* __restrict__ points,
    const int64_t* __restrict__ pointers, int64_t n, int64_t num_seg,
    int32_t* __restrict__ perm32, int32_t* __restrict__ rowptr32, int32_t* __restrict__ flag) {
  int bad = 0;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  for (int64_t s = tid; s <= num_seg; s += nthreads) {
    const int64_t p = pointers[s];
    if (s < num_seg && p > pointers[s + 1]) bad |= 1;
    if ((s == 0 && p != 0) || (s == num_seg && p != n)) bad |= 1;
    rowptr32[s] = (int32_t)(p < 0 ? 0 : (p > n ? n : p));
  }
  for (int64_t j = tid; j < n; j += nthreads) {
    const int64_t p = points[j];
    const bool inr = p >= 0 && p < n;
    perm32[j] = (int32_t)(inr ? p : (p < 0 ? 0 : n - 1));
    if (!inr) {
      bad |= 2;
      continue;
    }
    const int64_t s = idx[p];
    if (s < 0 || s >= num_seg) {
      bad |= 4;
      continue;
    }
    const int64_t a = pointers[s], b = pointers[s + 1];
    if (!(a <= j && j < b)) bad |= 4;
    else if (j > 0 && j > a && points[j - 1] >= p) bad |= 8;
  }
  if (bad) atomicOr(flag, bad);
}

}  // namespace spt

extern "C" int spt_csr_adopt_i64(const int64_t* idx, const int64_t* points, const int64_t* pointers,
                                 int64_t n, int64_t num_seg, int32_t* perm32, int32_t* rowptr32,
                                 int32_t* flag, spt_stream_t stream_) {
  SPT_CHECK_ARG(n >= 0 && num_seg >= 1 && n < (int64_t)INT32_MAX, "bad shape");
  SPT_CHECK_ARG(pointers && flag && rowptr32 && (n == 0 || (idx && points && perm32)), "null pointer");
  const int64_t work = n > num_seg + 1 ? n : num_seg + 1;
  spt::csr_adopt_kernel<<<spt::stream_grid(work, 256), 256, 0, (hipStream_t)stream_>>>(
      idx, points, pointers, n, num_seg, perm32, rowptr32, flag);
  SPT_CHECK_LAUNCH();
  return 0;
}

extern "C" int spt_csr_check_i64(const int64_t* idx, const int64_t* points, const int64_t* pointers,
                                 int64_t n, int64_t num_seg, int check_ascending, int32_t* flag,
                                 spt_stream_t stream_) {
  SPT_CHECK_ARG(n >= 0 && num_seg >= 1, "bad shape");
  SPT_CHECK_ARG(pointers && flag && (n == 0 || (idx && points)), "null pointer");
  const int64_t work = n > num_seg + 1 ? n : num_seg + 1;
  spt::csr_check_kernel<<<spt::stream_grid(work, 256), 256, 0, (hipStream_t)stream_>>>(
      idx, points, pointers, n, num_seg, check_ascending, flag);
  SPT_CHECK_LAUNCH();
  return 0;
}
