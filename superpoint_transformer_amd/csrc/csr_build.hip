// CSR view of an unsorted segment index: stable LSD radix sort of (idx, iota)
// followed by boundary detection.  Hand-written for gfx950 (wave64 ballots for
// the stable in-wave ranking; 4-wave workgroups; LDS digit counters).
//
// Why a sort at all: the reference groups rows by index with float atomics in
// torch_scatter's COO kernels on every call (src/nn/pool.py:61-62,
// src/nn/norm.py:118-126, src/nn/attention.py:307,315).  We group ONCE per
// batch and level, deterministically, and every later segment kernel streams
// rows through (perm, rowptr) without atomics.
#include "common.hpp"

namespace spt {

constexpr int SORT_THREADS = 256;
constexpr int SORT_WAVES = SORT_THREADS / 64;
constexpr int SORT_ITEMS = 16;                        // keys per lane
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;  // 4096 keys per workgroup
constexpr int WAVE_CHUNK = SORT_TILE / SORT_WAVES;    // 1024 consecutive keys per wave
constexpr int MAX_BINS = 256;

template <bool FIRST>
__device__ __forceinline__ uint32_t load_key(const int64_t* __restrict__ idx,
                                             const uint32_t* __restrict__ keys,
                                             int64_t i) {
  if constexpr (FIRST)
    return (uint32_t)idx[i];
  else
    return keys[i];
}

// ---- per-workgroup digit histogram -----------------------------------------
template <bool FIRST>
__global__ __launch_bounds__(SORT_THREADS) void radix_hist_kernel(
    const int64_t* __restrict__ idx, const uint32_t* __restrict__ keys_in,
    int64_t n, int shift, int bits, uint32_t* __restrict__ blockhist,
    int nblocks) {
  __shared__ uint32_t h[MAX_BINS];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * SORT_TILE;
  const uint32_t mask = (1u << bits) - 1u;
#pragma unroll
  for (int k = 0; k < SORT_ITEMS; ++k) {
    int64_t i = base + (int64_t)k * SORT_THREADS + threadIdx.x;
    if (i < n) {
      uint32_t d = (load_key<FIRST>(idx, keys_in, i) >> shift) & mask;
      atomicAdd(&h[d], 1u);
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < (1 << bits))
    blockhist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// ---- stable scatter ----------------------------------------------------------
// Order inside a tile is (wave, round, lane) == ascending input position, so
// equal digits keep their relative order: the sort is stable.
template <bool FIRST>
__global__ __launch_bounds__(SORT_THREADS) void radix_scatter_kernel(
    const int64_t* __restrict__ idx, const uint32_t* __restrict__ keys_in,
    const uint32_t* __restrict__ vals_in, int64_t n, int shift, int bits,
    const uint32_t* __restrict__ blockoff, int nblocks,
    uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
  __shared__ uint32_t cnt[SORT_WAVES][MAX_BINS];
  const int w = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  for (int j = threadIdx.x; j < SORT_WAVES * MAX_BINS; j += SORT_THREADS)
    (&cnt[0][0])[j] = 0;
  __syncthreads();

  const int64_t wbase = (int64_t)blockIdx.x * SORT_TILE + (int64_t)w * WAVE_CHUNK;
  const uint32_t mask = (1u << bits) - 1u;
  const uint64_t lt = lanemask_lt();

  uint32_t key[SORT_ITEMS];
  uint32_t rank[SORT_ITEMS];
#pragma unroll
  for (int r = 0; r < SORT_ITEMS; ++r) {
    int64_t i = wbase + r * 64 + lane;
    key[r] = (i < n) ? load_key<FIRST>(idx, keys_in, i) : 0xffffffffu;
  }
#pragma unroll
  for (int r = 0; r < SORT_ITEMS; ++r) {
    const int64_t i = wbase + r * 64 + lane;
    const bool valid = i < n;
    const uint32_t d = (key[r] >> shift) & mask;
    uint64_t peers = __ballot(valid);
    for (int b = 0; b < bits; ++b) {
      const bool bit = (d >> b) & 1u;
      const uint64_t m = __ballot(valid && bit);
      peers &= bit ? m : ~m;
    }
    const uint32_t pre = cnt[w][d];
    const uint32_t rk = __popcll(peers & lt);
    const uint32_t tot = __popcll(peers);
    rank[r] = pre + rk;
    __builtin_amdgcn_wave_barrier();
    if (valid && rk == tot - 1) cnt[w][d] = pre + tot;  // highest peer lane
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  if ((int)threadIdx.x < (1 << bits)) {
    const int d = threadIdx.x;
    uint32_t base = blockoff[(size_t)d * nblocks + blockIdx.x];
#pragma unroll
    for (int w2 = 0; w2 < SORT_WAVES; ++w2) {
      uint32_t t = cnt[w2][d];
      cnt[w2][d] = base;
      base += t;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < SORT_ITEMS; ++r) {
    const int64_t i = wbase + r * 64 + lane;
    if (i < n) {
      const uint32_t d = (key[r] >> shift) & mask;
      const uint32_t dst = cnt[w][d] + rank[r];
      keys_out[dst] = key[r];
      if constexpr (FIRST)
        vals_out[dst] = (uint32_t)i;
      else
        vals_out[dst] = vals_in[i];
    }
  }
}

// ---- device-wide exclusive scan (3 launches) -------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v,
                                                         uint32_t* total) {
  __shared__ uint32_t wsum[SCAN_THREADS / 64];
  const int w = threadIdx.x >> 6;
  const uint32_t inc = wave_inclusive_scan(v);
  __syncthreads();  // protects wsum reuse across calls
  if ((threadIdx.x & 63) == 63) wsum[w] = inc;
  __syncthreads();
  uint32_t off = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < SCAN_THREADS / 64; ++k) {
    uint32_t s = wsum[k];
    if (k < w) off += s;
    tot += s;
  }
  if (total) *total = tot;
  return off + inc - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(
    const uint32_t* __restrict__ in, int64_t m, uint32_t* __restrict__ partial) {
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k)
    if (base + k < m) s += in[base + k];
  uint32_t tot;
  block_exclusive_scan(s, &tot);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// single workgroup: exclusive scan of the chunk totals, carry across rounds
__global__ __launch_bounds__(SCAN_THREADS) void scan_partials_kernel(
    uint32_t* __restrict__ partial, int nchunks) {
  uint32_t carry = 0;
  for (int base = 0; base < nchunks; base += SCAN_THREADS) {
    const int i = base + threadIdx.x;
    const uint32_t v = (i < nchunks) ? partial[i] : 0u;
    uint32_t tot;
    const uint32_t ex = block_exclusive_scan(v, &tot);
    if (i < nchunks) partial[i] = carry + ex;
    carry += tot;
  }
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_kernel(
    uint32_t* __restrict__ data, int64_t m, const uint32_t* __restrict__ partial) {
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    v[k] = (base + k < m) ? data[base + k] : 0u;
    s += v[k];
  }
  uint32_t run = block_exclusive_scan(s, nullptr) + partial[blockIdx.x];
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    if (base + k < m) data[base + k] = run;
    run += v[k];
  }
}

// ---- boundaries of the sorted key array -> rowptr ----------------------------
__global__ void rowptr_from_sorted_kernel(const uint32_t* __restrict__ skeys,
                                          int64_t n, int64_t num_seg,
                                          int32_t* __restrict__ rowptr) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += stride) {
    int64_t lo, hi;  // rowptr[lo+1 .. hi] = i
    if (i == n) {
      lo = (n > 0) ? (int64_t)skeys[n - 1] : -1;
      hi = num_seg;
    } else {
      hi = (int64_t)skeys[i];
      lo = (i > 0) ? (int64_t)skeys[i - 1] : -1;
    }
    if (lo > num_seg - 1) lo = num_seg - 1;  // out-of-range keys: stay in bounds
    if (hi > num_seg) hi = num_seg;
    if (i < n && hi > num_seg - 1) hi = num_seg - 1;
    for (int64_t s = lo + 1; s <= hi; ++s) rowptr[s] = (int32_t)i;
  }
}

__global__ void iota_kernel(int32_t* __restrict__ perm, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    perm[i] = (int32_t)i;
}

__global__ void rowptr_single_kernel(int32_t* rowptr, int64_t n) {
  rowptr[0] = 0;
  rowptr[1] = (int32_t)n;
}

static int bits_for(int64_t num_seg) {
  int b = 1;
  while (((int64_t)1 << b) < num_seg) ++b;
  return b;
}

struct SortPlan {
  int64_t n;
  int nblocks, nbits, passes;
  int64_t hist_len;  // 256 * nblocks (upper bound)
  int nchunks;
  size_t off_keys0, off_keys1, off_vals, off_hist, off_part, total;
};

static SortPlan make_plan(int64_t n, int64_t num_seg) {
  SortPlan p;
  p.n = n;
  p.nblocks = (int)ceil_div(n > 0 ? n : 1, SORT_TILE);
  p.nbits = bits_for(num_seg);
  p.passes = (num_seg <= 1) ? 0 : (p.nbits + 7) / 8;
  p.hist_len = (int64_t)MAX_BINS * p.nblocks;
  p.nchunks = (int)ceil_div(p.hist_len, SCAN_TILE);
  size_t o = 0;
  const size_t nb = align_up((size_t)(n > 0 ? n : 1) * 4, 256);
  p.off_keys0 = o; o += nb;
  p.off_keys1 = o; o += nb;
  p.off_vals = o;  o += nb;
  p.off_hist = o;  o += align_up((size_t)p.hist_len * 4, 256);
  p.off_part = o;  o += align_up((size_t)p.nchunks * 4, 256);
  p.total = o;
  return p;
}

}  // namespace spt

using namespace spt;

extern "C" size_t spt_csr_build_workspace_bytes(int64_t n, int64_t num_seg) {
  if (n < 0 || num_seg < 1) return 0;
  return make_plan(n, num_seg).total;
}

extern "C" int spt_csr_build(const int64_t* idx, int64_t n, int64_t num_seg,
                             int32_t* perm, int32_t* rowptr, void* ws,
                             size_t ws_bytes, spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(n >= 0 && n < ((int64_t)1 << 31) - SORT_TILE, "n out of range");
  SPT_CHECK_ARG(num_seg >= 1 && num_seg < ((int64_t)1 << 31), "num_seg out of range");
  SPT_CHECK_ARG(rowptr != nullptr, "rowptr is null");
  SPT_CHECK_ARG(n == 0 || (idx && perm), "null idx/perm");
  const SortPlan p = make_plan(n, num_seg);
  SPT_CHECK_ARG(ws_bytes >= p.total && (ws || p.total == 0), "workspace too small");

  if (n == 0) {
    rowptr_from_sorted_kernel<<<stream_grid(num_seg + 1, 256), 256, 0, stream>>>(
        nullptr, 0, num_seg, rowptr);
    SPT_CHECK_LAUNCH();
    return 0;
  }
  if (p.passes == 0) {
    iota_kernel<<<stream_grid(n, 256), 256, 0, stream>>>(perm, n);
    rowptr_single_kernel<<<1, 1, 0, stream>>>(rowptr, n);
    SPT_CHECK_LAUNCH();
    return 0;
  }
  char* base = (char*)ws;
  uint32_t* kbuf[2] = {(uint32_t*)(base + p.off_keys0), (uint32_t*)(base + p.off_keys1)};
  uint32_t* vtmp = (uint32_t*)(base + p.off_vals);
  uint32_t* hist = (uint32_t*)(base + p.off_hist);
  uint32_t* part = (uint32_t*)(base + p.off_part);

  // balanced digit widths, e.g. 19 bits -> 7+6+6
  const int per = (p.nbits + p.passes - 1) / p.passes;
  const uint32_t* kin = nullptr;
  const uint32_t* vin = nullptr;
  int shift = 0;
  for (int pass = 0; pass < p.passes; ++pass) {
    const int bits = (p.nbits - shift < per) ? (p.nbits - shift) : per;
    uint32_t* kout = kbuf[pass & 1];
    // value ping-pong arranged so that the LAST pass lands in perm
    uint32_t* vout = (((p.passes - 1 - pass) & 1) == 0) ? (uint32_t*)perm : vtmp;
    const int64_t hl = (int64_t)(1 << bits) * p.nblocks;
    const int nch = (int)ceil_div(hl, SCAN_TILE);
    if (pass == 0)
      radix_hist_kernel<true><<<p.nblocks, SORT_THREADS, 0, stream>>>(
          idx, nullptr, n, shift, bits, hist, p.nblocks);
    else
      radix_hist_kernel<false><<<p.nblocks, SORT_THREADS, 0, stream>>>(
          nullptr, kin, n, shift, bits, hist, p.nblocks);
    scan_reduce_kernel<<<nch, SCAN_THREADS, 0, stream>>>(hist, hl, part);
    scan_partials_kernel<<<1, SCAN_THREADS, 0, stream>>>(part, nch);
    scan_apply_kernel<<<nch, SCAN_THREADS, 0, stream>>>(hist, hl, part);
    if (pass == 0)
      radix_scatter_kernel<true><<<p.nblocks, SORT_THREADS, 0, stream>>>(
          idx, nullptr, nullptr, n, shift, bits, hist, p.nblocks, kout, vout);
    else
      radix_scatter_kernel<false><<<p.nblocks, SORT_THREADS, 0, stream>>>(
          nullptr, kin, vin, n, shift, bits, hist, p.nblocks, kout, vout);
    kin = kout;
    vin = vout;
    shift += bits;
  }
  rowptr_from_sorted_kernel<<<stream_grid(n + 1, 256), 256, 0, stream>>>(
      kin, n, num_seg, rowptr);
  SPT_CHECK_LAUNCH();
  return 0;
}
