// Stable LSD radix sort of (u32 key, u32 value) pairs + device-wide u32 scan,
// shared by the CSR builder (csr_build.hip) and the segment sampler
// (sampling.hip).  gfx950 only: wave64 ballots rank equal digits inside a wave,
// 4-wave workgroups, LDS digit counters, <= 8 bits per pass.
//
// First-pass source (template MODE of the hist / scatter kernels):
//   0  u32 keys + u32 values            (every later pass)
//   1  int64 index array, values = iota (CSR build)
//   2  u32 keys, values = iota
#pragma once
#include "common.hpp"

namespace spt {


constexpr int SORT_THREADS = 256;
constexpr int SORT_WAVES = SORT_THREADS / 64;
constexpr int SORT_ITEMS = 16;                        // keys per lane
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;  // 4096 keys per workgroup
constexpr int WAVE_CHUNK = SORT_TILE / SORT_WAVES;    // 1024 consecutive keys per wave
constexpr int MAX_BINS = 256;

template <int MODE>
__device__ __forceinline__ uint32_t load_key(const int64_t* __restrict__ idx,
                                             const uint32_t* __restrict__ keys,
                                             int64_t i) {
  if constexpr (MODE == 1)
    return (uint32_t)idx[i];
  else
    return keys[i];
}

// ---- per-workgroup digit histogram -----------------------------------------
template <int MODE>
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
      uint32_t d = (load_key<MODE>(idx, keys_in, i) >> shift) & mask;
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
template <int MODE>
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
    key[r] = (i < n) ? load_key<MODE>(idx, keys_in, i) : 0xffffffffu;
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
      if constexpr (MODE != 0)
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

static __global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(
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
static __global__ __launch_bounds__(SCAN_THREADS) void scan_partials_kernel(
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

static __global__ __launch_bounds__(SCAN_THREADS) void scan_apply_kernel(
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
// rowptr[s] = first position of a key >= s.  Element i owns the entries (skeys[i-1], skeys[i]]:
// one for a dense id space, but a sparse one (the cells of a kNN grid over surfaces: most of the
// volume is empty) has runs of thousands of empty ids between two occupied ones - a lane filling
// its run alone kept the whole launch waiting (5.4 ms for 29 M cells / 12 M points).  Runs longer
// than 4 entries are filled by the whole wave, 64 consecutive entries per store.
static __global__ void rowptr_from_sorted_kernel(const uint32_t* __restrict__ skeys,
                                          int64_t n, int64_t num_seg,
                                          int32_t* __restrict__ rowptr) {
  const int lane = threadIdx.x & 63;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t base = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63); base <= n;
       base += stride) {                                  // (wave-uniform trip count)
    const int64_t i = base + lane;
    int64_t lo = 0, hi = -1;  // rowptr[lo+1 .. hi] = i
    if (i <= n) {
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
    }
    const bool longrun = hi - lo > 4;
    if (!longrun)
      for (int64_t s = lo + 1; s <= hi; ++s) rowptr[s] = (int32_t)i;
    uint64_t todo = __ballot(longrun);
    while (todo) {
      const int src = __ffsll((unsigned long long)todo) - 1;
      todo &= todo - 1;
      const int64_t l2 = __shfl(lo, src, 64), h2 = __shfl(hi, src, 64);
      const int32_t v = (int32_t)(base + src);
      for (int64_t s = l2 + 1 + lane; s <= h2; s += 64) rowptr[s] = v;
    }
  }
}

static __global__ void iota_kernel(int32_t* __restrict__ perm, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    perm[i] = (int32_t)i;
}

static __global__ void rowptr_single_kernel(int32_t* rowptr, int64_t n) {
  rowptr[0] = 0;
  rowptr[1] = (int32_t)n;
}


static inline int bits_for(int64_t num_seg) {
  int b = 1;
  while (((int64_t)1 << b) < num_seg) ++b;
  return b;
}

// Scratch of one sort: two key buffers, two value buffers, per-workgroup digit
// histograms and scan partials.  `bytes(n)` is what `carve` consumes.
struct RadixScratch {
  uint32_t *k0, *k1, *v0, *v1, *hist, *part;
  static size_t bytes(int64_t n) {
    const int64_t m = n > 0 ? n : 1;
    const int nblocks = (int)ceil_div(m, SORT_TILE);
    const int64_t hist_len = (int64_t)MAX_BINS * nblocks;
    return 4 * align_up((size_t)m * 4, 256) + align_up((size_t)hist_len * 4, 256) +
           align_up((size_t)ceil_div(hist_len, SCAN_TILE) * 4, 256);
  }
  char* carve(char* base, int64_t n) {
    const int64_t m = n > 0 ? n : 1;
    const size_t nb = align_up((size_t)m * 4, 256);
    const int nblocks = (int)ceil_div(m, SORT_TILE);
    const int64_t hist_len = (int64_t)MAX_BINS * nblocks;
    k0 = (uint32_t*)base; base += nb;
    k1 = (uint32_t*)base; base += nb;
    v0 = (uint32_t*)base; base += nb;
    v1 = (uint32_t*)base; base += nb;
    hist = (uint32_t*)base; base += align_up((size_t)hist_len * 4, 256);
    part = (uint32_t*)base; base += align_up((size_t)ceil_div(hist_len, SCAN_TILE) * 4, 256);
    return base;
  }
};

// In-place exclusive scan of m u32 values (3 launches).
static inline void device_exclusive_scan(uint32_t* data, int64_t m, uint32_t* part,
                                         hipStream_t stream) {
  const int nch = (int)ceil_div(m > 0 ? m : 1, SCAN_TILE);
  scan_reduce_kernel<<<nch, SCAN_THREADS, 0, stream>>>(data, m, part);
  scan_partials_kernel<<<1, SCAN_THREADS, 0, stream>>>(part, nch);
  scan_apply_kernel<<<nch, SCAN_THREADS, 0, stream>>>(data, m, part);
}

// Sort n pairs by the low `nbits` bits of the key.  MODE selects the first-pass
// source (see top).  The last pass writes its values to `final_vals` when given
// (else into the scratch); returns the sorted keys / values through the out
// pointers.  nbits >= 1, n >= 1.
template <int MODE>
static inline void radix_sort_pairs(const int64_t* idx64, const uint32_t* keys32,
                                    const uint32_t* vals32, int64_t n, int nbits,
                                    RadixScratch& s, uint32_t* final_vals,
                                    const uint32_t** keys_sorted,
                                    const uint32_t** vals_sorted, hipStream_t stream) {
  const int nblocks = (int)ceil_div(n, SORT_TILE);
  const int passes = (nbits + 7) / 8;
  const int per = (nbits + passes - 1) / passes;   // balanced digits, e.g. 19 -> 7+6+6
  uint32_t* kbuf[2] = {s.k0, s.k1};
  uint32_t* vbuf[2] = {s.v0, s.v1};
  const uint32_t* kin = keys32;
  const uint32_t* vin = vals32;
  int shift = 0;
  for (int pass = 0; pass < passes; ++pass) {
    const int bits = (nbits - shift < per) ? (nbits - shift) : per;
    uint32_t* kout = kbuf[pass & 1];
    uint32_t* vout = vbuf[pass & 1];
    if (pass == passes - 1 && final_vals) vout = final_vals;
    const int64_t hl = (int64_t)(1 << bits) * nblocks;
    if (pass == 0) {
      radix_hist_kernel<MODE><<<nblocks, SORT_THREADS, 0, stream>>>(
          idx64, kin, n, shift, bits, s.hist, nblocks);
      device_exclusive_scan(s.hist, hl, s.part, stream);
      radix_scatter_kernel<MODE><<<nblocks, SORT_THREADS, 0, stream>>>(
          idx64, kin, vin, n, shift, bits, s.hist, nblocks, kout, vout);
    } else {
      radix_hist_kernel<0><<<nblocks, SORT_THREADS, 0, stream>>>(
          nullptr, kin, n, shift, bits, s.hist, nblocks);
      device_exclusive_scan(s.hist, hl, s.part, stream);
      radix_scatter_kernel<0><<<nblocks, SORT_THREADS, 0, stream>>>(
          nullptr, kin, vin, n, shift, bits, s.hist, nblocks, kout, vout);
    }
    kin = kout;
    vin = vout;
    shift += bits;
  }
  *keys_sorted = kin;
  *vals_sorted = vin;
}

}  // namespace spt
