// Radius-bounded exact kNN on a uniform grid (replaces the FRNN CUDA extension
// behind src/utils/neighbors.py:24-48 <- knn_1 :51-123, knn_2 :186-242).
//
// Contract (bit-exact, see oracle/spt_oracle.py:frnn_grid_points): for every
// query the K search points of smallest squared distance d2 < r^2, ascending by
// (d2, index); d2 = (dx*dx + dy*dy) + dz*dz in f32 without fma; missing
// neighbours are index -1 / distance -1.
//
// FRNN scans every point of the 27..125 cells of size r/2 around a query: at
// the reference's settings (r = 2 m on 3 cm voxels, r = 10 m on 10 cm voxels)
// that is 1e4..1e5 candidates per query for K = 46 results.  Here the grid is
// FINE (a few points per cell), points are counting-sorted by cell, and one
// wave per query walks the cells in rings of growing Chebyshev radius, stopping
// as soon as the K-th best distance is below the radius the finished rings
// guarantee - typically ~200 candidates.  Candidates are (d2,index) packed in
// one u64 (order = lexicographic = the tie rule); accepted ones are compacted
// into a 64-slot pending buffer and merged into the wave-resident sorted list
// with a 64-lane bitonic network only when the buffer fills.
#include <math.h>

#include "common.hpp"
#include "geof_core.hpp"

namespace spt {

constexpr uint64_t KNN_EMPTY = ~0ull;
constexpr int KNN_WAVES = 4;

struct Grid {
  float ox, oy, oz, inv_s, s;
  int dx, dy, dz;
};

__device__ __forceinline__ int cell_coord(float p, float o, float inv_s) {
  return (int)floorf((p - o) * inv_s);
}

__global__ void knn_cell_ids_kernel(const float* __restrict__ pos, int64_t n, Grid g,
                                    int64_t* __restrict__ cell) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int x = cell_coord(pos[i * 3 + 0], g.ox, g.inv_s);
    int y = cell_coord(pos[i * 3 + 1], g.oy, g.inv_s);
    int z = cell_coord(pos[i * 3 + 2], g.oz, g.inv_s);
    x = x < 0 ? 0 : (x >= g.dx ? g.dx - 1 : x);
    y = y < 0 ? 0 : (y >= g.dy ? g.dy - 1 : y);
    z = z < 0 ? 0 : (z >= g.dz ? g.dz - 1 : z);
    cell[i] = ((int64_t)z * g.dy + y) * g.dx + x;
  }
}

// positions in cell order, original index in .w
__global__ void knn_gather_sorted_kernel(const float* __restrict__ pos,
                                         const int32_t* __restrict__ perm, int64_t n,
                                         float4* __restrict__ sorted) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
    const int32_t i = perm[j];
    sorted[j] = make_float4(pos[(int64_t)i * 3], pos[(int64_t)i * 3 + 1],
                            pos[(int64_t)i * 3 + 2], __int_as_float(i));
  }
}

// lane ^ J exchange without the LDS crossbar: DPP row operations inside a 16-lane row (every lane
// has a source: no `old` value, no register initialisation), gfx950's v_permlane16_swap /
// v_permlane32_swap across rows.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_all(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}
template <int J>
__device__ __forceinline__ uint32_t xor_lane_u32(uint32_t v) {
  if constexpr (J == 1) {
    return dpp_all<0xB1>(v);                      // quad [1,0,3,2]
  } else if constexpr (J == 2) {
    return dpp_all<0x4E>(v);                      // quad [2,3,0,1]
  } else if constexpr (J == 4) {
    return dpp_all<0x141>(dpp_all<0x1B>(v));      // ^3 (quad [3,2,1,0]) then ^7 (row_half_mirror)
  } else if constexpr (J == 8) {
    return dpp_all<0x128>(v);                     // row_ror:8
  } else if constexpr (J == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);   // {[v0 v0 v2 v2], [v1 v1 v3 v3]}
    return (threadIdx.x & 16) ? r[0] : r[1];
  } else {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);   // {[lo lo], [hi hi]}
    return (threadIdx.x & 32) ? r[0] : r[1];
  }
}
template <int J>
__device__ __forceinline__ uint64_t xor_lane_u64(uint64_t v) {
  return ((uint64_t)xor_lane_u32<J>((uint32_t)(v >> 32)) << 32) | xor_lane_u32<J>((uint32_t)v);
}

__device__ __forceinline__ uint64_t row_mirror_u64(uint64_t v) {
  return ((uint64_t)dpp_all<0x140>((uint32_t)(v >> 32)) << 32) | dpp_all<0x140>((uint32_t)v);
}
// value of a wave-uniform lane index (v_readlane, no LDS permute)
__device__ __forceinline__ uint64_t read_lane_u64(uint64_t v, int src) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
  return ((uint64_t)hi << 32) | lo;
}

// One compare-exchange of the bitonic network: the lane keeps min or max of (its key, the key of
// lane ^ J).  ONE 64-bit compare; which side a lane keeps is a compile-time lane mask combined
// with the compare's mask on the scalar unit: 5 vector instructions per step inside a row (two
// DPP moves, the compare, two selects), 7 for J = 4 and across rows.
//   across rows: the swap instruction leaves (A, B) = (key, partner) in even rows / row halves and
//   (partner, key) in odd ones, and the lane's result is `(A < B) == wants_min ? A : B` in both
//   (an odd lane that wants the minimum takes A = its partner exactly when A < B).
template <int K, int J>
__device__ __forceinline__ uint64_t bitonic_step(uint64_t key, int lane) {
  const bool want_min = ((lane & K) == 0) == ((lane & J) == 0);
  if constexpr (J >= 16) {
    const uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
    uint64_t A, B;
    if constexpr (J == 16) {
      const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
      const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
      A = ((uint64_t)rh[0] << 32) | rl[0];
      B = ((uint64_t)rh[1] << 32) | rl[1];
    } else {
      const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
      const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
      A = ((uint64_t)rh[0] << 32) | rl[0];
      B = ((uint64_t)rh[1] << 32) | rl[1];
    }
    return ((A < B) == want_min) ? A : B;
  } else {
    const uint64_t other = xor_lane_u64<J>(key);
    return ((key < other) == want_min) ? key : other;
  }
}
template <int K, int J>
__device__ __forceinline__ uint64_t bitonic_merge_steps(uint64_t key, int lane) {
  key = bitonic_step<K, J>(key, lane);
  if constexpr (J > 1) key = bitonic_merge_steps<K, J / 2>(key, lane);
  return key;
}

// ascending bitonic sort of one key per lane
__device__ __forceinline__ uint64_t wave_sort(uint64_t key, int lane) {
  key = bitonic_merge_steps<2, 1>(key, lane);
  key = bitonic_merge_steps<4, 2>(key, lane);
  key = bitonic_merge_steps<8, 4>(key, lane);
  key = bitonic_merge_steps<16, 8>(key, lane);
  key = bitonic_merge_steps<32, 16>(key, lane);
  key = bitonic_merge_steps<64, 32>(key, lane);   // lane & 64 == 0: ascending everywhere
  return key;
}

// two independent ascending sorts, lanes 0..31 and lanes 32..63 (15 compare-exchange steps
// instead of 21, and two lists per pass): the 16-blocks alternate direction as in the full network,
// the last merge runs ascending in both halves
__device__ __forceinline__ uint64_t wave_sort_halves(uint64_t key, int lane) {
  key = bitonic_merge_steps<2, 1>(key, lane);
  key = bitonic_merge_steps<4, 2>(key, lane);
  key = bitonic_merge_steps<8, 4>(key, lane);
  key = bitonic_merge_steps<16, 8>(key, lane);
  key = bitonic_merge_steps<64, 16>(key, lane);   // partners stay inside a half; ascending in both
  return key;
}

// merge two ascending 64-lists, keep the 64 smallest, ascending
__device__ __forceinline__ uint64_t wave_merge(uint64_t best, uint64_t cand_sorted, int lane) {
  // lane 63 - l = l ^ 63: mirror inside the rows (^15), then swap rows (^16) and halves (^32)
  const uint64_t rev = xor_lane_u64<32>(xor_lane_u64<16>(row_mirror_u64(cand_sorted)));
  const uint64_t key = best < rev ? best : rev;  // bitonic
  return bitonic_merge_steps<64, 32>(key, lane);
}

struct KnnState {
  uint64_t best;      // lane i: i-th smallest key so far
  uint64_t kth;       // key of rank K-1 (uniform)
  int npend;          // pending accepted candidates in LDS (uniform)
};

__device__ __forceinline__ void knn_flush(KnnState& st, uint64_t* pend, int K, int lane) {
  if (st.npend == 0) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  uint64_t c = lane < st.npend ? pend[lane] : KNN_EMPTY;
  c = wave_sort(c, lane);
  st.best = wave_merge(st.best, c, lane);
  st.kth = read_lane_u64(st.best, K - 1);
  st.npend = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// scan sorted points [start, start+len) against the query
__device__ __forceinline__ void knn_scan(KnnState& st, uint64_t* pend,
                                         const float4* __restrict__ sorted, int start,
                                         int len, float qx, float qy, float qz, float r2,
                                         bool inclusive, int K, int lane, uint64_t lb = 0,
                                         bool has_lb = false) {
  for (int b = 0; b < len; b += 64) {
    const int j = b + lane;
    uint64_t key = KNN_EMPTY;
    if (j < len) {
      const float4 p = sorted[start + j];
      const float ddx = qx - p.x, ddy = qy - p.y, ddz = qz - p.z;
      const float d2 = (ddx * ddx + ddy * ddy) + ddz * ddz;  // -ffp-contract=off: no fma
      const bool in = inclusive ? (d2 <= r2) : (d2 < r2);
      if (in) key = ((uint64_t)__float_as_uint(d2) << 32) | (uint32_t)__float_as_int(p.w);
      // continuation search: only neighbours strictly after (d2, index) = lb
      if (has_lb && key <= lb) key = KNN_EMPTY;
    }
    const bool acc = key < st.kth;
    const uint64_t m = __ballot(acc);
    if (m == 0) continue;
    const int n = __popcll(m);
    if (st.npend + n > 64) knn_flush(st, pend, K, lane);
    // the flush may have tightened kth: re-test so the buffer never overflows uselessly
    const bool acc2 = key < st.kth;
    const uint64_t m2 = __ballot(acc2);
    if (acc2) pend[st.npend + __popcll(m2 & lanemask_lt())] = key;
    st.npend += __popcll(m2);
  }
}

__global__ __launch_bounds__(KNN_WAVES * 64) void knn_search_kernel(
    const float* __restrict__ query, int64_t nq, const int32_t* __restrict__ qorder,
    const float4* __restrict__ sorted, const int32_t* __restrict__ rowptr, Grid g, int K,
    float r, int inclusive, int squared, int64_t* __restrict__ out_idx,
    float* __restrict__ out_dist, const int32_t* __restrict__ nq_dev,
    const int64_t* __restrict__ after_idx = nullptr,
    const float* __restrict__ after_d2 = nullptr) {
  __shared__ uint64_t pend_all[KNN_WAVES][64];
  if (nq_dev) nq = *nq_dev;          // leftovers of knn_cell_kernel: the count lives on the device
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  uint64_t* pend = pend_all[wid];
  const float r2 = r * r;
  const int64_t wave = (int64_t)blockIdx.x * KNN_WAVES + wid;
  const int64_t nwaves = (int64_t)gridDim.x * KNN_WAVES;

  for (int64_t w = wave; w < nq; w += nwaves) {
    const int64_t qi = qorder ? qorder[w] : w;
    const float qx = query[qi * 3], qy = query[qi * 3 + 1], qz = query[qi * 3 + 2];
    const int cx = cell_coord(qx, g.ox, g.inv_s);
    const int cy = cell_coord(qy, g.oy, g.inv_s);
    const int cz = cell_coord(qz, g.oz, g.inv_s);
    KnnState st;
    st.best = KNN_EMPTY;
    st.kth = KNN_EMPTY;
    st.npend = 0;
    // continuation (spt_grid_knn_after_f32): neighbours ranked strictly after the given one;
    // a query whose previous list was not full (index -1) has nothing left
    const bool has_lb = after_idx != nullptr;
    uint64_t lb = 0;
    if (has_lb) {
      const int64_t ai = after_idx[qi];
      lb = ai < 0 ? KNN_EMPTY - 1
                  : (((uint64_t)__float_as_uint(after_d2[qi]) << 32) | (uint32_t)ai);
    }

    for (int rho = 0;; ++rho) {
      if (rho > 1) {
        // Rings 0..rho-1 are done: an unexplored point sits in a cell at
        // Chebyshev distance >= rho, i.e. more than (rho-1) whole cells away along
        // one axis, so it is farther than (rho - 1 - 0.01) s  (0.01 cell of slack
        // covers the f32 rounding of the cell coordinates).
        const float gr = ((float)(rho - 1) - 0.01f) * g.s;
        const float g2 = gr * gr;
        if (g2 >= r2) break;                       // nothing within r is left
        knn_flush(st, pend, K, lane);              // exact K-th distance so far
        if (st.kth != KNN_EMPTY && __uint_as_float((uint32_t)(st.kth >> 32)) <= g2) break;
      }
      if (rho > 0) {
        const int in = rho - 1;                    // cube of finished rings covers the grid?
        if (cx - in <= 0 && cx + in >= g.dx - 1 && cy - in <= 0 && cy + in >= g.dy - 1 &&
            cz - in <= 0 && cz + in >= g.dz - 1)
          break;
      }
      const int side = 2 * rho + 1;
      const int nrows = side * side;
      for (int base = 0; base < nrows; base += 64) {
        const int row = base + lane;
        int sA = 0, lA = 0, sB = 0, lB = 0;
        if (row < nrows) {
          const int dz = row / side - rho, dy = row % side - rho;
          const int z = cz + dz, y = cy + dy;
          if (z >= 0 && z < g.dz && y >= 0 && y < g.dy) {
            const int64_t rb = ((int64_t)z * g.dy + y) * g.dx;
            const bool face = (dz == rho || dz == -rho || dy == rho || dy == -rho);
            if (face) {
              int x0 = cx - rho, x1 = cx + rho;
              x0 = x0 < 0 ? 0 : x0;
              x1 = x1 >= g.dx ? g.dx - 1 : x1;
              if (x0 <= x1) {
                sA = rowptr[rb + x0];
                lA = rowptr[rb + x1 + 1] - sA;
              }
            } else {
              const int xa = cx - rho, xb = cx + rho;
              if (xa >= 0 && xa < g.dx) {
                sA = rowptr[rb + xa];
                lA = rowptr[rb + xa + 1] - sA;
              }
              if (xb >= 0 && xb < g.dx) {
                sB = rowptr[rb + xb];
                lB = rowptr[rb + xb + 1] - sB;
              }
            }
          }
        }
        uint64_t mA = __ballot(lA > 0);
        while (mA) {
          const int l = __ffsll((unsigned long long)mA) - 1;
          mA &= mA - 1;
          knn_scan(st, pend, sorted, __shfl(sA, l, 64), __shfl(lA, l, 64), qx, qy, qz, r2,
                   inclusive != 0, K, lane, lb, has_lb);
        }
        uint64_t mB = __ballot(lB > 0);
        while (mB) {
          const int l = __ffsll((unsigned long long)mB) - 1;
          mB &= mB - 1;
          knn_scan(st, pend, sorted, __shfl(sB, l, 64), __shfl(lB, l, 64), qx, qy, qz, r2,
                   inclusive != 0, K, lane, lb, has_lb);
        }
      }
    }
    knn_flush(st, pend, K, lane);
    if (lane < K) {
      const bool ok = st.best != KNN_EMPTY;
      float d = __uint_as_float((uint32_t)(st.best >> 32));
      if (!squared) d = sqrtf(d);
      out_idx[qi * K + lane] = ok ? (int64_t)(uint32_t)(st.best & 0xffffffffu) : -1;
      out_dist[qi * K + lane] = ok ? d : -1.0f;
    }
  }
}

// ---- self-search, cell-centric fast path ---------------------------------------------------
// The wave-per-query kernel above spends ~2 000 instructions per query: every query walks the
// same 27 cells as its cell mates, and every ~64 accepted candidates cost a 64-lane sort +
// merge.  For the preprocessing call (every point of a cloud searches the cloud itself,
// src/utils/neighbors.py:51-123) the queries ARE the cell-sorted search points, so here a wave
// owns 64 consecutive sorted points = 1-3 neighbouring cells of one grid row and shares the
// candidate stream among its lanes:
//   round : leader = first unfinished lane; group = lanes of the leader's row with cell x in
//           [lx, lx + KC_W); candidates = the 9 rows (dz, dy in -1..1), x in [lx-1, lx+KC_W]
//           - one contiguous range of `sorted` per row - staged 64 at a time through LDS and
//           read back as broadcasts: every lane evaluates ITS distance to the same candidate;
//   pass A: per-lane histogram (KC_NBINS bins of d2 over [0, B2], B2 = min(guaranteed radius^2
//           of ring 1, r^2)) -> smallest bin edge holding >= K candidates;
//   pass B: candidates at or below that bin are appended to the lane's list (<= 64 entries) -
//           one comparison per candidate against the largest d2 of that bin (found once per
//           round by one-ulp steps around bin edge / scale, so that it is EXACTLY pass A's set);
//   sort  : per query one 64-lane bitonic sort of its list, K results written.
// Round 4 halved the vector instruction count - candidates staged per coordinate (packed f32
// arithmetic on candidate pairs, no register shuffles), 5-7 instructions per compare-exchange of
// the sort network instead of 12-16, the row of a list entry by one table read instead of a
// 9-range search - and keeps four queries' candidate rows in flight per group of sorts: the sort
// stage went from 8.5 to 6.0 ms at scene S, the two candidate passes stayed at ~13 ms: they are
// bound by the LDS pipe (broadcast reads + per-lane atomics / stores), DESIGN.md 7.3 item 8.
// A lane whose ring-1 neighbourhood does not guarantee its K-th neighbour (fewer than K
// candidates within B2 while r reaches further; a bin so dense that the list would overflow)
// is appended to `todo` and finished by knn_search_kernel afterwards: same exact contract.
constexpr int KC_WAVES = 4;
constexpr int KC_NBINS = 16;
constexpr int KC_CAP = 64;
// measurement builds (tools/build_variant.sh ... -DSPT_KC_W=<cells> -DSPT_KC_MAXBLK=<blocks>)
#ifndef SPT_KC_W
#define SPT_KC_W 3
#endif
#ifndef SPT_KC_G
#define SPT_KC_G 4
#endif
#ifndef SPT_KC_MASKED_ADD
#define SPT_KC_MASKED_ADD 0
#endif
#ifndef SPT_KC_SKIP     /* 1: no sort stage, 2: no pass B and no sort, 4: no distance work in pass A */
#define SPT_KC_SKIP 0
#endif
#ifndef SPT_KC_MAXBLK
#define SPT_KC_MAXBLK 256
#endif
constexpr int KC_W = SPT_KC_W;
// 64-candidate blocks per round (16 384 candidates; denser rounds go to the wave-per-query kernel).
// A wave's LDS is 14.3 KB: two workgroups per CU - three (a shorter table) measured 3 % slower,
// the candidate passes are bound by the LDS pipe (broadcast reads + histogram atomics), not by
// latency
constexpr int KC_MAXBLK = SPT_KC_MAXBLK;

struct KcLds {
  uint32_t hist[KC_NBINS][64];
  uint16_t list[KC_CAP + 1][64];   // + 1: the branch-free append always stores, then maybe advances
  // the staged candidates, one array per coordinate: a 16-byte broadcast read hands every lane the
  // same coordinate of FOUR candidates in adjacent registers - the operand pairs of the packed f32
  // instructions as they come, no register shuffles in front of the distance arithmetic
  float sx[64], sy[64], sz[64];
  // the round's candidate stream as blocks of <= 64 consecutive positions of the cell order: first
  // position and length; a list entry is 64 * block + slot, a candidate's row one table read away
  int32_t blk[KC_MAXBLK];
  uint8_t blkm[KC_MAXBLK];
};

__device__ __forceinline__ void kc_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int kc_clamped_cell(float p, float o, float inv_s, int d) {
  int c = cell_coord(p, o, inv_s);
  return c < 0 ? 0 : (c >= d ? d - 1 : c);
}

__device__ __forceinline__ float kc_d2(const float4& q, const float4& p) {
  const float ddx = q.x - p.x, ddy = q.y - p.y, ddz = q.z - p.z;
  return (ddx * ddx + ddy * ddy) + ddz * ddz;       // the contract's rounding order
}

// squared distances of the query to staged candidates t .. t + 3 (the contract's rounding order),
// two candidates per packed f32 instruction
typedef float kc_f32x2 __attribute__((ext_vector_type(2)));
typedef float kc_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void kc_d2x4(const KcLds& L, int t, const float4& q, float (&d2)[4]) {
  const kc_f32x4 X = *reinterpret_cast<const kc_f32x4*>(&L.sx[t]);
  const kc_f32x4 Y = *reinterpret_cast<const kc_f32x4*>(&L.sy[t]);
  const kc_f32x4 Z = *reinterpret_cast<const kc_f32x4*>(&L.sz[t]);
  const kc_f32x2 qx = {q.x, q.x}, qy = {q.y, q.y}, qz = {q.z, q.z};
  const kc_f32x2 ax = qx - X.xy, ay = qy - Y.xy, az = qz - Z.xy;
  const kc_f32x2 bx = qx - X.zw, by = qy - Y.zw, bz = qz - Z.zw;
  const kc_f32x2 a = (ax * ax + ay * ay) + az * az;
  const kc_f32x2 b = (bx * bx + by * by) + bz * bz;
  d2[0] = a.x;
  d2[1] = a.y;
  d2[2] = b.x;
  d2[3] = b.y;
}

// GEOF: the eigenfeatures of every finished query's neighbourhood (the K winners = the point
// itself + its K - 1 nearest: what geometric_features(xyz, knn_1(xyz, K - 1)) describes) leave the
// kernel with the neighbour lists.  The winners' rows were gathered for the sort a moment ago; a
// query's lane walks ITS list once more (rows from L1 / L2), keeps the entries at or below the
// K-th key the sort found, and sums their moments about the query in f64 - the sums
// point_geof_dense_kernel forms from the stored index rows, without the 8 (K - 1)-byte index row
// and the K - 1 position gathers per point coming back from HBM.  One Jacobi per lane at the end
// of the unit.
template <bool GEOF>
__global__ __launch_bounds__(KC_WAVES * 64) void knn_cell_kernel(
    const float4* __restrict__ sorted, int64_t ns, const int32_t* __restrict__ rowptr, Grid g,
    int K, float r, int inclusive, int squared, int64_t* __restrict__ out_idx,
    float* __restrict__ out_dist, int32_t* __restrict__ todo, int32_t* __restrict__ todo_count,
    int k_min, int post, float* __restrict__ feats) {
  __shared__ KcLds lds_all[KC_WAVES];
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  KcLds& L = lds_all[wid];
  const float r2 = r * r;
  const float gr = (1.0f - 0.01f) * g.s;   // ring 1 done: unexplored points are farther than this
  const float g2 = gr * gr;
  const bool ring_covers_r = g2 >= r2;
  const float B2 = ring_covers_r ? r2 : g2;
  // one comparison per candidate: d2 <= lim  <=>  d2 <= g2 and (d2 < r2, or <= when inclusive)
  const float lim = (ring_covers_r && !inclusive) ? __uint_as_float(__float_as_uint(r2) - 1u) : B2;
  const float bin_scale = (float)KC_NBINS / B2;
  const int64_t nunits = (ns + 63) / 64;

  for (int64_t unit = (int64_t)blockIdx.x * KC_WAVES + wid; unit < nunits;
       unit += (int64_t)gridDim.x * KC_WAVES) {
    const int64_t j = unit * 64 + lane;
    const bool have = j < ns;
    const float4 q = have ? sorted[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int cx = kc_clamped_cell(q.x, g.ox, g.inv_s, g.dx);
    const int cy = kc_clamped_cell(q.y, g.oy, g.inv_s, g.dy);
    const int cz = kc_clamped_cell(q.z, g.oz, g.inv_s, g.dz);
    uint64_t pending = __ballot(have);
    // GEOF: moment sums of the lane's query (filled in the one round that finishes it)
    double s1[3] = {0.0, 0.0, 0.0}, s2[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    int cnt = 0;
    bool featured = false;
    while (pending) {
      const int leader = __ffsll((unsigned long long)pending) - 1;
      const int lx = __builtin_amdgcn_readlane(cx, leader), ly = __builtin_amdgcn_readlane(cy, leader),
                lz = __builtin_amdgcn_readlane(cz, leader);
      const bool mine = ((pending >> lane) & 1) && cy == ly && cz == lz && cx >= lx &&
                        cx < lx + KC_W;
      pending &= ~__ballot(mine);
      // the 9 candidate ranges (lanes 0..8)
      int rs = 0, rl = 0;
      if (lane < 9) {
        const int z = lz + lane / 3 - 1, y = ly + lane % 3 - 1;
        if (z >= 0 && z < g.dz && y >= 0 && y < g.dy) {
          const int x0 = lx - 1 < 0 ? 0 : lx - 1;
          const int x1 = lx + KC_W >= g.dx ? g.dx - 1 : lx + KC_W;
          const int64_t rb = ((int64_t)z * g.dy + y) * g.dx;
          rs = rowptr[rb + x0];
          rl = rowptr[rb + x1 + 1] - rs;
        }
      }
      // the round's candidate stream as a table of <= 64-candidate blocks (first position in
      // `sorted`, length): lanes 0..8 write the blocks of their range at its prefix offset
      const int nbk = (rl + 63) >> 6;
      const int bpre = (int)wave_inclusive_scan((uint32_t)nbk) - nbk;
      const int nblk = __builtin_amdgcn_readlane(bpre + nbk, 8);
      bool ok = false;
      int bK = KC_NBINS - 1, len = 0;
      // candidate rows of block bi, one per lane (slots past the block's end hold a point that is
      // out of every radius: the consumers run whole groups of 4 without a tail)
      auto fetch_blk = [&](int bi) -> float4 {
        float4 p = make_float4(1e30f, 1e30f, 1e30f, 0.f);
        if (bi < nblk) {
          const int base = L.blk[bi], m = (int)L.blkm[bi];
          if (lane < m) p = sorted[base + lane];
        }
        return p;
      };
      // block bi into the staging arrays (its rows were requested one block earlier and travelled
      // while the previous block was evaluated); returns its length
      auto stage_blk = [&](int bi, const float4& p) -> int {
        kc_wave_sync();                                   // every lane is done with the previous block
        L.sx[lane] = p.x;
        L.sy[lane] = p.y;
        L.sz[lane] = p.z;
        kc_wave_sync();
        return __builtin_amdgcn_readfirstlane((int)L.blkm[bi]);
      };
      if (nblk <= KC_MAXBLK) {
        kc_wave_sync();                                   // (the previous round's table readers)
        for (int jb = 0; jb < nbk; ++jb) {
          L.blk[bpre + jb] = rs + 64 * jb;
          L.blkm[bpre + jb] = (uint8_t)(rl - 64 * jb < 64 ? rl - 64 * jb : 64);
        }
        // ---- pass A: per-lane histogram of d2 (lanes outside the group count too: their
        //      columns are simply not read) --------------------------------------------------
#pragma unroll
        for (int b = 0; b < KC_NBINS; ++b) L.hist[b][lane] = 0u;
        kc_wave_sync();
        float4 nxt = fetch_blk(0);
        for (int bi = 0; bi < nblk; ++bi) {
          const int m = stage_blk(bi, nxt);
          nxt = fetch_blk(bi + 1);
          for (int t = 0; t < ((SPT_KC_SKIP & 4) ? 0 : m); t += 4) {
            float d2[4];
            kc_d2x4(L, t, q, d2);
            const kc_f32x2 s01 = (kc_f32x2){d2[0], d2[1]} * bin_scale;     // packed: d2 * bin_scale
            const kc_f32x2 s23 = (kc_f32x2){d2[2], d2[3]} * bin_scale;
            const float sc[4] = {s01.x, s01.y, s23.x, s23.y};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              int bin = (int)sc[u];
              bin = bin > KC_NBINS - 1 ? KC_NBINS - 1 : bin;
              __hip_atomic_fetch_add(&L.hist[bin][lane], d2[u] <= lim ? 1u : 0u, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
          }
        }
        kc_wave_sync();
        int cum = 0, found = -1;
#pragma unroll
        for (int b = 0; b < KC_NBINS; ++b) {
          cum += (int)L.hist[b][lane];
          if (found < 0 && cum >= K) {
            found = b;
            len = cum;
          }
        }
        if (found >= 0) {
          bK = found;
          ok = len <= KC_CAP;
        } else {                     // fewer than K within B2: final only if B2 is the radius
          len = cum;
          ok = ring_covers_r;
        }
      }
      ok = ok && mine;
#if SPT_KC_SKIP & 4
      ok = mine;
#endif
      // ---- lanes the fast path cannot finish ----------------------------------------------
      const uint64_t fb = __ballot(mine && !ok);
      if (fb) {
        int base = 0;
        if (lane == 0) base = atomicAdd(todo_count, __popcll(fb));
        base = __builtin_amdgcn_readfirstlane(base);
        if (mine && !ok) todo[base + __popcll(fb & lanemask_lt())] = __float_as_int(q.w);
      }
      uint64_t todo_sort = __ballot(ok);
      if (todo_sort == 0) continue;
      uint64_t kth = KNN_EMPTY;         // GEOF: the K-th key of the lane's query (lists shorter than K: all)
#if SPT_KC_SKIP & 2     /* measurement: rounds end after pass A */
      continue;
#endif
      // ---- pass B: the candidates at or below the lane's bin, as 64 * block + slot
      //      (branch-free append: store, then advance if taken) --------------------------------
      // take <=> d2 <= lim and bin(d2) <= bK, with bin(d2) = min(int(d2 * bin_scale), NBINS - 1) as
      // in pass A.  bin is monotone in d2, so the second condition is d2 <= D for the largest float
      // D with fl(D * bin_scale) < bK + 1: found once per round (a division and one-ulp steps), and
      // the per-candidate test is ONE comparison against min(lim, D)
      float dlim = -1.0f;                                  // lanes not in the round take nothing
      if (ok) {
        dlim = lim;
        if (bK < KC_NBINS - 1) {
          const float thr = (float)(bK + 1);
          float D = thr / bin_scale;
          while (D * bin_scale >= thr) D = __uint_as_float(__float_as_uint(D) - 1u);
          while (__uint_as_float(__float_as_uint(D) + 1u) * bin_scale < thr)
            D = __uint_as_float(__float_as_uint(D) + 1u);
          dlim = D < lim ? D : lim;
        }
      }
      int nl = 0;
      {
        float4 nxt = fetch_blk(0);
        for (int bi = 0; bi < nblk; ++bi) {
          const int m = stage_blk(bi, nxt);
          nxt = fetch_blk(bi + 1);
          const int pb = 64 * bi;
          for (int t = 0; t < m; t += 4) {
            float d2[4];
            kc_d2x4(L, t, q, d2);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              L.list[nl][lane] = (uint16_t)(pb + t + u);
              nl += d2[u] <= dlim ? 1 : 0;
            }
          }
        }
      }
      kc_wave_sync();
#if SPT_KC_SKIP & 1     /* measurement: no per-query sort / output */
      continue;
#endif
      // ---- per query: 64-lane sort of its list, K results.  The candidate rows of a query are a
      //      64-lane gather (L2 latency >> one sort): the rows of the NEXT four queries (pairs of
      //      queries) are in flight while the current four sort ---------------------------------
      // lists of <= 32 candidates (K <= 32: the DALES setting k = 25) are sorted two queries at a
      // time, one per half wave
#ifdef SPT_KNN_NO_PAIR   /* measurement builds only */
      const uint64_t small = 0ull;
#else
      const uint64_t small = (K <= 32) ? __ballot(ok && len <= 32) : 0ull;
#endif
      uint64_t big = todo_sort & ~small;
      uint64_t pairs = small;
      auto gidx = [&](int pos) { return L.blk[pos >> 6] + (pos & 63); };
      constexpr int KC_G = SPT_KC_G;                        // queries (pairs) per prefetch group
      if (pairs) {
        auto next_pair = [&](int& qa, int& qb) {
          qa = qb = -1;
          if (!pairs) return;
          qa = __ffsll((unsigned long long)pairs) - 1;
          pairs &= pairs - 1;
          qb = pairs ? __ffsll((unsigned long long)pairs) - 1 : qa;            // odd one out: alone
          pairs &= pairs - 1;
        };
        auto fetch2 = [&](int qa, int qb) -> float4 {
          float4 pc = make_float4(0.f, 0.f, 0.f, 0.f);
          if (qa < 0) return pc;
          const int qh = (lane < 32) ? qa : qb;                                // the query of this half
          const int n = __shfl(len, qh, 64);
          if ((lane & 31) < n && (lane < 32 || qb != qa)) pc = sorted[gidx((int)L.list[lane & 31][qh])];
          return pc;
        };
        auto sort2 = [&](int qa, int qb, const float4& pc2) {
          const int qh = (lane < 32) ? qa : qb;
          const int hl = lane & 31;
          const bool live = lane < 32 || qb != qa;
          const int n = __shfl(len, qh, 64);
          float4 qq;
          qq.x = __shfl(q.x, qh, 64);
          qq.y = __shfl(q.y, qh, 64);
          qq.z = __shfl(q.z, qh, 64);
          const int64_t qi = (int64_t)(uint32_t)__shfl(__float_as_int(q.w), qh, 64);
          uint64_t key = KNN_EMPTY;
          if (hl < n && live)
            key = ((uint64_t)__float_as_uint(kc_d2(qq, pc2)) << 32) | (uint32_t)__float_as_int(pc2.w);
          key = wave_sort_halves(key, lane);
          if constexpr (GEOF) {
            const uint32_t alo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, K - 1);
            const uint32_t ahi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), K - 1);
            const uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, 32 + K - 1);
            const uint32_t bhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), 32 + K - 1);
            if (lane == qa) kth = ((uint64_t)ahi << 32) | alo;
            if (qb != qa && lane == qb) kth = ((uint64_t)bhi << 32) | blo;
          }
          if (hl < K && live) {
            const bool okk = key != KNN_EMPTY;
            float d = __uint_as_float((uint32_t)(key >> 32));
            if (!squared) d = sqrtf(d);
            out_idx[qi * K + hl] = okk ? (int64_t)(uint32_t)(key & 0xffffffffu) : -1;
            out_dist[qi * K + hl] = okk ? d : -1.0f;
          }
        };
        int qa[KC_G], qb[KC_G];
        float4 pc2[KC_G];
#pragma unroll
        for (int gq = 0; gq < KC_G; ++gq) next_pair(qa[gq], qb[gq]);
#pragma unroll
        for (int gq = 0; gq < KC_G; ++gq) pc2[gq] = fetch2(qa[gq], qb[gq]);
        while (qa[0] >= 0) {
          int na[KC_G], nb2[KC_G];
          float4 pn2[KC_G];
#pragma unroll
          for (int gq = 0; gq < KC_G; ++gq) next_pair(na[gq], nb2[gq]);
#pragma unroll
          for (int gq = 0; gq < KC_G; ++gq) pn2[gq] = fetch2(na[gq], nb2[gq]);
#pragma unroll
          for (int gq = 0; gq < KC_G; ++gq)
            if (qa[gq] >= 0) sort2(qa[gq], qb[gq], pc2[gq]);
#pragma unroll
          for (int gq = 0; gq < KC_G; ++gq) {
            qa[gq] = na[gq];
            qb[gq] = nb2[gq];
            pc2[gq] = pn2[gq];
          }
        }
      }
      todo_sort = big;
      if (todo_sort != 0) {
      auto next_q = [&]() {
        const int ql = todo_sort ? __ffsll((unsigned long long)todo_sort) - 1 : -1;
        todo_sort &= todo_sort - 1;
        return ql;
      };
      auto fetch = [&](int ql) -> float4 {
        float4 pc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ql < 0) return pc;
        const int n = __builtin_amdgcn_readlane(len, ql);
        if (lane < n) pc = sorted[gidx((int)L.list[lane][ql])];
        return pc;
      };
      auto sort1 = [&](int ql, const float4& pc) {
        const int n = __builtin_amdgcn_readlane(len, ql);
        float4 qq;
        qq.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.x), ql));
        qq.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.y), ql));
        qq.z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.z), ql));
        const int64_t qi = (int64_t)(uint32_t)__builtin_amdgcn_readlane(__float_as_int(q.w), ql);
        uint64_t key = KNN_EMPTY;
        if (lane < n)
          key = ((uint64_t)__float_as_uint(kc_d2(qq, pc)) << 32) | (uint32_t)__float_as_int(pc.w);
        key = wave_sort(key, lane);
        if constexpr (GEOF) {
          const uint32_t klo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, K - 1);
          const uint32_t khi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), K - 1);
          if (lane == ql) kth = ((uint64_t)khi << 32) | klo;
        }
        if (lane < K) {
          const bool okk = key != KNN_EMPTY;
          float d = __uint_as_float((uint32_t)(key >> 32));
          if (!squared) d = sqrtf(d);
          out_idx[qi * K + lane] = okk ? (int64_t)(uint32_t)(key & 0xffffffffu) : -1;
          out_dist[qi * K + lane] = okk ? d : -1.0f;
        }
      };
      int qc[KC_G];
      float4 pc[KC_G];
#pragma unroll
      for (int gq = 0; gq < KC_G; ++gq) qc[gq] = next_q();
#pragma unroll
      for (int gq = 0; gq < KC_G; ++gq) pc[gq] = fetch(qc[gq]);
      while (qc[0] >= 0) {
        int qn[KC_G];
        float4 pn[KC_G];
#pragma unroll
        for (int gq = 0; gq < KC_G; ++gq) qn[gq] = next_q();
#pragma unroll
        for (int gq = 0; gq < KC_G; ++gq) pn[gq] = fetch(qn[gq]);
#pragma unroll
        for (int gq = 0; gq < KC_G; ++gq)
          if (qc[gq] >= 0) sort1(qc[gq], pc[gq]);
#pragma unroll
        for (int gq = 0; gq < KC_G; ++gq) {
          qc[gq] = qn[gq];
          pc[gq] = pn[gq];
        }
      }
      }  // big lists
      if constexpr (GEOF) {
        // ---- moments of the round's finished queries, one query per lane: list entry -> row
        //      (GU gathers in flight, unconditional: idle lanes read the round's first row and
        //      take nothing) -> key as the sort formed it -> at or below the K-th: a winner ----------
        constexpr int GU = 8;                                 // gathers in flight per lane
        const double qx = (double)q.x, qy = (double)q.y, qz = (double)q.z;
        // rows of list entries t .. t + GU - 1: every LDS read and every gather unconditional (an idle
        // lane reads entry 0 of the round's first block), so that the requests leave together
        auto request = [&](int t, float4 (&row)[GU], bool (&act)[GU]) {
          int pos[GU], gi[GU];
#pragma unroll
          for (int u = 0; u < GU; ++u) pos[u] = (int)L.list[t + u < KC_CAP ? t + u : KC_CAP][lane];
#pragma unroll
          for (int u = 0; u < GU; ++u) {
            act[u] = ok && t + u < len;
            pos[u] = act[u] ? pos[u] : 0;
          }
#pragma unroll
          for (int u = 0; u < GU; ++u) gi[u] = gidx(pos[u]);
#pragma unroll
          for (int u = 0; u < GU; ++u) row[u] = sorted[gi[u]];
        };
        float4 row[GU];
        bool act[GU];
        request(0, row, act);
        for (int t = 0; t < KC_CAP; t += GU) {
          if (__ballot(ok && t < len) == 0) break;
          float4 nrow[GU];
          bool nact[GU];
          request(t + GU, nrow, nact);
#pragma unroll
          for (int u = 0; u < GU; ++u) {
            const uint64_t key = ((uint64_t)__float_as_uint(kc_d2(q, row[u])) << 32) |
                                 (uint32_t)__float_as_int(row[u].w);
            const bool take = act[u] && key <= kth;
            const double dx = take ? (double)row[u].x - qx : 0.0;
            const double dy = take ? (double)row[u].y - qy : 0.0;
            const double dz = take ? (double)row[u].z - qz : 0.0;
            s1[0] += dx; s1[1] += dy; s1[2] += dz;
            s2[0] += dx * dx; s2[1] += dx * dy; s2[2] += dx * dz;
            s2[3] += dy * dy; s2[4] += dy * dz; s2[5] += dz * dz;
            cnt += take ? 1 : 0;
          }
#pragma unroll
          for (int u = 0; u < GU; ++u) {
            row[u] = nrow[u];
            act[u] = nact[u];
          }
        }
        featured = featured || ok;
      }
    }
    if constexpr (GEOF) {
      if (featured)
        finish_features(s1, s2, cnt, k_min, post, feats + (int64_t)(uint32_t)__float_as_int(q.w) * 11);
    }
  }
}

// Eigenfeatures of the queries the cell kernel left to knn_search_kernel (or of all points when the
// cell kernel did not run): one lane per listed point, its K found neighbours - itself among them,
// at distance 0 - read back from the index rows just written.  The same sums about the point's own
// position as in knn_cell_kernel<true>.
__global__ __launch_bounds__(256) void knn_geof_rows_kernel(
    const float* __restrict__ xyz, int64_t n_all, const int32_t* __restrict__ list,
    const int32_t* __restrict__ n_dev, const int64_t* __restrict__ idx, int K, int k_min, int post,
    float* __restrict__ feats) {
  const int64_t n = n_dev ? (int64_t)*n_dev : n_all;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int64_t p = list ? (int64_t)list[i] : i;
    const double px = xyz[p * 3], py = xyz[p * 3 + 1], pz = xyz[p * 3 + 2];
    double s1[3] = {0.0, 0.0, 0.0}, s2[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    int cnt = 0;
    for (int c = 0; c < K; ++c) {
      const int64_t t = idx[p * K + c];
      if (t < 0) continue;
      const double dx = (double)xyz[t * 3] - px, dy = (double)xyz[t * 3 + 1] - py,
                   dz = (double)xyz[t * 3 + 2] - pz;
      s1[0] += dx; s1[1] += dy; s1[2] += dz;
      s2[0] += dx * dx; s2[1] += dx * dy; s2[2] += dx * dz;
      s2[3] += dy * dy; s2[4] += dy * dz; s2[5] += dz * dz;
      ++cnt;
    }
    finish_features(s1, s2, cnt, k_min, post, feats + p * 11);
  }
}

struct KnnPlan {
  size_t off_cell, off_perm, off_rowptr, off_sorted, off_sortws, off_todo, off_count, sortws, total;
};

}  // namespace spt

using namespace spt;

static KnnPlan knn_plan(int64_t ns, int64_t ncells) {
  KnnPlan p;
  size_t o = 0;
  p.off_cell = o;   o += align_up((size_t)(ns > 0 ? ns : 1) * 8, 256);
  p.off_perm = o;   o += align_up((size_t)(ns > 0 ? ns : 1) * 4, 256);
  p.off_rowptr = o; o += align_up((size_t)(ncells + 1) * 4, 256);
  p.off_sorted = o; o += align_up((size_t)(ns > 0 ? ns : 1) * 16, 256);
  p.sortws = spt_csr_build_workspace_bytes(ns, ncells);
  p.off_sortws = o; o += align_up(p.sortws, 256);
  p.off_todo = o;   o += align_up((size_t)(ns > 0 ? ns : 1) * 4, 256);   // queries left to the slow path
  p.off_count = o;  o += 256;
  p.total = o;
  return p;
}

// ---- bounding box --------------------------------------------------------------------
// torch's column reductions of a tall [n, 3] tensor take ~8 ms each at 15 M points; the
// grid description needs min and max before anything else can start.
__device__ __forceinline__ uint32_t bbox_key(float f) {      // order-preserving f32 -> u32
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void bbox_init_kernel(uint32_t* __restrict__ keys) {
  if (threadIdx.x < 3) keys[threadIdx.x] = 0xffffffffu;        // running minima
  else if (threadIdx.x < 6) keys[threadIdx.x] = 0u;            // running maxima
}

__global__ __launch_bounds__(256) void bbox_kernel(const float* __restrict__ xyz, int64_t n,
                                                   uint32_t* __restrict__ keys) {
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const float v = xyz[i * 3 + q];
      lo[q] = fminf(lo[q], v);
      hi[q] = fmaxf(hi[q], v);
    }
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    lo[q] = wave_reduce_min(lo[q]);
    hi[q] = wave_reduce_max(hi[q]);
  }
  // one set of atomics per workgroup (per wave they were ~100 k contended updates of six words:
  // 1.1 ms at 12 M points, 40x the time of the reads)
  __shared__ float sl[4][6];
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      sl[threadIdx.x >> 6][q] = lo[q];
      sl[threadIdx.x >> 6][3 + q] = hi[q];
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {                                       // integer atomics: order-free
    const int q = threadIdx.x;
    float v = sl[0][q];
    for (int w = 1; w < 4; ++w) v = q < 3 ? fminf(v, sl[w][q]) : fmaxf(v, sl[w][q]);
    if (q < 3) atomicMin(&keys[q], bbox_key(v));
    else atomicMax(&keys[q], bbox_key(v));
  }
}

__global__ void bbox_decode_kernel(const uint32_t* __restrict__ keys, float* __restrict__ out) {
  if (threadIdx.x < 6) {
    const uint32_t k = keys[threadIdx.x];
    out[threadIdx.x] = __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
  }
}

// ---- spatial (cell) order of a cloud -----------------------------------------------------
// order[j] = index of the j-th point when points are grouped by grid cell (z, y, x major):
// consumers that gather neighbourhoods (point_geof.hip) visit points in this order so that
// the rows one wave touches overlap and stay in L2.
extern "C" int spt_bbox_f32(const float* xyz, int64_t n, float* lo_hi, spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(n >= 1 && xyz && lo_hi, "bad arguments");
  uint32_t* keys = (uint32_t*)(lo_hi + 6);                     // caller provides 12 floats
  bbox_init_kernel<<<1, 256, 0, stream>>>(keys);
  {
    int grid = stream_grid(n, 256 * 8);
    if (grid > 1024) grid = 1024;
    bbox_kernel<<<grid, 256, 0, stream>>>(xyz, n, keys);
  }
  bbox_decode_kernel<<<1, 64, 0, stream>>>(keys, lo_hi);
  SPT_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t spt_spatial_order_workspace_bytes(int64_t n, int64_t ncells) {
  if (n < 0 || ncells < 1) return 0;
  return align_up((size_t)(n > 0 ? n : 1) * 8, 256) + align_up((size_t)(ncells + 1) * 4, 256) +
         align_up(spt_csr_build_workspace_bytes(n, ncells), 256);
}

extern "C" int spt_spatial_order(const float* xyz, int64_t n, float cell_size, const float* origin,
                                 const int32_t* dims, int32_t* order, void* ws, size_t ws_bytes,
                                 spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(n >= 0 && cell_size > 0.f && origin && dims, "bad arguments");
  SPT_CHECK_ARG(dims[0] >= 1 && dims[1] >= 1 && dims[2] >= 1, "bad grid dims");
  const int64_t ncells = (int64_t)dims[0] * dims[1] * dims[2];
  SPT_CHECK_ARG(ncells < ((int64_t)1 << 31), "grid has more than 2^31 cells: enlarge cell_size");
  if (n == 0) return 0;
  SPT_CHECK_ARG(xyz && order, "null pointer");
  SPT_CHECK_ARG(ws && ws_bytes >= spt_spatial_order_workspace_bytes(n, ncells), "workspace too small");
  Grid g;
  g.ox = origin[0]; g.oy = origin[1]; g.oz = origin[2];
  g.s = cell_size; g.inv_s = 1.0f / cell_size;
  g.dx = dims[0]; g.dy = dims[1]; g.dz = dims[2];
  char* base = (char*)ws;
  int64_t* cell = (int64_t*)base;
  int32_t* rowptr = (int32_t*)(base + align_up((size_t)n * 8, 256));
  char* sortws = (char*)rowptr + align_up((size_t)(ncells + 1) * 4, 256);
  knn_cell_ids_kernel<<<stream_grid(n, 256), 256, 0, stream>>>(xyz, n, g, cell);
  return spt_csr_build(cell, n, ncells, order, rowptr, sortws,
                       spt_csr_build_workspace_bytes(n, ncells), stream_);
}

// ---- the host-side grid description's probes (neighbors._grid_for) as two kernels ---------------
// The cell size is chosen from the occupancy of the non-empty cells on a spatially coherent
// subsample: whole coarse cells (edge `coarse`) drawn by a hash of their coordinates.  As torch
// expressions on the full cloud that was ~25 elementwise launches and three [n, 3] int64
// temporaries per search (2.5 ms of a 25 ms preprocessing call at 15 M points); here one pass
// compacts the kept points (order irrelevant: they are only counted per cell) ...
constexpr int SUB_PER_THREAD = 16;
__global__ __launch_bounds__(256) void knn_subsample_kernel(
    const float* __restrict__ xyz, int64_t n, float lx, float ly, float lz, float coarse,
    int thresh, float* __restrict__ out, int32_t* __restrict__ count) {
  // one reservation of output rows per 4 096-point tile (per wave it was 230 k atomics on one
  // word at 15 M points: 2.6 ms): keep bits per thread, block scan, then the kept points again
  __shared__ int wave_tot[4];
  __shared__ int tile_base;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t tile = (int64_t)blockDim.x * SUB_PER_THREAD;
  for (int64_t t0 = (int64_t)blockIdx.x * tile; t0 < n; t0 += (int64_t)gridDim.x * tile) {
    uint32_t bits = 0;
#pragma unroll
    for (int j = 0; j < SUB_PER_THREAD; ++j) {
      const int64_t i = t0 + (int64_t)j * blockDim.x + threadIdx.x;
      if (i < n) {
        const float x = xyz[i * 3], y = xyz[i * 3 + 1], z = xyz[i * 3 + 2];
        const int64_t cx = (int64_t)floorf((x - lx) / coarse), cy = (int64_t)floorf((y - ly) / coarse),
                      cz = (int64_t)floorf((z - lz) / coarse);
        const int64_t key = (cx * 73856093ll) ^ (cy * 19349663ll) ^ (cz * 83492791ll);
        if ((int)(key & 0xFFFF) < thresh) bits |= 1u << j;
      }
    }
    const int mine = __popc(bits);
    const int incl = (int)wave_inclusive_scan((uint32_t)mine);
    if (lane == 63) wave_tot[wid] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
      const int total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
      tile_base = total ? atomicAdd(count, total) : 0;
    }
    __syncthreads();
    int64_t o = (int64_t)tile_base + (incl - mine);
    for (int w = 0; w < wid; ++w) o += wave_tot[w];
#pragma unroll
    for (int j = 0; j < SUB_PER_THREAD; ++j)
      if (bits & (1u << j)) {
        const int64_t i = t0 + (int64_t)j * blockDim.x + threadIdx.x;
        out[o * 3] = xyz[i * 3]; out[o * 3 + 1] = xyz[i * 3 + 1]; out[o * 3 + 2] = xyz[i * 3 + 2];
        ++o;
      }
    __syncthreads();                                          // wave_tot / tile_base are reused
  }
}

extern "C" int spt_knn_subsample_f32(const float* xyz, int64_t n, const float* lo, float coarse,
                                     int thresh, float* out, int32_t* count, spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(n >= 0 && lo && coarse > 0.f, "bad arguments");
  SPT_CHECK_ARG(count, "null pointer");
  (void)hipMemsetAsync(count, 0, 4, stream);
  if (n == 0) return 0;
  SPT_CHECK_ARG(xyz && out, "null pointer");
  knn_subsample_kernel<<<stream_grid(n, 256 * SUB_PER_THREAD), 256, 0, stream>>>(xyz, n, lo[0], lo[1], lo[2], coarse,
                                                               thresh, out, count);
  SPT_CHECK_LAUNCH();
  return 0;
}

// ... and the linear cell id of every point at a candidate cell size (counted by the caller)
extern "C" int spt_grid_cell_ids_f32(const float* xyz, int64_t n, float cell_size,
                                     const float* origin, const int32_t* dims, int64_t* cell,
                                     spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(n >= 0 && cell_size > 0.f && origin && dims, "bad arguments");
  SPT_CHECK_ARG(dims[0] >= 1 && dims[1] >= 1 && dims[2] >= 1, "bad grid dims");
  if (n == 0) return 0;
  SPT_CHECK_ARG(xyz && cell, "null pointer");
  Grid g;
  g.ox = origin[0]; g.oy = origin[1]; g.oz = origin[2];
  g.s = cell_size; g.inv_s = 1.0f / cell_size;
  g.dx = dims[0]; g.dy = dims[1]; g.dz = dims[2];
  knn_cell_ids_kernel<<<stream_grid(n, 256), 256, 0, stream>>>(xyz, n, g, cell);
  SPT_CHECK_LAUNCH();
  return 0;
}

// ... or, without the sort a `unique` needs, their NUMBER: one bit per cell, set by the points,
// counted afterwards (exact; the bitmap is ncells / 8 bytes of the caller's workspace)
__global__ __launch_bounds__(256) void knn_mark_cells_kernel(const float* __restrict__ pos, int64_t n,
                                                             Grid g, uint32_t* __restrict__ bits) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int x = cell_coord(pos[i * 3 + 0], g.ox, g.inv_s);
    int y = cell_coord(pos[i * 3 + 1], g.oy, g.inv_s);
    int z = cell_coord(pos[i * 3 + 2], g.oz, g.inv_s);
    x = x < 0 ? 0 : (x >= g.dx ? g.dx - 1 : x);
    y = y < 0 ? 0 : (y >= g.dy ? g.dy - 1 : y);
    z = z < 0 ? 0 : (z >= g.dz ? g.dz - 1 : z);
    const int64_t c = ((int64_t)z * g.dy + y) * g.dx + x;
    const uint32_t bit = 1u << (c & 31);
    if (!(bits[c >> 5] & bit)) atomicOr(&bits[c >> 5], bit);
  }
}

__global__ __launch_bounds__(256) void knn_count_bits_kernel(const uint32_t* __restrict__ bits,
                                                             int64_t nwords,
                                                             unsigned long long* __restrict__ count) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += stride)
    acc += (uint32_t)__popc(bits[i]);
  acc = wave_reduce_sum(acc);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(count, (unsigned long long)acc);
}

extern "C" size_t spt_grid_count_cells_workspace_bytes(int64_t ncells) {
  if (ncells < 1) return 0;
  return align_up((size_t)((ncells + 31) / 32) * 4, 256) + 256;
}

extern "C" int spt_grid_count_cells_f32(const float* xyz, int64_t n, float cell_size,
                                        const float* origin, const int32_t* dims, int64_t* count,
                                        void* ws, size_t ws_bytes, spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(n >= 0 && cell_size > 0.f && origin && dims && count, "bad arguments");
  SPT_CHECK_ARG(dims[0] >= 1 && dims[1] >= 1 && dims[2] >= 1, "bad grid dims");
  const int64_t ncells = (int64_t)dims[0] * dims[1] * dims[2];
  SPT_CHECK_ARG(ws && ws_bytes >= spt_grid_count_cells_workspace_bytes(ncells), "workspace too small");
  SPT_CHECK_ARG(xyz || n == 0, "null pointer");
  Grid g;
  g.ox = origin[0]; g.oy = origin[1]; g.oz = origin[2];
  g.s = cell_size; g.inv_s = 1.0f / cell_size;
  g.dx = dims[0]; g.dy = dims[1]; g.dz = dims[2];
  const int64_t nwords = (ncells + 31) / 32;
  uint32_t* bits = (uint32_t*)ws;
  (void)hipMemsetAsync(bits, 0, (size_t)nwords * 4, stream);
  (void)hipMemsetAsync(count, 0, 8, stream);
  if (n > 0) {
    knn_mark_cells_kernel<<<stream_grid(n, 256), 256, 0, stream>>>(xyz, n, g, bits);
    knn_count_bits_kernel<<<stream_grid(nwords, 256), 256, 0, stream>>>(
        bits, nwords, (unsigned long long*)count);
  }
  SPT_CHECK_LAUNCH();
  return 0;
}

// process-wide switch between the two self-search kernels (tests cross-check them; 1 = shared
// candidate streams, 0 = wave per query); returns the previous setting
static std::atomic<int> g_knn_cell_path{1};
static thread_local int tl_knn_cell_path = -1;   // per-call choice of spt_grid_knn_ex_f32
static bool knn_cell_path_enabled() {
  return tl_knn_cell_path >= 0 ? tl_knn_cell_path != 0 : g_knn_cell_path != 0;
}
extern "C" int spt_knn_use_cell_path(int on) {
  const int prev = g_knn_cell_path;
  g_knn_cell_path = on ? 1 : 0;
  return prev;
}

extern "C" size_t spt_grid_knn_workspace_bytes(int64_t ns, int64_t ncells) {
  if (ns < 0 || ncells < 1) return 0;
  return knn_plan(ns, ncells).total;
}

static int grid_knn_impl(const float* query, int64_t nq, const float* search, int64_t ns, int K,
                         float r, float cell_size, const float* origin, const int32_t* dims,
                         int order_queries_by_cell, int inclusive, int squared,
                         const int64_t* after_idx, const float* after_d2, int64_t* idx,
                         float* dist, int32_t* cell_order, void* ws, size_t ws_bytes,
                         spt_stream_t stream_, float* feats = nullptr, int k_min = 1, int post = 0);

extern "C" int spt_grid_knn_f32(const float* query, int64_t nq, const float* search,
                                int64_t ns, int K, float r, float cell_size,
                                const float* origin, const int32_t* dims,
                                int order_queries_by_cell, int inclusive, int squared,
                                int64_t* idx, float* dist, int32_t* cell_order, void* ws,
                                size_t ws_bytes, spt_stream_t stream_) {
  return grid_knn_impl(query, nq, search, ns, K, r, cell_size, origin, dims,
                       order_queries_by_cell, inclusive, squared, nullptr, nullptr, idx, dist,
                       cell_order, ws, ws_bytes, stream_);
}

// Same with the formulation chosen per call (-1: the process default; 0: one wave per query;
// 1: cell-centric self-search where it applies) - no process-wide state involved.
extern "C" int spt_grid_knn_ex_f32(const float* query, int64_t nq, const float* search,
                                   int64_t ns, int K, float r, float cell_size,
                                   const float* origin, const int32_t* dims,
                                   int order_queries_by_cell, int inclusive, int squared,
                                   int64_t* idx, float* dist, int32_t* cell_order, int formulation,
                                   void* ws, size_t ws_bytes, spt_stream_t stream_) {
  const int prev = tl_knn_cell_path;
  tl_knn_cell_path = formulation < 0 ? -1 : (formulation != 0);
  const int st = grid_knn_impl(query, nq, search, ns, K, r, cell_size, origin, dims,
                               order_queries_by_cell, inclusive, squared, nullptr, nullptr, idx,
                               dist, cell_order, ws, ws_bytes, stream_);
  tl_knn_cell_path = prev;
  return st;
}

// Self-search WITH the eigenfeatures of what it finds (KNN + PointFeatures of the reference's
// preprocessing chain in one call: src/transforms/neighbors.py:11-95 -> src/transforms/point.py:
// 160-180 -> src/utils/geometry.py:80-126).  Every point searches the cloud for its K nearest
// (itself included, column 0: K = k + 1 of knn_1); feats[i] = the 11 features (pgeof's column order)
// of the neighbourhood {found points}: exactly geometric_features(xyz, idx[:, 1:], add_self) of
// spt_point_geof_dense_f32, the moment sums taken in f64 about the point itself (a different
// summation order: equal whenever the sums are exact, i.e. for coordinates of similar magnitude,
// and within an ulp of the f32 outputs otherwise).  formulation as in spt_grid_knn_ex_f32.
extern "C" int spt_grid_knn_geof_f32(const float* xyz, int64_t n, int K, float r, float cell_size,
                                     const float* origin, const int32_t* dims, int inclusive,
                                     int squared, int k_min, int post, int64_t* idx, float* dist,
                                     float* feats, int32_t* cell_order, int formulation, void* ws,
                                     size_t ws_bytes, spt_stream_t stream_) {
  if (!feats) return ::spt::fail(-1, "%s: null feature pointer", __func__);
  const int prev = tl_knn_cell_path;
  tl_knn_cell_path = formulation < 0 ? -1 : (formulation != 0);
  const int st = grid_knn_impl(xyz, n, xyz, n, K, r, cell_size, origin, dims, 1, inclusive, squared,
                               nullptr, nullptr, idx, dist, cell_order, ws, ws_bytes, stream_, feats,
                               k_min, post);
  tl_knn_cell_path = prev;
  return st;
}

// The K neighbours ranked strictly AFTER (after_d2[q], after_idx[q]) in the (squared distance,
// index) order of the contract: chained after a K = 64 search it yields neighbours 65..128
// (k_max = 100 of cluster_radius_nn_graph, src/utils/neighbors.py:491).  after_idx[q] < 0 (the
// previous list was not full) -> nothing left for that query.
extern "C" int spt_grid_knn_after_f32(const float* query, int64_t nq, const float* search,
                                      int64_t ns, int K, float r, float cell_size,
                                      const float* origin, const int32_t* dims, int inclusive,
                                      int squared, const int64_t* after_idx,
                                      const float* after_d2, int64_t* idx, float* dist, void* ws,
                                      size_t ws_bytes, spt_stream_t stream_) {
  if (!after_idx || !after_d2) return ::spt::fail(-1, "%s: null continuation keys", __func__);
  return grid_knn_impl(query, nq, search, ns, K, r, cell_size, origin, dims, 1, inclusive,
                       squared, after_idx, after_d2, idx, dist, nullptr, ws, ws_bytes, stream_);
}

static int grid_knn_impl(const float* query, int64_t nq, const float* search, int64_t ns, int K,
                         float r, float cell_size, const float* origin, const int32_t* dims,
                         int order_queries_by_cell, int inclusive, int squared,
                         const int64_t* after_idx, const float* after_d2, int64_t* idx,
                         float* dist, int32_t* cell_order, void* ws, size_t ws_bytes,
                         spt_stream_t stream_, float* feats, int k_min, int post) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(nq >= 0 && ns >= 0, "bad shape");
  SPT_CHECK_ARG(!feats || (query == search && nq == ns && !after_idx),
                "features come with a self-search only");
  SPT_CHECK_ARG(K >= 1 && K <= 64, "K must be in [1, 64]");
  SPT_CHECK_ARG(r > 0.f && cell_size > 0.f && origin && dims, "bad grid");
  SPT_CHECK_ARG(dims[0] >= 1 && dims[1] >= 1 && dims[2] >= 1, "bad grid dims");
  const int64_t ncells = (int64_t)dims[0] * dims[1] * dims[2];
  SPT_CHECK_ARG(ncells < ((int64_t)1 << 31), "grid has more than 2^31 cells: enlarge cell_size");
  SPT_CHECK_ARG(ns < ((int64_t)1 << 31) - 4096, "too many search points");
  if (nq == 0) return 0;
  SPT_CHECK_ARG(query && idx && dist && (search || ns == 0), "null pointer");
  const KnnPlan p = knn_plan(ns, ncells);
  SPT_CHECK_ARG(ws && ws_bytes >= p.total, "workspace too small");
  Grid g;
  g.ox = origin[0]; g.oy = origin[1]; g.oz = origin[2];
  g.s = cell_size; g.inv_s = 1.0f / cell_size;
  g.dx = dims[0]; g.dy = dims[1]; g.dz = dims[2];
  char* base = (char*)ws;
  int64_t* cell = (int64_t*)(base + p.off_cell);
  int32_t* perm = (int32_t*)(base + p.off_perm);
  int32_t* rowptr = (int32_t*)(base + p.off_rowptr);
  float4* sorted = (float4*)(base + p.off_sorted);
  if (ns > 0)
    knn_cell_ids_kernel<<<stream_grid(ns, 256), 256, 0, stream>>>(search, ns, g, cell);
  int st = spt_csr_build(cell, ns, ncells, perm, rowptr, base + p.off_sortws, p.sortws, stream_);
  if (st != 0) return st;
  if (ns > 0)
    knn_gather_sorted_kernel<<<stream_grid(ns, 256), 256, 0, stream>>>(search, perm, ns, sorted);
  // self-search: visiting queries in cell order keeps candidate cells L2-resident
  const int32_t* qorder = (order_queries_by_cell && query == search && nq == ns) ? perm : nullptr;
  // the cell order of the search points is a by-product other gather kernels can reuse
  if (cell_order && ns > 0)
    (void)hipMemcpyAsync(cell_order, perm, (size_t)ns * 4, hipMemcpyDeviceToDevice, stream);
  const int grid = (int)(ceil_div(nq, KNN_WAVES) < 256 * 8 ? ceil_div(nq, KNN_WAVES) : 256 * 8);
  if (qorder && knn_cell_path_enabled() && !after_idx) {
    // self-search in cell order: shared candidate streams (knn_cell_kernel), leftovers through
    // the wave-per-query kernel with their count read on the device
    int32_t* todo = (int32_t*)(base + p.off_todo);
    int32_t* count = (int32_t*)(base + p.off_count);
    (void)hipMemsetAsync(count, 0, 4, stream);
    const int64_t units = ceil_div(ns, 64);
    const int cgrid = (int)(ceil_div(units, KC_WAVES) < 256 * 8 ? ceil_div(units, KC_WAVES) : 256 * 8);
    if (feats)
      knn_cell_kernel<true><<<cgrid, KC_WAVES * 64, 0, stream>>>(
          sorted, ns, rowptr, g, K, r, inclusive, squared, idx, dist, todo, count, k_min, post, feats);
    else
      knn_cell_kernel<false><<<cgrid, KC_WAVES * 64, 0, stream>>>(
          sorted, ns, rowptr, g, K, r, inclusive, squared, idx, dist, todo, count, 0, 0, nullptr);
    knn_search_kernel<<<grid, KNN_WAVES * 64, 0, stream>>>(query, nq, todo, sorted, rowptr, g, K, r,
                                                           inclusive, squared, idx, dist, count);
    if (feats)      // the leftovers' features, from the rows the kernel above just wrote
      knn_geof_rows_kernel<<<256, 256, 0, stream>>>(search, ns, todo, count, idx, K, k_min, post, feats);
  } else {
    knn_search_kernel<<<grid, KNN_WAVES * 64, 0, stream>>>(query, nq, qorder, sorted, rowptr, g, K,
                                                           r, inclusive, squared, idx, dist, nullptr,
                                                           after_idx, after_d2);
    if (feats)
      knn_geof_rows_kernel<<<stream_grid(ns, 256), 256, 0, stream>>>(search, ns, nullptr, nullptr, idx,
                                                                     K, k_min, post, feats);
  }
  SPT_CHECK_LAUNCH();
  return 0;
}
