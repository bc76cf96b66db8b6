// TEST INFRASTRUCTURE - CPU twins (OpenMP) of two entries of libspt_hip.so, same C signatures
// minus the stream and workspace:
//
//   spt_cpu_grid_knn_f32    <-> spt_grid_knn_f32    (src/utils/neighbors.py:51-123 -> FRNN)
//   spt_cpu_point_geof_f32  <-> spt_point_geof_dense_f32 (src/utils/geometry.py:80-126, 236-338)
//
// They restate the CONTRACT, not the GPU code: an exact radius-bounded kNN on a uniform grid
// (K nearest search points with d2 < r^2 - <= when inclusive -, ascending by (d2, index), ties by
// ascending search index, d2 = (dx*dx + dy*dy) + dz*dz in f32 without fma, missing entries
// idx = -1 / dist = -1) and the eigenfeatures of geometry.py in f64.  Used by tests/ (a second,
// independent bit-exact check of the HIP kNN at full DALES / S3DIS size, where the exhaustive
// Python oracle cannot go) and by bench.py's preprocess cpu_baseline leg.  Nothing under
// superpoint_transformer_amd/ may load this library (tests/test_abi.py).
//
// Build: g++ -O3 -fopenmp -ffp-contract=off -shared -fPIC (oracle/cpu/build.py, called by
// __graft_entry__.build()).
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

struct Key {               // (d2 bits, index): lexicographic order = the contract's order
  float d2;
  int64_t idx;
};
inline bool key_less(const Key& a, const Key& b) {
  return a.d2 < b.d2 || (a.d2 == b.d2 && a.idx < b.idx);
}

struct Grid {
  float cell;
  float lo[3];
  int dims[3];
  std::vector<int64_t> start;    // [ncells + 1]
  std::vector<int64_t> order;    // search indices sorted by cell (stable: ascending index inside a cell)
};

inline int cell_coord(float v, float lo, float inv, int dim) {
  int c = (int)floorf((v - lo) * inv);
  return c < 0 ? 0 : (c >= dim ? dim - 1 : c);
}

void build_grid(const float* s, int64_t ns, float cell, Grid& g) {
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int64_t i = 0; i < ns; ++i)
    for (int d = 0; d < 3; ++d) {
      lo[d] = std::min(lo[d], s[3 * i + d]);
      hi[d] = std::max(hi[d], s[3 * i + d]);
    }
  g.cell = cell;
  const float inv = 1.0f / cell;
  int64_t ncells = 1;
  for (int d = 0; d < 3; ++d) {
    g.lo[d] = lo[d];
    g.dims[d] = std::max(1, (int)floorf((hi[d] - lo[d]) * inv) + 1);
    ncells *= g.dims[d];
  }
  g.start.assign(ncells + 1, 0);
  std::vector<int64_t> cid(ns);
  for (int64_t i = 0; i < ns; ++i) {
    const int cx = cell_coord(s[3 * i], g.lo[0], inv, g.dims[0]);
    const int cy = cell_coord(s[3 * i + 1], g.lo[1], inv, g.dims[1]);
    const int cz = cell_coord(s[3 * i + 2], g.lo[2], inv, g.dims[2]);
    cid[i] = ((int64_t)cz * g.dims[1] + cy) * g.dims[0] + cx;
    ++g.start[cid[i] + 1];
  }
  for (int64_t c = 0; c < ncells; ++c) g.start[c + 1] += g.start[c];
  g.order.resize(ns);
  std::vector<int64_t> fill(g.start.begin(), g.start.end() - 1);
  for (int64_t i = 0; i < ns; ++i) g.order[fill[cid[i]]++] = i;
}

}  // namespace

extern "C" {

// A cell size that keeps ~K/2 points per cell of the occupied cells (any size gives the same
// result): estimated from the bounding box and a surface-like occupancy.
float spt_cpu_knn_cell_size(const float* search, int64_t ns, int K, float r) {
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int64_t i = 0; i < ns; ++i)
    for (int d = 0; d < 3; ++d) {
      lo[d] = std::min(lo[d], search[3 * i + d]);
      hi[d] = std::max(hi[d], search[3 * i + d]);
    }
  // probe: count occupied cells at a trial size, then scale for the target occupancy assuming
  // a 2-D (surface) distribution - clouds here are voxelised surfaces
  float cell = r > 0 ? r / 4 : 1.0f;
  for (int it = 0; it < 6; ++it) {
    const float inv = 1.0f / cell;
    int dims[3];
    double nc = 1;
    for (int d = 0; d < 3; ++d) {
      dims[d] = std::max(1, (int)floorf((hi[d] - lo[d]) * inv) + 1);
      nc *= dims[d];
    }
    if (nc > 4e8) { cell *= 2; continue; }
    std::vector<uint8_t> occ((size_t)nc, 0);
    int64_t used = 0;
    const int64_t step = std::max<int64_t>(1, ns / 2000000);
    for (int64_t i = 0; i < ns; i += step) {
      const int cx = cell_coord(search[3 * i], lo[0], inv, dims[0]);
      const int cy = cell_coord(search[3 * i + 1], lo[1], inv, dims[1]);
      const int cz = cell_coord(search[3 * i + 2], lo[2], inv, dims[2]);
      uint8_t& o = occ[((size_t)cz * dims[1] + cy) * dims[0] + cx];
      used += !o;
      o = 1;
    }
    const double per = (double)ns / (double)std::max<int64_t>(used, 1);
    const double target = std::max(4.0, K / 2.0);
    if (per > 0.5 * target && per < 2.0 * target) break;
    cell *= (float)sqrt(target / per);
  }
  return cell;
}

int spt_cpu_grid_knn_f32(const float* query, int64_t nq, const float* search, int64_t ns, int K,
                         float r, float cell_size, int inclusive, int squared, int64_t* idx,
                         float* dist, int nthreads) {
  if (nq < 0 || ns < 0 || K < 1 || !(r > 0) || !(cell_size > 0)) return -1;
  Grid g;
  build_grid(search, ns, cell_size, g);
  const float r2 = r * r;
  const float inv = 1.0f / g.cell;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
  {
    std::vector<Key> heap;           // max-heap of the K best so far
    heap.reserve(K + 1);
    auto cmp = [](const Key& a, const Key& b) { return key_less(a, b); };
#pragma omp for schedule(dynamic, 256)
    for (int64_t qi = 0; qi < nq; ++qi) {
      const float qx = query[3 * qi], qy = query[3 * qi + 1], qz = query[3 * qi + 2];
      const int cx = cell_coord(qx, g.lo[0], inv, g.dims[0]);
      const int cy = cell_coord(qy, g.lo[1], inv, g.dims[1]);
      const int cz = cell_coord(qz, g.lo[2], inv, g.dims[2]);
      heap.clear();
      // a query outside the grid's box is clamped into a border cell: its distance to that cell
      // only adds to the guarantee below, which stays valid (and conservative)
      const int max_ring = std::max(std::max(g.dims[0], g.dims[1]), g.dims[2]);
      for (int ring = 0; ring <= max_ring; ++ring) {
        for (int dz = -ring; dz <= ring; ++dz) {
          const int z = cz + dz;
          if (z < 0 || z >= g.dims[2]) continue;
          for (int dy = -ring; dy <= ring; ++dy) {
            const int y = cy + dy;
            if (y < 0 || y >= g.dims[1]) continue;
            const bool shell_zy = (dz == -ring || dz == ring || dy == -ring || dy == ring);
            const int xstep = shell_zy ? 1 : 2 * ring;      // interior rows: only the two end cells
            for (int dx = -ring; dx <= ring; dx += (xstep > 0 ? xstep : 1)) {
              const int x = cx + dx;
              if (x < 0 || x >= g.dims[0]) continue;
              const int64_t c = ((int64_t)z * g.dims[1] + y) * g.dims[0] + x;
              for (int64_t p = g.start[c]; p < g.start[c + 1]; ++p) {
                const int64_t si = g.order[p];
                const float ddx = qx - search[3 * si], ddy = qy - search[3 * si + 1],
                            ddz = qz - search[3 * si + 2];
                const float d2 = (ddx * ddx + ddy * ddy) + ddz * ddz;
                if (inclusive ? !(d2 <= r2) : !(d2 < r2)) continue;
                const Key k{d2, si};
                if ((int)heap.size() < K) {
                  heap.push_back(k);
                  std::push_heap(heap.begin(), heap.end(), cmp);
                } else if (key_less(k, heap.front())) {
                  std::pop_heap(heap.begin(), heap.end(), cmp);
                  heap.back() = k;
                  std::push_heap(heap.begin(), heap.end(), cmp);
                }
              }
              if (ring == 0) break;
            }
          }
        }
        // every unvisited point lies beyond `ring` whole cells from the query's cell: at
        // distance >= ring * cell (minus a rounding margin: cell coordinates and d2 are f32).
        // Stop when the K-th best cannot be beaten, or r is covered.
        const float guard = std::max(0.0f, ((float)ring - 1e-3f) * g.cell);
        if (guard > r * 1.00001f) break;
        if ((int)heap.size() == K && heap.front().d2 < guard * guard * 0.99999f) break;
      }
      std::sort_heap(heap.begin(), heap.end(), cmp);
      for (int j = 0; j < K; ++j) {
        if (j < (int)heap.size()) {
          idx[qi * K + j] = heap[j].idx;
          dist[qi * K + j] = squared ? heap[j].d2 : sqrtf(heap[j].d2);
        } else {
          idx[qi * K + j] = -1;
          dist[qi * K + j] = -1.0f;
        }
      }
    }
  }
  return 0;
}

// ---- eigenfeatures ---------------------------------------------------------------------------
static void eigh3_jacobi(double a[3][3], double w[3], double v[3][3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) v[i][j] = i == j;
  for (int sweep = 0; sweep < 32; ++sweep) {
    const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (fabs(a[p][q]) < 1e-300) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {
          const double akp = a[k][p], akq = a[k][q];
          a[k][p] = c * akp - s * akq;
          a[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {
          const double apk = a[p][k], aqk = a[q][k];
          a[p][k] = c * apk - s * aqk;
          a[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          const double vkp = v[k][p], vkq = v[k][q];
          v[k][p] = c * vkp - s * vkq;
          v[k][q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < 3; ++i) w[i] = a[i][i];
  // ascending eigenvalues (torch.linalg.eigh's order), eigenvectors in the columns of v
  for (int i = 0; i < 2; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (w[j] < w[i]) {
        std::swap(w[i], w[j]);
        for (int k = 0; k < 3; ++k) std::swap(v[k][i], v[k][j]);
      }
}

// geometry.py:236-338 with k_step = -1: 11 columns in pgeof's order (geometry.py:165-174)
// [linearity, planarity, scattering, verticality, nx, ny, nz, length, surface, volume,
// curvature]; `post`: verticality * 2 and the normal flipped to z >= 0 (geometry.py:121-124).
// nn [n, k] int64 (-1 = missing); add_self: the point itself is prepended (geometry.py:95-96).
int spt_cpu_point_geof_f32(const float* xyz, int64_t n, const int64_t* nn, int k, int add_self,
                           int k_min, int post, float* feats, int nthreads) {
  if (n < 0 || k < 0) return -1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static, 4096)
  for (int64_t i = 0; i < n; ++i) {
    double s1[3] = {0, 0, 0}, s2[6] = {0, 0, 0, 0, 0, 0};
    int cnt = 0;
    bool have = false;
    double px = 0, py = 0, pz = 0;
    auto add = [&](int64_t j) {
      if (!have) {              // moments about the first element: no cancellation
        px = xyz[3 * j]; py = xyz[3 * j + 1]; pz = xyz[3 * j + 2];
        have = true;
      }
      const double dx = xyz[3 * j] - px, dy = xyz[3 * j + 1] - py, dz = xyz[3 * j + 2] - pz;
      s1[0] += dx; s1[1] += dy; s1[2] += dz;
      s2[0] += dx * dx; s2[1] += dx * dy; s2[2] += dx * dz;
      s2[3] += dy * dy; s2[4] += dy * dz; s2[5] += dz * dz;
      ++cnt;
    };
    if (add_self) add(i);
    for (int c = 0; c < k; ++c)
      if (nn[i * k + c] >= 0) add(nn[i * k + c]);
    float* f = feats + i * 11;
    if (cnt == 0 || cnt < k_min) {
      // geometry.py:312-321: too small a neighbourhood -> all zeros (an empty one has the
      // (1,1,1) / identity fallback of scatter_pca first, then the same masking when k_min >= 1)
      for (int q = 0; q < 11; ++q) f[q] = 0.f;
      if (cnt == 0 && k_min <= 0) {
        // eigenvalues (1,1,1), identity vectors
        const double l = 1.0;
        f[0] = 0.f; f[1] = 0.f; f[2] = (float)(l / (l + 1e-3));
        f[3] = (float)((1.0 / (sqrt(3.0) + 1e-8)) * (post ? 2.0 : 1.0));
        f[4] = 1.f; f[5] = 0.f; f[6] = 0.f;
        f[7] = 1.f; f[8] = (float)sqrt(1.0 + 1e-6); f[9] = (float)cbrt(1.0 + 1e-9);
        f[10] = (float)(l / (3.0 + 1e-3));
      }
      continue;
    }
    const double inv = 1.0 / cnt;
    const double mx = s1[0] * inv, my = s1[1] * inv, mz = s1[2] * inv;
    double a[3][3], w[3], v[3][3];
    a[0][0] = s2[0] * inv - mx * mx;
    a[0][1] = a[1][0] = s2[1] * inv - mx * my;
    a[0][2] = a[2][0] = s2[2] * inv - mx * mz;
    a[1][1] = s2[3] * inv - my * my;
    a[1][2] = a[2][1] = s2[4] * inv - my * mz;
    a[2][2] = s2[5] * inv - mz * mz;
    eigh3_jacobi(a, w, v);
    for (int q = 0; q < 3; ++q) w[q] = w[q] > 0 ? w[q] : 0;      // scatter.py:123
    const double l1 = sqrt(w[2]), l2 = sqrt(w[1]), l3 = sqrt(w[0]);
    double nx = v[0][0], ny = v[1][0], nz = v[2][0];              // geometry.py:290
    double u[3];
    for (int r = 0; r < 3; ++r) u[r] = fabs(v[r][0]) * w[0] + fabs(v[r][1]) * w[1] + fabs(v[r][2]) * w[2];
    double vert = u[2] / (sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]) + 1e-8);
    if (post) {
      vert *= 2.0;
      if (nz < 0) { nx = -nx; ny = -ny; nz = -nz; }
    }
    f[0] = (float)((l1 - l2) / (l1 + 1e-3));
    f[1] = (float)((l2 - l3) / (l1 + 1e-3));
    f[2] = (float)(l3 / (l1 + 1e-3));
    f[3] = (float)vert;
    f[4] = (float)nx; f[5] = (float)ny; f[6] = (float)nz;
    f[7] = (float)l1;
    f[8] = (float)sqrt(l1 * l2 + 1e-6);
    f[9] = (float)cbrt(l1 * l2 * l3 + 1e-9);
    f[10] = (float)(l3 / (l1 + l2 + l3 + 1e-3));
  }
  return 0;
}

int spt_cpu_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

}  // extern "C"
