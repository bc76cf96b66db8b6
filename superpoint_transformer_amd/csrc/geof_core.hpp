// Symmetric 3x3 eigen-decomposition (cyclic Jacobi, f64, in registers) and the 11 SPG
// eigenfeatures of one neighbourhood from its moment sums: shared by point_geof.hip (features
// from stored neighbour lists) and grid_knn.hip (features out of the kNN kernel itself, while
// the neighbours' rows are still at hand).  Column order: src/utils/geometry.py:165-174.
#pragma once
#include <math.h>

#include "common.hpp"

namespace spt {

__device__ __forceinline__ void jacobi_rotate(double (&a)[3][3], double (&v)[3][3], int p,
                                              int q) {
  const double apq = a[p][q];
  if (fabs(apq) <= 1e-300) return;
  const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
  const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
  const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
  const int r = 3 - p - q;
  const double app = a[p][p], aqq = a[q][q], arp = a[r][p], arq = a[r][q];
  a[p][p] = app - t * apq;
  a[q][q] = aqq + t * apq;
  a[p][q] = a[q][p] = 0.0;
  a[r][p] = a[p][r] = c * arp - s * arq;
  a[r][q] = a[q][r] = s * arp + c * arq;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double vkp = v[k][p], vkq = v[k][q];
    v[k][p] = c * vkp - s * vkq;
    v[k][q] = s * vkp + c * vkq;
  }
}

// eigenvalues ascending in w, eigenvectors in the COLUMNS of v
__device__ __forceinline__ void eigh3(double (&a)[3][3], double (&w)[3], double (&v)[3][3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) v[i][j] = (i == j) ? 1.0 : 0.0;
  const double scale = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]) + 1e-300;
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
    if (off <= 1e-18 * scale) break;
    jacobi_rotate(a, v, 0, 1);
    jacobi_rotate(a, v, 0, 2);
    jacobi_rotate(a, v, 1, 2);
  }
  w[0] = a[0][0]; w[1] = a[1][1]; w[2] = a[2][2];
  // sort ascending (3-element network), swapping columns of v alongside
#define SPT_SWAP(i, j)                                    \
  if (w[i] > w[j]) {                                      \
    const double tw = w[i]; w[i] = w[j]; w[j] = tw;       \
    _Pragma("unroll") for (int k = 0; k < 3; ++k) {       \
      const double tv = v[k][i]; v[k][i] = v[k][j]; v[k][j] = tv; \
    }                                                     \
  }
  SPT_SWAP(0, 1)
  SPT_SWAP(1, 2)
  SPT_SWAP(0, 1)
#undef SPT_SWAP
}

// moments (sums of d and d d^T about an origin, count) -> the 11 features of one point
__device__ __forceinline__ void finish_features(const double s1[3], const double s2[6], int cnt,
                                                int k_min, int post, float* __restrict__ out) {
  double w[3] = {1.0, 1.0, 1.0};
  double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  if (cnt > 0) {                       // cnt == 0: scatter.py:113-118 -> (1,1,1) / I
    const double inv = 1.0 / (double)cnt;
    const double mx = s1[0] * inv, my = s1[1] * inv, mz = s1[2] * inv;
    double a[3][3];
    a[0][0] = s2[0] * inv - mx * mx;
    a[0][1] = a[1][0] = s2[1] * inv - mx * my;
    a[0][2] = a[2][0] = s2[2] * inv - mx * mz;
    a[1][1] = s2[3] * inv - my * my;
    a[1][2] = a[2][1] = s2[4] * inv - my * mz;
    a[2][2] = s2[5] * inv - mz * mz;
    eigh3(a, w, v);
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) w[q] = w[q] > 0.0 ? w[q] : 0.0;     // scatter.py:123
  // geometry.py:292-315
  const double l1 = sqrt(w[2]), l2 = sqrt(w[1]), l3 = sqrt(w[0]);
  double f[11];
  f[0] = (l1 - l2) / (l1 + 1e-3);
  f[1] = (l2 - l3) / (l1 + 1e-3);
  f[2] = l3 / (l1 + 1e-3);
  double un[3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
    un[r] = fabs(v[r][0]) * w[0] + fabs(v[r][1]) * w[1] + fabs(v[r][2]) * w[2];
  f[3] = un[2] / (sqrt(un[0] * un[0] + un[1] * un[1] + un[2] * un[2]) + 1e-8);
  f[4] = v[0][0]; f[5] = v[1][0]; f[6] = v[2][0];                 // normal = smallest eigvec
  f[7] = l1;
  f[8] = sqrt(l1 * l2 + 1e-6);
  f[9] = cbrt(l1 * l2 * l3 + 1e-9);
  f[10] = l3 / (l1 + l2 + l3 + 1e-3);
  if (cnt < k_min) {                                              // geometry.py:318-327
#pragma unroll
    for (int q = 0; q < 11; ++q) f[q] = 0.0;
  }
  if (post) {                                                     // geometry.py:121,124
    f[3] *= 2.0;
    if (f[6] < 0.0) { f[4] = -f[4]; f[5] = -f[5]; f[6] = -f[6]; }
  }
#pragma unroll
  for (int q = 0; q < 11; ++q) out[q] = (float)f[q];
}

}  // namespace spt
