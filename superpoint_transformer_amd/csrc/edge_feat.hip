// On-the-fly horizontal edge features + symmetrisation + self loops, fused
// (the step that runs on device every batch right before
// the attention stages).  Replaces
//   _on_the_fly_horizontal_edge_features   src/transforms/graph.py:1135-1277
//   NAGAddSelfLoops._process               src/transforms/graph.py:1419-1452
// From the E trimmed (i<j) edges with their 7 stored attributes
// [mean_off(3), std_off(3), mean_dist] and the node attributes pos, normal,
// log_length, log_surface, log_volume, log_size it writes the final
//   edge_index [2, 2E (+N)] = [ (s,t) | (t,s) | (i,i) ]
//   edge_attr  [2E (+N), 18] = [mean_off(3) std_off(3) mean_dist angle_source
//        angle_target normal_angle log_length log_surface log_volume log_size
//        centroid_dir(3) centroid_dist]          (f_list order, graph.py:1186-1275)
// in one pass: the reference materialises ~14 [E, .] temporaries and two cats.
#include <math.h>

#include "common.hpp"

namespace spt {

__device__ __forceinline__ float clip1(float v) {
  // graph.py:1206-1208: NaN -> 0 (0/0 directions), then clip to [-1, 1]
  if (v != v) return 0.f;
  return fminf(fmaxf(v, -1.f), 1.f);
}

__global__ __launch_bounds__(256) void edge_feat_kernel(
    const int64_t* __restrict__ se, int64_t E, int64_t N, const float* __restrict__ ea7,
    const float* __restrict__ pos, const float* __restrict__ normal,
    const float* __restrict__ ll, const float* __restrict__ ls,
    const float* __restrict__ lv, const float* __restrict__ lz, int self_loops,
    int64_t* __restrict__ ei, float* __restrict__ out) {
  const int64_t etot = 2 * E + (self_loops ? N : 0);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < E + (self_loops ? N : 0);
       i += stride) {
    if (i >= E) {  // self loop: zero features (add_self_loops(fill_value=0))
      const int64_t n = i - E, row = 2 * E + n;
      ei[row] = n;
      ei[etot + row] = n;
#pragma unroll
      for (int k = 0; k < 18; ++k) out[row * 18 + k] = 0.f;
      continue;
    }
    const int64_t s = se[i], t = se[E + i];
    float f[18], fb[18];
    const float mx = ea7[i * 7 + 0], my = ea7[i * 7 + 1], mz = ea7[i * 7 + 2];
    const float nrm = sqrtf(mx * mx + my * my + mz * mz);
    const float dx = clip1(mx / nrm), dy = clip1(my / nrm), dz = clip1(mz / nrm);
    const float nsx = normal[s * 3], nsy = normal[s * 3 + 1], nsz = normal[s * 3 + 2];
    const float ntx = normal[t * 3], nty = normal[t * 3 + 1], ntz = normal[t * 3 + 2];
    f[0] = mx; f[1] = my; f[2] = mz;
    f[3] = ea7[i * 7 + 3]; f[4] = ea7[i * 7 + 4]; f[5] = ea7[i * 7 + 5];
    f[6] = ea7[i * 7 + 6];
    f[7] = fabsf(dx * nsx + dy * nsy + dz * nsz);       // angle_source
    f[8] = fabsf(dx * ntx + dy * nty + dz * ntz);       // angle_target
    f[9] = fabsf(nsx * ntx + nsy * nty + nsz * ntz);    // normal_angle
    f[10] = ll[s] - ll[t];
    f[11] = ls[s] - ls[t];
    f[12] = lv[s] - lv[t];
    f[13] = lz[s] - lz[t];
    const float cx = pos[t * 3] - pos[s * 3], cy = pos[t * 3 + 1] - pos[s * 3 + 1],
                cz = pos[t * 3 + 2] - pos[s * 3 + 2];
    const float cd = sqrtf(cx * cx + cy * cy + cz * cz);
    f[14] = clip1(cx / cd); f[15] = clip1(cy / cd); f[16] = clip1(cz / cd);
    f[17] = sqrtf(cd);                                   // graph.py:1251: sqrt of the distance
    // the flipped edge (graph.py: torch.cat((f, +-f)))
#pragma unroll
    for (int k = 0; k < 18; ++k) fb[k] = f[k];
    fb[0] = -f[0]; fb[1] = -f[1]; fb[2] = -f[2];
    fb[10] = -f[10]; fb[11] = -f[11]; fb[12] = -f[12]; fb[13] = -f[13];
    fb[14] = -f[14]; fb[15] = -f[15]; fb[16] = -f[16];
    ei[i] = s;
    ei[etot + i] = t;
    ei[E + i] = t;
    ei[etot + E + i] = s;
#pragma unroll
    for (int k = 0; k < 18; ++k) {
      out[i * 18 + k] = f[k];
      out[(E + i) * 18 + k] = fb[k];
    }
  }
}

// ---- vertical (child -> parent) edge features, src/transforms/graph.py:1335-1416 -------------
// out[i] = [dir(3), sqrt(dist), |<n_i, n_p>|, log_length_p - log_length_i, ... surface, volume,
// size] with p = super_index[i]; 0/0 directions -> 0, directions clipped to [-1, 1].
__global__ __launch_bounds__(256) void vertical_edge_feat_kernel(
    const int64_t* __restrict__ sup, int64_t n, const float* __restrict__ cpos,
    const float* __restrict__ cnrm, const float* __restrict__ cll, const float* __restrict__ cls,
    const float* __restrict__ clv, const float* __restrict__ clz, const float* __restrict__ ppos,
    const float* __restrict__ pnrm, const float* __restrict__ pll, const float* __restrict__ pls,
    const float* __restrict__ plv, const float* __restrict__ plz, float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int64_t p = sup[i];
    const float dx = ppos[p * 3] - cpos[i * 3], dy = ppos[p * 3 + 1] - cpos[i * 3 + 1],
                dz = ppos[p * 3 + 2] - cpos[i * 3 + 2];
    const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
    float* o = out + i * 9;
    o[0] = clip1(dx / dist); o[1] = clip1(dy / dist); o[2] = clip1(dz / dist);
    o[3] = sqrtf(dist);
    o[4] = fabsf(cnrm[i * 3] * pnrm[p * 3] + cnrm[i * 3 + 1] * pnrm[p * 3 + 1] +
                 cnrm[i * 3 + 2] * pnrm[p * 3 + 2]);
    o[5] = pll[p] - cll[i];
    o[6] = pls[p] - cls[i];
    o[7] = plv[p] - clv[i];
    o[8] = plz[p] - clz[i];
  }
}

// ---- symmetric edge features of the panoptic edge-affinity head ---------------------------
// src/models/panoptic.py:477-480:  x_edge = x[obj_edge_index];
//   out[e] = cat(|x[a_e] - x[b_e]|, (x[a_e] + x[b_e]) / 2)          [E, 2C]
// One kernel instead of a [2, E, C] gather + sub / abs / add / mul / cat (5 passes over
// [E, C]).  A lane group of C/4 lanes owns an edge: two 16-byte row reads, two row writes.
__global__ __launch_bounds__(256) void edge_affinity_fwd_kernel(
    const float* __restrict__ x, const int64_t* __restrict__ ea, const int64_t* __restrict__ eb,
    int64_t E, int C, float* __restrict__ out) {
  const int lpr = C >> 2;                       // lanes per edge
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < E * lpr; t += stride) {
    const int64_t e = t / lpr;
    const int k = (int)(t - e * lpr) << 2;
    const float4 a = *reinterpret_cast<const float4*>(x + ea[e] * C + k);
    const float4 b = *reinterpret_cast<const float4*>(x + eb[e] * C + k);
    float* o = out + e * 2 * C + k;
    *reinterpret_cast<float4*>(o) =
        make_float4(fabsf(a.x - b.x), fabsf(a.y - b.y), fabsf(a.z - b.z), fabsf(a.w - b.w));
    *reinterpret_cast<float4*>(o + C) = make_float4((a.x + b.x) / 2.f, (a.y + b.y) / 2.f,
                                                    (a.z + b.z) / 2.f, (a.w + b.w) / 2.f);
  }
}

// gend[e] = d loss / d x[a_e] through edge e, gend[E + e] = the same for x[b_e]:
//   sign(x_a - x_b) * g_abs + g_mean / 2   and   -sign(x_a - x_b) * g_abs + g_mean / 2
// (torch's abs backward: sign(0) = 0).  The per-node sums are a segment reduce over
// cat(a, b) on the CSR kernels - deterministic, no atomics.
__global__ __launch_bounds__(256) void edge_affinity_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gout, const int64_t* __restrict__ ea,
    const int64_t* __restrict__ eb, int64_t E, int C, float* __restrict__ gend) {
  const int lpr = C >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < E * lpr; t += stride) {
    const int64_t e = t / lpr;
    const int k = (int)(t - e * lpr) << 2;
    const float4 a = *reinterpret_cast<const float4*>(x + ea[e] * C + k);
    const float4 b = *reinterpret_cast<const float4*>(x + eb[e] * C + k);
    const float4 g1 = *reinterpret_cast<const float4*>(gout + e * 2 * C + k);
    const float4 g2 = *reinterpret_cast<const float4*>(gout + e * 2 * C + C + k);
    auto sgn = [](float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); };
    const float4 s = make_float4(sgn(a.x - b.x), sgn(a.y - b.y), sgn(a.z - b.z), sgn(a.w - b.w));
    *reinterpret_cast<float4*>(gend + e * C + k) = make_float4(
        s.x * g1.x + g2.x / 2.f, s.y * g1.y + g2.y / 2.f, s.z * g1.z + g2.z / 2.f, s.w * g1.w + g2.w / 2.f);
    *reinterpret_cast<float4*>(gend + (E + e) * C + k) = make_float4(
        -s.x * g1.x + g2.x / 2.f, -s.y * g1.y + g2.y / 2.f, -s.z * g1.z + g2.z / 2.f, -s.w * g1.w + g2.w / 2.f);
  }
}

}  // namespace spt

using namespace spt;

extern "C" int spt_vertical_edge_features_f32(
    const int64_t* super_index, int64_t n, const float* child_pos, const float* child_normal,
    const float* child_log_length, const float* child_log_surface, const float* child_log_volume,
    const float* child_log_size, const float* parent_pos, const float* parent_normal,
    const float* parent_log_length, const float* parent_log_surface,
    const float* parent_log_volume, const float* parent_log_size, float* v_edge_attr,
    spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(n >= 0, "bad shape");
  if (n == 0) return 0;
  SPT_CHECK_ARG(super_index && child_pos && child_normal && child_log_length && child_log_surface &&
                child_log_volume && child_log_size && parent_pos && parent_normal &&
                parent_log_length && parent_log_surface && parent_log_volume && parent_log_size &&
                v_edge_attr, "null pointer");
  vertical_edge_feat_kernel<<<stream_grid(n, 256), 256, 0, stream>>>(
      super_index, n, child_pos, child_normal, child_log_length, child_log_surface,
      child_log_volume, child_log_size, parent_pos, parent_normal, parent_log_length,
      parent_log_surface, parent_log_volume, parent_log_size, v_edge_attr);
  SPT_CHECK_LAUNCH();
  return 0;
}

extern "C" int spt_edge_affinity_features_f32(const float* x, int64_t n, int C,
                                              const int64_t* edge_a, const int64_t* edge_b,
                                              int64_t e, float* out, spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(n >= 0 && e >= 0 && C >= 4 && C % 4 == 0, "bad shape (C must be a multiple of 4)");
  if (e == 0) return 0;
  SPT_CHECK_ARG(x && edge_a && edge_b && out, "null pointer");
  edge_affinity_fwd_kernel<<<stream_grid(e * (C / 4), 256), 256, 0, stream>>>(x, edge_a, edge_b, e,
                                                                             C, out);
  SPT_CHECK_LAUNCH();
  return 0;
}

extern "C" int spt_edge_affinity_features_bwd_f32(const float* x, const float* gout, int64_t n,
                                                  int C, const int64_t* edge_a,
                                                  const int64_t* edge_b, int64_t e, float* gend,
                                                  spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(n >= 0 && e >= 0 && C >= 4 && C % 4 == 0, "bad shape (C must be a multiple of 4)");
  if (e == 0) return 0;
  SPT_CHECK_ARG(x && gout && edge_a && edge_b && gend, "null pointer");
  edge_affinity_bwd_kernel<<<stream_grid(e * (C / 4), 256), 256, 0, stream>>>(x, gout, edge_a, edge_b,
                                                                             e, C, gend);
  SPT_CHECK_LAUNCH();
  return 0;
}

extern "C" int spt_horizontal_edge_features_f32(
    const int64_t* se, int64_t e, int64_t n, const float* edge_attr7, const float* pos,
    const float* normal, const float* log_length, const float* log_surface,
    const float* log_volume, const float* log_size, int add_self_loops,
    int64_t* edge_index_out, float* edge_attr_out, spt_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPT_CHECK_ARG(e >= 0 && n >= 0, "bad shape");
  const int64_t work = e + (add_self_loops ? n : 0);
  if (work == 0) return 0;
  SPT_CHECK_ARG(edge_index_out && edge_attr_out, "null output");
  SPT_CHECK_ARG(e == 0 || (se && edge_attr7 && pos && normal && log_length && log_surface &&
                           log_volume && log_size), "null input");
  edge_feat_kernel<<<stream_grid(work, 256), 256, 0, stream>>>(
      se, e, n, edge_attr7, pos, normal, log_length, log_surface, log_volume, log_size,
      add_self_loops, edge_index_out, edge_attr_out);
  SPT_CHECK_LAUNCH();
  return 0;
}
