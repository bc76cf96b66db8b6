"""Golden fixtures for the segment-level preprocessing rows:
``sparse_sample`` pointers, ``scatter_mean_orientation`` and the segment
geometric features, produced by the REFERENCE'S OWN functions
(src/utils/sparse.py, src/utils/scatter.py, src/utils/geometry.py imported
verbatim by path through make_golden.install_reference_import_hooks; the absent
torch_scatter is bound to the CPU restatement of oracle/spt_oracle.py, the absent
pgeof C++ branch is routed to the reference's own torch branch).

Usage (build container only): python tests/golden/make_golden_segment.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def synth_segments(gen, sizes):
    """Unsorted segment index with the given sizes (some of them 0)."""
    idx = torch.repeat_interleave(torch.arange(len(sizes)), torch.as_tensor(sizes))
    return idx[torch.randperm(idx.numel(), generator=gen)]


def main():
    U, _ = mg.install_reference_import_hooks()
    import src.utils.geometry as G
    # CPU tensors make geometric_features take the pgeof (C++) branch; route it to
    # the reference's torch implementation of the same features
    G.geometric_features_pgeof = lambda xyz, nn, k_min=5, k_step=-1, k_min_search=25: \
        G.geometric_features_torch(xyz, nn, k_min=k_min, k_step=k_step, k_min_search=k_min_search)
    gen = torch.Generator().manual_seed(20240607)
    torch.manual_seed(7)
    # the reference's CPU sampler draws from numpy's GLOBAL generator (sparse_sample ->
    # fast_randperm -> np.random.shuffle, src/utils/tensor.py:330-339): seed it, or two runs of
    # this script store different draws
    np.random.seed(20240607)

    out = {}
    # --- 1. sampling counts over a sweep of segment sizes ---------------------------------
    sizes = list(range(0, 420)) + [1000, 2047, 5000]
    sizes[0] = 1                      # the last segment must exist for idx.max() + 1
    sizes = sizes[::-1]               # ... and a few empty ones in the middle
    sizes[10] = 0
    sizes[200] = 0
    idx = synth_segments(gen, sizes)
    out["sweep_idx"] = idx
    cases = [(32, 1), (32, 5), (128, 32), (8, 0), (0, 0), (3, 3)]
    for c, (n_max, n_min) in enumerate(cases):
        s, p = U.sparse_sample(idx, n_max=n_max, n_min=n_min, return_pointers=True)
        out[f"sweep_ptr_{c}"] = p
        out[f"sweep_samples_{c}"] = s
    out["sweep_cases"] = np.asarray(cases)
    mask = torch.rand(idx.numel(), generator=gen) < 0.4
    s, p = U.sparse_sample(idx, n_max=32, n_min=5, mask=mask, return_pointers=True)
    out["sweep_mask"] = mask
    out["sweep_ptr_mask"] = p
    out["sweep_samples_mask"] = s

    # --- 2. a small scene: planar / linear / volumetric segments ---------------------------
    nseg = 60
    ssz = torch.randint(1, 120, (nseg,), generator=gen).tolist()
    ssz[7] = 1                 # (an empty segment makes the reference's CPU eigh raise on NaNs)
    ssz[20] = 2
    ssz[21] = 4
    ssz[22] = 5
    sidx = synth_segments(gen, ssz)
    n0 = sidx.numel()
    centre = torch.randn(nseg, 3, generator=gen) * 5
    shape = torch.rand(nseg, 3, generator=gen) * torch.tensor([2.0, 1.0, 0.3])
    kind = torch.arange(nseg) % 3
    shape[kind == 1, 1:] *= 0.05          # lines
    shape[kind == 2, 2] *= 0.02           # planes
    rot = torch.linalg.qr(torch.randn(nseg, 3, 3, generator=gen))[0]
    local = torch.randn(n0, 3, generator=gen) * shape[sidx]
    pos = (centre[sidx] + torch.einsum("nij,nj->ni", rot[sidx], local)).float()
    normal = torch.nn.functional.normalize(
        rot[sidx][:, :, 2] + 0.3 * torch.randn(n0, 3, generator=gen), dim=1)
    normal = (normal * torch.where(torch.rand(n0, 1, generator=gen) < 0.5, -1.0, 1.0)).float()
    feat = torch.rand(n0, 5, generator=gen).float()
    out.update(scene_pos=pos, scene_idx=sidx, scene_normal=normal, scene_feat=feat,
               scene_num_seg=np.int64(nseg))

    samples, ptr = U.sparse_sample(sidx, n_max=32, n_min=5, return_pointers=True)
    out.update(scene_samples=samples, scene_ptr=ptr)
    # graph.py:234-246 (without the 1e-8 jitter: below f32 resolution at these coordinates)
    nn = U.csr_to_dense(ptr, U.arange_interleave(ptr[1:] - ptr[:-1]), samples, fill_value=-1)
    torch.set_default_dtype(torch.float64)      # scatter_pca allocates `cov` in the default dtype
    f = G.geometric_features(pos.double(), nn, add_self_as_neighbor=False)
    torch.set_default_dtype(torch.float32)
    for k, v in f.items():
        out[f"scene_geof_{k}"] = v
    out["scene_mean_normal"] = U.scatter_mean_orientation(normal.double(), sidx)
    out["scene_mean_normal_f32"] = U.scatter_mean_orientation(normal, sidx)
    mg.save("segment_features.npz", **out)


if __name__ == "__main__":
    main()
