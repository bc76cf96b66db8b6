"""GPU parity of the segment-level preprocessing rows: the
per-segment sampler, scatter_std, scatter_mean_orientation and the segment
features, through the C ABI.

Bars: sample counts / pointers BIT-EXACT against the reference's own output
(tests/golden/segment_features.npz) and the oracle; which elements are drawn is
RNG-dependent, so the draw is held to the sampler's contract (membership, no
duplicates, mask honoured, reproducible per seed, uniform inclusion
frequencies).  Float outputs: 1e-5 (std, mean orientation) and 1e-4
(eigen-features, away from degenerate spectra) against the f64 reference."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import spt_oracle as O

pytestmark = pytest.mark.gpu

G = load_golden("segment_features.npz")


def t(name):
    return torch.from_numpy(G[name])


def test_sample_pointers_are_bit_exact_and_draw_honours_the_contract(dev):
    from superpoint_transformer_amd.segment import sparse_sample
    idx = t("sweep_idx")
    for c, (n_max, n_min) in enumerate(G["sweep_cases"].tolist()):
        s, p = sparse_sample(idx.to(dev), n_max=n_max, n_min=n_min, return_pointers=True, seed=c)
        assert torch.equal(p.cpu(), t(f"sweep_ptr_{c}")), (n_max, n_min)
        assert O.check_sparse_sample(idx, s.cpu(), p.cpu(), n_max, n_min) == []
    mask = t("sweep_mask")
    s, p = sparse_sample(idx.to(dev), n_max=32, n_min=5, mask=mask.to(dev), return_pointers=True,
                         seed=11)
    assert torch.equal(p.cpu(), t("sweep_ptr_mask"))
    assert O.check_sparse_sample(idx, s.cpu(), p.cpu(), 32, 5, mask) == []
    # positions instead of a boolean mask (tensor_idx semantics, sparse.py:196-199)
    s2, p2 = sparse_sample(idx.to(dev), n_max=32, n_min=5, mask=torch.where(mask)[0].to(dev),
                           return_pointers=True, seed=11)
    assert torch.equal(p2, p) and torch.equal(s2, s)


def test_sampler_is_reproducible_per_seed_and_varies_across_seeds(dev):
    from superpoint_transformer_amd.segment import sparse_sample
    idx = t("sweep_idx").to(dev)
    a = sparse_sample(idx, 32, 5, seed=123)
    b = sparse_sample(idx, 32, 5, seed=123)
    c = sparse_sample(idx, 32, 5, seed=124)
    assert torch.equal(a, b)
    assert a.shape == c.shape and not torch.equal(a, c)
    torch.manual_seed(5)
    d = sparse_sample(idx, 32, 5)
    torch.manual_seed(5)
    e = sparse_sample(idx, 32, 5)
    assert torch.equal(d, e)


def test_sampler_draws_uniformly(dev):
    """Inclusion frequency of every element of a segment = n_samples / size, and of
    every PAIR of elements of a small segment = the hypergeometric value (chi-square,
    5 sigma)."""
    from superpoint_transformer_amd.segment import sparse_sample
    sizes = [7, 40, 300, 3, 64]
    idx = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    perm = torch.randperm(idx.numel(), generator=torch.Generator().manual_seed(3))
    idx = idx[perm]
    n_samples, _ = O.sparse_sample_counts(idx, 8, 2)
    trials = 4000
    hits = torch.zeros(idx.numel(), dtype=torch.long)
    seg0 = torch.where(idx == 0)[0]
    pos_in_seg0 = {int(e): j for j, e in enumerate(seg0)}
    pair = np.zeros((7, 7))
    d = idx.to(dev)
    for s in range(trials):
        smp = sparse_sample(d, 8, 2, seed=1000 + s).cpu()
        hits[smp] += 1
        mine = [pos_in_seg0[int(e)] for e in smp if int(e) in pos_in_seg0]
        for a in mine:
            for b in mine:
                pair[a, b] += 1
    for g, size in enumerate(sizes):
        k = int(n_samples[g])
        h = hits[idx == g].double()
        p = k / size
        if k == size:
            assert bool((h == trials).all())
            continue
        chi2 = float(((h - trials * p) ** 2 / (trials * p * (1 - p))).sum())
        # sum of `size` ~N(0,1)^2 terms with one linear constraint, scaled by (size-1)/size
        dof = size - 1
        assert chi2 < dof + 5 * (2 * dof) ** 0.5, (g, chi2, dof)
    k = int(n_samples[0])
    ppair = k * (k - 1) / (7 * 6)
    off = pair[~np.eye(7, dtype=bool)]
    z = (off - trials * ppair) / (trials * ppair * (1 - ppair)) ** 0.5
    assert np.abs(z).max() < 5, z


def test_sampler_edge_cases(dev):
    from superpoint_transformer_amd.segment import sparse_sample
    # one element, one segment
    s, p = sparse_sample(torch.zeros(1, dtype=torch.long, device=dev), 32, 1, return_pointers=True)
    assert s.tolist() == [0] and p.tolist() == [0, 1]
    # a single huge segment next to empty ones
    idx = torch.full((100000,), 3, dtype=torch.long)
    s, p = sparse_sample(idx.to(dev), 128, 32, return_pointers=True, seed=1, num_segments=6)
    assert p.tolist() == [0, 0, 0, 0, 128, 128, 128]
    assert torch.unique(s).numel() == 128
    # everything masked out
    s, p = sparse_sample(idx.to(dev), 128, 32, mask=torch.zeros(100000, dtype=torch.bool, device=dev),
                         return_pointers=True, num_segments=4)
    assert s.numel() == 0 and p.tolist() == [0, 0, 0, 0, 0]
    with pytest.raises(ValueError):
        sparse_sample(idx.to(dev), 4, 5)
    with pytest.raises(RuntimeError):
        sparse_sample(idx, 4, 1)                       # CPU tensor: no fallback


def test_scatter_std_matches_the_f64_oracle(dev):
    from superpoint_transformer_amd.segment import scatter_std
    idx, feat = t("scene_idx"), t("scene_feat")
    n = int(G["scene_num_seg"])
    for x in (feat, feat[:, :1].contiguous(), torch.cat([feat, feat * 3 + 100, feat], 1)):
        got = scatter_std(x.to(dev), idx.to(dev), n).cpu().double()
        ref = O.scatter_std(x.double(), idx, 0, None, n)
        assert torch.allclose(got, ref, atol=1e-5, rtol=1e-5)
    got = scatter_std(feat[:, 0].contiguous().to(dev), idx.to(dev), n + 3)   # 1-D, empty tail
    assert got.shape == (n + 3,) and bool((got[n:] == 0).all())


def test_mean_orientation_matches_the_reference(dev):
    from superpoint_transformer_amd.segment import scatter_mean_orientation
    got = scatter_mean_orientation(t("scene_normal").to(dev), t("scene_idx").to(dev),
                                   int(G["scene_num_seg"])).cpu().double()
    ref = t("scene_mean_normal")
    assert torch.allclose(got, ref, atol=1e-5), (got - ref).abs().max()
    # empty segments -> 0 (scatter_mean of nothing, then 0 / (0 + 1e-4))
    got = scatter_mean_orientation(t("scene_normal").to(dev), t("scene_idx").to(dev),
                                   int(G["scene_num_seg"]) + 2)
    assert bool((got[-2:] == 0).all())


def test_segment_features_match_the_reference_on_its_own_samples(dev):
    from superpoint_transformer_amd.segment import segment_features
    pos, idx, n = t("scene_pos"), t("scene_idx"), int(G["scene_num_seg"])
    attrs = {"normal": t("scene_normal"), "feat": t("scene_feat")}
    f = segment_features(pos.to(dev), idx.to(dev), n,
                         samples=(t("scene_samples").to(dev), t("scene_ptr").to(dev)),
                         point_attrs={k: v.to(dev) for k, v in attrs.items()})
    ref = O.segment_features(pos.double(), idx, n, t("scene_samples"), t("scene_ptr"),
                             point_attrs={k: v.double() for k, v in attrs.items()})
    # eigen-features are ill-conditioned where two eigenvalues coincide: judge where the
    # f64 spectrum is separated (same criterion as tests/test_neighbors_gpu.py)
    width = t("scene_ptr")[1:] - t("scene_ptr")[:-1]
    ok = width >= 5
    for key in ("linearity", "planarity", "scattering", "curvature", "log_length", "log_surface",
                "log_volume"):
        g = f[key].cpu().double()
        assert torch.allclose(g[ok], ref[key][ok], atol=1e-4), (key, (g - ref[key]).abs().max())
        assert bool((f[key].cpu()[~ok] == 0).all())          # < k_min = 5 samples -> zeroed
    for key in ("linearity", "planarity", "scattering", "verticality", "curvature"):
        assert torch.allclose(f[key].cpu().double(), t(f"scene_geof_{key}"), atol=2e-4), key
    # normals: up to the eigen-solver's sign, already canonicalised to z >= 0
    gn, rn = f["normal"].cpu().double(), t("scene_geof_normal")
    planar = (t("scene_geof_planarity").view(-1) > 0.3) & ok
    assert torch.allclose(gn[planar], rn[planar], atol=1e-3)
    assert torch.allclose(f["log_size"].cpu().double(), ref["log_size"].double(), atol=1e-6)
    assert torch.allclose(f["mean_normal"].cpu().double(), t("scene_mean_normal"), atol=1e-5)
    assert torch.allclose(f["mean_feat"].cpu().double(), ref["mean_feat"], atol=1e-6)
    assert torch.allclose(f["std_feat"].cpu().double(), ref["std_feat"], atol=1e-5)
    assert torch.allclose(f["std_normal"].cpu().double(), ref["std_normal"], atol=1e-5)


def test_segment_features_with_their_own_draw(dev):
    """End to end with the GPU sampler: same statistics as with the reference's draw
    for the segments that are sampled entirely (size <= n_min)."""
    from superpoint_transformer_amd.segment import segment_features
    pos, idx, n = t("scene_pos"), t("scene_idx"), int(G["scene_num_seg"])
    a = segment_features(pos.to(dev), idx.to(dev), n, n_max=32, n_min=5, seed=1)
    b = segment_features(pos.to(dev), idx.to(dev), n, n_max=32, n_min=5, seed=1)
    for k in a:
        assert torch.equal(a[k], b[k])
    full = torch.bincount(idx, minlength=n) <= 5
    ref = O.segment_features(pos.double(), idx, n, t("scene_samples"), t("scene_ptr"))
    for key in ("linearity", "planarity", "log_length"):
        assert torch.allclose(a[key].cpu().double()[full], ref[key][full], atol=1e-4)


def test_sampler_at_scene_scale(dev):
    """15 M points / 428 571 segments: contract by size-independent properties."""
    from superpoint_transformer_amd.segment import sparse_sample
    n, s = 15_000_000, 428_571
    idx = torch.randint(0, s, (n,), device=dev, generator=torch.Generator(dev).manual_seed(0))
    smp, ptr = sparse_sample(idx, 32, 5, return_pointers=True, seed=9, num_segments=s)
    size = torch.bincount(idx, minlength=s)
    ns = (32 * torch.tanh(size / 32)).floor().long().clamp(min=5).clamp(max=size)
    got = ptr[1:] - ptr[:-1]
    # the device tanh may round differently from torch's exactly at integer crossings
    assert int((got != ns).sum()) == 0 or bool(((got - ns).abs() <= 1).all())
    assert int(ptr[-1]) == smp.numel()
    seg = torch.repeat_interleave(torch.arange(s, device=dev), got)
    assert torch.equal(idx[smp], seg)
    assert torch.unique(smp).numel() == smp.numel()


def test_sampler_and_statistics_on_degenerate_inputs(dev):
    from superpoint_transformer_amd.segment import (sparse_sample, scatter_std,
                                                    scatter_mean_orientation)
    empty = torch.zeros(0, dtype=torch.long, device=dev)
    s, p = sparse_sample(empty, 8, 1, return_pointers=True, num_segments=3)
    assert s.numel() == 0 and p.tolist() == [0, 0, 0, 0]
    x = torch.rand(5, 2, device=dev)
    idx = torch.tensor([2, 2, 2, 0, 2], device=dev)
    std = scatter_std(x, idx, 4)
    assert std.shape == (4, 2) and bool((std[[0, 1, 3]] == 0).all())     # singletons / empties
    ref = O.scatter_std(x.cpu().double(), idx.cpu(), 0, None, 4)
    assert torch.allclose(std.cpu().double(), ref, atol=1e-6)
    o = scatter_mean_orientation(torch.tensor([[0.0, 0.0, -2.0]], device=dev),
                                 torch.zeros(1, dtype=torch.long, device=dev), 1)
    assert torch.allclose(o.cpu(), torch.tensor([[0.0, 0.0, 1.0]]), atol=1e-4)   # flipped to z+
