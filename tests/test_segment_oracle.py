"""CPU: the oracle's restatement of the segment-level preprocessing rows (f2)
against fixtures produced by the reference's own functions
(tests/golden/make_golden_segment.py)."""
import os

import numpy as np
import torch

from oracle import spt_oracle as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "segment_features.npz"))


def t(name):
    return torch.from_numpy(G[name])


def test_sample_counts_and_pointers_match_the_reference():
    idx = t("sweep_idx")
    for c, (n_max, n_min) in enumerate(G["sweep_cases"].tolist()):
        _, ptr = O.sparse_sample_counts(idx, n_max, n_min)
        assert torch.equal(ptr, t(f"sweep_ptr_{c}")), (n_max, n_min)
        # the reference's own draw satisfies the contract the GPU sampler is held to
        assert O.check_sparse_sample(idx, t(f"sweep_samples_{c}"), t(f"sweep_ptr_{c}"),
                                     n_max, n_min) == []
    mask = t("sweep_mask")
    _, ptr = O.sparse_sample_counts(idx, 32, 5, mask)
    assert torch.equal(ptr, t("sweep_ptr_mask"))
    assert O.check_sparse_sample(idx, t("sweep_samples_mask"), ptr, 32, 5, mask) == []


def test_mean_orientation_matches_the_reference():
    got = O.scatter_mean_orientation(t("scene_normal").double(), t("scene_idx"))
    assert torch.allclose(got, t("scene_mean_normal"), atol=1e-12)


def test_segment_geometric_features_match_the_reference():
    pos, idx = t("scene_pos").double(), t("scene_idx")
    f = O.segment_features(pos, idx, int(G["scene_num_seg"]), t("scene_samples"), t("scene_ptr"))
    for key in ("linearity", "planarity", "scattering", "verticality", "curvature", "normal"):
        assert torch.allclose(f[key], t(f"scene_geof_{key}"), atol=1e-9), key
    for key in ("length", "surface", "volume"):
        assert torch.allclose(f[f"log_{key}"], torch.log(t(f"scene_geof_{key}") + 1), atol=1e-9), key
