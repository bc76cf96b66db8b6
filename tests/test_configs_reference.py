"""The model configurations the benchmark runs ARE the reference's: ``hotpath.spt64_config`` /
``spt128_config`` / ``nano2_config`` / ``panoptic_config`` against the reference's own Hydra
YAML trees (configs/model/**, configs/datamodule/**, with the overrides of
configs/experiment/**), composed and resolved by ``tests/hydra_lite.py`` (Hydra / OmegaConf are
not installed here; the ``${eval:...}`` expressions are the configs' own python).  Build
container only: the YAMLs live under /root/reference."""
import os

import pytest

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/configs"),
                                reason="reference tree only exists in the build container")

# keys of the ``net`` node that are not constructor numbers / switches
SKIP = {"mlp_norm", "norm", "mlp_activation", "activation", "output_stage_wise"}


def _check(cfg, mine, ignore=()):
    for k, v in mine.items():
        if k in SKIP or k in ignore:
            continue
        assert cfg.get("model.net." + k) == v, (k, cfg.get("model.net." + k), v)


def test_spt64_is_the_s3dis_experiment():
    """configs/experiment/semantic/s3dis.yaml: datamodule semantic/s3dis, model semantic/spt-2."""
    import hydra_lite as H
    from superpoint_transformer_amd import hotpath
    c = H.compose("semantic/s3dis", "semantic/spt-2")
    assert (c.get("datamodule.num_hf_point"), c.get("datamodule.num_hf_edge"),
            c.get("datamodule.num_hf_segment"), c.get("datamodule.num_classes")) == (8, 18, 0, 13)
    _check(c, hotpath.spt64_config(8, 18))
    net = c.get("model.net")
    assert net["down_num_heads"] == 16 and net["qk_dim"] == 4 and net["in_rpe_dim"] == 32
    assert net["k_delta_rpe"] is False and net["down_attn_drop"] is None and net["pool"] == "max"
    assert c.get("model.multi_stage_loss_lambdas") == [1, 50]          # hotpath.SPTTrainStep.lambdas
    assert hotpath.NUM_CLASSES == 13


def test_spt128_is_the_kitti360_experiment():
    """configs/experiment/semantic/kitti360.yaml:22-27: widths 128, FFN on, ratio 1."""
    import hydra_lite as H
    from superpoint_transformer_amd import hotpath
    over = {"model": {"_down_dim": [128] * 4, "_up_dim": [128] * 3,
                      "net": {"no_ffn": False, "down_ffn_ratio": 1}}}
    c = H.compose("semantic/kitti360", "semantic/spt-2", over)
    pt, ed = c.get("datamodule.num_hf_point"), c.get("datamodule.num_hf_edge")
    _check(c, hotpath.spt128_config(pt, ed))


def test_nano2_is_the_s3dis_nano_experiment():
    """configs/experiment/semantic/s3dis_nano.yaml: no level-0 stage; ``point_mlp`` resolves in
    the YAML but SPT ignores it when ``nano`` (spt.py:486-521)."""
    import hydra_lite as H
    from superpoint_transformer_amd import hotpath
    c = H.compose("semantic/s3dis_nano", "semantic/nano-2")
    assert c.get("model.net.nano") is True and c.get("datamodule.num_hf_segment") == 8
    _check(c, hotpath.nano2_config(8, 18), ignore=("point_mlp",))


def test_panoptic_config_and_edge_affinity_head():
    """configs/experiment/panoptic/s3dis.yaml: model panoptic/spt-2 (point encoder [32, 64, 64],
    edge-affinity head MLP [128, 32, 16, 1] without norm, loss weight 1)."""
    import hydra_lite as H
    from superpoint_transformer_amd import hotpath
    c = H.compose("panoptic/s3dis", "panoptic/spt-2")
    _check(c, hotpath.panoptic_config(8, 18))
    head = c.get("model.edge_affinity_head")
    model = hotpath.SPTPanoptic(**hotpath.panoptic_config(8, 18))
    mine = [model.edge_affinity_head.mlp[0].in_features] + [
        m.out_features for m in model.edge_affinity_head.mlp if hasattr(m, "out_features")]
    assert head["dims"] == mine == [128, 32, 16, 1]
    assert head["norm"] is None and head["last_norm"] is False and head["last_activation"] is False
    assert c.get("model.edge_affinity_loss_lambda") == 1
    # the per-batch target construction of this configuration (tools/instance_bench.py)
    assert (c.get("datamodule.instance_k_max"), c.get("datamodule.instance_radius")) == (30, 0.1)


def test_preprocessing_leg_uses_the_datasets_knn_settings():
    """bench.PRE_CFG (voxel, k, r of the kNN + geometric-feature leg) against
    configs/datamodule/semantic/{s3dis,dales}.yaml; ``knn_step: -1`` = no optimal-neighbourhood
    search in the shipped configs (the k_step >= 0 route exists and is tested separately)."""
    import importlib.util
    import hydra_lite as H
    spec = importlib.util.spec_from_file_location(
        "bench_module", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for scene, dm in (("S", "semantic/s3dis"), ("D", "semantic/dales")):
        c = H.compose(dm, "semantic/spt-2")
        got = (c.get("datamodule.voxel"), c.get("datamodule.knn"), float(c.get("datamodule.knn_r")))
        assert got == bench.PRE_CFG[scene], (scene, got)
        assert c.get("datamodule.knn_step") == -1
