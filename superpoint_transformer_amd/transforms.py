"""Per-batch on-device graph transforms of the hot path's "next" rows
(SURVEY.md 8f): on-the-fly horizontal edge features fused with the edge
symmetrisation and the self loops."""
import torch

from . import _lib

__all__ = ["horizontal_edge_features", "EDGE_FEATURE_COLUMNS"]

EDGE_FEATURE_COLUMNS = [
    "mean_off_x", "mean_off_y", "mean_off_z", "std_off_x", "std_off_y", "std_off_z",
    "mean_dist", "angle_source", "angle_target", "normal_angle", "log_length",
    "log_surface", "log_volume", "log_size", "centroid_dir_x", "centroid_dir_y",
    "centroid_dir_z", "centroid_dist"]


def horizontal_edge_features(edge_index, edge_attr, pos, normal, log_length, log_surface,
                             log_volume, log_size, add_self_loops=True):
    """``OnTheFlyHorizontalEdgeFeatures`` (all default keys) + ``NAGAddSelfLoops``
    of one level (src/transforms/graph.py:1135-1277, 1419-1452) in one kernel.

    ``edge_index`` [2,E] trimmed edges, ``edge_attr`` [E,7] (f16 on disk: cast
    here like the reference's ``.float()``), node attributes of the level.
    Returns (edge_index [2, 2E(+N)], edge_attr [2E(+N), 18])."""
    _lib.require_cuda(edge_index, edge_attr, pos)
    dev = pos.device
    se = edge_index.long().contiguous()
    e, n = se.shape[1], pos.shape[0]

    def f32(t, cols=None):
        t = t.detach().float().contiguous()
        return t.view(n, cols) if cols else t.view(-1)

    ea7 = edge_attr.detach().float().contiguous()
    if ea7.shape != (e, 7):
        raise ValueError(f"edge_attr must be [E,7], got {tuple(ea7.shape)}")
    etot = 2 * e + (n if add_self_loops else 0)
    ei_out = torch.empty((2, etot), dtype=torch.int64, device=dev)
    ea_out = torch.empty((etot, 18), dtype=torch.float32, device=dev)
    args = [f32(pos, 3), f32(normal, 3), f32(log_length), f32(log_surface), f32(log_volume),
            f32(log_size)]
    with torch.cuda.device(dev):
        st = _lib.lib.spt_horizontal_edge_features_f32(
            _lib.ptr(se), e, n, _lib.ptr(ea7), *[_lib.ptr(a) for a in args],
            int(add_self_loops), _lib.ptr(ei_out), _lib.ptr(ea_out), _lib.stream_ptr(dev))
    _lib.check(st, "spt_horizontal_edge_features_f32")
    return ei_out, ea_out
