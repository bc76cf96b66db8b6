"""ctypes binding of libspt_hip.so (C ABI declared in include/spt_hip.h).

The library is the product: there is NO CPU or PyTorch fallback.  Importing
this module without the built shared object raises, and every wrapper raises
``RuntimeError`` with ``spt_last_error()`` on a non-zero status.
"""
import ctypes
import os

import torch  # noqa: F401  (loads torch's libamdhip64.so.7 first so ours binds to it)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libspt_hip.so")

_c = ctypes
_i64 = _c.c_int64
_int = _c.c_int
_p = _c.c_void_p
_sz = _c.c_size_t
_f32 = _c.c_float

# name -> (restype, argtypes).  Must list EVERY symbol of include/spt_hip.h:
# tests/test_abi.py parses the header and checks both directions.
SIGNATURES = {
    "spt_version": (_int, []),
    "spt_last_error": (_c.c_char_p, []),
    "spt_csr_build_workspace_bytes": (_sz, [_i64, _i64]),
    "spt_csr_build": (_int, [_p, _i64, _i64, _p, _p, _p, _sz, _p]),
    "spt_csr_pos_seg": (_int, [_p, _i64, _i64, _p, _p]),
    "spt_csr_gather_i64_i32": (_int, [_p, _p, _i64, _p, _p]),
    "spt_segcsr_reduce_f32": (_int, [_int, _p, _p, _p, _i64, _i64, _int, _p, _p, _p]),
    "spt_segcsr_max_affine_f32": (_int, [_p, _p, _p, _i64, _i64, _int, _p, _p, _p, _f32, _p, _p, _p, _p]),
    "spt_segcsr_max_affine_bf16_supported": (_int, [_int, _i64]),
    "spt_segcsr_max_affine_bf16": (_int, [_p, _p, _p, _i64, _i64, _int, _p, _p, _p, _f32, _p, _p, _p, _p]),
    "spt_segcsr_max_affine_raw_supported": (_int, [_int, _i64]),
    "spt_segcsr_max_affine_raw_f32": (_int, [_p, _int, _p, _p, _i64, _i64, _int, _p, _p, _p, _f32, _p, _p, _p,
                                             _p, _p]),
    "spt_graphnorm_bwd_stats_sparse_raw_f32": (_int, [_p, _p, _p, _p, _p, _i64, _i64, _int, _int, _p, _p, _p,
                                                      _f32, _p, _p, _sz, _p]),
    "spt_fused_linear_storage_supported": (_int, [_int, _int]),
    "spt_graphnorm_bwd_stats_sparse_ex_f32": (_int, [_p, _int, _p, _p, _p, _p, _i64, _i64, _int, _int, _p, _p,
                                                     _p, _f32, _p, _p, _sz, _p]),
    "spt_segcsr_reduce_bwd_f32": (_int, [_int, _p, _p, _p, _p, _p, _i64, _i64, _int, _p, _p]),
    "spt_segcsr_sum_i64": (_int, [_p, _p, _p, _i64, _i64, _int, _p, _p]),
    "spt_gather_rows_f32": (_int, [_p, _p, _i64, _i64, _int, _p, _p]),
    "spt_graphnorm_workspace_bytes": (_sz, [_i64, _int, _int]),
    "spt_graphnorm_fwd_f32": (_int, [_p, _p, _i64, _int, _int, _p, _p, _p, _f32, _f32,
                                     _p, _p, _p, _p, _sz, _p]),
    "spt_graphnorm_bwd_f32": (_int, [_p, _p, _p, _i64, _int, _int, _p, _p, _p, _p, _p,
                                     _f32, _p, _p, _p, _p, _p, _sz, _p]),
    "spt_graphnorm_stats_f32": (_int, [_p, _p, _i64, _int, _int, _p, _p, _f32, _p, _p, _p, _p,
                                       _p, _sz, _p]),
    "spt_graphnorm_bwd_acc_f32": (_int, [_p, _p, _p, _i64, _int, _int, _p, _p, _p, _p, _p,
                                         _f32, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "spt_edge_attn_fwd_f32": (_int, [_p, _i64, _int, _int, _int, _p, _p, _p, _i64, _p, _int,
                                     _p, _p, _p, _p, _p, _p, _int, _f32, _p, _p, _p, _p]),
    "spt_edge_attn_bwd_workspace_bytes": (_sz, [_int, _int, _int, _int]),
    "spt_attn_use_mfma": (_int, [_int]),
    "spt_attn_bwd_packed": (_int, [_int]),
    "spt_attn_bwd_el_full_line": (_int, [_int]),
    "spt_attn_bwd_el_target_order": (_int, [_int]),
    "spt_attn_tile_record_ints": (_int, []),
    "spt_attn_pack_tile_ids_ex": (_int, [_p, _p, _p, _p, _i64, _p, _p]),
    "spt_attn_tile_record_ints_m": (_int, [_int]),
    "spt_attn_pack_tile_ids_m": (_int, [_p, _p, _p, _p, _i64, _int, _p, _p]),
    "spt_attn_mirror_prepare": (_int, [_p, _p, _i64, _i64, _p, _p, _p]),
    "spt_attn_pack_tile_ids_mirror": (_int, [_p, _p, _p, _p, _i64, _i64, _p, _p]),
    "spt_attn_split_pack_f32": (_int, [_p, _i64, _int, _int, _p, _p]),
    "spt_attn_split_grad_f32": (_int, [_p, _i64, _int, _int, _p, _p]),
    "spt_edge_attn_bwd_f32": (_int, [_p, _i64, _int, _int, _int, _p, _p, _p, _i64, _p, _int,
                                     _p, _p, _p, _p, _p, _p, _int, _f32, _p, _p, _p, _p,
                                     _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "spt_edge_attn_bwd_acc_f32": (_int, [_p, _i64, _int, _int, _int, _p, _p, _p, _i64, _p, _int,
                                         _p, _p, _p, _p, _p, _p, _int, _f32, _p, _p, _p, _p,
                                         _p, _p, _int, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "spt_edge_attn_fwd_ex_f32": (_int, [_p, _i64, _int, _int, _int, _p, _p, _p, _i64, _p, _int,
                                        _p, _p, _p, _p, _p, _p, _int, _f32, _p, _p, _p, _int, _p]),
    "spt_edge_attn_bwd_ex_workspace_bytes": (_sz, [_i64, _i64, _int, _int, _int, _int]),
    "spt_edge_attn_bwd_el_supported": (_int, [_int, _int, _int, _int, _int]),
    "spt_attn_pack_tile_ids": (_int, [_p, _p, _p, _i64, _p, _p]),
    "spt_edge_attn_bwd_ex_f32": (_int, [_p, _i64, _int, _int, _int, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _int,
                                        _p, _p, _p, _p, _p, _p, _int, _f32, _p, _p, _p, _p,
                                        _p, _p, _int, _p, _p, _p, _p, _p, _p, _int, _p, _sz, _p]),
    "spt_grid_knn_workspace_bytes": (_sz, [_i64, _i64]),
    "spt_knn_use_cell_path": (_int, [_int]),
    "spt_segcsr_use_stream": (_int, [_int]),
    "spt_segcsr_reduce_ex_f32": (_int, [_int, _p, _p, _p, _i64, _i64, _int, _p, _p, _int, _p]),
    "spt_segcsr_max_affine_ex_f32": (_int, [_p, _p, _p, _i64, _i64, _int, _p, _p, _p, _f32, _p, _p, _p,
                                            _int, _p]),
    "spt_grid_knn_ex_f32": (_int, [_p, _i64, _p, _i64, _int, _f32, _f32, _p, _p, _int, _int, _int,
                                   _p, _p, _p, _int, _p, _sz, _p]),
    "spt_grid_knn_geof_f32": (_int, [_p, _i64, _int, _f32, _f32, _p, _p, _int, _int, _int, _int,
                                     _p, _p, _p, _p, _int, _p, _sz, _p]),
    "spt_grid_knn_after_f32": (_int, [_p, _i64, _p, _i64, _int, _f32, _f32, _p, _p, _int, _int, _p, _p,
                                      _p, _p, _p, _sz, _p]),
    "spt_grid_knn_f32": (_int, [_p, _i64, _p, _i64, _int, _f32, _f32, _p, _p, _int, _int, _int,
                                _p, _p, _p, _p, _sz, _p]),
    "spt_point_geof_dense_f32": (_int, [_p, _i64, _p, _int, _int, _int, _int, _p, _p, _p]),
    "spt_point_geof_dense_ld_f32": (_int, [_p, _i64, _p, _int, _i64, _int, _int, _int, _p, _p, _p]),
    "spt_bbox_f32": (_int, [_p, _i64, _p, _p]),
    "spt_knn_subsample_f32": (_int, [_p, _i64, _p, _f32, _int, _p, _p, _p]),
    "spt_grid_cell_ids_f32": (_int, [_p, _i64, _f32, _p, _p, _p, _p]),
    "spt_grid_count_cells_workspace_bytes": (_sz, [_i64]),
    "spt_grid_count_cells_f32": (_int, [_p, _i64, _f32, _p, _p, _p, _p, _sz, _p]),
    "spt_spatial_order_workspace_bytes": (_sz, [_i64, _i64]),
    "spt_spatial_order": (_int, [_p, _i64, _f32, _p, _p, _p, _p, _sz, _p]),
    "spt_point_geof_csr_f32": (_int, [_p, _i64, _p, _p, _int, _int, _int, _p, _p]),
    "spt_vertical_edge_features_f32": (_int, [_p, _i64] + [_p] * 13 + [_p]),
    "spt_edge_affinity_features_f32": (_int, [_p, _i64, _int, _p, _p, _i64, _p, _p]),
    "spt_edge_affinity_features_bwd_f32": (_int, [_p, _p, _i64, _int, _p, _p, _i64, _p, _p]),
    "spt_scatter_pca_f32": (_int, [_p, _p, _p, _i64, _p, _p, _p]),
    "spt_neighbors_dense_to_csr_workspace_bytes": (_sz, [_i64]),
    "spt_neighbors_dense_to_csr": (_int, [_p, _i64, _int, _p, _p, _p, _p, _sz, _p]),
    "spt_horizontal_edge_features_f32": (_int, [_p, _i64, _i64, _p, _p, _p, _p, _p, _p, _p, _int,
                                                _p, _p, _p]),
    "spt_graphnorm_tables_f32": (_int, [_p, _int, _int, _p, _p, _f32, _p, _p, _p, _p, _p]),
    "spt_graphnorm_apply_f32": (_int, [_p, _p, _i64, _int, _int, _p, _p, _p, _f32, _p, _p]),
    "spt_graphnorm_bwd_stats_f32": (_int, [_p, _p, _p, _i64, _int, _int, _p, _p, _p, _f32, _p,
                                           _p, _sz, _p]),
    "spt_graphnorm_bwd_stats_sparse_workspace_bytes": (_sz, [_i64, _int, _int]),
    "spt_graphnorm_bwd_stats_sparse_f32": (_int, [_p, _p, _p, _p, _p, _i64, _i64, _int, _int, _p, _p,
                                                   _p, _f32, _p, _p, _sz, _p]),
    "spt_graphnorm_bwd_tables_f32": (_int, [_p, _int, _int, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                            _p, _p]),
    "spt_fused_linear_supported": (_int, [_int, _int]),
    "spt_fused_linear_use_split_bf16": (_int, [_int]),
    "spt_fused_linear_bwd_use_dma": (_int, [_int]),
    "spt_fused_linear_fwd_use_x3": (_int, [_int]),
    "spt_fused_linear_pooled_supported": (_int, [_int, _int]),
    "spt_fused_linear_bwd_pooled_f32": (_int, [_p, _p, _p, _p, _p, _i64, _i64, _int, _p, _p, _p, _f32,
                                               _p, _p, _p, _p, _int, _p, _p, _p, _f32, _p, _p, _p,
                                               _int, _p, _p, _sz, _p]),
    "spt_fused_linear_pooled_supported_ex": (_int, [_int, _int, _int]),
    "spt_fused_linear_fwd_ex_f32": (_int, [_p, _i64, _i64, _int, _p, _int, _p, _p, _p, _f32, _p, _p,
                                           _int, _p, _sz, _p]),
    "spt_fused_linear_bwd_ex_f32": (_int, [_p, _p, _i64, _i64, _int, _p, _p, _p, _f32, _p, _p, _p,
                                           _p, _int, _p, _p, _p, _f32, _p, _p, _p, _int, _p, _int, _p,
                                           _sz, _p]),
    "spt_fused_linear_bwd_pooled_ex_f32": (_int, [_p, _p, _p, _p, _p, _i64, _i64, _int, _p, _p, _p, _f32,
                                                  _p, _p, _p, _p, _int, _p, _p, _p, _f32, _p, _p, _p,
                                                  _int, _p, _int, _p, _sz, _p]),
    "spt_fused_linear_workspace_bytes": (_sz, [_int, _int]),
    "spt_fused_linear_fwd_runs_f32": (_int, [_p, _int, _p, _p, _p, _int, _int, _p, _int, _p, _p, _p,
                                             _f32, _p, _p, _int, _p, _sz, _p]),
    "spt_fused_linear_bwd_runs_f32": (_int, [_p, _p, _int, _p, _p, _p, _int, _int, _p, _p, _p, _f32,
                                             _p, _p, _p, _p, _int, _p, _p, _p, _f32, _p, _p, _p, _p,
                                             _int, _p, _sz, _p]),
    "spt_fused_linear_bwd_pooled_runs_f32": (_int, [_p, _p, _p, _p, _p, _int, _p, _p, _p, _int, _int,
                                                    _p, _p, _p, _f32, _p, _p, _p, _p, _int, _p, _p,
                                                    _p, _f32, _p, _p, _p, _p, _int, _p, _sz, _p]),
    "spt_fused_linear_fwd_runs_gn_f32": (_int, [_p, _int, _p, _p, _p, _int, _int, _p, _int, _p, _p, _p,
                                                _f32, _p, _p, _int, _p, _sz, _p, _p]),
    "spt_fused_linear_bwd_runs_gn_f32": (_int, [_p, _p, _int, _p, _p, _p, _int, _int, _p, _p, _p, _f32,
                                                _p, _p, _p, _p, _int, _p, _p, _p, _f32, _p, _p, _p, _p,
                                                _int, _p, _sz, _p, _p]),
    "spt_fused_linear_bwd_pooled_runs_gn_f32": (_int, [_p, _p, _p, _p, _p, _int, _p, _p, _p, _int, _int,
                                                       _p, _p, _p, _f32, _p, _p, _p, _p, _int, _p, _p,
                                                       _p, _f32, _p, _p, _p, _p, _int, _p, _sz, _p, _p]),
    "spt_fused_linear_bwd_pool_runs_gn_f32": (_int, [_p, _p, _p, _p, _p, _p, _i64, _int, _p, _p, _p, _int,
                                                     _int, _p, _p, _p, _f32, _p, _p, _p, _p, _int, _p, _p,
                                                     _p, _f32, _p, _p, _p, _p, _p, _p, _int, _p, _sz, _p, _p]),
    "spt_csr_check_i64": (_int, [_p, _p, _p, _i64, _i64, _int, _p, _p]),
    "spt_csr_adopt_i64": (_int, [_p, _p, _p, _i64, _i64, _p, _p, _p, _p]),
    "spt_fused_linear_pool_supported": (_int, [_int, _int, _int]),
    "spt_fused_linear_pool_gram_len": (_sz, [_int]),
    "spt_fused_linear_pool_workspace_bytes": (_sz, [_int, _int]),
    "spt_fused_linear_fwd_pool_runs_f32": (_int, [_p, _p, _p, _p, _p, _i64, _i64, _int, _p, _p, _p, _int,
                                                  _int, _p, _int, _p, _p, _p, _f32, _f32, _p, _p, _p,
                                                  _f32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _int, _p,
                                                  _sz, _p]),
    "spt_fused_linear_bwd_pool_runs_f32": (_int, [_p, _p, _p, _p, _p, _p, _i64, _int, _p, _p, _p, _int,
                                                  _int, _p, _p, _p, _f32, _p, _p, _p, _p, _int, _p, _p,
                                                  _p, _f32, _p, _p, _p, _p, _p, _p, _int, _p, _sz, _p]),
    "spt_fused_linear_fwd_f32": (_int, [_p, _i64, _i64, _int, _p, _int, _p, _p, _p, _f32, _p, _p,
                                        _p, _sz, _p]),
    "spt_fused_linear_bwd_f32": (_int, [_p, _p, _i64, _i64, _int, _p, _p, _p, _f32, _p, _p, _p,
                                        _p, _int, _p, _p, _p, _f32, _p, _p, _p, _int, _p, _p,
                                        _sz, _p]),
    "spt_unit_sphere_norm_f32": (_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _p, _p, _p, _p]),
    "spt_unit_sphere_assemble_f32": (_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _p, _int, _p, _p, _p, _p]),
    "spt_unit_sphere_workspace_bytes": (_sz, [_i64, _i64]),
    "spt_unit_sphere_norm_ws_f32": (_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _p, _p, _p, _p, _sz, _p]),
    "spt_unit_sphere_assemble_ws_f32": (_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _p, _int, _p, _p, _p,
                                               _p, _sz, _p]),
    "spt_sparse_sample_workspace_bytes": (_sz, [_i64, _i64]),
    "spt_sparse_sample": (_int, [_p, _i64, _i64, _p, _int, _int, _c.c_uint64, _p, _p, _p, _sz, _p]),
    "spt_segment_std_f32": (_int, [_p, _p, _p, _i64, _int, _p, _p]),
    "spt_segment_mean_orientation_f32": (_int, [_p, _p, _p, _i64, _p, _p]),
    "spt_cluster_graph_edges_workspace_bytes": (_sz, [_i64, _int]),
    "spt_cluster_graph_edges": (_int, [_p, _p, _p, _i64, _int, _f32, _int, _p, _p, _p, _p, _sz, _p]),
    "spt_skinny_linear_supported": (_int, [_int, _int]),
    "spt_narrow_linear_bwd_supported": (_int, [_int, _int]),
    "spt_narrow_linear_bwd_workspace_bytes": (_sz, [_int, _int]),
    "spt_narrow_linear_bwd_f32": (_int, [_p, _p, _p, _i64, _int, _int, _p, _p, _p, _p, _sz, _p]),
    "spt_cross_entropy_workspace_bytes": (_sz, [_i64]),
    "spt_cross_entropy_fwd_f32": (_int, [_p, _p, _i64, _int, _i64, _p, _p, _p, _p, _sz, _p]),
    "spt_cross_entropy_bwd_f32": (_int, [_p, _p, _p, _i64, _int, _i64, _p, _p, _p, _p]),
    "spt_skinny_dw_supported": (_int, [_int, _int]),
    "spt_skinny_dw_workspace_bytes": (_sz, [_int, _int]),
    "spt_skinny_dw_f32": (_int, [_p, _p, _i64, _int, _int, _p, _p, _p, _sz, _p]),
    "spt_skinny_linear_f32": (_int, [_p, _i64, _int, _p, _p, _int, _p, _p]),
    "spt_skinny_linear_wt_f32": (_int, [_p, _i64, _int, _p, _int, _p, _p]),
    "spt_skinny_pre_supported": (_int, [_int, _int, _int]),
    "spt_skinny_linear_pre_f32": (_int, [_p, _i64, _int, _p, _p, _int, _p, _p, _p, _p, _p, _int, _p, _p]),
    "spt_skinny_dw_pre_f32": (_int, [_p, _p, _i64, _int, _int, _p, _p, _p, _p, _p, _p, _int, _p, _sz, _p]),
    "spt_skinny_use_split_bf16": (_int, [_int]),
    "spt_skinny_linear_pre_m_f32": (_int, [_p, _i64, _int, _p, _p, _int, _p, _p, _p, _p, _p, _int, _p, _int, _p]),
    "spt_skinny_linear_wt_m_f32": (_int, [_p, _i64, _int, _p, _int, _p, _int, _p]),
    "spt_skinny_dw_pre_m_f32": (_int, [_p, _p, _i64, _int, _int, _p, _p, _p, _p, _p, _p, _int, _int, _p, _sz, _p]),
    "spt_index_inverse": (_int, [_p, _i64, _i64, _p, _p]),
    "spt_select_edges_workspace_bytes": (_sz, [_i64]),
    "spt_select_edges": (_int, [_p, _i64, _i64, _p, _i64, _p, _i64, _p, _p, _p, _sz, _p]),
    "spt_cluster_select_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "spt_cluster_select": (_int, [_p, _p, _i64, _p, _i64, _i64, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "spt_relabel_consecutive_workspace_bytes": (_sz, [_i64]),
    "spt_relabel_consecutive": (_int, [_p, _p, _i64, _i64, _p, _p, _p, _p, _sz, _p]),
    "spt_radius_ball_workspace_bytes": (_sz, [_i64]),
    "spt_radius_ball_f32": (_int, [_p, _i64, _p, _f32, _int, _p, _i64, _p, _p, _p, _sz, _p]),
    "spt_cluster_pair_anchors_f32": (_int, [_p, _p, _p, _p, _p, _i64, _i64, _int, _int, _p, _p, _p]),
}


class GnFwdTables(ctypes.Structure):
    """``spt_gn_fwd_tables`` of include/spt_hip.h (host struct of device pointers)."""
    _fields_ = [("weight", _p), ("mean_scale", _p), ("eps", _f32), ("mean", _p), ("rstd", _p),
                ("am", _p), ("scale", _p)]


class GnBwdTables(ctypes.Structure):
    """``spt_gn_bwd_tables`` of include/spt_hip.h."""
    _fields_ = [("weight", _p), ("mean_scale", _p), ("mean", _p), ("rstd", _p), ("c1", _p), ("c2", _p),
                ("c3", _p), ("gweight", _p), ("gbias", _p), ("gmean_scale", _p)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (hipcc --offload-arch=gfx950). There is no fallback path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def last_error():
    return lib.spt_last_error().decode("utf-8", "replace")


def check(status, what):
    if status != 0:
        raise RuntimeError(f"{what} failed ({status}): {last_error()}")


def ptr(t):
    """Device pointer of a tensor (or None)."""
    return None if t is None else t.data_ptr()


def stream_ptr(device=None):
    """hipStream_t of torch's current stream, as an integer."""
    return torch.cuda.current_stream(device).cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "superpoint_transformer_amd ops run on the MI355X only: got a "
                f"{t.device} tensor (no CPU fallback is provided)")
