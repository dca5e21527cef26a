"""ctypes binding of libsug_amd.so (the C ABI declared in include/sug_amd.h).

There is no CPU or eager fallback: if the library is missing or a tensor is not on
a HIP device, the calling op raises.
"""
import ctypes
import os

# torch must be imported before libsug_amd.so is opened: both need libamdhip64.so.7 and the
# process must end up with ONE HIP runtime (torch's bundled copy), otherwise launches from this
# library see "no ROCm-capable device".
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libsug_amd.so')

_vp, _i32, _i64, _f32, _f64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double

# symbol -> argtypes, in the order of include/sug_amd.h (restype int unless noted)
SIGNATURES = {
    'sug_knn': [_vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp],
    'sug_knn_reverse': [_vp, _i32, _i32, _i32, _vp, _vp, _vp],
    'sug_fps': [_vp, _vp, _i32, _i32, _i32, _vp, _vp],
    'sug_ball_query': [_vp, _vp, _i32, _i32, _i32, _f32, _i32, _vp, _vp],
    'sug_knn_query': [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp],
    'sug_knn_query_direct': [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp],
    'sug_three_nn': [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp],
    'sug_gather_rows': [_vp, _i64, _vp, _i32, _i32, _i32, _i32, _vp, _i64, _vp],
    'sug_scatter_add_rows': [_vp, _i64, _vp, _i32, _i32, _i32, _i32, _vp, _i64, _vp],
    'sug_scatter_rows_ordered_supported': [_i32, _i32, _i32],
    'sug_scatter_rows_ordered': [_vp, _i64, _vp, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _vp],
    'sug_group_max': [_vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp],
    'sug_group_max_bwd': [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _i64, _vp],
    'sug_edgeconv_fwd': [_vp, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    'sug_bn_finalize': [_vp, _vp, _vp, _i32, _f64, _f32, _f32, _vp, _vp, _vp, _vp],
    'sug_affine_act': [_vp, _i64, _vp, _i64, _i32, _f32, _vp, _i64, _vp],
    'sug_col_stats': [_vp, _i64, _i64, _i32, _vp, _vp, _vp],
    'sug_edgeconv_bwd_reduce': [_vp, _i64, _vp, _vp, _i64, _i32, _f32, _vp, _vp, _vp, _vp],
    'sug_edgeconv_bwd_scatter': [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32,
                                 _vp, _i64, _vp],
    'sug_edgeconv_layer_fwd': [_vp, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _vp, _vp,
                               _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp],
    'sug_edgeconv_fused_layer_fwd': [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _f32,
                                     _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp],
    'sug_edgeconv_fused_supported': [_i32, _i32, _i32, _i32],
    'sug_edgeconv_layer_bwd': [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32,
                               _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp],
    'sug_bn_act_rows_fwd': [_vp, _i64, _i64, _i32, _i32, _vp, _vp, _i32, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _i64,
                            _vp, _vp, _vp],
    'sug_bn_act_rows_bwd': [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp],
    'sug_bn_act_pool_layer_fwd': [_vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _f32, _f32, _f32, _vp, _vp, _vp,
                                  _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp],
    'sug_bn_act_pool_layer_bwd': [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _vp,
                                  _i64, _vp, _vp],
    'sug_fold_groups': [_vp, _i32, _i32, _vp, _vp],
    'sug_edgeconv_fwd_bn': [_vp, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp,
                            _vp, _vp],
    'sug_col_stats_bn': [_vp, _i64, _i64, _i32, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _vp, _vp],
    'sug_col_stats_bn_grouped': [_vp, _i64, _i64, _i32, _i32, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _vp, _vp],
    'sug_bn_replay': [_vp, _i32, _i32, _f32, _vp, _vp, _vp],
    'sug_bn_replay_multi': [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    'sug_ptran_pos1_fwd': [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
    'sug_ptran_pos1_bwd': [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp],
    'sug_ptran_qk_fwd': [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
    'sug_ptran_qk_bwd': [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp],
    'sug_ptran_attn_fwd': [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _vp],
    'sug_ptran_fused_supported': [_i32, _i32, _i32, _i32],
    'sug_ptran_fused_fwd': [_vp] * 13 + [_i32, _i32, _i32, _i32, _f32, _i32] + [_vp] * 9,
    'sug_ptran_attn_bwd': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp,
                           _vp, _vp, _vp],
    'sug_ptran_relu_bwd_db': [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp],
    'sug_mmd_rbf_value': [_vp, _i64, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp],
    'sug_mmd_rbf_rows': [_vp, _i64, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp],
    'sug_mmd_rbf_rows_bwd': [_vp, _i64, _vp, _i32, _i32, _i32, _i32, _vp, _f32, _vp, _i64, _vp],
    'sug_mmd_rbf_bwd': [_vp, _i64, _vp, _i32, _i32, _vp, _vp, _i64, _vp],
    'sug_colsum': [_vp, _i64, _i64, _i32, _i32, _f32, _vp, _vp, _vp],
    'sug_gate_fwd': [_vp, _vp, _i64, _vp, _vp],
    'sug_gate_bwd': [_vp, _vp, _vp, _i64, _vp, _vp, _vp],
    'sug_mmd_assemble': [_vp, _i64, _vp, _i64, _vp, _vp, _i32, _i32, _i32, _f32, _vp, _vp],
    'sug_edge_weight_split': [_vp, _i32, _i32, _i32, _vp, _vp],
    'sug_edge_weight_split_multi': [_vp, _vp, _vp, _i32, _i32, _vp, _vp],
    'sug_soft_mmd_multi_fwd': [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp],
    'sug_soft_mmd_multi_bwd': [_i32, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp],
    'sug_sda_prob_weights': [_vp, _i64, _vp, _i64, _vp, _vp, _i32, _i32, _f32, _i32, _vp, _vp],
    'sug_sda_prob_weights_multi': [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _i32, _vp, _vp],
    'sug_adam_step': [_vp, _vp, _vp, _i32, _vp, _f64, _f64, _f64, _f64, _f64, _f64, _f64, _vp],
    'sug_adam_step_capturable': [_vp, _vp, _vp, _i32, _vp, _f64, _f64, _f64, _f64, _f64, _vp, _vp, _vp, _vp],
    'sug_adam_chain_step': [_vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp],
    'sug_linear_dw': [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp],
    'sug_linear_dw_fold': [_vp, _i32, _i64, _vp, _vp],
    'sug_sa_first_fwd': [_vp, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp],
    'sug_sa_first_bwd': [_vp, _vp, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                         _vp, _vp, _vp],
    'sug_pointmlp_max_bwd_coef': [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp],
    'sug_pointmlp_max_bwd_dwfix': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp],
    'sug_head_linear_supported': [_i32, _i32, _i32, _i32, _i32],
    'sug_head_linear_fwd': [_i32, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _vp],
    'sug_head_ln_bwd': [_i32, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _f32, _vp],
    'sug_head_linear_bwd': [_i32, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _f32, _f32, _vp],
    'sug_gate_bn_fwd': [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _f32, _f32, _vp, _vp, _vp],
    'sug_gate_bn_bwd': [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp],
    'sug_copy_rows2d': [_vp, _i64, _vp, _i64, _i64, _i32, _vp],
    'sug_ln_act_fwd': [_vp, _vp, _vp, _i32, _i32, _f32, _f32, _vp, _vp, _vp],
    'sug_ln_act_bwd': [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp],
    'sug_interp3_cat_bwd_lists': [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    'sug_linear_dw_bias': [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _vp],
    'sug_node_offset_fwd': [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp],
    'sug_node_offset_bwd': [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp],
    'sug_interp3_cat_fwd': [_vp, _i64, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _i64, _vp],
    'sug_interp3_cat_bwd': [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp],
    'sug_bn_bwd_apply': [_vp, _vp, _i64, _vp, _vp, _i64, _i32, _vp, _i64, _vp],
    'sug_bn_act_pool_fwd': [_vp, _i64, _vp, _i32, _i32, _i32, _f32, _vp, _vp, _i64, _vp, _vp, _vp],
    'sug_bn_act_pool_bwd': [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _vp, _i64, _vp],
    'sug_rows_gemm': [_vp, _i64, _i64, _i32, _vp, _vp, _i32, _vp, _i64, _vp],
    'sug_pointmlp_max_fwd': [_vp, _i64, _i64, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp],
    'sug_pointmlp_max_layer_fwd': [_vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _vp,
                                   _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp],
    'sug_pointmlp_max_layer_fwd_xf': [_vp, _i64, _i64, _i32, _vp, _f32, _vp, _i64, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32,
                                      _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp],
    'sug_pointmlp_max_bwd_sparse': [_vp, _vp, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _vp],
    'sug_mmd_rbf': [_vp, _i64, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp],
    'sug_chamfer': [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp],
    'sug_chamfer_weights': [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp],
    'sug_sa_first_geo_fwd': [_vp, _i64, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _f32, _f32,
                             _vp, _vp, _vp, _vp, _vp, _vp],
    'sug_sa_first_geo_bwd': [_vp, _vp, _i64, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp,
                             _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    'sug_calayer_supported': [_i32, _i32, _i32, _i32],
    'sug_calayer_fwd': [_i32, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    'sug_calayer_bwd': [_i32, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    'sug_ce_pair_fwd': [_vp, _vp, _i64, _vp, _i32, _i32, _f32, _i64, _vp, _vp, _vp],
    'sug_ce_pair_bwd': [_vp, _vp, _i64, _vp, _i32, _i32, _i32, _f32, _i64, _vp, _vp, _vp, _vp, _vp],
    'sug_loss_combine_fwd': [_vp, _vp, _vp, _vp, _f32, _f32, _vp, _vp],
    'sug_loss_combine_bwd': [_vp, _f32, _f32, _vp, _vp],
}

STATS_BLOCKS = 1024        # SUG_STATS_BLOCKS

_lib = None


def lib():
    """Load (once) and return the library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'sug_amd: %s not found -- build it with `make -C sug_amd/csrc` or '
                '`python -c "import __graft_entry__ as g; g.build()"`; there is no fallback path'
                % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = ctypes.c_int
        L.sug_linear_dw_workspace.restype = ctypes.c_int64
        L.sug_linear_dw_workspace.argtypes = [_i64, _i32, _i32]
        L.sug_pointmlp_max_bwd_workspace.restype = ctypes.c_int64
        L.sug_pointmlp_max_bwd_workspace.argtypes = [_i64, _i32, _i32, _i32]
        L.sug_colsum_workspace.restype = ctypes.c_int64
        L.sug_colsum_workspace.argtypes = [_i64, _i32]
        L.sug_ptran_colsum_workspace.restype = ctypes.c_int64
        L.sug_ptran_colsum_workspace.argtypes = [_i64]
        L.sug_chamfer_workspace.restype = ctypes.c_int64
        L.sug_scatter_rows_workspace.restype = ctypes.c_int64
        L.sug_scatter_rows_workspace.argtypes = [_i32, _i32, _i32]
        L.sug_chamfer_workspace.argtypes = [_i32, _i32, _i32]
        L.sug_adam_chunk.restype = ctypes.c_int
        L.sug_adam_chunk.argtypes = []
        L.sug_adam_chain_chunk.restype = ctypes.c_int
        L.sug_adam_chain_chunk.argtypes = []
        L.sug_last_error.restype = ctypes.c_char_p
        L.sug_last_error.argtypes = []
        L.sug_abi_version.restype = ctypes.c_int
        L.sug_abi_version.argtypes = []
        if L.sug_abi_version() != 7:
            raise RuntimeError('sug_amd: ABI version mismatch, rebuild libsug_amd.so')
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError('%s failed (%d): %s' % (what, rc, lib().sug_last_error().decode()))
