"""ctypes binding of libdgn_hip.so (C ABI in include/dgn_hip.h).

There is NO fallback: if the shared library is missing or cannot be loaded, every
entry point raises -- the product path never computes on the CPU or through torch ops.
Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``dgn_amd/csrc/build.sh`` (hipcc, --offload-arch=gfx950).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

DGN_MAX_AGG = 16
DGN_MAX_CH = 4
DGN_MAX_SCALERS = 4
ABI_VERSION = 28

LIB_PATH = os.environ.get("DGN_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdgn_hip.so")

# symbols include/dgn_hip.h declares (checked by tests/test_abi.py without a GPU)
EXPORTS = ("dgn_abi_version", "dgn_sizeof", "dgn_last_error", "dgn_set_option", "dgn_get_option", "dgn_edge_weights_workspace_bytes", "dgn_edge_weights",
           "dgn_agg_workspace_bytes", "dgn_agg_forward", "dgn_agg_aux_bytes", "dgn_agg_forward_aux", "dgn_agg_backward_aux", "dgn_towers_layer_agg_aux_bytes", "dgn_dense_layer_agg_aux_bytes", "dgn_agg_backward_workspace_bytes", "dgn_agg_edge_table_workspace_bytes", "dgn_agg_backward", "dgn_linear_forward_bn", "dgn_linear_wgrad_bn", "dgn_linear_forward_act", "dgn_linear_act_supported", "dgn_linear_forward_add", "dgn_linear_add_supported", "dgn_linear_forward_bn_act", "dgn_linear_act_mask_bytes", "dgn_linear_forward_bn_act_mask", "dgn_linear_forward_act_mask", "dgn_linear_bnb_supported", "dgn_linear_wgrad_bn_act_mask", "dgn_linear_forward_act_mask_bnb", "dgn_linear_combine_backward_weight_bias", "dgn_towers_layer_zmask_supported",
           "dgn_scale_combine_forward", "dgn_scale_combine_backward_workspace_bytes", "dgn_scale_combine_backward",
           "dgn_bn_tail_workspace_bytes", "dgn_bn_tail_forward", "dgn_bn_tail_backward",
           "dgn_bias_act_forward", "dgn_bias_act_backward", "dgn_dropout_mask_bytes", "dgn_dropout_forward", "dgn_dropout_backward",
           "dgn_layer_fused_supported", "dgn_layer_fused_forward",
           "dgn_gemm_supported", "dgn_gemm_forward", "dgn_gemm_wgrad_workspace_bytes", "dgn_gemm_wgrad",
           "dgn_agg_f_valid_supported", "dgn_graph_build_workspace_bytes", "dgn_graph_build", "dgn_graph_build_csc", "dgn_graph_build_cuts",
           "dgn_assemble_params", "dgn_towers_layer_supported", "dgn_towers_layer_forward_workspace_bytes", "dgn_towers_layer_forward",
           "dgn_towers_layer_backward_workspace_bytes", "dgn_towers_layer_backward",
           "dgn_linear_supported", "dgn_linear_forward", "dgn_linear_combine_forward", "dgn_linear_combine_backward_input", "dgn_linear_combine_backward_weight", "dgn_linear_wgrad_workspace_bytes", "dgn_linear_wgrad",
           "dgn_dense_layer_supported", "dgn_dense_layer_forward_workspace_bytes", "dgn_dense_layer_forward", "dgn_dense_layer_backward_workspace_bytes",
           "dgn_dense_layer_backward",
           "dgn_linear_bd_supported", "dgn_linear_bd_forward", "dgn_linear_bd_backward_input", "dgn_linear_bd_wgrad_workspace_bytes", "dgn_linear_bd_wgrad",
           "dgn_dc_supported", "dgn_dc_wgrad_supported", "dgn_dc_fold", "dgn_dc_gemm", "dgn_dc_wgrad_workspace_bytes", "dgn_dc_wgrad",
           "dgn_block_layer_supported", "dgn_block_layer_param_grad_floats", "dgn_block_layer_forward_workspace_bytes", "dgn_block_layer_forward",
           "dgn_block_layer_backward_workspace_bytes", "dgn_block_layer_backward")

DGN_DC_CLASSES, DGN_DC_UNIT = 32, 64


class DgnGraph(C.Structure):
    _fields_ = [("n_nodes", C.c_int64), ("n_edges", C.c_int64), ("indptr", C.c_void_p), ("src", C.c_void_p),
                ("n_hub", C.c_int64), ("hub_rows", C.c_void_p), ("hub_chunk_ptr", C.c_void_p),
                ("n_chunks", C.c_int64), ("chunk_hub", C.c_void_p), ("hub_threshold", C.c_int32),
                ("hub_chunk", C.c_int32), ("csc_ptr", C.c_void_p), ("csc_pos", C.c_void_p), ("max_in_degree", C.c_int32),
                ("n_src", C.c_int64), ("row_base", C.c_int64), ("blk_cut", C.c_void_p), ("blk_gap", C.c_int32),
                ("gblk_desc", C.c_void_p), ("n_gblk", C.c_int64), ("gblk_rows", C.c_int32), ("csc_order", C.c_void_p), ("dst_csr", C.c_void_p)]


class DgnChannel(C.Structure):
    _fields_ = [("kind", C.c_int32), ("eig_col", C.c_int32), ("alpha", C.c_float), ("eps", C.c_float)]


class DgnAggSpec(C.Structure):
    _fields_ = [("n_agg", C.c_int32), ("agg_op", C.c_int32 * DGN_MAX_AGG), ("agg_ch", C.c_int32 * DGN_MAX_AGG),
                ("n_ch", C.c_int32), ("n_scalers", C.c_int32), ("scaler", C.c_int32 * DGN_MAX_SCALERS),
                ("avg_log", C.c_float), ("eps", C.c_float), ("n_towers", C.c_int32), ("agg_total", C.c_int32),
                ("agg_offset", C.c_int32), ("tower_stride", C.c_int64)]


class DgnMsg(C.Structure):
    _fields_ = [("F", C.c_int64), ("x_src", C.c_void_p), ("ld_src", C.c_int64), ("x_dst", C.c_void_p),
                ("ld_dst", C.c_int64), ("m_edge", C.c_void_p), ("ld_edge", C.c_int64), ("x_in", C.c_void_p),
                ("ld_in", C.c_int64), ("edge_type", C.c_void_p), ("n_edge_types", C.c_int32), ("f_valid", C.c_int32)]


class DgnMsgGrad(C.Structure):
    _fields_ = [("g_src", C.c_void_p), ("ld_src", C.c_int64), ("g_dst", C.c_void_p), ("ld_dst", C.c_int64),
                ("g_edge", C.c_void_p), ("ld_edge", C.c_int64), ("g_in", C.c_void_p), ("ld_in", C.c_int64),
                ("accumulate", C.c_int32)]


class DgnBnGrad(C.Structure):
    _fields_ = [("g_out", C.c_void_p), ("y", C.c_void_p), ("ld", C.c_int64), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("mean", C.c_void_p), ("invstd", C.c_void_p), ("sums", C.c_void_p), ("relu", C.c_int32), ("n_valid", C.c_void_p)]


class DgnTowersLayer(C.Structure):
    _fields_ = [("graph", C.POINTER(DgnGraph)), ("spec", C.POINTER(DgnAggSpec)), ("w", C.c_void_p), ("ld_w", C.c_int64),
                ("log_deg", C.c_void_p), ("n_towers", C.c_int32), ("f_in", C.c_int32), ("f_out", C.c_int32), ("n_scalers", C.c_int32),
                ("residual", C.c_int32), ("momentum", C.c_float), ("eps", C.c_float), ("slope", C.c_float),
                ("h", C.c_void_p), ("snorm", C.c_void_p), ("scale", C.c_void_p), ("w_sd", C.c_void_p), ("bias_sd", C.c_void_p),
                ("w_post", C.c_void_p), ("b_post", C.c_void_p), ("bn_gamma", C.c_void_p), ("bn_beta", C.c_void_p),
                ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("w_mix", C.c_void_p), ("b_mix", C.c_void_p),
                ("pq", C.c_void_p), ("aggx", C.c_void_p), ("y0", C.c_void_p), ("save_mean", C.c_void_p), ("save_invstd", C.c_void_p),
                ("y1", C.c_void_p), ("z", C.c_void_p), ("out", C.c_void_p), ("ws", C.c_void_p), ("ws_bytes", C.c_size_t), ("n_valid", C.c_void_p),
                ("zmask", C.c_void_p), ("agg_aux", C.c_void_p),
                ("drop_p", C.c_float), ("drop_seed", C.c_void_p), ("drop_offset", C.c_uint64), ("drop_mask", C.c_void_p), ("id_slot1", C.c_int32),
                ("num_batches_tracked", C.c_void_p), ("n_nbt", C.c_int32)]


class DgnTowersGrads(C.Structure):
    _fields_ = [("g_out", C.c_void_p), ("g_h", C.c_void_p), ("g_w_sd", C.c_void_p), ("g_bias_sd", C.c_void_p), ("g_w_post", C.c_void_p),
                ("g_b_post", C.c_void_p), ("g_gamma", C.c_void_p), ("g_beta", C.c_void_p), ("g_w_mix", C.c_void_p), ("g_b_mix", C.c_void_p)]


class DgnDegreeClasses(C.Structure):
    _fields_ = [("n_units", C.c_int64), ("vperm", C.c_void_p), ("unit_class", C.c_void_p), ("present", C.c_void_p), ("scale", C.c_void_p)]


class DgnDcLayout(C.Structure):
    _fields_ = [("n_agg", C.c_int32), ("f_pad", C.c_int32), ("f_in", C.c_int32), ("h_off", C.c_int32), ("id_slot", C.c_int32), ("ld", C.c_int64)]


class DgnDenseLayer(C.Structure):
    _fields_ = [("graph", C.POINTER(DgnGraph)), ("spec", C.POINTER(DgnAggSpec)), ("w", C.c_void_p), ("ld_w", C.c_int64), ("log_deg", C.c_void_p),
                ("type", C.c_int32), ("f_in", C.c_int32), ("f_out", C.c_int32), ("n_scalers", C.c_int32), ("n_agg", C.c_int32), ("id_slot", C.c_int32),
                ("residual", C.c_int32), ("momentum", C.c_float), ("eps", C.c_float),
                ("h", C.c_void_p), ("snorm", C.c_void_p), ("scale", C.c_void_p), ("w_pre", C.c_void_p), ("b_pre", C.c_void_p),
                ("w_post", C.c_void_p), ("b_post", C.c_void_p), ("bn_gamma", C.c_void_p), ("bn_beta", C.c_void_p),
                ("running_mean", C.c_void_p), ("running_var", C.c_void_p),
                ("hp", C.c_void_p), ("pq", C.c_void_p), ("agg", C.c_void_p), ("y", C.c_void_p), ("wf", C.c_void_p), ("wsd", C.c_void_p),
                ("save_mean", C.c_void_p), ("save_invstd", C.c_void_p), ("out", C.c_void_p), ("ws", C.c_void_p), ("ws_bytes", C.c_size_t),
                ("n_valid", C.c_void_p), ("agg_aux", C.c_void_p), ("dc", C.POINTER(DgnDegreeClasses)),
                ("num_batches_tracked", C.c_void_p)]


class DgnDenseGrads(C.Structure):
    _fields_ = [("g_out", C.c_void_p), ("g_h", C.c_void_p), ("g_w_pre", C.c_void_p), ("g_b_pre", C.c_void_p), ("g_w_post", C.c_void_p),
                ("g_b_post", C.c_void_p), ("g_gamma", C.c_void_p), ("g_beta", C.c_void_p)]


DGN_BLK_MAX_TOWERS = 8


class DgnBlockTable(C.Structure):
    _fields_ = [("n_blocks", C.c_int32), ("max_rows", C.c_int32), ("max_edges", C.c_int32), ("desc", C.c_void_p)]


class DgnBlockLayer(C.Structure):
    _fields_ = [("graph", C.POINTER(DgnGraph)), ("blocks", C.POINTER(DgnBlockTable)), ("spec", C.POINTER(DgnAggSpec)), ("channels", C.POINTER(DgnChannel)),
                ("eig", C.c_void_p), ("ld_eig", C.c_int64), ("n_eig_cols", C.c_int32), ("log_deg", C.c_void_p),
                ("type", C.c_int32), ("n_towers", C.c_int32), ("f_in", C.c_int32), ("f_out", C.c_int32), ("residual", C.c_int32),
                ("momentum", C.c_float), ("eps", C.c_float), ("slope", C.c_float), ("h", C.c_void_p), ("snorm", C.c_void_p),
                ("w_pre", C.POINTER(C.c_void_p)), ("b_pre", C.POINTER(C.c_void_p)), ("w_post", C.POINTER(C.c_void_p)), ("b_post", C.POINTER(C.c_void_p)),
                ("gamma", C.POINTER(C.c_void_p)), ("beta", C.POINTER(C.c_void_p)), ("w_mix", C.c_void_p), ("b_mix", C.c_void_p),
                ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("num_batches_tracked", C.c_void_p), ("n_nbt", C.c_int32),
                ("y0", C.c_void_p), ("save_mean", C.c_void_p), ("save_invstd", C.c_void_p), ("out", C.c_void_p), ("ws", C.c_void_p), ("ws_bytes", C.c_size_t),
                ("n_valid", C.c_void_p), ("overflow", C.c_void_p), ("eval_mode", C.c_int32),
                ("dbg_agg", C.c_void_p), ("dbg_gagg", C.c_void_p), ("dbg_time", C.c_void_p),
                ("drop_p", C.c_float), ("drop_seed", C.c_void_p), ("drop_offset", C.c_uint64), ("drop_mask", C.c_void_p)]


class DgnBlockGrads(C.Structure):
    _fields_ = [("g_out", C.c_void_p), ("g_h", C.c_void_p), ("g_params", C.c_void_p), ("g_gamma", C.c_void_p), ("g_beta", C.c_void_p)]


class DgnError(RuntimeError):
    pass


_lib = None
_lock = threading.Lock()


def load() -> C.CDLL:
    """Load the shared library once; raise loudly if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise DgnError(f"{LIB_PATH} not found: the HIP extension is not built (run __graft_entry__.build() "
                           "or dgn_amd/csrc/build.sh). dgn_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        lib.dgn_abi_version.restype = C.c_int
        lib.dgn_last_error.restype = C.c_char_p
        lib.dgn_sizeof.restype = C.c_size_t
        lib.dgn_sizeof.argtypes = [C.c_char_p]
        lib.dgn_set_option.restype = C.c_int
        lib.dgn_set_option.argtypes = [C.c_char_p, C.c_int64]
        lib.dgn_get_option.restype = C.c_int64
        lib.dgn_get_option.argtypes = [C.c_char_p]
        lib.dgn_edge_weights_workspace_bytes.restype = C.c_size_t
        lib.dgn_edge_weights_workspace_bytes.argtypes = [C.POINTER(DgnGraph), C.c_int32]
        lib.dgn_edge_weights.restype = C.c_int
        lib.dgn_edge_weights.argtypes = [C.POINTER(DgnGraph), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                         C.POINTER(DgnChannel), C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t,
                                         C.c_void_p]
        lib.dgn_agg_workspace_bytes.restype = C.c_size_t
        lib.dgn_agg_workspace_bytes.argtypes = [C.POINTER(DgnGraph), C.POINTER(DgnAggSpec), C.c_int64]
        lib.dgn_agg_backward_workspace_bytes.restype = C.c_size_t
        lib.dgn_agg_backward_workspace_bytes.argtypes = [C.POINTER(DgnGraph), C.POINTER(DgnAggSpec), C.c_int64, C.c_int32]
        vp = C.c_void_p
        lib.dgn_linear_forward_bn.argtypes = [C.c_int64, C.c_int32, C.c_int32, vp, vp, C.c_int64, C.c_int32, vp, vp, vp, vp, vp, vp, vp]
        lib.dgn_linear_forward_act.argtypes = [C.c_int64, C.c_int32, C.c_int32, vp, vp, vp, C.c_int32, C.c_float, vp, C.c_int64, C.c_int32, vp, vp, vp]
        lib.dgn_linear_forward_add.argtypes = [C.c_int64, C.c_int32, C.c_int32, vp, vp, C.c_int64, C.c_int32, vp, vp, vp, vp]
        lib.dgn_linear_forward_bn_act.argtypes = [C.c_int64, C.c_int32, C.c_int32, vp, vp, C.c_int64, vp, vp, vp, vp, vp, C.c_int32, C.c_float, vp, vp, vp, vp]
        lib.dgn_linear_wgrad_bn.argtypes = [C.c_int64, C.c_int32, C.c_int32, vp, vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
        lib.dgn_linear_bnb_supported.restype = C.c_int
        lib.dgn_linear_bnb_supported.argtypes = [C.c_int32, C.c_int32]
        lib.dgn_linear_wgrad_bn_act_mask.restype = C.c_int
        lib.dgn_linear_wgrad_bn_act_mask.argtypes = [C.c_int64, C.c_int32, C.c_int32, vp, vp, C.c_int32, C.c_float, vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp,
                                                     C.c_size_t, vp]
        lib.dgn_linear_forward_act_mask_bnb.restype = C.c_int
        lib.dgn_linear_forward_act_mask_bnb.argtypes = [C.c_int64, C.c_int32, C.c_int32, vp, vp, C.c_int32, C.c_float, vp, C.c_int64, C.c_int32, vp, vp, vp, vp, vp,
                                                        vp, C.c_int32, vp, C.c_int64, vp]
        lib.dgn_linear_combine_backward_weight_bias.restype = C.c_int
        lib.dgn_linear_combine_backward_weight_bias.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, C.c_int64, vp, vp, C.c_int64, vp,
                                                                C.c_int64, C.c_int64, vp, vp, C.c_size_t, vp]
        lib.dgn_linear_act_mask_bytes.restype = C.c_size_t
        lib.dgn_linear_act_mask_bytes.argtypes = [C.c_int64, C.c_int32]
        lib.dgn_linear_forward_bn_act_mask.restype = C.c_int
        lib.dgn_linear_forward_bn_act_mask.argtypes = [C.c_int64, C.c_int32, C.c_int32, vp, vp, C.c_int64, vp, vp, vp, vp, vp, C.c_int32, C.c_float, vp, vp, vp, vp]
        lib.dgn_linear_forward_act_mask.restype = C.c_int
        lib.dgn_linear_forward_act_mask.argtypes = [C.c_int64, C.c_int32, C.c_int32, vp, vp, C.c_int32, C.c_float, vp, C.c_int64, C.c_int32, vp, vp, vp]
        lib.dgn_towers_layer_zmask_supported.restype = C.c_int
        lib.dgn_towers_layer_zmask_supported.argtypes = [C.c_int32, C.c_int32]
        lib.dgn_agg_edge_table_workspace_bytes.restype = C.c_size_t
        lib.dgn_agg_edge_table_workspace_bytes.argtypes = [C.c_int64, C.c_int32]
        lib.dgn_agg_forward.restype = C.c_int
        lib.dgn_agg_forward.argtypes = [C.POINTER(DgnGraph), C.POINTER(DgnAggSpec), C.POINTER(DgnMsg), C.c_void_p,
                                        C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t,
                                        C.c_void_p]
        lib.dgn_agg_aux_bytes.restype = C.c_size_t
        lib.dgn_agg_aux_bytes.argtypes = [C.POINTER(DgnGraph), C.POINTER(DgnAggSpec), C.POINTER(DgnMsg)]
        lib.dgn_agg_forward_aux.restype = C.c_int
        lib.dgn_agg_forward_aux.argtypes = [C.POINTER(DgnGraph), C.POINTER(DgnAggSpec), C.POINTER(DgnMsg), C.c_void_p, C.c_int64, C.c_void_p,
                                            C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.dgn_agg_backward_aux.restype = C.c_int
        lib.dgn_agg_backward_aux.argtypes = [C.POINTER(DgnGraph), C.POINTER(DgnAggSpec), C.POINTER(DgnMsg), C.c_void_p, C.c_int64, C.c_void_p,
                                             C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(DgnMsgGrad), C.c_void_p, C.c_size_t, C.c_void_p]
        lib.dgn_dense_layer_agg_aux_bytes.restype = C.c_size_t
        lib.dgn_dense_layer_agg_aux_bytes.argtypes = [C.POINTER(DgnDenseLayer)]
        lib.dgn_towers_layer_agg_aux_bytes.restype = C.c_size_t
        lib.dgn_towers_layer_agg_aux_bytes.argtypes = [C.POINTER(DgnTowersLayer)]
        lib.dgn_agg_backward.restype = C.c_int
        lib.dgn_agg_backward.argtypes = [C.POINTER(DgnGraph), C.POINTER(DgnAggSpec), C.POINTER(DgnMsg), C.c_void_p,
                                         C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(DgnMsgGrad),
                                         C.c_void_p, C.c_size_t, C.c_void_p]
        lib.dgn_scale_combine_forward.restype = C.c_int
        lib.dgn_scale_combine_forward.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        lib.dgn_scale_combine_backward.restype = C.c_int
        lib.dgn_scale_combine_backward.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p,
                                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(DgnBnGrad),
                                                   C.c_void_p]
        lib.dgn_scale_combine_backward_workspace_bytes.restype = C.c_size_t
        lib.dgn_scale_combine_backward_workspace_bytes.argtypes = [C.c_int64, C.c_int32, C.c_int32]
        lib.dgn_bn_tail_workspace_bytes.restype = C.c_size_t
        lib.dgn_bn_tail_workspace_bytes.argtypes = [C.c_int64, C.c_int32]
        lib.dgn_bn_tail_forward.restype = C.c_int
        lib.dgn_bn_tail_forward.argtypes = [C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        lib.dgn_bn_tail_backward.restype = C.c_int
        lib.dgn_bn_tail_backward.argtypes = [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_size_t, C.c_void_p, C.c_void_p]
        lib.dgn_bias_act_forward.restype = C.c_int
        lib.dgn_bias_act_forward.argtypes = [C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_float, C.c_void_p,
                                             C.c_void_p, C.c_void_p]
        lib.dgn_dropout_mask_bytes.restype = C.c_size_t
        lib.dgn_dropout_mask_bytes.argtypes = [C.c_int64]
        lib.dgn_dropout_forward.restype = C.c_int
        lib.dgn_dropout_forward.argtypes = [C.c_int64, C.c_void_p, C.c_float, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.dgn_dropout_backward.restype = C.c_int
        lib.dgn_dropout_backward.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
        lib.dgn_bias_act_backward.restype = C.c_int
        lib.dgn_bias_act_backward.argtypes = [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_float,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.dgn_linear_supported.restype = C.c_int
        lib.dgn_linear_supported.argtypes = [C.c_int32, C.c_int32, C.c_int32]
        lib.dgn_linear_forward.restype = C.c_int
        lib.dgn_linear_forward.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                           C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                           C.c_void_p]
        lib.dgn_linear_combine_forward.restype = C.c_int
        lib.dgn_linear_combine_forward.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p,
                                                   C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                                   C.c_void_p]
        lib.dgn_linear_combine_backward_input.restype = C.c_int
        lib.dgn_linear_combine_backward_input.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                                          C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
        lib.dgn_linear_combine_backward_weight.restype = C.c_int
        lib.dgn_linear_combine_backward_weight.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                                           C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                                           C.c_size_t, C.c_void_p]
        lib.dgn_linear_wgrad_workspace_bytes.restype = C.c_size_t
        lib.dgn_linear_wgrad_workspace_bytes.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int32]
        lib.dgn_linear_wgrad.restype = C.c_int
        lib.dgn_linear_wgrad.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                         C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                         C.c_size_t, C.c_void_p]
        lib.dgn_dense_layer_supported.restype = C.c_int
        lib.dgn_dense_layer_supported.argtypes = [C.c_int32] * 5
        lib.dgn_dense_layer_forward_workspace_bytes.restype = C.c_size_t
        lib.dgn_dense_layer_forward_workspace_bytes.argtypes = [C.POINTER(DgnDenseLayer)]
        lib.dgn_dense_layer_backward_workspace_bytes.restype = C.c_size_t
        lib.dgn_dense_layer_backward_workspace_bytes.argtypes = [C.POINTER(DgnDenseLayer)]
        lib.dgn_dense_layer_forward.restype = C.c_int
        lib.dgn_dense_layer_forward.argtypes = [C.POINTER(DgnDenseLayer), vp]
        lib.dgn_dense_layer_backward.restype = C.c_int
        lib.dgn_dense_layer_backward.argtypes = [C.POINTER(DgnDenseLayer), C.POINTER(DgnDenseGrads), vp]
        lib.dgn_linear_bd_supported.restype = C.c_int
        lib.dgn_linear_bd_supported.argtypes = [C.c_int32, C.c_int32]
        lib.dgn_linear_bd_forward.restype = C.c_int
        lib.dgn_linear_bd_forward.argtypes = [C.c_int64, C.c_int32, C.c_int32, vp, vp, C.c_int64, vp, vp, vp]
        lib.dgn_linear_bd_backward_input.restype = C.c_int
        lib.dgn_linear_bd_backward_input.argtypes = [C.c_int64, C.c_int32, C.c_int32, vp, vp, C.c_int64, vp, vp, vp, vp]
        lib.dgn_linear_bd_wgrad_workspace_bytes.restype = C.c_size_t
        lib.dgn_linear_bd_wgrad_workspace_bytes.argtypes = [C.c_int64, C.c_int32, C.c_int32]
        lib.dgn_linear_bd_wgrad.restype = C.c_int
        lib.dgn_linear_bd_wgrad.argtypes = [C.c_int64, C.c_int32, C.c_int32, vp, vp, vp, C.c_int64, vp, vp, C.c_size_t, vp]
        for name in ("dgn_dc_supported", "dgn_dc_wgrad_supported"):
            getattr(lib, name).restype = C.c_int
            getattr(lib, name).argtypes = [C.c_int32, C.c_int32]
        lib.dgn_dc_fold.restype = C.c_int
        lib.dgn_dc_fold.argtypes = [C.POINTER(DgnDegreeClasses), C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, C.POINTER(DgnDcLayout), vp, vp, vp]
        lib.dgn_dc_gemm.restype = C.c_int
        lib.dgn_dc_gemm.argtypes = [C.POINTER(DgnDegreeClasses), C.c_int32, C.c_int32, C.c_int32, vp, C.c_int64, C.c_int64, vp, C.c_int64, C.c_int64,
                                    C.c_int64, vp, vp, vp, C.c_int64, C.c_int64, C.c_int32, vp]
        lib.dgn_dc_wgrad_workspace_bytes.restype = C.c_size_t
        lib.dgn_dc_wgrad_workspace_bytes.argtypes = [C.c_int64, C.c_int32, C.c_int32]
        lib.dgn_dc_wgrad.restype = C.c_int
        lib.dgn_dc_wgrad.argtypes = [C.POINTER(DgnDegreeClasses), C.c_int32, C.c_int32, C.c_int32, vp, C.c_int64, vp, C.c_int64, vp, C.c_int64,
                                     C.POINTER(DgnDcLayout), vp, C.c_size_t, vp]
        lib.dgn_gemm_supported.restype = C.c_int
        lib.dgn_gemm_supported.argtypes = [C.c_int32, C.c_int32]
        lib.dgn_gemm_forward.restype = C.c_int
        lib.dgn_gemm_forward.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p,
                                         C.c_void_p, C.c_int64, C.c_void_p]
        lib.dgn_gemm_wgrad_workspace_bytes.restype = C.c_size_t
        lib.dgn_gemm_wgrad_workspace_bytes.argtypes = [C.c_int64, C.c_int32, C.c_int32]
        lib.dgn_gemm_wgrad.restype = C.c_int
        lib.dgn_gemm_wgrad.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                       C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.dgn_layer_fused_supported.restype = C.c_int
        lib.dgn_layer_fused_supported.argtypes = [C.POINTER(DgnGraph), C.POINTER(DgnAggSpec), C.c_int64, C.c_int32, C.c_int32]
        lib.dgn_layer_fused_forward.restype = C.c_int
        lib.dgn_layer_fused_forward.argtypes = [C.POINTER(DgnGraph), C.POINTER(DgnAggSpec), C.POINTER(DgnMsg), C.c_void_p, C.c_int64,
                                                C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        lib.dgn_graph_build_workspace_bytes.restype = C.c_size_t
        lib.dgn_graph_build_workspace_bytes.argtypes = [C.c_int64, C.c_int64]
        lib.dgn_graph_build.restype = C.c_int
        lib.dgn_graph_build.argtypes = [C.c_int64, C.c_int64] + [C.c_void_p] * 9 + [C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.dgn_graph_build_cuts.restype = C.c_int
        lib.dgn_graph_build_cuts.argtypes = [C.c_int64, C.c_int64] + [C.c_void_p] * 4 + [C.c_void_p, C.c_size_t, C.c_void_p]
        lib.dgn_graph_build_csc.restype = C.c_int
        lib.dgn_graph_build_csc.argtypes = [C.c_int64, C.c_int64] + [C.c_void_p] * 4 + [C.c_void_p, C.c_size_t, C.c_void_p]
        lib.dgn_assemble_params.restype = C.c_int
        lib.dgn_assemble_params.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.dgn_towers_layer_supported.restype = C.c_int
        lib.dgn_towers_layer_supported.argtypes = [C.c_int32] * 5
        for name in ("dgn_towers_layer_forward_workspace_bytes", "dgn_towers_layer_backward_workspace_bytes"):
            getattr(lib, name).restype = C.c_size_t
            getattr(lib, name).argtypes = [C.POINTER(DgnTowersLayer)]
        lib.dgn_towers_layer_forward.restype = C.c_int
        lib.dgn_towers_layer_forward.argtypes = [C.POINTER(DgnTowersLayer), C.c_void_p]
        lib.dgn_towers_layer_backward.restype = C.c_int
        lib.dgn_towers_layer_backward.argtypes = [C.POINTER(DgnTowersLayer), C.POINTER(DgnTowersGrads), C.c_void_p]
        lib.dgn_block_layer_supported.restype = C.c_int
        lib.dgn_block_layer_supported.argtypes = [C.POINTER(DgnBlockLayer)]
        lib.dgn_block_layer_param_grad_floats.restype = C.c_int64
        lib.dgn_block_layer_param_grad_floats.argtypes = [C.POINTER(DgnBlockLayer)]
        for name in ("dgn_block_layer_forward_workspace_bytes", "dgn_block_layer_backward_workspace_bytes"):
            getattr(lib, name).restype = C.c_size_t
            getattr(lib, name).argtypes = [C.POINTER(DgnBlockLayer)]
        lib.dgn_block_layer_forward.restype = C.c_int
        lib.dgn_block_layer_forward.argtypes = [C.POINTER(DgnBlockLayer), vp]
        lib.dgn_block_layer_backward.restype = C.c_int
        lib.dgn_block_layer_backward.argtypes = [C.POINTER(DgnBlockLayer), C.POINTER(DgnBlockGrads), vp]
        if lib.dgn_abi_version() != ABI_VERSION:
            raise DgnError(f"libdgn_hip.so ABI {lib.dgn_abi_version()} != binding {ABI_VERSION}: rebuild")
        _lib = lib
    return _lib


class _Options:
    """The library's process-wide options as attributes (``dgn_set_option`` / ``dgn_get_option`` of the C ABI): ``options.blk_min_nodes = 0``.
    ``monkeypatch.setattr(_lib.options, name, value)`` switches one for a test and restores it."""

    def __getattr__(self, name):
        v = load().dgn_get_option(name.encode())
        if v == -2 ** 63:
            raise AttributeError(f"unknown libdgn_hip option '{name}'")
        return v

    def __setattr__(self, name, value):
        check(load().dgn_set_option(name.encode(), int(value)), "dgn_set_option")


options = _Options()


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise DgnError(f"{what} failed (rc={rc}): {load().dgn_last_error().decode()}")


def stream_ptr(device) -> int:
    """Raw handle of torch's current stream on ``device`` (what every entry point takes as its ``stream`` argument).
    ``torch.cuda.current_stream(d).cuda_stream`` builds a Stream object per call (~7 us of an eager step that is launch-bound)."""
    import torch
    idx = device.index if getattr(device, "index", None) is not None else torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(idx)
