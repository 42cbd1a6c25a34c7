"""ctypes binding of libwsmgmap.so (C ABI declared in include/wsmgmap.h).

This is the only place the shared library is touched.  There is NO fallback: if the
library is missing, or a call returns non-zero, a WsmgError is raised — the product path
never silently degrades to PyTorch or CPU code.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libwsmgmap.so")
if os.environ.get("WSMG_LIB"):       # experiments only: another build of the same library (read once, here)
    LIB_PATH = os.environ["WSMG_LIB"]

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_l = ctypes.c_int64
c_f = ctypes.c_float


class WsmgError(RuntimeError):
    pass


class CopyDesc(ctypes.Structure):         # WsmgCopyDesc of include/wsmgmap.h
    _fields_ = [("dst", ctypes.c_void_p), ("src", ctypes.c_void_p), ("bytes", ctypes.c_longlong)]


class RelayoutDesc(ctypes.Structure):     # WsmgRelayoutDesc of include/wsmgmap.h
    _fields_ = [("w_oihw", c_p), ("w_ohwi", c_p), ("w_ihwo", c_p), ("O", c_i), ("I", c_i), ("KH", c_i), ("KW", c_i),
                ("I_pad", c_i), ("reserved", c_i)]


class ColsumDesc(ctypes.Structure):       # WsmgColsumDesc of include/wsmgmap.h (round 6)
    _fields_ = [("x", c_p), ("out", c_p), ("rows", c_i), ("cols", c_i)]


# name -> argtypes (all return int unless listed in _RESTYPE)
_SIG = {
    "wsmg_abi_version": [],
    "wsmg_build_info": [],
    "wsmg_bev_index": [c_p, c_i, c_i, c_i, c_f, c_i, c_i, c_i, c_f, c_p, c_p],
    "wsmg_bev_scatter_max": [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p],
    "wsmg_bev_rotate": [c_p, c_p, c_f, c_i, c_i, c_i, c_p, c_p],
    "wsmg_map_fuse": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p],
    "wsmg_map_retrieve": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_p],
    "wsmg_map_retrieve_fused": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_p],
    "wsmg_map_retrieve_tiled": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_p],
    "wsmg_bev_scatter_rotate": [c_p, c_p, c_p, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p],
    "wsmg_map_fuse_planes": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p],
    "wsmg_conv2d_fwd": [c_p, c_p, c_p, c_p] + [c_i] * 11 + [c_p],
    "wsmg_conv2d_bwd_data": [c_p, c_p, c_p] + [c_i] * 11 + [c_p],
    "wsmg_conv2d_bwd_weight": [c_p, c_p, c_p] + [c_i] * 11 + [c_p],
    "wsmg_channel_sum": [c_p, c_l, c_i, c_p, c_p, c_l, c_p],
    "wsmg_channel_reduce_workspace_bytes": [c_l, c_i],
    "wsmg_bn_act_fwd": [c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_i, c_i, c_l, c_i, c_p, c_p, c_p, c_p, c_l, c_p],
    "wsmg_bn_act_bwd": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_l, c_i, c_p, c_p, c_p, c_p, c_p, c_l, c_p],
    "wsmg_relu_fwd": [c_p, c_p, c_l, c_p],
    "wsmg_relu_bwd": [c_p, c_p, c_p, c_l, c_p],
    "wsmg_token_grad_merge": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "wsmg_maxpool3x3s2_fwd": [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "wsmg_maxpool3x3s2_bwd": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "wsmg_upsample2x_fwd": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "wsmg_upsample2x_bwd": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "wsmg_avgpool2_fwd": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "wsmg_avgpool2_bwd": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "wsmg_nchw_to_nhwc": [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "wsmg_nhwc_to_nchw": [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "wsmg_attn_fwd": [c_p, c_p, c_p, c_p, c_f, c_i, c_i, c_i, c_p, c_p, c_p],
    "wsmg_attn_bwd": [c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_i, c_i, c_i, c_p, c_p, c_p, c_p],
}
# bf16 storage variants: identical argument lists except the convs' extra `out_f32` int
for _n in ["wsmg_token_grad_merge", "wsmg_channel_sum", "wsmg_bn_act_fwd", "wsmg_bn_act_bwd", "wsmg_relu_fwd", "wsmg_relu_bwd",
           "wsmg_maxpool3x3s2_fwd", "wsmg_maxpool3x3s2_bwd", "wsmg_upsample2x_fwd", "wsmg_upsample2x_bwd",
           "wsmg_avgpool2_fwd", "wsmg_avgpool2_bwd", "wsmg_nchw_to_nhwc", "wsmg_nhwc_to_nchw", "wsmg_attn_fwd",
           "wsmg_attn_bwd", "wsmg_conv2d_bwd_weight"]:
    _SIG[_n + "_bf16"] = list(_SIG[_n])
_SIG["wsmg_conv2d_fwd_bf16"] = [c_p, c_p, c_p, c_p, c_i] + [c_i] * 11 + [c_p]
_SIG["wsmg_conv2d_bwd_data_bf16"] = [c_p, c_p, c_p, c_i] + [c_i] * 11 + [c_p]
_SIG["wsmg_conv_transpose2d_infer_bf16"] = [c_p, c_p, c_p, c_p, c_i] + [c_i] * 11 + [c_p]
_SIG["wsmg_conv2d_splitk_plan"] = [c_i] * 7 + [c_p, c_p]
_SIG["wsmg_conv2d_fwd_bf16_splitk"] = [c_p, c_p, c_p, c_p, c_i, c_i, c_p] + [c_i] * 11 + [c_p]
_SIG["wsmg_conv2d_fwd_bf16_stats"] = [c_p, c_p, c_p, c_p, c_i, c_p, c_i] + [c_i] * 11 + [c_p]
_SIG["wsmg_conv2d_bwd_data_bf16_stats"] = [c_p, c_p, c_p, c_i, c_p, c_i] + [c_i] * 11 + [c_p]
_SIG["wsmg_bn_act_fwd_bf16_pre"] = [c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_i, c_l, c_i, c_p, c_p, c_p, c_p, c_i, c_p]
_SIG["wsmg_attn_shared_fwd"] = [c_p, c_p, c_p, c_p, c_p, c_f, c_i, c_i, c_i, c_p, c_p, c_p]
_SIG["wsmg_attn_shared_bwd"] = [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_i, c_i, c_i, c_p, c_p, c_p]
_SIG["wsmg_attn_shared_fwd_bf16"] = list(_SIG["wsmg_attn_shared_fwd"])
_SIG["wsmg_attn_shared_bwd_bf16"] = list(_SIG["wsmg_attn_shared_bwd"])
_SIG["wsmg_attn_fp8_fold"] = [c_p, c_p, c_i, c_i, c_i, c_p, c_p]
_SIG["wsmg_attn_fp8_splits"] = [c_i, c_i]
_SIG["wsmg_attn_fp8_workspace_bytes"] = [c_i, c_i]
_SIG["wsmg_attn_fp8_fwd"] = [c_p, c_p, c_p, c_p, c_f, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p]
_SIG["wsmg_attn_fp8_bwd"] = [c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_i, c_i, c_i, c_p, c_p, c_p]
_SIG["wsmg_quantize_e4m3_dev"] = [c_p, c_l, c_p, c_p, c_p]
_SIG["wsmg_quantize_e4m3"] = [c_p, c_l, c_f, c_p, c_p]
_SIG["wsmg_weight_relayout"] = [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p]
_SIG["wsmg_weight_relayout_bf16"] = list(_SIG["wsmg_weight_relayout"])
_SIG["wsmg_weight_relayout_multi"] = [c_p, c_i, c_i, c_p]
_SIG["wsmg_weight_grad_to_oihw"] = [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p]
_SIG["wsmg_conv2d_bwd_weight_plan"] = [c_i] * 11 + [c_p, c_p]
_SIG["wsmg_conv2d_bwd_weight_bf16_plan"] = list(_SIG["wsmg_conv2d_bwd_weight_plan"])
_SIG["wsmg_conv2d_bwd_weight_slabs"] = [c_p, c_p, c_p, c_i, c_l] + [c_i] * 11 + [c_p]
_SIG["wsmg_conv2d_bwd_weight_bf16_slabs"] = list(_SIG["wsmg_conv2d_bwd_weight_slabs"])
_SIG["wsmg_weight_grad_reduce_oihw"] = [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p]
_SIG["wsmg_update_heads_fwd"] = [c_p] * 6 + [c_i] * 3 + [c_p] * 3 + [c_p]
_SIG["wsmg_update_heads_bwd"] = [c_p] * 8 + [c_i] * 3 + [c_p] * 5 + [c_p]
_SIG["wsmg_aux_reduce_fwd"] = [c_p, c_p, c_i, c_p, c_i, c_p, c_p]
_SIG["wsmg_aux_reduce_bwd"] = [c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_p]
_SIG["wsmg_dagger_loss_fwd"] = [c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p]
_SIG["wsmg_dagger_loss_bwd"] = [c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p]
_SIG["wsmg_cls_tail_workspace_floats"] = [c_i]
_SIG["wsmg_cls_tail_fwd_bf16"] = [c_p] * 7 + [c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p]
_SIG["wsmg_cls_tail_bwd_bf16"] = [c_p] * 7 + [c_i, c_p, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_l, c_p, c_p, c_p, c_p, c_p]
_SIG["wsmg_bn_stats_finalize"] = [c_p, c_i, c_i, c_l, c_f, c_f, c_p, c_p, c_p, c_p, c_p]
_SIG["wsmg_bn_bwd_apply_bf16"] = [c_p] * 7 + [c_l, c_i, c_p, c_p]
_SIG["wsmg_attn_fp8_mfma_fwd"] = [c_p] * 9 + [c_f, c_i, c_i, c_i, c_i, c_p, c_p, c_p]
_SIG["wsmg_maxpool3x3s2_fwd_idx_bf16"] = [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]
_SIG["wsmg_maxpool3x3s2_bwd_idx_bf16"] = [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]
_SIG["wsmg_add3_bf16"] = [c_p, c_p, c_p, c_p, c_l, c_p]
_SIG["wsmg_attn_fp8_prep"] = [c_p] * 4 + [c_i, c_i, c_i, c_i, c_f, c_f, c_f] + [c_p] * 8
_SIG["wsmg_mean_rows"] = [c_p, c_l, c_i, c_p, c_p]
_SIG["wsmg_bn_act_bwd_ld"] = [c_p, c_l] + _SIG["wsmg_bn_act_bwd"][1:]
_SIG["wsmg_bn_act_bwd_ld_bf16"] = list(_SIG["wsmg_bn_act_bwd_ld"])
_SIG["wsmg_bn_act_bwd_ld_bf16_lo"] = [c_p, c_l] + [c_p] * 6 + [c_i, c_l, c_i] + [c_p] * 6 + [c_l, c_p]
_SIG["wsmg_upsample2x_bwd_ld"] = [c_p, c_l, c_p, c_i, c_i, c_i, c_i, c_p]
_SIG["wsmg_upsample2x_bwd_ld_bf16"] = list(_SIG["wsmg_upsample2x_bwd_ld"])
_SIG["wsmg_relu_bwd_rows_bf16"] = [c_p, c_l, c_p, c_p, c_l, c_i, c_p]
_SIG["wsmg_cat_channels"] = [c_p, c_p, c_p, c_l, c_i, c_i, c_p]
_SIG["wsmg_upsample2x_cat_bf16"] = [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]
_SIG["wsmg_ce_nhwc_fwd"] = [c_p, c_p, c_l, c_i, c_p, c_p]
_SIG["wsmg_ce_nhwc_bwd"] = [c_p, c_p, c_p, c_l, c_i, c_p, c_p]
_SIG["wsmg_ce_nhwc_fwd_bf16"] = list(_SIG["wsmg_ce_nhwc_fwd"])
_SIG["wsmg_ce_nhwc_bwd_bf16"] = list(_SIG["wsmg_ce_nhwc_bwd"])
_SIG["wsmg_collate_pad"] = [c_p, c_p, c_i, c_i, c_l, c_i, c_f, c_p, c_p]
_SIG["wsmg_collate_pad_nhwc_bf16"] = [c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_p]
_SIG["wsmg_gru_workspace_bytes"] = [c_i]
_SIG["wsmg_gru_fwd"] = [c_p] * 5 + [c_i] * 3 + [c_p] * 6 + [c_p]
_SIG["wsmg_gru_bwd"] = [c_p] * 10 + [c_i] * 3 + [c_p] * 4 + [c_p]
_SIG["wsmg_gru_fwd_owned"] = list(_SIG["wsmg_gru_fwd"])
_SIG["wsmg_gru_bwd_owned"] = list(_SIG["wsmg_gru_bwd"])
_SIG["wsmg_lstm_workspace_bytes"] = [c_i]
_SIG["wsmg_lstm_fwd"] = [c_p] * 4 + [c_i] * 3 + [c_p] * 4 + [c_p]
_SIG["wsmg_lstm_bwd"] = [c_p] * 5 + [c_i] * 3 + [c_p] * 2 + [c_p]
_SIG["wsmg_group_norm_nhwc_bf16"] = [c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_i, c_p, c_p]
_SIG["wsmg_rnn_status"] = [c_i]
_SIG["wsmg_rnn_debug_spin_limit"] = [ctypes.c_uint]
_SIG["wsmg_rnn_debug_inject"] = [ctypes.c_uint]
_SIG["wsmg_instruction_dedup"] = [c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p]
_SIG["wsmg_conv_debug_win3_tile"] = [c_i]
_SIG["wsmg_copy_multi"] = [c_p, c_i, c_p]
_SIG["wsmg_linear_rows"] = [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]
_SIG["wsmg_act_heads"] = [c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]
_SIG["wsmg_path_kl_fwd"] = [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_p]
_SIG["wsmg_path_kl_bwd"] = [c_p, c_p, c_p, c_i, c_i, c_p, c_p]
_SIG["wsmg_adam_step_multi"] = [c_p, c_i, c_f, c_f, c_f, c_f, c_f, ctypes.c_double, ctypes.c_double, c_p]
_SIG["wsmg_adam_step_multi_dev"] = [c_p, c_i, c_f, c_f, c_f, c_f, c_f, c_p, c_p]
_SIG["wsmg_rows_gemm_f32"] = ([c_p, c_i, c_i] * 3 + [c_p, c_i, c_i, c_p, c_p, c_i, c_i] + [c_p, c_i, c_i] * 3 + [c_p, c_i] * 3
                              + [c_i, c_p, ctypes.c_uint, c_p, c_i, c_p, c_p])
_SIG["wsmg_rows_gemm_workgroups"] = [c_i, c_i]
_SIG["wsmg_rows_gemm_supported"] = [c_i]
_SIG["wsmg_gru_fwd_chain"] = [c_p] * 5 + [c_i] * 3 + [c_p] * 6 + [c_i, c_p, ctypes.c_uint, c_p, c_p]
_SIG["wsmg_gru_bwd_chain"] = [c_p] * 10 + [c_i] * 3 + [c_p] * 4 + [c_i, c_p, ctypes.c_uint, c_p, c_p]
_SIG["wsmg_gru_chain_workgroups"] = []
_SIG["wsmg_attn_fp8_mfma_fused"] = [c_p] * 5 + [c_f] * 4 + [c_i] * 4 + [c_p, ctypes.c_uint, c_i, c_p, c_p, c_p, c_p]
_SIG["wsmg_attn_fp8_mfma_fused_arrivals"] = [c_i] * 4
_SIG["wsmg_collate_ego_sparse_nhwc_bf16"] = [c_p] * 5 + [c_i] * 4 + [c_f, c_p, c_p]
_SIG["wsmg_debug_occupy"] = [c_i, c_i, c_i, c_p, c_p, c_p]
_SIG["wsmg_conv2d_bwd_data_bf16_ex"] = [c_p] * 5 + [c_i] * 12 + [c_p]
_SIG["wsmg_conv2d_fwd_bf16_ex"] = [c_p] * 4 + [c_i, c_p, c_i, c_i] + [c_i] * 11 + [c_p]
_SIG["wsmg_bev_index_compact"] = [c_p, c_i, c_i, c_i, c_f, c_i, c_i, c_i, c_f, c_p, c_p, c_p, c_p]
_SIG["wsmg_bev_scatter_rotate_compact"] = [c_p, c_p, c_p, c_p, c_f] + [c_i] * 6 + [c_p, c_p]
_SIG["wsmg_colsum_multi"] = [c_p, c_i, c_p]
_SIG["wsmg_attn_fp8_row_fwd"] = [c_p] * 5 + [c_f, c_i, c_i, c_i] + [c_p] * 4
_RESTYPE = {"wsmg_cls_tail_workspace_floats": c_l, "wsmg_attn_fp8_workspace_bytes": c_l, "wsmg_lstm_workspace_bytes": c_l, "wsmg_build_info": ctypes.c_char_p, "wsmg_channel_reduce_workspace_bytes": c_l, "wsmg_gru_workspace_bytes": c_l}

_lib = None


def lib():
    """The loaded library; raises WsmgError (never falls back) if it cannot be loaded."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WsmgError(
                f"{LIB_PATH} not found: build it with `make -C ws-mgmap_amd/csrc` "
                "(or __graft_entry__.build()). The HIP kernels are mandatory; there is no fallback path.")
        try:
            L = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise WsmgError(f"cannot load {LIB_PATH}: {e}") from e
        for name, args in _SIG.items():
            try:
                fn = getattr(L, name)
            except AttributeError as e:
                raise WsmgError(f"{LIB_PATH} does not export {name}") from e
            fn.argtypes = args
            fn.restype = _RESTYPE.get(name, c_i)
        info = L.wsmg_build_info() or b""
        if b"rnn-nopk" not in info:
            # the GRU / LSTM gradient kernels are only correct when wsmg_rnn.hip was compiled without packed-fp32 instructions
            # (csrc/Makefile: EXTRA_wsmg_rnn; 177 wrong tensors in 60 loaded repeats otherwise): a stale object or another build
            # of the library (WSMG_LIB) must not bring that back silently
            raise WsmgError(f"{LIB_PATH} ({info.decode(errors='replace')}) was not built with the RNN kernels' compiler flags "
                            "(no 'rnn-nopk' in wsmg_build_info()): rebuild with `make -C ws-mgmap_amd/csrc`")
        _lib = L
    return _lib


def call(name, *args):
    """Call an int-returning entry point; non-zero is an error."""
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        kind = {-1: "WSMG_EINVAL (rejected arguments)", -2: "WSMG_ENOMEM (workspace too small)"}.get(rc, f"hipError_t {rc}")
        raise WsmgError(f"{name} failed: {kind}")


rnn_timeouts = 0      # persistent-RNN timeouts reported so far in this process


# Under a process group an exception that ONE rank raises between two collectives leaves its peers blocked in theirs.  While a
# GradAllReducer is active the forward pass's status checks therefore do nothing (`defer_rnn_status`); the reducer's `finish()`
# reads the word (force=True) and the ranks agree on the error before any of them raises (wsmgmap/parallel.py).
defer_rnn_status = False

STATUS_BITS = ((1, "gru_fwd"), (2, "gru_bwd"), (4, "lstm_fwd"), (8, "lstm_bwd"), (16, "attn_fp8_fused barrier"))


def check_rnn_status(sync=False, force=False):
    """Raise WsmgError if a persistent RNN kernel reported a timeout since the last check.  Reads a word in host-mapped
    pinned memory: no device synchronisation by itself.  Called at the host's natural sync points (the instruction dedup
    read-back of every forward pass, GradAllReducer.finish(), the end of an update in bench / tests) — those see every kernel
    that had FINISHED by then, so a timeout of the current update can surface one update late.  sync=True waits for the device
    first: the exact form, for a trainer that wants the guarantee that NaN-filled outputs never reach optimizer.step()
    (`BasePolicy.check_status()`).  force: also while a GradAllReducer defers the checks (its own call)."""
    global rnn_timeouts
    if defer_rnn_status and not force:
        return
    if sync:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    v = lib().wsmg_rnn_status(1)
    if v:
        rnn_timeouts += 1       # (owners of persistent, never-cleared RNN workspaces re-zero them: the error word in them is sticky)
        names = [n for b, n in STATUS_BITS if v & b]
        raise WsmgError("persistent kernel(s) timed out waiting for their cooperating workgroups: " + ", ".join(names) +
                        " — their outputs were filled with NaN; results since the previous check are invalid "
                        "(the persistent kernels need all their workgroups resident at once: too many concurrent persistent "
                        "kernels for the free CUs?  net.recurrent_chunks = 0 runs one at a time)")


def take_rnn_status():
    """The status word's bits since the last read, cleared, WITHOUT raising (0: nothing timed out) — for callers that handle a
    timeout themselves (bench.py's in-process fallback).  Counts as a reported timeout for the owners of never-cleared workspaces."""
    global rnn_timeouts
    v = int(lib().wsmg_rnn_status(1))
    if v:
        rnn_timeouts += 1
    return v


def status_names(bits):
    return [n for b, n in STATUS_BITS if bits & b]


def exported_names():
    return list(_SIG)
