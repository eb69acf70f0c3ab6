"""ctypes binding of libmulactseg_hip.so (the C ABI declared in include/mulactseg_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, this module
raises.  PyTorch is used only for device memory and streams (tensor.data_ptr(), current stream).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MAS_LIB: another build of the same ABI (A/B runs of a kernel change inside one GPU session); the default is the in-tree library
LIB_PATH = os.environ.get("MAS_LIB") or os.path.join(_HERE, "libmulactseg_hip.so")

ID_I64, ID_I32, ID_U16 = 0, 1, 2
MAX_CLASSES = 32
SCORE_FRAC, PROB_FRAC, LOSS_FRAC = 40, 23, 32
LOSS_CE, LOSS_GROUP, LOSS_GROUP_ONLY_MULTI, LOSS_DECOMP, LOSS_TCE = 1, 2, 4, 8, 16
ACC_WORDS = 8
GRAD_FRAC = 44

SK_DMA, SK_NOSPLIT = 1, 2
LOWRES_GENERIC = 1
ABI_VERSION = 7        # MAS_ABI_VERSION of include/mulactseg_hip.h this table was written against (load() refuses any other library)

_c = ctypes
_vp, _i, _f, _i64, _d = _c.c_void_p, _c.c_int, _c.c_float, _c.c_int64, _c.c_double


class SkOpts(ctypes.Structure):
    """mas_sk_opts of include/mulactseg_hip.h: the per-call options of the stream-K convolution."""
    _fields_ = [("flags", _c.c_uint), ("spin_limit", _c.c_uint), ("stamps", _c.c_void_p)]


# name -> (restype, argtypes); mirrors include/mulactseg_hip.h one to one
SIGNATURES = {
    "mas_abi_version": (_i, []),
    "mas_error_string": (_c.c_char_p, [_i]),
    "mas_class_prob_sum": (_i, [_vp, _i, _i, _i, _i, _f, _vp, _vp]),
    "mas_bvsb_region_accum": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp]),
    "mas_region_finalize": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "mas_region_keys": (_i, [_vp, _vp, _vp, _i64, _i, _vp, _vp]),
    "mas_select_workspace_bytes": (_c.c_size_t, [_i64]),
    "mas_sort_keys_desc": (_i, [_vp, _i64, _vp, _vp, _c.c_size_t, _vp]),
    "mas_budget_walk": (_i, [_vp, _i64, _vp, _vp, _i, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _c.c_size_t, _vp]),
    "mas_region_reweight": (_i, [_vp, _vp, _i64, _i, _vp, _vp]),
    "mas_dominant_hist": (_i, [_vp, _i64, _i, _vp, _vp]),
    "mas_minmax_normalize": (_i, [_vp, _i64, _vp, _vp]),
    "mas_iou_counts": (_i, [_vp, _vp, _vp, _i64, _i, _i64, _vp, _vp]),
    "mas_logits_iou_counts": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i64, _vp, _vp]),
    "mas_single_pass_accum": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "mas_single_pass_accum_lowres": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "mas_single_pass_accum_lowres_opt": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _c.c_uint, _vp]),
    "mas_class_weight": (_i, [_vp, _i, _i, _i64, _i, _i, _d, _vp, _vp, _vp, _vp]),
    "mas_region_finalize_weighted": (_i, [_vp, _vp, _i64, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "mas_stage2_gather_protos": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "mas_stage2_assign": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "mas_stage2_adjacency": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "mas_stage2_propagate": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mas_aspp_dw3_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "mas_aspp_dw3_bwd_x": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "mas_aspp_dw3_bwd_w": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "mas_train_augment": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp,
                               _vp, _i, _i64, _vp, _i, _vp, _i, _i64, _vp, _i, _vp, _vp]),
    "mas_upsample_bilinear_fwd": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _vp]),
    "mas_upsample_bilinear_bwd": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _vp]),
    "mas_bn_workspace_bytes": (_i64, [_i, _i, _i]),
    "mas_bn_mask_bytes": (_i64, [_i, _i, _i]),
    "mas_bn_act_train_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mas_bn_act_train_fwd_stats": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mas_bn_act_eval_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp, _vp]),
    "mas_bn_act_train_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mas_cosine_head_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp]),
    "mas_cosine_head_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "mas_dense_small_fwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "mas_dense_small_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "mas_maxpool3s2_fwd": (_i, [_vp, _i64, _i, _i, _vp, _vp, _vp]),
    "mas_maxpool3s2_bwd": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp]),
    "mas_conv1x1_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "mas_stem_conv_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "mas_conv_chunk": (_i, [_i, _i]),
    "mas_conv_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "mas_conv_bx_supported": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "mas_conv_bx_packed_bytes": (_i64, [_i, _i, _i, _i]),
    "mas_conv_bx_pack": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "mas_conv_bx_fwd_dual": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "mas_conv_bx_pack_job_bytes": (_c.c_size_t, []),
    "mas_conv_bx_pack_job": (_c.c_uint, [_vp, _vp, _i, _i, _i, _i, _vp, _c.c_uint]),
    "mas_conv_bx_pack_multi": (_i, [_vp, _i, _c.c_uint, _vp]),
    "mas_conv_bx_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "mas_bx3_bytes": (_c.c_longlong, [_i, _i, _i, _i]),
    "mas_bx3_split": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "mas_conv_bx_fwd_pre": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "mas_conv_bx_train_plan": (_i, [_i, _i, _i, _i, _i, _i, _i, _vp]),
    "mas_conv_bx_train_workspace_bytes": (_c.c_size_t, [_i, _i, _i, _i, _i]),
    "mas_conv_bx_train": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp, _c.c_size_t, _vp, _vp]),
    "mas_conv_bx_train_stat_slots": (_i, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "mas_conv_wgrad_bx_supported": (_i, [_i, _i, _i, _i, _i]),
    "mas_conv_wgrad_bx_workspace_bytes": (_c.c_size_t, [_i, _i]),
    "mas_conv_wgrad_bx": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _c.c_size_t, _vp]),
    "mas_conv_wgrad_bx3_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "mas_conv_wgrad_bx3_workspace_bytes": (_c.c_size_t, [_i, _i]),
    "mas_conv_wgrad_bx3": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _c.c_size_t, _vp]),
    "mas_conv_sk_workspace_bytes": (_c.c_size_t, []),
    "mas_conv_sk_packed_elems": (_c.c_size_t, [_i, _i, _i, _i, _i]),
    "mas_conv_sk_pack": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "mas_conv_sk_pack_job_bytes": (_c.c_size_t, []),
    "mas_conv_sk_pack_job": (_c.c_uint, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _c.c_uint]),
    "mas_conv_sk_pack_multi": (_i, [_vp, _i, _c.c_uint, _vp]),
    "mas_conv_sk": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _c.c_size_t, _c.c_uint, _vp, _vp]),
    "mas_conv_sk_error": (_i, [_vp, _vp]),
    "mas_conv_sk_stats_slots": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _c.c_uint]),
    "mas_conv_sk_stats": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _c.c_size_t, _c.c_uint, _vp, _vp]),
    "mas_conv_sk_dgrad_s2": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _c.c_size_t, _c.c_uint, _vp, _vp]),
    "mas_conv_wgrad_workspace_bytes": (_c.c_size_t, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "mas_conv_wgrad_plan": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "mas_conv_wgrad": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _c.c_size_t, _vp]),
    "mas_depthwise3x3_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "mas_depthwise3x3_bwd_x": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "mas_depthwise3x3_bwd_w": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "mas_target_bits": (_i, [_vp, _i64, _i, _i, _vp, _vp]),
    "mas_partial_loss_fwd": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp]),
    "mas_group_finalize": (_i, [_vp, _i64, _vp, _vp]),
    "mas_loss_values": (_i, [_vp, _i, _vp, _vp]),
    "mas_loss_scales": (_i, [_vp, _vp, _i, _vp, _vp]),
    "mas_partial_loss_bwd": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _vp]),
    "mas_partial_loss_fwd_lowres": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp]),
    "mas_partial_loss_bwd_lowres": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _vp]),
    "mas_loss_values_weighted": (_i, [_vp, _i, _vp, _vp, _vp]),
    "mas_loss_scales_weighted": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "mas_fix_to_float": (_i, [_vp, _i64, _i, _vp, _vp]),
    "mas_partial_loss_work_bytes": (_c.c_size_t, [_i, _i, _i, _i]),
    "mas_partial_loss_fwd_fused": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _c.c_size_t, _vp, _vp]),
    "mas_adamw_job_bytes": (_c.c_size_t, []),
    "mas_adamw_max_groups": (_i, []),
    "mas_adamw_job": (_c.c_uint, [_vp, _vp, _vp, _vp, _vp, _c.c_longlong, _i, _c.c_uint]),
    "mas_adamw_multi": (_i, [_vp, _i, _c.c_uint, _vp, _i, _d, _d, _d, _d, _vp, _vp, _vp]),
    "mas_partial_loss_bwd_fused": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp]),
}

_lib = None


class MulActSegHipError(RuntimeError):
    pass


def load():
    """Load the HIP library once; raise loudly when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own HIP runtime (same SONAME as /opt/rocm's).  It must be the one that is
    # resident before this library is mapped, otherwise two runtimes end up in the process and the second
    # one sees no device ("no ROCm-capable device is detected").
    import torch  # noqa: F401
    # a variant build (tools/build_variant.sh, MAS_LIB) lives under build/, never beside the product library: a stale A/B library inside
    # the package directory would travel with every snapshot and be one environment variable away from being the product
    if os.environ.get("MAS_LIB"):
        real = os.path.realpath(LIB_PATH)
        if real.startswith(os.path.realpath(_HERE) + os.sep) and real != os.path.realpath(os.path.join(_HERE, "libmulactseg_hip.so")):
            raise MulActSegHipError("MAS_LIB=%s points inside the package directory; variant builds belong under build/ "
                                    "(tools/build_variant.sh)" % LIB_PATH)
    if not os.path.exists(LIB_PATH):
        raise MulActSegHipError(
            "libmulactseg_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C mulactseg_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    # the version first: a library built from another header (a stale variant named by MAS_LIB, tools/build_variant.sh) may export
    # every symbol of the table with OTHER argument lists -- ctypes would bind them without complaint
    try:
        lib.mas_abi_version.restype = ctypes.c_int
        lib.mas_abi_version.argtypes = []
        got = int(lib.mas_abi_version())
    except AttributeError:
        got = None
    if got != ABI_VERSION:
        raise MulActSegHipError("%s reports ABI version %s, this package binds version %d (include/mulactseg_hip.h MAS_ABI_VERSION): "
                                "rebuild it with `make -C mulactseg_amd/csrc`" % (LIB_PATH, got, ABI_VERSION))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().mas_error_string(code)
        raise MulActSegHipError("%s failed with code %d: %s" % (what, code, msg.decode() if msg else "?"))
