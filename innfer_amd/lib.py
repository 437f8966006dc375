"""ctypes binding of libinnfer_amd.so (the C ABI declared in include/innfer_amd.h).

The library is the product: there is NO fallback.  If the shared object is
missing the import of this module raises; if a call fails the Python side
raises the exception class the reference would have raised for the same
condition (ValueError / NotImplementedError / RuntimeError).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_DEFAULT_LIB = os.path.join(_HERE, "lib", "libinnfer_amd.so")
LIB_PATH = os.environ.get("INNFER_LIB") or _DEFAULT_LIB

F16, F32, U8 = 0, 1, 2
OK, ERR_INVALID, ERR_HIP, ERR_UNSUPPORTED, ERR_NOMEM, ERR_WORKSPACE = 0, -1, -2, -3, -4, -5


class InnferError(RuntimeError):
    pass


def _load():
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "innfer_amd has no CPU or PyTorch fallback for its hot path.")
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.7; it must be the ONE HIP runtime of the
    # process (device pointers, streams and events are shared with torch), so torch is imported
    # first and libinnfer_amd.so's NEEDED libamdhip64.so.7 binds to the copy already loaded.
    import torch  # noqa: F401
    return C.CDLL(LIB_PATH)


_lib = _load()


class ConvArgs(C.Structure):
    _fields_ = [
        ("d_in", C.c_void_p), ("in_group_stride", C.c_int64), ("C", C.c_int),
        ("d_packed", C.c_void_p), ("d_bias", C.c_void_p),
        ("d_out", C.c_void_p), ("out_group_stride", C.c_int64), ("out_ch_off", C.c_int), ("K", C.c_int),
        ("N", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("act", C.c_int), ("upsample2x", C.c_int),
        ("d_res1", C.c_void_p), ("res1_group_stride", C.c_int64), ("res1_scale", C.c_float),
        ("d_res2", C.c_void_p), ("res2_group_stride", C.c_int64), ("res2_scale", C.c_float),
        ("row_begin", C.c_int), ("row_end", C.c_int),
        ("reflect_pad", C.c_int), ("dilation", C.c_int), ("dilation_groups", C.c_int), ("pixel_shuffle2", C.c_int),
        ("stride2_k4", C.c_int), ("transposed2x", C.c_int), ("column7", C.c_int),
        ("split", C.c_int), ("in_lo", C.c_int64), ("out_lo", C.c_int64), ("res1_lo", C.c_int64), ("res2_lo", C.c_int64),
        ("reserved0", C.c_int), ("res1_from_input", C.c_int), ("plane_rows", C.c_int),
    ]


# name -> (restype, argtypes); every symbol declared in include/innfer_amd.h
SIGNATURES = {
    "innfer_version": (C.c_int, []),
    "innfer_last_error": (C.c_char_p, []),
    "innfer_rrdbnet_create": (C.c_int, [C.POINTER(C.c_void_p)] + [C.c_int] * 7),
    "innfer_srresnet_create": (C.c_int, [C.POINTER(C.c_void_p)] + [C.c_int] * 5),
    "innfer_net_destroy": (None, [C.c_void_p]),
    "innfer_net_num_convs": (C.c_int, [C.c_void_p]),
    "innfer_net_conv_info": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t,
                                       C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "innfer_net_set_conv": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "innfer_net_scale": (C.c_int, [C.c_void_p]),
    "innfer_net_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "innfer_net_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "innfer_net_forward_timed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                           C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p,
                                           C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                           C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "innfer_rrdbnet_create_ex": (C.c_int, [C.POINTER(C.c_void_p)] + [C.c_int] * 10),
    "innfer_srresnet_create_ex": (C.c_int, [C.POINTER(C.c_void_p)] + [C.c_int] * 6 + [C.c_float, C.c_int]),
    "innfer_net_set_conv_input_map": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "innfer_net_set_outm": (C.c_int, [C.c_void_p, C.c_int]),
    "innfer_timer_start": (C.c_int, []),
    "innfer_timer_stop": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "innfer_net_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "innfer_pack_conv3x3_split": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "innfer_nchw_to_slab_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64] + [C.c_int] * 5 + [C.c_void_p]),
    "innfer_slab_split_to_nchw": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]),
    "innfer_net_set_band_rows": (C.c_int, [C.c_void_p, C.c_int]),
    "innfer_net_set_fused_tail": (C.c_int, [C.c_void_p, C.c_int]),
    "innfer_net_set_hr_chain": (C.c_int, [C.c_void_p, C.c_int]),
    "innfer_net_set_residual_lds": (C.c_int, [C.c_void_p, C.c_int]),
    "innfer_net_set_upconv_phases": (C.c_int, [C.c_void_p, C.c_int]),
    "innfer_net_set_final_act": (C.c_int, [C.c_void_p, C.c_int]),
    "innfer_net_flops": (C.c_double, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "innfer_unet_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int]),
    "innfer_unet_create_ex": (C.c_int, [C.POINTER(C.c_void_p)] + [C.c_int] * 6),
    "innfer_unet_destroy": (None, [C.c_void_p]),
    "innfer_unet_num_params": (C.c_int, [C.c_void_p]),
    "innfer_unet_param_info": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "innfer_unet_set_param": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "innfer_unet_set_eval": (C.c_int, [C.c_void_p, C.c_int]),
    "innfer_unet_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "innfer_unet_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "innfer_unet_flops": (C.c_double, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "innfer_unet_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "innfer_pan_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "innfer_pan_create_ex": (C.c_int, [C.POINTER(C.c_void_p)] + [C.c_int] * 9),
    "innfer_pan_destroy": (None, [C.c_void_p]),
    "innfer_pan_num_params": (C.c_int, [C.c_void_p]),
    "innfer_pan_param_info": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "innfer_pan_set_param": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "innfer_pan_set_fused_scpa": (C.c_int, [C.c_void_p, C.c_int]),
    "innfer_pan_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "innfer_pan_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "innfer_pan_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "innfer_ppon_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]),
    "innfer_ppon_destroy": (None, [C.c_void_p]),
    "innfer_ppon_num_params": (C.c_int, [C.c_void_p]),
    "innfer_ppon_param_info": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "innfer_ppon_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "innfer_ppon_set_param": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "innfer_ppon_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "innfer_ppon_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "innfer_resnet_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int]),
    "innfer_resnet_create_ex": (C.c_int, [C.POINTER(C.c_void_p)] + [C.c_int] * 8),
    "innfer_resnet_set_eval": (C.c_int, [C.c_void_p, C.c_int]),
    "innfer_resnet_destroy": (None, [C.c_void_p]),
    "innfer_resnet_num_params": (C.c_int, [C.c_void_p]),
    "innfer_resnet_param_info": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "innfer_resnet_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "innfer_resnet_set_param": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "innfer_resnet_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "innfer_resnet_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "innfer_wbc_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    "innfer_wbc_destroy": (None, [C.c_void_p]),
    "innfer_wbc_num_params": (C.c_int, [C.c_void_p]),
    "innfer_wbc_param_info": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "innfer_wbc_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "innfer_wbc_set_param": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "innfer_wbc_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "innfer_wbc_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "innfer_guided_filter_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "innfer_guided_filter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                       C.c_void_p, C.c_size_t, C.c_void_p]),
    "innfer_conv3x3_packed_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "innfer_pack_conv3x3": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "innfer_pack_conv3x3_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "innfer_pack_conv3x3_shuffle2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "innfer_pack_convt2x_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "innfer_conv3x3_f16": (C.c_int, [C.POINTER(ConvArgs), C.c_void_p]),
    "innfer_filter2d": (C.c_int, [C.c_void_p, C.c_int, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "innfer_conv7x1_packed_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "innfer_pack_conv7x1": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "innfer_conv4x4s2_packed_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "innfer_pack_conv4x4s2": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "innfer_convt2x_packed_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "innfer_pack_convt2x": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "innfer_nchw_to_slab": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "innfer_slab_to_nchw": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "innfer_chop_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int),
                                   C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.POINTER(C.c_int)]),
    "innfer_extract_tiles": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                       C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "innfer_blend_profile": (C.c_int, [C.c_int, C.c_double, C.c_int, C.POINTER(C.c_float)]),
    "innfer_recompose": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_double, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "innfer_shard_tiles": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "innfer_comm_unique_id": (C.c_int, [C.c_void_p]),
    "innfer_comm_init": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int]),
    "innfer_comm_destroy": (None, [C.c_void_p]),
    "innfer_comm_rank": (C.c_int, [C.c_void_p]),
    "innfer_comm_size": (C.c_int, [C.c_void_p]),
    "innfer_gather_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "innfer_comm_broadcast": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "innfer_u8hwc_to_nchw": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "innfer_srgb_to_linear": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "innfer_linear_to_srgb": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "innfer_color_fix_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "innfer_color_fix": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_size_t, C.c_void_p]),
    "innfer_nchw_to_u8hwc": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "innfer_linear_resize": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "innfer_extract_tiles_u8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "innfer_recompose_u8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "innfer_net_set_u8_io": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "innfer_guided_filter_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                          C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "innfer_inthwc_to_nchw": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p]),
    "innfer_nchw_to_inthwc": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
}

# INNFER_ABI_ANY=1 (measurement only: scripts/evidence_r5.sh A/Bs an OLDER build of the library against the current one on one box): bind the entry points that
# library has and skip the revision check -- calls into entry points it lacks fail with AttributeError.  Never set for product use.
#  Only honoured together with INNFER_LIB pointing at a NON-default library, and loudly: struct layouts are not checked in this mode.
_ABI_ANY = os.environ.get("INNFER_ABI_ANY") == "1" and bool(os.environ.get("INNFER_LIB")) and os.path.realpath(os.environ["INNFER_LIB"]) != os.path.realpath(_DEFAULT_LIB)
if os.environ.get("INNFER_ABI_ANY") == "1":
    import warnings
    warnings.warn("INNFER_ABI_ANY=1: " + ("binding %s without the ABI revision / symbol check (measurement only; struct layouts are unchecked)" % LIB_PATH if _ABI_ANY
                  else "ignored -- it needs INNFER_LIB to name a library other than the in-tree default"), RuntimeWarning, stacklevel=2)
for _name, (_res, _args) in SIGNATURES.items():
    if _ABI_ANY and not hasattr(_lib, _name):
        continue
    _fn = getattr(_lib, _name)          # AttributeError here = header/library mismatch
    _fn.restype = _res
    _fn.argtypes = _args

lib = _lib

ABI_VERSION = 113          # the header revision this binding was written against (INNFER_ABI_VERSION)
if _lib.innfer_version() != ABI_VERSION and not _ABI_ANY:
    raise ImportError(f"{LIB_PATH} speaks ABI {_lib.innfer_version()}, this binding {ABI_VERSION}: rebuild with `make`")


def last_error():
    return _lib.innfer_last_error().decode(errors="replace")


def check(rc):
    """Map a status code to the exception the reference raises for that condition."""
    if rc == OK:
        return
    msg = last_error()
    if rc == ERR_INVALID:
        raise ValueError(msg)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == ERR_NOMEM:                 # the class torch's allocator raises: parallel.run_tile_batches halves the tile batch on it
        import torch
        raise torch.OutOfMemoryError(msg)
    raise InnferError(f"innfer_amd status {rc}: {msg}")


COMM_ID_BYTES = 128


def shard_tiles(n_tiles, nranks, rank):
    first, count = C.c_int(), C.c_int()
    check(_lib.innfer_shard_tiles(n_tiles, nranks, rank, C.byref(first), C.byref(count)))
    return first.value, count.value


def chop_plan(H, W, patch=200, step=0.5):
    """(ps, ys, xs): host-side tile geometry."""
    ps, nh, nw = C.c_int(), C.c_int(), C.c_int()
    check(_lib.innfer_chop_plan(H, W, patch, step, C.byref(ps), C.byref(nh), C.byref(nw), None, None))
    ys, xs = (C.c_int * nh.value)(), (C.c_int * nw.value)()
    check(_lib.innfer_chop_plan(H, W, patch, step, C.byref(ps), C.byref(nh), C.byref(nw), ys, xs))
    return ps.value, list(ys), list(xs)


def blend_profile(P, step=0.5, scale=1):
    import numpy as np
    buf = (C.c_float * P)()
    check(_lib.innfer_blend_profile(P, step, scale, buf))
    return np.frombuffer(buf, dtype=np.float32).copy()


def timed_launches(fn, stream=None, cap=4096, name_cap=64):
    """fn() under the library's launch timer (innfer_timer_start / innfer_timer_stop): [(kernel family, ms, algorithmic flops, algorithmic bytes)] of
    every instrumented launch fn made on this thread, in issue order.  Measurement only."""
    import torch
    if stream is None:
        stream = torch.cuda.current_stream().cuda_stream
    names = C.create_string_buffer(cap * name_cap)
    ms, fl, by, n = (C.c_float * cap)(), (C.c_double * cap)(), (C.c_double * cap)(), C.c_int()
    check(_lib.innfer_timer_start())
    try:
        fn()
    finally:
        rc = _lib.innfer_timer_stop(stream, cap, names, name_cap, ms, fl, by, C.byref(n))
    check(rc)
    raw = names.raw
    return [(raw[i * name_cap:(i + 1) * name_cap].split(b"\0", 1)[0].decode(), ms[i], fl[i], by[i]) for i in range(min(n.value, cap))]
