"""ctypes binding of libvdm4cdm_hip.so (C-ABI: include/vdm4cdm_hip.h).

The library is the product's only compute path on the GPU.  There is NO fallback: if the shared
object is missing or a symbol is absent, ``lib()`` raises.  ``build()`` compiles it in-tree with
hipcc for gfx950 (vdm4cdm_amd/csrc/Makefile).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# VDM4CDM_LIB: load another build of the same library (tools/conv_timeline.py uses the stamped diagnostic build)
# VDM4CDM_FP32_EXACT=1: fp32 storage computes with the exact fp32 MFMA (1/16 of the bf16 rate) instead of three bf16 MFMAs per product
FP32_EXACT = os.environ.get("VDM4CDM_FP32_EXACT", "0") == "1"
LIB_PATH = os.environ.get("VDM4CDM_LIB") or os.path.join(_HERE, "libvdm4cdm_hip_fp32exact.so" if FP32_EXACT else "libvdm4cdm_hip.so")
CSRC = os.path.join(_HERE, "csrc")

VDM_F32, VDM_BF16 = 0, 1
PAD_ZEROS, PAD_CIRCULAR = 0, 1
PACK_FWD, PACK_DGRAD = 0, 1
GN_STATS_WS_BYTES = 2048 * 2 * 64 * 4


class ConvDesc(C.Structure):
    """vdm_conv_desc (include/vdm4cdm_hip.h)."""
    _fields_ = [(k, C.c_int32) for k in
                ("n", "od", "oh", "ow", "cin", "cout", "ksize", "stride", "upsample", "pad_mode", "dtype", "out_f32")]


class PackItem(C.Structure):
    """vdm_pack_item"""
    _fields_ = [("w_master", C.c_void_p), ("w_packed", C.c_void_p)] + \
               [(k, C.c_int32) for k in ("taps", "cout", "cin", "nc", "nchunks", "nkb", "dgrad", "variant", "cls_kind", "dtype")] + \
               [("elems", C.c_int64)]


class PackChunk(C.Structure):
    """vdm_pack_chunk"""
    _fields_ = [("item", C.c_int32), ("count", C.c_int32), ("first", C.c_int64)]


class GnFold(C.Structure):
    """vdm_gn_fold"""
    _fields_ = [("x1", C.c_void_p), ("x2", C.c_void_p), ("c1", C.c_int32), ("c2", C.c_int32), ("groups", C.c_int32),
                ("stats", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("eps", C.c_float), ("inv_keep", C.c_float),
                ("keep_mask", C.c_void_p), ("partials", C.c_void_p)]


class CondMlp(C.Structure):
    """vdm_cond_mlp"""
    _fields_ = [("input", C.c_void_p), ("in_dim", C.c_int32), ("dim", C.c_int32), ("sinusoid", C.c_int32), ("reserved", C.c_int32),
                ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p), ("wproj", C.c_void_p),
                ("dw1", C.c_void_p), ("db1", C.c_void_p), ("dw2", C.c_void_p), ("db2", C.c_void_p), ("dwproj", C.c_void_p)]


class AugmentChannel(C.Structure):
    """vdm_augment_channel"""
    _fields_ = [("field", C.c_void_p), ("out", C.c_void_p), ("alpha", C.c_float), ("mean", C.c_float), ("std", C.c_float)]


class AugmentSample(C.Structure):
    """vdm_augment_sample"""
    _fields_ = [("sim", C.c_int32), ("anchor", C.c_int32 * 3), ("flip", C.c_int32 * 3), ("perm", C.c_int32 * 3)]


PACK_CHUNK = 16384                   # VDM_PACK_CHUNK
_p, _i, _i64, _u64, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float, C.c_size_t
_D = C.POINTER(ConvDesc)
ABI_VERSION = 11                      # VDM_ABI_VERSION of include/vdm4cdm_hip.h this binding was written for

# name -> (restype, argtypes); mirrors include/vdm4cdm_hip.h one to one
SIGNATURES = {
    "vdm_last_error": (C.c_char_p, []),
    "vdm_abi_version": (_i, []),
    "vdm_device_info": (_i, [_i, C.POINTER(_i), C.POINTER(_i), C.c_char_p]),
    "vdm_conv_packed_bytes": (_sz, [_D, _i]),
    "vdm_conv_pack_weights": (_i, [_D, _i, _p, _p, _p]),
    "vdm_conv_pack_plan": (_i, [_D, _i, _p, _p, C.POINTER(PackItem)]),
    "vdm_conv_pack_many": (_i, [_p, _p, _i, _i, _p]),
    "vdm_conv_gn_tiles": (_i, [_D]),
    "vdm_conv_fwd": (_i, [_D, _p, _p, _p, _p, _i64, _p, _p, _p, _p]),
    "vdm_conv_fwd_gn_supported": (_i, [_D]),
    "vdm_conv_fwd_gn": (_i, [_D, _p, _p, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _i, _f, _p]),
    "vdm_conv_dgrad": (_i, [_D, _p, _p, _p, _p, _p]),
    "vdm_conv_dgrad_gn_tiles": (_i, [_D]),
    "vdm_conv_dgrad_gn": (_i, [_D, _p, _p, _p, C.POINTER(GnFold), _p]),
    "vdm_conv_kernel_variant": (_i, [_D, _i]),
    "vdm_conv_dgw_supported": (_i, [_D]),
    "vdm_conv_dgw_tiles": (_i, [_D]),
    "vdm_conv_dgw_workspace_bytes": (_sz, [_D]),
    "vdm_conv_dgrad_gn_wgrad": (_i, [_D, _p, _p, _p, _p, C.POINTER(GnFold), _p, _p, _i, _p, _sz, _p]),
    "vdm_conv_wgrad_workspace_bytes": (_sz, [_D]),
    "vdm_conv_wgrad": (_i, [_D, _p, _p, _p, _p, _i, _p, _sz, _p]),
    "vdm_gn_stats": (_i, [_p, _i, _p, _i, _i, _i64, _i, _i, _p, _p, _p, _i, _p, _i, _p, _p]),
    "vdm_gn_silu_fwd": (_i, [_p, _i, _p, _i, _i, _i64, _i, _i, _p, _p, _p, _f, _f, _u64, _p, _p, _i, _p, _p]),
    "vdm_gn_dyh": (_i, [_p, _i, _p, _i, _i, _i64, _i, _i, _p, _p, _p, _f, _f, _u64, _p, _p, _i, _p, _p]),
    "vdm_gn_bwd_finalize": (_i, [_p, _i, _i, _i, _i, _i64, _p, _p, _f, _p, _p, _p, _p, _i64, _p]),
    "vdm_gn_bwd_apply": (_i, [_p, _i, _p, _i, _i, _i64, _i, _i, _p, _p, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "vdm_gn_bwd_apply_wgrad_thin": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p, _f, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _sz, _p]),
    "vdm_gn_skip_supported": (_i, [_i, _i, _i, _i]),
    "vdm_gn_skip_ws_floats": (_sz, [_i, _i, _i, _i, _i64]),
    "vdm_gn_silu_skip_fwd": (_i, [_p, _i, _p, _i, _i, _i64, _i, _i, _p, _p, _p, _f, _p, _p, _p, _i, _p, _p, _p]),
    "vdm_gn_bwd_apply_skip": (_i, [_p, _i, _p, _i, _i, _i64, _i, _i, _p, _p, _f, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "vdm_pack_input": (_i, [_p, _p, _i64, _i, _i, _p, _p]),
    "vdm_cond_saved_floats": (_sz, [C.POINTER(CondMlp), _i, _i]),
    "vdm_cond_bwd_scratch_floats": (_sz, [C.POINTER(CondMlp), _i, _i, _i]),
    "vdm_cond_table_fwd": (_i, [C.POINTER(CondMlp), _i, _i, _i, _p, _p, _p]),
    "vdm_cond_table_bwd": (_i, [C.POINTER(CondMlp), _i, _i, _i, _p, _i64, _p, _p, _p, _p]),
    "vdm_cond_table_step": (_i, [_p, _p, _p, _i, _i, _p, _p]),
    "vdm_augment_batch": (_i, [C.POINTER(AugmentChannel), _i, _i, _i, C.POINTER(AugmentSample), _i, _p]),
    "vdm_attn_split_heads": (_i, [_p, _i64, _i64, _i, _i64, _i, _i, _i, _p, _p, _p]),
    "vdm_attn_fwd": (_i, [_p, _p, _p, _i, _i64, _i, _i, _i, _f, _p, _p, _p]),
    "vdm_attn_rowdot": (_i, [_p, _p, _i, _i64, _i, _i, _i, _p, _p]),
    "vdm_attn_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i64, _i, _i, _i, _f, _p, _p]),
    "vdm_channel_sums": (_i, [_p, _i64, _i, _i, _p, _p]),
    "vdm_channel_dot_sums": (_i, [_p, _p, _i, _p, _i, _i, _i64, _i, _p, _p]),
    "vdm_diffuse": (_i, [_p, _p, _p, _p, _i, _i64, _p, _p]),
    "vdm_loss_terms": (_i, [_p, _p, _p, _p, _f, _p, _i, _i64, _p, _p, _p, _p]),
    "vdm_diffuse_pack": (_i, [_p, _p, _p, _u64, _u64, _p, _p, _p, _i, _i64, _i, _p, _p, _p]),
    "vdm_loss_terms_rng": (_i, [_p, _p, _u64, _u64, _p, _p, _u64, _u64, _p, _f, _p, _i, _i64, _p, _p, _p, _p]),
    "vdm_ancestral_step": (_i, [_p, _p, _p, _p, _p, _u64, _i64, _p]),
    "vdm_ancestral_step_cfg": (_i, [_p, _p, _p, _f, _p, _p, _p, _u64, _i64, _p]),
    "vdm_randn": (_i, [_p, _i64, _u64, _u64, _p, _p]),
    "vdm_step_inc": (_i, [_p, _p]),
    "vdm_sumsq": (_i, [_p, _i64, _p, _p, _p]),
    "vdm_train_scalars": (_i, [_p, _p, _i, _i, _i, _f, _f, _f, _p, _u64, _p, _p]),
    "vdm_elbo_assemble": (_i, [_p, _p, _i, _f, _f, _f, _f, _p, _p]),
    "vdm_clip_scale": (_i, [_p, _i64, _p, _f, _p]),
}


class VdmError(RuntimeError):
    pass


_lib = None


def build(force=False):
    """Compile libvdm4cdm_hip.so in-tree with hipcc --offload-arch=gfx950."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"])
    subprocess.check_call(["make", "-C", CSRC, "-j4"])
    return LIB_PATH


def lib():
    """Load the shared object (once) and type its entry points.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VdmError(
            f"{LIB_PATH} is missing: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C vdm4cdm_amd/csrc`). There is no CPU fallback for the HIP backend.")
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    ver = handle.vdm_abi_version()
    if ver != ABI_VERSION:
        raise VdmError(f"libvdm4cdm_hip.so ABI version {ver} != {ABI_VERSION}")
    _lib = handle
    return _lib


def check(status, what=""):
    if status != 0:
        msg = lib().vdm_last_error().decode("utf-8", "replace")
        raise VdmError(f"{what} failed with status {status}: {msg}")
