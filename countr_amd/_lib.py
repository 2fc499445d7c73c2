"""ctypes binding of libcountr_hip.so (C ABI declared in include/countr_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails, this raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("COUNTR_LIB", os.path.join(_HERE, "libcountr_hip.so"))
# the same sources built with IEEE fp16 as the 16-bit storage / matrix-operand type (precision="fp16"; csrc/common.hpp, build.py)
LIB_PATH_F16 = os.environ.get("COUNTR_LIB_F16", os.path.join(_HERE, "libcountr_hip_f16.so"))

F32, BF16 = 0, 1
ABI_VERSION = 9
OP_ROW, OP_COL, OP_IM2ROW, OP_IM2COL = 0, 1, 2, 3
ACT_NONE, ACT_GELU, ACT_GELU_BWD = 0, 1, 2


class CountrError(RuntimeError):
    pass


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("C2", C.c_void_p),
        ("bias", C.c_void_p), ("resid", C.c_void_p), ("partial", C.c_void_p),
        ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64), ("ldres", C.c_int64),
        ("sA0", C.c_int64), ("sA1", C.c_int64), ("sB0", C.c_int64), ("sB1", C.c_int64),
        ("sC0", C.c_int64), ("sC1", C.c_int64),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("res_mod", C.c_int32), ("act", C.c_int32), ("out_bf16", C.c_int32),
        ("nbatch", C.c_int32), ("nb1", C.c_int32), ("splitk", C.c_int32),
        ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
        ("alpha", C.c_float),
        ("rowsum_partial", C.c_void_p),
        ("ln_xcopy", C.c_void_p), ("ln_stats_out", C.c_void_p), ("ln_stats", C.c_void_p), ("ln_colsum", C.c_void_p),
        ("ln_nblk", C.c_int32), ("ln_eps", C.c_float),
        ("rowsum_slabs", C.c_int32),
        ("prefetch", C.c_void_p), ("prefetch_bytes", C.c_int64),
        ("gn_rows", C.c_void_p),
    ]


_libs = {}


def lib(variant=""):
    """Load (once) and return the ctypes handle of a library variant ("" = bf16 build, "f16" = fp16 build); raises CountrError if it
    is not built."""
    L = _libs.get(variant)
    if L is None:
        path = LIB_PATH_F16 if variant == "f16" else LIB_PATH
        if not os.path.exists(path):
            raise CountrError(
                "%s is not built (run `python -m countr_amd.build` or __graft_entry__.build()); "
                "the HIP path has no CPU fallback" % os.path.basename(path))
        L = C.CDLL(path)
        _declare(L)
        if L.countr_version() != ABI_VERSION:      # a stale build: its countr_gemm_args is shorter than GemmArgs above
            raise CountrError("%s has ABI version %d, this package needs %d: rebuild (python -m countr_amd.build)"
                              % (os.path.basename(path), L.countr_version(), ABI_VERSION))
        _libs[variant] = L
    return L


def variant_of(precision):
    return "f16" if precision == "fp16" else ""


def _declare(L):
    vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
    L.countr_last_error.restype = C.c_char_p
    L.countr_init.argtypes = [i32]
    L.countr_version.argtypes = []
    L.countr_gemm.argtypes = [C.POINTER(GemmArgs), i32, i32, i32, vp]
    L.countr_gemm_rowsum_slabs.argtypes = [C.POINTER(GemmArgs), i32, i32, i32]
    L.countr_gemm_tiles.argtypes = [C.POINTER(GemmArgs), i32, i32, i32]
    L.countr_gemm_gn_rows.argtypes = [C.POINTER(GemmArgs), i32, i32, i32]
    L.countr_gemm_group.argtypes = [C.POINTER(GemmArgs), i32, i32, i32, i32, vp]
    L.countr_gemm_group_tiles.argtypes = [C.POINTER(GemmArgs), i32, i32, i32, i32]
    L.countr_splitk_reduce.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]
    L.countr_reduce_table.argtypes = [vp, i32, i32, vp]
    for name, sig in _SIGS.items():
        fn = getattr(L, name)  # AttributeError here means the .so is stale: rebuild
        fn.argtypes = sig
        fn.restype = _RESTYPES.get(name, C.c_int)


_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float
_SIGS = {
    "countr_layernorm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp],
    "countr_layernorm_bwd_nblocks": [],
    "countr_layernorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "countr_colsum_partials": [_vp, _vp, _i, _i, _i, _vp],
    "countr_groupnorm_nsplit": [_i],
    "countr_groupnorm_bwd_image_sums_offset": [_i, _i],
    "countr_groupnorm_relu_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp],
    "countr_groupnorm_relu_fwd_rows": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp],
    "countr_groupnorm_relu_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "countr_instnorm_workspace_floats": [_i, _i],
    "countr_instnorm_relu_pool_fwd": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _i, _vp],
    "countr_instnorm_relu_pool_bwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp],
    "countr_attn_fwd": [_vp, _vp, _vp, _i, _i, _i, _i, _f, _vp],
    "countr_attn_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp],
    "countr_softmax_fwd": [_vp, _vp, _i64, _i, _i, _vp],
    "countr_softmax_fwd_ld": [_vp, _vp, _i64, _i, _i, _i, _vp],
    "countr_softmax_bwd": [_vp, _vp, _vp, _i64, _i, _f, _i, _vp],
    "countr_xattn_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp],
    "countr_xattn_bwd_workspace_floats": [_i, _i, _i, _i],
    "countr_xattn_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp],
    "countr_im2patch": [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    "countr_conv3x3_c3_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "countr_conv3x3_c3_wgrad_nblocks": [],
    "countr_conv3x3_c3_wgrad": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "countr_upsample2x_fwd": [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    "countr_upsample2x_bwd": [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    "countr_gelu_bwd": [_vp, _vp, _vp, _i64, _i, _vp],
    "countr_colsum_nparts": [],
    "countr_colsum": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "countr_cast_permute": [_vp, _vp, _i64, _i, _i, _i, _i, _i, _vp],
    "countr_copy_multi": [_i, _vp, _vp, _vp, _vp],
    "countr_step_prologue_record_bytes": [],
    "countr_step_prologue_copy_blocks": [],
    "countr_step_prologue": [_vp, _i, _vp, _vp, _vp, _i, _vp],
    "countr_gather_rows": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "countr_mae_indices": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "countr_patch_mse_workspace_floats": [_i, _i, _i, _i],
    "countr_patch_mse": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp],
    "countr_conv_shadows": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "countr_transpose16": [_i, _vp, _vp, _vp, _vp, _vp],
    "countr_masked_mse_workspace_floats": [_i],
    "countr_masked_mse": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp],
    "countr_masked_mse_amp": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp],
    "countr_patch_mse_amp": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _vp],
    "countr_adamw_step_amp": [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _i, _f, _vp, _vp, _vp, _vp],
    "countr_adamw_gnorm_floats": [],
    "countr_splitk_finish": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "countr_window_gather": [_vp, _vp, _vp, _i, _i, _vp, _vp],
    "countr_window_blend": [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp],
    "countr_window_blend_blocks": [_i, _i],
    "countr_adamw_step": [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _i, _f, _vp, _vp, _vp],
}
_RESTYPES = {"countr_xattn_bwd_workspace_floats": C.c_int64, "countr_groupnorm_bwd_image_sums_offset": C.c_int64}


def check(rc, what=""):
    if rc != 0:
        msgs = [m.decode() for m in (L.countr_last_error() for L in _libs.values()) if m]      # (the error text is per library and thread)
        raise CountrError("%s failed (rc=%d): %s" % (what or "countr call", rc, " | ".join(msgs)))


def exported_symbols():
    """Names declared in include/countr_hip.h (parsed), used by the CPU-side ABI test."""
    import re
    hdr = os.path.join(_HERE, "..", "include", "countr_hip.h")
    txt = open(hdr).read()
    return sorted(set(re.findall(r"\b(countr_[a-z0-9_]+)\s*\(", txt)))
