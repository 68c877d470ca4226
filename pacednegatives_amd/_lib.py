"""ctypes binding of libpaced_hip.so (the C ABI declared in include/paced_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (plain ``hipcc -shared``).  There is no
CPU fallback: if the shared object is missing or a symbol cannot be resolved, importing the product
path raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PACED_HIP_LIB selects another build of the same library (A/B runs of two builds in one job)
LIB_PATH = os.environ.get("PACED_HIP_LIB") or os.path.join(_HERE, "lib", "libpaced_hip.so")

_vp, _i, _f, _u32, _sz, _ll = C.c_void_p, C.c_int, C.c_float, C.c_uint32, C.c_size_t, C.c_longlong

class GemmOpts(C.Structure):
    """``struct paced_gemm_opts`` of include/paced_hip.h (a negative input field = the process default)."""
    _fields_ = [("grid_limit", _i), ("tile", _i), ("ring", _i), ("kernel", _i)]


# paced_abi_version() this binding was written against: argument lists changed between versions, so a library of another
# version that happens to export every symbol would be called with shifted arguments
EXPECTED_ABI = 4

# name -> (restype, argtypes); must list every symbol include/paced_hip.h declares
SIGNATURES = {
    "paced_abi_version": (_i, []),
    "paced_last_error": (C.c_char_p, []),
    "paced_gemm_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _u32, _i, _vp]),
    "paced_gemm_bf16_heads": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _u32, _i, _ll, _ll, _ll, _vp]),
    "paced_gemm_bf16_batched": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _ll, _ll, _ll, _vp]),
    "paced_gemm_bf16_ex": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _u32, _i, _ll, _ll, _ll,
                                _i, _ll, _ll, _ll, C.POINTER(GemmOpts), _vp]),
    "paced_gemm_set_tile": (_i, [_i]),
    "paced_gemm_set_ring": (_i, [_i]),
    "paced_gemm_set_grid": (_i, [_i]),
    "paced_gemm_get_grid": (_i, []),
    "paced_gemm_last_kernel": (_i, []),
    "paced_gemm_tile_bits_bytes": (_ll, [_i, _i]),
    "paced_rmsnorm_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _u32, _vp]),
    "paced_rmsnorm_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _u32, _f, _u32, _vp]),
    "paced_embedding_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _u32, _vp]),
    "paced_embedding_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _u32, _vp]),
    "paced_dropout_apply": (_i, [_vp, _vp, _sz, _f, _u32, _vp]),
    "paced_dropout_mask": (_i, [_vp, _sz, _f, _u32, _vp]),
    "paced_cast_f32_bf16": (_i, [_vp, _vp, _sz, _vp]),
    "paced_relbias_table_fwd": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "paced_relbias_table_bwd": (_i, [_vp, _i, _vp, _vp, _i, _i, _vp]),
    "paced_enc_attn_fwd": (_i, [_vp, _vp, _vp, _i, _ll, _vp, _i, _ll, _vp, _vp, _i, _i, _i, _i, _f, _u32, _vp]),
    "paced_enc_attn_bwd": (_i, [_vp, _vp, _vp, _i, _ll, _vp, _i, _ll, _vp, _i, _ll, _vp, _vp, _vp, _vp, _vp, _i, _ll, _vp, _i, _i, _i, _i, _i, _f, _u32, _vp]),
    "paced_dec_self_attn_fwd": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _f, _u32, _vp]),
    "paced_dec_self_attn_bwd": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _f, _u32, _vp]),
    "paced_cross_attn_fwd": (_i, [_vp, _i, _vp, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _f, _u32, _vp]),
    "paced_cross2_fwd": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _f, _u32, _vp]),
    "paced_cross2_bwd": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _vp, _ll, _i, _i, _i, _i, _f, _u32, _vp]),
    "paced_cross_attn_bwd": (_i, [_vp, _i, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _u32, _vp]),
    "paced_cross_entropy": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _ll, _f, _vp]),
    "paced_lce_group": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "paced_lce_weights": (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    "paced_eta_weights": (_i, [_vp, _vp, _f, _i, _vp, _vp, _i, _vp]),
    "paced_eta_bound_meta": (_i, [_vp, _vp, _vp, _f, _vp, _vp, _i, _vp]),
    "paced_adamw": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _f, _i, _vp]),
    "paced_pool_topk_workspace_bytes": (_sz, [_i, _i]),
    "paced_pool_topk": (_i, [_vp, _vp, _i, _ll, _i, _i, _ll, _vp, _vp, _vp, _sz, _vp]),
    "paced_pool_topk_static_bytes": (_sz, []),
    "paced_pool_topk_static_max_queries": (_i, []),
    "paced_pool_topk_static_max_k": (_i, []),
    "paced_pool_topk_static": (_i, [_vp, _vp, _i, _ll, _i, _i, _ll, _vp, _vp, _vp]),
    "paced_topk_merge": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
}


class PacedHipError(RuntimeError):
    pass


_lib = None


def load():
    """Load the shared library once and bind every symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PacedHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the lceT5 hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    got = lib.paced_abi_version()
    if got != EXPECTED_ABI:
        raise PacedHipError(f"{LIB_PATH} has ABI version {got}, this binding expects {EXPECTED_ABI}: rebuild the library "
                            "(python __graft_entry__.py) or point PACED_HIP_LIB at a build of this source tree")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().paced_last_error().decode("utf-8", "replace")
        raise PacedHipError(f"{what or 'paced_hip call'} failed ({rc}): {msg}")
