"""ctypes binding of libneko_hip.so (include/neko_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, this module
raises.  Build with ``python -m neko_amd.build`` (or ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NEKO_HIP_LIB") or os.path.join(_HERE, "csrc", "libneko_hip.so")   # override: kernel A/B builds

_vp, _i, _l, _f = C.c_void_p, C.c_int, C.c_long, C.c_float

#: name -> argtypes, exactly the prototypes of include/neko_hip.h
SIGNATURES = {
    "neko_gemm_bf16": [_vp, _l, _i, _vp, _l, _i, _i, _i, _i, _f, _vp, _vp, _vp, _l, _i, _vp, _l, _vp, _l,
                       _vp, _l, _i, _vp, _l, _i, _i, _vp, _i, C.c_uint, _f, _i, _vp],
    "neko_gemm_colsum_ws_floats": [_i, _i],
    "neko_gemm_dgrad_gelu_colsum": [_vp, _l, _vp, _l, _i, _i, _i, _vp, _l, _i, _vp, _l, _vp, _vp, _vp],
    "neko_dropout_f32": [_vp, _vp, _l, _i, C.c_uint, _f, _vp],
    "neko_gather_rows_bf16": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "neko_scatter_rows_f32": [_vp, _vp, _vp, _i, _i, _vp],
    "neko_layernorm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp],
    "neko_layernorm_bwd_blocks": [_i],
    "neko_layernorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, C.c_uint, _f, _vp, _vp],
    "neko_layernorm_bwd_rows": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, C.c_uint, _f, _vp, _vp],
    "neko_layernorm_bwd_bf16dy": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, C.c_uint, _f, _vp, _vp],
    "neko_mask_bias": [_vp, _vp, _vp, _i, _i, _vp],
    "neko_attn_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, C.c_uint, _f, _vp, _vp],
    "neko_attn_mask_dwords": [_i, _i, _i, _i],
    "neko_attn_set_path": [_i],
    "neko_attn_bwd_reproducible": [_i],
    "neko_gemm_set_mainloop": [_i],
    "neko_gemm_last_mainloop": [],
    "neko_attn_varlen_supported": [_i, _i],
    "neko_attn_fwd_varlen": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, C.c_uint, _f, _vp, _vp],
    "neko_attn_bwd_varlen": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _l, _i, _i, _i, _i, C.c_uint, _f, _vp, _vp],
    "neko_attn_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, C.c_uint, _f, _vp, _vp],
    "neko_gemv_bf16": [_vp, _l, _vp, _l, _i, _i, _i, _i, _vp, _vp, _l, _i, _vp, _l, _vp, _l, _vp],
    "neko_attn_decode": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "neko_ce_fwd_bwd": [_vp, _l, _i, _i, _vp, _vp, _vp, _vp, _l, _i, _vp],
    "neko_ce_bf16_inplace": [_vp, _l, _i, _i, _vp, _vp, _vp, _i, _i, _vp],
    "neko_pack_embed_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _i, _i, _i, _vp],
    "neko_pack_embed_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "neko_pack_embed_bwd_det_ws_bytes": [_i, _i],
    "neko_pack_embed_bwd_det": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _l, _vp],
    "neko_patch_pos_add_bwd_det_ws_bytes": [_i, _i],
    "neko_patch_pos_add_bwd_det": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _l, _vp],
    "neko_tokenize_continuous": [_vp, _vp, _l, _i, _f, _f, _i, _i, _vp],
    "neko_cast_f32_bf16": [_vp, _vp, _l, _vp],
    "neko_colsum_bf16": [_vp, _l, _i, _i, _vp, _i, _vp],
    "neko_sqnorm_f32": [_vp, _l, _vp, _vp],
    "neko_geglu_fwd": [_vp, _vp, _l, _vp],
    "neko_geglu_bwd": [_vp, _vp, _vp, _vp, _vp, _l, _vp],
    "neko_adamw_step": [_vp, _vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _f, _vp, _f, _vp, _vp, _vp, _vp, _vp],
    "neko_set_drop_salt": [_vp],
    "neko_patch_resblock_fwd": [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp],
    "neko_patch_resblock_bwd": [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                _vp, _vp],
    "neko_patch_resblock_fwd_stats": [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp],
    "neko_patch_resblock_bwd_stats": [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                      _vp, _vp],
    "neko_patch_resblock_bwd_ws_floats": [_i],
    "neko_pack_embed_bwd_sorted_ws_bytes": [_i, _i],
    "neko_pack_embed_bwd_sorted": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _l, _vp],
    "neko_patch_pos_add_bwd_sorted_ws_bytes": [_i, _i],
    "neko_patch_pos_add_bwd_sorted": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _l, _vp],
    "neko_patch_pos_add": [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "neko_patch_pos_add_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "neko_abi_version": [],
}

_lib = None


class NekoHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libneko_hip.so (once). Raises NekoHipError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64: import it FIRST so libneko_hip.so binds to the same HIP runtime
    # (loading ours first drags in /opt/rocm's copy -> two runtimes in one process, ours sees no device).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise NekoHipError(
            f"{LIB_PATH} not found: the HIP extension is not built (run `python -m neko_amd.build`). "
            "neko_amd has no CPU/PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the library lacks a declared symbol
        fn.argtypes = args
        fn.restype = _i
    lib.neko_attn_mask_dwords.restype = C.c_long
    lib.neko_gemm_colsum_ws_floats.restype = C.c_long
    lib.neko_pack_embed_bwd_det_ws_bytes.restype = C.c_long
    lib.neko_patch_pos_add_bwd_det_ws_bytes.restype = C.c_long
    lib.neko_pack_embed_bwd_sorted_ws_bytes.restype = C.c_long
    lib.neko_patch_pos_add_bwd_sorted_ws_bytes.restype = C.c_long
    lib.neko_status_string.argtypes = [_i]
    lib.neko_status_string.restype = C.c_char_p
    _lib = lib
    return lib


def check(rc: int, name: str) -> None:
    if rc != 0:
        msg = load().neko_status_string(rc).decode()
        raise NekoHipError(f"{name} failed with code {rc}: {msg}")


def call(name: str, *args) -> None:
    check(getattr(load(), name)(*args), name)
