"""ctypes binding of libuvtg.so (include/uvtg.h).  There is no fallback: if the library is missing the
import of any compute entry point fails loudly with the build instruction."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UVTG_LIB_PATH") or os.path.join(_HERE, "libuvtg.so")      # override: A/B of two builds (dev)


class Dims(C.Structure):
    """struct uvtg_dims (include/uvtg.h)."""
    _fields_ = [("struct_size", C.c_int), ("B", C.c_int), ("Lv", C.c_int), ("Lt", C.c_int),
                ("d", C.c_int), ("H", C.c_int), ("F", C.c_int), ("E", C.c_int),
                ("Dv", C.c_int), ("Dt", C.c_int), ("n_proj", C.c_int),
                ("precise", C.c_int), ("training", C.c_int), ("proj_precise", C.c_int),
                ("p_in", C.c_float), ("p_attn", C.c_float), ("p_path", C.c_float),
                ("seed", C.c_ulonglong), ("loss_only", C.c_int), ("use_txt_pos", C.c_int), ("max_q_l", C.c_int)]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_size = C.sizeof(Dims)


ABI_VERSION = 304          # uvtg_version() this binding was written against


_P, _I, _F, _LL = C.c_void_p, C.c_int, C.c_float, C.c_longlong
_DP = C.POINTER(Dims)
# name -> (restype, argtypes); the authoritative prototypes are in include/uvtg.h
SIGNATURES = {
    "uvtg_version": (_I, []),
    "uvtg_dev_config_from_env": (_I, []),
    "uvtg_dev_config_set": (_I, [C.c_char_p, C.c_char_p]),
    "uvtg_strerror": (C.c_char_p, [_I]),
    "uvtg_param_count": (_I, [_DP]),
    "uvtg_param_numel": (_I, [_DP, _I, C.POINTER(_LL)]),
    "uvtg_param_offsets": (_I, [_DP, C.POINTER(_LL)]),
    "uvtg_workspace_bytes": (C.c_size_t, [_DP]),
    "uvtg_wcache_bytes": (C.c_size_t, [_DP]),
    "uvtg_loss_ws_floats": (_LL, [_I, _I, _I]),
    "uvtg_prepare_weights": (_I, [_DP, _P, _P, _P]),
    "uvtg_forward": (_I, [_DP, _P, _P] + [_P] * 5 + [_P] * 6 + [_P, _P] + [_P]),
    "uvtg_backward": (_I, [_DP, _P, _P] + [_P] * 4 + [_P] * 4 + [_P] * 5 + [_LL, _LL] + [_P, _P] + [_P, _P, _P] + [_P, _I] + [_P]),
    "uvtg_criterion_fwd": (_I, [_I, _I, _I, _I, _F, _P, _P, _P, _LL, _LL, _P] + [_P] * 6 + [_P, _P] + [_P] * 3 + [_P]),
    "uvtg_criterion_bwd": (_I, [_I, _I, _I, _I, _F, _P, _P, _P, _LL, _LL, _P] + [_P] * 6 + [_P, _P, _P] + [_P] * 6 + [_P] * 3 + [_P]),
    "uvtg_forward_saliency_stats": (_I, [_DP, _P, _P, _P, _P]),
    "uvtg_backward_event_groups": (_I, [_I, _P]),
    "uvtg_cls_nce_ws_floats": (_LL, [_I, _I]),
    "uvtg_cls_nce_fwd": (_I, [_I, _I, _I, _P, _LL, _LL, _P, _P, _P, _P, _P, _P, _P]),
    "uvtg_cls_nce_bwd": (_I, [_I, _I, _I, _P, _LL, _LL, _P, _P, _P, _P, _P, _P, _P, _LL, _LL, _P, _P]),
    "uvtg_linear_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "uvtg_split_f16": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "uvtg_linear_split": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "uvtg_wgrad_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "uvtg_wgrad_scratch_floats": (_LL, [_I, _I, _I]),
    "uvtg_wgrad_bf16_ws": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _LL, _P]),
    "uvtg_wgrad_multi_slab_floats": (_LL, [_I]),
    "uvtg_wgrad_bf16_multi": (_I, [_I, _P, _P, _P, _P, _P, _P, _I, _P, _LL, _P, _I, _P]),
    "uvtg_cast_bf16": (_I, [_P, _P, _LL, _P]),
    "uvtg_layernorm_fwd": (_I, [_P] * 6 + [_I, _I, _P]),
    "uvtg_layernorm_bwd": (_I, [_P] * 8 + [_I, _I, _P]),
    "uvtg_attention_fwd": (_I, [_P] * 4 + [_I] * 5 + [_P]),
    "uvtg_attention_bwd": (_I, [_P] * 7 + [_F] + [_I] * 4 + [_P]),
    "uvtg_ragged_to_padded": (_I, [_P, _I, _P, _I, _I, _I, _P, _P, _P]),
    "uvtg_sine_position": (_I, [_P] * 5 + [_I] * 4 + [_P]),
    "uvtg_adamw_clip_step": (_I, [_P, _P, _P, _P, _LL, _F, _F, _F, _F, _F, _I, _F, _F, _P, _P]),
    "uvtg_adamw_clip_step_prenorm": (_I, [_P, _P, _P, _P, _LL, _F, _F, _F, _F, _F, _I, _F, _F, _P, _P]),
    "uvtg_backward_gradnorm2": (_P, [_P, _P]),
    "uvtg_debug_force_nt_tile": (_I, [_I]),
    "uvtg_debug_force_nt_bm": (_I, [_I]),
    "uvtg_debug_gemm_cus": (_I, [_I]),
    "uvtg_set_reserved_cus": (_I, [_I]),
    "uvtg_cast_f32": (_I, [_P, _P, _LL, _P]),
    "uvtg_debug_nt_tile_rows": (_I, [_I, _I, _I, _I, _I]),
    "uvtg_debug_nt_plan": (_I, [_I, _I, _I, _I, _I, _I, _P]),
    "uvtg_debug_layernorm_fwd_bf16": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _I, _P]),
    "uvtg_debug_ln_fwd_lean": (_I, [_I]),
    "uvtg_debug_delta_fuse": (_I, [_I]),
    "uvtg_debug_attn_ws": (_I, [_I]),
    "uvtg_debug_attn_fwd_dma": (_I, [_I]),
    "uvtg_debug_last_layer_clip": (_I, [_I]),
    "uvtg_debug_tn_conv_defer": (_I, [_I]),
    "uvtg_debug_nt_plan2": (_I, [_I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "uvtg_linear_sk_ws_floats": (_LL, []),
    "uvtg_linear_bf16_sk": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "uvtg_linear_split_sk": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "uvtg_debug_nt_splitk": (_I, [_I]),
    "uvtg_debug_nt_small": (_I, [_I]),
    "uvtg_debug_nt_plan_override": (_I, [_I, _I, _I, _I, _I]),
    "uvtg_debug_nt_cgw": (_I, [_I]),
    "uvtg_debug_nt_loader_waves": (_I, [_I]),
    "uvtg_debug_nt_splitk_parts": (_I, [_I, _I, _I, _I, _I]),
    "uvtg_debug_nt_small_tile": (_I, [_I, _I, _I, _I, _I]),
    "uvtg_profile_start": (_I, []),
    "uvtg_profile_stop": (_I, [_P, _P, _P]),
    "uvtg_profile_event_floor_ms": (C.c_double, []),
    "uvtg_profile_bytes": (_I, [_P]),
    "uvtg_profile_sections_start": (_I, []),
    "uvtg_profile_sections_stop": (_I, [_P, _P]),
    "uvtg_detr_criterion": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _P, _P, _I, _I, _F, _F, _F, _P, _P,
                                 _P, _P, _P, _P, _P, _P, _P]),
    "uvtg_hungarian": (_I, [_P, _I, _P, _I, _I, _P, _P, _I, _F, _F, _F, _P, _P, _P, _P, _P]),
    "uvtg_decode_rank_nms": (_I, [_P] * 5 + [_I, _I, C.c_double, _I, _I] + [_P] * 4 + [_P]),
    "uvtg_postprocess_mr": (_I, [_P] * 6 + [_I, _I, _F, _I, C.c_double, _I, _I] + [_P] * 5 + [_P]),
}

_lib = None


def load():
    """dlopen libuvtg.so once; raises RuntimeError (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the UniVTG MI355X kernels are not built. Run `python -m univtg_amd.build` "
            "(needs hipcc; cross-compiles for gfx950 without a GPU). There is no CPU / PyTorch fallback.")
    # torch first: libuvtg.so must share the HIP runtime instance (and its device context / streams) that
    # PyTorch-ROCm loaded; loading it before torch binds a second runtime that sees no device.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if name.startswith("uvtg_debug_") and not hasattr(lib, name) and os.environ.get("UVTG_LIB_PATH"):
            continue                     # an older build loaded for an A/B (dev): experiment knobs may be missing
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.uvtg_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH}: uvtg_version() = {lib.uvtg_version()}, this binding expects {ABI_VERSION}; rebuild with "
                           "`python -m univtg_amd.build --force`")
    # developer switches (include/uvtg_dev.h): the library never reads the environment on its own; the tools opt in with UVTG_DEV_ENV=1
    if os.environ.get("UVTG_DEV_ENV") == "1":
        lib.uvtg_dev_config_from_env()
    _lib = lib
    return lib


def check(code: int, what: str = "uvtg"):
    if code != 0:
        msg = load().uvtg_strerror(code)
        raise RuntimeError(f"{what} failed with code {code}: {msg.decode() if msg else '?'}")
