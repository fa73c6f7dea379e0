"""ctypes binding of the C ABI declared in include/controlanimate_hip.h.

The product path has NO CPU / eager fallback: if the shared library is missing, `lib()` raises
(`CAHipUnavailable`) with the build command.  Struct layouts mirror the header field by field.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CA_HIP_LIB") or os.path.join(_HERE, "csrc", "libcontrolanimate_hip.so")  # (CA_HIP_LIB: another build of the same ABI, for same-box A/B timing)

CA_BF16, CA_F16 = 0, 1
CA_ACT_NONE, CA_ACT_SILU = 0, 1
ABI_VERSION = 13


class CAHipUnavailable(RuntimeError):
    pass


class CAHipError(RuntimeError):
    pass


class GemmArgs(C.Structure):
    _fields_ = [
        ("a", C.c_void_p), ("a2", C.c_void_p), ("w", C.c_void_p), ("c", C.c_void_p),
        ("bias", C.c_void_p), ("rowbias", C.c_void_p), ("residual", C.c_void_p),
        ("lda", C.c_int64), ("lda2", C.c_int64), ("ldc", C.c_int64), ("ld_res", C.c_int64),
        ("ld_rowbias", C.c_int64),
        ("m", C.c_int32), ("n", C.c_int32), ("k1", C.c_int32), ("k2", C.c_int32),
        ("rows_per_group", C.c_int32),
        ("alpha", C.c_float), ("post_scale", C.c_float),
        ("act", C.c_int32), ("geglu", C.c_int32), ("out_f32", C.c_int32), ("dtype", C.c_int32),
        ("ln_stats", C.c_void_p), ("ln_colsum", C.c_void_p), ("ln_eps", C.c_float),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("row_sums_out", C.c_void_p), ("ln_parts", C.c_int32),
        ("w_frag", C.c_void_p),
    ]


class FfArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w1_frag", C.c_void_p), ("bias1", C.c_void_p), ("colsum1", C.c_void_p), ("ln_stats", C.c_void_p),
        ("w2_frag", C.c_void_p), ("bias2", C.c_void_p), ("residual", C.c_void_p), ("y", C.c_void_p),
        ("lda", C.c_int64), ("ldc", C.c_int64), ("ld_res", C.c_int64),
        ("m", C.c_int32), ("c", C.c_int32), ("inner", C.c_int32), ("ln_eps", C.c_float), ("dtype", C.c_int32),
        ("w_out_frag", C.c_void_p), ("bias_out", C.c_void_p), ("residual_out", C.c_void_p), ("ld_res_out", C.c_int64),  # ABI v12
    ]


class TattnArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w_frag", C.c_void_p), ("gamma", C.c_void_p), ("bias_pe", C.c_void_p), ("o", C.c_void_p),
        ("lda", C.c_int64), ("ldo", C.c_int64), ("ld_bias_pe", C.c_int64),
        ("batch", C.c_int32), ("frames", C.c_int32), ("tokens", C.c_int32), ("heads", C.c_int32), ("c", C.c_int32),
        ("ln_eps", C.c_float), ("scale", C.c_float), ("dtype", C.c_int32),
        ("w_out_frag", C.c_void_p), ("bias_out", C.c_void_p), ("residual", C.c_void_p), ("ld_res", C.c_int64),  # ABI v12
    ]


ATTN_WOUT_FRAG_ELEMS = 102400  # CA_ATTN_WOUT_FRAG_ELEMS
TATTN_W_FRAG_ELEMS = 368640  # CA_TATTN_W_FRAG_ELEMS


class XattnArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("wq_frag", C.c_void_p), ("bias", C.c_void_p), ("kv_frag", C.c_void_p), ("o", C.c_void_p),
        ("lda", C.c_int64), ("ldo", C.c_int64),
        ("m", C.c_int32), ("tokens", C.c_int32), ("frames_per_kv", C.c_int32), ("kv_mod", C.c_int32), ("kv_batches", C.c_int32),
        ("nk", C.c_int32), ("heads", C.c_int32), ("c", C.c_int32),
        ("ln_eps", C.c_float), ("dtype", C.c_int32),
        ("w_out_frag", C.c_void_p), ("bias_out", C.c_void_p), ("residual", C.c_void_p), ("ld_res", C.c_int64),  # ABI v12
        ("kv_frag_ip", C.c_void_p), ("nk_ip", C.c_int32), ("ip_scale", C.c_float),                              # ABI v13
    ]


XATTN_W_FRAG_ELEMS = 122880   # CA_XATTN_W_FRAG_ELEMS
XATTN_KV_FRAG_ELEMS = 7680    # CA_XATTN_KV_FRAG_ELEMS, per (text batch, head)


class ConvArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("x2", C.c_void_p), ("w", C.c_void_p), ("y", C.c_void_p),
        ("bias", C.c_void_p), ("rowbias", C.c_void_p), ("residual", C.c_void_p),
        ("ld_res", C.c_int64), ("ld_rowbias", C.c_int64),
        ("images", C.c_int32), ("hin", C.c_int32), ("win", C.c_int32),
        ("cin1", C.c_int32), ("cin2", C.c_int32), ("cout", C.c_int32),
        ("stride", C.c_int32), ("upsample", C.c_int32), ("rows_per_group", C.c_int32),
        ("alpha", C.c_float), ("post_scale", C.c_float),
        ("act", C.c_int32), ("out_f32", C.c_int32), ("dtype", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64), ("pad_asym", C.c_int32),
        ("w_wino", C.c_void_p), ("x_is_wino_v", C.c_int32),  # ABI v12
    ]


class GroupNormArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("x2", C.c_void_p), ("y", C.c_void_p),
        ("gamma", C.c_void_p), ("beta", C.c_void_p), ("partials", C.c_void_p),
        ("images", C.c_int32), ("hw", C.c_int32), ("c1", C.c_int32), ("c2", C.c_int32),
        ("groups", C.c_int32), ("frames_per_stat", C.c_int32),
        ("eps", C.c_float), ("act", C.c_int32), ("dtype", C.c_int32),
        ("wino_v", C.c_void_p), ("wino_h", C.c_int32), ("wino_w", C.c_int32),  # ABI v12
    ]


class LayerNormArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("y", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
        ("pos", C.c_void_p),
        ("rows", C.c_int64),
        ("c", C.c_int32), ("rows_per_frame", C.c_int32), ("frames", C.c_int32),
        ("eps", C.c_float), ("dtype", C.c_int32), ("stats", C.c_void_p),
    ]


class AttnArgs(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("o", C.c_void_p),
        ("q_outer", C.c_int64), ("q_inner", C.c_int64), ("q_row", C.c_int64),
        ("o_outer", C.c_int64), ("o_inner", C.c_int64), ("o_row", C.c_int64),
        ("k_outer", C.c_int64), ("k_inner", C.c_int64), ("k_row", C.c_int64),
        ("inner_count", C.c_int32), ("kv_inner_count", C.c_int32), ("kv_div", C.c_int32), ("kv_mod", C.c_int32),
        ("batches", C.c_int32), ("heads", C.c_int32), ("head_dim", C.c_int32),
        ("nq", C.c_int32), ("nk", C.c_int32),
        ("scale", C.c_float), ("out_scale", C.c_float),
        ("accumulate", C.c_int32), ("dtype", C.c_int32), ("causal", C.c_int32),
        ("key_mask", C.c_void_p), ("key_mask_stride", C.c_int64),
    ]


# symbol -> (restype, argtypes); this table is also what tests/test_capi_symbols.py checks
# against the declarations in include/controlanimate_hip.h.
SYMBOLS = {
    "ca_abi_version": (C.c_int, []),
    "ca_last_error": (C.c_char_p, []),
    "ca_gemm": (C.c_int, [C.POINTER(GemmArgs), C.c_void_p]),
    "ca_pack_w_frag": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "ca_pack_w2_frag": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "ca_ff_fused": (C.c_int, [C.POINTER(FfArgs), C.c_void_p]),
    "ca_ff_fused_supported": (C.c_int, [C.POINTER(FfArgs)]),
    "ca_tattn_fused": (C.c_int, [C.POINTER(TattnArgs), C.c_void_p]),
    "ca_tattn_fused_supported": (C.c_int, [C.POINTER(TattnArgs)]),
    "ca_pack_w_tattn": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "ca_pack_w_out": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "ca_xattn_fused": (C.c_int, [C.POINTER(XattnArgs), C.c_void_p]),
    "ca_xattn_fused_supported": (C.c_int, [C.POINTER(XattnArgs)]),
    "ca_xattn_pack_w": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "ca_xattn_pack_kv": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p, C.c_void_p]),
    "ca_gemm_ln_inline_supported": (C.c_int, [C.POINTER(GemmArgs)]),
    "ca_gemm_wants_finished_stats": (C.c_int, [C.POINTER(GemmArgs)]),
    "ca_ln_finish_sums": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "ca_gemm_workspace_bytes": (C.c_int64, [C.POINTER(GemmArgs)]),
    "ca_gemm_row_sums_parts": (C.c_int, [C.POINTER(GemmArgs)]),
    "ca_gemm_plan_name": (C.c_int, [C.POINTER(GemmArgs), C.c_char_p, C.c_int32]),
    "ca_conv3x3_plan_name": (C.c_int, [C.POINTER(ConvArgs), C.c_char_p, C.c_int32]),
    "ca_conv3x3": (C.c_int, [C.POINTER(ConvArgs), C.c_void_p]),
    "ca_pack_w_wino": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "ca_conv3x3_workspace_bytes": (C.c_int64, [C.POINTER(ConvArgs)]),
    "ca_softmax_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_float, C.c_int32, C.c_void_p]),
    "ca_groupnorm_partials_floats": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "ca_groupnorm_stats": (C.c_int, [C.POINTER(GroupNormArgs), C.c_void_p]),
    "ca_groupnorm_apply": (C.c_int, [C.POINTER(GroupNormArgs), C.c_void_p]),
    "ca_groupnorm": (C.c_int, [C.POINTER(GroupNormArgs), C.c_void_p]),
    "ca_groupnorm_wino_supported": (C.c_int, [C.POINTER(GroupNormArgs)]),
    "ca_layernorm": (C.c_int, [C.POINTER(LayerNormArgs), C.c_void_p]),
    "ca_attention": (C.c_int, [C.POINTER(AttnArgs), C.c_void_p]),
    "ca_attention_plan_name": (C.c_int, [C.POINTER(AttnArgs), C.c_char_p, C.c_int32]),
    "ca_add_bcast": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]),
    "ca_repeat": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "ca_silu_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "ca_timestep_embedding": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ca_latents_to_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p]),
    "ca_nhwc_to_ncfhw_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ca_ncfhw_to_nhwc": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.c_void_p, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ca_cfg_scheduler_step": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         C.POINTER(C.c_float), C.c_float, C.c_void_p]),
    "ca_lincomb": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_float), C.c_int32, C.c_int64, C.c_void_p]),
}

_lib = None


def lib() -> C.CDLL:
    """Loads (once) and returns the shared library; raises CAHipUnavailable if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime: it must be loaded FIRST so that this library binds to the
    # same libamdhip64 instance (otherwise the process ends up with two runtimes and launches from
    # here fail with "no ROCm-capable device is detected").
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise CAHipUnavailable(
            f"{LIB_PATH} is missing: the HIP extension is not built. Run "
            "`python -m controlanimate_amd._build` (needs hipcc); there is no CPU fallback.")
    try:
        handle = C.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 missing
        raise CAHipUnavailable(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(handle, name)
        fn.restype = res
        fn.argtypes = args
    if handle.ca_abi_version() != ABI_VERSION:
        raise CAHipUnavailable(f"ABI mismatch: library {handle.ca_abi_version()} vs binding {ABI_VERSION}; rebuild")
    _lib = handle
    return handle


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().ca_last_error()
        raise CAHipError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")
