"""A/B switches for timing runs -- OUTSIDE the product package.

The product modules read no tuning environment variable: which one-launch / fused form runs is `controlanimate_amd.context.dispatch`,
all on.  Same-box A/B measurements (`bench.py` interleaved inside one gpurun call) still want to switch a form off from the
command line, so `bench.py` and the tools call `apply_from_env()`, which maps the variable names used in rounds 2-4 onto the
dispatch attributes.  Nothing in `controlanimate_amd/` imports this file."""
import os

ENV = {
    "CA_GEMM_AR_PY": "gemm_ar",
    "CA_FF_FUSED": "ff_fused",
    "CA_TATTN_FUSED": "tattn_fused",
    "CA_XATTN_FUSED": "xattn_fused",
    "CA_ATTN_OUT_FUSED": "attn_out_fused",
    "CA_XATTN_IP_FUSED": "xattn_ip_fused",
    "CA_CONV_WINOGRAD": "conv_winograd",
    "CA_GN_WINOGRAD": "gn_winograd",
    "CA_LN_ROWSUMS": "ln_row_sums",
    "CA_REPEAT_KERNEL": "repeat_kernel",
    "CA_LN_FOLD": "ln_fold",
    "CA_CFG_SHARED": "cfg_shared",
    "CA_CN_DEDUP": "cn_cfg_dedup",
}


def apply_from_env() -> dict:
    """Sets dispatch.<attr> = False for every CA_* variable that is "0"; returns what was switched off."""
    from controlanimate_amd.context import dispatch
    off = {}
    for var, attr in ENV.items():
        if os.environ.get(var, "1") == "0":
            setattr(dispatch, attr, False)
            off[var] = attr
    if os.environ.get("CA_CONTROLNET_STREAMS"):  # (an integer: streams a multi-ControlNet stack is spread over)
        dispatch.controlnet_streams = int(os.environ["CA_CONTROLNET_STREAMS"])
        off["CA_CONTROLNET_STREAMS"] = "controlnet_streams=%d" % dispatch.controlnet_streams
    return off
