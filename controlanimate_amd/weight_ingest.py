"""Weight ingest for the HIP models: everything between a checkpoint file and
`model.load_state_dict(...); model.prepare()`.  Pure host code, no kernels.

Mirrors (behaviour, not code):
  animatediff/utils/util.py:101-175                      load_weights: motion module, DreamBooth (LDM single file ->
                                                         diffusers names for UNet + VAE), LoRA fuse, motion LoRA
  animatediff/utils/convert_from_ckpt.py:328-510         convert_ldm_unet_checkpoint
  animatediff/utils/convert_from_ckpt.py:557-665         convert_ldm_vae_checkpoint
  animatediff/utils/convert_lora_safetensor_to_diffusers.py:28-47    convert_motion_lora_ckpt_to_diffusers
  animatediff/utils/convert_lora_safetensor_to_diffusers.py:51-115   convert_lora (kohya names, W += alpha * up @ down)
  diffusers 0.23 `load_lora_weights` + `fuse_lora(lora_scale)` (third party, what util.py:151-153 actually calls):
        W += lora_scale * (network_alpha / rank) * up @ down

The LDM -> diffusers renaming is generated from the model configuration (which levels have attention,
layers per block) instead of pattern-rewriting checkpoint keys; tests/golden/ldm_keymaps.json holds the
maps the REFERENCE converter produces for the SD1.5 layouts and tests/test_weight_ingest.py compares.
Fusing happens on the fp32 master parameters; call `model.prepare()` afterwards to repack the arena.
"""
from __future__ import annotations

import logging
import os
import re
from typing import Dict, Iterable, List, Mapping, Optional, Sequence, Tuple

import torch

logger = logging.getLogger(__name__)

UNET_PREFIX = "model.diffusion_model."
VAE_PREFIX = "first_stage_model."

_RES_LDM_TO_HF = (("in_layers.0", "norm1"), ("in_layers.2", "conv1"), ("emb_layers.1", "time_emb_proj"),
                  ("out_layers.0", "norm2"), ("out_layers.3", "conv2"), ("skip_connection", "conv_shortcut"))


def _cfg(config, key, default=None):
    if isinstance(config, Mapping):
        return config.get(key, default)
    return getattr(config, key, default)


# ------------------------------------------------------------------------------------ UNet (LDM -> diffusers)
def ldm_unet_module_map(config) -> List[Tuple[str, str]]:
    """(ldm module prefix, diffusers module prefix) pairs for an SD-style UNet described by `config`
    (block_out_channels, layers_per_block, down_block_types / up_block_types).  A ResBlock's or
    transformer's inner parameter names are handled by the caller."""
    lpb = int(_cfg(config, "layers_per_block", 2))
    down_types = list(_cfg(config, "down_block_types"))
    up_types = list(_cfg(config, "up_block_types"))
    nb = len(down_types)
    pairs: List[Tuple[str, str]] = [("time_embed.0", "time_embedding.linear_1"), ("time_embed.2", "time_embedding.linear_2"),
                                    ("input_blocks.0.0", "conv_in"), ("out.0", "conv_norm_out"), ("out.2", "conv_out")]
    idx = 1
    for b in range(nb):
        attn = "CrossAttn" in down_types[b]
        for l in range(lpb):
            pairs.append((f"input_blocks.{idx}.0", f"down_blocks.{b}.resnets.{l}"))
            if attn:
                pairs.append((f"input_blocks.{idx}.1", f"down_blocks.{b}.attentions.{l}"))
            idx += 1
        if b < nb - 1:
            pairs.append((f"input_blocks.{idx}.0.op", f"down_blocks.{b}.downsamplers.0.conv"))
            idx += 1
    pairs += [("middle_block.0", "mid_block.resnets.0"), ("middle_block.1", "mid_block.attentions.0"),
              ("middle_block.2", "mid_block.resnets.1")]
    idx = 0
    for b in range(nb):
        attn = "CrossAttn" in up_types[b]
        for l in range(lpb + 1):
            pairs.append((f"output_blocks.{idx}.0", f"up_blocks.{b}.resnets.{l}"))
            if attn:
                pairs.append((f"output_blocks.{idx}.1", f"up_blocks.{b}.attentions.{l}"))
            if l == lpb and b < nb - 1:
                pairs.append((f"output_blocks.{idx}.{2 if attn else 1}.conv", f"up_blocks.{b}.upsamplers.0.conv"))
            idx += 1
    return pairs


def _rename_unet_key(key: str, modules: Sequence[Tuple[str, str]]) -> Optional[str]:
    # longest prefix first ("input_blocks.3.0.op" before "input_blocks.3.0")
    for src, dst in modules:
        if key == src or key.startswith(src + "."):
            rest = key[len(src):]
            if ".resnets." in dst:
                for a, b in _RES_LDM_TO_HF:
                    if rest.startswith("." + a + "."):
                        rest = "." + b + rest[len(a) + 1:]
                        break
            return dst + rest
    return None


def convert_ldm_unet_checkpoint(checkpoint: Mapping[str, torch.Tensor], config, extract_ema: bool = False) -> Dict[str, torch.Tensor]:
    """Single-file LDM / DreamBooth checkpoint -> diffusers UNet2DConditionModel names (the spatial part of
    UNet3DConditionModel; load with strict=False, the motion modules come from their own checkpoint)."""
    modules = sorted(ldm_unet_module_map(config), key=lambda p: -len(p[0]))
    # EMA checkpoints keep a second copy under `model_ema.` with dots stripped; > 100 such keys marks one
    has_ema = sum(k.startswith("model_ema") for k in checkpoint) > 100
    out: Dict[str, torch.Tensor] = {}
    for key, val in checkpoint.items():
        if not key.startswith(UNET_PREFIX):
            continue
        inner = key[len(UNET_PREFIX):]
        if has_ema and extract_ema:
            ema_key = "model_ema." + "".join(key.split(".")[1:])
            val = checkpoint.get(ema_key, val)
        new = _rename_unet_key(inner, modules)
        if new is not None:
            out[new] = val
    return out


# ------------------------------------------------------------------------------------ VAE (LDM -> diffusers)
def ldm_vae_module_map(config) -> List[Tuple[str, str]]:
    boc = list(_cfg(config, "block_out_channels"))
    lpb = int(_cfg(config, "layers_per_block", 2))
    nb = len(boc)
    pairs = [("encoder.norm_out", "encoder.conv_norm_out"), ("decoder.norm_out", "decoder.conv_norm_out")]
    for side in ("encoder", "decoder"):
        pairs += [(f"{side}.mid.block_1", f"{side}.mid_block.resnets.0"), (f"{side}.mid.block_2", f"{side}.mid_block.resnets.1"),
                  (f"{side}.mid.attn_1", f"{side}.mid_block.attentions.0")]
    for i in range(nb):
        for j in range(lpb):
            pairs.append((f"encoder.down.{i}.block.{j}", f"encoder.down_blocks.{i}.resnets.{j}"))
        pairs.append((f"encoder.down.{i}.downsample.conv", f"encoder.down_blocks.{i}.downsamplers.0.conv"))
        for j in range(lpb + 1):  # LDM numbers decoder levels from the output side
            pairs.append((f"decoder.up.{nb - 1 - i}.block.{j}", f"decoder.up_blocks.{i}.resnets.{j}"))
        pairs.append((f"decoder.up.{nb - 1 - i}.upsample.conv", f"decoder.up_blocks.{i}.upsamplers.0.conv"))
    return pairs


_VAE_ATTN = (("norm", "group_norm"), ("q", "query"), ("k", "key"), ("v", "value"), ("proj_out", "proj_attn"))


def convert_ldm_vae_checkpoint(checkpoint: Mapping[str, torch.Tensor], config) -> Dict[str, torch.Tensor]:
    """`first_stage_model.*` -> diffusers AutoencoderKL names.  Attention projections come out under the
    names the reference emits (query/key/value/proj_attn, 1x1 convs squeezed to matrices); AutoencoderKL
    (ours and diffusers') renames them to to_q/to_k/to_v/to_out.0 on load."""
    modules = sorted(ldm_vae_module_map(config), key=lambda p: -len(p[0]))
    out: Dict[str, torch.Tensor] = {}
    for key, val in checkpoint.items():
        if not key.startswith(VAE_PREFIX):
            continue
        inner = key[len(VAE_PREFIX):]
        new = inner
        for src, dst in modules:
            if inner.startswith(src + "."):
                rest = inner[len(src) + 1:]
                if ".attentions." in dst:
                    head, _, tail = rest.partition(".")
                    rest = dict(_VAE_ATTN).get(head, head) + "." + tail
                    if val.dim() > 2 and tail == "weight" and head != "norm":  # 1x1 conv projections -> matrices
                        val = val[:, :, 0, 0]
                elif ".resnets." in dst:
                    rest = re.sub(r"^nin_shortcut\.", "conv_shortcut.", rest)
                new = dst + "." + rest
                break
        out[new] = val
    return out


# ------------------------------------------------------------------------------------ LoRA
def _module_weight(root: torch.nn.Module, path: str) -> torch.nn.Parameter:
    mod = root.get_submodule(path)
    w = getattr(mod, "weight", None)
    if w is None:
        raise KeyError(f"{path} has no weight")
    return w


def _add_delta(w: torch.nn.Parameter, up: torch.Tensor, down: torch.Tensor, scale: float):
    up2, down2 = up.float().flatten(1), down.float().flatten(1)
    if down.dim() == 4 and down.shape[2:] != (1, 1):  # 3x3 LoRA-down followed by a 1x1 up (LoCon-style conv LoRA)
        delta = torch.einsum("or,rikl->oikl", up2, down.float())
    else:
        delta = up2 @ down2
    w.data += (scale * delta).reshape(w.shape).to(device=w.device, dtype=w.dtype)


def _kohya_module_index(root: torch.nn.Module) -> Dict[str, str]:
    """'down_blocks_0_attentions_0_transformer_blocks_0_attn1_to_q' -> dotted module path.  (The reference
    resolves such names by trial-and-error getattr; an index is exact and O(1).)"""
    return {name.replace(".", "_"): name for name, m in root.named_modules() if getattr(m, "weight", None) is not None}


def fuse_lora(unet: torch.nn.Module, state_dict: Mapping[str, torch.Tensor], lora_scale: float = 1.0,
              use_network_alpha: bool = True, prefix: str = "lora_unet") -> List[str]:
    """Adds LoRA deltas to the UNet's master weights; returns the fused module paths.
    Accepts kohya files (`lora_unet_<path_with_underscores>.lora_{down,up}.weight` + `.alpha`; LCM-LoRA and
    most community LoRAs) and diffusers files (`unet.<path>.{processor.<proj>_lora|lora}.{down,up}.weight`).
    use_network_alpha=True is diffusers' fuse_lora (scale * alpha / rank), False is the reference's own
    convert_lora (ignores `.alpha`)."""
    fused: List[str] = []
    index = None
    for key in state_dict:
        if key.endswith(".lora_down.weight") and key.startswith(prefix + "_"):
            if index is None:
                index = _kohya_module_index(unet)
            stem = key[: -len(".lora_down.weight")]
            path = index.get(stem[len(prefix) + 1:])
            if path is None:
                raise KeyError(f"LoRA target {stem} not found in the UNet")
            down, up = state_dict[key], state_dict[stem + ".lora_up.weight"]
            scale = lora_scale
            if use_network_alpha and (stem + ".alpha") in state_dict:
                scale *= float(state_dict[stem + ".alpha"]) / down.shape[0]
            _add_delta(_module_weight(unet, path), up, down, scale)
            fused.append(path)
        elif key.endswith("down.weight") and ("_lora." in key or ".lora." in key) and "text_encoder" not in key:
            stem = key[: -len("down.weight")]
            up = state_dict[stem + "up.weight"]
            path = key[len("unet."):] if key.startswith("unet.") else key
            m = re.match(r"(.*)\.processor\.(to_q|to_k|to_v|to_out)_lora\.down\.weight$", path)
            if m:
                path = f"{m.group(1)}.{m.group(2)}" + (".0" if m.group(2) == "to_out" else "")
            else:
                path = path[: -len(".lora.down.weight")]
            scale = lora_scale
            alpha_key = stem[:-1] + ".alpha" if stem.endswith(".") else stem + "alpha"
            if use_network_alpha and alpha_key in state_dict:
                scale *= float(state_dict[alpha_key]) / state_dict[key].shape[0]
            _add_delta(_module_weight(unet, path), up, state_dict[key], scale)
            fused.append(path)
    return fused


def lora_text_encoder_keys(state_dict: Mapping[str, torch.Tensor]) -> List[str]:
    """Keys of a LoRA file that target the CLIP text encoder (kohya `lora_te_*`, diffusers `text_encoder.*`)."""
    return [k for k in state_dict if k.startswith("lora_te_") or k.startswith("text_encoder.")]


def fuse_lora_text_encoder(text_encoder: torch.nn.Module, state_dict: Mapping[str, torch.Tensor], lora_scale: float = 1.0,
                           use_network_alpha: bool = True) -> List[str]:
    """Text-encoder half of diffusers' load_lora_weights + fuse_lora (what util.py:155-156 runs): kohya keys
    `lora_te_text_model_encoder_layers_<i>_{self_attn_{q,k,v,out}_proj,mlp_fc{1,2}}.lora_{down,up}.weight` (+ `.alpha`)
    and diffusers keys `text_encoder.<path>.lora_linear_layer.{down,up}.weight` are added to the master weights
    with the same scale * alpha / rank rule as the UNet half."""
    fused = fuse_lora(text_encoder, {k: v for k, v in state_dict.items() if k.startswith("lora_te_")}, lora_scale,
                      use_network_alpha, prefix="lora_te")
    for key, down in state_dict.items():
        if not (key.startswith("text_encoder.") and key.endswith("down.weight")):
            continue
        stem = key[: -len("down.weight")]
        up = state_dict[stem + "up.weight"]
        path = key[len("text_encoder."):]
        path = re.sub(r"\.(lora_linear_layer|lora)\.down\.weight$", "", path)
        path = re.sub(r"\.(to_q|to_k|to_v|to_out)_lora\.down\.weight$", lambda m: "." + {"to_q": "q_proj", "to_k": "k_proj", "to_v": "v_proj", "to_out": "out_proj"}[m.group(1)], path)
        scale = lora_scale
        alpha_key = stem[:-1] + ".alpha" if stem.endswith(".") else stem + "alpha"
        if use_network_alpha and alpha_key in state_dict:
            scale *= float(state_dict[alpha_key]) / down.shape[0]
        _add_delta(_module_weight(text_encoder, path), up, down, scale)
        fused.append(path)
    return fused


def convert_ldm_clip_checkpoint(checkpoint: Mapping[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """convert_from_ckpt.py:716-728: the CLIP text tower of an LDM checkpoint is its `cond_stage_model.transformer.`
    sub-dict under transformers' own key names (returned as a state dict; the reference instantiates the model)."""
    pre = "cond_stage_model.transformer."
    return {k[len(pre):]: v for k, v in checkpoint.items() if k.startswith(pre)}


def convert_lora(unet: torch.nn.Module, state_dict: Mapping[str, torch.Tensor], alpha: float = 0.6) -> List[str]:
    """The reference's own kohya fuse (convert_lora_safetensor_to_diffusers.py:51-115): `.alpha` entries ignored."""
    return fuse_lora(unet, state_dict, lora_scale=alpha, use_network_alpha=False)


def fuse_motion_lora(unet: torch.nn.Module, state_dict: Mapping[str, torch.Tensor], alpha: float = 1.0) -> List[str]:
    """AnimateDiff motion LoRA (`...motion_modules.0...attention_blocks.0.processor.to_q_lora.down.weight`):
    W += alpha * up @ down on the temporal attention projections."""
    fused = []
    for key, down in state_dict.items():
        if "up." in key:
            continue
        up = state_dict[key.replace(".down.", ".up.")]
        path = key.replace("processor.", "").replace("_lora", "").replace("down.", "").replace("up.", "")
        path = path.replace("to_out.", "to_out.0.")
        path = path[len("unet."):] if path.startswith("unet.") else path
        path = path.rsplit(".", 1)[0]  # drop "weight"
        _add_delta(_module_weight(unet, path), up, down, alpha)
        fused.append(path)
    return fused


# ------------------------------------------------------------------------------------ files and orchestration
def read_checkpoint(path: str, allow_pickle: Optional[bool] = None) -> Dict[str, torch.Tensor]:
    """.safetensors or torch pickle (.ckpt/.pth/.bin); unwraps a top-level "state_dict".
    Pickles are read with `weights_only=True` (tensors and plain containers only).  Community .ckpt files that
    carry arbitrary Python objects need the explicit opt-in `allow_pickle=True` (or CA_ALLOW_PICKLE=1): that
    executes code from the file, exactly as the reference's bare `torch.load` does (util.py:117,127)."""
    if path.endswith(".safetensors"):
        from safetensors import safe_open
        out = {}
        with safe_open(path, framework="pt", device="cpu") as f:
            for k in f.keys():
                out[k] = f.get_tensor(k)
        return out
    if allow_pickle is None:
        allow_pickle = os.environ.get("CA_ALLOW_PICKLE", "0") == "1"
    try:
        sd = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as exc:
        if not allow_pickle:
            raise RuntimeError(f"{path}: not loadable with weights_only=True ({type(exc).__name__}); pass allow_pickle=True "
                               f"(or CA_ALLOW_PICKLE=1) only for files you trust") from exc
        sd = torch.load(path, map_location="cpu", weights_only=False)
    return sd["state_dict"] if isinstance(sd, dict) and "state_dict" in sd else sd


def load_motion_module(unet: torch.nn.Module, state_dict: Mapping[str, torch.Tensor]) -> List[str]:
    """util.py:113-120: only `motion_modules.` tensors are taken; nothing may be unexpected."""
    sub = {k: v for k, v in state_dict.items() if "motion_modules." in k}
    missing, unexpected = unet.load_state_dict(sub, strict=False)
    if unexpected:
        raise RuntimeError(f"unexpected motion-module keys: {unexpected[:4]} ...")
    return list(sub)


def load_weights(animation_pipeline, motion_module_path: str = "", motion_module_lora_configs: Iterable[dict] = (),
                 dreambooth_model_path: str = "", lora_model_path: Sequence[str] = (), lora_alpha: Sequence[float] = ()):
    """Same arguments and order of operations as the reference's load_weights (util.py:101-175), including
    the text-encoder halves (dreambooth CLIP tower, `lora_te_*` deltas) when the pipeline carries a fusable text_encoder;
    when it does not, the skipped tensors are logged (never silently dropped)."""
    unet = animation_pipeline.unet
    if motion_module_path:
        load_motion_module(unet, read_checkpoint(motion_module_path))
    if dreambooth_model_path:
        sd = read_checkpoint(dreambooth_model_path)
        vae = getattr(animation_pipeline, "vae", None)
        if vae is not None and hasattr(vae, "load_state_dict"):
            vae.load_state_dict(convert_ldm_vae_checkpoint(sd, vae.config))
        unet.load_state_dict(convert_ldm_unet_checkpoint(sd, unet.config), strict=False)
        te = getattr(animation_pipeline, "text_encoder", None)
        te_sd = convert_ldm_clip_checkpoint(sd)
        if te_sd:
            if te is not None and hasattr(te, "load_state_dict"):
                te.load_state_dict(te_sd, strict=False)
            else:
                logger.warning("dreambooth checkpoint carries %d text-encoder tensors but the pipeline has no text_encoder "
                               "to load them into: prompts will be embedded by the base CLIP weights", len(te_sd))
    if isinstance(lora_model_path, str):
        lora_model_path = [lora_model_path] if lora_model_path else []
    if isinstance(lora_alpha, (int, float)):
        lora_alpha = [lora_alpha] * len(lora_model_path)
    for path, alpha in zip(lora_model_path, lora_alpha):
        lsd = read_checkpoint(path)
        fuse_lora(unet, lsd, lora_scale=float(alpha))
        # the reference's load_lora_weights + fuse_lora (util.py:155-156) also fuses the text-encoder half
        te = getattr(animation_pipeline, "text_encoder", None)
        te_keys = lora_text_encoder_keys(lsd)
        if te_keys and te is not None and isinstance(te, torch.nn.Module):
            fuse_lora_text_encoder(te, lsd, lora_scale=float(alpha))
        elif te_keys:
            logger.warning("%s: %d text-encoder LoRA tensors skipped (no fusable text_encoder attached); prompt embeddings "
                           "will differ from the reference for this LoRA", path, len(te_keys))
    for cfg in motion_module_lora_configs:
        fuse_motion_lora(unet, read_checkpoint(cfg["path"]), float(cfg["alpha"]))
    for m in (unet, getattr(animation_pipeline, "vae", None), getattr(animation_pipeline, "text_encoder", None)):
        if m is not None and getattr(m, "arena", None) is not None:
            m.prepare()  # repack the device arena from the updated master weights
    return animation_pipeline
