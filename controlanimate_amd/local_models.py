"""Building the pipeline's models from LOCAL files, the way the reference's constructors do from
`from_pretrained` (there is no network here: a name that is not on disk is an error that says where it was looked for).

    modules/controlanimate_pipeline.py:26-48   tokenizer / text_encoder / vae / unet from `pretrained_model_path/<subfolder>`
    modules/controlresiduals_pipeline.py:25-38 ControlNetModel.from_pretrained(name) per ControlNet name
    modules/ip_adapter.py:83-94                CLIPVisionModelWithProjection.from_pretrained(image_encoder_path)
    animatediff/pipelines/controlanimation_pipeline.py:160-163  VaeImageProcessor(do_normalize=False) for control images
    diffusers TextualInversionLoaderMixin (third party): load_textual_inversion / maybe_convert_prompt, restated for the
    one embedding file the reference loads (models/TI/easynegative.safetensors, controlanimate_pipeline.py:118-121)

Host-side file handling only; the models themselves run on the HIP path.
"""
from __future__ import annotations

import glob
import json
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .weight_ingest import read_checkpoint

WEIGHT_NAMES = ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.fp16.safetensors", "model.safetensors",
                "diffusion_pytorch_model.bin", "pytorch_model.bin")


class Config(dict):
    """A config mapping with attribute access and `.get` -- what the reference reads from its OmegaConf object
    (`config.steps`, `model_config.get("dreambooth_path", "")`)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def load_config(path: str, **overrides) -> Config:
    import yaml
    with open(path) as fh:
        cfg = Config(yaml.safe_load(fh) or {})
    cfg.update(overrides)
    return cfg


def resolve_model_dir(name: str, subfolder: Optional[str] = None, extra_roots: Sequence[str] = ()) -> str:
    """A `from_pretrained` name -> an existing local directory: the path itself, `models/<basename>`, one of
    `extra_roots`, or a snapshot of the Hugging Face cache (`$HF_HOME/hub/models--org--name/snapshots/<rev>`)."""
    tried: List[str] = []

    def ok(d):
        d = os.path.join(d, subfolder) if subfolder else d
        tried.append(d)
        return d if os.path.isdir(d) else None

    cands = [name, os.path.join("models", os.path.basename(name.rstrip("/")))]
    cands += [os.path.join(r, os.path.basename(name.rstrip("/"))) for r in extra_roots]
    cands += [os.path.join(r, name) for r in extra_roots]
    hub = os.path.join(os.environ.get("HF_HOME", os.path.join(os.path.expanduser("~"), ".cache", "huggingface")), "hub")
    cands += sorted(glob.glob(os.path.join(hub, "models--" + name.replace("/", "--"), "snapshots", "*")))
    for c in cands:
        d = ok(c)
        if d:
            return d
    raise FileNotFoundError(f"model '{name}'" + (f" (subfolder '{subfolder}')" if subfolder else "") +
                            " is not on disk and there is no network to download it; looked in: " + ", ".join(tried))


_SKELETON = False


class skeleton_weights:
    """`with skeleton_weights(): ...` -- the model loaders below build their modules from the config files WITHOUT reading the
    weight files (parameters keep their default initialisation).  The ranks > 0 of a window-sharded run build their models
    this way, pack them (`prepare()`: the arenas get their final layout) and then RECEIVE the arena contents from rank 0
    over RCCL (vid2vid.run_video_sharded, window_shard.broadcast_weights): one process reads the checkpoints, not eight."""

    def __enter__(self):
        global _SKELETON
        self._old, _SKELETON = _SKELETON, True
        return self

    def __exit__(self, *exc):
        global _SKELETON
        _SKELETON = self._old
        return False


def is_skeleton() -> bool:
    return _SKELETON


def load_dir_state_dict(model_dir: str) -> Dict[str, torch.Tensor]:
    if _SKELETON:
        return {}
    for n in WEIGHT_NAMES:
        f = os.path.join(model_dir, n)
        if os.path.isfile(f):
            return read_checkpoint(f)
    raise FileNotFoundError(f"no weight file ({', '.join(WEIGHT_NAMES)}) in {model_dir}")


def load_dir_config(model_dir: str) -> dict:
    f = os.path.join(model_dir, "config.json")
    if not os.path.isfile(f):
        raise FileNotFoundError(f"{f} does not exist")
    with open(f) as fh:
        return {k: v for k, v in json.load(fh).items() if not k.startswith("_")}


def load_controlnet(name: str, extra_roots: Sequence[str] = ("models/ControlNet", "models/controlnet")):
    """ControlNetModel.from_pretrained(name) (controlresiduals_pipeline.py:33) from a local directory."""
    from .controlnet import ControlNetModel
    d = resolve_model_dir(name, extra_roots=extra_roots)
    cfg = load_dir_config(d)
    known = ControlNetModel.__init__.__code__.co_varnames
    net = ControlNetModel.from_config({k: v for k, v in cfg.items() if k in known})
    missing, unexpected = net.load_state_dict(load_dir_state_dict(d), strict=False)
    if unexpected:
        raise RuntimeError(f"{d}: unexpected ControlNet keys {unexpected[:4]} ...")
    return net


def load_vae(pretrained_model_path: str, vae_path: str = ""):
    """AutoencoderKL.from_pretrained(path, subfolder="vae") or .from_single_file(vae_path) (:37-40)."""
    from .vae import AutoencoderKL
    from .weight_ingest import convert_ldm_vae_checkpoint
    if vae_path:
        vae = AutoencoderKL.from_config()
        if _SKELETON:
            return vae
        sd = read_checkpoint(vae_path)
        if any(k.startswith("first_stage_model.") for k in sd) or any(k.startswith("encoder.down.") for k in sd):
            if not any(k.startswith("first_stage_model.") for k in sd):
                sd = {"first_stage_model." + k: v for k, v in sd.items()}
            sd = convert_ldm_vae_checkpoint(sd, vae.config)
        vae.load_state_dict(sd)
        return vae
    d = resolve_model_dir(pretrained_model_path, "vae")
    cfg = load_dir_config(d)
    from .vae import VAE_CONFIG
    vae = AutoencoderKL.from_config({k: v for k, v in cfg.items() if k in VAE_CONFIG})
    vae.load_state_dict(load_dir_state_dict(d), strict=not _SKELETON)
    return vae


def load_text_encoder(pretrained_model_path: str):
    """CLIPTextModel.from_pretrained(path, subfolder="text_encoder") (:35)."""
    from .clip import TEXT_CONFIG, CLIPTextModel
    d = resolve_model_dir(pretrained_model_path, "text_encoder")
    cfg = load_dir_config(d)
    te = CLIPTextModel.from_config({k: v for k, v in cfg.items() if k in TEXT_CONFIG})
    te.load_state_dict(load_dir_state_dict(d), strict=not _SKELETON)
    return te


def load_tokenizer(pretrained_model_path: str):
    """CLIPTokenizer.from_pretrained(path, subfolder="tokenizer") (:34): transformers' tokenizer on the local files."""
    from transformers import CLIPTokenizer
    d = resolve_model_dir(pretrained_model_path, "tokenizer")
    with open(os.path.join(d, "vocab.json")) as fh:
        vocab = json.load(fh)
    with open(os.path.join(d, "merges.txt"), encoding="utf-8") as fh:
        lines = [ln.rstrip("\n") for ln in fh if ln.strip() and not ln.startswith("#version")]
    try:  # transformers >= 5: CLIPTokenizer(vocab: dict, merges: list of pairs)
        return CLIPTokenizer(vocab=vocab, merges=[tuple(ln.split(" ")) for ln in lines], model_max_length=77)
    except TypeError:  # transformers 4.x: file paths
        return CLIPTokenizer(os.path.join(d, "vocab.json"), os.path.join(d, "merges.txt"), model_max_length=77)


def load_image_encoder(image_encoder_path: str):
    """CLIPVisionModelWithProjection.from_pretrained(image_encoder_path) (ip_adapter.py:83-86)."""
    from .clip import VISION_CONFIG, CLIPVisionModelWithProjection
    d = resolve_model_dir(image_encoder_path)
    cfg = load_dir_config(d)
    cfg = dict(cfg.get("vision_config", {}), **{k: v for k, v in cfg.items() if k in VISION_CONFIG})
    enc = CLIPVisionModelWithProjection.from_config({k: v for k, v in cfg.items() if k in VISION_CONFIG})
    enc.load_state_dict(load_dir_state_dict(d), strict=False)
    return enc


class VaeImageProcessor:
    """diffusers.image_processor.VaeImageProcessor as the pipeline builds it (controlanimation_pipeline.py:159-163):
    PIL / ndarray / tensor -> float32 [B,3,H,W]; resize to (height, width) or down to a multiple of `vae_scale_factor`
    (lanczos), /255, optional RGB conversion, optional 2x-1 normalisation (off for the control-image processor)."""

    def __init__(self, vae_scale_factor: int = 8, do_resize: bool = True, do_normalize: bool = True, do_convert_rgb: bool = False):
        self.vae_scale_factor, self.do_resize, self.do_normalize, self.do_convert_rgb = vae_scale_factor, do_resize, do_normalize, do_convert_rgb

    def preprocess(self, image, height: Optional[int] = None, width: Optional[int] = None) -> torch.Tensor:
        from PIL import Image
        images = image if isinstance(image, (list, tuple)) else [image]
        out = []
        for im in images:
            if torch.is_tensor(im):
                t = im.float()
                t = t[None] if t.dim() == 3 else t
                if height and width and tuple(t.shape[-2:]) != (height, width):
                    t = torch.nn.functional.interpolate(t, size=(height, width), mode="bilinear", align_corners=False)
            else:
                if not isinstance(im, Image.Image):
                    arr = np.asarray(im)
                    im = Image.fromarray(arr if arr.dtype == np.uint8 else (arr * 255).round().astype(np.uint8))
                if self.do_convert_rgb:
                    im = im.convert("RGB")
                if self.do_resize:
                    w, h = (width, height) if (height and width) else im.size
                    w, h = w - w % self.vae_scale_factor, h - h % self.vae_scale_factor
                    if (w, h) != im.size:
                        im = im.resize((w, h), resample=Image.LANCZOS)
                arr = np.asarray(im).astype(np.float32) / 255.0
                if arr.ndim == 2:
                    arr = arr[..., None]
                t = torch.from_numpy(arr).permute(2, 0, 1)[None]
            if self.do_normalize:
                t = 2.0 * t - 1.0
            out.append(t)
        return torch.cat(out)


# ------------------------------------------------------------------------------------ textual inversion
def read_textual_inversion(path: str, token: Optional[str] = None):
    """-> (tokens, embeddings [n, dim]).  A file with a single tensor (the reference's easynegative.safetensors:
    `emb_params` [8,768]) or an A1111 `.pt` (`string_to_param`); a multi-vector embedding becomes
    token, token_1, ..., token_{n-1} (diffusers TextualInversionLoaderMixin)."""
    sd = read_checkpoint(path)
    if "string_to_param" in sd:
        name = sd.get("name", None)
        emb = next(iter(sd["string_to_param"].values()))
    elif len(sd) == 1:
        name, emb = next(iter(sd.items()))
    else:
        raise ValueError(f"{path}: {len(sd)} tensors; a textual-inversion file holds exactly one embedding")
    token = token or name
    emb = emb.float()
    if emb.dim() == 1:
        emb = emb[None]
    tokens = [token] + [f"{token}_{i}" for i in range(1, emb.shape[0])]
    return tokens, emb


def maybe_convert_prompt(prompt, tokenizer):
    """TextualInversionLoaderMixin.maybe_convert_prompt: a multi-vector token `tok` in the prompt is replaced by
    `tok tok_1 tok_2 ...` for as long as those exist among the tokenizer's added tokens."""
    if isinstance(prompt, (list, tuple)):
        return [maybe_convert_prompt(p, tokenizer) for p in prompt]
    added = getattr(tokenizer, "added_tokens_encoder", None) or {}
    if not added or not hasattr(tokenizer, "tokenize") or not isinstance(prompt, str):
        return prompt  # nothing was added to this tokenizer (or it is a caller-supplied callable without a vocabulary)
    seen = []
    for tok in tokenizer.tokenize(prompt):
        if tok in seen or tok not in added:
            continue
        seen.append(tok)
        repl, i = tok, 1
        while f"{tok}_{i}" in added:
            repl += f" {tok}_{i}"
            i += 1
        prompt = prompt.replace(tok, repl)
    return prompt
