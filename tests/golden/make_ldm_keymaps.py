"""Generates tests/golden/ldm_keymaps.json by running the REFERENCE's LDM->diffusers converters
(animatediff/utils/convert_from_ckpt.py: convert_ldm_unet_checkpoint, convert_ldm_vae_checkpoint) on a
synthetic SD1.5-layout single-file checkpoint whose tensors are unique tags.  Container-only
(needs /root/reference and the diffusers stub);  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_ldm_keymaps.py

The LDM key list is written from the LDM module layout (openaimodel.UNetModel / autoencoder ddconfig of
v1-inference.yaml), independently of controlanimate_amd/weight_ingest.py."""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refstub  # noqa: E402

_refstub.install()
from animatediff.utils.convert_from_ckpt import convert_ldm_unet_checkpoint, convert_ldm_vae_checkpoint  # noqa: E402


def ldm_unet_keys():
    keys = []
    wb = lambda p: [p + ".weight", p + ".bias"]  # noqa: E731

    def res(p, skip):
        out = wb(p + ".in_layers.0") + wb(p + ".in_layers.2") + wb(p + ".emb_layers.1") + wb(p + ".out_layers.0") + wb(p + ".out_layers.3")
        return out + (wb(p + ".skip_connection") if skip else [])

    def tx(p):
        out = wb(p + ".norm") + wb(p + ".proj_in") + wb(p + ".proj_out")
        t = p + ".transformer_blocks.0"
        for a in ("attn1", "attn2"):
            out += [f"{t}.{a}.to_q.weight", f"{t}.{a}.to_k.weight", f"{t}.{a}.to_v.weight"] + wb(f"{t}.{a}.to_out.0")
        out += wb(t + ".ff.net.0.proj") + wb(t + ".ff.net.2") + wb(t + ".norm1") + wb(t + ".norm2") + wb(t + ".norm3")
        return out

    keys += wb("time_embed.0") + wb("time_embed.2") + wb("input_blocks.0.0") + wb("out.0") + wb("out.2")
    # channel_mult (1,2,4,4), num_res_blocks 2, attention at the first three levels
    i, ch = 1, 320
    for level, mult in enumerate((1, 2, 4, 4)):
        for _ in range(2):
            keys += res(f"input_blocks.{i}.0", skip=(320 * mult != ch))
            ch = 320 * mult
            if level < 3:
                keys += tx(f"input_blocks.{i}.1")
            i += 1
        if level < 3:
            keys += wb(f"input_blocks.{i}.0.op")
            i += 1
    keys += res("middle_block.0", False) + tx("middle_block.1") + res("middle_block.2", False)
    i = 0
    for level in (3, 2, 1, 0):
        for j in range(3):
            keys += res(f"output_blocks.{i}.0", skip=True)  # every decoder ResBlock sees a concatenated skip
            if level < 3:
                keys += tx(f"output_blocks.{i}.1")
            if j == 2 and level > 0:
                keys += wb(f"output_blocks.{i}.{2 if level < 3 else 1}.conv")
            i += 1
    return ["model.diffusion_model." + k for k in keys]


def ldm_vae_keys():
    keys = []
    wb = lambda p: [p + ".weight", p + ".bias"]  # noqa: E731

    def res(p, skip):
        return wb(p + ".norm1") + wb(p + ".conv1") + wb(p + ".norm2") + wb(p + ".conv2") + (wb(p + ".nin_shortcut") if skip else [])

    def mid(p):
        out = res(p + ".block_1", False) + res(p + ".block_2", False)
        for n in ("norm", "q", "k", "v", "proj_out"):
            out += wb(f"{p}.attn_1.{n}")
        return out

    keys += wb("encoder.conv_in") + wb("encoder.norm_out") + wb("encoder.conv_out")
    ch = 128
    for i, mult in enumerate((1, 2, 4, 4)):
        for j in range(2):
            keys += res(f"encoder.down.{i}.block.{j}", skip=(128 * mult != ch))
            ch = 128 * mult
        if i < 3:
            keys += wb(f"encoder.down.{i}.downsample.conv")
    keys += mid("encoder.mid")
    keys += wb("decoder.conv_in") + wb("decoder.norm_out") + wb("decoder.conv_out") + mid("decoder.mid")
    ch = 512
    for i, mult in reversed(list(enumerate((1, 2, 4, 4)))):
        for j in range(3):
            keys += res(f"decoder.up.{i}.block.{j}", skip=(128 * mult != ch))
            ch = 128 * mult
        if i > 0:
            keys += wb(f"decoder.up.{i}.upsample.conv")
    keys += wb("quant_conv") + wb("post_quant_conv")
    return ["first_stage_model." + k for k in keys]


def tagged(keys, four_d=()):
    sd = {}
    for n, k in enumerate(keys):
        shape = (1, 1, 1, 1) if any(s in k for s in four_d) else (1,)
        sd[k] = torch.full(shape, float(n))
    return sd


def main():
    ukeys, vkeys = ldm_unet_keys(), ldm_vae_keys()
    usd = tagged(ukeys)
    ucfg = dict(layers_per_block=2, class_embed_type=None)
    uout = convert_ldm_unet_checkpoint(dict(usd), ucfg)
    tag2key = {int(v.flatten()[0]): k for k, v in usd.items()}
    umap = {k: tag2key[int(v.flatten()[0])] for k, v in uout.items()}
    vsd = tagged(vkeys, four_d=(".attn_1.q.", ".attn_1.k.", ".attn_1.v.", ".attn_1.proj_out."))
    vcfg = dict(block_out_channels=(128, 256, 512, 512), layers_per_block=2)
    vout = convert_ldm_vae_checkpoint(dict(vsd), vcfg)
    tag2key = {int(v.flatten()[0]): k for k, v in vsd.items()}
    vmap = {k: tag2key[int(v.flatten()[0])] for k, v in vout.items()}
    vdims = {k: v.dim() for k, v in vout.items() if ".attentions." in k}
    out = {"generator": "tests/golden/make_ldm_keymaps.py (reference converters, SD1.5 layout)",
           "unet": umap, "vae": vmap, "vae_attention_dims": vdims}
    with open(os.path.join(HERE, "ldm_keymaps.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print(f"unet: {len(ukeys)} ldm keys -> {len(umap)} diffusers keys; vae: {len(vkeys)} -> {len(vmap)}")


if __name__ == "__main__":
    main()
