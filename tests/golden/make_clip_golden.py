"""Generates tests/golden/clip_text_tiny.npz and clip_vision_tiny.npz by running the real transformers
CLIPTextModel / CLIPVisionModelWithProjection (the reference's third-party encoders, importable in the build
container) on tiny seeded configurations.  python tests/golden/make_clip_golden.py"""
import os

import numpy as np
import torch
from transformers import CLIPTextConfig, CLIPTextModel, CLIPVisionConfig, CLIPVisionModelWithProjection

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    torch.manual_seed(0)
    tcfg = CLIPTextConfig(vocab_size=100, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                          max_position_embeddings=16, hidden_act="quick_gelu", eos_token_id=2, bos_token_id=0, pad_token_id=1)
    tm = CLIPTextModel(tcfg).eval()
    for p in tm.parameters():  # default init is nearly identity-free noise; make LN affine and biases non-trivial
        if p.dim() == 1:
            p.data += 0.1 * torch.randn_like(p)
    ids = torch.randint(3, 99, (2, 16), generator=torch.Generator().manual_seed(1))
    ids[0, 9], ids[1, 15] = 99, 99  # the highest id plays EOS (legacy eos_token_id == 2 -> argmax)
    with torch.no_grad():
        out = tm(input_ids=ids, output_hidden_states=True)
        # the same prompts with some tokens hidden as attention KEYS (transformers' attention_mask, added to the causal mask):
        # what Compel's default DownweightMode.MASK passes for a down-weighted fragment
        amask = torch.ones(2, 16, dtype=torch.long)
        amask[0, 3:6] = 0
        amask[1, 7] = 0
        amask[1, 11:13] = 0
        out_m = tm(input_ids=ids, attention_mask=amask, output_hidden_states=True)
    # checkpoint naming (transformers 4.x, what the SD1.5 text_encoder files use): `text_model.` prefix;
    # transformers 5.x dropped the wrapper level in state_dict()
    sd = {(k if k.startswith("text_model.") else "text_model." + k): v.numpy() for k, v in tm.state_dict().items() if "position_ids" not in k}
    np.savez_compressed(os.path.join(HERE, "clip_text_tiny.npz"), input_ids=ids.numpy(), last_hidden_state=out.last_hidden_state.numpy(),
                        pooler_output=out.pooler_output.numpy(), hidden_1=out.hidden_states[1].numpy(),
                        attention_mask=amask.numpy(), masked_last_hidden_state=out_m.last_hidden_state.numpy(),
                        masked_pooler_output=out_m.pooler_output.numpy(),
                        **{"w:" + k: v for k, v in sd.items()})

    torch.manual_seed(1)
    vcfg = CLIPVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, image_size=28, patch_size=14,
                            projection_dim=32, hidden_act="gelu")
    vm = CLIPVisionModelWithProjection(vcfg).eval()
    for p in vm.parameters():
        if p.dim() == 1:
            p.data += 0.1 * torch.randn_like(p)
    px = torch.randn(2, 3, 28, 28, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        vo = vm(pixel_values=px)
    sd = {k: v.numpy() for k, v in vm.state_dict().items() if "position_ids" not in k}
    np.savez_compressed(os.path.join(HERE, "clip_vision_tiny.npz"), pixel_values=px.numpy(), image_embeds=vo.image_embeds.numpy(),
                        last_hidden_state=vo.last_hidden_state.numpy(), **{"w:" + k: v for k, v in sd.items()})
    print("text keys", len(tm.state_dict()), "vision keys", len(vm.state_dict()))


if __name__ == "__main__":
    main()
