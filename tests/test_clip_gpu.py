"""GPU parity of the HIP CLIP encoders (controlanimate_amd/clip.py) against the oracle (oracle/clip.py, itself
pinned to transformers by tests/test_clip_cpu.py): the transformers-generated tiny fixtures and seeded models
at the real widths (ViT-L text 768/12 heads/77 tokens, ViT-H vision 1280/16 heads/257 tokens)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item()


def test_text_tiny_fixture_from_transformers():
    from controlanimate_amd.clip import CLIPTextModel
    from tests.test_clip_cpu import TEXT_TINY
    z = np.load(os.path.join(GOLD, "clip_text_tiny.npz"))
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    m = CLIPTextModel.from_config(TEXT_TINY)
    m.load_state_dict(sd)
    m.to(DEV).prepare(DEV)
    out = m(torch.from_numpy(z["input_ids"]).to(DEV), output_hidden_states=True)
    assert rel(out[0], torch.from_numpy(z["last_hidden_state"])) < 5e-3
    assert rel(out.pooler_output, torch.from_numpy(z["pooler_output"])) < 5e-3
    assert rel(out.hidden_states[1], torch.from_numpy(z["hidden_1"])) < 5e-3 and len(out.hidden_states) == 3


def test_text_tiny_fixture_with_attention_mask_from_transformers():
    """ca_attention's key mask (ABI v7) under the causal mask == transformers' CLIPTextModel(attention_mask=...)."""
    from controlanimate_amd.clip import CLIPTextModel
    from tests.test_clip_cpu import TEXT_TINY
    z = np.load(os.path.join(GOLD, "clip_text_tiny.npz"))
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    m = CLIPTextModel.from_config(TEXT_TINY)
    m.load_state_dict(sd)
    m.to(DEV).prepare(DEV)
    ids, am = torch.from_numpy(z["input_ids"]).to(DEV), torch.from_numpy(z["attention_mask"]).to(DEV)
    out = m(ids, attention_mask=am)
    assert rel(out[0], torch.from_numpy(z["masked_last_hidden_state"])) < 5e-3
    assert rel(out.pooler_output, torch.from_numpy(z["masked_pooler_output"])) < 5e-3
    plain = m(ids)
    assert rel(plain[0], torch.from_numpy(z["last_hidden_state"])) < 5e-3
    assert rel(out[0], plain[0]) > 1e-2  # the mask matters
    # an all-ones mask is the plain call, bit for bit
    assert torch.equal(m(ids, attention_mask=torch.ones_like(am))[0], plain[0])


def test_vision_tiny_fixture_from_transformers():
    from controlanimate_amd.clip import CLIPVisionModelWithProjection
    from tests.test_clip_cpu import VIS_TINY
    z = np.load(os.path.join(GOLD, "clip_vision_tiny.npz"))
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    m = CLIPVisionModelWithProjection.from_config(VIS_TINY)
    m.load_state_dict(sd)
    m.to(DEV).prepare(DEV)
    out = m(torch.from_numpy(z["pixel_values"]).to(DEV))
    assert rel(out.image_embeds, torch.from_numpy(z["image_embeds"])) < 5e-3
    assert rel(out.last_hidden_state, torch.from_numpy(z["last_hidden_state"])) < 5e-3


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 1e-2), (torch.bfloat16, 3e-2)])
def test_text_real_width_77_tokens_causal(dtype, tol):
    from controlanimate_amd.clip import CLIPTextModel
    from oracle.clip import CLIPTextConfig, clip_text_forward, init_clip_weights
    over = dict(num_hidden_layers=3, vocab_size=1000)
    cfg = CLIPTextConfig(**over)
    sd = init_clip_weights(cfg, "text", seed=3)
    ids = torch.randint(0, 999, (2, 77), generator=torch.Generator().manual_seed(4))
    ids[0, 20], ids[1, 76] = 999, 999
    with torch.no_grad():
        last, pooled, _ = clip_text_forward(sd, cfg, ids)
    m = CLIPTextModel.from_config(over)
    m.load_state_dict(sd)
    m.to(DEV).prepare(DEV, dtype)
    out = m(ids.to(DEV))
    assert out.last_hidden_state.shape == (2, 77, 768)
    assert rel(out.last_hidden_state, last) < tol and rel(out.pooler_output, pooled) < tol, (rel(out.last_hidden_state, last),)
    # causal: changing a later token must not change earlier positions
    ids2 = ids.clone()
    ids2[0, 50] = 7
    out2 = m(ids2.to(DEV))
    assert torch.equal(out2.last_hidden_state[0, :50], out.last_hidden_state[0, :50])
    assert not torch.equal(out2.last_hidden_state[0, 50:], out.last_hidden_state[0, 50:])


def test_vision_vit_h_width_257_tokens():
    from controlanimate_amd.clip import CLIPVisionModelWithProjection
    from oracle.clip import CLIPVisionConfig, clip_vision_forward, init_clip_weights
    over = dict(num_hidden_layers=2)
    cfg = CLIPVisionConfig(**over)
    sd = init_clip_weights(cfg, "vision", seed=5)
    px = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        emb, last, _ = clip_vision_forward(sd, cfg, px)
    m = CLIPVisionModelWithProjection.from_config(over)
    m.load_state_dict(sd)
    m.to(DEV).prepare(DEV)
    out = m(px.to(DEV))
    assert out.image_embeds.shape == (2, 1024) and out.last_hidden_state.shape == (2, 257, 1280)
    assert rel(out.image_embeds, emb) < 1e-2 and rel(out.last_hidden_state, last) < 1e-2, (rel(out.image_embeds, emb), rel(out.last_hidden_state, last))
