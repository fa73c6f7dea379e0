"""CPU: the CLIP oracle (oracle/clip.py) against fixtures produced by the real transformers modules
(tests/golden/make_clip_golden.py), and the HIP models' checkpoint-key compatibility."""
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TEXT_TINY = dict(vocab_size=100, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, max_position_embeddings=16)
VIS_TINY = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, image_size=28, patch_size=14, projection_dim=32)


def _load(name):
    z = np.load(os.path.join(GOLD, name))
    return z, {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}


def test_text_oracle_matches_transformers():
    from oracle.clip import CLIPTextConfig, clip_text_forward
    z, sd = _load("clip_text_tiny.npz")
    last, pooled, hs = clip_text_forward(sd, CLIPTextConfig(**TEXT_TINY), torch.from_numpy(z["input_ids"]))
    assert torch.allclose(last, torch.from_numpy(z["last_hidden_state"]), atol=2e-5)
    assert torch.allclose(pooled, torch.from_numpy(z["pooler_output"]), atol=2e-5)
    assert torch.allclose(hs[1], torch.from_numpy(z["hidden_1"]), atol=2e-5) and len(hs) == 3


def test_text_oracle_with_attention_mask_matches_transformers():
    """transformers' CLIPTextModel with an attention_mask (tokens hidden as keys, on top of the causal mask): the call
    Compel's default down-weighting makes (modules/controlanimate_pipeline.py:133-135)."""
    from oracle.clip import CLIPTextConfig, clip_text_forward
    z, sd = _load("clip_text_tiny.npz")
    am = torch.from_numpy(z["attention_mask"])
    assert int((am == 0).sum()) == 6
    last, pooled, _ = clip_text_forward(sd, CLIPTextConfig(**TEXT_TINY), torch.from_numpy(z["input_ids"]), attention_mask=am)
    assert torch.allclose(last, torch.from_numpy(z["masked_last_hidden_state"]), atol=2e-5)
    assert torch.allclose(pooled, torch.from_numpy(z["masked_pooler_output"]), atol=2e-5)
    assert not torch.allclose(last, torch.from_numpy(z["last_hidden_state"]), atol=1e-3)  # the mask matters


def test_vision_oracle_matches_transformers():
    from oracle.clip import CLIPVisionConfig, clip_vision_forward
    z, sd = _load("clip_vision_tiny.npz")
    emb, last, _ = clip_vision_forward(sd, CLIPVisionConfig(**VIS_TINY), torch.from_numpy(z["pixel_values"]))
    assert torch.allclose(emb, torch.from_numpy(z["image_embeds"]), atol=2e-5)
    assert torch.allclose(last, torch.from_numpy(z["last_hidden_state"]), atol=2e-5)


def test_hip_models_take_the_checkpoint_keys():
    from controlanimate_amd.clip import CLIPTextModel, CLIPVisionModelWithProjection
    from oracle.clip import CLIPTextConfig, CLIPVisionConfig, clip_param_shapes
    _, tsd = _load("clip_text_tiny.npz")
    tm = CLIPTextModel.from_config(TEXT_TINY)
    assert set(tm.state_dict()) == set(tsd) == set(clip_param_shapes(CLIPTextConfig(**TEXT_TINY), "text"))
    tm.load_state_dict(tsd, strict=True)
    tm.load_state_dict({k[len("text_model."):]: v for k, v in tsd.items()}, strict=True)  # transformers 5.x naming
    _, vsd = _load("clip_vision_tiny.npz")
    vm = CLIPVisionModelWithProjection.from_config(VIS_TINY)
    assert set(vm.state_dict()) == set(vsd) == set(clip_param_shapes(CLIPVisionConfig(**VIS_TINY), "vision"))
    vm.load_state_dict(vsd, strict=True)
    # full-size key counts: SD1.5 text encoder 196 tensors (+position_ids buffer in old files), ViT-H image encoder 520
    assert len(CLIPTextModel.from_config().state_dict()) == 196
    full_v = clip_param_shapes(CLIPVisionConfig(), "vision")
    assert len(full_v) == 32 * 16 + 8 and full_v["vision_model.embeddings.position_embedding.weight"] == (257, 1280)


def test_clip_preprocess_matches_transformers_image_processor():
    import pytest
    transformers = pytest.importorskip("transformers")
    from PIL import Image
    from controlanimate_amd.clip import clip_preprocess
    rng = np.random.default_rng(0)
    ims = [Image.fromarray(rng.integers(0, 255, s, dtype=np.uint8)) for s in ((300, 400, 3), (512, 320, 3), (224, 224, 3))]
    ref = transformers.CLIPImageProcessor()(images=ims, return_tensors="pt").pixel_values
    assert torch.allclose(clip_preprocess(ims), ref, atol=1e-6)
