"""GPU: the facade built from local files (`ControlAnimatePipeline(config)`, as scripts/vid2vid.py:152) runs a window end
to end on the HIP path: prompt (textual inversion expanded) -> CLIP -> VAE encode -> canny hints -> ControlNet + UNet3D
loop -> VAE decode -> PIL frames; the same seed gives the same frames."""
import numpy as np
import pytest

from test_facade_from_files_cpu import _config, model_tree  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def test_animate_from_a_config_only(model_tree):  # noqa: F811
    from PIL import Image
    from controlanimate_amd.controlanimate_pipeline import ControlAnimatePipeline
    cfg = _config(model_tree)
    pipe = ControlAnimatePipeline(cfg)
    rng = np.random.default_rng(5)
    frames = [Image.fromarray(rng.integers(0, 255, (64, 64, 3), dtype=np.uint8)) for _ in range(4)]
    out1 = pipe.animate(frames, None, cfg)
    out2 = pipe.animate(frames, None, cfg)
    assert len(out1) == 4 and out1[0].size == (64, 64)
    a, b = np.stack([np.asarray(x) for x in out1]), np.stack([np.asarray(x) for x in out2])
    assert a.std() > 0 and np.array_equal(a, b)


def test_weighted_prompt_through_the_hip_text_encoder(model_tree):  # noqa: F811
    """The facade encodes prompts the way the reference does (Compel syntax, modules/controlanimate_pipeline.py:133-135):
    an unweighted prompt is the plain CLIP encoding, `(x)++` scales x's tokens' distance from the empty-prompt encoding."""
    import torch
    from controlanimate_amd.controlanimate_pipeline import ControlAnimatePipeline
    from controlanimate_amd.prompt_weighting import Compel
    pipe = ControlAnimatePipeline(_config(model_tree))
    assert isinstance(pipe.encode_prompt, Compel)
    enc = pipe.encode_prompt
    plain = enc("a photo of a cat")
    assert torch.equal(plain, pipe._encode_plain("a photo of a cat").float())
    up = enc("a photo of a (cat)++")
    ids, w = enc.token_ids_and_weights([("a photo of a", 1.0), ("cat", 1.21)])
    z0 = pipe._encode_plain("").float()
    assert float(w.max()) == pytest.approx(1.21) and not torch.equal(up, plain)
    assert torch.allclose(up, z0 + (plain - z0) * w.to(plain.device)[..., None], atol=2e-3)
