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
