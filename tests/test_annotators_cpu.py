"""Canny annotator (controlanimate_amd/annotators.py; unpinned numpy restatement of cv2.Canny): algorithmic properties."""
import numpy as np
from PIL import Image


def test_step_edge_is_one_pixel_wide_and_thresholds_apply():
    from controlanimate_amd.annotators import canny_edges
    img = np.zeros((32, 32), np.uint8)
    img[:, 16:] = 200                                   # vertical step: Sobel dx = 4 * 200 = 800 on two columns
    e = canny_edges(img, 100, 200)
    cols = np.where(e.any(0))[0]
    assert len(cols) == 1 and cols[0] in (15, 16)        # thinned by non-maximum suppression
    assert (e[:, cols[0]] == 255).all()
    weak = np.zeros((32, 32), np.uint8)
    weak[:, 16:] = 30                                   # gradient 120: above low, below high, no strong seed -> dropped
    assert canny_edges(weak, 100, 200).sum() == 0
    assert canny_edges(weak, 50, 100).sum() > 0


def test_hysteresis_links_weak_to_strong():
    from controlanimate_amd.annotators import canny_edges
    img = np.zeros((40, 40), np.uint8)
    img[:20, 20:] = 200                                  # strong upper half of the edge
    img[20:, 20:] = 40                                   # weak lower half (gradient 160 in [100, 200])
    e = canny_edges(img, 100, 200)
    col = np.where(e[:18].any(0))[0][0]
    assert (e[2:38, col] == 255).mean() > 0.9            # the weak part survives because it touches the strong part
    only_weak = np.zeros((40, 40), np.uint8)
    only_weak[:, 20:] = 40                               # the same weak edge without any strong seed
    assert canny_edges(only_weak, 100, 200).sum() == 0


def test_colour_input_and_pil_interface():
    from controlanimate_amd.annotators import canny, canny_edges
    rgb = np.zeros((24, 24, 3), np.uint8)
    rgb[:, 12:, 1] = 255                                 # edge only in the green channel
    e = canny_edges(rgb)
    assert e[:, 11:13].any() and e.dtype == np.uint8 and set(np.unique(e)) <= {0, 255}
    out = canny(Image.fromarray(rgb))
    a = np.asarray(out)
    assert a.shape == (24, 24, 3) and (a[..., 0] == a[..., 1]).all() and (a[..., 0] == e).all()


def test_pipeline_uses_builtin_canny():
    from controlanimate_amd.controlresiduals_pipeline import MultiControlNetResidualsPipeline
    from controlanimate_amd.configs import controlnet_config
    from controlanimate_amd.controlnet import ControlNetModel
    net = ControlNetModel.from_config(controlnet_config(block_out_channels=(32, 64, 64, 64)))
    pipe = MultiControlNetResidualsPipeline(["lllyasviel/control_v11p_sd15_canny"], [1.0], use_lcm=False, controlnets=[net], device="cpu")
    rgb = np.zeros((16, 16, 3), np.uint8)
    rgb[:, 8:] = 255
    out = pipe.prepare_controlnet_input_image("lllyasviel/control_v11p_sd15_canny", Image.fromarray(rgb))
    assert np.asarray(out).max() == 255 and np.asarray(out).shape == (16, 16, 3)
    # tensors are control images that are already preprocessed (bench.py, the pipeline tests): never annotated
    import torch
    t = torch.rand(3, 16, 16)
    assert pipe.prepare_controlnet_input_image("lllyasviel/control_v11p_sd15_canny", t) is t
