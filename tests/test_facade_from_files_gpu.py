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


def test_skeleton_rank_receives_the_weights_and_reproduces_the_frames(model_tree):  # noqa: F811
    """Ranks > 0 of `vid2vid.run_video_sharded` build `ControlAnimatePipeline(config, skeleton=True)`: same modules, same arenas,
    NO weight file read (local_models.skeleton_weights), then receive `weight_buffers()` from rank 0.  That only works when both
    sides list the same tensors in the same order with the same shapes -- checked here on one GPU: identical arena layouts and
    manifests, different contents before the copy (the skeleton really read nothing), and bit-identical frames after copying the
    buffers over (what the broadcast does)."""
    import torch
    from PIL import Image
    from controlanimate_amd import local_models
    from controlanimate_amd.controlanimate_pipeline import ControlAnimatePipeline
    cfg = _config(model_tree)
    full = ControlAnimatePipeline(cfg)
    reads = []
    orig = local_models.read_checkpoint
    local_models.read_checkpoint = lambda *a, **k: (reads.append(a), orig(*a, **k))[1]
    try:
        skel = ControlAnimatePipeline(cfg, skeleton=True)
    finally:
        local_models.read_checkpoint = orig
    # (the textual-inversion file IS read: its vector count fixes how many tokens the tokenizer and the token table grow by --
    #  25 KB, and the values are overwritten by the broadcast like everything else)
    reads = [r for r in reads if "easynegative" not in str(r[0])]
    assert not reads, f"the skeleton read weight files: {reads}"
    assert not local_models.is_skeleton()  # the process-global flag is restored
    bf, bs = full.weight_buffers(), skel.weight_buffers()
    assert [(tuple(b.shape), b.dtype) for b in bf] == [(tuple(b.shape), b.dtype) for b in bs] and len(bf) >= 4

    def layout(p):
        models = [p.pipeline.unet] + list(p.multicontrolnetresiduals_pipeline.controlnets) + [p.pipeline.vae, p.pipeline.text_encoder]
        return [[(it.offset, it.nbytes, tuple(it.shape), it.dtype) for it in m.arena.items] for m in models if m is not None]

    assert layout(full) == layout(skel)
    assert any(not torch.equal(a, b) for a, b in zip(bf, bs))  # random init vs checkpoint contents
    for a, b in zip(bf, bs):
        b.copy_(a)
    rng = np.random.default_rng(5)
    frames = [Image.fromarray(rng.integers(0, 255, (64, 64, 3), dtype=np.uint8)) for _ in range(4)]
    out_f, out_s = full.animate(frames, None, cfg), skel.animate(frames, None, cfg)
    a, b = np.stack([np.asarray(x) for x in out_f]), np.stack([np.asarray(x) for x in out_s])
    assert a.std() > 0 and np.array_equal(a, b)
