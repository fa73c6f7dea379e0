"""End-to-end on the GPU with every model on the HIP path: ControlAnimatePipeline.animate (facade) ->
CLIP text encoder -> VAE encode of the input frames -> canny annotator -> ControlNet + UNet3D denoising loop ->
VAE decode -> PIL frames, as scripts/vid2vid.py drives it (reduced-width seeded models)."""
import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SMALL = (64, 128, 256, 256)


class _Tok:
    """Stand-in for CLIPTokenizer (host string processing): deterministic ids, BOS/EOS framing, 77 tokens."""
    model_max_length, bos_token_id, eos_token_id, pad_token_id = 77, 0, 99, 99

    def __call__(self, text, padding=None, max_length=77, truncation=True, return_tensors=None):
        ids = [0] + [3 + (ord(ch) % 90) for ch in text][: max_length - 2] + [99]
        if padding == "max_length":
            ids += [99] * (max_length - len(ids))
        from types import SimpleNamespace
        return SimpleNamespace(input_ids=torch.tensor([ids]) if return_tensors == "pt" else ids)


def _components():
    from controlanimate_amd.clip import CLIPTextModel
    from controlanimate_amd.configs import controlnet_config, unet_config
    from controlanimate_amd.controlnet import ControlNetModel
    from controlanimate_amd.unet import UNet3DConditionModel
    from controlanimate_amd.vae import AutoencoderKL
    from oracle.clip import CLIPTextConfig, init_clip_weights
    from oracle.controlnet import ControlNetConfig, init_controlnet_weights
    from oracle.unet3d import UNet3DConfig, init_unet3d_weights
    from oracle.vae import VAEConfig, init_vae_weights
    unet = UNet3DConditionModel.from_config(unet_config("v2", block_out_channels=SMALL))
    unet.load_state_dict(init_unet3d_weights(UNet3DConfig.v2(block_out_channels=SMALL), seed=31))
    net = ControlNetModel.from_config(controlnet_config(block_out_channels=SMALL))
    net.load_state_dict(init_controlnet_weights(ControlNetConfig(block_out_channels=SMALL), seed=32))
    vae = AutoencoderKL.from_config(dict(block_out_channels=(32, 64, 64, 64)))
    vae.load_state_dict(init_vae_weights(VAEConfig(block_out_channels=(32, 64, 64, 64)), seed=33))
    tcfg = dict(vocab_size=100, num_hidden_layers=2)
    text = CLIPTextModel.from_config(tcfg)
    text.load_state_dict(init_clip_weights(CLIPTextConfig(**tcfg), "text", seed=34))
    return dict(unet=unet.to(DEV), controlnets=[net.to(DEV)], vae=vae.to(DEV), text_encoder=text.to(DEV), tokenizer=_Tok())


def test_animate_end_to_end_all_hip():
    from controlanimate_amd.controlanimate_pipeline import ControlAnimatePipeline
    cfg = dict(use_lcm=0, controlnets=["lllyasviel/control_v11p_sd15_canny"], cond_scale=[0.8], scheduler="LCMScheduler",
               prompt="a red fox running", n_prompt="blurry", seed=7, width=64, height=64, steps=3, strength=0.6, guidance_scale=1.3,
               frame_count=8, overlaps=0, epoch=0, guess_mode=0, use_img2img=True)
    pipe = ControlAnimatePipeline(cfg, _components(), device=DEV)
    rng = np.random.default_rng(0)
    frames_in = [Image.fromarray(rng.integers(0, 255, (64, 64, 3), dtype=np.uint8)) for _ in range(8)]
    out1 = pipe.animate(frames_in, None, cfg)
    out2 = pipe.animate(frames_in, None, cfg)
    assert len(out1) == 8 and all(isinstance(f, Image.Image) and f.size == (64, 64) for f in out1)
    a1, a2 = np.stack([np.asarray(f) for f in out1]), np.stack([np.asarray(f) for f in out2])
    assert np.array_equal(a1, a2)                       # same seed -> same frames (deterministic kernels + seeded RNG)
    assert a1.std() > 1.0                               # not a constant image
    cfg2 = dict(cfg, seed=8)
    a3 = np.stack([np.asarray(f) for f in pipe.animate(frames_in, None, cfg2)])
    assert not np.array_equal(a1, a3)


def _sharded_worker(rank, world, port, cfg, seed_frames, out_path):
    """One rank of run_video_sharded; both ranks share GPU 0 (gloo rendezvous: the 1-GPU rehearsal of the node-level flow)."""
    import os
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      CA_DIST_BACKEND="gloo")
    import numpy as np
    import torch
    from PIL import Image
    from controlanimate_amd import vid2vid
    rng = np.random.default_rng(seed_frames)
    frames = [Image.fromarray(rng.integers(0, 255, (64, 64, 3), dtype=np.uint8)) for _ in range(20)]
    comps = _components()
    if rank != 0:  # what a skeleton rank has before the broadcast: the right shapes, the wrong numbers
        for m in (comps["unet"], comps["controlnets"][0], comps["vae"], comps["text_encoder"]):
            for p in m.parameters():
                p.data.normal_(std=0.02)
    out = vid2vid.run_video_sharded(cfg, frames, components=comps, device="cuda:0")
    if rank == 0:
        np.save(out_path, np.stack([np.asarray(f) for f in out]))
        assert vid2vid.run_video_sharded.last_broadcast_bytes > 10_000_000
    else:
        assert out is None
    torch.distributed.destroy_process_group()


def test_sharded_video_equals_sequential(tmp_path):
    """scripts/vid2vid.py's window loop sharded over two ranks (windows r, r + 2, ...; weights broadcast from rank 0, whose
    arenas OVERWRITE rank 1's differently-initialised ones; colour match + cross-fade on rank 0) == the sequential loop
    (`run_windows`) on one process, byte for byte -- with overlap_strength >= 1, no loop-back, a deterministic sampler."""
    import socket
    import torch.multiprocessing as mp
    from controlanimate_amd import vid2vid
    from controlanimate_amd.controlanimate_pipeline import ControlAnimatePipeline
    cfg = dict(use_lcm=0, controlnets=["lllyasviel/control_v11p_sd15_canny"], cond_scale=[0.8], scheduler="DDIMScheduler",
               prompt="a red fox running", n_prompt="blurry", seed=7, width=64, height=64, steps=3, strength=1.0, overlap_strength=1.0,
               guidance_scale=1.3, frame_count=8, overlap_length=4, overlaps=0, epoch=0, guess_mode=0, use_img2img=True, loop_back_frames=False)
    # ---- sequential reference, this process
    rng = np.random.default_rng(123)
    frames = [Image.fromarray(rng.integers(0, 255, (64, 64, 3), dtype=np.uint8)) for _ in range(20)]
    pipe = ControlAnimatePipeline(cfg, _components(), device=DEV)
    wc = vid2vid.WindowConfig(frame_count=8, overlap_length=4, strength=1.0, overlap_strength=1.0, loop_back_frames=False)

    def animate(batch, last, c):
        return pipe.animate(batch, last, dict(cfg, frame_count=c.frame_count, strength=c.strength, overlaps=c.overlaps, epoch=c.epoch))

    seq = [f for win in vid2vid.run_windows(frames, animate, wc) for f in win]
    seq = np.stack([np.asarray(f) for f in seq])
    assert seq.shape == (20, 64, 64, 3)
    del pipe
    torch.cuda.empty_cache()
    # ---- two ranks on this GPU
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out_path = str(tmp_path / "sharded.npy")
    mp.spawn(_sharded_worker, args=(2, port, cfg, 123, out_path), nprocs=2, join=True)
    sharded = np.load(out_path)
    assert sharded.shape == seq.shape
    assert np.array_equal(sharded, seq), f"{int((sharded != seq).sum())} bytes differ"


def test_sharded_video_with_two_windows_in_flight_equals_sequential():
    """run_video_sharded(chains_per_gpu=2): this rank's windows on two facades (`ControlAnimatePipeline.twin()`: one set of models, two
    denoising / residuals pipelines, two host threads and HIP streams -- VAE encode, text encoder, ControlNet + UNet3D loop, VAE decode of
    two windows in flight) == the sequential loop of scripts/vid2vid.py:168-268 (`run_windows`), byte for byte.  36 frames = 8 windows."""
    from controlanimate_amd import vid2vid
    from controlanimate_amd.controlanimate_pipeline import ControlAnimatePipeline
    cfg = dict(use_lcm=0, controlnets=["lllyasviel/control_v11p_sd15_canny"], cond_scale=[0.8], scheduler="DDIMScheduler",
               prompt="a red fox running", n_prompt="blurry", seed=7, width=64, height=64, steps=3, strength=1.0, overlap_strength=1.0,
               guidance_scale=1.3, frame_count=8, overlap_length=4, overlaps=0, epoch=0, guess_mode=0, use_img2img=True, loop_back_frames=False)
    rng = np.random.default_rng(321)
    frames = [Image.fromarray(rng.integers(0, 255, (64, 64, 3), dtype=np.uint8)) for _ in range(36)]
    comps = _components()
    pipe = ControlAnimatePipeline(cfg, comps, device=DEV)
    wc = vid2vid.WindowConfig(frame_count=8, overlap_length=4, strength=1.0, overlap_strength=1.0, loop_back_frames=False)

    def animate(batch, last, c):
        return pipe.animate(batch, last, dict(cfg, frame_count=c.frame_count, strength=c.strength, overlaps=c.overlaps, epoch=c.epoch))

    seq = np.stack([np.asarray(f) for win in vid2vid.run_windows(frames, animate, wc) for f in win])
    assert seq.shape == (36, 64, 64, 3)
    two = vid2vid.run_video_sharded(cfg, frames, components=comps, device=DEV, chains_per_gpu=2)
    two = np.stack([np.asarray(f) for f in two])
    assert two.shape == seq.shape and np.array_equal(two, seq), f"{int((two != seq).sum())} bytes differ"
    with pytest.raises(ValueError, match="use_lcm"):
        vid2vid.run_video_sharded(dict(cfg, use_lcm=1), frames, components=comps, device=DEV, chains_per_gpu=2)


# ---- BASELINE config 4's shape of problem: IP-Adapter + window sharding (fixed image prompt) ----------------------------------
def _img_enc(pil):
    """Stand-in for the CLIP vision tower (PIL -> [1, 1024] image embedding), deterministic in the pixels."""
    a = np.asarray(pil.convert("RGB").resize((16, 16)), dtype=np.float32).reshape(-1) / 255.0 - 0.5
    return torch.from_numpy(np.concatenate([a, np.zeros(1024 - a.size, np.float32)])[None])


_IP_CFG = dict(use_lcm=0, use_ipadapter=1, ipa_scale=0.6, controlnets=["lllyasviel/control_v11p_sd15_canny"], cond_scale=[0.8],
               scheduler="DDIMScheduler", prompt="a red fox running", n_prompt="blurry", seed=7, width=64, height=64, steps=3, strength=1.0,
               overlap_strength=1.0, guidance_scale=1.3, frame_count=8, overlap_length=4, overlaps=0, epoch=0, guess_mode=0, use_img2img=True,
               loop_back_frames=False)


def _ip_pipe(perturb=False):
    from controlanimate_amd.controlanimate_pipeline import ControlAnimatePipeline
    comps = dict(_components(), image_encoder=_img_enc)
    torch.manual_seed(2024)  # the IP processors' to_k_ip / to_v_ip and the ImageProjModel are initialised from the global RNG
    return ControlAnimatePipeline(_IP_CFG, comps, device=DEV)


def _sharded_ip_worker(rank, world, port, seed_frames, ref_image, out_path):
    import os
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      CA_DIST_BACKEND="gloo")
    from controlanimate_amd import vid2vid
    rng = np.random.default_rng(seed_frames)
    frames = [Image.fromarray(rng.integers(0, 255, (64, 64, 3), dtype=np.uint8)) for _ in range(20)]
    comps = dict(_components(), image_encoder=_img_enc)
    torch.manual_seed(2024 if rank == 0 else 77)  # rank 1: other IP / projection weights, replaced by the broadcast
    if rank != 0:
        for m in (comps["unet"], comps["controlnets"][0]):
            for p in m.parameters():
                p.data.normal_(std=0.02)
    out = vid2vid.run_video_sharded(_IP_CFG, frames, components=comps, device="cuda:0", ip_reference_image=ref_image)
    if rank == 0:
        np.save(out_path, np.stack([np.asarray(f) for f in out]))
        np.save(out_path + ".tok.npy", vid2vid.run_video_sharded.last_image_prompt.numpy())
    else:
        assert out is None
    torch.distributed.destroy_process_group()


def test_sharded_ip_adapter_video_equals_sequential(tmp_path):
    """use_ipadapter + window sharding: with a FIXED image prompt (here `ip_reference_image`, embedded on rank 0 and broadcast) the
    windows are independent, and the two-rank result equals the sequential window loop driven with the same tokens through the
    reference's animate(image_prompt_embeds=, uncond_image_prompt_embeds=) parameters -- byte for byte."""
    import socket
    import torch.multiprocessing as mp
    from controlanimate_amd import vid2vid
    rng = np.random.default_rng(321)
    frames = [Image.fromarray(rng.integers(0, 255, (64, 64, 3), dtype=np.uint8)) for _ in range(20)]
    ref_image = Image.fromarray(np.random.default_rng(5).integers(0, 255, (64, 64, 3), dtype=np.uint8))
    pipe = _ip_pipe()
    tok, untok = pipe.pipeline.ip_adapter.get_image_embeds_4controlanimate(pil_image=ref_image, scale=_IP_CFG["ipa_scale"])
    assert tok.shape == (1, 4, 768) and untok.shape == (1, 4, 768) and float((tok - untok).abs().max()) > 0
    wc = vid2vid.WindowConfig(frame_count=8, overlap_length=4, strength=1.0, overlap_strength=1.0, loop_back_frames=False)

    def animate(batch, last, c, **kw):
        return pipe.animate(batch, last, dict(_IP_CFG, frame_count=c.frame_count, strength=c.strength, overlaps=c.overlaps, epoch=c.epoch), **kw)

    seq = np.stack([np.asarray(f) for win in vid2vid.run_windows(
        frames, lambda b, l, c: animate(b, l, c, image_prompt_embeds=tok, uncond_image_prompt_embeds=untok), wc) for f in win])
    # the image prompt is live: without it (zero tokens on both halves, reference :707-710) the video differs
    wc0 = vid2vid.WindowConfig(frame_count=8, overlap_length=4, strength=1.0, overlap_strength=1.0, loop_back_frames=False)
    plain = np.stack([np.asarray(f) for f in animate(frames[:8], None, wc0)])
    first_fixed = np.stack([np.asarray(f) for f in animate(frames[:8], None, wc0, image_prompt_embeds=tok, uncond_image_prompt_embeds=untok)])
    assert not np.array_equal(plain, first_fixed)
    # refused without a fixed prompt: the windows would form a chain
    with pytest.raises(ValueError, match="FIXED image prompt"):
        vid2vid.run_video_sharded(_IP_CFG, frames, components=None, device=DEV)
    del pipe
    torch.cuda.empty_cache()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out_path = str(tmp_path / "sharded_ip.npy")
    mp.spawn(_sharded_ip_worker, args=(2, port, 321, ref_image, out_path), nprocs=2, join=True)
    sharded = np.load(out_path)
    assert np.allclose(np.load(out_path + ".tok.npy"), torch.stack([tok, untok]).cpu().numpy())
    assert sharded.shape == seq.shape == (20, 64, 64, 3)
    assert np.array_equal(sharded, seq), f"{int((sharded != seq).sum())} bytes differ"


def test_sharded_ip_adapter_initial_generation_baseline():
    """`do_initial_generation` (scripts/vid2vid.py:199-203): rank 0 generates window 0 without an image prompt and its first output
    frame becomes the baseline image of the whole video (single process: no group, the same code path minus the broadcast)."""
    from controlanimate_amd import vid2vid
    rng = np.random.default_rng(11)
    frames = [Image.fromarray(rng.integers(0, 255, (64, 64, 3), dtype=np.uint8)) for _ in range(12)]
    cfg = dict(_IP_CFG, do_initial_generation=True)
    comps = dict(_components(), image_encoder=_img_enc)
    torch.manual_seed(2024)
    out = vid2vid.run_video_sharded(cfg, frames, components=comps, device=DEV)
    assert len(out) == 12
    got = vid2vid.run_video_sharded.last_image_prompt
    pipe = _ip_pipe()
    base = pipe.animate(frames[:8], None, dict(cfg, frame_count=8))[0]
    tok, untok = pipe.pipeline.ip_adapter.get_image_embeds_4controlanimate(pil_image=base, scale=cfg["ipa_scale"])
    assert torch.allclose(got, torch.stack([tok, untok]).cpu().float(), atol=1e-6)
