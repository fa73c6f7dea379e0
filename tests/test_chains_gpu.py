"""Two denoising chains in flight on one GPU (controlanimate_amd/chains.py): independent windows on two host threads / HIP streams, one
set of models with a per-window cache slot per chain -- results equal to the sequential run of the same windows, bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _jobs(f, hw, n, n_steps, seed):
    g = torch.Generator().manual_seed(seed)
    jobs = []
    for k in range(n):
        jobs.append(dict(video_length=f, input_frames=None, height=8 * hw, width=8 * hw, num_inference_steps=n_steps, strength=1.0, guidance_scale=1.3,
                         generator=torch.Generator(device="cpu").manual_seed(100 + k), prompt_embeds=torch.randn(1, 77, 768, generator=g) * 0.5,
                         negative_prompt_embeds=torch.randn(1, 77, 768, generator=g) * 0.5, use_lcm=False, guess_mode=False,
                         latents=torch.randn(1, 4, f, hw, hw, generator=g), control_images={"n0": [h for h in torch.rand(f, 3, 8 * hw, 8 * hw, generator=g)]},
                         output_type="latent"))
    return jobs


def test_two_chains_equal_the_sequential_run_bit_for_bit():
    from controlanimate_amd.chains import ChainSet
    from controlanimate_amd.configs import NOISE_SCHEDULER_KWARGS
    from controlanimate_amd.controlanimation_pipeline import ControlAnimationPipeline
    from controlanimate_amd.controlresiduals_pipeline import MultiControlNetResidualsPipeline
    from controlanimate_amd.schedulers import get_scheduler
    from tests.test_pipeline_gpu import build
    ucfg, uw, unet, ccfg, cws, nets = build("v2", seed=91, n_controlnets=1)
    f, hw, n_steps, n_jobs = 8, 8, 4, 6

    def fresh():
        pipe = ControlAnimationPipeline(vae=None, text_encoder=None, tokenizer=None, unet=unet,
                                        scheduler=get_scheduler("LCMScheduler", **NOISE_SCHEDULER_KWARGS)).to(DEV)
        cn = MultiControlNetResidualsPipeline(["n0"], [0.8], use_lcm=False, controlnets=nets, device=DEV)
        return pipe, cn

    pipe, cn = fresh()
    seq = []
    for job in _jobs(f, hw, n_jobs, n_steps, seed=7):
        seq.append(pipe(multicontrolnetresiduals_pipeline=cn, **job).videos.clone())
    torch.cuda.synchronize()
    assert pipe.graph_replays == n_steps  # (a later window: every step replays)

    pipe2, cn2 = fresh()
    chains = ChainSet(pipe2, cn2, chains=2)
    assert chains.pipes[1].unet is pipe2.unet and chains.cns[1].controlnets[0] is nets[0] and chains.pipes[1].scheduler is not pipe2.scheduler
    threads_before = torch.get_num_threads()
    outs = chains.map(_jobs(f, hw, n_jobs, n_steps, seed=7))
    torch.cuda.synchronize()
    assert torch.get_num_threads() == threads_before and all(p_.single_host_thread for p_ in chains.pipes)  # (one switch around all chains, restored)
    for p_ in chains.pipes:
        assert p_.graph_fallback_reason is None and p_.graph_replays == n_steps   # both chains replay their own capture
    # each chain kept its own cache slots in the shared models: nobody re-captured after the priming window
    assert len(unet._slots) >= 2 and len(nets[0]._hints) >= 2
    for k, (a, b) in enumerate(zip(seq, outs)):
        assert torch.isfinite(b.videos).all() and torch.equal(a, b.videos), f"window {k} differs between the sequential run and two chains"
    assert not torch.equal(seq[0], seq[1])


def test_model_cache_slots_keep_two_pipelines_graphs_alive():
    """a second pipeline (or a facade alternating two configurations) on the same models used to evict the first one's
    per-window caches, so every alternation captured again.  With one cache slot per prompt / control-image tensor both keep replaying."""
    from controlanimate_amd.configs import NOISE_SCHEDULER_KWARGS
    from controlanimate_amd.controlanimation_pipeline import ControlAnimationPipeline
    from controlanimate_amd.controlresiduals_pipeline import MultiControlNetResidualsPipeline
    from controlanimate_amd.schedulers import get_scheduler
    from tests.test_pipeline_gpu import build
    ucfg, uw, unet, ccfg, cws, nets = build("v2", seed=93, n_controlnets=1)
    f, hw, n_steps = 8, 8, 4
    pipes = []
    for _ in range(2):
        p_ = ControlAnimationPipeline(vae=None, text_encoder=None, tokenizer=None, unet=unet, scheduler=get_scheduler("LCMScheduler", **NOISE_SCHEDULER_KWARGS)).to(DEV)
        pipes.append((p_, MultiControlNetResidualsPipeline(["n0"], [0.8], use_lcm=False, controlnets=nets, device=DEV)))
    jobs = _jobs(f, hw, 6, n_steps, seed=11)
    replays, reasons = [], []
    for k, job in enumerate(jobs):  # alternate the two pipelines window by window
        p_, cn_ = pipes[k % 2]
        p_(multicontrolnetresiduals_pipeline=cn_, **job)
        torch.cuda.synchronize()
        replays.append(p_.graph_replays)
        reasons.append(p_.graph_recapture_reason)
    assert replays == [n_steps - 1, n_steps - 1, n_steps, n_steps, n_steps, n_steps], (replays, reasons)  # one capture each, then replays only
