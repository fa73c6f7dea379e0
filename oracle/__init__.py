"""CPU oracle: an fp32 restatement of the reference's denoising-loop arithmetic.

TEST INFRASTRUCTURE ONLY.  Nothing under `controlanimate_amd/` may import this package; only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg use it, and only as the
checker / the timed CPU baseline -- never as the product path.

What pins it (see DESIGN.md "Oracle"):
  * in-tree arithmetic of the reference (animatediff/models/*.py, modules/attention_processor.py,
    the custom LCMScheduler, get_w_embedding): checked here against the reference's OWN modules,
    imported in the build container behind a stub of its missing third-party imports
    (tests/golden/make_golden.py), and frozen as fixtures under tests/golden/.
  * third-party arithmetic that is NOT in /root/reference (diffusers==0.23.0: Attention, GEGLU,
    FeedForward, Timesteps, TimestepEmbedding, ControlNetModel, DDIM/LCM/Euler schedulers):
    restated from the published algorithm.  The reference has no tests or
    golden vectors for these, diffusers is not installable here (no network), so for those
    pieces the status is: PARITY UNPINNED (restatement only).
"""
