"""Several independent denoising chains in flight on ONE GPU (a throughput mode of the window-sharded video).

The windows of a sharded video are independent problems.  One chain leaves the GPU under-filled for a good part of a step -- one
kernel in flight for 65 % of it, grids of 64-256 blocks at the 16x16- and 8x8-latent levels (profiles/round6_timeline_gaps.txt) -- and a
second, phase-shifted chain fills that: measured +2.6 % frames/s with two RANKS on one GPU in round 5.  `ChainSet` is the same inside one
process: every chain is its own `ControlAnimationPipeline` (own sampler object, own captured hipGraph and static buffers) and its own
`MultiControlNetResidualsPipeline` (own control-image tensors and side streams) on its own HIP stream and host thread; the MODELS -- weight
arenas, and one per-window cache slot per chain inside them (HipModelMixin._slots) -- are shared.  A window's latency doubles; use it where
throughput is what counts.  Results are those of the sequential run, bit for bit (tests/test_chains_gpu.py): a window's kernels do not
depend on what else is in flight.

The reference has no counterpart (single device, one window at a time: scripts/vid2vid.py:168-268).

Limits: samplers that draw from torch's GLOBAL generator (the in-tree LCM of `use_lcm=1`, reference :1601) interleave their draws across
threads -- pass such jobs a `generator` or run them on one chain.
"""
from __future__ import annotations

import copy
import threading
import time
from typing import Any, Dict, List, Optional, Sequence

import torch

from .controlanimation_pipeline import ControlAnimationPipeline
from .controlresiduals_pipeline import MultiControlNetResidualsPipeline

_PIPE_FLAGS = ("use_hip_graph", "window_graph", "overlap_controlnet", "fuse_controlnet_adds", "steps_in_flight", "pace_wait", "pace_poll_s",
               "pace_timeout_s", "single_host_thread", "record_eps")


def clone_pipeline(pipe: ControlAnimationPipeline) -> ControlAnimationPipeline:
    """A second loop over the SAME models: own sampler instance, own graph / noise state, the flags of `pipe`."""
    twin = ControlAnimationPipeline(vae=pipe.vae, text_encoder=pipe.text_encoder, tokenizer=pipe.tokenizer, unet=pipe.unet,
                                    scheduler=copy.deepcopy(pipe.scheduler))
    twin.device = pipe.device
    twin.ip_adapter = pipe.ip_adapter
    for f in _PIPE_FLAGS:
        setattr(twin, f, getattr(pipe, f))
    return twin


def clone_residuals_pipeline(cn: Optional[MultiControlNetResidualsPipeline]) -> Optional[MultiControlNetResidualsPipeline]:
    if cn is None:
        return None
    twin = MultiControlNetResidualsPipeline(cn.controlnet_names, cn.cond_scale, cn.use_lcm, controlnets=cn.controlnets, device=cn.device,
                                            annotators=cn.annotators)
    twin.ip_adapter = cn.ip_adapter
    twin.detect_identical_halves = cn.detect_identical_halves
    return twin


class one_host_thread:
    """torch's intra-op thread count is process-wide: with several chains in flight it is set to 1 ONCE around all of them (and restored), and
    the pipelines' own per-call switch (`single_host_thread`: read, set 1, restore) is off meanwhile -- two threads restoring each other's
    readings would leave the process at one thread."""

    def __init__(self, pipes):
        self.pipes, self.flags, self.prev = list(pipes), [], None

    def __enter__(self):
        self.prev = torch.get_num_threads()
        self.flags = [p.single_host_thread for p in self.pipes]
        for p in self.pipes:
            p.single_host_thread = False
        if self.prev != 1:
            torch.set_num_threads(1)
        return self

    def __exit__(self, *exc):
        for p, f in zip(self.pipes, self.flags):
            p.single_host_thread = f
        if torch.get_num_threads() != self.prev:
            torch.set_num_threads(self.prev)
        return False


class ChainSet:
    def __init__(self, pipe: ControlAnimationPipeline, cn: Optional[MultiControlNetResidualsPipeline] = None, chains: int = 2):
        if chains < 1:
            raise ValueError("chains must be >= 1")
        self.pipes = [pipe] + [clone_pipeline(pipe) for _ in range(chains - 1)]
        self.cns = [cn] + [clone_residuals_pipeline(cn) for _ in range(chains - 1)]
        self.streams = [torch.cuda.Stream(device=pipe.device) for _ in range(chains)]

    def map(self, jobs: Sequence[Dict[str, Any]], prime: bool = True, timeout_s: Optional[float] = None) -> List[Any]:
        """Runs `pipe(**job)` for every job -- chain k takes jobs k, k + chains, ... -- and returns the results in job order.  prime: the
        first job of every chain runs alone (its eager step and the hipGraph capture; pass False once every chain has captured), the rest
        concurrently.  timeout_s: give up (TimeoutError) when the concurrent part takes longer -- the worker threads are daemons."""
        with one_host_thread(self.pipes):
            return self._map(jobs, prime, timeout_s)

    def _map(self, jobs, prime, timeout_s):
        n = len(self.pipes)
        results: List[Any] = [None] * len(jobs)
        errors: List[Optional[BaseException]] = [None] * n
        dev = self.pipes[0].device

        def run(k: int, idxs):
            try:
                torch.cuda.set_device(dev)
                with torch.cuda.stream(self.streams[k]):
                    for i in idxs:
                        kw = dict(jobs[i])
                        if self.cns[k] is not None:
                            kw["multicontrolnetresiduals_pipeline"] = self.cns[k]
                        results[i] = self.pipes[k](**kw)
                self.streams[k].synchronize()
            except BaseException as exc:  # noqa: BLE001 -- re-raised in the caller's thread
                errors[k] = exc

        first = 0
        if prime:
            for k in range(min(n, len(jobs))):  # priming: one chain at a time
                run(k, [k])
                if errors[k] is not None:
                    raise errors[k]
            first = n
        threads = [threading.Thread(target=run, args=(k, list(range(k + first, len(jobs), n))), name=f"chain{k}", daemon=True) for k in range(n)]
        for t in threads:
            t.start()
        deadline = None if timeout_s is None else time.monotonic() + float(timeout_s)
        for t in threads:
            t.join(None if deadline is None else max(0.0, deadline - time.monotonic()))
            if t.is_alive():
                raise TimeoutError(f"chain thread {t.name} did not finish within {timeout_s} s")
        for e in errors:
            if e is not None:
                raise e
        return results
