"""Multi-GPU execution: independent 16-frame windows sharded over the GPUs of one node.

The reference is single-device and processes windows sequentially (scripts/vid2vid.py:168-268).
Windows are independent denoising problems once their inputs exist, so the MI355X
design is: one process per GPU, replicated weights, rank r takes windows r, r+W, r+2W, ...; NO
collective in the denoising loop.  The only communication is
  * a one-time RCCL broadcast of the packed weight arenas from rank 0 (bf16/fp16 UNet3D 2.6 GB +
    0.7 GB per ControlNet) -- `broadcast_weights`; each arena is ONE contiguous buffer so this is a
    few large xGMI transfers, not thousands of small ones;
  * an optional gather of the per-window results on rank 0 (latents, 1 MB per window).
Overlap blending / colour matching of neighbouring windows stays on the host, in window order, as
in the reference (vid2vid.py:216-226).
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def init_distributed(backend: Optional[str] = None, single_rank_group: bool = False) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment; initialises the default group when
    WORLD_SIZE > 1 ("nccl" = RCCL on ROCm when a GPU is present, "gloo" on CPU test runs).

    `single_rank_group=True` initialises it for WORLD_SIZE == 1 as well: the collectives of this module then really run
    (RCCL communicator of one rank: init, device broadcast, object gather, barrier) instead of returning early -- the 1-GPU
    proof that the `nccl` branch works (tests/test_window_shard_nccl_gpu.py).

    Rendezvous: MASTER_ADDR defaults to 127.0.0.1 (one node).  MASTER_PORT has NO fixed default: a launcher (torchrun,
    `bench.py --gpus N`) chooses it and hands the same value to every rank; a single-rank group picks a free port itself.  (A
    hard-wired 29500 collides as soon as two jobs share a host.)"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if (world > 1 or single_rank_group) and not dist.is_initialized():
        if backend is None:
            # CA_DIST_BACKEND=gloo: several ranks sharing ONE GPU (rehearsal of the multi-rank flow on a 1-GPU box)
            backend = os.environ.get("CA_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world > 1:
                raise RuntimeError("init_distributed: WORLD_SIZE > 1 but MASTER_PORT is not set -- every rank must be given the same "
                                   "port by its launcher (torchrun --master-port, bench.py --gpus N); there is no fixed default")
            os.environ["MASTER_PORT"] = str(_free_port())
        kw = {}
        if backend == "nccl":
            dev = torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1))
            torch.cuda.set_device(dev)
            kw["device_id"] = dev  # binds the communicator to this GPU: barrier() needs no device_ids guess
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def _group_active() -> bool:
    """True when collectives should run: an initialised default group (of any size, a single-rank group included)."""
    return dist.is_available() and dist.is_initialized()


def window_plan(num_frames: int, frame_count: int, overlap_length: int) -> List[Tuple[int, int]]:
    """[start, end) input-frame ranges of the sliding windows of vid2vid.py:168-189: the first window
    takes `frame_count` frames, every later one re-feeds the last `overlap_length` input frames and
    adds frame_count - overlap_length new ones; a short tail window is kept (the reference processes
    whatever frames remain)."""
    if frame_count <= overlap_length:
        raise ValueError("frame_count must exceed overlap_length")
    plan = []
    start = 0
    while start < num_frames:
        end = min(start + frame_count, num_frames)
        plan.append((start, end))
        if end >= num_frames:
            break
        start = end - overlap_length
    return plan


def windows_for_rank(num_windows: int, rank: int, world: int) -> List[int]:
    """Round-robin assignment: rank r takes windows r, r+world, ..."""
    return list(range(rank, num_windows, world))


def broadcast_weights(buffers: Sequence[torch.Tensor], src: int = 0) -> int:
    """Broadcasts the packed weight arenas (one contiguous uint8 tensor each) from `src`.
    Returns the number of bytes moved. No-op for a single process."""
    if not _group_active():
        return 0
    # every rank must hand over the same list -- a rank whose skeleton packs one tensor more, less or differently would
    # otherwise hang in a broadcast or run on garbage weights: compare (numel, dtype) manifests first, on ALL ranks, so that
    # every rank raises together instead of some waiting forever
    manifest = [(int(b.numel()), str(b.dtype)) for b in buffers]
    everyone: List[Optional[list]] = [None] * dist.get_world_size()
    dist.all_gather_object(everyone, manifest)
    for r, other in enumerate(everyone):
        if other != everyone[src]:
            diff = next((i for i, (x, y) in enumerate(zip(other, everyone[src])) if x != y), min(len(other), len(everyone[src])))
            raise RuntimeError(f"broadcast_weights: rank {r} holds {len(other)} weight buffers, rank {src} {len(everyone[src])}; "
                               f"first difference at index {diff}: {other[diff] if diff < len(other) else None} vs "
                               f"{everyone[src][diff] if diff < len(everyone[src]) else None}")
    total = 0
    for buf in buffers:
        dist.broadcast(buf, src=src)
        total += buf.numel() * buf.element_size()
    return total


def gather_window_results(local: List[Tuple[int, torch.Tensor]], num_windows: int, dst: int = 0) -> Optional[List[torch.Tensor]]:
    """Collects (window index, result) pairs on `dst`, returned in window order (None elsewhere)."""
    if not _group_active():
        out: List[Optional[torch.Tensor]] = [None] * num_windows
        for i, t in local:
            out[i] = t
        return out  # type: ignore[return-value]
    world, rank = dist.get_world_size(), dist.get_rank()
    payload = [(i, t.detach().cpu()) for i, t in local]
    gathered: List[Optional[list]] = [None] * world if rank == dst else None  # type: ignore[assignment]
    dist.gather_object(payload, gathered, dst=dst)
    if rank != dst:
        return None
    out = [None] * num_windows
    for part in gathered:
        for i, t in part:
            out[i] = t
    return out  # type: ignore[return-value]


def broadcast_tensor(t: torch.Tensor, src: int = 0) -> torch.Tensor:
    """In-place broadcast of one (device or host) tensor from `src`: the fixed IP-Adapter image prompt of a sharded video."""
    if _group_active():
        dist.broadcast(t, src=src)
    return t


def barrier() -> None:
    if _group_active():
        dist.barrier()


def blend_overlap(prev_tail: torch.Tensor, cur_head: torch.Tensor) -> torch.Tensor:
    """Linear cross-fade of the overlap frames (vid2vid.py:225-226): frame i of n takes weight
    (n - i - 0.5)/n from the previous window. Tensors: [n, ...] decoded frames."""
    n = prev_tail.shape[0]
    w = ((n - torch.arange(n, dtype=torch.float32) - 0.5) / n).view(n, *([1] * (prev_tail.dim() - 1))).to(prev_tail.device)
    return cur_head * (1.0 - w) + prev_tail * w


def run_sharded(num_windows: int, run_window: Callable[[int], torch.Tensor], rank: int, world: int,
                gather: bool = True) -> Optional[List[torch.Tensor]]:
    """Runs `run_window(i)` for this rank's windows and gathers the results on rank 0."""
    local = [(i, run_window(i)) for i in windows_for_rank(num_windows, rank, world)]
    return gather_window_results(local, num_windows) if gather else None
