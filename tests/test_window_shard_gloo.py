"""CPU, world_size 2 (gloo): the multi-GPU path -- weight-arena broadcast bit-equality, window
round-robin + ordered gather, and rank-independence of a window's result."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from controlanimate_amd import window_shard as WS
    r, w, _ = WS.init_distributed("gloo")
    assert (r, w) == (rank, world)
    # 1) packed arenas: rank 0 holds the weights, the others garbage; after the broadcast all equal
    g = torch.Generator().manual_seed(11)
    a0 = torch.randint(0, 255, (4096 + 256,), dtype=torch.uint8, generator=g)
    a1 = torch.randint(0, 255, (1024,), dtype=torch.uint8, generator=g)
    bufs = [a0.clone(), a1.clone()] if rank == 0 else [torch.zeros_like(a0), torch.full_like(a1, 7)]
    moved = WS.broadcast_weights(bufs)
    assert moved == a0.numel() + a1.numel()
    assert torch.equal(bufs[0], a0) and torch.equal(bufs[1], a1)
    # 1b) a rank whose buffer list differs (a skeleton that packs differently) makes EVERY rank raise -- nobody hangs
    bad = [a0.clone(), a1.clone()] if rank == 0 else [a0.clone(), torch.zeros(1000, dtype=torch.uint8)]
    with pytest.raises(RuntimeError, match="first difference at index 1"):
        WS.broadcast_weights(bad)
    with pytest.raises(RuntimeError, match="holds"):
        WS.broadcast_weights([a0.clone()] if rank == 0 else [a0.clone(), a1.clone()])
    # 2) windows: deterministic function of the window index only -> any rank computes the same thing
    def run_window(i):
        gg = torch.Generator().manual_seed(1000 + i)
        return torch.randn(1, 4, 2, 2, 2, generator=gg) * (i + 1)
    res = WS.run_sharded(5, run_window, rank, world)
    if rank == 0:
        assert len(res) == 5
        for i, t in enumerate(res):
            assert torch.equal(t, run_window(i)), i
        ret["ok"] = True
    else:
        assert res is None
    assert WS.windows_for_rank(5, rank, world) == ([0, 2, 4] if rank == 0 else [1, 3])
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_window_sharding_world2():
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret.get("ok") is True


def test_single_process_paths_are_noops():
    from controlanimate_amd import window_shard as WS
    assert WS.broadcast_weights([torch.zeros(8, dtype=torch.uint8)]) == 0
    out = WS.run_sharded(3, lambda i: torch.tensor([i]), 0, 1)
    assert [int(t) for t in out] == [0, 1, 2]
