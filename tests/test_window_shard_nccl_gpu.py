"""The RCCL ("nccl") branch of window_shard on ONE GPU: a single-rank process group -- communicator init bound to the device,
`broadcast_weights` of device uint8 arenas (manifest exchange through all_gather_object + one broadcast per arena), the object
gather of window results, a device-tensor broadcast, barrier -- run for real instead of returning early.  Multi-rank
correctness of the same calls is covered on CPU over gloo (tests/test_window_shard_gloo.py); what only a GPU box can show is
that the nccl backend initialises and moves device memory here."""
import subprocess
import sys
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import os, sys, torch
sys.path.insert(0, os.environ["CA_ROOT"])
import torch.distributed as dist
from controlanimate_amd import window_shard as WS
for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "CA_DIST_BACKEND"):
    os.environ.pop(k, None)
rank, world, local = WS.init_distributed(single_rank_group=True)
assert (rank, world, local) == (0, 1, 0)
assert dist.is_initialized() and dist.get_backend() == "nccl", dist.get_backend()
assert "MASTER_PORT" in os.environ and os.environ["MASTER_PORT"] != "29500"
g = torch.Generator(device="cuda").manual_seed(3)
arenas = [torch.randint(0, 256, (n,), device="cuda", dtype=torch.uint8, generator=g) for n in (64 << 20, 3 << 20, 257)]
before = [a.clone() for a in arenas]
moved = WS.broadcast_weights(arenas)
torch.cuda.synchronize()
assert moved == sum(a.numel() for a in arenas), moved
assert all(torch.equal(a, b) for a, b in zip(arenas, before))
t = torch.arange(24, device="cuda", dtype=torch.float32).view(2, 1, 4, 3)
WS.broadcast_tensor(t)
assert torch.equal(t.cpu(), torch.arange(24, dtype=torch.float32).view(2, 1, 4, 3))
res = WS.run_sharded(3, lambda i: torch.full((2, 2), float(i), device="cuda"), rank, world)
assert [float(r[0, 0]) for r in res] == [0.0, 1.0, 2.0] and all(r.device.type == "cpu" for r in res)
WS.barrier()
dist.destroy_process_group()
print("NCCL_SINGLE_RANK_OK", moved)
"""


def test_single_rank_nccl_group_runs_the_collectives():
    env = dict(os.environ, CA_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "NCCL_SINGLE_RANK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
