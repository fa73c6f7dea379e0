"""`python bench.py --gpus N` must start N ranks by itself (VERDICT r1 item 1 / ADVICE): CPU rehearsal over
gloo with the hot path replaced by a sleep (`--plumbing-only`; the product path has no CPU fallback)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(CA_DIST_BACKEND="gloo", **(env_extra or {}))
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus2_spawns_two_ranks_and_broadcasts():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "0", "--plumbing-only"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2
    assert out["config"]["weight_broadcast_bytes"] == 2 * (1 << 20)
    assert out["dry_run"] is True and out["value"] is None  # a rehearsal can never be read as a measurement


def test_gpus1_stays_single_process():
    r = _run(["--gpus", "1", "--steps", "2", "--warmup", "0", "--plumbing-only"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["config"]["weight_broadcast_bytes"] == 0


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4", "--plumbing-only"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "MASTER_PORT": "29999"})
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)


def test_failing_rank_fails_the_launch():
    # without a GPU the real hot path refuses to run: every rank exits non-zero and so must the parent
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present: the real path would run")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
