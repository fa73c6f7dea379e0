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


def test_committed_bench_line_keeps_the_driver_contract():
    """profiles/round3_bench.json is a `python bench.py` line of the final code: the keys the driver reads are there and the
    numbers are consistent with each other (value = frames per window / (steps x s per step), roofline.frac = achieved / peak,
    HBM traffic per launch not below what the dominant kernel must move at least once)."""
    path = os.path.join(ROOT, "profiles", "round3_bench.json")
    d = json.load(open(path))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"] == "frames_per_sec" and d["unit"] == "frames/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    cfg = d["config"]
    assert "workload" in cfg and "model" not in cfg and cfg["baseline_config"] == 2
    fps = cfg["frames_per_window"] / (cfg["steps_per_window"] * d["ms_per_step"] * 1e-3)
    assert abs(fps - d["value"]) / d["value"] < 1e-3
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-3 and 0 < r["frac"] < 1
    assert r["traffic"] is None or r["traffic"] > 0.5 * r["algorithmic_bytes_per_launch"]
    m = r["matrix_rate_vs_operand_data"]  # the matrix pipes' rate on real data on the same box
    assert m["vendor_gemm_8192x8192x8192"]["zero_data"] > m["vendor_gemm_8192x8192x8192"]["normal_data"] > 0
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1
    assert c["config2_step"]["sec_per_step"] > 100 * d["ms_per_step"] * 1e-3  # the CPU port beside it, never the target
