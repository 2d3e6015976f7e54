"""bench.py command-line contract that needs no GPU: --gpus is never ignored."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env_over):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_over)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, timeout=300)


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "2"], WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    assert r.returncode == 2 and b"WORLD_SIZE=3" in r.stderr and not r.stdout.strip()
    r = _run(["--gpus", "1"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode == 2 and not r.stdout.strip()


def test_gpus_without_launcher_spawns_ranks_and_fails_loudly_without_gpus():
    """no GPU here: the self-launched ranks must fail (no CPU path), and no JSON line may appear"""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-only check")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and not [l for l in r.stdout.decode().splitlines() if l.strip().startswith("{")]
    assert b"needs a GPU" in r.stderr


def test_bad_gpus_value():
    assert _run(["--gpus", "0"]).returncode == 2
