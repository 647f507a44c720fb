"""bench.py command-line contract that can be checked without a GPU: `--gpus N` must never silently run fewer
ranks than it reports (`bench.py --gpus 8` used to print n_gpus = 1)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_gpus_flag_refuses_to_run_with_fewer_visible_gpus():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a box with fewer than 2 GPUs")
    r = _run(["--gpus", "2"])
    assert r.returncode != 0
    assert "only" in r.stderr and "GPU(s) visible" in r.stderr
    assert '"n_gpus"' not in r.stdout


def test_gpus_flag_must_match_world_size():
    r = _run(["--gpus", "4"], {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in r.stderr
    assert '"n_gpus"' not in r.stdout
