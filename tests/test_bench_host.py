"""bench.py's host logic that needs no GPU: byte accounting, and that it refuses to run without the hardware it measures."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _run(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=300, env=env)


def test_byte_accounting_matches_the_layout():
    import bench
    # state 25 doubles + F_ext 3 + command 4 + flags word + (position cascade) 24 PID doubles, read once and written once
    assert bench.BYTES_PER_UAV_STEP == {"actuator": 492, "position": 876}
    assert bench.BYTES_MOVED_PER_UAV_STEP == {"actuator": 412, "position": 796}
    assert bench.STATE_BYTES_PER_UAV == 692
    for key in ("actuator", "position"):
        assert bench.BYTES_MOVED_PER_UAV_STEP[key] < bench.BYTES_PER_UAV_STEP[key]


def test_refuses_more_gpus_than_the_machine_has():
    import torch
    have = torch.cuda.device_count()
    want = max(have + 1, 2)  # (--gpus 1 does not spawn anything)
    r = _run("--gpus", str(want), "--steps", "1", "--warmup", "0")
    assert r.returncode != 0 and f"this machine shows {have} GPU(s)" in r.stderr, r.stderr[-400:]
    assert r.stdout.strip() == ""  # no JSON line for a run that did not happen


def test_no_cpu_fallback_for_the_hot_path():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the refusal path is not reachable")
    r = _run("--steps", "1", "--warmup", "0", "--traffic", "off", "--no-cpu-baseline")
    assert r.returncode != 0 and "no CPU fallback" in r.stderr, r.stderr[-400:]
    assert r.stdout.strip() == ""
