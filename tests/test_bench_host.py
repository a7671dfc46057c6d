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
    # f550 hexarotors (BASELINE configs[1]) and the eight-motor airframe: SURVEY 8d's 908 B / 940 B
    assert bench.bytes_per_uav_step("position", 6) == 908 and bench.bytes_per_uav_step("position", 8) == 940
    assert bench.bytes_moved_per_uav_step("position", 6) < 908


def test_config2_inputs_are_the_tmux_grid():
    import numpy as np
    import bench
    st, cmd = bench.make_inputs(400, "config2", seed=0)
    assert st["x"].shape == (400, 3) and np.all(st["x"][:, 2] == 0) and st["x"][:, 0].max() == 76.0 and np.all(st["heading"] == 0)
    d = np.sort(np.unique(st["x"][:, 0]))
    assert np.all(np.diff(d) == 4.0)  # 20 x 20, 4 m pitch (tmux/standalone_400_uavs/custom_configs/simulator.yaml)
    assert np.all(np.abs(cmd[:, :2]) <= 40) and cmd[:, 2].min() >= 2 and cmd[:, 2].max() <= 20 and np.all(np.abs(cmd[:, 3]) <= 3.14)


def test_mean_search_candidates_counts_the_27_cell_neighbourhood():
    import numpy as np
    import bench
    x = np.array([[0.1, 0.1, 0.1], [0.2, 0.2, 0.2], [1.5, 0.1, 0.1], [10.0, 10.0, 10.0]])  # cells of edge 1: (0,0,0) x2, (1,0,0), far away
    assert bench.mean_search_candidates(x, 1.0) == (2 + 2 + 2 + 0) / 4


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


def test_guarded_peer_record_turns_a_failed_child_run_into_a_record():
    """bench.py's `config5_peer` leg runs in child processes of rank 0; whatever happens to them (here: no GPU at all) becomes an
    {"error": ...} record, never an exception or an exit status of the parent run"""
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the child run would succeed")
    import bench
    sys.argv = ["bench.py", "--gpus", "1", "--steps", "2", "--warmup", "0", "--config5-timeout", "200"]
    args = bench.parse()
    rec = bench.peer_leg_in_children(args)
    assert "error" in rec and rec.get("transport") == "peer", rec


def test_cell_order_is_a_permutation_that_follows_space():
    """mrs_cell_order (host only): a permutation; UAVs that follow each other in it are close in space; unusable positions go last"""
    import numpy as np
    import mrs_multirotor_simulator_amd as M
    rng = np.random.default_rng(8)
    x = rng.uniform(0, 200, (20000, 3)) * [1, 1, 0.2]
    x[123] = np.nan
    o = M.cell_order(x)
    assert sorted(o.tolist()) == list(range(len(x))) and o[-1] == 123
    xs = x[o[:-1]]
    step_sorted = np.linalg.norm(np.diff(xs, axis=0), axis=1)
    step_random = np.linalg.norm(np.diff(x[:-1][x[:-1, 0] == x[:-1, 0]], axis=0), axis=1)
    assert np.median(step_sorted) < 0.1 * np.median(step_random)
    assert np.array_equal(M.cell_order(x, 2.25), o)  # (the default cell edge)


def test_explain_prints_the_notes_without_a_gpu():
    r = _run("--explain")
    import json
    notes = json.loads(r.stdout)
    assert r.returncode == 0 and {"timing", "roofline", "roofline_collision", "sub_records", "rehearsal"} <= set(notes)
