"""`bench.py --gpus N` end to end on the ONE GPU of the test box: MRS_BENCH_REHEARSAL=1 puts every rank on cuda:0
with a gloo process group and runs everything else as the 8-GPU node will — the torch.distributed.run child started before the
parent touches the GPU, one process per rank, barrier + MAX over ranks, rank 0's ONE JSON line, the config-5 leg through
mrs_swarm_tick_sharded_n on every rank (collective: host all-gather over gloo, RCCL refuses two ranks on a device), the guarded
peer-window child run, and a LOST rank turning into `config5.error` + a non-zero exit — never a hang.  Exercises code, measures nothing."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
DROP = ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK")


def _bench(*args, timeout=600, **env):
    e = {k: v for k, v in os.environ.items() if k not in DROP and not k.startswith("TORCHELASTIC_")}
    e.update(MRS_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **env)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout, env=e)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, lines, time.time() - t0


COMMON = ["--steps", "20", "--warmup", "5", "--uavs", "20000", "--config5-uavs", "200000", "--min-measure-ms", "5"]


def test_bench_gpus_2_rehearsal_prints_one_line_with_both_config5_records():
    r, lines, _ = _bench("--gpus", "2", *COMMON, "--config5-timeout", "200")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout[-2000:]
    assert len(lines[0]) < 8000  # the driver keeps the last 8 KB of stdout: every number must be in it
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_devices"] == 1 and "rehearsal" in d and d["steps"] == 20 and d["warmup"] == 5
    assert d["scaling"] == "weak" and d["config"]["uavs_per_gpu"] == 20000
    assert d["value"] == pytest.approx(2 * 20000 * 20 / (d["ms_per_step"] * 1e-3 * 20), rel=1e-9)  # whole job: both ranks' UAVs
    c5 = d["config5"]
    assert "error" not in c5, c5
    assert c5["n_gpus"] == 2 and c5["n_devices"] == 1 and c5["n_total"] == 200000 and c5["uavs_per_rank"] == 100000
    assert c5["rccl_ranks"] == 0 and "gloo" in c5["transport"]  # reported truthfully: RCCL was not the collective here
    assert c5["export_set_of_rank0"] > 0 and c5["search_ticks"] >= 1 and c5["collective_bytes_per_rank_per_tick"] > 0
    peer = d["config5_peer"]  # the guarded child run: a record of its own, or its failure written into it — never missing
    assert peer.get("transport") == "peer", peer
    if "error" not in peer:
        assert peer["n_gpus"] == 2 and peer["rccl_ranks"] == 0 and peer["export_set_of_rank0"] > 0


def test_bench_rehearsal_a_lost_rank_becomes_config5_error_not_a_hang():
    r, lines, el = _bench("--gpus", "2", *COMMON, "--config5-timeout", "90", "--config5-peer", "off", MRS_BENCH_KILL_RANK="1", timeout=400)
    assert r.returncode != 0
    assert el < 300, f"took {el:.0f} s"
    assert len(lines) == 1, (r.stdout[-1000:], r.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0           # the headline leg had completed on both ranks: its line stands
    assert "error" in d["config5"], d["config5"]           # ... with the failure of the second leg recorded
