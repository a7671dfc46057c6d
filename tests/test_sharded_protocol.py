"""The host-side decisions of the sharded collision tick (csrc/sharded_protocol.h — what tick_sharded.hip and tick_single.hip call)
driven through a multi-rank model of launches, collectives and pinned mirror words under random host skew, WITHOUT a GPU: all ranks
must issue the same launches whatever the interleaving, no launch may run beyond a stall index, and a protocol whose reports travel
slower than its constants allow must be caught (negative control).  The device side of the same protocol runs on the GPU box:
tests/test_sharded_chaos_gpu.py, tests/cpp/sharded_tick_test.cpp."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharded_protocol_model_cpp():
    exe = os.path.join(ROOT, "tests", "cpp", "sharded_protocol_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", os.path.join(ROOT, "tests", "cpp", "sharded_protocol_test.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    for tag in ("functions", "model", "negative_control"):
        assert f"ok {tag}" in out.stdout, out.stdout
