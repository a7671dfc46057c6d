#!/usr/bin/env python3
"""Timing-only ablation of k_query (collide.hip): stop after phase A / B / C to see where the time goes (results wrong)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mrs_multirotor_simulator_amd", "csrc"); OBJ = os.path.join(ROOT, "mrs_multirotor_simulator_amd", "build")
sys.path.insert(0, ROOT)
from mrs_multirotor_simulator_amd import build
build.build_library()
os.makedirs("/tmp/abl", exist_ok=True)
for stop in (1, 3, 0):
    o = f"/tmp/abl/collide_{stop}.o"
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off", f"-DMRS_QUERY_STOP={stop}", "-c",
                           os.path.join(CSRC, "collide.hip"), "-o", o])
    lib = f"/tmp/abl/lib_{stop}.so"
    subprocess.check_call(["hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib, o] + [os.path.join(OBJ, f) for f in
                          ("step_kernel_literal.o", "step_kernel_fast.o", "outputs.o", "host_api.o", "tick_single.o", "tick_sharded.o", "transport_rccl.o", "transport_local.o", "transport_peer.o")])
    env = dict(os.environ, MRS_SWARM_LIB=lib, TMPDIR="/tmp")
    d = f"/tmp/abl/prof_{stop}"
    subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py"),
                    "--steps", "100", "--warmup", "10", "--no-cpu-baseline", "--workload", "position+collisions"], env=env, capture_output=True)
    for root, _, files in os.walk(d):
        for f in files:
            if f.endswith("kernel_stats.csv"):
                for ln in open(os.path.join(root, f)):
                    if "k_query" in ln:
                        print("stop after phase", stop or "none", ": k_query avg ns", ln.split('",')[1].split(",")[2])
