#!/usr/bin/env python3
"""Per-phase durations inside k_query (collide.hip) from in-kernel timestamps: builds a -DMRS_QUERY_CLOCK library in /tmp whose
query writes the A/B/C phase durations (10-ns ticks of the constant clock) into the force columns.  Timing aid only."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mrs_multirotor_simulator_amd", "csrc"); OBJ = os.path.join(ROOT, "mrs_multirotor_simulator_amd", "build")
sys.path.insert(0, ROOT)
subprocess.check_call([sys.executable, "-m", "mrs_multirotor_simulator_amd.build"], cwd=ROOT, stdout=subprocess.DEVNULL)
os.makedirs("/tmp/qph", exist_ok=True)
o = "/tmp/qph/collide.o"
subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off", "-DMRS_QUERY_CLOCK=" + os.environ.get("QPH_CLOCK", "1")] + sys.argv[2:] + ["-c",
                       os.path.join(CSRC, "collide.hip"), "-o", o])
lib = "/tmp/qph/libmrs_swarm_clock.so"
subprocess.check_call(["hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib, o] + [os.path.join(OBJ, f) for f in
                      ("step_kernel_literal.o", "step_kernel_fast.o", "outputs.o", "host_api.o", "tick_single.o", "tick_sharded.o", "transport_rccl.o", "transport_local.o", "transport_peer.o")])
os.environ["MRS_SWARM_LIB"] = lib
if os.environ.get("QPH_LISTS", "0") == "1":
    sys.exit("the list-building query has its own stamps: tools/search_phases.py (this tool times the plain search-every-tick query)")
LISTS = False
os.environ["MRS_NEIGHBOUR_LISTS"] = "1" if LISTS else "0"
import mrs_multirotor_simulator_amd as M
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
rng = np.random.default_rng(4)
side = (64.0 * n) ** (1.0 / 3.0)
pos = rng.uniform(0, 1, (n, 3)) * [side * 2, side * 2, side / 4] + [0, 0, 5]
g = M.Swarm(n)
g.construct(0, n, M.model_params("x500"), pos, np.zeros(n))
for _ in range(5):
    if LISTS:
        st0 = g.get_state(0, 1)
        g.set_state(0, 1, st0["x"], st0["v"], st0["R"], st0["omega"], st0["motor_rpm"])
    g.handle_collisions(True, False, 100.0)
f = g.get_external_force()[::64]  # one record per wavefront
print("per-wavefront phase durations, us (mean / p50 / p95 / max):")
names = ("A heads+clear", "B list build", "C sweep") if os.environ.get("QPH_CLOCK", "1") == "1" else ("A+B", "C sweep", "D hits+lists")
for k, name in enumerate(names):
    d = f[:, k] * 0.01
    print(f"  {name:14s} {d.mean():7.2f} {np.percentile(d, 50):7.2f} {np.percentile(d, 95):7.2f} {d.max():7.2f}")
