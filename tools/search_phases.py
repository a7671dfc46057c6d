"""Where a wave of the list-building query spends its time: python tools/search_phases.py [n_uavs] with the timing build
(bash tools/build_variants.sh collideflag "-DMRS_Q2_CLOCK=1"; MRS_SWARM_LIB=variants/...so): every wave leaves six 100-MHz stamps in
the list rows of its first UAV (start, own position there, bucket heads there, items emitted, sweeps done, end)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402
import mrs_multirotor_simulator_amd as M  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
st, cmd = bench.make_inputs(n, "position+collisions", 1234, 64.0)
g = M.Swarm(n, arith=M.ARITH_FAST)
g.construct(0, n, M.model_params("x500", ground_enabled=True))
g.set_state(0, n, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
g.debug_search_ms(reps=4)
count, nbr, cap, radius = g.debug_neighbour_lists()
first = np.nonzero(count == 6)[0]
lpu = int(os.environ.get("MRS_QUERY_LPU", "3" if n <= 500000 else "2"))  # (collide.hip query_lpu)
first = first[first % (64 // lpu) == 0]
ts = nbr[:6, first].astype(np.int64).T  # [waves, 6]
t0 = ts[:, 0].min()
names = ["own position", "bucket heads", "decide + emit", "sweeps", "lists + forces"]
print(f"{n} UAVs, {len(first)} waves; kernel span (first start to last end) {(ts[:, 5].max() - t0) / 100:.2f} us; starts spread over {(ts[:, 0].max() - t0) / 100:.2f} us")
for k in range(5):
    d = (ts[:, k + 1] - ts[:, k]) / 100.0
    print(f"  {names[k]:16s} mean {d.mean():6.2f} us   median {np.median(d):6.2f}   p95 {np.percentile(d, 95):6.2f}")
life = (ts[:, 5] - ts[:, 0]) / 100.0
print(f"  wave lifetime    mean {life.mean():6.2f} us   median {np.median(life):6.2f}   p95 {np.percentile(life, 95):6.2f}")
h, edges = np.histogram((ts[:, 0] - t0) / 100.0, bins=10)
print("  wave starts per tenth of the start window:", h.tolist())
