#!/usr/bin/env python3
"""When the launches of a SHORT run of split steps really start and end on the device: a library built with -DMRS_TS_STEP=1
(tools/build_variants.sh stepflag "-DMRS_TS_STEP=1") lets the first and the last block of every plain step launch stamp the 100-MHz
wall clock at entry and behind their stores; this runs regions of K steps of the headline workload (synchronised on both sides, as
bench.py times them) and prints the last region launch by launch — no tracer in the way.
usage: MRS_SWARM_LIB=$PWD/variants/libmrs_stepflag__DMRS_TS_STEP_1.so region_timeline.py [K]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import mrs_multirotor_simulator_amd as M

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 100_000
st, cmd = bench.make_inputs(n, "actuator", 3)
g = M.Swarm(n, arith=M.ARITH_FAST)
g.construct(0, n, M.model_params("x500", ground_enabled=True))
g.set_state(0, n, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
g.set_input(0, n, M.ACTUATOR_CMD, cmd)
g.step_n(0.001, 200)
g.synchronize()
walls = []
for rep in range(8):
    g.synchronize()
    t0 = time.perf_counter()
    g.step_n(0.001, K)
    t1 = time.perf_counter()
    g.synchronize()
    walls.append(((t1 - t0) * 1e6, (time.perf_counter() - t0) * 1e6))
lib = M.load_library()
buf = (C.c_ulonglong * 4096)()
cnt = C.c_uint()
lib.mrs_debug_ts_step_read_fast.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_uint)]
assert lib.mrs_debug_ts_step_read_fast(buf, C.byref(cnt)) == 0
t = np.array(buf, dtype=np.uint64).reshape(1024, 4)
total = cnt.value
per_region = 2 * K * 2  # launches x stamped blocks
idx = [(total - per_region + k) & 1023 for k in range(per_region)]
rows = t[idx]
rows = rows[np.argsort(rows[:, 0])]
t0 = float(rows[:, 0].min())
print(f"K = {K}: last region by the host: call returned after {walls[-1][0]:.1f} us, region {walls[-1][1]:.1f} us (median of 8: {np.median([w[1] for w in walls]):.1f})")
launches = {}
for r in rows:
    blk0, blk = int(r[2] >> np.uint64(32)), int(r[2] & np.uint64(0xFFFFFFFF))
    key = (blk0, None)
    launches.setdefault(blk0, []).append(((float(r[0]) - t0) / 100.0, (float(r[1]) - t0) / 100.0, blk))
for blk0 in sorted(launches):
    ev = sorted(launches[blk0])
    # pair first-block and last-block stamps of the same launch: consecutive in time
    print(f" stream of blocks from {blk0}: (first block in -> last stamped block out) per launch")
    first = [e for e in ev if e[2] == 0]
    last = [e for e in ev if e[2] != 0]
    for k, (a, b) in enumerate(zip(first, last)):
        print(f"   launch {k:2d}: first block in {a[0]:7.2f}  out {a[1]:7.2f} | last block in {b[0]:7.2f}  out {b[1]:7.2f} | launch span {max(a[1], b[1]) - a[0]:5.2f}")
print(f" device span of the region (first entry to last exit): {(float(rows[:, 1].max()) - t0) / 100.0:.1f} us")
