#!/usr/bin/env python3
"""Where a wave of a sharded rank's launch spends its time, for the launch kind the library variant stamps: wall-clock stamps (100 MHz)
by the first lane of sampled blocks at the joints of the step kernel (tools/build_variants.sh stepflag "-DMRS_TS=1 -DMRS_TS_PART=<p>":
2 boundary launch (default), 1 interior launch, 0 full launch of the serial form — run that one with MRS_SHARD_SPLIT=0).
Runs tools/sharded_interior_alone.py's scenario and prints, for the last stamped launch, the median over blocks of every interval.
usage: MRS_SWARM_LIB=$PWD/variants/libmrs_stepflag__DMRS_TS_1__DMRS_TS_PART_1.so launch_phases.py [latency_us] [ticks]"""
import ctypes as C, os, runpy, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
runpy.run_path(os.path.join(ROOT, "tools", "sharded_interior_alone.py"), run_name="__main__")
import mrs_multirotor_simulator_amd as M
lib = M.load_library()
buf = (C.c_ulonglong * (128 * 16))()
lib.mrs_debug_ts_read.argtypes = [C.POINTER(C.c_ulonglong)]
assert lib.mrs_debug_ts_read(buf) == 0
t = np.array(buf, dtype=np.float64).reshape(128, 16)[:, :10]
rows = np.arange(128)[t[:, 0] > 0]
t = t[t[:, 0] > 0]
last = 9 if (t[:, 9] > t[:, 0]).sum() > len(t) // 2 else 7  # (a full launch publishes no epoch word: stamps 8 and 9 stay empty)
rows = rows[t[:, last] > t[:, 0]]
t = t[t[:, last] > t[:, 0]]
names = ["entry->stall decision (headers, control words)", "->partner gather issued (polls of layer-1 / boundary blocks)", "->state arrived (cascade starts)", "->cascade done",
         "->collision evaluation done", "->motors done", "->RK4 + post-step done", "->stores issued (publish)", "->stores drained, epoch stored"][:last]
print(f"{len(t)} blocks; span of the sampled blocks {((t[:, last].max() - t[:, 0].min()) / 100):.1f} us; block start spread {((t[:, 0].max() - t[:, 0].min()) / 100):.1f} us")
for k, nm in enumerate(names):
    d = (t[:, k + 1] - t[:, k]) / 100.0
    print(f"  {nm:62s} median {np.median(d):6.2f} us   p90 {np.percentile(d, 90):6.2f}   max {d.max():6.2f}")
print(f"  one block, entry to last stamp: median {np.median(t[:, last] - t[:, 0]) / 100:.2f} us   p90 {np.percentile(t[:, last] - t[:, 0], 90) / 100:.2f}")
life = (t[:, last] - t[:, 0]) / 100.0
t0 = t[:, 0].min()
print("  slowest sampled blocks (row = block / 15 for the big launches): row, start, life, phases")
for k in np.argsort(-life)[:10]:
    print(f"    row {rows[k]:3d}  start {(t[k, 0] - t0) / 100:5.2f}  life {life[k]:6.2f}  " + " ".join(f"{(t[k, j + 1] - t[k, j]) / 100:5.2f}" for j in range(last)))
print("  life by row (every 8th):", " ".join(f"{rows[k]}:{life[k]:.1f}" for k in range(0, len(t), 8)))
