#!/usr/bin/env python3
"""Where a boundary launch of the split sharded tick spends its time: wall-clock stamps (100 MHz) taken by the first lane of each of
its blocks at the joints of the kernel (a -DMRS_TS=1 build: tools/build_variants.sh stepflag "-DMRS_TS=1", MRS_SWARM_LIB=variants/...).
Runs tools/sharded_rank_cost.py's scenario and prints, for the LAST boundary launch, the median over blocks of every interval."""
import ctypes as C, os, runpy, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0], "125000", "8", "300", "20", "split"]
runpy.run_path(os.path.join(ROOT, "tools", "sharded_rank_cost.py"), run_name="__main__")
import mrs_multirotor_simulator_amd as M
lib = M.load_library()
buf = (C.c_ulonglong * (128 * 16))()
lib.mrs_debug_ts_read.argtypes = [C.POINTER(C.c_ulonglong)]
assert lib.mrs_debug_ts_read(buf) == 0
t = np.array(buf, dtype=np.float64).reshape(128, 16)[:, :10]
t = t[t[:, 0] > 0]
t = t[t[:, 9] > t[:, 0]]
names = ["entry->stall decision (headers, control words)", "->partner gather issued", "->state arrived (cascade starts)", "->cascade done", "->collision evaluation done",
         "->motors done", "->RK4 + post-step done", "->stores issued (publish)", "->stores drained, epoch stored"]
print(f"{len(t)} blocks; kernel span {((t[:, 9].max() - t[:, 0].min()) / 100):.1f} us; block start spread {((t[:, 0].max() - t[:, 0].min()) / 100):.1f} us")
for k, nm in enumerate(names):
    d = (t[:, k + 1] - t[:, k]) / 100.0
    print(f"  {nm:52s} median {np.median(d):6.2f} us   max {d.max():6.2f}")
print(f"  one block, entry to end: median {np.median(t[:, 9] - t[:, 0]) / 100:.2f} us")
