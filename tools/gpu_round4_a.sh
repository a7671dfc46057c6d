#!/bin/bash
# round 4, first call: the whole GPU suite on the tree as it stands + the default bench line (new sub-records) at the driver's K
set -o pipefail
mkdir -p gpurun_out/r4a
python -m pytest tests -m gpu -x -q > gpurun_out/r4a/gpu_tests.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r4a/gpu_tests.log
tail -5 gpurun_out/r4a/gpu_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r4a/bench_k20.json 2> gpurun_out/r4a/bench_k20.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4a/bench_k20.json"))
print("headline", d["value"], d["ms_per_step"], d.get("device_ms_per_step"), d["roofline"]["frac"])
for k in ("hbm_streaming", "config4", "literal", "config2"):
    r = d.get(k, {})
    print(k, r.get("value"), r.get("ms_per_step"), r.get("device_ms_per_step"), r.get("roofline", {}).get("frac"), (r.get("cpu_baseline") or {}).get("value"))
print("rc", d["config4"].get("roofline_collision"))
print("standin", {k: v for k, v in d.get("sharded_rank_standin", {}).items() if k != "runs"})
print("config5", d.get("config5"))
PY
