#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4f
timeout -k 10 500 python -m pytest tests/test_bench_launch_gpu.py tests/test_soak_cut_gpu.py -m gpu -q -x -s > gpurun_out/r4f/tests.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed|FAILED|^E  |config 4" gpurun_out/r4f/tests.log | tail -8
for ov in 1 0; do
  MRS_OVERLAP_TICKS=$ov timeout -k 10 200 python bench.py --workload position+collisions --steps 300 --warmup 100 --no-cpu-baseline --traffic off --sub-records off --config5 off 2>gpurun_out/r4f/bench_ov$ov.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('overlap $ov: wall %.2f us  device %.2f us per tick' % (d['ms_per_step']*1e3, d['device_ms_per_step']*1e3), d['config'].get('neighbour_searches'), d['config'].get('stale_list_stalls'), d['config'].get('launches_replayed'))"
done
