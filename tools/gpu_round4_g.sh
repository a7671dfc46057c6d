#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4g
run() {  # env assignments...
  env "$@" timeout -k 10 200 python bench.py --workload position+collisions --steps 300 --warmup 100 --no-cpu-baseline --traffic off --sub-records off --config5 off 2>gpurun_out/r4g/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print('$*: wall %.2f device %.2f us/tick; searches %s stalls %s replayed %s ahead %s' % (d['ms_per_step']*1e3, d['device_ms_per_step']*1e3, c.get('neighbour_searches'), c.get('stale_list_stalls'), c.get('launches_replayed'), c.get('searches_queued_ahead')))"
}
run MRS_OVERLAP_TICKS=0
run MRS_OVERLAP_TICKS=1
run MRS_OVERLAP_TICKS=1 MRS_WARN_FRACTION=0.65
run MRS_OVERLAP_TICKS=1 MRS_WARN_FRACTION=0.55
run MRS_OVERLAP_TICKS=1 MRS_WARN_FRACTION=0.65 MRS_FUSED_LEAD=2
run MRS_OVERLAP_TICKS=0 MRS_WARN_FRACTION=0.65
