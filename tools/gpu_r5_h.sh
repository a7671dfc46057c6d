#!/bin/bash
# round 5, call 8: the search beside the launches with an earlier warning
mkdir -p gpurun_out; OUT=gpurun_out/r05_h.log; : > $OUT
timeout -k 10 600 python -m pytest tests/test_async_search_gpu.py -x -q -m gpu -s > gpurun_out/r05_h_tests.log 2>&1; echo "tests rc=$?" >> $OUT; grep -v amdgpu.ids gpurun_out/r05_h_tests.log | tail -4 >> $OUT
run() { # label env...
  label=$1; shift
  for vol in 64 16; do
  env "$@" timeout -k 10 200 python bench.py --workload position+collisions --volume-per-uav $vol --steps 300 --warmup 100 --no-cpu-baseline --traffic off --sub-records off --config5 off 2>gpurun_out/r05_h.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print('$label vol $vol: wall %.2f device %.2f us/tick; searches %d stalls %d replayed %d ahead %d' % (d['ms_per_step']*1e3, d['device_ms_per_step']*1e3, c['neighbour_searches'], c['stale_list_stalls'], c['launches_replayed'], c['searches_queued_ahead']))" >> $OUT || { echo "$label FAILED" >> $OUT; tail -3 gpurun_out/r05_h.err >> $OUT; }
  done
}
for rep in 1 2; do
run "in order          " MRS_ASYNC_SEARCH=0
run "depth 2 warn 0.65 " MRS_ASYNC_SEARCH=2 MRS_WARN_FRACTION_ASYNC=0.65
run "depth 2 warn 0.62 " MRS_ASYNC_SEARCH=2 MRS_WARN_FRACTION_ASYNC=0.62
run "depth 2 warn 0.55 " MRS_ASYNC_SEARCH=2 MRS_WARN_FRACTION_ASYNC=0.55
run "depth 1 warn 0.68 " MRS_ASYNC_SEARCH=1 MRS_WARN_FRACTION_ASYNC=0.68
run "depth 1 warn 0.62 " MRS_ASYNC_SEARCH=1 MRS_WARN_FRACTION_ASYNC=0.62
done
cat $OUT
