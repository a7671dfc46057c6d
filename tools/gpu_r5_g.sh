#!/bin/bash
# round 5, call 7: the search beside the launches — parity tests, then config 4 with it off / depth 1 / depth 2 (twice each)
mkdir -p gpurun_out; OUT=gpurun_out/r05_g.log; : > $OUT
timeout -k 10 600 python -m pytest tests/test_async_search_gpu.py -x -q -m gpu -s > gpurun_out/r05_g_tests.log 2>&1; echo "tests rc=$?" >> $OUT; grep -v amdgpu.ids gpurun_out/r05_g_tests.log | tail -15 >> $OUT
for rep in 1 2; do
for d in 0 1 2; do
  for vol in 64 16; do
  MRS_ASYNC_SEARCH=$d timeout -k 10 200 python bench.py --workload position+collisions --volume-per-uav $vol --steps 300 --warmup 100 --no-cpu-baseline --traffic off --sub-records off --config5 off 2>gpurun_out/r05_g.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print('async depth $d vol $vol: wall %.2f device %.2f us/tick; searches %d stalls %d replayed %d ahead %d' % (d['ms_per_step']*1e3, d['device_ms_per_step']*1e3, c['neighbour_searches'], c['stale_list_stalls'], c['launches_replayed'], c['searches_queued_ahead']))" >> $OUT || { echo "depth $d FAILED" >> $OUT; tail -3 gpurun_out/r05_g.err >> $OUT; }
  done
done
done
cat $OUT
