#!/bin/bash
# round 5, call 15: the driver's K = 20 region with the runtime's active-wait window (ROC_ACTIVE_WAIT_TIMEOUT, microseconds) 
mkdir -p gpurun_out; OUT=gpurun_out/r05_p.log; : > $OUT
for rep in 1 2 3; do
for v in none 100 1000 200000; do
  if [ $v = none ]; then E="X=1"; else E="ROC_ACTIVE_WAIT_TIMEOUT=$v"; fi
  env $E timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic off --sub-records off --config5 off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=20 wait $v'.ljust(22), 'wall %.2f device %.2f us/step, first region %.2f, regions %d' % (d['ms_per_step']*1e3, d['device_ms_per_step']*1e3, d['first_region_wall_ms_per_step']*1e3, d['regions']))" >> $OUT
done
done
sort $OUT
