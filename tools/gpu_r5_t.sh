#!/bin/bash
# round 5: the second stream of a run of steps joined lazily (MRS_LAZY_JOIN, default 1) — whole suite, then short regions A/B
mkdir -p gpurun_out; OUT=gpurun_out/r05_t.log; : > $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r05_t_tests.log 2>&1; echo "tests rc=$?" >> $OUT; tail -3 gpurun_out/r05_t_tests.log >> $OUT
for rep in 1 2 3; do
for K in 20 40 160 1000; do
  for lz in 1 0; do
    MRS_LAZY_JOIN=$lz timeout -k 10 200 python bench.py --steps $K --warmup 5 --no-cpu-baseline --traffic off --sub-records off --config5 off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=%4d lazy=$lz' % $K, 'wall %.2f device %.2f us/step' % (d['ms_per_step']*1e3, d['device_ms_per_step']*1e3))" >> $OUT
  done
done
done
sort $OUT
