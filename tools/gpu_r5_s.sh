#!/bin/bash
# round 5: runs of K steps as one launch per step (MRS_SPLIT_STREAMS=0) against two half-swarm launches on two streams, by K
mkdir -p gpurun_out; OUT=gpurun_out/r05_s.log; : > $OUT
for rep in 1 2 3; do
for K in 20 40 80 160; do
  for sp in 1 0; do
    MRS_SPLIT_STREAMS=$sp timeout -k 10 200 python bench.py --steps $K --warmup 5 --no-cpu-baseline --traffic off --sub-records off --config5 off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=%4d split=$sp' % $K, 'wall %.2f device %.2f us/step' % (d['ms_per_step']*1e3, d['device_ms_per_step']*1e3))" >> $OUT
  done
done
done
sort $OUT
