#!/bin/bash
# round 5: a helper thread enqueues the second stream's launches of a run of split steps (tick_single.hip launch_steps_split)
set -o pipefail
mkdir -p gpurun_out
OUT=gpurun_out/r05_w2.log; : > $OUT
for on in 1 0 1 0; do
  echo "== MRS_ENQUEUE_THREAD=$on region_overhead" >> $OUT
  MRS_ENQUEUE_THREAD=$on timeout -k 10 200 python tools/region_overhead.py 20 40 100 1000 2>&1 | grep -v amdgpu.ids >> $OUT || exit 1
done
S="--sub-records off --config5 off --no-cpu-baseline"
for on in 1 0 1 0; do
  echo "== MRS_ENQUEUE_THREAD=$on bench --steps 20 --warmup 5 / default" >> $OUT
  MRS_ENQUEUE_THREAD=$on timeout -k 10 300 python bench.py --steps 20 --warmup 5 $S 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('K=20 value %.4g wall %.3f us dev %.3f us regions %d'%(d['value'], d['ms_per_step']*1e3, d['device_ms_per_step']*1e3, d['regions']))" >> $OUT || exit 1
  MRS_ENQUEUE_THREAD=$on timeout -k 10 300 python bench.py $S 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('K=1000 value %.4g wall %.3f us dev %.3f us regions %d'%(d['value'], d['ms_per_step']*1e3, d['device_ms_per_step']*1e3, d['regions']))" >> $OUT || exit 1
done
cat $OUT
