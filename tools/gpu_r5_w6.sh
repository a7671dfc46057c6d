#!/bin/bash
# round 5: plain step launches through hipModuleLaunchKernel (MRS_MODULE_LAUNCH=1, default) against hipLaunchKernelGGL (=0)
set -o pipefail
mkdir -p gpurun_out
OUT=gpurun_out/r05_w6.log; : > $OUT
timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -x -q -m gpu > gpurun_out/r05_w6_tests.log 2>&1; echo "tests rc=$?" >> $OUT; tail -2 gpurun_out/r05_w6_tests.log >> $OUT
for on in 1 0 1 0 1 0; do
  echo "== MRS_MODULE_LAUNCH=$on" >> $OUT
  MRS_MODULE_LAUNCH=$on timeout -k 10 200 python tools/region_overhead.py 20 40 1000 2>&1 | grep -v amdgpu.ids | cut -c1-200 >> $OUT || exit 1
done
S="--sub-records off --config5 off --no-cpu-baseline"
for on in 1 0 1 0 1 0; do
  MRS_MODULE_LAUNCH=$on timeout -k 10 300 python bench.py --steps 20 --warmup 5 $S 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('MODULE=$on K=20 value %.4g wall %.3f us dev %.3f us regions %d'%(d['value'], d['ms_per_step']*1e3, d['device_ms_per_step']*1e3, d['regions']))" >> $OUT || exit 1
done
cat $OUT
