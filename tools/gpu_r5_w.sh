#!/bin/bash
# (experiment) position and command columns requested first in the prologue (-DMRS_EARLY_CMD=1)
mkdir -p gpurun_out; OUT=gpurun_out/r05_w.log; : > $OUT
V=$PWD/variants/libmrs_stepflag__DMRS_EARLY_CMD_1.so
MRS_SWARM_LIB=$V timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu 2>&1 | tail -2 >> $OUT
run() { for w in position position+collisions; do
    env $2 timeout -k 10 300 python bench.py --workload $w --steps 300 --warmup 100 --no-cpu-baseline --traffic off --sub-records off --config5 off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['config']['workload'][:44].ljust(44), 'wall %.2f device %.2f' % (d['ms_per_step']*1e3, d['device_ms_per_step']*1e3))" >> $OUT
  done; }
run base X=1; run early MRS_SWARM_LIB=$V; run base X=1; run early MRS_SWARM_LIB=$V; run base X=1; run early MRS_SWARM_LIB=$V
sort $OUT
