#!/bin/bash
# round 5, call 14: boundary launches issued before the previous tick's collective has completed (MRS_EARLY_BOUNDARY=1)
mkdir -p gpurun_out; OUT=gpurun_out/r05_o.log; : > $OUT
MRS_EARLY_BOUNDARY=1 timeout -k 10 900 python -m pytest tests/test_sharded_chaos_gpu.py tests/test_export_sets_gpu.py tests/test_config5_gpu.py tests/test_sharded_multiprocess_gpu.py -x -q -m gpu > gpurun_out/r05_o_tests.log 2>&1; echo "sharded tests with early boundary launches rc=$?" >> $OUT; tail -3 gpurun_out/r05_o_tests.log >> $OUT
for rep in 1 2; do
  for lat in 0 10 20 30; do
    timeout -k 10 200 python tools/sharded_interior_alone.py $lat 600 2>/dev/null | sed "s/^/base  /" | cut -c1-120 >> $OUT
    MRS_EARLY_BOUNDARY=1 timeout -k 10 200 python tools/sharded_interior_alone.py $lat 600 2>/dev/null | sed "s/^/early /" | cut -c1-120 >> $OUT
  done
done
cat $OUT
MRS_EARLY_BOUNDARY=1 bash tools/gpu_rank_trace.sh 125000 20 MRS_EARLY_BOUNDARY=1 > gpurun_out/r05_o_trace.txt 2>&1; tail -14 gpurun_out/r05_o_trace.txt
