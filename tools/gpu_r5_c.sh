#!/bin/bash
# round 5, call 3: what bounds the sharded rank's split tick + pipelined I/O after the copy-stream split
mkdir -p gpurun_out; OUT=gpurun_out/r05_c.log; : > $OUT
python -m pytest tests/test_outputs.py tests/test_simulator_loop.py -x -q -m gpu -s > gpurun_out/r05_c_tests.log 2>&1; echo "tests rc=$?" >> $OUT; tail -4 gpurun_out/r05_c_tests.log >> $OUT
for ch in 1 2; do
MRS_IO_CHUNKS=$ch timeout -k 10 200 python -c "
import bench, json, sys
sys.argv=['bench.py']; a=bench.parse()
r=bench.io_tick_record(a); print('io_tick chunks=$ch', json.dumps({k:r[k] for k in ('serial','pipelined','speedup')}))" >> $OUT 2>gpurun_out/r05_c_io.err || tail -3 gpurun_out/r05_c_io.err >> $OUT
done
echo "--- standard library: one rank of 8 x 125000, stand-in collective" >> $OUT
for lat in 0 10 20; do
  for nt in 1 0; do
    MRS_INTERIOR_NT=$nt timeout -k 10 200 python tools/sharded_interior_alone.py $lat 400 >> $OUT 2>gpurun_out/r05_c.err || tail -3 gpurun_out/r05_c.err >> $OUT
  done
done
MRS_SHARD_SPLIT=0 timeout -k 10 200 python tools/sharded_interior_alone.py 10 400 >> $OUT 2>&1
echo "--- measurement build (-DMRS_WAIT_TICKS=0: waits give up at once), parts of the split tick left out" >> $OUT
V=variants/libmrs_stepflag__DMRS_WAIT_TICKS_0ll.so
for skip in 0 3 1 2 6 4 5; do
  for nt in 1 0; do
    MRS_SWARM_LIB=$V MRS_EXP_SPLIT_SKIP=$skip MRS_INTERIOR_NT=$nt timeout -k 10 200 python tools/sharded_interior_alone.py 10 400 >> $OUT 2>gpurun_out/r05_c.err || { echo "skip=$skip nt=$nt FAILED" >> $OUT; tail -3 gpurun_out/r05_c.err >> $OUT; }
  done
done
cat $OUT
bash tools/gpu_rank_trace.sh 125000 10 > gpurun_out/r05_c_trace.txt 2>&1; tail -14 gpurun_out/r05_c_trace.txt
