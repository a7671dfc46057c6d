#!/bin/bash
# round 5, call 10: the interior launch's kernel with its PID columns fetched with the state (-DMRS_INTERIOR_PIDPRE=1)
mkdir -p gpurun_out; OUT=gpurun_out/r05_j.log; : > $OUT
V=$PWD/variants/libmrs_stepflag__DMRS_INTERIOR_PIDPRE_1.so
for rep in 1 2 3; do
  for lat in 10 20 0; do
    timeout -k 10 200 python tools/sharded_interior_alone.py $lat 600 2>/dev/null | sed "s/^/base   /" >> $OUT
    MRS_SWARM_LIB=$V timeout -k 10 200 python tools/sharded_interior_alone.py $lat 600 2>/dev/null | sed "s/^/pidpre /" >> $OUT
  done
done
sort $OUT | cut -c1-130
echo "=== phases of an interior wave with the PID columns prefetched"
MRS_SWARM_LIB=$PWD/variants/libmrs_stepflag__DMRS_INTERIOR_PIDPRE_1__DMRS_TS_1__DMRS_TS_PART_1.so timeout -k 10 200 python tools/launch_phases.py 10 400 2>/dev/null | head -14
