#!/bin/bash
# round 5, call 9: where a wave's time goes — interior launch of the split tick against the full launch of the serial form (and the boundary launch)
mkdir -p gpurun_out; OUT=gpurun_out/r05_i.log; : > $OUT
V=$PWD/variants/libmrs_stepflag__DMRS_TS_1__DMRS_TS_PART
echo "=== interior launch of a split tick (10 us)" >> $OUT
MRS_SWARM_LIB=${V}_1.so timeout -k 10 200 python tools/launch_phases.py 10 400 2>/dev/null >> $OUT
echo "=== (cached interior accesses skipped)" >> $OUT
echo "=== full launch of the serial form (MRS_SHARD_SPLIT=0, 10 us)" >> $OUT
MRS_SWARM_LIB=${V}_0.so MRS_SHARD_SPLIT=0 timeout -k 10 200 python tools/launch_phases.py 10 400 2>/dev/null >> $OUT
cat $OUT
