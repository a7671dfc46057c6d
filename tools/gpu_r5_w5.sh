#!/bin/bash
# round 5: device-side timeline of 20-step regions (tools/region_timeline.py), plain library and the helper-thread experiment
set -o pipefail
mkdir -p gpurun_out
for v in stepflag__DMRS_TS_STEP_1 ts_step_helper; do
  for K in 20 ${EXTRA_K}; do
  echo "=== $v K=$K" >> gpurun_out/r05_w5_timeline.txt
  MRS_SWARM_LIB=$PWD/variants/libmrs_$v.so timeout -k 10 200 python tools/region_timeline.py $K 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_w5_timeline.txt || { tail gpurun_out/r05_w5_timeline.txt; exit 1; }
  done
done
