#!/bin/bash
# round 5: the second stream of a run of split steps starts behind a short delay kernel (MRS_STAGGER_US), with and without the helper thread
set -o pipefail
mkdir -p gpurun_out
OUT=gpurun_out/r05_w3.log; : > $OUT
for rep in 1 2; do
for th in 1 0; do
for st in 0 2 3.5 5; do
  echo "== MRS_ENQUEUE_THREAD=$th MRS_STAGGER_US=$st" >> $OUT
  MRS_ENQUEUE_THREAD=$th MRS_STAGGER_US=$st timeout -k 10 200 python tools/region_overhead.py 20 40 1000 2>&1 | grep -v amdgpu.ids | cut -c1-230 >> $OUT || exit 1
done
done
done
cat $OUT
