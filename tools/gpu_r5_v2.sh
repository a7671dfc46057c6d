#!/bin/bash
# round 5: the interior launch of a split tick hands out a slab's blocks from both ends inwards (-DMRS_INTERIOR_ENDS_FIRST=1) against index order
set -o pipefail
mkdir -p gpurun_out; OUT=gpurun_out/r05_v2.log; : > $OUT
V=$PWD/variants/libmrs_stepflag__DMRS_INTERIOR_ENDS_FIRST_1.so
MRS_SWARM_LIB=$V timeout -k 10 400 python -m pytest tests/test_sharded_chaos_gpu.py tests/test_config5_gpu.py -x -q -m gpu > gpurun_out/r05_v2_tests.log 2>&1; echo "tests with the variant rc=$?" >> $OUT; tail -1 gpurun_out/r05_v2_tests.log >> $OUT
for rep in 1 2 3 4 5; do
  for lat in 10 20; do
    printf "ends_first lat=$lat " >> $OUT; MRS_SWARM_LIB=$V timeout -k 10 200 python tools/sharded_interior_alone.py $lat 600 2>&1 | grep "us per tick" | cut -c1-120 >> $OUT || exit 1
    printf "index_order lat=$lat " >> $OUT; timeout -k 10 200 python tools/sharded_interior_alone.py $lat 600 2>&1 | grep "us per tick" | cut -c1-120 >> $OUT || exit 1
  done
done
python3 - <<'PY'
import re,collections,statistics
d=collections.defaultdict(list)
for l in open('gpurun_out/r05_v2.log'):
    m=re.match(r'(\w+) lat=(\d+) .*?: ([\d.]+) us per tick',l)
    if m: d[(m.group(1),m.group(2))].append(float(m.group(3)))
for k in sorted(d): print(k[0].ljust(12),'latency',k[1],'median %.2f'%statistics.median(d[k]),'runs',' '.join('%.2f'%v for v in sorted(d[k])))
PY
head -2 $OUT
