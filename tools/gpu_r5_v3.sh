#!/bin/bash
# round 5: the interior launch hands out its layer-1 blocks first (MRS_INTERIOR_L1_FIRST=1) against index order
set -o pipefail
mkdir -p gpurun_out; OUT=gpurun_out/r05_v3.log; : > $OUT
MRS_INTERIOR_L1_FIRST=1 timeout -k 10 600 python -m pytest tests/test_sharded_chaos_gpu.py tests/test_config5_gpu.py tests/test_search_halo_gpu.py tests/test_export_sets_gpu.py -x -q -m gpu > gpurun_out/r05_v3_tests.log 2>&1; rc=$?; echo "tests with MRS_INTERIOR_L1_FIRST=1 rc=$rc" >> $OUT; tail -1 gpurun_out/r05_v3_tests.log >> $OUT; [ $rc -eq 0 ] || { cat $OUT; tail -30 gpurun_out/r05_v3_tests.log; exit 1; }
for rep in 1 2 3 4 5; do
  for lat in 0 10 20; do
    for on in 1 0; do
      printf "l1_first=$on lat=$lat " >> $OUT; MRS_INTERIOR_L1_FIRST=$on timeout -k 10 200 python tools/sharded_interior_alone.py $lat 600 2>&1 | grep "us per tick" | cut -c1-120 >> $OUT || exit 1
    done
  done
done
python3 - <<'PY'
import re,collections,statistics
d=collections.defaultdict(list)
for l in open('gpurun_out/r05_v3.log'):
    m=re.match(r'l1_first=(\d) lat=(\d+) .*?: ([\d.]+) us per tick',l)
    if m: d[(m.group(1),int(m.group(2)))].append(float(m.group(3)))
for k in sorted(d): print('l1_first',k[0],'latency %2d'%k[1],'median %.2f'%statistics.median(d[k]),'runs',' '.join('%.2f'%v for v in sorted(d[k])))
PY
head -2 $OUT
