#!/bin/bash
# round 5: where between 0 and 10 us of collective latency MRS_INTERIOR_L1_FIRST stops paying (three runs each)
set -o pipefail
mkdir -p gpurun_out; OUT=gpurun_out/r05_v5.log; : > $OUT
for rep in 1 2 3; do for lat in 2 4 6 8; do for on in 1 0; do
  printf "l1_first=$on lat=$lat " >> $OUT; MRS_INTERIOR_L1_FIRST=$on timeout -k 10 200 python tools/sharded_interior_alone.py $lat 600 2>&1 | grep "us per tick" | cut -c1-120 >> $OUT || exit 1
done; done; done
python3 - <<'PY'
import re,collections,statistics
d=collections.defaultdict(list)
for l in open('gpurun_out/r05_v5.log'):
    m=re.match(r'l1_first=(\d) lat=(\d+) .*?: ([\d.]+) us per tick',l)
    if m: d[(int(m.group(2)),m.group(1))].append(float(m.group(3)))
for k in sorted(d): print('latency %2d'%k[0],'l1_first',k[1],'median %.2f'%statistics.median(d[k]),'runs',' '.join('%.2f'%v for v in sorted(d[k])))
PY
