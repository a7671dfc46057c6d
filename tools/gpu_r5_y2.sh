#!/bin/bash
# round 5: halo exchange against full gather on search ticks, five runs of each setting (the per-rank figure scatters by 1-2 us from run to run)
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r05_y2_rank.log
for rep in 1 2 3 4 5; do
for halo in 1 0; do
  for gbps in 0 300 150; do
    printf "halo=$halo gbps=$gbps " >> gpurun_out/r05_y2_rank.log
    MRS_SEARCH_HALO=$halo MRS_STANDIN_GBPS=$gbps timeout -k 10 200 python tools/sharded_interior_alone.py 10 600 2>&1 | grep "us per tick" >> gpurun_out/r05_y2_rank.log || exit 1
  done
done
done
python3 - <<'PY'
import re,collections,statistics
d=collections.defaultdict(list)
for l in open('gpurun_out/r05_y2_rank.log'):
    m=re.match(r'halo=(\d) gbps=(\d+) .*?: ([\d.]+) us per tick',l)
    if m: d[(m.group(1),m.group(2))].append(float(m.group(3)))
for k in sorted(d): print('halo',k[0],'gbps',k[1].rjust(4),'median %.2f'%statistics.median(d[k]),'runs',' '.join('%.2f'%v for v in sorted(d[k])))
PY
