#!/bin/bash
# round 5: kernel timeline of the split tick with the interior launch's layer-1 blocks first, against index order (10 us stand-in collective)
set -o pipefail
mkdir -p gpurun_out
for on in 1 0; do
  bash tools/gpu_rank_trace.sh 125000 10 MRS_INTERIOR_L1_FIRST=$on > gpurun_out/r05_v4_trace_$on.txt 2>&1 || { tail gpurun_out/r05_v4_trace_$on.txt; exit 1; }
  f=$(ls -t gpurun_out/ranktrace_125000/prof/*/*_kernel_trace.csv | head -1)
  echo "=== MRS_INTERIOR_L1_FIRST=$on"
  python3 - "$f" <<'PY'
import csv,sys,statistics
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
n=len(rows); rows=rows[n//2:n//2+400]
d={}
for r in rows:
    k=r['Kernel_Name'].split('(')[0].replace('(anonymous namespace)::','')[:34]
    d.setdefault(k,[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items():
    if len(v)>20: print('%-36s n %3d median %5.1f p90 %5.1f'%(k,len(v),statistics.median(v),sorted(v)[int(len(v)*0.9)]))
b=[r for r in rows if 'bnd' in r['Kernel_Name']]
per=[(int(b[i+1]['Start_Timestamp'])-int(b[i]['Start_Timestamp']))/1e3 for i in range(len(b)-1)]
print('boundary launch period: median %.1f us'%statistics.median(per))
PY
done
