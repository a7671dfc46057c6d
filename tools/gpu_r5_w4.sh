#!/bin/bash
# round 5: kernel by kernel through 20-step regions of the headline workload (rocprofv3 --kernel-trace of bench.py --steps 20)
set -o pipefail
ROOT=$PWD; OUT=$PWD/gpurun_out/r05_w4; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -- python3 $ROOT/bench.py --steps 20 --warmup 5 --sub-records off --config5 off --no-cpu-baseline --min-measure-ms 5 > $OUT/bench.log 2>&1 || { tail -20 $OUT/bench.log; exit 1; }
cd $ROOT
python3 - "$OUT" <<'PY'
import csv,glob,os,sys
f=sorted(glob.glob(sys.argv[1]+'/prof/*/*_kernel_trace.csv'), key=os.path.getmtime)[-1]
rows=[r for r in csv.DictReader(open(f)) if 'mrs_uav' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# regions: gaps > 30 us between consecutive kernel starts
regs=[[rows[0]]]
for a,b in zip(rows,rows[1:]):
    if int(b['Start_Timestamp'])-int(a['End_Timestamp'])>30000: regs.append([])
    regs[-1].append(b)
regs=[r for r in regs if len(r)==40]
print(len(regs),'regions of 40 launches')
import statistics
for k in range(40):
    d=[(int(r[k]['End_Timestamp'])-int(r[k]['Start_Timestamp']))/1e3 for r in regs]
    s=[(int(r[k]['Start_Timestamp'])-int(r[0]['Start_Timestamp']))/1e3 for r in regs]
    print('launch %2d q%s start %7.1f dur %5.1f'%(k, regs[0][k]['Queue_Id'], statistics.median(s), statistics.median(d)))
span=[(max(int(x['End_Timestamp']) for x in r)-int(r[0]['Start_Timestamp']))/1e3 for r in regs]
print('span median %.1f us'%statistics.median(span))
PY
