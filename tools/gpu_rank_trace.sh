# kernel timeline of ONE rank of the sharded tick in the split form (tools/sharded_rank_cost.py under rocprofv3 --kernel-trace)
# usage: gpu_rank_trace.sh <uavs per rank> <latency us> [env assignments...]
N=${1:-125000}; LAT=${2:-20}; shift; shift
OUT=$PWD/gpurun_out/ranktrace_$N; rm -rf $OUT; mkdir -p $OUT
ROOT=$PWD; cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -- python3 $ROOT/tools/sharded_rank_cost.py $N 8 400 $LAT split > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
cd $ROOT
tail -1 $OUT/trace.log | cut -c1-160
python3 - "$OUT" <<'PY'
import csv,glob,os,sys
f=sorted(glob.glob(sys.argv[1]+'/prof/*/*_kernel_trace.csv'), key=os.path.getmtime)[-1]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
i0=len(rows)*3//4
t0=int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i0+12]:
    print(r['Kernel_Name'][:32].ljust(32), 'q',r['Queue_Id'], 'blocks',int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']), 'start %.1f'%((int(r['Start_Timestamp'])-t0)/1e3), 'end %.1f'%((int(r['End_Timestamp'])-t0)/1e3), 'dur %.1f'%((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
PY
