#!/bin/bash
# round 5, call 4: per-kernel durations of one rank's split tick under variants of the interior kernel (rocprofv3 kernel trace)
mkdir -p gpurun_out; OUT=$PWD/gpurun_out/r05_d.log; : > $OUT
ROOT=$PWD
run() { # label, lat, env...
  label=$1; lat=$2; shift; shift
  D=$ROOT/gpurun_out/r05_d_prof; rm -rf $D; mkdir -p $D
  ( cd /tmp && export TMPDIR=/tmp && for kv in "$@"; do export "$kv"; done && timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $ROOT/tools/sharded_interior_alone.py $lat 400 > $D/run.log 2>&1 )
  line=$(grep "us per tick" $D/run.log | cut -c1-70)
  python3 - "$D" "$label" "$line" >> $OUT <<'PY'
import csv,glob,os,sys,collections
fs=sorted(glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv'), key=os.path.getmtime)
if not fs: print(sys.argv[2], 'NO TRACE', sys.argv[3]); sys.exit(0)
rows=list(csv.DictReader(open(fs[-1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=rows[len(rows)//3:]  # the timed part
d=collections.defaultdict(list)
for r in rows: d[r['Kernel_Name'].split('(')[0][-34:]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
keep=[k for k in d if any(t in k for t in ('step_coll','xcoll','k_standin'))]
def med(v): v=sorted(v); return v[len(v)//2]
print(sys.argv[2].ljust(44), '|', sys.argv[3], '|', '; '.join(f"{k.replace('mrs_uav_step_','')} med {med(d[k]):.1f} n={len(d[k])}" for k in sorted(keep)))
PY
}
W=$ROOT/variants/libmrs_stepflag__DMRS_WAIT_TICKS_0ll
run "nowait interior alone nt=1" 10 MRS_SWARM_LIB=$W.so MRS_EXP_SPLIT_SKIP=3
run "nowait interior alone nt=0" 10 MRS_SWARM_LIB=$W.so MRS_EXP_SPLIT_SKIP=3 MRS_INTERIOR_NT=0
run "nowait+nodrain interior alone nt=1" 10 MRS_SWARM_LIB=${W}__DMRS_EXP_NO_DRAIN_1.so MRS_EXP_SPLIT_SKIP=3
run "nowait+nodrain interior alone nt=0" 10 MRS_SWARM_LIB=${W}__DMRS_EXP_NO_DRAIN_1.so MRS_EXP_SPLIT_SKIP=3 MRS_INTERIOR_NT=0
run "nowait+nodrain full tick nt=1" 10 MRS_SWARM_LIB=${W}__DMRS_EXP_NO_DRAIN_1.so
run "nowait+ld0 interior alone" 10 MRS_SWARM_LIB=${W}__DMRS_NT_LD_AUX_0.so MRS_EXP_SPLIT_SKIP=3
run "nowait+st0 interior alone" 10 MRS_SWARM_LIB=${W}__DMRS_NT_ST_AUX_0.so MRS_EXP_SPLIT_SKIP=3
run "nowait+ld0+nodrain interior alone" 10 MRS_SWARM_LIB=${W}__DMRS_NT_LD_AUX_0__DMRS_EXP_NO_DRAIN_1.so MRS_EXP_SPLIT_SKIP=3
run "ld0 (correct protocol) split lat 0" 0 MRS_SWARM_LIB=$ROOT/variants/libmrs_stepflag__DMRS_NT_LD_AUX_0.so
run "ld0 (correct protocol) split lat 10" 10 MRS_SWARM_LIB=$ROOT/variants/libmrs_stepflag__DMRS_NT_LD_AUX_0.so
run "ld0 (correct protocol) split lat 20" 20 MRS_SWARM_LIB=$ROOT/variants/libmrs_stepflag__DMRS_NT_LD_AUX_0.so
cat $OUT
