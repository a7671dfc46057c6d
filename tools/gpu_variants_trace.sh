set -x
export TMPDIR=/tmp
OUT=gpurun_out/vtrace
rm -rf $OUT; mkdir -p $OUT
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/$tag -- python bench.py --steps 400 --warmup 50 --no-cpu-baseline --workload position+collisions > $OUT/$tag.json 2> $OUT/$tag.err
}
run base A=1
run unfused MRS_FUSED_COLLISIONS=0
for v in variants/libmrs_stepflag_*.so; do t=$(basename $v .so); run $t MRS_SWARM_LIB=$PWD/$v; done
python - <<'PY'
import csv,glob,json,os
import numpy as np
for d in sorted(glob.glob('gpurun_out/vtrace/*/')):
    tag=os.path.basename(d.rstrip('/'))
    f=sorted(glob.glob(d+'*/*kernel_trace.csv'))
    if not f: print(tag,'no trace'); continue
    rows=list(csv.DictReader(open(f[-1])))
    k={}
    for r in rows:
        k.setdefault(r['Kernel_Name'][:48],[]).append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
    tick=None
    try:
        for l in open(f'gpurun_out/vtrace/{tag}.json'):
            if l.startswith('{'): tick=json.loads(l)['ms_per_step']*1e3
    except Exception: pass
    out=[]
    for n,v in k.items():
        v=np.array(v)
        if n.startswith('mrs_uav_step'): v=v[v>6000]
        if len(v)>20: out.append(f"{n[:36]}: n {len(v)} median {np.median(v)/1e3:.2f}")
    print(tag.ljust(40),'tick', tick, ' | '.join(out))
PY
