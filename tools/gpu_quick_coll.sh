set -x
export TMPDIR=/tmp
OUT=gpurun_out/quick
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 120 python -m pytest "tests/test_parity_gpu.py::test_neighbour_lists_are_reused_and_rebuilt_with_identical_results" "tests/test_parity_gpu.py::test_full_size_100k_collision_tick_against_oracle" tests/test_export_sets_gpu.py -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
MRS_FUSED_LEAD=3 timeout -k 10 200 python bench.py --no-cpu-baseline --workload position+collisions > $OUT/bench_coll.json 2> $OUT/bench_coll.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_coll -- python bench.py --steps 500 --warmup 50 --no-cpu-baseline --workload position+collisions > $OUT/bench_coll_trace.json 2> $OUT/trace_coll.err
python - <<'PY'
import csv,glob,json
import numpy as np
for l in open('gpurun_out/quick/bench_coll.json'):
    if l.startswith('{'):
        d=json.loads(l); print('tick us %.2f'%(d['ms_per_step']*1e3), {k:v for k,v in d['config'].items() if 'search' in k or 'stall' in k})
f=sorted(glob.glob('gpurun_out/quick/trace_coll/*/*kernel_trace.csv'))[-1]
rows=list(csv.DictReader(open(f)))
d=[(r['Kernel_Name'][:60], int(r['End_Timestamp'])-int(r['Start_Timestamp'])) for r in rows]
fk=np.array([t for n,t in d if n.startswith('mrs_uav_step_coll')]); full=fk[fk>6000]
print('FK full launches: n',len(full),'mean %.0f median %.0f'%(full.mean(),np.median(full)))
for key in ('k_query','k_pack_insert'):
    q=np.array([t for n,t in d if key in n]); print(key, len(q), q.mean() if len(q) else None)
PY
