set -x
OUT=gpurun_out/live; rm -rf $OUT; mkdir -p $OUT
( time timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/time.log; tail -3 $OUT/time.log; tail -3 $OUT/bench.err
python - <<'PY'
import json
for l in open('gpurun_out/live/bench.json'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('traffic', r['traffic'], r['traffic_source'], 'value %.3e'%d['value'])
PY
timeout -k 10 600 python bench.py --no-cpu-baseline --workload position+collisions > $OUT/bench_coll.json 2> $OUT/bench_coll.err
python - <<'PY'
import json
for l in open('gpurun_out/live/bench_coll.json'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('coll traffic', r['traffic'], r['traffic_source'], 'tick us %.2f'%(d['ms_per_step']*1e3))
PY
