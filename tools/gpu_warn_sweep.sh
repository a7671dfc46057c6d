OUT=gpurun_out/warn; rm -rf $OUT; mkdir -p $OUT
for wf in 0.75 0.82 0.88 0.94; do
  MRS_WARN_FRACTION=$wf timeout -k 10 200 python bench.py --no-cpu-baseline --traffic off --workload position+collisions > $OUT/b_$wf.json 2> $OUT/b_$wf.err || exit 1
  python - $wf <<'PY'
import json,sys
for l in open('gpurun_out/warn/b_%s.json'%sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[1], 'tick us %.2f'%(d['ms_per_step']*1e3), {k:v for k,v in d['config'].items() if 'search' in k or 'stall' in k or 'noop' in k})
PY
done
