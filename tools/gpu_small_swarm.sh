OUT=gpurun_out/small; rm -rf $OUT; mkdir -p $OUT
for n in 400 4000; do
for sub in 1 2 4 8 16 64; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --traffic off --uavs $n --workload position --steps 2048 --warmup 128 --substeps $sub > $OUT/b_${n}_$sub.json 2> $OUT/b_${n}_$sub.err || { tail $OUT/b_${n}_$sub.err; exit 1; }
  python - $n $sub <<'PY'
import json,sys
for l in open('gpurun_out/small/b_%s_%s.json'%(sys.argv[1],sys.argv[2])):
    if l.startswith('{'):
        d=json.loads(l); print('n',sys.argv[1],'substeps',sys.argv[2], 'us/step %.2f'%(d['ms_per_step']*1e3), 'value %.3e'%d['value'])
PY
done; done
