OUT=gpurun_out/k20c; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_bench_launch_gpu.py tests/test_parity_gpu.py tests/test_random_sequences_gpu.py -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log | cut -c1-300; exit 1; }
tail -n 1 $OUT/tests.log
for a in "--steps 20 --warmup 5" "--steps 20 --warmup 5" "--steps 100 --warmup 10" "--steps 1000 --warmup 100" "--steps 20 --warmup 5 --workload position"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --traffic off $a 2>> $OUT/b.err | tail -n 1 > $OUT/last.json
  python - "$a" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/k20c/last.json').read())
print(sys.argv[1].ljust(40), 'us/step %.2f'%(d['ms_per_step']*1e3), 'value %.3e'%d['value'], 'wall us/step %.2f'%(d['wall_ms_per_step']*1e3), 'regions', d['regions'])
PY
done
