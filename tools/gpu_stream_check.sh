OUT=gpurun_out/streamchk; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_bench_launch_gpu.py -x -q -m gpu -s > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log | cut -c1-300; exit 1; }
grep -i "passed\|UAVs x" $OUT/tests.log | cut -c1-200
for a in "--uavs 4000000 --steps 100 --warmup 20" "--uavs 2000000 --workload position --steps 100 --warmup 20" "--uavs 1000000 --steps 200 --warmup 20" "--steps 1000 --warmup 100"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline $a 2>> $OUT/b.err | tail -n 1 > $OUT/last.json
  python - <<'PY'
import json
d=json.loads(open('gpurun_out/streamchk/last.json').read()); r=d['roofline']
print(d['config']['uavs_per_gpu'], d['config']['workload'][:40], 'us/step %.1f'%(d['ms_per_step']*1e3), 'value %.3e'%d['value'], 'frac %.3f'%r['frac'], 'ach %.3f'%r['frac_of_achievable'], r['kernel'], 'traffic', r['traffic'] and round(r['traffic']/1e6,1), 'alg MB', round(r['algorithmic_bytes_per_uav_step']*d['config']['uavs_per_gpu']/1e6,1))
PY
done
