# whole tick (position cascade + collisions) of 100 k UAVs against the air space per UAV
OUT=gpurun_out/dens; rm -rf $OUT; mkdir -p $OUT
for v in 64 30 16 8 4; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --traffic off --workload position+collisions --volume-per-uav $v --steps 600 --warmup 100 > $OUT/x.json 2> $OUT/x.err || { tail -3 $OUT/x.err; exit 1; }
  python - $v <<'PY'
import json,sys
d=json.loads([l for l in open('gpurun_out/dens/x.json') if l.startswith('{')][-1]); c=d['config']
print(sys.argv[1].rjust(4),'m3/UAV: us/tick %.1f'%(d['ms_per_step']*1e3), 'searches', c['neighbour_searches'], 'of', c['collision_ticks'], 'stalls', c['stale_list_stalls'], 'fused', c['ticks_evaluated_by_the_next_step_launch'], flush=True)
PY
done
