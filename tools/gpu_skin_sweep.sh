set -x
OUT=gpurun_out/skin
rm -rf $OUT; mkdir -p $OUT
for v in variants/libmrs_skin_*.so; do
  t=$(basename $v .so)
  for extra in "" "--volume-per-uav 16"; do
    tag=${t}_$(echo "$extra" | tr -c 'A-Za-z0-9\n' '_')
    MRS_SWARM_LIB=$PWD/$v MRS_FUSED_LEAD=3 timeout -k 10 200 python bench.py --no-cpu-baseline --workload position+collisions $extra > $OUT/$tag.json 2> $OUT/$tag.err
    python - <<PY
import json
for l in open('$OUT/$tag.json'):
    if l.startswith('{'):
        d=json.loads(l); print('$tag'.ljust(40), 'tick us %.2f'%(d['ms_per_step']*1e3), {k:v for k,v in d['config'].items() if 'search' in k or 'stall' in k or k=='collision_ticks'})
PY
  done
done
