OUT=gpurun_out/quickv; rm -rf $OUT; mkdir -p $OUT
for v in "A=1" "MRS_SEARCH_MARGIN=0" "MRS_SEARCH_MARGIN=8" "MRS_WARN_FRACTION=0.75" "MRS_WARN_FRACTION=0.3" "MRS_WARN_FRACTION=0.3 MRS_SEARCH_MARGIN=1"; do
  tag=$(echo "$v" | tr -c 'A-Za-z0-9\n' '_')
  env $v MRS_FUSED_LEAD=3 timeout -k 10 200 python bench.py --no-cpu-baseline --workload position+collisions > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
for l in open('$OUT/$tag.json'):
    if l.startswith('{'):
        d=json.loads(l); print('$v'.ljust(44), 'tick us %.2f'%(d['ms_per_step']*1e3), {k:v for k,v in d['config'].items() if 'search' in k or 'stall' in k or k=='collision_ticks'})
PY
done
