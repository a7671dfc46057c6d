OUT=gpurun_out/wd; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 python bench.py --no-cpu-baseline --traffic off --steps 100 --warmup 10 --config5 on --config5-timeout 1 > $OUT/wd.json 2> $OUT/wd.err; echo "rc $?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/wd/wd.json') if x.startswith('{')]
d=json.loads(l[-1]); print(len(l), 'value %.3e'%d['value'], d.get('config5'))
PY
timeout -k 10 300 python bench.py --no-cpu-baseline --traffic off --steps 100 --warmup 10 --config5 on > $OUT/ok.json 2> $OUT/ok.err; echo "rc $?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/wd/ok.json') if x.startswith('{')]
d=json.loads(l[-1]); c=d['config5']; print(len(l), 'value %.3e'%d['value'], 'config5 us/tick %.1f'%(c['ms_per_tick']*1e3), c['search_ticks'], c['sharded_ticks'])
PY
timeout -k 10 300 python tools/sharded_rank_cost.py 125000 8 400 export 2>&1 | tail -n 1 | cut -c1-250
