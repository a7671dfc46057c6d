# the default bench.py run, timed, with the sub-records in short
mkdir -p gpurun_out/b
t0=$(date +%s.%N)
timeout -k 10 600 python bench.py > gpurun_out/b/default.json 2> gpurun_out/b/default.err || { tail -20 gpurun_out/b/default.err; exit 1; }
t1=$(date +%s.%N)
python3 -c "print('default bench.py: %.0f s' % ($t1 - $t0))"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/b/default.json").read().strip().splitlines()[-1])
print("headline", round(d["ms_per_step"] * 1e3, 2), "us per step, frac", round(d["roofline"]["frac"], 3))
for k in ("hbm_streaming", "config4"):
    print(k, round(d[k]["ms_per_step"] * 1e3, 2), "us per step")
print("sharded_rank_standin", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in d["sharded_rank_standin"].items() if k not in ("runs", "workload")})
print("config5", round(d["config5"]["ms_per_tick"] * 1e3, 1), "us per tick")
PY
