#!/bin/bash
# round 5, call 2: new tests (pipelined outputs, publisher facade, rehearsal) + the default bench line (size, io_tick)
mkdir -p gpurun_out
python -m pytest tests/test_outputs.py tests/test_simulator_loop.py tests/test_bench_rehearsal_gpu.py tests/test_random_sequences_gpu.py -x -q -m gpu -s > gpurun_out/r05_b_tests.log 2>&1
echo "tests rc=$?" ; tail -15 gpurun_out/r05_b_tests.log
timeout -k 10 600 python bench.py > gpurun_out/r05_b_bench.json 2> gpurun_out/r05_b_bench.err
echo "bench rc=$? bytes=$(wc -c < gpurun_out/r05_b_bench.json)"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_b_bench.json').read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'])
print('io_tick', json.dumps(d.get('io_tick')))
print('config4', d['config4']['ms_per_step'], d['config4']['device_ms_per_step'])
print('standin', {k: d['sharded_rank_standin'][k] for k in ('split_10us','split_20us','serial_10us')})
PY
MRS_BENCH_REHEARSAL=1 timeout -k 10 400 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r05_b_rehearsal2.json 2> gpurun_out/r05_b_rehearsal2.err
echo "rehearsal rc=$? bytes=$(wc -c < gpurun_out/r05_b_rehearsal2.json)"; head -c 1500 gpurun_out/r05_b_rehearsal2.json
