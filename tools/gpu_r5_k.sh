#!/bin/bash
# round 5, call 11: PID columns of the whole cascade requested at its start (-DMRS_PID_AT_CASCADE=1) — parity, then A/B
mkdir -p gpurun_out; OUT=gpurun_out/r05_k.log; : > $OUT
V=$PWD/variants/libmrs_stepflag__DMRS_PID_AT_CASCADE_1.so
MRS_SWARM_LIB=$V timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_components_gpu.py tests/test_bench_launch_gpu.py tests/test_random_sequences_gpu.py -x -q -m gpu > gpurun_out/r05_k_tests.log 2>&1; echo "tests rc=$?" >> $OUT; tail -3 gpurun_out/r05_k_tests.log >> $OUT
run() { # label env
  for w in position position+collisions; do
    env $2 timeout -k 10 300 python bench.py --workload $w --steps 300 --warmup 100 --no-cpu-baseline --traffic off --sub-records off --config5 off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['config']['workload'][:44].ljust(44), 'wall %.2f device %.2f' % (d['ms_per_step']*1e3, d['device_ms_per_step']*1e3))" >> $OUT
  done
  env $2 timeout -k 10 300 python bench.py --workload position --uavs 50000 --steps 300 --warmup 100 --no-cpu-baseline --traffic off --sub-records off --config5 off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '50k position'.ljust(44), 'wall %.2f device %.2f' % (d['ms_per_step']*1e3, d['device_ms_per_step']*1e3))" >> $OUT
  env $2 timeout -k 10 300 python bench.py --workload config2 --uavs 400 --steps 2000 --warmup 200 --no-cpu-baseline --traffic off --sub-records off --config5 off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '400 position'.ljust(44), 'wall %.2f device %.2f' % (d['ms_per_step']*1e3, d['device_ms_per_step']*1e3))" >> $OUT
  for lat in 10 20; do env $2 timeout -k 10 200 python tools/sharded_interior_alone.py $lat 600 2>/dev/null | sed "s/^/$1 /" | cut -c1-70 >> $OUT; done
}
run base X=1; run pidc MRS_SWARM_LIB=$V; run base X=1; run pidc MRS_SWARM_LIB=$V
cat $OUT
