#!/bin/bash
# round 5, call 13: fused collision kernels in a three-waves-per-SIMD register budget (big swarms), with 2 / 1 prefetched partners
mkdir -p gpurun_out; OUT=gpurun_out/r05_n.log; : > $OUT
V2=$PWD/variants/libmrs_stepflag__DMRS_COLL_W3_1__DMRS_NPRE_2.so
V1=$PWD/variants/libmrs_stepflag__DMRS_COLL_W3_1__DMRS_NPRE_1.so
MRS_SWARM_LIB=$V2 MRS_COLL_W3=1 timeout -k 10 600 python -m pytest tests/test_bench_launch_gpu.py tests/test_config5_gpu.py tests/test_soak_cut_gpu.py -x -q -m gpu > gpurun_out/r05_n_tests.log 2>&1; echo "tests (w3 forced, NPRE 2) rc=$?" >> $OUT; tail -2 gpurun_out/r05_n_tests.log >> $OUT
run() { # label uavs env...
  label=$1; n=$2; shift; shift
  env "$@" timeout -k 10 300 python bench.py --workload position+collisions --uavs $n --steps 200 --warmup 50 --no-cpu-baseline --traffic off --sub-records off --config5 off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', '$n'.rjust(8), 'wall %.2f device %.2f us/tick  %.3e UAV-steps/s' % (d['ms_per_step']*1e3, d['device_ms_per_step']*1e3, d['value']))" >> $OUT
}
for n in 100000 200000 400000 1000000; do
  run "base             " $n X=1
  run "NPRE2 2 waves    " $n MRS_SWARM_LIB=$V2 MRS_COLL_W3=0
  run "NPRE2 3 waves    " $n MRS_SWARM_LIB=$V2 MRS_COLL_W3=1
  run "NPRE1 3 waves    " $n MRS_SWARM_LIB=$V1 MRS_COLL_W3=1
done
c5() { env "$@" timeout -k 10 300 python bench.py --only-config5 --steps 200 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config5 one rank', '%.1f us/tick' % (d['ms_per_tick']*1e3))"; }
echo "base: $(c5 X=1)" >> $OUT
echo "NPRE2 3 waves: $(c5 MRS_SWARM_LIB=$V2)" >> $OUT
echo "NPRE1 3 waves: $(c5 MRS_SWARM_LIB=$V1)" >> $OUT
cat $OUT
