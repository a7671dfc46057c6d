set -x
export TMPDIR=/tmp
# (1) headline: 100 k actuator  (2) 4 M actuator: beyond the Infinity Cache  (3) collision tick 100 k
PROFILE_OUT=gpurun_out/prof_r02_100k BENCH_ARGS="--steps 300 --warmup 50 --no-cpu-baseline" timeout -k 10 600 bash tools/profile_round.sh > gpurun_out/prof_r02_100k.log 2>&1
PROFILE_OUT=gpurun_out/prof_r02_4M BENCH_ARGS="--steps 100 --warmup 20 --no-cpu-baseline --uavs 4000000" timeout -k 10 600 bash tools/profile_round.sh > gpurun_out/prof_r02_4M.log 2>&1
PROFILE_OUT=gpurun_out/prof_r02_coll BENCH_ARGS="--steps 500 --warmup 50 --no-cpu-baseline --workload position+collisions" timeout -k 10 600 bash tools/profile_round.sh > gpurun_out/prof_r02_coll.log 2>&1
tail -30 gpurun_out/prof_r02_4M.log
