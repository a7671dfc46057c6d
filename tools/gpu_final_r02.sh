set -x
export TMPDIR=/tmp
OUT=gpurun_out/final_r02
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1 || { echo "gpu tests failed"; tail -40 $OUT/gpu_tests.log; exit 1; }
tail -3 $OUT/gpu_tests.log
: > $OUT/bench_all.jsonl
b() { timeout -k 10 300 python bench.py "$@" 2>> $OUT/bench.err | tail -1 >> $OUT/bench_all.jsonl; }
b
b --steps 20 --warmup 5 --no-cpu-baseline
b --uavs 4000000 --steps 200 --warmup 20 --no-cpu-baseline
b --uavs 1000000 --no-cpu-baseline
b --workload position --no-cpu-baseline
b --workload position+collisions
b --workload position+collisions --uavs 50000 --no-cpu-baseline
b --workload position+collisions --volume-per-uav 16 --no-cpu-baseline
b --workload position --uavs 400 --no-cpu-baseline
b --config5 on --steps 200 --warmup 20 --no-cpu-baseline
wc -l $OUT/bench_all.jsonl
PROFILE_OUT=$OUT/prof_100k BENCH_ARGS="--steps 300 --warmup 50 --no-cpu-baseline" timeout -k 10 600 bash tools/profile_round.sh > $OUT/prof_100k.log 2>&1
PROFILE_OUT=$OUT/prof_4M BENCH_ARGS="--steps 100 --warmup 20 --no-cpu-baseline --uavs 4000000" timeout -k 10 600 bash tools/profile_round.sh > $OUT/prof_4M.log 2>&1
PROFILE_OUT=$OUT/prof_coll BENCH_ARGS="--steps 500 --warmup 50 --no-cpu-baseline --workload position+collisions" timeout -k 10 600 bash tools/profile_round.sh > $OUT/prof_coll.log 2>&1
PROFILE_OUT=$OUT/prof_pos BENCH_ARGS="--steps 300 --warmup 50 --no-cpu-baseline --workload position" timeout -k 10 600 bash tools/profile_round.sh > $OUT/prof_pos.log 2>&1
timeout -k 10 300 python tools/sharded_rank_cost.py 125000 8 400 export > $OUT/sharded_rank_cost_export.log 2>&1
timeout -k 10 300 python tools/sharded_rank_cost.py 125000 8 300 full > $OUT/sharded_rank_cost_full.log 2>&1
tail -1 $OUT/sharded_rank_cost_export.log | cut -c1-300
python __graft_entry__.py > $OUT/entry.log 2>&1; python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
