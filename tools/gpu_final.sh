#!/bin/bash
# The one call behind profiles/<round>_*: pytest -m gpu, the bench lines, the rocprofv3 evidence (tools/profile_round.sh), the sharded
# rank's cost and timeline, the facade loop, smoke().   usage (through gpurun, two calls of at most 20 minutes):
#   bash tools/gpu_final.sh r04 tests      (suite + bench lines)        bash tools/gpu_final.sh r04 profiles      (everything else)
R=${1:-r05}; PART=${2:-all}
set -x
export TMPDIR=/tmp
OUT=gpurun_out/final_$R
mkdir -p $OUT
if [ $PART != profiles ]; then
timeout -k 10 1000 python -m pytest tests -x -q -m gpu -s > $OUT/gpu_tests.log 2>&1 || { echo "gpu tests failed"; tail -40 $OUT/gpu_tests.log; exit 1; }
tail -3 $OUT/gpu_tests.log
: > $OUT/bench_all.jsonl
b() { timeout -k 10 600 python bench.py "$@" 2>> $OUT/bench.err | tail -1 >> $OUT/bench_all.jsonl; }
S="--sub-records off --config5 off --no-cpu-baseline"
b
b --steps 20 --warmup 5
b --uavs 1000000 $S
b --workload position $S
b --workload position+collisions --uavs 50000 $S
b --workload position+collisions --volume-per-uav 16 $S
wc -l $OUT/bench_all.jsonl
fi
[ $PART = tests ] && exit 0
P="--sub-records off --config5 off --no-cpu-baseline"
PROFILE_OUT=$OUT/prof_100k BENCH_ARGS="--steps 300 --warmup 50 $P" timeout -k 10 600 bash tools/profile_round.sh > $OUT/prof_100k.log 2>&1
PROFILE_OUT=$OUT/prof_4M BENCH_ARGS="--steps 100 --warmup 20 --uavs 4000000 $P" timeout -k 10 600 bash tools/profile_round.sh > $OUT/prof_4M.log 2>&1
PROFILE_OUT=$OUT/prof_coll BENCH_ARGS="--steps 500 --warmup 50 --workload position+collisions $P" timeout -k 10 600 bash tools/profile_round.sh > $OUT/prof_coll.log 2>&1
PROFILE_OUT=$OUT/prof_pos BENCH_ARGS="--steps 300 --warmup 50 --workload position $P" timeout -k 10 600 bash tools/profile_round.sh > $OUT/prof_pos.log 2>&1
for lat in 0 10 20 30; do for form in split serial; do timeout -k 10 300 python tools/sharded_rank_cost.py 125000 8 600 $lat $form 2>/dev/null | cut -c1-260 >> $OUT/sharded_rank_cost.log; done; done
cat $OUT/sharded_rank_cost.log
bash tools/gpu_rank_trace.sh 125000 10 > $OUT/rank_trace_10us.txt 2>&1
f=$(ls -t gpurun_out/ranktrace_125000/prof/*/*_kernel_trace.csv | head -1); python tools/search_timeline.py $f > $OUT/search_timeline.txt 2>&1; tail -3 $OUT/search_timeline.txt
python tools/search_rate.py 100000 125000 1000000 --volume 64 16 > $OUT/search_rate.txt 2>&1; tail -6 $OUT/search_rate.txt
V=variants/libmrs_collideflag__DMRS_Q2_CLOCK_1.so  # (bash tools/build_variants.sh collideflag "-DMRS_Q2_CLOCK=1" before the call)
[ -f $V ] && for n in 100000 125000; do MRS_SWARM_LIB=$V python tools/search_phases.py $n 2>&1 | grep -v amdgpu.ids >> $OUT/search_phases.txt; done
bash tools/gpu_pmc_search.sh 100000 > $OUT/search_pmc.txt 2>&1
./tests/cpp/facade_loop_test | grep -v STATE > $OUT/facade_loop.txt 2>&1; ./tests/cpp/facade_loop_test single | grep -v STATE >> $OUT/facade_loop.txt 2>&1; cat $OUT/facade_loop.txt
python __graft_entry__.py > $OUT/entry.log 2>&1; python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
