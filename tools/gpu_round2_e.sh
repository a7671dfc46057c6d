set -x
mkdir -p gpurun_out/r2f
timeout -k 10 200 python -m pytest "tests/test_parity_gpu.py::test_neighbour_lists_are_reused_and_rebuilt_with_identical_results" "tests/test_parity_gpu.py::test_neighbour_lists_follow_host_writes" tests/test_random_sequences_gpu.py tests/test_components_gpu.py tests/test_l0_classes.py -x -q -m gpu -s > gpurun_out/r2f/risky.log 2>&1 || { echo "risky tests failed rc=$?"; tail -40 gpurun_out/r2f/risky.log; exit 1; }
tail -3 gpurun_out/r2f/risky.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2f/gpu_tests.log 2>&1 || { echo "gpu tests failed rc=$?"; tail -40 gpurun_out/r2f/gpu_tests.log; exit 1; }
tail -3 gpurun_out/r2f/gpu_tests.log
for v in "" "MRS_WARN_FRACTION=0.6" "MRS_FUSED_LEAD=2" "MRS_FUSED_LEAD=3"; do
  tag=$(echo "$v" | tr -c 'A-Za-z0-9\n' '_')
  env $v timeout -k 10 200 python bench.py --no-cpu-baseline --workload position+collisions > gpurun_out/r2f/bench_coll_$tag.json 2> gpurun_out/r2f/bench_coll_$tag.err && echo "bench $v ok"
done
timeout -k 10 200 python bench.py --no-cpu-baseline --workload position+collisions --uavs 50000 > gpurun_out/r2f/bench_coll_50k.json 2> gpurun_out/r2f/bench_coll_50k.err && echo ok
timeout -k 10 200 python bench.py --no-cpu-baseline --workload position > gpurun_out/r2f/bench_pos.json 2> gpurun_out/r2f/bench_pos.err && echo ok
MRS_SPLIT_STREAMS=0 timeout -k 10 200 python bench.py --no-cpu-baseline --workload position > gpurun_out/r2f/bench_pos_1stream.json 2> gpurun_out/r2f/bench_pos_1stream.err && echo ok
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2f/trace_coll -- python bench.py --steps 500 --warmup 50 --no-cpu-baseline --workload position+collisions > gpurun_out/r2f/bench_coll_trace.json 2> gpurun_out/r2f/trace_coll.err
f=$(find gpurun_out/r2f/trace_coll -name "*kernel_stats.csv" | head -1); cat $f | cut -c1-160
