set -x
mkdir -p gpurun_out/r2c
timeout -k 10 600 python -m pytest tests/test_export_sets_gpu.py tests/test_config5_gpu.py "tests/test_parity_gpu.py::test_neighbour_lists_are_reused_and_rebuilt_with_identical_results" -x -q -m gpu -s > gpurun_out/r2c/tests.log 2>&1; echo "tests rc=$?"
grep -i "export-set\|config 5\|fused\|passed\|failed" gpurun_out/r2c/tests.log | cut -c1-900
for w in position+collisions; do
timeout -k 10 300 python bench.py --no-cpu-baseline --workload $w > gpurun_out/r2c/bench_coll.json 2> gpurun_out/r2c/bench_coll.err; echo "bench rc=$?"
MRS_FUSED_COLLISIONS=0 timeout -k 10 300 python bench.py --no-cpu-baseline --workload $w > gpurun_out/r2c/bench_coll_unfused.json 2> gpurun_out/r2c/bench_coll_unfused.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline --workload $w --uavs 50000 > gpurun_out/r2c/bench_coll_50k.json 2> gpurun_out/r2c/bench_coll_50k.err; echo "bench rc=$?"
done
timeout -k 10 300 python bench.py --no-cpu-baseline --workload position > gpurun_out/r2c/bench_pos.json 2> gpurun_out/r2c/bench_pos.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline --config5 on --steps 200 --warmup 20 > gpurun_out/r2c/bench_c5.json 2> gpurun_out/r2c/bench_c5.err; echo "bench rc=$?"
tail -c 300 gpurun_out/r2c/*.err
