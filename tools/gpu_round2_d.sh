set -x
mkdir -p gpurun_out/r2e
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r2e/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -5 gpurun_out/r2e/gpu_tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline --workload position+collisions > gpurun_out/r2e/bench_coll.json 2> gpurun_out/r2e/bench_coll.err; echo "bench rc=$?"
MRS_WARN_FRACTION=0.85 timeout -k 10 300 python bench.py --no-cpu-baseline --workload position+collisions > gpurun_out/r2e/bench_coll_w85.json 2> gpurun_out/r2e/bench_coll_w85.err; echo "bench rc=$?"
MRS_WARN_FRACTION=0.6 timeout -k 10 300 python bench.py --no-cpu-baseline --workload position+collisions > gpurun_out/r2e/bench_coll_w60.json 2> gpurun_out/r2e/bench_coll_w60.err; echo "bench rc=$?"
MRS_FUSED_LEAD=2 timeout -k 10 300 python bench.py --no-cpu-baseline --workload position+collisions > gpurun_out/r2e/bench_coll_lead2.json 2> gpurun_out/r2e/bench_coll_lead2.err; echo "bench rc=$?"
MRS_FUSED_LEAD=8 timeout -k 10 300 python bench.py --no-cpu-baseline --workload position+collisions > gpurun_out/r2e/bench_coll_lead8.json 2> gpurun_out/r2e/bench_coll_lead8.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline --workload position+collisions --uavs 50000 > gpurun_out/r2e/bench_coll_50k.json 2> gpurun_out/r2e/bench_coll_50k.err; echo "bench rc=$?"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2e/trace_coll -- python bench.py --steps 500 --warmup 50 --no-cpu-baseline --workload position+collisions > gpurun_out/r2e/bench_coll_trace.json 2> gpurun_out/r2e/trace_coll.err
f=$(find gpurun_out/r2e/trace_coll -name "*kernel_stats.csv" | head -1); cat $f | cut -c1-160
