set -x
mkdir -p gpurun_out/r2a
timeout -k 10 900 python -m pytest tests/test_bench_launch_gpu.py tests/test_config5_gpu.py -x -q -m gpu -s > gpurun_out/r2a/newtests.log 2>&1; echo "newtests rc=$?"
tail -5 gpurun_out/r2a/newtests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r2a/bench_driver_like.json 2> gpurun_out/r2a/bench_driver_like.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r2a/bench_default.json 2> gpurun_out/r2a/bench_default.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline --workload position+collisions > gpurun_out/r2a/bench_coll.json 2> gpurun_out/r2a/bench_coll.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline --uavs 4000000 --steps 200 --warmup 20 > gpurun_out/r2a/bench_4M.json 2> gpurun_out/r2a/bench_4M.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline --config5 on --steps 100 --warmup 10 > gpurun_out/r2a/bench_c5.json 2> gpurun_out/r2a/bench_c5.err; echo "bench rc=$?"
tail -c 600 gpurun_out/r2a/*.err
