set -x
OUT=gpurun_out/shardchk; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_export_sets_gpu.py tests/test_config5_gpu.py "tests/test_parity_gpu.py::test_library_driven_sharded_tick_with_a_one_rank_communicator" -x -q -m gpu -s > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log | cut -c1-400; exit 1; }
grep -i "export-set exchange\|config 5\|passed" $OUT/tests.log | cut -c1-600
timeout -k 10 300 python tools/sharded_rank_cost.py 125000 8 400 export > $OUT/rank_cost.log 2>&1; tail -1 $OUT/rank_cost.log | cut -c1-400
timeout -k 10 300 python bench.py --no-cpu-baseline --config5 on --steps 200 --warmup 20 2> $OUT/c5.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config5'])"
