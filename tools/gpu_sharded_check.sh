set -x
OUT=gpurun_out/shardchk; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 500 python -m pytest tests/test_export_sets_gpu.py tests/test_config5_gpu.py tests/test_simulator_loop.py "tests/test_parity_gpu.py::test_library_driven_sharded_tick_with_a_one_rank_communicator" "tests/test_parity_gpu.py::test_gathered_collisions_reuse_neighbour_lists_over_many_ticks" "tests/test_parity_gpu.py::test_gathered_collisions_from_two_virtual_shards" "tests/test_parity_gpu.py::test_sharded_swarm_world1_on_gpu" -x -q -m gpu -s > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log | cut -c1-400; exit 1; }
grep -i "passed" $OUT/tests.log | cut -c1-600
timeout -k 10 300 python tools/sharded_rank_cost.py 125000 8 400 export > $OUT/rank_cost.log 2>&1; tail -1 $OUT/rank_cost.log | cut -c1-400
timeout -k 10 300 python tools/sharded_rank_cost.py 125000 8 300 full > $OUT/rank_cost_full.log 2>&1; tail -1 $OUT/rank_cost_full.log | cut -c1-400
timeout -k 10 300 python tools/export_tick_rate.py 125000 8 100 export slabs > $OUT/vshards.log 2>&1; tail -1 $OUT/vshards.log | cut -c1-300
