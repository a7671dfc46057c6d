#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4c
python -m pytest tests/test_facade_cpp.py tests/test_ground_clamp_gpu.py tests/test_pool_primitives_gpu.py -m gpu -q -s -k "pooled or fast_single or copy_uavs" > gpurun_out/r4c/tests.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed|FAILED|facade loop|FAST,|^E  " gpurun_out/r4c/tests.log | tail -12
for lat in 10 20; do
  timeout -k 10 200 python tools/sharded_rank_cost.py 125000 8 600 $lat split 2>&1 | tail -1
  MRS_SWARM_LIB=$PWD/variants/libmrs_stepflag__DMRS_EXP_BND_NOCASCADE_1.so timeout -k 10 200 python tools/sharded_rank_cost.py 125000 8 600 $lat split 2>&1 | tail -1
  MRS_INTERIOR_NT=0 MRS_SWARM_LIB=$PWD/variants/libmrs_stepflag__DMRS_EXP_BND_NOCASCADE_1.so timeout -k 10 200 python tools/sharded_rank_cost.py 125000 8 600 $lat split 2>&1 | tail -1
done
