#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4h
timeout -k 10 700 python -m pytest tests/test_sharded_chaos_gpu.py tests/test_export_sets_gpu.py tests/test_config5_gpu.py tests/test_sharded_multiprocess_gpu.py tests/test_peer_window_gpu.py -m gpu -q -x > gpurun_out/r4h/tests.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed|FAILED|^E  " gpurun_out/r4h/tests.log | tail -8
for lat in 10 20; do
  timeout -k 10 200 python tools/sharded_rank_cost.py 125000 8 600 $lat split 2>&1 | tail -1
done
