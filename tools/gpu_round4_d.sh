#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4d
python -m pytest tests/test_facade_cpp.py tests/test_pool_primitives_gpu.py tests/test_sharded_chaos_gpu.py tests/test_export_sets_gpu.py tests/test_sharded_multiprocess_gpu.py tests/test_peer_window_gpu.py tests/test_config5_gpu.py tests/test_facade_cpp.py -m gpu -q > gpurun_out/r4d/tests.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed|FAILED|^E  " gpurun_out/r4d/tests.log | tail -12
for lat in 10 20; do
  timeout -k 10 200 python tools/sharded_rank_cost.py 125000 8 600 $lat split 2>&1 | tail -1
  MRS_EARLY_SEARCH=0 timeout -k 10 200 python tools/sharded_rank_cost.py 125000 8 600 $lat split 2>&1 | tail -1
done
bash tools/gpu_rank_trace.sh 125000 10 > gpurun_out/r4d/trace10.txt 2>&1; tail -3 gpurun_out/r4d/trace10.txt
f=$(ls -t gpurun_out/ranktrace_125000/prof/*/*_kernel_trace.csv | head -1)
python tools/search_timeline.py $f > gpurun_out/r4d/search_timeline.txt 2>&1; cat gpurun_out/r4d/search_timeline.txt
