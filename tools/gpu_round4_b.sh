#!/bin/bash
# round 4: the whole GPU suite (no -x: every failure at once), then the facade loop's own output
set -o pipefail
mkdir -p gpurun_out/r4b
python -m pytest tests -m gpu -q -s > gpurun_out/r4b/gpu_tests.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r4b/gpu_tests.log
grep -E "passed|failed|FAILED|facade loop|FAST,|LITERAL," gpurun_out/r4b/gpu_tests.log | tail -30
./tests/cpp/facade_loop_test > gpurun_out/r4b/facade_loop.txt 2>&1; grep -v STATE gpurun_out/r4b/facade_loop.txt
./tests/cpp/facade_loop_test single > gpurun_out/r4b/facade_loop_single.txt 2>&1; grep -v STATE gpurun_out/r4b/facade_loop_single.txt
