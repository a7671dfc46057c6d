#!/bin/bash
# round 5: halo tests, kernel timeline of a halo search and the per-rank tick (medians of five) on the final form of the kernels
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_search_halo_gpu.py tests/test_export_sets_gpu.py tests/test_sharded_chaos_gpu.py tests/test_sharded_multiprocess_gpu.py -x -q -m gpu > gpurun_out/r05_z2_tests.log 2>&1; rc=$?; tail -2 gpurun_out/r05_z2_tests.log; [ $rc -eq 0 ] || exit $rc
bash tools/gpu_r5_z.sh > gpurun_out/r05_z2_timeline.log 2>&1 || exit 1
grep -E "halo|bbox|pack" gpurun_out/r05_z_search_timeline.txt
bash tools/gpu_r5_y2.sh
