#!/bin/bash
# round 5: the halo exchange of a search tick — its tests, the sharded suites around it, and what it does to a rank's tick when the
# collective's bytes cost wire time (MRS_STANDIN_GBPS).   usage: gpu_r5_y.sh [timing]   (timing: skip the tests)
set -o pipefail
ROOT=$(pwd)
mkdir -p gpurun_out
if [ "$1" != timing ]; then
MRS_HALO_TRACE=1 timeout -k 10 900 python -m pytest tests/test_search_halo_gpu.py tests/test_export_sets_gpu.py tests/test_sharded_chaos_gpu.py tests/test_config5_gpu.py tests/test_sharded_multiprocess_gpu.py tests/test_simulator_loop.py -x -q -m gpu -s > gpurun_out/r05_y_tests.log 2>&1
rc=$?
echo "tests rc=$rc"; tail -5 gpurun_out/r05_y_tests.log
[ $rc -eq 0 ] || exit $rc
fi
: > gpurun_out/r05_y_rank.log
for rep in 1 2; do
for halo in 1 0; do
  for gbps in 0 300 150; do
    echo "== MRS_SEARCH_HALO=$halo MRS_STANDIN_GBPS=$gbps" >> gpurun_out/r05_y_rank.log
    MRS_HALO_TRACE=1 MRS_SEARCH_HALO=$halo MRS_STANDIN_GBPS=$gbps timeout -k 10 200 python tools/sharded_interior_alone.py 10 600 >> gpurun_out/r05_y_rank.log 2>&1 || exit 1
  done
done
done
grep -v "amdgpu.ids" gpurun_out/r05_y_rank.log | grep -v "flags 0$"
