#!/bin/bash
# round 5: kernel timeline of a halo search on one rank of 8 x 125 000 (stand-in collective, 10 us)
set -o pipefail
mkdir -p gpurun_out
bash tools/gpu_rank_trace.sh 125000 10 MRS_HALO_TRACE=1 > gpurun_out/r05_z_trace.txt 2>&1 || { tail -20 gpurun_out/r05_z_trace.txt; exit 1; }
f=$(ls -t gpurun_out/ranktrace_125000/prof/*/*_kernel_trace.csv | head -1)
python tools/search_timeline.py $f > gpurun_out/r05_z_search_timeline.txt 2>&1
cat gpurun_out/r05_z_search_timeline.txt
grep "mrs halo" gpurun_out/ranktrace_125000/trace.log | tail -4
