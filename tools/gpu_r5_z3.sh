#!/bin/bash
# round 5, final library: kernel timeline of one rank (halo search included) and the rank-cost table, as tools/gpu_final.sh writes them
set -o pipefail
OUT=gpurun_out/final_r05; mkdir -p $OUT
bash tools/gpu_rank_trace.sh 125000 10 > $OUT/rank_trace_10us.txt 2>&1 || { tail $OUT/rank_trace_10us.txt; exit 1; }
f=$(ls -t gpurun_out/ranktrace_125000/prof/*/*_kernel_trace.csv | head -1); python tools/search_timeline.py $f > $OUT/search_timeline.txt 2>&1; cat $OUT/search_timeline.txt | cut -c1-100
python tools/search_timeline.py $f 2 > $OUT/search_timeline_b.txt 2>&1; tail -1 $OUT/search_timeline_b.txt | cut -c1-60
: > $OUT/sharded_rank_cost.log
for lat in 0 10 20 30; do for form in split serial; do timeout -k 10 300 python tools/sharded_rank_cost.py 125000 8 600 $lat $form 2>/dev/null | cut -c1-260 >> $OUT/sharded_rank_cost.log || exit 1; done; done
cut -c1-120 $OUT/sharded_rank_cost.log
