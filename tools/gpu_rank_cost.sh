set -x
export TMPDIR=/tmp
OUT=gpurun_out/rank_cost
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 python tools/sharded_rank_cost.py 125000 8 400 export > $OUT/export.log 2>&1; tail -2 $OUT/export.log | cut -c1-600
timeout -k 10 300 python tools/sharded_rank_cost.py 125000 8 300 full > $OUT/full.log 2>&1; tail -2 $OUT/full.log | cut -c1-600
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python tools/sharded_rank_cost.py 125000 8 400 export > $OUT/trace.log 2> $OUT/trace.err
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); cut -c1-150 $f | head -16
