set -x
export TMPDIR=/tmp
OUT=gpurun_out/export_rate
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 python tools/export_tick_rate.py 125000 8 300 export slabs > $OUT/export_slabs.log 2>&1; tail -1 $OUT/export_slabs.log | cut -c1-700
timeout -k 10 300 python tools/export_tick_rate.py 125000 8 300 export index > $OUT/export_index.log 2>&1; tail -1 $OUT/export_index.log | cut -c1-700
timeout -k 10 300 python tools/export_tick_rate.py 125000 8 200 full slabs > $OUT/full_slabs.log 2>&1; tail -1 $OUT/full_slabs.log | cut -c1-700
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python tools/export_tick_rate.py 125000 8 300 export slabs > $OUT/trace.log 2> $OUT/trace.err
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); cut -c1-150 $f | head -14
