# per-rank cost of the sharded tick with a stand-in collective of 20 / 10 / 30 / 0 us, split and serial form (tools/sharded_rank_cost.py),
# then a kernel trace of the split form at 20 us
OUT=$PWD/gpurun_out/rankcost; rm -rf $OUT; mkdir -p $OUT
for lat in 20 10 30 0; do
  timeout -k 10 200 python tools/sharded_rank_cost.py 125000 8 600 $lat split >> $OUT/log.txt 2>$OUT/err.txt || { tail -20 $OUT/err.txt; exit 1; }
  timeout -k 10 200 python tools/sharded_rank_cost.py 125000 8 600 $lat serial >> $OUT/log.txt 2>$OUT/err.txt || { tail -20 $OUT/err.txt; exit 1; }
done
cat $OUT/log.txt
ROOT=$PWD; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $ROOT/tools/sharded_rank_cost.py 125000 8 600 20 split > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
cd $ROOT
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-200
