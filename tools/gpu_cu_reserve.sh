# split sharded tick with the boundary chain on reserved compute units (MRS_SPLIT_CU_RESERVE): one rank of 8 x 125 000, stand-in collective
OUT=$PWD/gpurun_out/cureserve; rm -rf $OUT; mkdir -p $OUT
for r in ${RESERVES:-0 16 32}; do
  for lat in ${LATS:-20 10 0}; do
    echo "reserve $r:" >> $OUT/log.txt
    MRS_SPLIT_CU_RESERVE=$r timeout -k 10 200 python tools/sharded_rank_cost.py 125000 8 600 $lat split >> $OUT/log.txt 2>$OUT/err.txt || { tail -20 $OUT/err.txt; exit 1; }
  done
done
cat $OUT/log.txt
