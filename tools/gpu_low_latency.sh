# the sharded tick at the latencies of the peer-window exchange (3-8 us): split with / without non-temporal interior launch, serial form
OUT=$PWD/gpurun_out/lowlat; rm -rf $OUT; mkdir -p $OUT
for lat in ${LATS:-3 6 8}; do
  for v in "MRS_INTERIOR_NT=1 split" "MRS_INTERIOR_NT=0 split" "MRS_INTERIOR_NT=1 serial"; do
    set -- $v
    echo "$1 $2:" >> $OUT/log.txt
    env $1 timeout -k 10 200 python tools/sharded_rank_cost.py 125000 8 600 $lat $2 2>$OUT/err.txt | cut -c1-120 >> $OUT/log.txt || { tail -20 $OUT/err.txt; exit 1; }
  done
done
cat $OUT/log.txt
