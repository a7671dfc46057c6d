#!/bin/bash
# round 5, call 12: warning fraction of the skin for sharded swarms (MRS_WARN_FRACTION, default 0.75), one rank of 8 x 125000, 10 / 20 us
mkdir -p gpurun_out; OUT=gpurun_out/r05_l.log; : > $OUT
for rep in 1 2 3; do
  for f in 0.75 0.8 0.85 0.9; do
    for lat in 10 20; do
      MRS_WARN_FRACTION=$f timeout -k 10 200 python tools/sharded_interior_alone.py $lat 1200 2>/dev/null | sed "s/^/warn $f /" | cut -c1-140 >> $OUT
    done
  done
done
sort $OUT
