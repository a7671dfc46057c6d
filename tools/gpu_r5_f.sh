#!/bin/bash
# round 5, call 6: skin of sharded swarms, three repetitions per point (one rank of 8 x 125000, stand-in collective, 600 ticks)
mkdir -p gpurun_out; OUT=gpurun_out/r05_f.log; : > $OUT
for rep in 1 2 3; do
  for lat in 10 20; do
    timeout -k 10 200 python tools/sharded_interior_alone.py $lat 600 2>/dev/null | sed "s/^/skin 1.0   /" >> $OUT
    for sk in 0_5 0_625 0_75 0_875; do
      MRS_SWARM_LIB=$PWD/variants/libmrs_collideflag__DMRS_SKIN_SHARDED_$sk.so timeout -k 10 200 python tools/sharded_interior_alone.py $lat 600 2>/dev/null | sed "s/^/skin $sk /" >> $OUT
    done
  done
done
sort $OUT | cut -c1-150
echo "--- serial-form launches back to back (no collective), measurement only"
MRS_SHARD_SPLIT=0 MRS_EXP_SPLIT_SKIP=2 timeout -k 10 200 python tools/sharded_interior_alone.py 10 600 2>&1 | tail -1
MRS_SHARD_SPLIT=0 timeout -k 10 200 python tools/sharded_interior_alone.py 10 600 2>&1 | tail -1
MRS_SHARD_SPLIT=0 timeout -k 10 200 python tools/sharded_interior_alone.py 0 600 2>&1 | tail -1
