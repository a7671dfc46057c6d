#!/bin/bash
# round 5, call 5: whole GPU suite on the current tree; sharded rank: search wait by stamp vs stream synchronisation, skin sweep
mkdir -p gpurun_out; OUT=gpurun_out/r05_e.log; : > $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r05_e_tests.log 2>&1; echo "tests rc=$?" >> $OUT; tail -5 gpurun_out/r05_e_tests.log >> $OUT
echo "--- one rank of 8 x 125000, stand-in collective: search wait" >> $OUT
for rep in 1 2; do
for lat in 10 20; do
  timeout -k 10 200 python tools/sharded_interior_alone.py $lat 600 2>/dev/null | sed 's/^/stamp /' >> $OUT
  MRS_SEARCH_WAIT=sync timeout -k 10 200 python tools/sharded_interior_alone.py $lat 600 2>/dev/null | sed 's/^/sync  /' >> $OUT
done
done
echo "--- skin of sharded swarms (MRS_SKIN_SHARDED; default 1.0)" >> $OUT
for sk in 0_75 1_25 1_5; do
  for lat in 10 20; do
    MRS_SWARM_LIB=$PWD/variants/libmrs_collideflag__DMRS_SKIN_SHARDED_$sk.so timeout -k 10 200 python tools/sharded_interior_alone.py $lat 600 2>/dev/null | sed "s/^/skin $sk /" >> $OUT
  done
done
cat $OUT
