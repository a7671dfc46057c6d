OUT=gpurun_out/soak; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 500 python tests/campaigns/soak.py 20000 2000 literal 30 local > $OUT/soak_30.log 2>&1 || { tail -5 $OUT/soak_30.log; exit 1; }
tail -n 2 $OUT/soak_30.log | cut -c1-250
timeout -k 10 500 python tests/campaigns/soak.py 20000 1000 literal 8 local > $OUT/soak_8.log 2>&1 || { tail -5 $OUT/soak_8.log; exit 1; }
tail -n 2 $OUT/soak_8.log | cut -c1-250
timeout -k 10 500 python tests/campaigns/soak.py 20000 1000 fast 16 sharded > $OUT/soak_16s.log 2>&1 || { tail -5 $OUT/soak_16s.log; exit 1; }
tail -n 2 $OUT/soak_16s.log | cut -c1-250
