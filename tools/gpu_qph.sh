OUT=gpurun_out/qph; rm -rf $OUT; mkdir -p $OUT
QPH_LISTS=0 QPH_CLOCK=1 timeout -k 10 200 python tools/query_phases.py 100000 > $OUT/nolists.log 2>&1 || { tail $OUT/nolists.log; exit 1; }
QPH_LISTS=1 QPH_CLOCK=1 timeout -k 10 200 python tools/query_phases.py 100000 > $OUT/lists1.log 2>&1 || { tail $OUT/lists1.log; exit 1; }
QPH_LISTS=1 QPH_CLOCK=2 timeout -k 10 200 python tools/query_phases.py 100000 > $OUT/lists2.log 2>&1 || { tail $OUT/lists2.log; exit 1; }
for f in nolists lists1 lists2; do echo == $f; tail -n 4 $OUT/$f.log; done
