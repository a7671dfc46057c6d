OUT=gpurun_out/qt; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 python -m pytest "$@" -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -50 $OUT/tests.log | cut -c1-300; exit 1; }
tail -3 $OUT/tests.log
