OUT=gpurun_out/gdb; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 /opt/rocm/bin/rocgdb -batch -ex run -ex bt -ex "info threads" --args python -m pytest "$@" -x -q -m gpu -p no:faulthandler > $OUT/gdb.log 2>&1
grep -n "SIGSEGV" -A40 $OUT/gdb.log | cut -c1-250 | head -80
