set -x
export TMPDIR=/tmp
OUT=gpurun_out/r2d
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_coll -- python bench.py --steps 500 --warmup 50 --no-cpu-baseline --workload position+collisions > $OUT/bench_coll_trace.json 2> $OUT/trace_coll.err
f=$(find $OUT/trace_coll -name "*kernel_stats.csv" | head -1); cat $f | cut -c1-200
