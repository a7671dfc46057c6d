#!/bin/bash
# per-kernel durations of one bench workload (rocprofv3 kernel trace only): bash tools/gpu_kstats.sh <tag> <bench args...>
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/kstats_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py "$@" --traffic profile > $OUT/bench.json 2> $OUT/trace.err
f=$(ls $OUT/trace/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cut -d, -f1-8 "$f" | head -12 | cut -c1-230
