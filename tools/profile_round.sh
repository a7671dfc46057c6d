#!/bin/bash
# Produces the rocprofv3 evidence for profiles/ (run on the GPU box through gpurun; outputs under gpurun_out/profile_round).
# Kernel-trace/stats and every --pmc pass are separate runs (gpurun refuses combined modes).
set -u
OUT=${PROFILE_OUT:-gpurun_out/profile_round}
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="${BENCH_ARGS:---steps 300 --warmup 50 --no-cpu-baseline} --traffic profile"  # (no profiler inside a profiled run)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py $ARGS > $OUT/bench_under_trace.json 2> $OUT/trace.err
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d" " -f1)
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$tag -- python bench.py $ARGS > /dev/null 2> $OUT/pmc_$tag.err
done
# FETCH_SIZE calibration on a kernel with a known byte count and the same 8-B/lane SoA access pattern
hipcc -O3 --offload-arch=gfx950 tools/mem_floor.hip -o /tmp/mem_floor 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -- /tmp/mem_floor > $OUT/cal_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cal_write -- /tmp/mem_floor > $OUT/cal_write.log 2>&1
python tools/summarize_profile.py $OUT > $OUT/summary.md
cat $OUT/summary.md
