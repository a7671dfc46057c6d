#!/bin/bash
# copies what tools/gpu_final.sh left under gpurun_out/final_<round>/ into profiles/<round>_* (run HERE, after the two gpurun calls)
R=${1:-r05}; S=gpurun_out/final_$R; P=profiles
cp $S/bench_all.jsonl $P/${R}_bench_all.jsonl
cp $S/gpu_tests.log $P/${R}_gpu_tests.log
cp $S/sharded_rank_cost.log $P/${R}_sharded_rank_cost.log
cp $S/search_timeline.txt $P/${R}_search_timeline.txt
cp $S/facade_loop.txt $P/${R}_facade_loop.txt
for f in search_rate search_phases search_pmc; do [ -f $S/$f.txt ] && grep -v "amdgpu.ids" $S/$f.txt > $P/${R}_$f.txt; done
{ echo "# kernel timeline of ONE rank of 8 x 125 000 UAVs in the split form, stand-in collective of 10 us (tools/gpu_rank_trace.sh 125000 10; rocprofv3 --kernel-trace;"
  echo "# queue 1 = boundary launch + collective, queue 2 = interior launch; times in us)"
  echo "kernel,queue,blocks,start_us,end_us,duration_us"
  grep " q [0-9] blocks " $S/rank_trace_10us.txt | sed -E 's/^(.*[^ ]) +q ([0-9]+) blocks ([0-9]+) start ([0-9.-]+) end ([0-9.-]+) dur ([0-9.]+)$/"\1",\2,\3,\4,\5,\6/'; } > $P/${R}_split_tick_timeline.csv
for pair in prof_100k:step_kernel_100k_fast prof_4M:step_kernel_4M_fast prof_coll:collision_tick_100k_fast prof_pos:position_cascade_100k_fast; do
  d=${pair%%:*}; n=${pair##*:}
  cp $S/$d/bench_under_trace.json $P/${R}_${n}_bench_under_trace.json
  cp $(ls -t $S/$d/trace/*/*kernel_stats.csv | head -1) $P/${R}_${n}_kernel_stats.csv
  cp $S/$d/summary.md $P/${R}_${n}_summary.md
  [ -f $S/$d/summary.json ] && cp $S/$d/summary.json $P/${R}_${n}_summary.json
done
ls $P | grep -c "^${R}_"
