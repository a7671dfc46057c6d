#!/bin/bash
# round 5, VERDICT r4 item 1 step 1: the fused collision tick on spatially ordered slots, with and without the overlapped-halves
# patch of round 4 (variants/libmrs_overlap.so = profiles/r04_overlapped_ticks_experiment.patch re-applied on HEAD)
mkdir -p gpurun_out
OUT=gpurun_out/r05_order.log; : > $OUT
run() { # label, env...
  label=$1; shift
  for order in random xcell morton; do
    env "$@" timeout -k 10 200 python bench.py --workload position+collisions --steps 300 --warmup 100 --order $order --no-cpu-baseline --traffic ${TRAFFIC:-off} --sub-records off --config5 off 2>gpurun_out/r05_order.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; rc=d['roofline_collision']; r=d['roofline']
print('$label', '$order'.ljust(7), 'wall %.2f device %.2f us/tick; searches %d stalls %d replayed %d ahead %d; search %.1f us; traffic %s' % (d['ms_per_step']*1e3, d['device_ms_per_step']*1e3, c['neighbour_searches'], c['stale_list_stalls'], c['launches_replayed'], c['searches_queued_ahead'], rc['search_ms']*1e3, r['traffic']))
" >> $OUT || { echo "$label $order FAILED" >> $OUT; tail -5 gpurun_out/r05_order.err >> $OUT; }
  done
}
TRAFFIC=live run "base           " X=1
run "base repeat    " X=1
run "overlap-lib off" MRS_SWARM_LIB=variants/libmrs_overlap.so MRS_OVERLAP_TICKS=0
run "overlap on w.65" MRS_SWARM_LIB=variants/libmrs_overlap.so MRS_OVERLAP_TICKS=1 MRS_WARN_FRACTION=0.65
run "overlap on w.75" MRS_SWARM_LIB=variants/libmrs_overlap.so MRS_OVERLAP_TICKS=1
cat $OUT
