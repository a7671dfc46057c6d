#!/bin/bash
# One bench line per combination of environment switches / library variants, on one box in one call — what the one-off sweep scripts
# of rounds 1-3 did (skin, warning fraction, stream priority, interior kernel, split thresholds, sizes, densities; MEASUREMENTS.md).
# usage: tools/gpu_sweep.sh "<bench.py args>" VAR=a,b,c [VAR2=x,y ...]     (run on the GPU box through gpurun)
#   e.g. tools/gpu_sweep.sh "--workload position+collisions" MRS_WARN_FRACTION=0.75,0.65 MRS_FUSED_LEAD=2,3
#        tools/gpu_sweep.sh "--uavs 4000000 --steps 100" MRS_SWARM_LIB=,variants/libmrs_skin_1_0.so      (empty value: variable unset)
# Prints device and wall time per step and, for collision workloads, searches / stalls / replayed launches.
ARGS=$1; shift
combos=("")
for spec in "$@"; do
  var=${spec%%=*}; IFS=, read -ra vals <<< "${spec#*=}"
  [ "${spec: -1}" = "," ] && vals+=("")
  next=()
  for c in "${combos[@]}"; do for v in "${vals[@]}"; do next+=("$c $var=$v"); done; done
  combos=("${next[@]}")
done
for c in "${combos[@]}"; do
  envs=(); for kv in $c; do [ -n "${kv#*=}" ] && envs+=("$kv"); done
  env "${envs[@]}" timeout -k 10 300 python bench.py $ARGS --no-cpu-baseline --traffic off --sub-records off --config5 off 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read()); c = d['config']
extra = ' searches %s stalls %s replayed %s' % (c['neighbour_searches'], c['stale_list_stalls'], c['launches_replayed']) if 'neighbour_searches' in c else ''
print('%-60s device %.2f wall %.2f us per step%s' % ('$c'.strip() or '(defaults)', d['device_ms_per_step'] * 1e3, d['ms_per_step'] * 1e3, extra))"
done
