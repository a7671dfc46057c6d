# same-box A/B of two builds of the library: the tree's own and variants/libmrs_base_head.so (or $1), four workloads each, twice
BASE=${1:-variants/libmrs_base_head.so}
run() { # label, env
  for w in actuator position position+collisions; do
    env $2 timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --traffic off --sub-records off --config5 off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['config']['workload'][:44].ljust(44), round(d['device_ms_per_step']*1e3,2))"
  done
  env $2 timeout -k 10 300 python bench.py --uavs 1000000 --steps 300 --no-cpu-baseline --traffic off --sub-records off --config5 off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '1M actuator'.ljust(44), round(d['device_ms_per_step']*1e3,2))"
}
run new X=1; run base MRS_SWARM_LIB=$BASE; run new X=1; run base MRS_SWARM_LIB=$BASE
