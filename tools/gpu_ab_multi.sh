# same-box comparison of the tree's library and several variants on the position and position+collisions workloads (two rounds)
run() { for w in position position+collisions; do env $2 timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --traffic off --sub-records off --config5 off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1'.ljust(50), d['config']['workload'][:30].ljust(30), round(d['ms_per_step']*1e3,2))"; done; }
for round in 1 2; do run tree X=1; for v in "$@"; do run $(basename $v) MRS_SWARM_LIB=$v; done; done
