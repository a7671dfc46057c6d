# same-box A/B of the tree's library against a variant ($1) on the position and position+collisions workloads, twice each
V=${1:?variant .so}
run() { for w in position position+collisions; do env $2 timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --traffic off --sub-records off --config5 off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['config']['workload'][:44].ljust(44), round(d['ms_per_step']*1e3,2))"; done; }
run new X=1; run variant MRS_SWARM_LIB=$V; run new X=1; run variant MRS_SWARM_LIB=$V
