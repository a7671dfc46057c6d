# collision tick + search kernel time under compile-time variants of collide.hip (variants/ from tools/build_variants.sh collideflag ...)
export TMPDIR=/tmp
OUT=gpurun_out/cvar; rm -rf $OUT; mkdir -p $OUT
run() { tag=$1; lib=$2
  MRS_SWARM_LIB=$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python bench.py --steps 600 --warmup 60 --no-cpu-baseline --traffic off --workload position+collisions ${EXTRA:-} > $OUT/$tag.json 2> $OUT/$tag.err || { tail -3 $OUT/$tag.err; return 1; }
  python - $tag <<'PY'
import csv,glob,sys
f=sorted(glob.glob('gpurun_out/cvar/%s/*/*kernel_stats.csv'%sys.argv[1]))[-1]
out=[]
for r in csv.DictReader(open(f)):
    for key in ('k_query','k_pack_insert','mrs_uav_step_coll'):
        if key in r['Name']: out.append('%s %d x %.1f us'%(key, int(r['Calls']), float(r['AverageNs'])/1e3))
print(sys.argv[1].ljust(28), ' | '.join(out), flush=True)
PY
}
run base "" || exit 1
for v in variants/libmrs_collideflag_*.so; do run $(basename $v .so | sed 's/libmrs_collideflag_//') $PWD/$v || exit 1; done
