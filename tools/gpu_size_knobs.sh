# large swarms: three-wave kernel vs two-wave kernel with / without non-temporal accesses, by swarm size
OUT=gpurun_out/ksz; rm -rf $OUT; mkdir -p $OUT
WL=${1:-actuator}
run() { n=$1; tag=$2; shift 2; env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --traffic off --workload $WL --uavs $n --steps 100 --warmup 20 > $OUT/${n}_$tag.json 2> $OUT/${n}_$tag.err || { tail -3 $OUT/${n}_$tag.err; return 1; }
  python - $n $tag <<'PY'
import json,sys
d=json.loads([l for l in open('gpurun_out/ksz/%s_%s.json'%(sys.argv[1],sys.argv[2])) if l.startswith('{')][-1]); r=d['roofline']
print(sys.argv[1].rjust(9), sys.argv[2].ljust(8),'us/step %.1f'%(d['ms_per_step']*1e3),'moved TB/s %.2f'%(r['moved_GBps']/1e3), flush=True)
PY
}
for n in ${SIZES:-200000 300000 500000 1000000 2000000 4000000 8000000}; do
  run $n w3 MRS_THREE_WAVES=1 || exit 1
  run $n w2 MRS_THREE_WAVES=0 MRS_NT_ACCESSES=0 || exit 1
  run $n nt_w2 MRS_THREE_WAVES=0 MRS_NT_ACCESSES=1 || exit 1
done
