# the HBM-streaming regime (4 M UAVs) under the launcher's tuning switches
OUT=gpurun_out/k4m; rm -rf $OUT; mkdir -p $OUT
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --traffic off --uavs 4000000 --steps 100 --warmup 20 > $OUT/$tag.json 2> $OUT/$tag.err || { tail -3 $OUT/$tag.err; return 1; }
  python - $tag <<'PY'
import json,sys
d=json.loads([l for l in open('gpurun_out/k4m/%s.json'%sys.argv[1]) if l.startswith('{')][-1]); r=d['roofline']
print(sys.argv[1].ljust(28),'us/step %.1f'%(d['ms_per_step']*1e3),'moved TB/s %.2f'%(r['moved_GBps']/1e3),'frac_of_achievable %.3f'%r['frac_of_achievable'], r['kernel'])
PY
}
run base A=1 && run nosplit MRS_SPLIT_STREAMS=0 && run w2 MRS_THREE_WAVES=0 && run w2_nosplit MRS_THREE_WAVES=0 MRS_SPLIT_STREAMS=0 && run nt MRS_NT_ACCESSES=1 && run nt_w2 MRS_NT_ACCESSES=1 MRS_THREE_WAVES=0 && run nobuf MRS_NO_BUFFER_ADDRESSING=1
