# two-stream split threshold: swarms below 1024 blocks, split forced on (MRS_SPLIT_MIN_BLOCKS=1) vs off
OUT=gpurun_out/split; rm -rf $OUT; mkdir -p $OUT
for wl in actuator position; do for n in 16000 24000 32000 40000 50000 60000 65000; do for sp in 1 100000; do
  MRS_SPLIT_MIN_BLOCKS=$sp timeout -k 10 200 python bench.py --no-cpu-baseline --traffic off --workload $wl --uavs $n --steps 1000 --warmup 100 > $OUT/x.json 2> $OUT/x.err || { tail -3 $OUT/x.err; exit 1; }
  python - $wl $n $sp <<'PY'
import json,sys
d=json.loads([l for l in open('gpurun_out/split/x.json') if l.startswith('{')][-1])
print(sys.argv[1].ljust(9), sys.argv[2].rjust(7), 'split' if sys.argv[3]=='1' else 'one  ', 'us/step %.2f'%(d['ms_per_step']*1e3), flush=True)
PY
done; done; done
