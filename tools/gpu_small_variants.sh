# small swarms (one wave per SIMD): the step kernels under other instruction-scheduling strategies (variants/ built by tools/build_variants.sh stepflag ...)
OUT=gpurun_out/smallvar; rm -rf $OUT; mkdir -p $OUT
run() { # tag lib
  for n in 400 16000; do for sub in 1 64; do
    MRS_SWARM_LIB=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --traffic off --uavs $n --workload position --steps 2048 --warmup 128 --substeps $sub > $OUT/$1_${n}_$sub.json 2> $OUT/$1_${n}_$sub.err || { tail -3 $OUT/$1_${n}_$sub.err; return 1; }
    python - $1 $n $sub <<'PY'
import json,sys
d=json.loads([l for l in open('gpurun_out/smallvar/%s_%s_%s.json'%tuple(sys.argv[1:4])) if l.startswith('{')][-1])
print(sys.argv[1][:60].ljust(60),'n',sys.argv[2],'substeps',sys.argv[3],'us/step %.2f'%(d['ms_per_step']*1e3))
PY
  done; done
}
run base "" || exit 1
for v in variants/libmrs_stepflag_*.so; do run $(basename $v .so | sed 's/libmrs_stepflag_//') $PWD/$v || exit 1; done
