export TMPDIR=/tmp
OUT=gpurun_out/ranktrace; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/t -- python tools/sharded_rank_cost.py 125000 8 400 export > $OUT/log.txt 2>&1 || { tail -20 $OUT/log.txt; exit 1; }
tail -1 $OUT/log.txt | cut -c1-200
python - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/ranktrace/t/*/*kernel_stats.csv'))[-1]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), '%9.0f'%float(r['AverageNs']), '%6.2f%%'%float(r['Percentage']))
PY
