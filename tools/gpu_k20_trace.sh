export TMPDIR=/tmp
OUT=gpurun_out/k20; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic off > $OUT/b.json 2> $OUT/b.err || { tail -5 $OUT/b.err; exit 1; }
python - <<'PY'
import csv,glob,json
import numpy as np
d=json.loads([l for l in open('gpurun_out/k20/b.json') if l.startswith('{')][-1]); print('bench under trace: us/step %.2f regions %d'%(d['ms_per_step']*1e3, d['regions']))
f=sorted(glob.glob('gpurun_out/k20/t/*/*kernel_trace.csv'))[-1]
rows=[r for r in csv.DictReader(open(f)) if r['Kernel_Name'].startswith('mrs_uav_model_step')]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
S=np.array([int(r['Start_Timestamp']) for r in rows]); E=np.array([int(r['End_Timestamp']) for r in rows])
# regions: gaps > 30 us between consecutive starts
cut=np.flatnonzero(np.diff(S)>30000)+1
segs=np.split(np.arange(len(S)),cut)
segs=[s for s in segs if len(s)==40]
print('regions of 40 launches:',len(segs))
span=np.array([E[s].max()-S[s].min() for s in segs])/1e3
print('span first-start..last-end us: median %.1f  min %.1f  p90 %.1f'%(np.median(span),span.min(),np.percentile(span,90)))
s=segs[len(segs)//2]
print('one region, start offsets (us) and durations:')
print(np.round((S[s]-S[s][0])/1e3,1).tolist())
print(np.round((E[s]-S[s])/1e3,1).tolist())
PY
