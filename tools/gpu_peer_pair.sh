# the peer-window exchange kernel under a kernel trace: two ranks of 125 000 in one process (tools/peer_rank_pair.py)
OUT=$PWD/gpurun_out/peerpair; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 python tools/peer_rank_pair.py 125000 400 peer > $OUT/log.txt 2>$OUT/err.txt || { tail -20 $OUT/err.txt; exit 1; }
timeout -k 10 200 python tools/peer_rank_pair.py 125000 400 loopback >> $OUT/log.txt 2>$OUT/err.txt || { tail -20 $OUT/err.txt; exit 1; }
cat $OUT/log.txt
ROOT=$PWD; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $ROOT/tools/peer_rank_pair.py 125000 400 peer > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
cd $ROOT
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); head -14 "$f" | cut -c1-220
t=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv, sys, statistics
d = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(sys.argv[1])) if "k_peer_allgather" in r["Kernel_Name"]]
d.sort()
print(f"k_peer_allgather: {len(d)} launches, min {d[0]/1e3:.2f} us, median {statistics.median(d)/1e3:.2f} us, p90 {d[int(0.9*len(d))]/1e3:.2f} us, max {d[-1]/1e3:.1f} us")
PY
