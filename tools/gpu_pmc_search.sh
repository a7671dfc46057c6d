#!/bin/bash
# PMC counters of the two search kernels (one pass per counter set; no trace modes alongside): bash tools/gpu_pmc_search.sh <n_uavs>
N=${1:-100000}
export TMPDIR=/tmp
OUT=gpurun_out/pmc_search
rm -rf $OUT; mkdir -p $OUT
for set in "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d" " -f1)
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d $OUT/$tag -- python tools/search_rate.py $N > $OUT/$tag.log 2>&1
  f=$(ls $OUT/$tag/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    k = "k_query2" if "k_query2" in k else "k_pack_insert" if "k_pack_insert" in k else None
    if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
done
