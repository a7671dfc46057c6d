#!/bin/bash
# round 5: more seeds of the chaos campaign on the final library (halo searches on by default), with the halo statistics of every scenario
mkdir -p gpurun_out; OUT=gpurun_out/r05_chaos2.log
{ echo "# chaos_seeds.py 300 9000 (LITERAL) and chaos_seeds.py 80 11000 600 1200 fast on the final library of round 5 (halo searches on)"; } > $OUT
MRS_HALO_TRACE=1 timeout -k 10 900 python tests/campaigns/chaos_seeds.py 300 9000 > gpurun_out/r05_chaos2_a.txt 2>&1; echo "rc=$?" >> $OUT
grep -v "amdgpu.ids\|mrs halo" gpurun_out/r05_chaos2_a.txt | tail -4 >> $OUT
echo "halo searches traced: $(grep -c 'mrs halo' gpurun_out/r05_chaos2_a.txt); repeated on all records: $(grep -c 'repeated' gpurun_out/r05_chaos2_a.txt); with the MOVED flag: $(grep -c 'flags [13] ' gpurun_out/r05_chaos2_a.txt)" >> $OUT
MRS_HALO_TRACE=1 timeout -k 10 600 python tests/campaigns/chaos_seeds.py 80 11000 600 1200 fast > gpurun_out/r05_chaos2_b.txt 2>&1; echo "rc=$?" >> $OUT
grep -v "amdgpu.ids\|mrs halo" gpurun_out/r05_chaos2_b.txt | tail -3 >> $OUT
echo "halo searches traced: $(grep -c 'mrs halo' gpurun_out/r05_chaos2_b.txt); repeated on all records: $(grep -c 'repeated' gpurun_out/r05_chaos2_b.txt); with the MOVED flag: $(grep -c 'flags [13] ' gpurun_out/r05_chaos2_b.txt)" >> $OUT
cat $OUT
