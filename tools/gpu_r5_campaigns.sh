#!/bin/bash
# round 5: the campaigns outside the suite on the final tree (tests/campaigns/; the CPU oracle is the checker)
mkdir -p gpurun_out; OUT=gpurun_out/r05_campaigns.log
{ echo "# campaigns of round 5 on the final tree (tests/campaigns/; the CPU oracle is the checker)"
  echo "## soak.py 20000 2000 fast 30 local   and   soak.py 16000 1500 literal 16 sharded"; } > $OUT
timeout -k 10 500 python tests/campaigns/soak.py 20000 2000 fast 30 local 2>&1 | grep -v amdgpu.ids >> $OUT
timeout -k 10 500 python tests/campaigns/soak.py 16000 1500 literal 16 sharded 2>&1 | grep -v amdgpu.ids >> $OUT
echo "## chaos_seeds.py 120 5000 (LITERAL)   and   chaos_seeds.py 40 7000 600 1200 fast" >> $OUT
timeout -k 10 500 python tests/campaigns/chaos_seeds.py 120 5000 2>&1 | grep -v amdgpu.ids | tail -4 >> $OUT
timeout -k 10 500 python tests/campaigns/chaos_seeds.py 40 7000 600 1200 fast 2>&1 | grep -v amdgpu.ids | tail -3 >> $OUT
tail -12 $OUT
