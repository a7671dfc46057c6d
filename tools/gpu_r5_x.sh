#!/bin/bash
# stand-in collective with its bytes charged (MRS_STANDIN_GBPS): one rank of 8 x 125000, 10 / 20 us
for g in 0 1000 300 150; do for lat in 10 20; do MRS_STANDIN_GBPS=$g timeout -k 10 200 python tools/sharded_interior_alone.py $lat 600 2>/dev/null | sed "s/^/gbps $g /" | cut -c1-120; done; done
