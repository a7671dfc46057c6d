#!/bin/bash
# lazy join: where a region's time goes (tools/region_overhead.py), K = 20 / 40 / 160 / 1000, MRS_LAZY_JOIN 1 / 0, twice
for rep in 1 2; do for lz in 1 0; do echo "== MRS_LAZY_JOIN=$lz"; MRS_LAZY_JOIN=$lz timeout -k 10 200 python tools/region_overhead.py 20 40 160 1000 2>/dev/null; done; done
