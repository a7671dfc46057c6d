#!/bin/bash
# A/B of compile-time variants of the FAST step kernels: builds a library with the given extra -D flags in /tmp and runs bench.py on it.
# usage (on the GPU box): tools/variant_bench.sh "<flags>" [bench args...]
set -e
FLAGS="$1"; shift
CS=mrs_multirotor_simulator_amd/csrc; OBJ=mrs_multirotor_simulator_amd/build
python -m mrs_multirotor_simulator_amd.build > /dev/null
TAG=$(echo "$FLAGS" | tr -c 'A-Za-z0-9' '_')
hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=fast -fno-fast-math $FLAGS -c $CS/step_kernel_fast.hip -o /tmp/skf_$TAG.o
hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libmrs_$TAG.so $OBJ/step_kernel_literal.o /tmp/skf_$TAG.o $OBJ/collide.o $OBJ/outputs.o $OBJ/swarm_host.o
MRS_SWARM_LIB=/tmp/libmrs_$TAG.so python bench.py --no-cpu-baseline "$@" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$FLAGS', '|', ' '.join(sys.argv[1:]), '|', '%.4g UAV-steps/s' % d['value'], '%.2f us' % (d['ms_per_step']*1e3), 'frac %.3f' % d['roofline']['frac'])" "$@"
