#!/bin/bash
# Builds compile-time variants of the library HERE (hipcc cross-compiles; the GPU box's minutes are for measuring): one shared
# object per variant under variants/, picked up on the box through MRS_SWARM_LIB.  usage: tools/build_variants.sh skin 0.5 0.75 1.0 ...
set -e
CS=mrs_multirotor_simulator_amd/csrc; OBJ=mrs_multirotor_simulator_amd/build
HOST="$OBJ/host_api.o $OBJ/tick_single.o $OBJ/tick_sharded.o $OBJ/transport_rccl.o $OBJ/transport_local.o $OBJ/transport_peer.o"
python -m mrs_multirotor_simulator_amd.build > /dev/null
mkdir -p variants
kind=$1; shift
for v in "$@"; do
  tag=$(echo "${kind}_$v" | tr -c 'A-Za-z0-9\n' '_')
  case $kind in
    skin) hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -DMRS_SKIN=$v -c $CS/collide.hip -o /tmp/collide_$tag.o
          hipcc -shared -fPIC --offload-arch=gfx950 -o variants/libmrs_$tag.so $OBJ/step_kernel_literal.o $OBJ/step_kernel_fast.o /tmp/collide_$tag.o $OBJ/outputs.o $HOST ;;
    collideflag) hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math $v -c $CS/collide.hip -o /tmp/collide_$tag.o
          hipcc -shared -fPIC --offload-arch=gfx950 -o variants/libmrs_$tag.so $OBJ/step_kernel_literal.o $OBJ/step_kernel_fast.o /tmp/collide_$tag.o $OBJ/outputs.o $HOST ;;
    stepflag) hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=fast -fno-fast-math $v -c $CS/step_kernel_fast.hip -o /tmp/skf_$tag.o
          hipcc -shared -fPIC --offload-arch=gfx950 -o variants/libmrs_$tag.so $OBJ/step_kernel_literal.o /tmp/skf_$tag.o $OBJ/collide.o $OBJ/outputs.o $HOST ;;
    hostflag) unit=${v%% *}; flags=${v#* }  # "tick_sharded -DFOO=1": one host unit with extra flags
          hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math $flags -c $CS/$unit.hip -o /tmp/host_$tag.o
          hipcc -shared -fPIC --offload-arch=gfx950 -o variants/libmrs_$tag.so $OBJ/step_kernel_literal.o $OBJ/step_kernel_fast.o $OBJ/collide.o $OBJ/outputs.o $(echo $HOST | sed "s#$OBJ/$unit.o#/tmp/host_$tag.o#") ;;
  esac
  echo variants/libmrs_$tag.so
done
