#!/bin/bash
# Builds compile-time variants of the library HERE (hipcc cross-compiles; the GPU box's minutes are for measuring): one shared
# object per variant under variants/, picked up on the box through MRS_SWARM_LIB.  usage: tools/build_variants.sh skin 0.5 0.75 1.0 ...
set -e
CS=mrs_multirotor_simulator_amd/csrc; OBJ=mrs_multirotor_simulator_amd/build
python -m mrs_multirotor_simulator_amd.build > /dev/null
mkdir -p variants
kind=$1; shift
for v in "$@"; do
  tag=$(echo "${kind}_$v" | tr -c 'A-Za-z0-9\n' '_')
  case $kind in
    skin) hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -DMRS_SKIN=$v -c $CS/collide.hip -o /tmp/collide_$tag.o
          hipcc -shared -fPIC --offload-arch=gfx950 -o variants/libmrs_$tag.so $OBJ/step_kernel_literal.o $OBJ/step_kernel_fast.o /tmp/collide_$tag.o $OBJ/outputs.o $OBJ/swarm_host.o ;;
    collideflag) hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math $v -c $CS/collide.hip -o /tmp/collide_$tag.o
          hipcc -shared -fPIC --offload-arch=gfx950 -o variants/libmrs_$tag.so $OBJ/step_kernel_literal.o $OBJ/step_kernel_fast.o /tmp/collide_$tag.o $OBJ/outputs.o $OBJ/swarm_host.o ;;
    stepflag) hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=fast -fno-fast-math $v -c $CS/step_kernel_fast.hip -o /tmp/skf_$tag.o
          hipcc -shared -fPIC --offload-arch=gfx950 -o variants/libmrs_$tag.so $OBJ/step_kernel_literal.o /tmp/skf_$tag.o $OBJ/collide.o $OBJ/outputs.o $OBJ/swarm_host.o ;;
    hostflag) hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math $v -c $CS/swarm_host.hip -o /tmp/host_$tag.o
          hipcc -shared -fPIC --offload-arch=gfx950 -o variants/libmrs_$tag.so $OBJ/step_kernel_literal.o $OBJ/step_kernel_fast.o $OBJ/collide.o $OBJ/outputs.o /tmp/host_$tag.o ;;
  esac
  echo variants/libmrs_$tag.so
done
