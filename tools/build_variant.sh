#!/bin/bash
# dev: build armour_amd/lib/libarmour_hip_<name>.so from a COPY of the source tree (made by the caller under /tmp/var_<name>/, with whatever
# edits the variant needs) -- only p1_reach.o is rebuilt, the other objects are the in-tree ones.  Load it with ARMOUR_HIP_LIB=...
# usage: tools/build_variant.sh <name> [extra hipcc flags]
set -e
name=$1; shift
src=/tmp/var_$name
[ -d $src/armour_amd/csrc ] || { echo "no $src/armour_amd/csrc"; exit 1; }
cd $src/armour_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -mllvm -enable-ipra=false "$@" -c p1_reach.hip -o $src/p1_reach.o
L=/root/repo/armour_amd/lib
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -Wl,-rpath,/opt/rocm/lib -o $L/libarmour_hip_$name.so $L/api.o $L/p2_eval.o $src/p1_reach.o $L/solver.o $L/solver_device.o $L/controller.o $L/batch.o $L/relevance.o -lpthread
echo built $L/libarmour_hip_$name.so
