#!/bin/bash
# dev: armour_amd/lib/libarmour_hip_<name>.so = the in-tree objects with p2_eval.o rebuilt with extra flags (e.g. -DP2_TIMELINE -DP2_ABLATE=6)
# usage: tools/build_p2_variant.sh <name> [extra hipcc flags]
set -e
name=$1; shift
cd /root/repo/armour_amd/csrc
L=/root/repo/armour_amd/lib
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function "$@" -c p2_eval.hip -o /tmp/p2_eval_$name.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -Wl,-rpath,/opt/rocm/lib -o $L/libarmour_hip_$name.so $L/api.o /tmp/p2_eval_$name.o $L/p1_reach.o $L/solver.o $L/solver_device.o $L/controller.o $L/batch.o -lpthread
echo built $L/libarmour_hip_$name.so
