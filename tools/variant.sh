#!/bin/bash
# armour_amd/lib/libarmour_hip_<name>.so = the current tree with BOTH reach-set units rebuilt with extra flags (the other objects are the in-tree
# ones: run `make -C armour_amd/csrc` first).  Load it with ARMOUR_HIP_LIB=... or name it to tools/ab.py / tools/traffic.sh.
#   tools/variant.sh <name> [-s <source tree copy>] [hipcc flags...]      e.g.  tools/variant.sh occ2 -DP1_WAVES_PER_SIMD=2
# -s: compile the units of an edited COPY of the tree (cp -r armour_amd include /tmp/x/) instead of the tree itself.
# The narrow time-vectorised unit takes -DTV_GROW=<w> / the main unit -DP1_TV_NARROW=<w> when a flag -DROWS=<w> is given.
set -e
name=$1; shift
src=/root/repo
if [ "${1:-}" = "-s" ]; then src=$2; shift 2; fi
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function"   # (the shipped flags of both reach-set units: armour_amd/csrc/Makefile HIPFLAGS + P1FLAGS; pass -mllvm -enable-ipra=false for the fenced form of rounds 1-5)
A=(); B=()
for f in "$@"; do case "$f" in -DROWS=*) A+=("-DP1_TV_NARROW=${f#-DROWS=}"); B+=("-DTV_GROW=${f#-DROWS=}");; *) A+=("$f"); B+=("$f");; esac; done
O=/tmp/variant_$name; mkdir -p $O
cd $src/armour_amd/csrc
/opt/rocm/bin/hipcc $F "${A[@]}" -c p1_reach.hip -o $O/p1_reach.o 2>$O/p1_reach.err &
/opt/rocm/bin/hipcc $F "${B[@]}" -c p1_reach_tv50.hip -o $O/p1_reach_tv50.o 2>$O/p1_reach_tv50.err &
wait
L=/root/repo/armour_amd/lib
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -Wl,-rpath,/opt/rocm/lib -o $L/libarmour_hip_$name.so $L/api.o $L/p2_eval.o $O/p1_reach.o $O/p1_reach_tv50.o $L/solver.o $L/solver_device.o $L/controller.o $L/batch.o $L/relevance.o -lpthread
python3 /root/repo/tools/check_long_branches.py $L/libarmour_hip_$name.so
echo built $L/libarmour_hip_$name.so
