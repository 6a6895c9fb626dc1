#!/bin/bash
# dev: A/B the B=128 time-vectorised reach-set build between library variants (tools/build_variant.sh), interleaved on one box.
# usage: tools/gpu_tv_ab.sh <B> <variant> [<variant> ...]   ("head" = the in-tree library)
B=$1; shift
for i in 1 2 3; do for v in "$@"; do
  if [ $v = head ]; then unset ARMOUR_HIP_LIB; else export ARMOUR_HIP_LIB=$PWD/armour_amd/lib/libarmour_hip_$v.so; fi
  echo $v $(ARMOUR_P1_TRACE=1 timeout -k 10 120 python tools/p1_once.py $B 2>&1 | grep -o "arena each), [0-9.]* ms" | grep -o "[0-9.]* ms" | tr '\n' ' ')
done; done
