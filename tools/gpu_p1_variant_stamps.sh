#!/bin/bash
# dev: per-wave stamps of the per-step kernel at B = 1 for variant libraries (each built with -DP1_STAMPS): tools/gpu_p1_variant_stamps.sh <name>...
for v in "$@"; do
  echo "== variant $v"
  ARMOUR_HIP_LIB=$PWD/armour_amd/lib/libarmour_hip_$v.so ARMOUR_P1_TRACE=1 timeout -k 10 120 python tools/p1_once.py 1 2>&1 | grep "t=60\|\[P1\]" | tail -5
done
