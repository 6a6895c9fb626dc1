#!/bin/bash
# dev: per-wave stamps of the per-step reach-set kernel at B = 1 (variant "stamps" = -DP1_STAMPS): three-wave against four-wave blocks
export ARMOUR_HIP_LIB=$PWD/armour_amd/lib/libarmour_hip_stamps.so
for v in "3 0" "4 0" "4 1"; do set -- $v
  echo "== ARMOUR_P1_WAVES=$1 ARMOUR_P1_AUX3=$2"
  ARMOUR_P1_WAVES=$1 ARMOUR_P1_AUX3=$2 ARMOUR_P1_TRACE=1 timeout -k 10 120 python tools/p1_once.py 1 2>&1 | grep "t=60\|\[P1\]" | tail -6
done
