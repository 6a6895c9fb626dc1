#!/bin/bash
# dev: eight-wave blocks (variant "ded" = -DP1_TV_WAVES_PER_SIMD=2) against its own four-wave blocks and the shipped library at small batch sizes (no contention between blocks)
for B in 8 16 32 64; do
  export ARMOUR_HIP_LIB=$PWD/armour_amd/lib/libarmour_hip_ded.so
  for ded in 1 0; do echo "B=$B variant ded dedicated=$ded" $(ARMOUR_P1_TV=1 ARMOUR_P1_TV_DEDICATED=$ded ARMOUR_P1_TRACE=1 timeout -k 10 120 python tools/p1_once.py $B 2>&1 | grep -o "arena each), [0-9.]* ms" | grep -o "[0-9.]* ms" | tr '\n' ' '); done
  unset ARMOUR_HIP_LIB
  echo "B=$B shipped" $(ARMOUR_P1_TV=1 ARMOUR_P1_TRACE=1 timeout -k 10 120 python tools/p1_once.py $B 2>&1 | grep -o "arena each), [0-9.]* ms" | grep -o "[0-9.]* ms" | tr '\n' ' ')
done
