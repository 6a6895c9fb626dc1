#!/bin/bash
# dev: profile stamps of the eight-wave blocks (variant dprof) and tuning runs of the in-tree library
export ARMOUR_HIP_LIB=$PWD/armour_amd/lib/libarmour_hip_dprof.so
echo "== profile, shift 0"; ARMOUR_P1_TRACE=1 timeout -k 10 120 python tools/p1_once.py 128 2>&1 | grep "tv item 0\|P1 tv" | tail -14
unset ARMOUR_HIP_LIB
for sh in 0 1 2 3; do echo "shift=$sh" $(ARMOUR_P1_TV_HELP_SHIFT=$sh ARMOUR_P1_TRACE=1 timeout -k 10 120 python tools/p1_once.py 128 2>&1 | grep -o "arena each), [0-9.]* ms" | grep -o "[0-9.]* ms" | tr '\n' ' '); done
for mn in 64 96 128 256; do echo "help_min=$mn" $(ARMOUR_P1_TV_HELP_MIN=$mn ARMOUR_P1_TRACE=1 timeout -k 10 120 python tools/p1_once.py 128 2>&1 | grep -o "arena each), [0-9.]* ms" | grep -o "[0-9.]* ms" | tr '\n' ' '); done
for hn in 12 14 18 20; do echo "keep=$hn/32" $(ARMOUR_P1_TV_HELPERS=$hn ARMOUR_P1_TRACE=1 timeout -k 10 120 python tools/p1_once.py 128 2>&1 | grep -o "arena each), [0-9.]* ms" | grep -o "[0-9.]* ms" | tr '\n' ' '); done
