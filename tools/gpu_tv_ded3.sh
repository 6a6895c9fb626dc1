#!/bin/bash
export ARMOUR_HIP_LIB=$PWD/armour_amd/lib/libarmour_hip_dproff.so
for ded in 0 1; do echo "== full profile, dedicated=$ded"; ARMOUR_P1_TV_DEDICATED=$ded ARMOUR_P1_TRACE=1 timeout -k 10 120 python tools/p1_once.py 128 2>&1 | grep "tv item 0\|P1 tv" | head -16; done
