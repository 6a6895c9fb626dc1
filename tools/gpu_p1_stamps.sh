#!/bin/bash
# dev: per-wave stamps of the per-step reach-set kernel at B = 1 (variant "stamps" = -DP1_STAMPS)
export ARMOUR_HIP_LIB=$PWD/armour_amd/lib/libarmour_hip_stamps.so
ARMOUR_P1_TRACE=1 timeout -k 10 120 python tools/p1_once.py 1 2>&1 | grep "t=60\|\[P1\]" | head -24
