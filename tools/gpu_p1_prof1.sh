#!/bin/bash
# dev: -DP1_PROFILE build of the per-step kernel at B = 1: cycles by call size / sorter / phase for the profiled time steps
export ARMOUR_HIP_LIB=$PWD/armour_amd/lib/libarmour_hip_p1prof.so
ARMOUR_P1_TRACE=1 timeout -k 10 120 python tools/p1_once.py 1 2>&1 | grep "t=60\|t=70\|P1 profile\|P1 phases\|\[P1\]" | tail -16
