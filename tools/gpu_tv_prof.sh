#!/bin/bash
# dev: light profile build of the time-vectorised kernel (variant tprof = -DTV_PROFILE): per-wave stamps, waits and whole-call operator times of one B = 128 item
export ARMOUR_HIP_LIB=$PWD/armour_amd/lib/libarmour_hip_tprof.so
ARMOUR_P1_TRACE=1 timeout -k 10 120 python tools/p1_once.py 128 2>&1 | grep "tv item 0\|P1 tv" | tail -22
