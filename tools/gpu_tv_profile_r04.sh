#!/bin/bash
# dev: per-wave stamps of the time-vectorised kernel (block 0 of a B = 128 build, -DTV_PROFILE: whole-call cycles per operator kind):
# round 3's sources (tvp3) against the tree's (tvp)
R=$PWD
mkdir -p gpurun_out
for v in tvp3 tvp; do
  echo "==== $v"
  ARMOUR_HIP_LIB=$R/armour_amd/lib/libarmour_hip_$v.so timeout 300 python3 tools/p1_tv_once.py ${1:-128} 2>&1 | grep "whole calls\|rnea done\|forward done\|build ms" | tail -13
done > gpurun_out/r04_tv_profile_raw.txt 2>&1
cat gpurun_out/r04_tv_profile_raw.txt
