#!/bin/bash
# dev: per-wave stamps of the time-vectorised kernel (block 0 of a B = 128 build): -DTV_PROFILE (whole-call times) and -DTV_PROFILE_FULL (walk split)
R=$PWD
mkdir -p gpurun_out
for v in tvp tvpf; do
  echo "==== $v"
  ARMOUR_HIP_LIB=$R/armour_amd/lib/libarmour_hip_$v.so timeout 300 python3 tools/p1_tv_once.py 128 2>&1 | grep -v "^\[tv item [1-9]" | tail -60
done > gpurun_out/r04_tv_profile_raw.txt 2>&1
tail -130 gpurun_out/r04_tv_profile_raw.txt
