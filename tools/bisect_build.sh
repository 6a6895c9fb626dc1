#!/bin/bash
# dev: tools/bisect_build.sh <commit> -- export that commit's tree to ab_head/<commit>/ and build its library with two waves per SIMD for the reach-set kernels
c=$1
d=/root/repo/ab_head/$c
rm -rf $d; mkdir -p $d
git -C /root/repo archive $c armour_amd include tests/helpers.py | tar -x -C $d
make -C $d/armour_amd/csrc EXTRA=-DP1_WAVES_PER_SIMD=2 P1_OCC_ALLOW=2 > $d/build.log 2>&1
ls -la $d/armour_amd/lib/libarmour_hip.so 2>&1 | tail -1
grep -A8 "Function Name: .*chain_kernelILi1" $d/armour_amd/lib/p1_reach.resources.txt 2>/dev/null | grep -i "VGPRs:\|AGPRs\|Occupancy" | sed 's/.*remark: //' | tr '\n' ' '; echo
rm -f $d/armour_amd/lib/*.o
