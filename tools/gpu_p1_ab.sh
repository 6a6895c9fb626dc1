#!/bin/bash
# dev: A/B the B=128 reach-set build between ab_head/ (a build of HEAD: git archive HEAD armour_amd include | tar -x -C ab_head; make) and the
# in-tree library, interleaved on one box.
for i in 1 2 3; do for v in head new; do
  if [ $v = head ]; then export ARMOUR_HIP_LIB=$PWD/ab_head/armour_amd/lib/libarmour_hip.so; else unset ARMOUR_HIP_LIB; fi
  echo $v $(ARMOUR_P1_TRACE=1 timeout -k 10 120 python tools/p1_tv_once.py ${1:-128} 2>&1 | grep -o "arena each), [0-9.]* ms" | grep -o "[0-9.]* ms" | tr '\n' ' ')
done; done
