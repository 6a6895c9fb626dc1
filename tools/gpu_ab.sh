#!/bin/bash
# A/B the headline bench between ab_head/ (a reference build) and the in-tree library, interleaved.
for i in 1 2 3; do for v in head new; do
  if [ $v = head ]; then export ARMOUR_HIP_LIB=$PWD/ab_head/armour_amd/lib/libarmour_hip.so; else unset ARMOUR_HIP_LIB; fi
  timeout 200 python bench.py --no-cpu-baseline --headline-only "$@" 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['roofline']['launch_us'])"
done; done
