#!/bin/bash
# dev: like gpu_tv_ab.sh with an environment assignment applied to every run: tools/gpu_tv_ab_env.sh "VAR=value" <B> <variant> ...
ENVSET=$1; B=$2; shift; shift
for i in 1 2 3; do for v in "$@"; do
  if [ $v = head ]; then unset ARMOUR_HIP_LIB; else export ARMOUR_HIP_LIB=$PWD/armour_amd/lib/libarmour_hip_$v.so; fi
  echo "$ENVSET $v" $(env $ENVSET ARMOUR_P1_TRACE=1 timeout -k 10 120 python tools/p1_once.py $B 2>&1 | grep -o "arena each), [0-9.]* ms" | grep -o "[0-9.]* ms" | tr '\n' ' ')
done; done
