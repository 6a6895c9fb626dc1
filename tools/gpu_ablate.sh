#!/bin/bash
# dev only: rebuild the library with role/phase ablations on the GPU box and time each variant
# bits: 1 skip collision, 2 skip torque, 4 skip limits, 8 no plane loads, 16 no slicing, 32 no collision output writes
for ab in ${ABLATIONS:-6 14 22 38 62}; do
  make -C armour_amd/csrc -B EXTRA="-DP2_ABLATE=$ab" >/dev/null 2>&1
  echo -n "ablate=$ab: "
  timeout 200 python tools/gpu_launch_probe.py 2>&1 | grep -E "graph replay" | tail -1
done
