#!/bin/bash
# dev only: rebuild the library with role/phase ablations on the GPU box and time each variant with the headline bench
# bits: 1 skip collision, 2 skip torque, 4 skip limits, 8 no plane loads, 16 no slicing, 32 no collision output writes
for ab in ${ABLATIONS:-0 2 1 3 6 14 38}; do
  make -C armour_amd/csrc -B EXTRA="-DP2_ABLATE=$ab" >/dev/null 2>&1
  echo -n "ablate=$ab: "
  timeout 200 python bench.py --no-cpu-baseline --headline-only --no-check "$@" 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(d['roofline']['launch_us'])
except Exception as e: print('failed', e)"
done
make -C armour_amd/csrc -B >/dev/null 2>&1
