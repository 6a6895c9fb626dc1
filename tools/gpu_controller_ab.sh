#!/bin/bash
# dev: the tracking controller's two kernels (ARMOUR_CTL_SPLIT=0 one lane per state, =1 three waves per 64 states) over batch sizes
for s in 0 1; do echo "== ARMOUR_CTL_SPLIT=$s"; ARMOUR_CTL_SPLIT=$s timeout -k 10 300 python tools/gpu_controller_probe.py 1 8 64 1000 4096 16384 65536 262144 1000000 || exit 1; done
