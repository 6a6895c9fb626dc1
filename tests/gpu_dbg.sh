#!/bin/bash
make -C armour_amd/csrc -B EXTRA="-DDBG_CHECK_MERGE" 2>&1 | grep -E "error" 
python tests/gpu_p1_quick.py 2>&1 | grep -E "torque radius|Error|error" | head -4
