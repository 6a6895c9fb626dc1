#!/bin/bash
# Round-4 evidence for the headline launch (configs[1], B = 1): phase stamps of the shipped one-point kernel (development build
# libarmour_hip_tl.so = -DP2_TIMELINE -DP2_ABLATE=6: collision blocks only, stamps left in the limit rows), the launch floor of
# tools/micro/launch_floor.hip, and the kernel's argument bytes.  Writes gpurun_out/r04_p2_timeline.txt.
set -u
R=$PWD
OUT=$R/gpurun_out/r04_p2_timeline.txt
{
echo "Round 4 -- where the headline launch's microseconds go (one MI355X; $(date -u +%F))"
echo
echo "== phase stamps (s_memrealtime, 10 ns ticks) of three collision blocks of armour_p2_eval_kernel<true,true,false,false,true,6,true>, 64 launches back to back on one stream"
ARMOUR_HIP_LIB=$R/armour_amd/lib/libarmour_hip_tl.so python3 tools/gpu_p2_timeline.py
echo
echo "== launch floor (tools/micro/launch_floor.hip): back-to-back period of an almost empty kernel"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/micro/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
echo
echo "== kernel argument bytes: sizeof(P2Tables) + 3 pointers + sizeof(P2Launch)"
python3 - <<'PY'
import re, subprocess
src = r'''
#include "armour_amd/csrc/p2_tiles.h"
#include <cstdio>
int main() { printf("P2Tables %zu B, P2Launch %zu B, kernarg %zu B\n", sizeof(P2Tables), sizeof(p2::P2Launch), sizeof(P2Tables) + 24 + sizeof(p2::P2Launch)); }
'''
open('/tmp/ka.hip', 'w').write(src)
subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-std=c++17', '-I.', '/tmp/ka.hip', '-o', '/tmp/ka'], check=True)
print(subprocess.run(['/tmp/ka'], capture_output=True, text=True).stdout.strip())
PY
} > $OUT 2>&1
tail -40 $OUT
