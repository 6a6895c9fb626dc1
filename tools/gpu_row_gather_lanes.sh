#!/bin/bash
# dev: does a row of which only the first 8*act bytes are read cost less than a full 512-byte row?  (tools/micro/row_gather.hip, last argument)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value tools/micro/row_gather.hip -o /tmp/row_gather || exit 1
for R in 20000 200000; do for act in 64 56 50 48 32; do /tmp/row_gather 4 16 $R 0 400 $act; done; done
