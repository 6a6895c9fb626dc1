#!/bin/bash
# dev: large batches (more groups than CUs): one wave per group (default) against four-wave blocks looping over the groups (ARMOUR_P1_TV_WAVES=4)
for B in 160 200 256 384 512; do
  for w in 0 4; do
    echo "B=$B ARMOUR_P1_TV_WAVES=$w" $(ARMOUR_P1_TV_WAVES=$w ARMOUR_P1_TRACE=1 timeout -k 10 200 python tools/p1_once.py $B 2>&1 | grep -o "blocks of [0-9]* wave.*arena each), [0-9.]* ms, flags 0x[0-9a-f]*" | sed 's/(.*arena each),//' | tail -1)
  done
done
