#!/bin/bash
# dev: tools/micro/row_gather over waves per CU, terms in flight, table size and work per term
cd tools/micro
for R in 2000 20000 200000; do for work in 0 100; do for U in 4 16; do for W in 1 2 4 8; do timeout -k 5 60 ./row_gather $W $U $R $work 300 || exit 1; done; done; done; done
