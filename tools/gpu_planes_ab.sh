#!/bin/bash
# dev: planes-kernel time of the in-tree library against another build (ARMOUR_HIP_LIB), under rocprofv3 kernel stats
export TMPDIR=/tmp
for v in new other; do
  if [ $v = other ]; then export ARMOUR_HIP_LIB=$1; else unset ARMOUR_HIP_LIB; fi
  rm -rf /tmp/pl_$v; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl_$v -- python3 tools/p1_once.py 128 > /tmp/pl_$v.log 2>&1
  echo $v; python3 - $(find /tmp/pl_$v -name "*kernel_stats.csv") <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "planes_kernel" in r["Name"] or "tv_kernel" in r["Name"]: print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
done
