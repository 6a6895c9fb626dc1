#!/bin/bash
# dev: kernel-trace average of the headline launch (configs[1]) at HEAD and in a copy of an older commit's tree (ab_head/<name>, built there), interleaved on ONE box.
# the same instantiation averaged 5.30-5.33 us in the traces of rounds 2-4 and 5.82 us in round 5's.   bash tools/dev/headline_ab.sh r04
set -u
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/headline_ab; rm -rf $O; mkdir -p $O
HB="--no-cpu-baseline --headline-only --no-sync-probe --repeats 3 --steps 2000 --warmup 200"
for rnd in 1 2 3; do
  for v in head ${1:-r04}; do
    T=$R; [ $v != head ] && T=$R/ab_head/$v
    cd $T
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${v}_$rnd -- python3 $T/bench.py $HB > $O/${v}_$rnd.log 2>&1
    echo "$v round $rnd rc=$?"
    find $O/${v}_$rnd -name "*kernel_trace.csv" -delete
  done
done
cd $R
python3 - <<'PY'
import csv, glob
for f in sorted(glob.glob("gpurun_out/headline_ab/*/**/*kernel_stats.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "armour_p2_eval_kernel" in r["Name"]:
            print(f.split("/")[2], "calls", r["Calls"], "average %.1f ns" % float(r["AverageNs"]), "min", r["MinNs"], "max", r["MaxNs"], "stddev", r.get("StdDev", "")[:8])
PY
