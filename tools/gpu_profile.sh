#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (writes under gpurun_out/prof_*).
# Usage: bash tools/gpu_profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $OLDPWD/bench.py --no-cpu-baseline --headline-only "$@" > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $OLDPWD/bench.py --no-cpu-baseline --headline-only --steps 200 --warmup 20 "$@" > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 $OLDPWD/bench.py --no-cpu-baseline --headline-only --steps 200 --warmup 20 "$@" > $OUT/bench_pmc_write.log 2>&1
cd $OLDPWD
find $OUT -name "*.csv" | head -20
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; head -8 $f; done
python3 - <<PY
import csv, glob, collections
for tag in ("pmc_fetch", "pmc_write"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            k = (row["Kernel_Name"][:60], row["Counter_Name"])
            acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
        for k, (s, n) in sorted(acc.items()):
            print(tag, k, "mean per dispatch", s / n, "dispatches", n)
PY
tail -1 $OUT/bench_trace.log | cut -c1-400
