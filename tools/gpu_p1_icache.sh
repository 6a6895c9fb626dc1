#!/bin/bash
# dev: instruction-cache counters of the P1 chain kernel. Usage: bash tools/gpu_p1_icache.sh <B>
B=${1:-1}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/p1ic_B$B
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQ_INSTS_WAVE32_LDS" | sort -u | head -20
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $OUT/ic -- python3 $OLDPWD/tools/p1_once.py $B > $OUT/log.txt 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/if -- python3 $OLDPWD/tools/p1_once.py $B > $OUT/log2.txt 2>&1
cd $OLDPWD
tail -3 $OUT/log.txt
python3 - <<PY
import csv, glob, collections
for tag in ("ic", "if"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            if "chain" not in row["Kernel_Name"]: continue
            acc[row["Counter_Name"]][0] += float(row["Counter_Value"]); acc[row["Counter_Name"]][1] += 1
        for k, (s, n) in sorted(acc.items()):
            print(tag, k, "mean per dispatch %.4g" % (s / n))
PY
