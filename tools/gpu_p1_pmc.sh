#!/bin/bash
# dev: SQ counters of the P1 chain kernel. Usage: bash tools/gpu_p1_pmc.sh <B>
set -u
B=${1:-1}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/p1pmc_B$B
mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $OUT/sq -- python3 $OLDPWD/tools/p1_once.py $B > $OUT/log_sq.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/sq2 -- python3 $OLDPWD/tools/p1_once.py $B > $OUT/log_sq2.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $OLDPWD/tools/p1_once.py $B > $OUT/log_trace.txt 2>&1
cd $OLDPWD
tail -2 $OUT/log_sq.txt
python3 - <<PY
import csv, glob, collections
for tag in ("sq", "sq2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            if "chain" not in row["Kernel_Name"]: continue
            k = row["Counter_Name"]
            acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
        for k, (s, n) in sorted(acc.items()):
            print(tag, k, "mean per dispatch %.4g" % (s / n), "dispatches", n)
PY
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do head -6 $f | cut -c1-200; done
