#!/bin/bash
# dev: SQ / instruction-cache counters of the time-vectorised P1 kernel. Usage: bash tools/gpu_tv_pmc.sh <B>
set -u
B=${1:-8}
export TMPDIR=/tmp
export ARMOUR_P1_TV=1
OUT=$PWD/gpurun_out/tvpmc_B$B
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $OUT/sq -- python3 $OLDPWD/tools/p1_tv_once.py $B > $OUT/log_sq.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/sq2 -- python3 $OLDPWD/tools/p1_tv_once.py $B > $OUT/log_sq2.txt 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $OUT/ic -- python3 $OLDPWD/tools/p1_tv_once.py $B > $OUT/log_ic.txt 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/if -- python3 $OLDPWD/tools/p1_tv_once.py $B > $OUT/log_if.txt 2>&1
cd $OLDPWD
tail -2 $OUT/log_sq.txt
python3 - <<PY
import csv, glob, collections
for tag in ("sq", "sq2", "ic", "if"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            if "tv_kernel" not in row["Kernel_Name"]: continue
            k = row["Counter_Name"]
            acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
        for k, (s, n) in sorted(acc.items()):
            print(tag, k, "mean per dispatch %.5g" % (s / n), "dispatches", n)
PY
