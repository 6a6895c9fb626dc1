#!/bin/bash
# dev: where the waves of the reach-set kernels wait -- in-flight levels of the memory / LDS / instruction-fetch queues (LEVEL / count =
# average latency in cycles), per kernel.  Usage: bash tools/gpu_p1_latency_pmc.sh <B>
B=${1:-128}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/p1lat_B$B
rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VALU" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $set --output-format csv -d $OUT/s$i -- python3 $R/tools/p1_once.py $B > $OUT/s$i.log 2>&1
done
cd $R
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        kn = "tv" if "tv_kernel" in row["Kernel_Name"] else "chain" if "chain_kernel" in row["Kernel_Name"] else None
        if kn: a = acc[(kn, row["Counter_Name"])]; a[0] += float(row["Counter_Value"]); a[1] += 1
for (kn, c), (s, n) in sorted(acc.items()): print(kn, c, "%.6g" % (s / n))
PY
