#!/bin/bash
# dev: where do the time-vectorised kernel's waves wait?  PMC passes (one counter group each) over tools/p1_once.py 128
export TMPDIR=/tmp
R=$PWD; OUT=$R/gpurun_out/tvpmc2; rm -rf $OUT; mkdir -p $OUT; cd /tmp
i=0
for grp in "TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS" \
           "TCP_UTCL1_STALL_INFLIGHT_MAX TCP_UTCL1_THRASHING_STALL TCP_UTCL1_STALL_MULTI_MISS TCP_UTCL1_SERIALIZATION_STALL" \
           "TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCP_LATENCY TCP_TOTAL_ACCESSES" \
           "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_GATE_EN1" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" \
           "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_BRANCH SQ_IFETCH" \
           "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TD_TD_BUSY TD_TC_STALL" ; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 $R/tools/p1_once.py 128 > $OUT/g$i.log 2>&1; echo "group $i rc=$?"
done
cd $R
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('gpurun_out/tvpmc2/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'tv_kernel' in row['Kernel_Name']:
            a = acc[row['Counter_Name']]; a[0] += float(row['Counter_Value']); a[1] += 1
for k in sorted(acc): print(f"{k:45s} {acc[k][0] / acc[k][1]:16.1f}  ({acc[k][1]} dispatches)")
PY
find $OUT -name "*.csv" -size +1M -delete
