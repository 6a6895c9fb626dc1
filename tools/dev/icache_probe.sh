set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/s9_icache; rm -rf $O; mkdir -p $O
cd /tmp
for v in r5 eu4; do
  export ARMOUR_HIP_LIB=$R/armour_amd/lib/libarmour_hip_$v.so
  timeout -k 10 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH --output-format csv -d $O/$v -- python3 $R/tools/workload.py p1 1 > $O/$v.log 2>&1; echo "$v rc=$?"
done
cd $R
python3 - <<'PY'
import csv,glob,collections
for v in ("r5","eu4"):
    acc=collections.defaultdict(float); n=0
    for f in glob.glob(f"gpurun_out/s9_icache/{v}/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            if "chain_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]]+=float(r["Counter_Value"]); 
    print(v,dict(acc))
PY
