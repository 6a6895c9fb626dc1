#!/bin/bash
# dev: cache behaviour of the reach-set build kernels (armour_p1_chain_kernel at B = 1, armour_p1_tv_kernel at B = 128) (L2 hit rate, HBM traffic, vector-L1 traffic) at B = 1 and B = 128
set -u
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof_p1cache
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for B in 1 128; do
  # one counter group per pass, every pass under its own timeout: a refused combination ("exceeds the capabilities of the
  # hardware") makes rocprofv3 abort and then sit there until killed
  timeout 180 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/l2_B$B -- python3 $R/tools/p1_once.py $B > $OUT/l2_B$B.log 2>&1
  timeout 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_B$B -- python3 $R/tools/p1_once.py $B > $OUT/fetch_B$B.log 2>&1
  timeout 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_B$B -- python3 $R/tools/p1_once.py $B > $OUT/write_B$B.log 2>&1
  timeout 180 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/l1_B$B -- python3 $R/tools/p1_once.py $B > $OUT/l1_B$B.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/prof_p1cache/*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        if 'chain_kernel' in row['Kernel_Name'] or 'tv_kernel' in row['Kernel_Name']:
            a = acc[row['Counter_Name']]; a[0] += float(row['Counter_Value']); a[1] += 1
    print(f.split('/')[2], {k: round(v[0] / v[1], 1) for k, v in acc.items()}, 'dispatches', {k: v[1] for k, v in acc.items()})
PY
