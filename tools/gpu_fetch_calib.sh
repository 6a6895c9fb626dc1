#!/bin/bash
# FETCH_SIZE calibration for the plane-table access pattern (tools/micro/fetch_calib.hip); run on the GPU box.
set -u
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/fetch_calib
rm -rf $OUT; mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 -Wno-unused-value --offload-arch=gfx950 $R/tools/micro/fetch_calib.hip -o $OUT/fetch_calib || exit 1
cd /tmp
for w in 8 16; do
  $OUT/fetch_calib $w > $OUT/run_$w.txt 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_$w -- $OUT/fetch_calib $w > $OUT/pmc_$w.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, os
out = os.path.join(os.environ.get("PWD", "."), "gpurun_out", "fetch_calib")
known = 35000 * 128 * 60 * 8
for w in (8, 16):
    print(open(os.path.join(out, f"run_{w}.txt")).read().strip().splitlines()[-1])
    f = glob.glob(os.path.join(out, f"pmc_{w}", "**", "*counter_collection.csv"), recursive=True)
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == "FETCH_SIZE" and "read" in r["Kernel_Name"]]
    mean = sum(vals) / len(vals)
    print(f"read{w}: known bytes {known}, FETCH_SIZE {mean:.0f} KiB per dispatch -> bytes / (FETCH_SIZE * 1024) = {known / (mean * 1024):.3f}")
PY
