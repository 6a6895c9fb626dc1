#!/bin/bash
# dev: HBM traffic per block of the time-vectorised kernel at several batch sizes (2 x FETCH_SIZE + WRITE_SIZE KiB per dispatch / blocks): how much of B = 128's traffic is capacity misses of shared caches
export TMPDIR=/tmp
R=$PWD; OUT=$R/gpurun_out/tvtraffic; rm -rf $OUT; mkdir -p $OUT; cd /tmp
for B in 8 32 64 128; do
  for c in FETCH_SIZE WRITE_SIZE; do
    ARMOUR_P1_TV=1 timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $OUT/B${B}_$c -- python3 $R/tools/p1_once.py $B > $OUT/B${B}_$c.log 2>&1; echo "B=$B $c rc=$?"
  done
done
cd $R
python3 - <<'PY'
import csv, glob
for B in (8, 32, 64, 128):
    v = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        s, n = 0.0, 0
        for f in glob.glob(f"gpurun_out/tvtraffic/B{B}_{c}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if "tv_kernel" in row["Kernel_Name"] and row["Counter_Name"] == c: s += float(row["Counter_Value"]); n += 1
        v[c] = s / max(n, 1)
    blocks = 2 * B
    tot = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
    print(f"B={B}: {blocks} blocks, HBM bytes per dispatch {tot:.3e} (reads {2 * v['FETCH_SIZE'] * 1024:.3e}, writes {v['WRITE_SIZE'] * 1024:.3e}) = {tot / blocks / 1e6:.1f} MB per block")
PY
find $OUT -name "*.csv" -size +1M -delete
