#!/bin/bash
# Round-4 quick check on the GPU box: gpu tests, the default bench line, and the configs[2] counter passes (traffic per launch).
set -u
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r4_quick
rm -rf $OUT; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cd /tmp
HB="--no-cpu-baseline --headline-only --no-sync-probe --repeats 3"
C2="--batch 128 --obstacles 50 --steps 40 --warmup 4"
run() { d=$1; shift; timeout -k 10 240 "$@" > $OUT/$d.log 2>&1; echo "$d rc=$?"; }
run c2_trace rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2_trace -- python3 $R/bench.py $HB $C2
run c2_fetch rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c2_fetch -- python3 $R/bench.py $HB $C2
run c2_write rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/c2_write -- python3 $R/bench.py $HB $C2
cd $R
find $OUT -name "*kernel_trace.csv" -size +2M -delete
python3 - <<'PY'
import csv, glob, os, json, collections
out = os.path.join(os.getcwd(), "gpurun_out", "r4_quick")
d = json.load(open(os.path.join(out, "bench.json")))
print("headline", d["value"], d["ms_per_step"], d["roofline"].get("launch_us"))
for k, v in d["other_configs"].items(): print(k, v.get("ms_per_step"), v.get("launch_us"), v.get("p1_set_problems_ms_per_problem"), v.get("effective_bytes_per_launch"))
for sub in ("c2_fetch", "c2_write"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "armour_p2_eval_kernel" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for c, (s, n) in acc.items(): print(sub, c, s / n, n)
for f in glob.glob(os.path.join(out, "c2_trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "armour" in r["Name"]: print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
