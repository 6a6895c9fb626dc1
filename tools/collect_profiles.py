"""Copies the rocprofv3 summaries of tools/gpu_profile.sh (gpurun_out/prof_<tag>/) into profiles/ (tracked):
    profiles/<name>_kernel_stats.csv   the --kernel-trace --stats table
    profiles/<name>_pmc.json           per-kernel means of the PMC passes + the P2 HBM traffic per launch
Usage: python tools/collect_profiles.py <tag> <name> [workload label] [extra bench arguments of the run]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "BASELINE configs[1]: Kinova 7-DOF, O=20, T=100, B=1"
extra = (" " + sys.argv[4]) if len(sys.argv) > 4 else ""
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
newest = lambda pattern: max(glob.glob(pattern, recursive=True), key=os.path.getmtime)  # gpurun_out/ accumulates earlier runs
stats = newest(os.path.join(src, "trace", "**", "*kernel_stats.csv"))
shutil.copy(stats, os.path.join(ROOT, "profiles", name + "_kernel_stats.csv"))
out = {}
for sub in ("pmc_fetch", "pmc_write"):
    for f in [newest(os.path.join(src, sub, "**", "*counter_collection.csv"))]:
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            k = (row["Kernel_Name"], row["Counter_Name"])
            acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
        for (kn, cn), (s, n) in acc.items():
            short = next((x for x in ("armour_p2_eval_kernel", "armour_p1_chain_kernel", "armour_p1_planes_kernel") if x in kn), None)
            if short:
                out.setdefault(short, {})[cn] = {"mean_per_dispatch": s / n, "dispatches": n}
p2 = out["armour_p2_eval_kernel"]
summary = {
    "command": "rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --no-cpu-baseline --headline-only --steps 200 --warmup 20" + extra + "  "
               "(second pass: --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum; kernel stats: --kernel-trace --stats, default steps unless given)",
    "workload": workload,
    "counters": out,
    "p2_hbm_traffic_bytes_per_launch": (2 * p2["FETCH_SIZE"]["mean_per_dispatch"] + p2["WRITE_SIZE"]["mean_per_dispatch"]) * 1024,
    "note": "FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM section) prescribes for gfx950 wide coalesced reads",
    "l2_hit_rate_p2": p2["TCC_HIT_sum"]["mean_per_dispatch"] / (p2["TCC_HIT_sum"]["mean_per_dispatch"] + p2["TCC_MISS_sum"]["mean_per_dispatch"]),
}
json.dump(summary, open(os.path.join(ROOT, "profiles", name + "_pmc.json"), "w"), indent=1)
print(open(os.path.join(ROOT, "profiles", name + "_kernel_stats.csv")).read())
print("P2 traffic bytes/launch", summary["p2_hbm_traffic_bytes_per_launch"], "L2 hit rate", summary["l2_hit_rate_p2"])
