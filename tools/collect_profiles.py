"""python tools/collect_profiles.py <tag>: copies the rocprofv3 summaries of `tools/profile.sh <tag>` (gpurun_out/prof_<tag>/) into profiles/
(tracked), stamped with the commit and date they were taken at:
    profiles/<tag>_<what>_kernel_stats.csv   the --kernel-trace --stats tables
    profiles/<tag>_<what>_pmc.json           per-kernel means of the PMC passes (+ HBM traffic per launch: 2 x FETCH_SIZE + WRITE_SIZE KiB;
                                             the factor 2 is calibrated for this access pattern: profiles/r03_fetch_calibration.txt)
    profiles/<tag>_p1_cache.json, <tag>_p1_sq.json   the reach-set build kernels at B = 1 and B = 128"""
import sys
import collections
import csv
import datetime
import glob
import json
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"
SRC = os.path.join(ROOT, "gpurun_out", "prof_" + TAG)
DST = os.path.join(ROOT, "profiles")
KERNELS = ("armour_p2_eval_kernel", "armour_p1_chain_kernel", "armour_p1_tv_kernel", "armour_p1_planes_kernel", "armour_p1_plane_class_kernel",
           "armour_p1_plane_sample_kernel", "armour_solve_kernel", "armour_solve_scan_kernel")
# (the newest commit that touched what was measured: re-running this script after a documentation commit does not move the stamp)
STAMP = {"commit": subprocess.run(["git", "log", "-1", "--format=%h", "--", "armour_amd", "include", "bench.py", "tools/workload.py"], cwd=ROOT, capture_output=True, text=True).stdout.strip(),
         "date": datetime.date.today().isoformat()}


def newest(pattern):
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)


def counters(sub):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(newest(os.path.join(SRC, sub, "**", "*counter_collection.csv")))):
        short = next((x for x in KERNELS if x in row["Kernel_Name"]), None)
        if short:
            a = acc[(short, row["Counter_Name"])]
            a[0] += float(row["Counter_Value"]); a[1] += 1
    out = {}
    for (kn, cn), (s, n) in acc.items():
        out.setdefault(kn, {})[cn] = {"mean_per_dispatch": s / n, "dispatches": n}
    return out


def merge(*ds):
    out = {}
    for d in ds:
        for k, v in d.items():
            out.setdefault(k, {}).update(v)
    return out


def traffic(c):
    return (2 * c["FETCH_SIZE"]["mean_per_dispatch"] + c["WRITE_SIZE"]["mean_per_dispatch"]) * 1024


NOTE = ("FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM section) prescribes for gfx950 -- calibrated for the plane table's "
        "8- and 16-byte-per-lane access patterns: known bytes / (FETCH_SIZE * 1024) = 2.000 for both (profiles/r03_fetch_calibration.txt)")
for tag, name, kern, workload, cmd in (
        ("h", "bench_headline", "armour_p2_eval_kernel", "BASELINE configs[1]: Kinova 7-DOF, O=20, T=100, B=1",
         "python3 bench.py --no-cpu-baseline --headline-only --no-sync-probe --repeats 3 --steps 200 --warmup 20 (kernel stats: --steps 2000 --warmup 200)"),
        ("c2", "bench_configs2", "armour_p2_eval_kernel", "BASELINE configs[2]: Kinova 7-DOF, O=50, T=100, B=128",
         "python3 bench.py --no-cpu-baseline --headline-only --no-sync-probe --repeats 3 --batch 128 --obstacles 50 --steps 40 --warmup 4"),
        ("c4", "bench_configs4", "armour_p2_eval_kernel", "BASELINE configs[4]: Fetch preset (9 links, 7 factors), last link +-50 % payload, O=100, T=100, B=1",
         "python3 tools/workload.py c4 40"),
        ("c48", "bench_configs4_8factor", "armour_p2_eval_kernel", "BASELINE configs[4] as it reads: Fetch 8-DOF (torso yaw + arm: 9 links, 8 factors, 128-bit keys), gripper link +-50 % payload, O=100, T=100, B=1",
         "ARMOUR_KEY128=1 python3 tools/workload.py c4_8f 40"),
        ("s", "solve", "armour_solve_kernel", "armour_solve (device-resident form) of the reference's sample problem, T=100, O=10: 2 evaluation phases per launch",
         "python3 tools/workload.py solve 50")):
    shutil.copy(newest(os.path.join(SRC, tag + "_trace", "**", "*kernel_stats.csv")), os.path.join(DST, f"{TAG}_{name}_kernel_stats.csv"))
    c = merge(counters(tag + "_fetch"), counters(tag + "_write"))
    k = c[kern]
    summary = dict(STAMP, command=f"rocprofv3 --pmc FETCH_SIZE -- {cmd}   (second pass: --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum; kernel stats: --kernel-trace --stats)",
                   workload=workload, counters=c, note=NOTE, hbm_traffic_bytes_per_launch=traffic(k),
                   l2_hit_rate=k["TCC_HIT_sum"]["mean_per_dispatch"] / (k["TCC_HIT_sum"]["mean_per_dispatch"] + k["TCC_MISS_sum"]["mean_per_dispatch"]))
    if kern == "armour_p2_eval_kernel":
        summary["p2_hbm_traffic_bytes_per_launch"] = summary["hbm_traffic_bytes_per_launch"]
    json.dump(summary, open(os.path.join(DST, f"{TAG}_{name}_pmc.json"), "w"), indent=1)
    print(name, "traffic bytes/launch %.0f" % summary["hbm_traffic_bytes_per_launch"], "L2 hit rate %.3f" % summary["l2_hit_rate"])
cache, sq = {}, {}
for B in (1, 128):
    shutil.copy(newest(os.path.join(SRC, f"p1_trace_B{B}", "**", "*kernel_stats.csv")), os.path.join(DST, f"{TAG}_p1_B{B}_kernel_stats.csv"))
    c = merge(counters(f"p1_fetch_B{B}"), counters(f"p1_write_B{B}"), counters(f"p1_l2_B{B}"))
    for kern, v in c.items():
        if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
            continue
        e = {k: x["mean_per_dispatch"] for k, x in v.items()}
        e["dispatches"] = v["FETCH_SIZE"]["dispatches"]
        e["hbm_bytes_per_dispatch"] = traffic(v)
        if "TCC_HIT_sum" in e:
            e["l2_hit_rate"] = e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
        cache.setdefault(f"B={B}", {})[kern] = e
    both = merge(counters(f"p1_sqa_B{B}"), counters(f"p1_sqb_B{B}"))
    kern = "armour_p1_tv_kernel" if "armour_p1_tv_kernel" in both else "armour_p1_chain_kernel"   # batches are built time-vectorised
    v = {k: x["mean_per_dispatch"] for k, x in both[kern].items()}
    v["kernel"] = kern
    v["derived"] = {"valu_insts_per_wave": v["SQ_INSTS_VALU"] / v["SQ_WAVES"], "wait_fraction_of_wave_cycles": v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"],
                    "issue_fraction_of_wave_cycles": v["SQ_ACTIVE_INST_ANY"] / v["SQ_WAVE_CYCLES"],
                    "lds_bank_conflict_fraction": v["SQ_LDS_BANK_CONFLICT"] / max(1.0, v["SQ_LDS_IDX_ACTIVE"])}
    sq[f"B={B}"] = v
pl = counters("pl_fetch")
json.dump(dict(STAMP, command="tools/profile.sh: rocprofv3 --pmc <one group per pass> -- python3 tools/workload.py p1 B   (B = 1 and 128; random worlds O = 20, T = 100); "
                              "'planes at configs[2]': the FETCH_SIZE pass of the configs[2] bench command (B = 128, O = 50)",
               note=NOTE, kernels=cache, planes_at_configs2={k: v for k, v in pl.items() if "plane" in k}),
          open(os.path.join(DST, TAG + "_p1_cache.json"), "w"), indent=1)
json.dump(dict(STAMP, command="rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- python3 tools/workload.py p1 B  "
                              "(second pass: SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE)",
               counters=sq), open(os.path.join(DST, TAG + "_p1_sq.json"), "w"), indent=1)
for B in (1, 128):
    for kern, e in cache.get(f"B={B}", {}).items():
        print(f"P1 B={B} {kern}: HBM bytes per dispatch {e['hbm_bytes_per_dispatch']:.3e}, L2 hit rate {e.get('l2_hit_rate', float('nan')):.3f}, dispatches {e['dispatches']}")
    print(f"P1 B={B} SQ:", sq[f"B={B}"]["kernel"], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in sq[f"B={B}"]["derived"].items()})
shutil.copy(newest(os.path.join(SRC, "cull_trace", "**", "*kernel_stats.csv")), os.path.join(DST, TAG + "_cull_configs2_kernel_stats.csv"))
with open(os.path.join(DST, TAG + "_cull_configs2.txt"), "w") as f:   # (what the workload itself timed with HIP events, under the kernel trace)
    f.write(f"# python3 tools/workload.py cull 128 50 under rocprofv3 --kernel-trace --stats, commit {STAMP['commit']}, {STAMP['date']}\n")
    f.write("".join(l for l in open(os.path.join(SRC, "cull_trace.log")) if l.startswith(("full:", "culled:", "records identical"))))
for f in (TAG + "_bench_headline", TAG + "_bench_configs2", TAG + "_solve", TAG + "_p1_B1", TAG + "_p1_B128"):
    print("==", f); print("".join(open(os.path.join(DST, f + "_kernel_stats.csv")).readlines()[:6]))
