"""Transport A (the reference's process + text-file protocol) end to end: one planning iteration of the reference's sample problem through
`armour_main` with a resident planner, process wall around the client, the time the planner writes into armour.out, and the worker's own
stamps (ARMOUR_CLI_TIMING=1: reach sets / solve / diagnostics / files).  GPU box:  python tools/dev/transport_a_probe.py [reps=30]"""
import os
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from armour_amd import file_protocol as fp  # noqa: E402
from armour_amd.worlds import reference_sample_problem  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
exe = os.path.join(ROOT, "armour_amd", "bin", "armour_main")
p = reference_sample_problem()
os.environ["ARMOUR_CLI_TIMING"] = "1"
for T in (100, 128):
    with tempfile.TemporaryDirectory() as d:
        fp.write_armour_in(os.path.join(d, fp.IN_NAME), p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
        with fp.ResidentPlanner(d, T, 100) as rp:
            wall, written = [], []
            for r in range(reps + 3):
                t0 = time.perf_counter()
                pid = os.posix_spawn(exe, [exe, d, str(T)], os.environ)   # (no fork of this interpreter: what a C or MATLAB caller's spawn costs)
                _, status = os.waitpid(pid, 0)
                w = (time.perf_counter() - t0) * 1e3
                assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0
                k_opt, ms = fp.read_armour_out(os.path.join(d, "armour.out"))
                if r >= 3:
                    wall.append(w); written.append(ms)
            sizes = {n: os.path.getsize(os.path.join(d, n)) for n in sorted(os.listdir(d)) if n.endswith(".out")}
            log = rp.log()
        stamps = [l.strip() for l in log.splitlines() if "[timing]" in l][-1:] or ["(no stamps in the log)"]
        print(f"T = {T}: process wall around armour_main (posix_spawn + socket + iteration + five files) median {statistics.median(wall):.2f} ms, "
              f"min {min(wall):.2f}, max {max(wall):.2f}; time written to armour.out median {statistics.median(written):.2f} ms; feasible {k_opt is not None}")
        print("   worker stamps of the last iteration:", stamps[0].replace("HIP & C++: ", ""))
        print("   bytes written:", sizes)
