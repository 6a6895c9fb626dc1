"""What one planning iteration costs through the file protocol: stand-alone executable (pays the GPU start-up every time)
against the resident planner (`armour_main --serve`).  Development probe."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from armour_amd import file_protocol as fp  # noqa: E402
from helpers import SAMPLE_PROBLEM as p  # noqa: E402

exe = os.path.join(ROOT, "armour_amd", "bin", "armour_main")
with tempfile.TemporaryDirectory() as d:
    fp.write_armour_in(os.path.join(d, fp.IN_NAME), p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])

    def run(tag):
        t0 = time.perf_counter()
        rc = subprocess.run([exe, d, "128"], capture_output=True).returncode
        wall = (time.perf_counter() - t0) * 1e3
        print(f"{tag}: exit {rc}, process wall {wall:.1f} ms, planning time written to armour.out {open(os.path.join(d, 'armour.out')).read().split()[-1]} ms")

    for i in range(3):
        run("stand-alone")
    with fp.ResidentPlanner(d, 128, 100):
        for i in range(5):
            run("resident   ")
