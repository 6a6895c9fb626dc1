"""Host-side logic that needs no GPU: synthetic worlds, the reference's file protocol, sharding (incl. a
world_size-2 gloo run of the multi-rank plumbing bench.py uses)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_random_worlds_are_deterministic_and_in_range():
    from armour_amd.worlds import SPEED, STATE_LB, STATE_UB, random_batch, random_k, random_problem
    a, b = random_problem(3, 20), random_problem(3, 20)
    for k in a:
        assert np.array_equal(a[k], b[k])
    assert a["obstacles"].shape == (20, 12)
    lim = np.abs(STATE_LB) < 1000
    assert np.all(a["q0"][lim] >= STATE_LB[lim] + 0.3) and np.all(a["q0"][lim] <= STATE_UB[lim] - 0.3)
    assert np.all(np.abs(a["qd0"]) <= 0.5 * SPEED) and np.all(np.abs(a["qdd0"]) <= 1)
    # boxes are [c, diag(s/2)] column-major (SIM/worlds/obstacles/box_obstacle_zonotope.m:21-26)
    ob = a["obstacles"][0]
    assert ob[4] == ob[5] == ob[6] == ob[8] == ob[9] == ob[10] == 0 and 0.005 <= ob[3] <= 0.25
    bt = random_batch(10, 4, 5)
    assert bt["q0"].shape == (4, 7) and bt["obstacles"].shape == (4, 5, 12)
    assert np.array_equal(bt["q0"][2], random_problem(12, 5)["q0"])
    k = random_k(0, 64)
    assert k.shape == (64, 7) and np.all(np.abs(k) <= 1)


def test_scene_csv_loader():
    from armour_amd.worlds import load_scene_csv
    qs, qg, obs = load_scene_csv(os.path.join(ROOT, "tests", "golden", "scene_013_001.csv"))
    assert qs.shape == (7,) and qg.shape == (7,) and obs.shape == (6, 12)
    assert abs(qs[2] - 2.859) < 1e-12 and abs(obs[0, 3] - 0.037916 / 2) < 1e-15 and abs(obs[5, 11] - 0.37495 / 2) < 1e-15


def test_file_protocol_roundtrip(tmp_path):
    from armour_amd import file_protocol as fp
    from helpers import SAMPLE_PROBLEM as p
    path = tmp_path / fp.IN_NAME
    fp.write_armour_in(path, p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    lines = open(path).read().splitlines()
    assert len(lines) == 4 + 1 + 10 and lines[4].strip() == "10"
    assert lines[0].split()[0] == "0.6543000000"          # "%.10f " as uarmtd_planner.m:160
    back = fp.parse_armour_in(path)
    for k in p:
        assert np.allclose(back[k], p[k], rtol=0, atol=5e-11)
    # error conventions of RT/armour_main.cu:47-71
    with pytest.raises(ValueError):
        fp.parse_armour_in(tmp_path / "missing.in")
    with pytest.raises(ValueError):
        fp.parse_armour_in(path, max_obstacles=5)
    # outputs
    T, J, n, m = 4, 7, 7, 11
    rng = np.random.default_rng(0)
    cen, gens, tr, g = rng.normal(size=(T, J, 3)), rng.normal(size=(T, J, 3, 6)), rng.uniform(1, 2, (n, T)), rng.normal(size=m)
    fp.write_outputs(tmp_path, np.arange(7) / 10.0, 123.0, cen, gens, tr, g)
    k_opt, ms = fp.read_armour_out(tmp_path / "armour.out")
    assert np.allclose(k_opt, np.arange(7) / 10.0) and ms == 123.0
    assert np.loadtxt(tmp_path / "armour_joint_position_center.out").shape == (T * J, 3)
    assert np.loadtxt(tmp_path / "armour_joint_position_radius.out").shape == (T * J * 3, 6)
    assert np.allclose(np.loadtxt(tmp_path / "armour_control_input_radius.out"), tr.T, rtol=1e-9)
    assert np.allclose(np.loadtxt(tmp_path / "armour_constraints.out"), g, rtol=1e-5)
    fp.write_outputs(tmp_path, None, 5.0, cen, gens, tr, g)   # infeasible: single -1 then the time
    k_opt, ms = fp.read_armour_out(tmp_path / "armour.out")
    assert k_opt is None and ms == 5.0 and open(tmp_path / "armour.out").read().split()[0] == "-1"


def test_shard_ranges_cover_exactly_once():
    from armour_amd.sharding import shard_range, shard_seeds
    for total in (1, 7, 128, 1024, 1025):
        for world in (1, 2, 4, 8):
            rs = [shard_range(total, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == total
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in rs]
            assert max(sizes) - min(sizes) <= 1
    assert shard_seeds(100, 10, 1, 4) == [103, 104, 105]


_GLOO_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from armour_amd.sharding import shard_seeds, reduce_max_elapsed, gather_counts
from armour_amd.worlds import random_problem
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
seeds = shard_seeds(0, 9, rank, world)
# every rank builds only its own worlds; nothing is exchanged on the data path
probs = [random_problem(s, 3) for s in seeds]
counts = gather_counts(len(probs))
mx = reduce_max_elapsed(0.5 + rank)
all_seeds = [None] * world
dist.all_gather_object(all_seeds, seeds)
if rank == 0:
    flat = sorted(s for part in all_seeds for s in part)
    assert flat == list(range(9)), flat
    assert sum(counts) == 9 and counts == [5, 4], counts
    assert mx == 0.5 + (world - 1), mx
    print("GLOO_OK")
dist.destroy_process_group()
"""


def test_two_rank_gloo_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29531", str(script), ROOT], capture_output=True, text=True, timeout=300, env=env)
    assert "GLOO_OK" in out.stdout, out.stdout + out.stderr


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` as the driver invokes it (no torch.distributed.run around it): the parent starts one
    process per rank, they meet on 127.0.0.1 and rank 0 prints ONE JSON line.  --dry-run swaps the GPU work for a sleep and
    RCCL for gloo; launcher, rendezvous and the timing reduction (armour_amd.sharding) are the code of the real run."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "1", "--repeats", "3",
                          "--batch", "3", "--dry-run"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]   # (gloo prints a connection banner on stdout)
    assert len(lines) == 1, out.stdout
    js = json.loads(lines[0])
    assert js["n_gpus"] == 2 and js["steps"] == 20 and js["scaling"] == "weak" and js["config"]["problems_per_rank"] == [3, 3]
    # the slowest rank (rank 1 sleeps twice as long) sets the interval: value = 6 problems * 20 steps / max-rank time
    assert js["ms_per_step"] >= 2 * 0.1 * 0.9 and abs(js["value"] - 6 * 20 / (js["ms_per_step"] * 20e-3)) <= 1e-6 * js["value"]
    # a mismatch between --gpus and an existing WORLD_SIZE is refused, not silently re-interpreted
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                         timeout=120, env=dict(env, WORLD_SIZE="1", RANK="0"))
    assert bad.returncode != 0 and "does not match" in bad.stderr


def _matlab_desired_trajectory(q0, qd0, qdd0, q1, t):
    """KSI/uarmtd_planner.m:886-905 restated: match_deg5_bernstein_coefficients (final velocity and acceleration 0,
    T = 1) -> bernstein_to_poly -> power-basis evaluation of position, velocity and acceleration."""
    from math import comb
    out = []
    for j in range(len(q0)):
        beta = [q0[j], q0[j] + qd0[j] / 5, qdd0[j] / 20 + 2 * qd0[j] / 5 + q0[j], q1[j], q1[j], q1[j]]
        alpha = [sum((-1) ** (i - jj) * comb(5, i) * comb(i, jj) * beta[jj] for jj in range(i + 1)) for i in range(6)]
        q = sum(alpha[c] * t ** c for c in range(6))
        qd = sum(c * alpha[c] * t ** (c - 1) for c in range(1, 6))
        qdd = sum(c * (c - 1) * alpha[c] * t ** (c - 2) for c in range(2, 6))
        out.append((q, qd, qdd))
    return tuple(np.array(x) for x in zip(*out))


def test_desired_trajectory_matches_the_matlab_bernstein_formulation():
    """armour_desired_trajectory (the C++ closed form of RT/Trajectory.cu:542-602) against the MATLAB caller's
    Bernstein -> monomial evaluation of the same curve; end conditions; braking fallback."""
    from armour_amd.planner import desired_trajectory
    rng = np.random.default_rng(5)
    q0, qd0, qdd0 = rng.uniform(-3, 3, 7), rng.uniform(-1, 1, 7), rng.uniform(-1, 1, 7)
    k = rng.uniform(-1, 1, 7)
    kr = np.full(7, np.pi / 48)
    for t in (0.0, 0.13, 0.5, 0.77, 1.0):
        q, qd, qdd = desired_trajectory(q0, qd0, qdd0, k, t)
        rq, rqd, rqdd = _matlab_desired_trajectory(q0, qd0, qdd0, q0 + kr * k, t)
        assert np.abs(q - rq).max() <= 1e-12 and np.abs(qd - rqd).max() <= 1e-11 and np.abs(qdd - rqdd).max() <= 1e-10
    q, qd, qdd = desired_trajectory(q0, qd0, qdd0, k, 0.0)
    assert np.abs(q - q0).max() <= 1e-15 and np.abs(qd - qd0).max() <= 1e-14 and np.abs(qdd - qdd0).max() <= 1e-13
    q, qd, qdd = desired_trajectory(q0, qd0, qdd0, k, 1.0)
    assert np.abs(q - (q0 + kr * k)).max() <= 1e-14 and np.abs(qd).max() <= 1e-13 and np.abs(qdd).max() <= 1e-12
    # duration scaling: the curve over [0, D] with the same end conditions
    q2, qd2, qdd2 = desired_trajectory(q0, qd0, qdd0, k, 1.0, duration=2.0)
    h = 1e-6
    qa, _, _ = desired_trajectory(q0, qd0, qdd0, k, 1.0 + h, duration=2.0)
    qb, _, _ = desired_trajectory(q0, qd0, qdd0, k, 1.0 - h, duration=2.0)
    assert np.abs((qa - qb) / (2 * h) - qd2).max() <= 1e-8
    # no plan (k = NaN): follow the previous plan shifted by t_plan while moving, else hold
    prev = lambda tt: desired_trajectory(q0, qd0, qdd0, k, tt)
    nan = np.full(7, np.nan)
    bq, bqd, _ = desired_trajectory(q0 + 0.1, qd0, qdd0, nan, 0.2, previous=prev)
    pq, pqd, _ = prev(0.7)
    assert np.array_equal(bq, pq) and np.array_equal(bqd, pqd)
    hq, hqd, hqdd = desired_trajectory(q0, np.zeros(7), qdd0, nan, 0.2, previous=prev)
    assert np.array_equal(hq, q0) and not hqd.any() and not hqdd.any()
    hq, hqd, _ = desired_trajectory(q0, qd0, qdd0, None, 0.6, previous=prev)
    assert np.array_equal(hq, q0) and not hqd.any()


def test_resident_planner_protocol_without_a_gpu(tmp_path):
    """The socket protocol of `armour_main --serve` (armour_amd/csrc/cli_common.h), with no handles created up front so
    that it runs on a CPU-only host: requests reach the resident process, failures come back as the reference's
    conventions (-1 in the .out file, non-zero exit code), `--quit` ends it and removes the socket."""
    import os
    import subprocess
    import time
    from conftest import ROOT
    bindir = os.path.join(ROOT, "armour_amd", "bin")
    exe, exe2 = os.path.join(bindir, "armour_main"), os.path.join(bindir, "armtd_main")
    if not (os.path.exists(exe) and os.path.exists(exe2)):
        pytest.skip("executables not built (make -C armour_amd/csrc)")
    daemon = subprocess.Popen([exe, "--serve", str(tmp_path), "0", "0"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        t0 = time.time()
        while not os.path.exists(tmp_path / "armour.sock"):
            assert daemon.poll() is None and time.time() - t0 < 30
            time.sleep(0.01)
        for binary, out in ((exe, "armour.out"), (exe2, "armtd.out")):
            r = subprocess.run([binary, str(tmp_path)], capture_output=True, text=True, timeout=60)
            assert r.returncode == 1 and open(tmp_path / out).read().split() == ["-1"]
            assert r.stdout == "" and r.stderr == ""          # the iteration ran in the resident process, not here
        assert subprocess.run([exe, "--quit", str(tmp_path)], timeout=60).returncode == 0
        assert daemon.wait(timeout=30) == 0 and not os.path.exists(tmp_path / "armour.sock")
        log = daemon.stdout.read()
        assert log.count("Error reading input files") == 2 and "listening" in log
        # nobody listening any more: the executable runs the iteration itself (and reports the missing input)
        r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=60)
        assert r.returncode == 1 and "Error reading input files" in r.stderr
        assert subprocess.run([exe, "--quit", str(tmp_path)], timeout=60).returncode == 1
    finally:
        if daemon.poll() is None:
            daemon.kill()
            daemon.wait()


def test_resident_planner_survives_thousands_of_iterations_and_dies_with_its_front_end(tmp_path):
    """Two findings of the round-1 review, on a CPU-only host (no handles: every iteration fails fast with the reference's
    "-1" convention, which is all this needs).  (1) file_protocol.ResidentPlanner used to hand the planner a pipe nobody
    read: after ~64 KB of per-iteration "HIP & C++: ..." lines it blocked in fflush and every forwarding armour_main hung.
    The output now goes to a log file: 1500 iterations (> 64 KB of output) complete.  (2) `armour_main --serve` no longer
    exec()s the worker (an exec from a GPU-initialised process, e.g. under rocprofv3, takes the machine down on this pool);
    it spawns it, relays SIGTERM, and the worker asks for SIGTERM on its parent's death: killing the front end with
    SIGKILL does not leave a worker behind holding the socket."""
    import signal
    import time
    from armour_amd import file_protocol as fp
    bindir = os.path.join(ROOT, "armour_amd", "bin")
    exe = os.path.join(bindir, "armour_main")
    if not os.path.exists(exe):
        pytest.skip("executables not built (make -C armour_amd/csrc)")
    with fp.ResidentPlanner(tmp_path, 0, 0) as rp:
        for i in range(1500):
            r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=30)
            assert r.returncode == 1 and r.stdout == "" and r.stderr == "", (i, r.stderr)
        log = rp.log()
        assert log.count("Error reading input files") == 1500 and len(log) > 65536
    assert not os.path.exists(tmp_path / "armour.sock")
    # (2) SIGKILL the front end: the worker must go away by itself
    rp = fp.ResidentPlanner(tmp_path, 0, 0)
    assert subprocess.run([exe, str(tmp_path)], capture_output=True, timeout=30).returncode == 1      # served
    rp.proc.send_signal(signal.SIGKILL)
    rp.proc.wait()
    t0 = time.time()
    while True:
        # once the worker is gone nobody answers on the socket and the executable runs the iteration itself (stderr non-empty)
        r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=30)
        if r.stderr:
            break
        assert time.time() - t0 < 20, "the worker outlived its front end"
        time.sleep(0.05)
    # SIGTERM to the front end is relayed: clean shutdown, socket removed, exit code 0
    rp = fp.ResidentPlanner(tmp_path, 0, 0)
    rp.proc.send_signal(signal.SIGTERM)
    assert rp.proc.wait(timeout=30) == 0 and not os.path.exists(tmp_path / "armour.sock")


def test_output_files_are_formatted_as_the_reference_formats_them(tmp_path):
    """The five .out files are `ofstream << setprecision(10)` in the reference (RT/armour_main.cu:312-372; the constraints with precision 6).  The
    worker formats them with std::to_chars into one buffer per file (cli_common.h OutFile): the same bytes as the iostream form, number for
    number -- checked here on two million doubles of every magnitude, on the CPU."""
    import subprocess
    src = tmp_path / "fmt.cpp"
    src.write_text(r'''
#include <cmath>
#include <cstdio>
#include <cstring>
#include <iomanip>
#include <random>
#include <sstream>
#include "cli_common.h"
int main() {
    std::mt19937_64 r(7);
    int bad = 0;
    for (int i = 0; i < 2000000; i++) {
        const unsigned long long b = r();
        double v;
        if (i % 3 == 0) memcpy(&v, &b, 8);
        else if (i % 3 == 1) v = (double)(long long)(b % 2000001) / (double)((b >> 32) % 100000 + 1) - 7.0;
        else { v = std::ldexp((double)(b & 0xfffff) / 1048576.0, (int)((b >> 40) % 60) - 30); if (b & 1) v = -v; }
        if (!std::isfinite(v)) continue;
        for (int prec : {10, 6}) {
            cli::OutFile o(1);
            o.num(v, prec);
            std::ostringstream ref;
            ref << std::setprecision(prec) << v;
            if (ref.str() != std::string(o.buf.data(), o.len)) bad++;
        }
    }
    cli::OutFile o(4);
    o.integer(-1); o.ch('\n'); o.num(0.5, 10); o.ch(' '); o.num(-1e-7, 10);
    if (std::string(o.buf.data(), o.len) != "-1\n0.5 -1e-07") bad++;
    printf("bad %d\n", bad);
    return bad != 0;
}
''')
    exe = tmp_path / "fmt"
    inc = os.path.join(ROOT, "armour_amd", "csrc")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", inc, str(src), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "bad 0", out.stdout + out.stderr
