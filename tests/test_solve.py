"""armour_solve (SQP on the device callbacks) and the armour_main CLI on the GPU.

IPOPT is absent, so the check is against an independent solver on the CPU oracle's callbacks (scipy SLSQP) and
against the optimality / feasibility conditions themselves."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from helpers import SAMPLE_PROBLEM

pytestmark = pytest.mark.gpu


def _scipy_reference(o, m, gl, gu):
    from scipy.optimize import minimize
    two = gl > -1e18

    def cons(k):
        g, _ = o.eval_g_jac(k, want_jac=False)
        return np.concatenate([gu - g, (g - gl)[two]])

    def cons_jac(k):
        _, j = o.eval_g_jac(k, want_g=False)
        return np.concatenate([-j, j[two]])

    r = minimize(o.eval_f, np.zeros(7), jac=o.eval_grad_f, bounds=[(-1, 1)] * 7, method="SLSQP",
                 constraints=[dict(type="ineq", fun=cons, jac=cons_jac)], options=dict(maxiter=200, ftol=1e-12))
    return r


# (the last three: BASELINE's trajectory length, T = 100, with 5 and 8 obstacles -- 4 200 and 6 300 constraint rows; seeds on which the
#  independent solver ends feasible, so the comparison is one of optima and not of two failure modes)
@pytest.mark.parametrize("seed,O,T", [(3, 3, 10), (5, 6, 10), (11, 0, 20), (1, 5, 100), (4, 8, 100), (12, 8, 100)])
def test_solve_matches_independent_solver(seed, O, T):
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_problem
    from oracle.cpu_oracle import Oracle
    p = random_problem(seed, O)
    nlp = ArmourNLP(T=T).set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    sol = nlp.solve(tolerance=1e-7, max_iterations=100)[0]
    o = Oracle(T=T).set_problem(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    _, _, gl, gu = o.bounds()
    ref = _scipy_reference(o, o.m, gl, gu)
    g_ref, _ = o.eval_g_jac(ref.x, want_jac=False)
    ref_feasible = bool(np.all(g_ref <= gu + 1e-6) and np.all(g_ref >= gl - 1e-6))
    g, _ = o.eval_g_jac(sol["k_opt"], want_jac=False)
    if ref_feasible:
        # same NLP, same start: both SQP methods should land on the same local optimum
        assert sol["max_violation"] <= 1e-6
        assert np.all(np.abs(sol["k_opt"]) <= 1 + 1e-12)
        assert sol["cost"] <= ref.fun + 1e-6 * (1 + abs(ref.fun)), (sol, ref.fun)
        assert abs(sol["cost"] - o.eval_f(sol["k_opt"])) <= 1e-10
        assert np.all(g <= gu + 1e-6) and np.all(g >= gl - 1e-6)
        assert sol["feasible"]
    else:
        assert not sol["feasible"] or sol["max_violation"] <= 1e-2


def _same_solution(a, c):
    return (np.array_equal(a["k_opt"], c["k_opt"]) and a["feasible"] == c["feasible"] and a["iterations"] == c["iterations"]
            and a["evaluations"] == c["evaluations"] and a["status"] == c["status"] and a["cost"] == c["cost"]
            and a["max_violation"] == c["max_violation"])


@pytest.mark.parametrize("seed,O,T,B", [(3, 3, 20, 1), (7, 12, 100, 1), (8, 30, 100, 3), (30, 4, 20, 3), (0, 20, 100, 1), (2, 0, 100, 2)])
def test_device_resident_solve_equals_the_host_driven_one(seed, O, T, B, monkeypatch):
    """SURVEY.md 8f rank 1.  armour_solve runs the whole SQP iterate in one persistent kernel (solver_device.hip: evaluation,
    scan, candidate rows, Goldfarb-Idnani QP, merit line search, finalize_solution verdict -- no host round trip per
    evaluation).  The round-1 form (one launch per evaluation, QPs on the host) is kept behind `force_host_qp`; both must
    produce the SAME iterates: k_opt, cost, violation, counts and verdict bit for bit -- for feasible and infeasible
    problems, for batches (whose problems no longer run in lock step), and whatever number of blocks shares a problem."""
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch
    bp = random_batch(seed, B, O)
    nlp = ArmourNLP(T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    host = nlp.solve(host_qp=True)
    dev = nlp.solve(device_qp=True)          # (the automatic choice since round 6; asked for here)
    for a, c in zip(host, dev):
        assert _same_solution(a, c), (a, c)
    for a, c in zip(host, nlp.solve()):      # ... and whatever the automatic choice is, the same again
        assert _same_solution(a, c), (a, c)
    from armour_amd import _lib
    for blocks in (1, 5, 64):   # (ARMOUR_OPT_SOLVE_BLOCKS: a handle option since round 4)
        nlp.set_option(_lib.OPT_SOLVE_BLOCKS, blocks)
        for a, c in zip(host, nlp.solve(device_qp=True)):
            assert _same_solution(a, c), (blocks, a, c)
    nlp.set_option(_lib.OPT_SOLVE_BLOCKS, 0)
    # a batch cut into sub-batches launched back to back (round 3: what large batches with many obstacles do by themselves)
    if B > 1:
        for sub in (1, 2):
            nlp.set_option(_lib.OPT_SOLVE_SUB_BATCH, sub)
            for a, c in zip(host, nlp.solve(device_qp=True)):
                assert _same_solution(a, c), ("sub-batch", sub, a, c)
        nlp.set_option(_lib.OPT_SOLVE_SUB_BATCH, 0)
    # tighter tolerance / more iterations: longer iterate sequences
    for a, c in zip(nlp.solve(tolerance=1e-7, max_iterations=100, host_qp=True), nlp.solve(tolerance=1e-7, max_iterations=100, device_qp=True)):
        assert _same_solution(a, c), (a, c)


@pytest.mark.parametrize("seed,O,T,B", [(8, 30, 100, 3), (5000, 50, 100, 12), (7, 12, 100, 1), (2, 0, 100, 2), (3, 3, 20, 1)])
def test_culled_device_solve_equals_the_full_one(seed, O, T, B):
    """ARMOUR_OPT_SOLVE_CULL: the persistent kernel walks only the rows that can pass its candidate filter for some k (relevance.hip: the
    solver's mask; listed collision rows one per thread from their packed plane entries, listed torque tiles, the limit rows).  Every other
    row adds nothing the solver reads, and a listed row's g / Jacobian are the fused evaluation's bit for bit -- so k_opt, cost, violation,
    counts and verdict equal the full device form's and the host form's, for every block count and sub-batch cut."""
    from armour_amd import _lib
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch
    bp = random_batch(seed, B, O)
    nlp = ArmourNLP(T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    host = nlp.solve(host_qp=True)
    nlp.set_option(_lib.OPT_SOLVE_CULL, 0)
    full = nlp.solve(device_qp=True)
    nlp.set_option(_lib.OPT_SOLVE_CULL, 1)
    for blocks in (0, 1, 3, 64):
        nlp.set_option(_lib.OPT_SOLVE_BLOCKS, blocks)
        for b, (h, f, c) in enumerate(zip(host, full, nlp.solve(device_qp=True))):
            assert _same_solution(f, c) and _same_solution(h, c), (blocks, b, h, f, c)
    nlp.set_option(_lib.OPT_SOLVE_BLOCKS, 0)
    if B > 1:
        nlp.set_option(_lib.OPT_SOLVE_SUB_BATCH, 2)
        for f, c in zip(full, nlp.solve(device_qp=True)):
            assert _same_solution(f, c), ("sub-batch", f, c)
        nlp.set_option(_lib.OPT_SOLVE_SUB_BATCH, 0)
    for a, c in zip(nlp.solve(tolerance=1e-7, max_iterations=100, host_qp=True), nlp.solve(tolerance=1e-7, max_iterations=100, device_qp=True)):
        assert _same_solution(a, c), (a, c)
    nlp.close()


def test_accepted_steps_keep_the_trajectory_parameter_in_its_box():
    """Round 5 (tools/dev/solve_stress.py): a QP result was taken as feasible although an ACTIVE row -- a variable's bound -- had drifted by 1.6
    (ill-conditioned N'G^-1 N: the rows "satisfied with equality" are not looked at again), the SQP went to a trajectory parameter of -2.56 in a
    box of +-1, and the culled form -- whose row lists hold inside the box -- and the full form parted.  solve_qp now verifies its result against
    every row.  The world that showed it, and a sweep of batches: every k_opt inside the box, the three forms equal."""
    from armour_amd import _lib
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch
    nlp = ArmourNLP(T=100)
    cases = [(16, 3)] + [(s, O) for s in range(1, 25, 3) for O in (3, 20)]
    for seed, O in cases:
        B = 32
        bp = random_batch(9000 + 31 * seed + O, B, O)
        if seed % 3 == 1:
            bp["q_des"] = bp["q0"] + 0.05 * (bp["q_des"] - bp["q0"])   # goals near the start: part of the problems feasible
        nlp.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
        nlp.set_option(_lib.OPT_SOLVE_CULL, 1); cul = nlp.solve(device_qp=True)
        nlp.set_option(_lib.OPT_SOLVE_CULL, 0); full = nlp.solve(device_qp=True)
        host = nlp.solve(host_qp=True)
        for b, (c, f, h) in enumerate(zip(cul, full, host)):
            assert np.all(np.abs(c["k_opt"]) <= 1.0 + 1e-6), (seed, O, b, c["k_opt"])
            assert _same_solution(c, f) and _same_solution(c, h), (seed, O, b, c, f, h)
    nlp.close()


def test_culled_device_solve_in_armtd_mode():
    from armour_amd import _lib
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_problem, synthetic_offline_jrs
    T = 100
    p = random_problem(4, 8)
    jrs, kr = synthetic_offline_jrs(p["qd0"], T=T)
    nlp = ArmourNLP(T=T).set_option(_lib.OPT_SOLVE_CULL, 1).set_parameters_armtd(p["q0"], p["qd0"], p["q_des"], jrs, kr, p["obstacles"])
    a, c = nlp.solve(host_qp=True)[0], nlp.solve(device_qp=True)[0]
    assert _same_solution(a, c), (a, c)


def test_device_resident_solve_in_armtd_mode():
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_problem, synthetic_offline_jrs
    T = 100
    p = random_problem(4, 8)
    jrs, kr = synthetic_offline_jrs(p["qd0"], T=T)
    nlp = ArmourNLP(T=T).set_parameters_armtd(p["q0"], p["qd0"], p["q_des"], jrs, kr, p["obstacles"])
    a, c = nlp.solve(host_qp=True)[0], nlp.solve(device_qp=True)[0]
    assert _same_solution(a, c), (a, c)


def test_batched_solve_equals_single_solves():
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch
    T, O, B = 20, 4, 3
    bp = random_batch(30, B, O)
    batch = ArmourNLP(T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"]).solve()
    for b in range(B):
        one = ArmourNLP(T=T).set_parameters(bp["q0"][b], bp["qd0"][b], bp["qdd0"][b], bp["q_des"][b], bp["obstacles"][b]).solve()[0]
        assert np.allclose(batch[b]["k_opt"], one["k_opt"], atol=1e-9) and batch[b]["feasible"] == one["feasible"]


@pytest.mark.parametrize("seed,O,T", [(3, 3, 20), (7, 12, 100), (8, 30, 100)])
def test_device_verdict_and_violation_equal_the_host_ones(seed, O, T):
    """armour_solve leaves g / jac on the device and reads back a scan (violation sum, candidate rows, verdict): the
    verdict must be armtd_NLP::finalize_solution's (host armour_check_feasible on the full g) and the violation the L1
    violation of the bounds, for feasible and infeasible outcomes alike; a batch spans several scan segments."""
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch
    B = 3
    bp = random_batch(seed, B, O)
    nlp = ArmourNLP(T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    sols = nlp.solve()
    _, _, gl, gu = nlp.get_bounds_info()
    k = np.stack([s["k_opt"] for s in sols])
    g = nlp.eval_g(k)
    host = nlp.finalize_solution(g)
    for b in range(B):
        assert bool(sols[b]["feasible"]) == bool(host[b])
        v = np.maximum(g[b] - gu[b], 0).sum() + np.maximum(gl[b] - g[b], 0).sum()
        assert abs(sols[b]["max_violation"] - v) <= 1e-9 * max(1.0, v)


def test_row_buffer_overflow_falls_back_to_whole_linearisation(monkeypatch):
    """If a scan segment has more candidate rows than its slice of the row buffer holds, armour_solve copies that
    linearisation over whole and selects the rows on the host as it used to: same iterates, same answer."""
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch
    T, O, B = 20, 4, 3
    bp = random_batch(30, B, O)
    ref = ArmourNLP(T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"]).solve()
    from armour_amd import _lib
    small = ArmourNLP(T=T).set_option(_lib.OPT_SOLVE_ROW_CAP, 2).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"]).solve()
    for a, c in zip(ref, small):
        assert np.array_equal(a["k_opt"], c["k_opt"]) and a["feasible"] == c["feasible"] and a["iterations"] == c["iterations"]
        assert abs(a["max_violation"] - c["max_violation"]) <= 1e-12 * max(1.0, a["max_violation"])


def test_wall_time_limit_is_honoured(sample_problem):
    from armour_amd.planner import ArmourNLP
    p = sample_problem
    nlp = ArmourNLP(T=128).set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    sol = nlp.solve(max_wall_time_s=1e-6)[0]
    assert sol["status"] == 5 and sol["iterations"] == 0 and np.all(sol["k_opt"] == 0)   # stopped at the start point
    full = nlp.solve()[0]
    assert full["status"] in (1, 2) and full["cost"] <= sol["cost"] + 1e-12


def test_cli_drop_in_for_armour_main(tmp_path, sample_problem):
    """The armour_main binary over the reference's file protocol (KSI/uarmtd_planner.m:158-219)."""
    from armour_amd import file_protocol as fp
    from armour_amd.planner import ArmourNLP
    exe = os.path.join(ROOT, "armour_amd", "bin", "armour_main")
    assert os.path.exists(exe), "build with make -C armour_amd/csrc"
    p = sample_problem
    fp.write_armour_in(tmp_path / fp.IN_NAME, p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    out = subprocess.run([exe, str(tmp_path), "128"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    k_opt, ms = fp.read_armour_out(tmp_path / "armour.out")
    nlp = ArmourNLP(T=128).set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    sol = nlp.solve()[0]
    assert (k_opt is not None) == sol["feasible"] and ms > 0
    if k_opt is not None:
        assert np.allclose(k_opt, sol["k_opt"], atol=1e-8)
    g_file = np.loadtxt(tmp_path / "armour_constraints.out")
    assert g_file.shape == (nlp.m,) and np.allclose(g_file, nlp.eval_g(sol["k_opt"])[0], rtol=1e-5, atol=1e-12)
    assert np.loadtxt(tmp_path / "armour_joint_position_center.out").shape == (128 * 7, 3)
    assert np.loadtxt(tmp_path / "armour_joint_position_radius.out").shape == (128 * 7 * 3, 6)
    assert np.allclose(np.loadtxt(tmp_path / "armour_control_input_radius.out"), nlp.torque_radius()[0].T, rtol=1e-9)
    # missing input: -1 in armour.out and a non-zero exit code (RT/armour_main.cu:47-52)
    os.remove(tmp_path / fp.IN_NAME)
    bad = subprocess.run([exe, str(tmp_path), "128"], capture_output=True, text=True, timeout=60)
    assert bad.returncode != 0 and open(tmp_path / "armour.out").read().split()[0] == "-1"


def test_resident_planner_serves_both_file_protocols(tmp_path, sample_problem):
    """`armour_main --serve`: the per-iteration executables forward to the resident process and produce the same files
    as their stand-alone runs, without the GPU start-up in every iteration; `--quit` ends it and they fall back."""
    import time
    from armour_amd import file_protocol as fp
    from armour_amd.worlds import synthetic_offline_jrs
    bindir = os.path.join(ROOT, "armour_amd", "bin")
    exe, exe2 = os.path.join(bindir, "armour_main"), os.path.join(bindir, "armtd_main")
    p = sample_problem
    fp.write_armour_in(tmp_path / fp.IN_NAME, p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    jrs, kr = synthetic_offline_jrs(p["qd0"], 100)
    fp.write_armtd_in(tmp_path / fp.ARMTD_IN_NAME, p["q0"], p["qd0"], p["q_des"], jrs, kr, p["obstacles"])
    # stand-alone runs first
    assert subprocess.run([exe, str(tmp_path), "128"], capture_output=True, timeout=300).returncode == 0
    assert subprocess.run([exe2, str(tmp_path)], capture_output=True, timeout=300).returncode == 0
    names = ["armour.out", "armour_constraints.out", "armour_joint_position_center.out", "armour_joint_position_radius.out", "armour_control_input_radius.out",
             "armtd.out", "armtd_constraints.out", "armtd_joint_position_center.out", "armtd_joint_position_radius.out"]
    alone = {nm: open(tmp_path / nm).read().split() for nm in names}
    with fp.ResidentPlanner(tmp_path, 128, 100) as rp:
        for rep in range(3):
            for nm in names:
                os.remove(tmp_path / nm)
            t0 = time.perf_counter()
            assert subprocess.run([exe, str(tmp_path), "128"], capture_output=True, timeout=120).returncode == 0
            wall = time.perf_counter() - t0
            assert subprocess.run([exe2, str(tmp_path)], capture_output=True, timeout=120).returncode == 0
            for nm in names:
                got = open(tmp_path / nm).read().split()
                if nm in ("armour.out", "armtd.out"):     # last entry is the time in ms
                    assert got[:-1] == alone[nm][:-1] and 0 < float(got[-1]) < 100.0, (nm, got[-1])
                else:
                    assert got == alone[nm], nm
        assert wall < 1.0   # process start + forward + a ~3 ms iteration; the stand-alone run needs the GPU start-up
        # error conventions through the resident planner: missing input -> -1 and a non-zero exit code
        os.remove(tmp_path / fp.IN_NAME)
        bad = subprocess.run([exe, str(tmp_path), "128"], capture_output=True, timeout=60)
        assert bad.returncode != 0 and open(tmp_path / "armour.out").read().split()[0] == "-1"
        assert rp.proc.poll() is None
    assert rp.proc.returncode == 0 and not os.path.exists(tmp_path / "armour.sock")
    fp.write_armour_in(tmp_path / fp.IN_NAME, p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    assert subprocess.run([exe, str(tmp_path), "128"], capture_output=True, timeout=300).returncode == 0   # no resident planner: runs itself
    assert open(tmp_path / "armour.out").read().split()[:-1] == alone["armour.out"][:-1]


def test_large_batches_solve_in_sub_batches_on_the_device():
    """at B = 128, O = 50 a single persistent launch left every problem 4 blocks of 159 tiles (50 ms) and the
    solve fell back to the host-driven form (30 ms).  The batch is now cut into sub-batches with ~24 tiles per block; same iterates
    as the host form for every problem (a sampled third compared here), and the device form is the faster one."""
    import time
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch
    T, O, B = 100, 50, 128
    bp = random_batch(5000, B, O)
    nlp = ArmourNLP(T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    nlp.solve(); nlp.solve(host_qp=True)          # (code objects, buffers)
    import gc
    gc.collect(); gc.disable()                    # (a collector pause of this interpreter is tens of ms: not the library's)
    try:
        t_dev = t_host = 1e9
        for _ in range(2):
            t0 = time.perf_counter(); dev = nlp.solve(); t_dev = min(t_dev, time.perf_counter() - t0)
            t0 = time.perf_counter(); host = nlp.solve(host_qp=True); t_host = min(t_host, time.perf_counter() - t0)
    finally:
        gc.enable()
    for b in range(0, B, 3):
        assert _same_solution(host[b], dev[b]), (b, host[b], dev[b])
    print(f"armour_solve B={B} O={O}: device form {t_dev * 1e3:.1f} ms, host form {t_host * 1e3:.1f} ms")
    assert t_dev <= 1.1 * t_host, (t_dev, t_host)
    nlp.close()
