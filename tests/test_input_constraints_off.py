"""TURN_OFF_INPUT_CONSTRAINTS (RT/Parameters.h:46-47): the ARMOUR (Bezier) trajectory planned without the torque rows.

The reference switches at compile time: the reach-set build stops after the forward kinematics (RT/armour_main.cu:115,149-165), the
torque radius stays zero (:172-175) and its file is not written (:355); m = J T O + 4n with the collision rows FIRST
(RT/NLPclass.cu:46-54,289-301,361-373), bounds and finalize_solution without the torque block (:117,453).  Here it is
ArmourParams.input_constraints_off (include/armour_types.h).
"""
import os
import subprocess

import numpy as np
import pytest

from helpers import PZ_TESTS_K, SAMPLE_PROBLEM

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_pair(T, p):
    from oracle.cpu_oracle import Oracle, default_params
    full = Oracle(T=T).set_problem(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    pr = default_params(T)
    pr.input_constraints_off = 1
    off = Oracle(params=pr).set_problem(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    return full, off


def test_oracle_rows_are_the_full_problems_rows_without_the_torque_block():
    """CPU: with the switch on the restatement returns exactly the collision and limit rows of the full problem -- the reach sets the rows are
    sliced from (JRS, forward kinematics, half-space tables) do not depend on it -- in the order RT/NLPclass.cu:289-301 writes them, the
    bounds of RT/NLPclass.cu:129-164 behind offset 0, a zero torque radius, and the same cost."""
    T = 16
    p = SAMPLE_PROBLEM
    full, off = _oracle_pair(T, p)
    n, J, O = full.n, full.J, full.O
    assert off.m == J * T * O + 4 * n and full.m == off.m + n * T
    for k in (np.zeros(7), PZ_TESTS_K):
        g, jac = full.eval_g_jac(k)
        g2, jac2 = off.eval_g_jac(k)
        assert np.array_equal(g2, g[n * T:]) and np.array_equal(jac2, jac[n * T:])
        assert off.eval_f(k) == full.eval_f(k) and np.array_equal(off.eval_grad_f(k), full.eval_grad_f(k))
    xl, xu, gl, gu = full.bounds()
    xl2, xu2, gl2, gu2 = off.bounds()
    assert np.array_equal(gl2, gl[n * T:]) and np.array_equal(gu2, gu[n * T:]) and np.array_equal(xl, xl2) and np.array_equal(xu, xu2)
    assert not off.torque_radius().any()
    assert np.array_equal(off.link_generators(), full.link_generators())


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,O", [(1, 100, 10), (3, 100, 4), (40, 100, 3)], ids=["sample problem", "random batch, per-step kernel", "random batch, time-vectorised kernel"])
def test_device_rows_bounds_and_solution_against_the_oracle(B, T, O):
    """GPU: g / jac, bounds, feasibility verdict and the solver's optimum with the switch on against the oracle twin (tolerances of
    tests/test_p1_parity.py); the torque tables are empty, the radius zero, and the reach-set build ran the forward kinematics only."""
    from armour_amd.planner import ArmourNLP, default_params
    from armour_amd.worlds import random_batch, random_k
    from oracle.cpu_oracle import Oracle
    from oracle.cpu_oracle import default_params as oracle_params
    if B == 1:
        bp = {k: np.asarray(v, dtype=float)[None] for k, v in SAMPLE_PROBLEM.items()}
    else:
        bp = random_batch(321, B, O)
    pr, po = default_params(T), oracle_params(T)
    pr.input_constraints_off = 1; po.input_constraints_off = 1
    nlp = ArmourNLP(params=pr).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    n, J = nlp.n, nlp.J
    O_ = bp["obstacles"].reshape(B, -1, 12).shape[1]
    assert nlp.m == J * T * O_ + 4 * n
    full = ArmourNLP(T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    for x in (nlp, full):   # (the second build of each: code objects loaded)
        x.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    assert nlp.build_ms < full.build_ms   # (no RNEA)
    ks = random_k(9, B)
    g, jac = nlp.eval_g_jac(ks)
    gf, jf = full.eval_g_jac(ks)
    # the same device rows as the full problem's (the link tables do not depend on the switch: same kernels, same operands)
    assert np.array_equal(g, gf[:, n * T:]) and np.array_equal(jac, jf[:, n * T:])
    xl, xu, gl, gu = nlp.get_bounds_info()
    for b in sorted({0, B - 1}):
        o = Oracle(params=po).set_problem(bp["q0"][b], bp["qd0"][b], bp["qdd0"][b], bp["q_des"][b], bp["obstacles"][b])
        g_ref, jac_ref = o.eval_g_jac(ks[b])
        assert np.abs(g[b] - g_ref).max() <= 1e-9 and np.abs(jac[b] - jac_ref).max() <= 1e-8
        _, _, gl_ref, gu_ref = o.bounds()
        assert np.array_equal(gl[b], gl_ref) and np.array_equal(gu[b], gu_ref)
        assert abs(nlp.eval_f(ks)[b] - o.eval_f(ks[b])) <= 1e-12
    assert not nlp.torque_radius().any()
    c, ind, keys, co = nlp.pz("torque", 0, 0)
    assert len(keys) == 0
    # verdict of the row test (finalize_solution without the torque block) against numpy on the full rows
    feas = nlp.finalize_solution(g)
    viol = nlp.eval_violations(ks)
    is_col = np.arange(nlp.m) < J * T * O_
    for b in range(B):
        inside = np.where(is_col, g[b] <= 1e-4, (g[b] >= gl[b]) & (g[b] <= gu[b]))   # (RT/NLPclass.cu:472-536: collision rows with the 1e-4 m slack)
        assert bool(feas[b]) == bool(inside.all()) == bool(viol[b]["feasible"])
    # the solver: same optimum as scipy on the oracle's callbacks would need minutes at T = 100; here: both solver forms agree, and a
    # feasible result passes the oracle's own rows
    sols = nlp.solve()
    sols_host = nlp.solve(host_qp=True)
    for b in range(B):
        assert np.array_equal(sols[b]["k_opt"], sols_host[b]["k_opt"]) and sols[b]["feasible"] == sols_host[b]["feasible"]
    b = 0
    if sols[b]["feasible"]:
        o = Oracle(params=po).set_problem(bp["q0"][b], bp["qd0"][b], bp["qdd0"][b], bp["q_des"][b], bp["obstacles"][b])
        g_ref, _ = o.eval_g_jac(sols[b]["k_opt"])
        _, _, gl_ref, gu_ref = o.bounds()
        assert (g_ref <= gu_ref + 1e-4).all() and (g_ref[-4 * n:] >= gl_ref[-4 * n:]).all()
    nlp.close(); full.close()


@pytest.mark.gpu
def test_cli_honours_the_switch(tmp_path, sample_problem):
    """armour_main --no-input-constraints (or ARMOUR_TURN_OFF_INPUT_CONSTRAINTS=1 in the environment, for a caller that cannot change the
    command line): m rows without the torque block in armour_constraints.out, no armour_control_input_radius.out (RT/armour_main.cu:355)."""
    from armour_amd import file_protocol as fp
    exe = os.path.join(ROOT, "armour_amd", "bin", "armour_main")
    p = sample_problem
    T, n, J, O = 100, 7, 7, len(p["obstacles"])
    for how in ("flag", "env"):
        d = tmp_path / how
        d.mkdir()
        fp.write_armour_in(d / fp.IN_NAME, p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
        env = dict(os.environ)
        cmd = [exe, str(d), str(T)]
        if how == "flag":
            cmd.append("--no-input-constraints")
        else:
            env["ARMOUR_TURN_OFF_INPUT_CONSTRAINTS"] = "1"
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
        assert out.returncode == 0, out.stdout + out.stderr
        g_file = np.loadtxt(d / "armour_constraints.out")
        assert g_file.shape == (J * T * O + 4 * n,)
        assert not os.path.exists(d / "armour_control_input_radius.out")
        k_opt, ms = fp.read_armour_out(d / "armour.out")
        assert ms > 0
