"""ARMTD comparison mode: the reference's second planner, kinova_planner_realtime_armtd_comparison
("CMP"): constant-acceleration trajectory, offline cos/sin JRS tables, forward kinematics, no torque rows.

CPU tests pin the oracle's restatement of CMP/Trajectory.cu against properties of the curve itself; the GPU tests compare
armour_set_problems_armtd + the fused eval kernel with the oracle through the C ABI at the tolerances of
test_p1_parity.py.  PARITY UNPINNED: the reference ships neither golden vectors nor its offline JRS tables (the .mat
files of CMP/offline_jrs are not in the checkout) and cannot be built here, so the tables are the closed-form enclosure
of armour_amd.worlds.synthetic_offline_jrs."""
import os
import subprocess

import numpy as np
import pytest

from helpers import PZ_TESTS_K, SAMPLE_PROBLEM, load_golden

C_TOL, G_TOL, J_TOL = 1e-11, 1e-9, 1e-8
T = 100


def _problem(seed, O):
    from armour_amd.worlds import random_problem, synthetic_offline_jrs
    p = random_problem(seed, O)
    jrs, kr = synthetic_offline_jrs(p["qd0"], T)
    return dict(q0=p["q0"], qd0=p["qd0"], q_des=p["q_des"], jrs=jrs, k_range=kr, obstacles=p["obstacles"])


def _oracle(p):
    from oracle.cpu_oracle import Oracle
    return Oracle(T=T).set_problem_armtd(p["q0"], p["qd0"], p["q_des"], p["jrs"], p["k_range"], p["obstacles"])


# ---------------------------------------------------------------------------------------------- CPU: the oracle itself
def _curve(q0, qd0, ka, t):
    """CMP/Trajectory.h:6-15 evaluated pointwise: constant acceleration to t = 0.5, constant deceleration to rest at 1."""
    qp, vp = q0 + qd0 * 0.5 + 0.125 * ka, qd0 + 0.5 * ka
    s = t - 0.5
    q = np.where(t <= 0.5, q0 + qd0 * t + 0.5 * ka * t * t, qp + vp * s - vp * s * s)
    v = np.where(t <= 0.5, qd0 + ka * t, vp - 2 * vp * s)
    return q, v


def test_oracle_synthetic_tables_enclose_the_curve():
    """The stand-in JRS tables really contain cos/sin of the trajectory for every k and every velocity of the bin."""
    from armour_amd.worlds import synthetic_offline_jrs
    rng = np.random.default_rng(5)
    qd0 = rng.uniform(-1.5, 1.5, 7)
    jrs, kr = synthetic_offline_jrs(qd0, T)
    c_kvi = np.linspace(-np.pi, np.pi, 401)
    for i in range(7):
        vc = c_kvi[np.argmin(np.abs(qd0[i] - c_kvi))]
        k = rng.uniform(-1, 1, 4000)
        v = vc + rng.uniform(-1, 1, 4000) * (c_kvi[1] - c_kvi[0]) / 2
        j = rng.integers(0, T, 4000)
        t = (j + rng.uniform(0, 1, 4000)) / T
        th = _curve(0.0, v, kr[i] * k, t)[0]
        assert np.all(np.abs(np.cos(th) - (jrs[i, 0, j] + jrs[i, 1, j] * k)) <= jrs[i, 2, j] + 1e-12)
        assert np.all(np.abs(np.sin(th) - (jrs[i, 3, j] + jrs[i, 4, j] * k)) <= jrs[i, 5, j] + 1e-12)


def test_oracle_state_limit_rows_are_the_extrema_of_the_curve():
    """Rows q_min, q_max, qd_min, qd_max (CMP/Trajectory.cu:83-205) against a dense sampling of the curve."""
    rng = np.random.default_rng(11)
    t = np.linspace(0, 1, 20001)
    for trial in range(12):
        p = _problem(100 + trial, 0)
        if trial % 3 == 0:
            p["qd0"] = p["qd0"] * 0.05   # turning point inside (0, 0.5) for small initial velocities
            from armour_amd.worlds import synthetic_offline_jrs
            p["jrs"], p["k_range"] = synthetic_offline_jrs(p["qd0"], T)
        o = _oracle(p)
        assert o.m == 28
        k = rng.uniform(-1, 1, 7)
        g, jac = o.eval_g_jac(k)
        for i in range(7):
            q, v = _curve(p["q0"][i], p["qd0"][i], p["k_range"][i] * k[i], t)
            ext = [q.min(), q.max(), v.min(), v.max()]
            for r in range(4):
                # the sampled extremum can miss the exact turning point by O(dt^2)
                assert abs(g[r * 7 + i] - ext[r]) <= 1e-7, (trial, i, r)
        off = jac.copy()
        for r in range(4):
            off[r * 7 + np.arange(7), np.arange(7)] = 0
        assert not off.any()   # one entry per row (the reference's memset of :208 clears too little; zero is the intent)


def test_oracle_state_limit_gradient_keeps_the_reference_scaling():
    """CMP/Trajectory.cu:347-381 stores d/d(k_range*k) on the Jacobian diagonal, without the k_range factor: a central
    difference in k equals k_range times the stored value (a reference behaviour kept on purpose)."""
    p = _problem(7, 0)
    o = _oracle(p)
    rng = np.random.default_rng(3)
    k = rng.uniform(-0.9, 0.9, 7)
    jac = o.eval_g_jac(k)[1]
    eps = 1e-6
    for i in range(7):
        kp, km = k.copy(), k.copy()
        kp[i] += eps
        km[i] -= eps
        num = (o.eval_g_jac(kp)[0] - o.eval_g_jac(km)[0]) / (2 * eps)
        for r in range(4):
            assert abs(num[r * 7 + i] - p["k_range"][i] * jac[r * 7 + i, i]) <= 1e-6


def test_oracle_cost_and_bounds():
    """CMP/NLPclass.cu:183-243 (cost at the plan point q0 + qd0/2 + k_range k/8) and :73-140 (bounds)."""
    from oracle.cpu_oracle import Oracle
    p = _problem(9, 4)
    o = _oracle(p)
    assert o.m == 7 * T * 4 + 28
    k = np.linspace(-0.8, 0.8, 7)
    qp = p["q0"] + 0.5 * p["qd0"] + 0.125 * p["k_range"] * k
    e = p["q_des"] - qp
    e[::2] = (e[::2] + np.pi) % (2 * np.pi) - np.pi   # joints 1,3,5,7 are continuous
    assert abs(o.eval_f(k) - 10.0 * np.sum(e * e)) <= 1e-12
    eps = 1e-6
    num = np.array([(o.eval_f(k + eps * np.eye(7)[i]) - o.eval_f(k - eps * np.eye(7)[i])) / (2 * eps) for i in range(7)])
    assert np.abs(num - o.eval_grad_f(k)).max() <= 1e-6
    xl, xu, gl, gu = o.bounds()
    arm = Oracle(T=T).set_problem(p["q0"], p["qd0"], np.zeros(7), p["q_des"], p["obstacles"])
    agl, agu = arm.bounds()[2:]
    assert np.array_equal(gl, agl[7 * T:]) and np.array_equal(gu, agu[7 * T:])   # ARMOUR's rows minus the torque block


def _check_against_fixture(gd, obj, tab_tol, g_tol, j_tol):
    """obj: the oracle or the device planner, both built from the fixture's inputs."""
    is_dev = hasattr(obj, "get_bounds_info")
    first = (lambda a: a[0]) if is_dev else (lambda a: a)
    assert np.abs(first(obj.link_generators()) - gd["link_gens"]).max() <= tab_tol
    lc, lk = [], []
    for i in range(7):
        for t in range(T):
            keys = obj.pz("link", i, t)[2]
            lc.append(len(keys)); lk.append(keys)
    assert np.array_equal(np.array(lc), gd["link_count"]) and np.array_equal(np.concatenate(lk), gd["link_keys"])
    b = obj.get_bounds_info() if is_dev else obj.bounds()
    assert np.array_equal(first(b[2]), gd["g_l"]) and np.array_equal(first(b[3]), gd["g_u"])
    for tag, k in (("k0", np.zeros(7)), ("kt", PZ_TESTS_K)):
        g, jac = obj.eval_g_jac(k)
        assert np.abs(first(g) - gd[f"g_{tag}"]).max() <= g_tol
        assert np.abs(first(jac)[gd["jac_rows"]] - gd[f"jac_{tag}"]).max() <= j_tol
        assert abs(first(np.atleast_1d(obj.eval_f(k))) - float(gd[f"f_{tag}"])) <= 1e-12
        assert np.abs(first(np.atleast_2d(obj.eval_grad_f(k))) - gd[f"gradf_{tag}"]).max() <= 1e-12


def test_oracle_reproduces_armtd_fixture():
    """tests/golden/armtd_sample_T100.npz (make_golden.py): the sample problem of CMP/armtd_main.cu:17-33."""
    gd = load_golden("armtd_sample_T100")
    o = _oracle(gd)
    assert o.m == 7 * T * 10 + 28
    assert abs(o.min_margin() - float(gd["min_margin"])) <= 1e-6 * float(gd["min_margin"])
    _check_against_fixture(gd, o, 1e-13, 1e-12, 1e-12)


def test_oracle_armtd_without_obstacles_and_with_the_gripper_link():
    """Edge cases of the row layout: O = 0 leaves the 4n state-limit rows; the 8-link arm (fixed gripper joint last,
    CMP/KinovaInfo.h) has J = 8 link blocks and still n = 7 factors."""
    from oracle.cpu_oracle import Oracle, kinova_gripper_robot
    p = _problem(3, 0)
    o = _oracle(p)
    assert o.m == 28
    g0 = o.eval_g_jac(np.zeros(7))[0]
    p5 = _problem(3, 5)
    o5 = _oracle(p5)
    assert np.array_equal(o5.eval_g_jac(np.zeros(7))[0][-28:], g0)
    og = Oracle(robot=kinova_gripper_robot(), T=T).set_problem_armtd(p5["q0"], p5["qd0"], p5["q_des"], p5["jrs"], p5["k_range"], p5["obstacles"])
    assert og.m == 8 * T * 5 + 28
    gg = og.eval_g_jac(np.zeros(7))[0]
    assert np.array_equal(gg[-28:], g0) and np.array_equal(gg[:6 * T * 5], o5.eval_g_jac(np.zeros(7))[0][:6 * T * 5])   # links before the gripper are unchanged


# ---------------------------------------------------------------------------------------------- GPU: the HIP path
def _nlp(ps):
    from armour_amd.planner import ArmourNLP
    stack = lambda key: np.stack([p[key] for p in ps])
    return ArmourNLP(T=T).set_parameters_armtd(stack("q0"), stack("qd0"), stack("q_des"), stack("jrs"), stack("k_range"), stack("obstacles"))


@pytest.mark.gpu
def test_tables_and_eval_against_oracle():
    from armour_amd.worlds import random_k
    ps = [_problem(40 + b, 8) for b in range(3)]
    ps[1]["qd0"] = ps[1]["qd0"] * 0.05
    from armour_amd.worlds import synthetic_offline_jrs
    ps[1]["jrs"], ps[1]["k_range"] = synthetic_offline_jrs(ps[1]["qd0"], T)
    nlp = _nlp(ps)
    oracles = [_oracle(p) for p in ps]
    assert nlp.m == oracles[0].m == 7 * T * 8 + 28 and nlp.get_nlp_info() == (7, nlp.m, 7 * nlp.m)
    for b, o in enumerate(oracles):
        assert o.min_margin() > 1e-9
        for i in range(7):
            for t in range(T):
                c, ind, keys, co = o.pz("link", i, t)
                c2, ind2, keys2, co2 = nlp.pz("link", i, t, b=b)
                assert np.array_equal(keys, keys2), (b, i, t)
                if len(keys):
                    assert np.abs(co - co2).max() <= C_TOL
                assert np.abs(c - c2).max() <= C_TOL and np.abs(ind - ind2).max() <= C_TOL
                assert len(nlp.pz("torque", i, t, b=b)[2]) == 0
        assert np.abs(nlp.link_generators()[b] - o.link_generators()).max() <= C_TOL
    A2, d2, dl2 = nlp.hyperplanes()
    for b, o in enumerate(oracles):
        A, d, dl = o.hyperplanes()
        assert np.abs(A - A2[b]).max() <= C_TOL and np.abs(d - d2[b]).max() <= C_TOL and np.abs(dl - dl2[b]).max() <= C_TOL
    xl, xu, gl, gu = nlp.get_bounds_info()
    for trial in range(3):
        ks = random_k(70 + trial, 3) if trial else np.zeros((3, 7))
        g, jac = nlp.eval_g_jac(ks)
        assert np.array_equal(nlp.eval_g(ks), g) and np.array_equal(nlp.eval_jac_g(ks), jac)
        f, gf = nlp.eval_f(ks), nlp.eval_grad_f(ks)
        for b, o in enumerate(oracles):
            gr, jr = o.eval_g_jac(ks[b])
            assert np.abs(g[b] - gr).max() <= G_TOL and np.abs(jac[b] - jr).max() <= J_TOL
            assert np.array_equal(g[b][-28:], gr[-28:]) and np.array_equal(jac[b][-28:], jr[-28:])   # closed-form rows: bit-exact
            assert abs(f[b] - o.eval_f(ks[b])) <= 1e-12 and np.abs(gf[b] - o.eval_grad_f(ks[b])).max() <= 1e-12
            ob = o.bounds()
            assert np.array_equal(gl[b], ob[2]) and np.array_equal(gu[b], ob[3])


@pytest.mark.gpu
def test_device_against_armtd_fixture():
    gd = load_golden("armtd_sample_T100")
    assert float(gd["min_margin"]) > 1e-9
    _check_against_fixture(gd, _nlp([gd]), C_TOL, G_TOL, J_TOL)


@pytest.mark.gpu
def test_device_without_obstacles_and_with_the_gripper_link():
    from armour_amd.planner import ArmourNLP, kinova_gripper_robot
    from oracle.cpu_oracle import Oracle
    from oracle.cpu_oracle import kinova_gripper_robot as oracle_gripper
    p = _problem(3, 0)
    nlp = _nlp([p])
    assert nlp.m == 28
    g, jac = nlp.eval_g_jac(np.full(7, 0.3))
    gr, jr = _oracle(p).eval_g_jac(np.full(7, 0.3))
    assert np.array_equal(g[0], gr) and np.array_equal(jac[0], jr)
    sol = nlp.solve()[0]
    assert sol["feasible"] == bool(nlp.finalize_solution(nlp.eval_g(sol["k_opt"]))[0])
    p5 = _problem(3, 5)
    dev = ArmourNLP(robot=kinova_gripper_robot(), T=T).set_parameters_armtd(p5["q0"], p5["qd0"], p5["q_des"], p5["jrs"], p5["k_range"], p5["obstacles"])
    og = Oracle(robot=oracle_gripper(), T=T).set_problem_armtd(p5["q0"], p5["qd0"], p5["q_des"], p5["jrs"], p5["k_range"], p5["obstacles"])
    assert dev.m == og.m == 8 * T * 5 + 28
    k = np.linspace(-0.6, 0.6, 7)
    g, jac = dev.eval_g_jac(k)
    gr, jr = og.eval_g_jac(k)
    assert np.abs(g[0] - gr).max() <= G_TOL and np.abs(jac[0] - jr).max() <= J_TOL
    # the re-check looks at links 0..n-2 = 0..5 of the 8 (CMP/NLPclass.cu:391)
    ok = g.copy()
    ok[0, :8 * T * 5] = -1.0
    bad = ok.copy()
    bad[0, 6 * T * 5] = 1.0
    assert dev.finalize_solution(ok)[0] and dev.finalize_solution(bad)[0]
    bad[0, 6 * T * 5 - 1] = 1.0
    assert not dev.finalize_solution(bad)[0]


@pytest.mark.gpu
def test_batch_multi_point_and_pinned_paths_are_bit_identical():
    import torch
    from armour_amd.worlds import random_k
    ps = [_problem(60 + b, 5) for b in range(9)]   # B >= 8: the d-from-centre kernels
    nlp = _nlp(ps)
    ks = random_k(5, 9)
    g, jac = nlp.eval_g_jac(ks)
    gp, jp = nlp.eval_g_jac(ks, pinned=True)
    assert np.array_equal(g, gp) and np.array_equal(jac, jp)
    for b in (0, 4, 8):
        one = _nlp([ps[b]])
        g1, j1 = one.eval_g_jac(ks[b])
        assert np.array_equal(g[b], g1[0]) and np.array_equal(jac[b], j1[0])
        assert np.array_equal(nlp.link_generators()[b], one.link_generators()[0])
    one = _nlp([ps[2]])
    pts = random_k(6, 4)
    dk = torch.tensor(pts, device="cuda")
    dg = torch.zeros((4, one.m), dtype=torch.float64, device="cuda")
    dj = torch.zeros((4, one.m, 7), dtype=torch.float64, device="cuda")
    one.eval_g_jac_device_multi(dk.data_ptr(), 4, dg.data_ptr(), dj.data_ptr())
    torch.cuda.synchronize()
    for s in range(4):
        g1, j1 = one.eval_g_jac(pts[s])
        assert np.array_equal(dg[s].cpu().numpy(), g1[0]) and np.array_equal(dj[s].cpu().numpy(), j1[0])


@pytest.mark.gpu
def test_feasibility_recheck_follows_the_comparison_planner():
    """CMP/NLPclass.cu:391-402 re-checks the collision rows of links 0..n-2 only; limit rows as RT."""
    p = _problem(21, 3)
    nlp = _nlp([p])
    g = nlp.eval_g_jac(np.zeros(7))[0]
    Q = 7 * T * 3
    ok = g.copy()
    ok[0, :Q] = -1.0
    assert nlp.finalize_solution(ok)[0] == 1
    bad = ok.copy()
    bad[0, 6 * T * 3 + 5] = 1.0          # a row of the last link: not looked at
    assert nlp.finalize_solution(bad)[0] == 1
    bad = ok.copy()
    bad[0, 6 * T * 3 - 1] = 1.0          # last row of link 5: violation
    assert nlp.finalize_solution(bad)[0] == 0
    bad = ok.copy()
    bad[0, Q + 2 * 7] = 100.0            # qd_min of joint 0 above the speed limit
    assert nlp.finalize_solution(bad)[0] == 0


@pytest.mark.gpu
def test_solve_in_comparison_mode():
    """armour_solve on the comparison planner's NLP: the verdict equals the host re-check on the oracle's g at the
    returned k, and a feasible result does not cost more than the feasible start k = 0."""
    ps = [dict(q0=SAMPLE_PROBLEM["q0"], qd0=SAMPLE_PROBLEM["qd0"], q_des=SAMPLE_PROBLEM["q_des"], obstacles=SAMPLE_PROBLEM["obstacles"])]
    from armour_amd.worlds import synthetic_offline_jrs
    ps[0]["jrs"], ps[0]["k_range"] = synthetic_offline_jrs(ps[0]["qd0"], T)
    ps += [_problem(80 + b, 6) for b in range(3)]
    ps[0]["obstacles"] = ps[0]["obstacles"][:6]
    nlp = _nlp(ps)
    res = nlp.solve()
    k_opt = np.stack([r["k_opt"] for r in res])
    feasible = np.array([r["feasible"] for r in res])
    g = nlp.eval_g_jac(k_opt)[0]
    assert np.array_equal(feasible, nlp.finalize_solution(g))
    f0, f1 = nlp.eval_f(np.zeros((4, 7))), nlp.eval_f(k_opt)
    start_ok = nlp.finalize_solution(nlp.eval_g_jac(np.zeros((4, 7)))[0])
    for b in range(4):
        o = _oracle(ps[b])
        assert np.abs(o.eval_g_jac(k_opt[b])[0] - g[b]).max() <= G_TOL
        if feasible[b] and start_ok[b]:
            assert f1[b] <= f0[b] + 1e-9
    assert feasible.any()


@pytest.mark.gpu
def test_cli_drop_in_for_armtd_main(tmp_path):
    """The armtd_main binary over the comparison planner's file protocol (KSI/uarmtd_planner.m:257-345).  The tables
    pass through "%.10f" text, so the in-process reference below is built from the parsed-back values."""
    from armour_amd import file_protocol as fp
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "armour_amd", "bin", "armtd_main")
    assert os.path.exists(exe), "build with make -C armour_amd/csrc"
    p = _problem(33, 7)
    fp.write_armtd_in(tmp_path / fp.ARMTD_IN_NAME, p["q0"], p["qd0"], p["q_des"], p["jrs"], p["k_range"], p["obstacles"])
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    tok = np.array(open(tmp_path / fp.ARMTD_IN_NAME).read().split(), dtype=np.float64)
    q = dict(q0=tok[0:7], qd0=tok[7:14], q_des=tok[14:21])
    body = tok[21:21 + 7 * (6 * T + 1)].reshape(7, 6 * T + 1)
    q["jrs"], q["k_range"] = body[:, :6 * T].reshape(7, 6, T), body[:, -1]
    nobs = int(tok[21 + 7 * (6 * T + 1)])
    q["obstacles"] = tok[22 + 7 * (6 * T + 1):].reshape(nobs, 12)
    assert nobs == 7
    nlp = _nlp([q])
    lines = open(tmp_path / "armtd.out").read().split()
    sol = nlp.solve(tolerance=1e-7, max_wall_time_s=0.4)[0]
    assert (len(lines) == 8) == sol["feasible"] and float(lines[-1]) > 0
    k_used = sol["k_opt"]
    if len(lines) == 8:
        assert np.allclose(np.array(lines[:7], dtype=np.float64), sol["k_opt"], atol=1e-8)
    g_file = np.loadtxt(tmp_path / "armtd_constraints.out")
    assert g_file.shape == (nlp.m,) and np.allclose(g_file, nlp.eval_g(k_used)[0], rtol=1e-5, atol=1e-12)
    assert np.allclose(np.loadtxt(tmp_path / "armtd_joint_position_center.out").reshape(T, 7, 3), nlp.link_centers(k_used)[0], rtol=1e-9, atol=1e-12)
    assert np.allclose(np.loadtxt(tmp_path / "armtd_joint_position_radius.out").reshape(T, 7, 3, 6), nlp.link_generators()[0], rtol=1e-9, atol=1e-12)
    os.remove(tmp_path / fp.ARMTD_IN_NAME)   # missing input: -1 and a non-zero exit code (CMP/armtd_main.cu:57-62)
    bad = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=60)
    assert bad.returncode != 0 and open(tmp_path / "armtd.out").read().split()[0] == "-1"
