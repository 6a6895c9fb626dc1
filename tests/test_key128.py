"""An 8-factor arm through the whole path with 128-bit monomial keys (BASELINE configs[4] says "8-DOF").

The reference packs 9 bits per factor into a u64 (RT/PZsparse.h:23-40): 7 factors fill it, and neither its Kinova nor its Fetch header has an
eighth -- it cannot represent the arm at all.  The library's second ABI (-DARMOUR_KEY128: libarmour_hip_k128.so, oracle/liboracle_k128.so;
include/armour_types.h) keeps the packing and the plain integer addition of keys and widens the key to 128 bits: ARMOUR_MAX_FACTORS = 8.
Here: the Kinova Gen3 with one more revolute joint behind the wrist (a synthetic arm -- eight actuated joints, eight trajectory parameters,
72 key bits: cos / sin error fields of joints 1..7 lie beyond bit 63) at T = 100, O = 20,
  * reach sets by BOTH build kernels (step by step, time-vectorised) against the CPU restatement built with the same key type: identical
    monomial key sets of every link / torque table, coefficients and centres <= 1e-12, radii / torque radius / link generators <= 1e-10,
  * the fused evaluation (8 columns per Jacobian row, m = 8 T + 8 T O + 32) against it: |dg| <= 1e-9, |djac| <= 1e-8 (the tolerances of
    tests/test_p1_parity.py),
  * the two kernels against each other: keys, coefficients and centres bit for bit, radii to 1e-12 (the contract of ARMOUR_OPT_P1_BUILD).
The body runs in a process of its own: the ABI (struct layouts of the ctypes mirrors, which library) is chosen per process by ARMOUR_KEY128=1.
The 64-bit build is not touched by any of this (pz_key.h: pzkey_t = uint64_t there) -- its digests against round 3's library are the A/B of
profiles/r04_p1_ab.txt."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_eight_factor_arm(robot):
    """Kinova Gen3 (no gripper; the struct filled by the library's / the oracle's own preset: RT/KinovaWithoutGripperInfo.h) + joint 8: a revolute
    joint about z, 0.12 m further along the last link's axis, carrying a 0.6 kg link.  The same numbers go to the product and to the oracle."""
    J = robot.num_joints
    assert J == 7 and robot.num_factors == 7
    for e in range(3):
        robot.trans[(J + 1) * 3 + e] = robot.trans[J * 3 + e]        # the end frame moves out by one joint
    robot.trans[J * 3 + 0], robot.trans[J * 3 + 1], robot.trans[J * 3 + 2] = 0.0, 0.0, -0.12
    robot.rots[J * 3 + 0], robot.rots[J * 3 + 1], robot.rots[J * 3 + 2] = np.pi / 2, 0.0, 0.0
    robot.axes[J] = 3
    robot.mass[J] = 0.6
    robot.com[J * 3 + 0], robot.com[J * 3 + 1], robot.com[J * 3 + 2] = 0.0, 0.01, -0.03
    for e in range(9):
        robot.inertia[J * 9 + e] = 0.0
    robot.inertia[J * 9 + 0], robot.inertia[J * 9 + 4], robot.inertia[J * 9 + 8] = 6e-4, 6e-4, 4e-4
    robot.friction[J], robot.damping[J], robot.armature[J] = robot.friction[J - 1], robot.damping[J - 1], robot.armature[J - 1]
    robot.continuous[J] = 0
    robot.state_limits_lb[J], robot.state_limits_ub[J] = -2.2, 2.2
    robot.speed_limits[J], robot.torque_limits[J] = 1.2218, 13.0
    for e in range(3):
        robot.link_zonotope_center[J * 3 + e] = robot.link_zonotope_center[(J - 1) * 3 + e]
        robot.link_zonotope_generators[J * 3 + e] = robot.link_zonotope_generators[(J - 1) * 3 + e]
    robot.num_joints, robot.num_factors = J + 1, J + 1
    return robot


def _body():
    """(runs with ARMOUR_KEY128=1)"""
    sys.path.insert(0, ROOT)
    from armour_amd import _lib
    from armour_amd.planner import ArmourNLP, default_params, kinova_robot
    from oracle import cpu_oracle as orc
    assert _lib.MAXF == 8 and orc.MAXF == 8 and _lib.load().armour_abi_max_factors() == 8
    T, O, n = 100, 20, 8
    rng = np.random.default_rng(2024)
    lb = np.array([-np.pi, -2.41, -np.pi, -2.66, -np.pi, -2.23, -np.pi, -2.2]) + 0.3
    q0 = rng.uniform(lb, -lb)
    speed = np.array([1.3963, 1.3963, 1.3963, 1.3963, 1.2218, 1.2218, 1.2218, 1.2218])
    qd0, qdd0 = rng.uniform(-0.5, 0.5, n) * speed, rng.uniform(-1.0, 1.0, n)
    q_des = q0 + rng.uniform(-np.pi / 8, np.pi / 8, n)
    obs = np.zeros((O, 12))
    obs[:, 0:3] = rng.uniform([-0.8, -0.8, 0.05], [0.8, 0.8, 1.2], (O, 3))
    s = rng.uniform(0.01, 0.5, (O, 3))
    obs[:, 3], obs[:, 7], obs[:, 11] = s[:, 0] / 2, s[:, 1] / 2, s[:, 2] / 2
    def with_k_range(p):
        p.k_range[7] = p.k_range[6]   # (the oracle's generated table of RT/Parameters.h's k_range has the reference's seven entries)
        return p
    oracle = orc.Oracle(robot=make_eight_factor_arm(orc.kinova_robot()), params=with_k_range(orc.default_params(T))).set_problem(q0, qd0, qdd0, q_des, obs)
    k = rng.uniform(-1, 1, n)
    g_ref, jac_ref = oracle.eval_g_jac(k)
    m = n * T + 8 * T * O + 4 * n
    assert g_ref.shape == (m,) and jac_ref.shape == (m, n)
    built = {}
    for name, build, B in (("per_step", 1, 1), ("time_vectorised", 2, 3)):
        nlp = ArmourNLP(robot=make_eight_factor_arm(kinova_robot()), params=with_k_range(default_params(T)))
        nlp.set_option(_lib.OPT_P1_BUILD, build)
        rep = lambda a: np.repeat(np.asarray(a)[None], B, axis=0)
        nlp.set_parameters(rep(q0), rep(qd0), rep(qdd0), rep(q_des), rep(obs))
        info = nlp.build_info()
        assert info["kernel"] == name, info
        assert (nlp.n, nlp.J, nlp.m) == (8, 8, m)
        b = B - 1
        assert np.abs(nlp.torque_radius()[b] - oracle.torque_radius()).max() <= 1e-10
        assert np.abs(nlp.link_generators()[b] - oracle.link_generators()).max() <= 1e-10
        tabs = []
        for which, cnt in (("link", 8), ("torque", 8)):
            for i in range(cnt):
                for t in range(0, T, 9):
                    c1, r1, k1, co1 = nlp.pz(which, i, t, b=b)
                    c2, r2, k2, co2 = oracle.pz(which, i, t)
                    assert np.array_equal(k1, k2), (name, which, i, t, len(k1), len(k2))          # identical key sets, in the same order
                    assert np.abs(co1 - co2).max(initial=0.0) <= 1e-12 and np.abs(c1 - c2).max() <= 1e-12
                    assert np.abs(r1 - r2).max() <= 1e-10
                    tabs.append((c1, r1, k1, co1))
        g, jac = nlp.eval_g_jac(rep(k))
        dg, dj = np.abs(g[b] - g_ref).max(), np.abs(jac[b] - jac_ref).max()
        assert dg <= 1e-9 and dj <= 1e-8, (name, dg, dj)
        print(f"{name}: {info}, build {nlp.build_ms:.2f} ms, |dg| {dg:.2e} |djac| {dj:.2e}, table sizes {nlp.table_sizes()}", flush=True)
        if name == "time_vectorised":   # the NLP solve with 8 variables: the host-driven and the device-resident form give the same iterates
            host, dev = nlp.solve(host_qp=True), nlp.solve(device_qp=True)
            for a, c in zip(host, dev):
                assert np.array_equal(a["k_opt"], c["k_opt"]) and a["feasible"] == c["feasible"] and a["iterations"] == c["iterations"] and len(a["k_opt"]) == 8
            print(f"solve (8 variables): feasible {host[0]['feasible']}, {host[0]['iterations']} iterations, k_opt {np.round(host[0]['k_opt'], 4)}", flush=True)
        built[name] = tabs
        nlp.close()
    for (c1, r1, k1, co1), (c2, r2, k2, co2) in zip(built["per_step"], built["time_vectorised"]):
        assert np.array_equal(k1, k2) and np.array_equal(co1, co2) and np.array_equal(c1, c2) and np.abs(r1 - r2).max() <= 1e-12
    print("key128 ok", flush=True)


def _body_fetch8():
    """(runs with ARMOUR_KEY128=1)  BASELINE configs[4] as it reads -- "Fetch 8-DOF arm with payload-mass uncertainty, 100 obstacles": the Fetch arm
    behind a torso yaw joint (include/armour_robot_fetch.h: 9 links, 8 factors, mixed axes), +-50 % mass / inertia on the gripper link, O = 100,
    T = 100, m = 8 T + 9 T O + 32 = 90 832 rows.  Both reach-set kernels and the fused evaluation against the 128-bit oracle."""
    sys.path.insert(0, ROOT)
    from armour_amd import _lib
    from armour_amd.planner import ArmourNLP, default_params, fetch8_robot
    from armour_amd.worlds import random_fetch8_problem
    from oracle import cpu_oracle as orc
    T, O, n, J = 100, 100, 8, 9
    p = random_fetch8_problem(11, O)
    def params(mod):
        pr = mod(T)
        pr.k_range[7] = pr.k_range[6]
        return pr
    oracle = orc.Oracle(robot=orc.fetch8_robot(0.5), params=params(orc.default_params)).set_problem(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    assert oracle.min_margin() > 1e-9
    k = np.random.default_rng(3).uniform(-1, 1, n)
    g_ref, jac_ref = oracle.eval_g_jac(k)
    m = n * T + J * T * O + 4 * n
    assert g_ref.shape == (m,)
    built = {}
    for name, build, B in (("per_step", 1, 1), ("time_vectorised", 2, 3)):
        nlp = ArmourNLP(robot=fetch8_robot(0.5), params=params(default_params))
        nlp.set_option(_lib.OPT_P1_BUILD, build)
        rep = lambda a: np.repeat(np.asarray(a)[None], B, axis=0)
        for _ in range(2):
            nlp.set_parameters(rep(p["q0"]), rep(p["qd0"]), rep(p["qdd0"]), rep(p["q_des"]), rep(p["obstacles"]))
        info = nlp.build_info()
        assert info["kernel"] == name and info["waves"] >= 3, info   # (multi-wave blocks: half the key entries in the same LDS bytes, p1_reach.hip key_cap)
        assert (nlp.n, nlp.J, nlp.m) == (n, J, m)
        b = B - 1
        assert np.abs(nlp.torque_radius()[b] - oracle.torque_radius()).max() <= 1e-10
        assert np.abs(nlp.link_generators()[b] - oracle.link_generators()).max() <= 1e-10
        tabs = []
        for which, cnt in (("link", J), ("torque", n)):
            for i in range(cnt):
                for t in range(0, T, 9):
                    c1, r1, k1, co1 = nlp.pz(which, i, t, b=b)
                    c2, r2, k2, co2 = oracle.pz(which, i, t)
                    assert np.array_equal(k1, k2), (name, which, i, t, len(k1), len(k2))
                    assert np.abs(co1 - co2).max(initial=0.0) <= 1e-12 and np.abs(c1 - c2).max() <= 1e-12 and np.abs(r1 - r2).max() <= 1e-10
                    tabs.append((c1, r1, k1, co1))
        g, jac = nlp.eval_g_jac(rep(k))
        dg, dj = np.abs(g[b] - g_ref).max(), np.abs(jac[b] - jac_ref).max()
        assert dg <= 1e-9 and dj <= 1e-8, (name, dg, dj)
        print(f"fetch8 {name}: {info}, build {nlp.build_ms:.2f} ms ({B} problem(s)), |dg| {dg:.2e} |djac| {dj:.2e}, table sizes {nlp.table_sizes()}", flush=True)
        if name == "per_step":
            assert nlp.build_ms <= 2.0, nlp.build_ms   # (round 4's verdict: one-wave blocks took 7.2 ms for one 8-factor problem)
        built[name] = tabs
        nlp.close()
    for (c1, r1, k1, co1), (c2, r2, k2, co2) in zip(built["per_step"], built["time_vectorised"]):
        assert np.array_equal(k1, k2) and np.array_equal(co1, co2) and np.array_equal(c1, c2) and np.abs(r1 - r2).max() <= 1e-12
    print("fetch8 ok", flush=True)


def _run(code):
    env = dict(os.environ, ARMOUR_KEY128="1")
    env.pop("ARMOUR_HIP_LIB", None)
    return subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)


@pytest.mark.gpu
def test_eight_factor_arm_with_128_bit_keys_matches_the_oracle():
    r = _run("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_key128 as t; t._body()" % (ROOT, os.path.join(ROOT, "tests")))
    assert r.returncode == 0 and "key128 ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    print(r.stdout)


@pytest.mark.gpu
def test_fetch_8dof_payload_100_obstacles_matches_the_oracle():
    r = _run("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_key128 as t; t._body_fetch8()" % (ROOT, os.path.join(ROOT, "tests")))
    assert r.returncode == 0 and "fetch8 ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    print(r.stdout)


def test_fetch8_preset_is_the_same_in_product_and_oracle():
    """CPU: the "Fetch 8-DOF" preset is derived twice -- include/armour_robot_fetch.h (armour_fill_fetch8) and oracle/armour_oracle.cpp
    (oracle_fill_fetch8, from the oracle's own table of CMP/FetchInfo.h) -- and must agree byte for byte; the 64-bit-key library refuses it."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import ctypes as C\n"
            "from armour_amd import planner\n"
            "from oracle import cpu_oracle as orc\n"
            "a, b = planner.fetch8_robot(0.5), orc.fetch8_robot(0.5)\n"
            "assert C.sizeof(a) == C.sizeof(b) and bytes(a) == bytes(b)\n"
            "assert (a.num_joints, a.num_factors, list(a.axes)) == (9, 8, [3, 3, 2, 1, 2, 1, 2, 1, 0]) and a.mass_uncertainty_link[8] == 0.5\n"
            "print('preset ok')\n") % ROOT
    r = _run(code)
    assert r.returncode == 0 and "preset ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    from armour_amd import _lib, planner
    with pytest.raises(_lib.ArmourError):
        planner.fetch8_robot()


def test_key128_libraries_load_and_export_the_abi():
    """CPU: both ABIs' libraries exist, export every symbol of include/armour_hip.h, and say which ABI they are; the oracle's 128-bit build builds
    the 8-factor arm's reach sets and its callback (m = 8 T + 8 T O + 32 rows, 8 columns) -- the restatement itself needs no GPU."""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import ctypes as C, numpy as np\n"
            "from armour_amd import _lib\n"
            "from oracle import cpu_oracle as orc\n"
            "import test_key128 as t\n"
            "L = C.CDLL(_lib.LIB_PATH)\n"
            "assert _lib.LIB_PATH.endswith('libarmour_hip_k128.so') and L.armour_abi_max_factors() == 8\n"
            "assert all(hasattr(L, s) for s in _lib.EXPORTS)\n"
            "pp = orc.default_params(20); pp.k_range[7] = pp.k_range[6]\n"
            "o = orc.Oracle(robot=t.make_eight_factor_arm(orc.kinova_robot()), params=pp)\n"
            "o.set_problem(np.full(8, 0.3), np.full(8, 0.1), np.zeros(8), np.full(8, 0.4), np.array([[0.5, 0.2, 0.6, 0.1, 0, 0, 0, 0.1, 0, 0, 0, 0.1]]))\n"
            "g, jac = o.eval_g_jac(np.full(8, 0.25))\n"
            "assert g.shape == (8 * 20 + 8 * 20 + 32,) and jac.shape[1] == 8 and np.isfinite(g).all()\n"
            "print('abi ok')\n") % (ROOT, os.path.join(ROOT, "tests"))
    r = _run(code)
    assert r.returncode == 0 and "abi ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    import ctypes as C
    from armour_amd import _lib
    assert C.CDLL(os.path.join(ROOT, "armour_amd", "lib", "libarmour_hip.so")).armour_abi_max_factors() == 7 and _lib.MAXF == 7
