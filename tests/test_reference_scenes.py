"""The reference's own worlds as a workload: the 100 saved random scenes of its system test (kinova_src/saved_worlds/random/*.csv,
driven by kinova_src/scripts/kinova_run_100_worlds.m:115-186) and its seven hard scenarios (KSI/kinova_scenarios/
get_kinova_scenario_info.m:1-262), first planning iteration of each (armour_amd/scenes.py).

CPU: the loaders, the waypoint rule, the oracle against tests/golden/reference_scenes.npz (12 worlds x T = 100 / 128).
GPU: all 107 worlds as ONE batch through the C ABI -- tables, g and the Jacobian of EVERY world against the live oracle,
armour_solve's verdict and optimum of every world against scipy SLSQP on the oracle's callbacks; the golden worlds at T = 128
against the committed fixture; padding a world's obstacle list with the far box leaves its solve unchanged."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import ROOT
from helpers import PZ_TESTS_K

C_TOL, R_TOL, G_TOL, J_TOL = 1e-11, 1e-10, 1e-9, 1e-8   # tests/test_baseline_configs.py
GOLDEN = os.path.join(ROOT, "tests", "golden", "reference_scenes.npz")


# ------------------------------------------------------------------------------------------------ CPU
def test_the_107_worlds_load_and_the_waypoint_rule():
    from armour_amd import scenes
    ws = scenes.reference_worlds()
    assert len(ws) == 107 and len(set(n for n, _ in ws)) == 107
    counts = [p["obstacles"].shape[0] for _, p in ws]
    assert min(counts[:100]) == 5 and max(counts[:100]) == 14 and sum(counts[:100]) == 893        # the files hold fewer boxes than their names say
    assert counts[100:] == [1, 1, 2, 10, 4, 12, 4]
    for i, (name, p) in enumerate(ws):
        look = 1.0 if i < 100 else 0.1
        assert np.all(p["qd0"] == 0) and np.all(p["qdd0"] == 0)
        d = p["q_des"] - p["q0"]
        assert abs(np.linalg.norm(d) - look) <= 1e-12, name                                     # robot_arm_straight_line_HLP.m:55: no clipping at the goal
        e = p["goal"] - p["q0"]
        e[scenes.CONTINUOUS] = scenes.angdiff(p["q0"][scenes.CONTINUOUS], p["goal"][scenes.CONTINUOUS])
        assert np.abs(d / look - e / np.linalg.norm(e)).max() <= 1e-12
        obs = p["obstacles"]
        assert np.all(obs[:, [4, 5, 6, 8, 9, 10]] == 0) and np.all(obs[:, [3, 7, 11]] > 0)        # boxes: [c, diag(s / 2)]
    # a continuous joint takes the short way round: goal 3 rad, start -3 rad -> direction negative
    w = scenes.straight_line_waypoint(np.array([-3.0, 0, 0, 0, 0, 0, 0]), np.array([3.0, 0, 0, 0, 0, 0, 0]), 0.5)
    assert abs(w[0] - (-3.5)) <= 1e-12
    w = scenes.straight_line_waypoint(np.array([0, -2.0, 0, 0, 0, 0, 0.0]), np.array([0, 2.0, 0, 0, 0, 0, 0]), 0.5)   # joint 2 has limits: no wrap
    assert abs(w[1] - (-1.5)) <= 1e-12
    b = scenes.as_batch(ws)
    assert b["obstacles"].shape == (107, 14, 12) and b["q0"].shape == (107, 7)
    assert np.array_equal(b["obstacles"][100, 1], scenes.FAR_BOX) and np.array_equal(b["obstacles"][100, 0], ws[100][1]["obstacles"][0])
    with pytest.raises(ValueError):
        scenes.pad_obstacles(ws[105][1]["obstacles"], 4)


def test_hard_scenario_table_is_what_its_script_writes(tmp_path):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_hard_scenarios", os.path.join(ROOT, "tests", "golden", "make_hard_scenarios.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with open(os.path.join(ROOT, "tests", "golden", "scenes", "hard_scenarios.json")) as f:
        table = json.load(f)["scenarios"]
    for s in table:
        name, start, goal, boxes = mod.scenario(s["scenario"])
        assert name == s["name"] and np.allclose(start, s["start"], atol=0, rtol=0) and np.allclose(goal, s["goal"], atol=0, rtol=0)
        assert np.array_equal(np.array([np.concatenate(b) for b in boxes]), np.array(s["boxes"]))
    # get_kinova_scenario_info.m:13-19,254-261: the table of scenario 1, a 1 x 4 x 0.01 m slab at (1.1, 0, 0.8) in the Fetch frame, in the Kinova's
    assert np.allclose(table[0]["boxes"][0], [0.0, 0.0, 1.35, 0.01, 4.0, 1.0], atol=1e-15)
    # make_shelf_obstacle.m: two sides and three boards per shelf
    assert len(table[3]["boxes"]) == 10


def _golden():
    return np.load(GOLDEN)


def _golden_cases():
    return [(str(n), T) for n in np.load(GOLDEN)["names"] for T in (100, 128)]


def _key_digest(pz, T):
    h = hashlib.sha256()
    lc, tc = [], []
    for kind, cnt in (("link", lc), ("torque", tc)):
        for i in range(7):
            for t in range(T):
                keys = np.ascontiguousarray(pz(kind, i, t)[2], dtype=np.uint64)
                cnt.append(len(keys))
                h.update(keys.tobytes())
    return np.array(lc, np.int32), np.array(tc, np.int32), h.hexdigest()


@pytest.mark.parametrize("name,T", _golden_cases())
def test_oracle_reproduces_the_scene_goldens(name, T):
    from oracle.cpu_oracle import Oracle
    gd = _golden()
    pre = f"{name}/T{T}/"
    o = Oracle(T=T).set_problem(gd[f"{name}/q0"], np.zeros(7), np.zeros(7), gd[f"{name}/q_des"], gd[f"{name}/obstacles"])
    assert o.m == int(gd[pre + "m"])
    assert abs(o.min_margin() - float(gd[pre + "min_margin"])) <= 1e-6 * float(gd[pre + "min_margin"]) + 1e-15
    assert np.abs(o.torque_radius() - gd[pre + "torque_radius"]).max() <= 1e-12
    assert np.abs(o.link_generators()[::8] - gd[pre + "link_gens"]).max() <= 1e-13
    lc, tc, dig = _key_digest(o.pz, T)
    assert np.array_equal(lc, gd[pre + "link_count"]) and np.array_equal(tc, gd[pre + "torque_count"]) and dig == str(gd[pre + "key_digest"])
    _, _, gl, gu = o.bounds()
    assert np.abs(gl[:7 * T] - gd[pre + "g_l"]).max() <= 1e-12 and np.abs(gu[:7 * T] - gd[pre + "g_u"]).max() <= 1e-12
    rows = gd[pre + "jac_rows"]
    g0, _ = o.eval_g_jac(np.zeros(7))
    gt, jt = o.eval_g_jac(PZ_TESTS_K)
    assert np.abs(g0 - gd[pre + "g_k0"]).max() <= 1e-12 and np.abs(gt[rows] - gd[pre + "g_kt"]).max() <= 1e-12
    assert np.abs(jt[rows] - gd[pre + "jac_kt"]).max() <= 1e-12
    assert abs(o.eval_f(PZ_TESTS_K) - float(gd[pre + "f_kt"])) <= 1e-13 and np.abs(o.eval_grad_f(PZ_TESTS_K) - gd[pre + "gradf_kt"]).max() <= 1e-13


# ------------------------------------------------------------------------------------------------ GPU
def _slsqp(o):
    from scipy.optimize import minimize
    _, _, gl, gu = o.bounds()
    two = gl > -1e18

    def cons(k):
        g, _ = o.eval_g_jac(k, want_jac=False)
        return np.concatenate([gu - g, (g - gl)[two]])

    def cons_jac(k):
        _, j = o.eval_g_jac(k, want_g=False)
        return np.concatenate([-j, j[two]])

    r = minimize(o.eval_f, np.zeros(7), jac=o.eval_grad_f, bounds=[(-1, 1)] * 7, method="SLSQP",
                 constraints=[dict(type="ineq", fun=cons, jac=cons_jac)], options=dict(maxiter=200, ftol=1e-12))
    g, _ = o.eval_g_jac(r.x, want_jac=False)
    return r, bool(np.all(g <= gu + 1e-6) and np.all(g >= gl - 1e-6))



def _rows_differ_only_through_noise_planes(name, p, o, T, O, off, k, g_dev, g_ref):
    """Rows `off` of a padded world differ from the oracle's.  Accepted only if (i) every one is a collision row whose arg-max
    plane -- in the oracle's table or in the device's -- has a normal that is rounding noise (two generators parallel in exact
    arithmetic: helpers.noise_planes), (ii) the device's FULL table, built on a handle of this world alone, agrees with the
    oracle's on every other plane, and (iii) the device's row is the reference's row formula applied to the device's own table."""
    from armour_amd.planner import ArmourNLP
    from armour_amd.scenes import pad_obstacles
    from helpers import noise_planes
    n = J = 7
    obs = pad_obstacles(p["obstacles"], O)
    one = ArmourNLP(T=T).set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], obs)
    A2, d2, dl2 = (a[0] for a in one.hyperplanes())
    A, d, dl = o.hyperplanes()
    gens = o.link_generators()
    cen = o.slice_links(k)
    for r in off:
        q = r - n * T
        assert 0 <= q < J * T * O, (name, r)
        l, t, ob = q // (T * O), (q // O) % T, q % O
        noise = noise_planes(obs[ob], gens[t, l])
        assert noise.any(), (name, r)
        ok = ~noise
        assert np.abs(A[t, l, ob][ok] - A2[t, l, ob][ok]).max() <= C_TOL and np.abs(d[t, l, ob][ok] - d2[t, l, ob][ok]).max() <= C_TOL
        assert np.abs(dl[t, l, ob][ok] - dl2[t, l, ob][ok]).max() <= C_TOL
        vals = []
        for (Ax, dx, dlx) in ((A[t, l, ob], d[t, l, ob], dl[t, l, ob]), (A2[t, l, ob], d2[t, l, ob], dl2[t, l, ob])):
            v = Ax @ cen[t, l]
            live = np.linalg.norm(Ax, axis=1) > 0
            pos, neg = np.where(live, v - dx - dlx, -1e8), np.where(live, -v + dx - dlx, -1e8)
            both = np.concatenate([pos, neg])
            vals.append((-both.max(), int(np.argmax(both)) % 36))
        assert noise[vals[0][1]] or noise[vals[1][1]], (name, r, vals)          # a noise plane decides the row in one of the two tables
        assert abs(vals[0][0] - g_ref[r]) <= G_TOL and abs(vals[1][0] - g_dev[r]) <= G_TOL, (name, r, vals, g_ref[r], g_dev[r])
    one.close()
    return int(off.size)


@pytest.mark.gpu
def test_all_107_reference_worlds_as_one_batch():
    """Tables, g, Jacobian of every world against the live oracle; armour_solve of every world against SLSQP on the oracle's callbacks."""
    from armour_amd.planner import ArmourNLP
    from armour_amd.scenes import as_batch, reference_worlds
    from armour_amd.worlds import random_k
    from oracle.cpu_oracle import Oracle
    T = 100
    ws = reference_worlds()
    bp = as_batch(ws)
    B, O = bp["obstacles"].shape[:2]
    assert (B, O) == (107, 14)
    nlp = ArmourNLP(T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    n, J, m = nlp.n, nlp.J, nlp.m
    ks = np.stack([np.tile(PZ_TESTS_K, (B, 1)), random_k(77, B)])
    outs = [nlp.eval_g_jac(k) for k in ks]
    outs = [(g.copy(), j.copy()) for g, j in outs]
    tr, gens = nlp.torque_radius(), nlp.link_generators()
    sols = nlp.solve(tolerance=1e-7, max_iterations=100)
    host = nlp.solve(tolerance=1e-7, max_iterations=100, host_qp=True)
    worst = dict(coef=0.0, radius=0.0, g=0.0, jac=0.0, k=0.0, cost=0.0)
    n_feasible = n_ref_feasible = n_noise_rows = 0
    min_margin = np.inf
    for b, (name, p) in enumerate(ws):
        o = Oracle(T=T).set_problem(bp["q0"][b], bp["qd0"][b], bp["qdd0"][b], bp["q_des"][b], bp["obstacles"][b])
        min_margin = min(min_margin, o.min_margin())
        assert o.min_margin() > 1e-9, name
        for which, cnt in (("link", J), ("torque", n)):
            for i in range(cnt):
                for t in range(b % 4, T, 4):   # every fourth time step of every link / joint, the phase moving with the world
                    c, ind, keys, co = o.pz(which, i, t)
                    c2, ind2, keys2, co2 = nlp.pz(which, i, t, b=b)
                    assert np.array_equal(keys, keys2), (name, which, i, t)
                    if len(keys):
                        worst["coef"] = max(worst["coef"], np.abs(co - co2).max())
                    worst["coef"] = max(worst["coef"], np.abs(c - c2).max())
                    worst["radius"] = max(worst["radius"], np.abs(ind - ind2).max())
        worst["radius"] = max(worst["radius"], np.abs(tr[b] - o.torque_radius()).max())
        worst["coef"] = max(worst["coef"], np.abs(gens[b] - o.link_generators()).max())
        for s in range(2):
            gr, jr = o.eval_g_jac(ks[s, b])
            dg, dj = np.abs(outs[s][0][b] - gr), np.abs(outs[s][1][b] - jr).max(axis=1)
            off = np.nonzero((dg > G_TOL) | (dj > J_TOL))[0]
            if off.size:   # allowed only where the winning half-space is one whose normal is rounding noise (helpers.noise_planes)
                n_noise_rows += _rows_differ_only_through_noise_planes(name, p, o, T, O, off, ks[s, b], outs[s][0][b], gr)
                dg[off], dj[off] = 0.0, 0.0
            worst["g"] = max(worst["g"], dg.max())
            worst["jac"] = max(worst["jac"], dj.max())
        assert worst["coef"] <= C_TOL and worst["radius"] <= R_TOL and worst["g"] <= G_TOL and worst["jac"] <= J_TOL, (name, worst)
        # ---- the solve: the same verdict as an independent solver on the oracle's callbacks, and the same optimum
        sol = sols[b]
        assert np.array_equal(sol["k_opt"], host[b]["k_opt"]) and sol["feasible"] == host[b]["feasible"] and sol["cost"] == host[b]["cost"], name   # both forms, same iterates
        ref, ref_feasible = _slsqp(o)
        n_feasible += int(sol["feasible"])
        n_ref_feasible += int(ref_feasible)
        assert bool(sol["feasible"]) == ref_feasible, (name, sol, ref.x, ref.fun)
        if ref_feasible:
            gk, _ = o.eval_g_jac(sol["k_opt"], want_jac=False)
            _, _, gl, gu = o.bounds()
            assert np.all(gk <= gu + 1e-6) and np.all(gk >= gl - 1e-6), name
            assert np.all(np.abs(sol["k_opt"]) <= 1 + 1e-12)
            assert abs(sol["cost"] - o.eval_f(sol["k_opt"])) <= 1e-10
            worst["cost"] = max(worst["cost"], abs(sol["cost"] - ref.fun) / (1 + abs(ref.fun)))
            worst["k"] = max(worst["k"], np.abs(sol["k_opt"] - ref.x).max())
            assert abs(sol["cost"] - ref.fun) <= 1e-6 * (1 + abs(ref.fun)), (name, sol, ref.fun)
            assert np.abs(sol["k_opt"] - ref.x).max() <= 1e-4, (name, sol["k_opt"], ref.x)    # (the cost is strictly convex in k: one optimum)
    print(f"107 reference worlds: {n_feasible} feasible (independent solver: {n_ref_feasible}); worst deviations {worst}; smallest prune margin {min_margin:.3e}; "
          f"{n_noise_rows} collision row(s) decided by a half-space whose normal is rounding noise")
    assert n_noise_rows <= 16
    # ---- padding: worlds with their own obstacle count on handles of their own give the batch's optimum
    for b in (0, 55, 100, 104):
        p = ws[b][1]
        one = ArmourNLP(T=T).set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
        s1 = one.solve(tolerance=1e-7, max_iterations=100)[0]
        assert s1["feasible"] == sols[b]["feasible"] and np.abs(s1["k_opt"] - sols[b]["k_opt"]).max() <= 1e-9, (ws[b][0], s1, sols[b])
        own = p["obstacles"].shape[0]
        g1, _ = one.eval_g_jac(ks[0, b])
        col = outs[0][0][b, n * T:n * T + J * T * O].reshape(J * T, O)
        assert np.abs(col[:, :own].ravel() - g1[0, n * T:n * T + J * T * own]).max() <= 1e-12     # the world's own rows (a lone problem builds step by step: radii to rounding)
        assert col[:, own:].max(initial=-np.inf) < -40.0                                          # the far box: rows 40 m inside
        one.close()
    nlp.close()


@pytest.mark.gpu
@pytest.mark.parametrize("T", [100, 128])
def test_device_against_the_scene_goldens(T):
    """The committed fixture (no oracle at run time): the twelve golden worlds as one batch at T = 100 and at the reference's T = 128."""
    from armour_amd.planner import ArmourNLP
    from armour_amd.scenes import pad_obstacles
    gd = _golden()
    names = [str(n) for n in gd["names"]]
    own = [gd[f"{n}/obstacles"].shape[0] for n in names]
    O = max(own)
    z = np.zeros((len(names), 7))
    nlp = ArmourNLP(T=T).set_parameters(np.stack([gd[f"{n}/q0"] for n in names]), z, z, np.stack([gd[f"{n}/q_des"] for n in names]),
                                        np.stack([pad_obstacles(gd[f"{n}/obstacles"], O) for n in names]))
    n, J = nlp.n, nlp.J
    g0, _ = nlp.eval_g_jac(np.zeros((len(names), 7)))
    g0 = g0.copy()
    gt, jt = nlp.eval_g_jac(np.tile(PZ_TESTS_K, (len(names), 1)))
    tr, gens = nlp.torque_radius(), nlp.link_generators()
    for b, name in enumerate(names):
        pre = f"{name}/T{T}/"
        assert np.abs(tr[b] - gd[pre + "torque_radius"]).max() <= R_TOL
        assert np.abs(gens[b][::8] - gd[pre + "link_gens"]).max() <= C_TOL
        lc, tc, dig = _key_digest(lambda kind, i, t: nlp.pz(kind, i, t, b=b), T)
        assert np.array_equal(lc, gd[pre + "link_count"]) and np.array_equal(tc, gd[pre + "torque_count"]) and dig == str(gd[pre + "key_digest"]), name
        # rows of the padded batch -> rows of the world with its own obstacle count
        o_own = own[b]
        idx = np.concatenate([np.arange(n * T), (n * T + (np.arange(J * T)[:, None] * O + np.arange(o_own)[None, :])).ravel(), n * T + J * T * O + np.arange(4 * n)])
        assert idx.size == int(gd[pre + "m"])
        rows = gd[pre + "jac_rows"]
        assert np.abs(g0[b][idx] - gd[pre + "g_k0"]).max() <= G_TOL
        assert np.abs(gt[b][idx][rows] - gd[pre + "g_kt"]).max() <= G_TOL
        assert np.abs(jt[b][idx][rows] - gd[pre + "jac_kt"]).max() <= J_TOL
        assert abs(nlp.eval_f(np.tile(PZ_TESTS_K, (len(names), 1)))[b] - float(gd[pre + "f_kt"])) <= 1e-12
    nlp.close()


REF_SCENARIOS = "/root/reference/kinova_src/kinova_simulator_interfaces/kinova_scenarios/get_kinova_scenario_info.m"
REF_SCENES = "/root/reference/kinova_src/saved_worlds/random"


@pytest.mark.skipif(not os.path.exists(REF_SCENARIOS), reason="the reference checkout is only present in the build container")
def test_fixtures_are_the_references_own_numbers():
    """Where the reference checkout exists (this container, not the GPU box): the committed CSVs are byte for byte the reference's files, and start / goal of
    every hard scenario in the JSON table are the (uncommented) MATLAB statements of get_kinova_scenario_info.m, read as text and evaluated."""
    import re
    from armour_amd.scenes import SCENE_DIR
    names = sorted(f for f in os.listdir(REF_SCENES) if f.endswith(".csv"))
    assert len(names) == 100
    for f in names:
        assert open(os.path.join(REF_SCENES, f), "rb").read() == open(os.path.join(SCENE_DIR, f), "rb").read(), f
    text = open(REF_SCENARIOS).read()
    with open(os.path.join(SCENE_DIR, "hard_scenarios.json")) as fh:
        table = {s["scenario"]: s for s in json.load(fh)["scenarios"]}
    blocks = re.split(r"\n\s*case (\d) %", text)
    for k in range(1, len(blocks), 2):
        case, body = int(blocks[k]), blocks[k + 1]
        for field in ("start", "goal"):
            m = [ln for ln in body.splitlines() if re.match(rf"\s*{field} = \[", ln)]     # (commented alternatives start with %)
            vec = re.search(r"\[(.*?)\]", m[0]).group(1)
            vals = [float(eval(x.strip(), {"pi": np.pi})) for x in re.split(r"[;,]", vec) if x.strip()]
            assert np.allclose(vals, table[case][field], rtol=0, atol=1e-15), (case, field)
    assert len(table) == 7
