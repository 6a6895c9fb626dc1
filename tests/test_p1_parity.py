"""GPU parity of the whole product path -- armour_set_problems (device JRS / FK / RNEA / torque radius /
half-space tables) followed by the fused eval kernel -- against the committed golden fixtures and the live
CPU oracle, all through the C ABI.

Stated tolerances (BASELINE.md "parity tolerance"; north_star "within a stated fp tolerance"):
    identical monomial key sets per (link|joint, t);
    |d coeff|, |d centre|, |d independent radius| <= 1e-11;  link generators, half-space tables <= 1e-11
    torque radius <= 1e-10;  |dg| <= 1e-9;  |djac| <= 1e-8     (all absolute; observed ~1e-14)
valid when no monomial norm lies within 1e-9 (relative) of SIMPLIFY_THRESHOLD -- each fixture records that
margin (min_margin, ~1e-7 here), so a prune flip would be detected rather than silent.  Sources of the
residual: summation order of equal keys (the reference's std::sort leaves it unspecified), wave-parallel
abs-sums, device cos/sin vs glibc."""
import numpy as np
import pytest

from helpers import PZ_TESTS_K, load_golden

pytestmark = pytest.mark.gpu

CASES = [f"{n}_T{T}" for n in ("sample", "debug", "scene013") for T in (100, 128)]
C_TOL, R_TOL, G_TOL, J_TOL = 1e-11, 1e-10, 1e-9, 1e-8


def _nlp(gd):
    from armour_amd.planner import ArmourNLP
    return ArmourNLP(T=int(gd["T"])).set_parameters(gd["q0"], gd["qd0"], gd["qdd0"], gd["q_des"], gd["obstacles"])


@pytest.mark.parametrize("case", CASES)
def test_against_golden_fixtures(case):
    gd = load_golden(case)
    assert float(gd["min_margin"]) > 1e-9
    nlp = _nlp(gd)
    T = int(gd["T"])
    assert np.abs(nlp.torque_radius()[0] - gd["torque_radius"]).max() <= R_TOL
    assert np.abs(nlp.link_generators()[0] - gd["link_gens"]).max() <= C_TOL
    lc, lk, tc, tk = [], [], [], []
    for i in range(7):
        for t in range(T):
            k1 = nlp.pz("link", i, t)[2]; lc.append(len(k1)); lk.append(k1)
            k2 = nlp.pz("torque", i, t)[2]; tc.append(len(k2)); tk.append(k2)
    assert np.array_equal(np.array(lc), gd["link_count"]) and np.array_equal(np.concatenate(lk), gd["link_keys"])
    assert np.array_equal(np.array(tc), gd["torque_count"]) and np.array_equal(np.concatenate(tk), gd["torque_keys"])
    _, _, gl, gu = nlp.get_bounds_info()
    assert np.abs(gl[0] - gd["g_l"]).max() <= R_TOL and np.abs(gu[0] - gd["g_u"]).max() <= R_TOL
    for tag, k in (("k0", np.zeros(7)), ("kt", PZ_TESTS_K)):
        g, jac = nlp.eval_g_jac(k)
        assert np.array_equal(nlp.eval_g(k), g) and np.array_equal(nlp.eval_jac_g(k), jac)   # the g-only / jac-only kernels
        assert np.abs(g[0] - gd[f"g_{tag}"]).max() <= G_TOL
        assert np.abs(jac[0][gd["jac_rows"]] - gd[f"jac_{tag}"]).max() <= J_TOL
        assert abs(nlp.eval_f(k)[0] - float(gd[f"f_{tag}"])) <= 1e-12
        assert np.abs(nlp.eval_grad_f(k)[0] - gd[f"gradf_{tag}"]).max() <= 1e-12


def _compare_tables(nlp, oracles):
    for b, o in enumerate(oracles):
        assert o.min_margin() > 1e-9
        for which, cnt in (("link", o.J), ("torque", o.n)):
            for i in range(cnt):
                for t in range(o.T):
                    c, ind, keys, co = o.pz(which, i, t)
                    c2, ind2, keys2, co2 = nlp.pz(which, i, t, b=b)
                    assert np.array_equal(keys, keys2), (b, which, i, t)
                    if len(keys):
                        assert np.abs(co - co2).max() <= C_TOL
                    assert np.abs(c - c2).max() <= C_TOL and np.abs(ind - ind2).max() <= C_TOL
        assert np.abs(nlp.torque_radius()[b] - o.torque_radius()).max() <= R_TOL
        assert np.abs(nlp.link_generators()[b] - o.link_generators()).max() <= C_TOL
    if oracles[0].O:
        A2, d2, dl2 = nlp.hyperplanes()
        for b, o in enumerate(oracles):
            A, d, dl = o.hyperplanes()
            assert np.abs(A - A2[b]).max() <= C_TOL and np.abs(d - d2[b]).max() <= C_TOL and np.abs(dl - dl2[b]).max() <= C_TOL


def test_random_batch_against_live_oracle():
    """B = 4 independent worlds in one handle (the batched twin of the reference's single-problem process)."""
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch, random_k
    from oracle.cpu_oracle import Oracle
    T, O, B = 100, 9, 4
    bp = random_batch(20, B, O)
    nlp = ArmourNLP(T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    oracles = [Oracle(T=T).set_problem(bp["q0"][b], bp["qd0"][b], bp["qdd0"][b], bp["q_des"][b], bp["obstacles"][b]) for b in range(B)]
    _compare_tables(nlp, oracles)
    ks = random_k(99, B)
    g, jac = nlp.eval_g_jac(ks)
    for b, o in enumerate(oracles):
        gr, jr = o.eval_g_jac(ks[b])
        assert np.abs(g[b] - gr).max() <= G_TOL and np.abs(jac[b] - jr).max() <= J_TOL
    ts = nlp.table_sizes()
    assert ts["sum_link"] == sum(o.table_sizes()["sum_link"] for o in oracles)
    assert ts["sum_torque"] == sum(o.table_sizes()["sum_torque"] for o in oracles)


def _rotate_obstacles(obs, seed):
    """Turn the axis-aligned boxes into generally oriented ones: generators g_i -> R g_i with a random rotation R."""
    rng = np.random.default_rng(seed)
    out = obs.copy()
    for o in out.reshape(-1, 12):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        G = o[3:].reshape(3, 3)          # rows = generators
        o[3:] = (G @ q.T).ravel()
    return out


@pytest.mark.parametrize("B", [1, 8])
def test_rotated_obstacles_keep_every_plane(B):
    """With generally oriented obstacles no half-space is a duplicate of another (the reference's worlds only hold
    axis-aligned boxes, where 12 of 36 are): more than 24 planes stay live, so the fused kernel runs its 9-slot variants
    -- with the compact link x link normals (B = 1) and with d recomputed from the obstacle centres (B = 8)."""
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch, random_k
    from oracle.cpu_oracle import Oracle
    T, O = 100, 5
    bp = random_batch(60, B, O)
    bp["obstacles"] = _rotate_obstacles(bp["obstacles"], 61)
    nlp = ArmourNLP(T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    ks = random_k(5, B)
    g, jac = nlp.eval_g_jac(ks)
    for b in sorted({0, B - 1}):
        o = Oracle(T=T).set_problem(bp["q0"][b], bp["qd0"][b], bp["qdd0"][b], bp["q_des"][b], bp["obstacles"][b])
        gr, jr = o.eval_g_jac(ks[b])
        assert np.abs(g[b] - gr).max() <= G_TOL and np.abs(jac[b] - jr).max() <= J_TOL
    if B > 1:   # batched (d recomputed) against a single-problem handle (d read): bit for bit
        one = ArmourNLP(T=T).set_parameters(bp["q0"][B - 1], bp["qd0"][B - 1], bp["qdd0"][B - 1], bp["q_des"][B - 1], bp["obstacles"][B - 1])
        g1, j1 = one.eval_g_jac(ks[B - 1])
        assert np.array_equal(g[B - 1], g1[0]) and np.array_equal(jac[B - 1], j1[0])


def test_fast_initial_state_stress():
    """speed-limit initial velocity and |qdd0| = 3: the largest intermediate PZs we know of (~2.6k raw terms)."""
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import SPEED, random_problem
    from oracle.cpu_oracle import Oracle
    T = 100
    p = random_problem(0, 2)
    p["qd0"] = SPEED.copy(); p["qdd0"] = np.full(7, 3.0)
    nlp = ArmourNLP(T=T).set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    o = Oracle(T=T).set_problem(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    _compare_tables(nlp, [o])


def test_handle_reuse_and_state_errors():
    """set_problems can be called again on the same handle with a different (B, O); eval before set is an error;
    too-small table capacities are reported, never silently truncated."""
    from armour_amd import _lib
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_problem
    nlp = ArmourNLP(T=20)
    with pytest.raises(_lib.ArmourError) as ei:
        nlp.L.armour_get_sizes(nlp.h, None, None, None) and None
        _lib.check(nlp.L.armour_get_sizes(nlp.h, None, None, None))
    assert ei.value.code == _lib.ESTATE
    p = random_problem(1, 3)
    nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    g1 = nlp.eval_g(np.zeros(7))
    p2 = random_problem(2, 12)
    nlp.set_parameters(p2["q0"], p2["qd0"], p2["qdd0"], p2["q_des"], p2["obstacles"])
    assert nlp.m == 7 * 20 + 7 * 20 * 12 + 28
    nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    assert np.array_equal(nlp.eval_g(np.zeros(7)), g1)
    small = _lib.ArmourLimits(link_monomials=2)
    with pytest.raises(_lib.ArmourError) as ei:
        ArmourNLP(T=20, limits=small).set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    assert ei.value.code == _lib.ECAPACITY


def test_linearity_of_slices_in_duplicate_obstacles_and_permutation():
    """Size-independent properties at the full BASELINE size (T=100, O=20): permuting the obstacle list permutes
    the collision rows and nothing else; rows of a duplicated obstacle are identical."""
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_problem
    T, O = 100, 20
    p = random_problem(42, O)
    p["obstacles"][7] = p["obstacles"][3]
    nlp = ArmourNLP(T=T).set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    g, jac = nlp.eval_g_jac(PZ_TESTS_K)
    perm = np.random.default_rng(0).permutation(O)
    nlp2 = ArmourNLP(T=T).set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"][perm])
    g2, jac2 = nlp2.eval_g_jac(PZ_TESTS_K)
    n = 7
    a = g[0][n * T:-28].reshape(7, T, O); b = g2[0][n * T:-28].reshape(7, T, O)
    assert np.array_equal(a[:, :, perm], b)
    ja = jac[0][n * T:-28].reshape(7, T, O, n); jb = jac2[0][n * T:-28].reshape(7, T, O, n)
    assert np.array_equal(ja[:, :, perm], jb)
    assert np.array_equal(g[0][:n * T], g2[0][:n * T]) and np.array_equal(g[0][-28:], g2[0][-28:])
    assert np.array_equal(a[:, :, 7], a[:, :, 3]) and np.array_equal(ja[:, :, 7], ja[:, :, 3])


def test_file_protocol_planning_iteration(tmp_path, sample_problem):
    """armour.in -> device reach sets + constraints -> the five *.out files (RT/armour_main.cu:36-76,312-372)."""
    from armour_amd import file_protocol as fp
    from armour_amd.planner import ArmourNLP
    from oracle.cpu_oracle import Oracle
    p = sample_problem
    fp.write_armour_in(tmp_path / fp.IN_NAME, p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    nlp = ArmourNLP(T=128)
    feasible = fp.run_planning_iteration(nlp, tmp_path, k=PZ_TESTS_K)
    o = Oracle(T=128).set_problem(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    g_ref, _ = o.eval_g_jac(PZ_TESTS_K)
    g_file = np.loadtxt(tmp_path / "armour_constraints.out")
    assert g_file.shape == (nlp.m,) and np.allclose(g_file, g_ref, rtol=1e-5, atol=1e-12)   # setprecision(6)
    assert np.allclose(np.loadtxt(tmp_path / "armour_control_input_radius.out"), o.torque_radius().T, rtol=1e-9)
    assert np.allclose(np.loadtxt(tmp_path / "armour_joint_position_center.out").reshape(128, 7, 3), o.slice_links(PZ_TESTS_K), rtol=1e-9, atol=1e-12)
    assert np.allclose(np.loadtxt(tmp_path / "armour_joint_position_radius.out").reshape(128, 7, 3, 6), o.link_generators(), rtol=1e-9, atol=1e-12)
    k_opt, ms = fp.read_armour_out(tmp_path / "armour.out")
    assert (k_opt is not None) == feasible and ms > 0


def test_kinova_with_gripper_payload():
    """RT/KinovaInfo.h: 8 links (7 actuated + the 1.72 kg gripper on a fixed joint), 3 % mass / inertia uncertainty,
    many obstacles -- the closest configuration the reference's RT path supports to BASELINE configs[4]
    (its 7-factor u64 key cannot hold an 8-factor arm, and CMP/FetchInfo.h has no link zonotopes)."""
    from armour_amd.planner import ArmourNLP, kinova_gripper_robot
    from armour_amd.worlds import random_k, random_problem
    from oracle.cpu_oracle import Oracle
    from oracle.cpu_oracle import kinova_gripper_robot as oracle_robot
    T, O = 100, 100
    p = random_problem(8, O)
    nlp = ArmourNLP(robot=kinova_gripper_robot(), T=T).set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    o = Oracle(robot=oracle_robot(), T=T).set_problem(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    assert nlp.J == 8 and nlp.m == 7 * T + 8 * T * O + 28 == o.m
    _compare_tables(nlp, [o])
    for k in (PZ_TESTS_K, random_k(5, 1)[0]):
        g, jac = nlp.eval_g_jac(k)
        gr, jr = o.eval_g_jac(k)
        assert np.abs(g[0] - gr).max() <= G_TOL and np.abs(jac[0] - jr).max() <= J_TOL
    # the gripper makes the robust-input radius larger than on the bare arm for the same motion
    bare = ArmourNLP(T=T).set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"][:1])
    assert nlp.torque_radius()[0].mean() != bare.torque_radius()[0].mean()


def _fetch_problem(seed, O):
    from armour_amd.worlds import random_fetch_problem
    return random_fetch_problem(seed, O)


@pytest.mark.parametrize("mode", ["armour", "armtd"])
def test_fetch_mixed_axes_payload_uncertainty(mode):
    """BASELINE configs[4]: Fetch (CMP/FetchInfo.h: 9 links, 7 factors, joint axes {z,y,x,y,x,y,x}, two fixed links),
    +-50 % mass / inertia uncertainty on the last link (the payload), 100 obstacles, T = 100 -- device against the oracle.
    The first exercise of rotations about x and y (p1_reach.hip make_rotation, the joint-axis entries of the RNEA).
    Link boxes / M_max are stated stand-ins (include/armour_robot_fetch.h); the "8-DOF" arm of BASELINE.json would need
    an 8th factor, which the reference's own 64-bit monomial key cannot hold (RT/PZsparse.h:8-21).  `armtd` runs the same
    robot through the comparison planner's chain -- the code path FetchInfo.h belongs to in the reference."""
    from armour_amd.planner import ArmourNLP, default_params, fetch_robot
    from armour_amd.worlds import random_k, synthetic_offline_jrs
    from oracle.cpu_oracle import Oracle
    from oracle.cpu_oracle import default_params as oracle_params
    from oracle.cpu_oracle import fetch_robot as oracle_fetch
    T, O = 100, 100
    p = _fetch_problem(11, O)
    nlp = ArmourNLP(robot=fetch_robot(0.5), params=default_params(T))
    o = Oracle(robot=oracle_fetch(0.5), params=oracle_params(T))
    if mode == "armour":
        nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
        o.set_problem(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
        assert nlp.J == 9 and nlp.m == 7 * T + 9 * T * O + 28 == o.m
    else:
        jrs, kr = synthetic_offline_jrs(p["qd0"], T=T)
        nlp.set_parameters_armtd(p["q0"], p["qd0"], p["q_des"], jrs, kr, p["obstacles"])
        o.set_problem_armtd(p["q0"], p["qd0"], p["q_des"], jrs, kr, p["obstacles"])
        assert nlp.m == 9 * T * O + 28 == o.m
    _compare_tables(nlp, [o])
    for k in (PZ_TESTS_K, random_k(6, 1)[0]):
        g, jac = nlp.eval_g_jac(k)
        gr, jr = o.eval_g_jac(k)
        assert np.abs(g[0] - gr).max() <= G_TOL and np.abs(jac[0] - jr).max() <= J_TOL
    _, _, gl, gu = nlp.get_bounds_info()
    _, _, ogl, ogu = o.bounds()
    assert np.abs(gl[0] - ogl).max() <= R_TOL and np.abs(gu[0] - ogu).max() <= R_TOL
    if mode == "armour":
        # the payload interval only widens the robust-input radius: same polynomials as the 3 % robot, larger radii
        base = ArmourNLP(robot=fetch_robot(), params=default_params(T)).set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"][:1])
        tr, tb = nlp.torque_radius()[0], base.torque_radius()[0]
        assert (tr >= tb).all() and (tr > tb + 1e-6).any()
        assert np.array_equal(nlp.pz("torque", 3, 50)[2], base.pz("torque", 3, 50)[2])
        # and a batch of two such problems equals the single-problem handles bit for bit (1-wave blocks vs 3-wave blocks)
        p2 = _fetch_problem(12, O)
        st = lambda key: np.stack([p[key], p2[key]])
        two = ArmourNLP(robot=fetch_robot(0.5), params=default_params(T)).set_parameters(st("q0"), st("qd0"), st("qdd0"), st("q_des"), st("obstacles"))
        g2, j2 = two.eval_g_jac(np.stack([PZ_TESTS_K, PZ_TESTS_K]))
        g1, j1 = nlp.eval_g_jac(PZ_TESTS_K)
        assert np.array_equal(g2[0], g1[0]) and np.array_equal(j2[0], j1[0])


def test_more_work_items_than_resident_waves():
    """B*T = 1000 (problem, time step) items exceed the persistent grid (<= 4 waves x 256 CUs would hold them, the
    LDS-limited grid of 768 does not): waves loop over several items and must give the same tables as a
    single-problem handle, bit for bit."""
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch, random_k
    T, O, B = 100, 2, 10
    bp = random_batch(1000, B, O)
    nlp = ArmourNLP(T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    ks = random_k(3, B)
    g, jac = nlp.eval_g_jac(ks)
    for b in (0, 7, 9):
        one = ArmourNLP(T=T).set_parameters(bp["q0"][b], bp["qd0"][b], bp["qdd0"][b], bp["q_des"][b], bp["obstacles"][b])
        g1, j1 = one.eval_g_jac(ks[b])
        assert np.array_equal(g[b], g1[0]) and np.array_equal(jac[b], j1[0])
        assert np.array_equal(nlp.torque_radius()[b], one.torque_radius()[0])


def test_two_pass_build_of_large_batches():
    """The per-step kernel, from 16 items per CU on, first runs every (problem, time step) item with 2048-entry sort buffers
    (four waves per CU) and rebuilds only the items that overflowed them with the full buffers: the tables must equal those
    of single-problem handles (one pass, full buffers) bit for bit, including for a fast initial state whose products are
    the ones that overflow.  A batch of this size is built time-vectorised by default (pz_tv.h): the batch's handle is held to
    the per-step kernel by ARMOUR_OPT_P1_BUILD = 1 (round 4: a handle option; an environment switch read once per process before)."""
    from armour_amd import _lib
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch, random_k
    T, O, B = 100, 1, 42
    bp = random_batch(500, B, O)
    bp["qd0"][5] = 0.9 * np.array([1.3963, 1.3963, 1.3963, 1.3963, 1.2218, 1.2218, 1.2218])   # fast start: the largest PZs
    nlp = ArmourNLP(T=T).set_option(_lib.OPT_P1_BUILD, 1).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    assert nlp.build_info()["kernel"] == "per_step" and nlp.build_info()["launches"] >= 2
    ks = random_k(9, B)
    g, jac = nlp.eval_g_jac(ks)
    for b in (0, 5, 23, 41):
        one = ArmourNLP(T=T).set_parameters(bp["q0"][b], bp["qd0"][b], bp["qdd0"][b], bp["q_des"][b], bp["obstacles"][b])
        g1, j1 = one.eval_g_jac(ks[b])
        assert np.array_equal(g[b], g1[0]) and np.array_equal(jac[b], j1[0])
        assert np.array_equal(nlp.torque_radius()[b], one.torque_radius()[0])
        assert np.array_equal(nlp.link_generators()[b], one.link_generators()[0])
        one.close()
    nlp.close()


def test_batch_built_time_vectorised_against_single_problem_handles():
    """The same batch in the default configuration: B = 42 is built time-vectorised (three waves per group of 50 time
    steps), the single-problem handles step by step.  Keys, coefficients and centres equal bit for bit; radii, and through
    them g, to rounding (the two kernels add the pruned amounts in different orders)."""
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch, random_k
    T, O, B = 100, 1, 42
    bp = random_batch(500, B, O)
    bp["qd0"][5] = 0.9 * np.array([1.3963, 1.3963, 1.3963, 1.3963, 1.2218, 1.2218, 1.2218])
    nlp = ArmourNLP(T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    ks = random_k(9, B)
    g, jac = nlp.eval_g_jac(ks)
    for b in (0, 5, 23, 41):
        one = ArmourNLP(T=T).set_parameters(bp["q0"][b], bp["qd0"][b], bp["qdd0"][b], bp["q_des"][b], bp["obstacles"][b])
        g1, j1 = one.eval_g_jac(ks[b])
        assert np.abs(g[b] - g1[0]).max() <= 1e-12 * max(1.0, np.abs(g1[0]).max()) and np.abs(jac[b] - j1[0]).max() <= 1e-12 * max(1.0, np.abs(j1[0]).max())
        assert np.abs(nlp.torque_radius()[b] - one.torque_radius()[0]).max() <= 1e-12
        assert np.abs(nlp.link_generators()[b] - one.link_generators()[0]).max() <= 1e-12
        for which, cnt in (("link", nlp.J), ("torque", nlp.n)):
            for i in range(cnt):
                for t in (0, 49, 50, 99):
                    c1, i1, k1, co1 = nlp.pz(which, i, t, b=b)
                    c2, i2, k2, co2 = one.pz(which, i, t)
                    assert np.array_equal(k1, k2) and np.array_equal(co1, co2) and np.array_equal(c1, c2), (b, which, i, t)
                    assert np.abs(i1 - i2).max() <= 1e-12
        one.close()
    nlp.close()


def test_build_info_names_the_kernel_that_built_the_tables():
    """armour_get_build_info: a lone problem is built step by step by four-wave blocks in one launch, a batch of 18 time-vectorised by
    four-wave blocks in one launch -- for the 8-link Kinova and for the 9-link Fetch alike (a slot pool sized for 8 links once sent every
    Fetch batch through a failed time-vectorised launch and then down the per-step path, silently: the tables were right, the build slow)."""
    from armour_amd.planner import ArmourNLP, default_params, fetch_robot
    from armour_amd.worlds import random_batch
    T, O = 100, 2
    fp = [_fetch_problem(40 + b, O) for b in range(18)]
    fb = {k: np.stack([p[k] for p in fp]) for k in fp[0]}
    kb = random_batch(40, 18, O)
    for make, bp in ((lambda: ArmourNLP(T=T), kb), (lambda: ArmourNLP(robot=fetch_robot(0.5), params=default_params(T)), fb)):
        one = make().set_parameters(bp["q0"][0], bp["qd0"][0], bp["qdd0"][0], bp["q_des"][0], bp["obstacles"][0])
        assert one.build_info() == {"kernel": "per_step", "waves": 4, "sort_entries": 4096, "launches": 1}
        nlp = make().set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
        assert nlp.build_info() == {"kernel": "time_vectorised", "waves": 4, "sort_entries": 4096, "launches": 1}
        # ... and the batch's tables are the lone problem's
        for which, cnt in (("link", nlp.J), ("torque", nlp.n)):
            for i in range(cnt):
                for t in (0, 50, 99):
                    c1, i1, k1, co1 = nlp.pz(which, i, t, b=0)
                    c2, i2, k2, co2 = one.pz(which, i, t)
                    assert np.array_equal(k1, k2) and np.array_equal(co1, co2) and np.array_equal(c1, c2), (which, i, t)
                    assert np.abs(i1 - i2).max() <= 1e-12
        one.close(); nlp.close()


@pytest.mark.parametrize("threshold", [3e-4, 5e-3, 3e-2])
def test_other_simplify_thresholds_against_the_oracle(threshold):
    """The default threshold (5e-4, RT/Parameters.h) prunes the error terms of most rotations and nothing else of the JRS; a smaller one keeps more
    of them, large ones also prune velocity terms and k-dependent rotation terms -- the branches of the closed-form JRS constructors
    (p1_reach.hip jrs_*_direct, p1_tv.inc.h jrs_*_direct_tv) that the default never takes.  Both kernels against the oracle."""
    from armour_amd import _lib
    from armour_amd.planner import ArmourNLP, default_params
    from armour_amd.worlds import random_batch
    from oracle.cpu_oracle import Oracle
    from oracle.cpu_oracle import default_params as oracle_params
    T, O, B = 24, 2, 3
    pd, po = default_params(T), oracle_params(T)
    pd.simplify_threshold = threshold; po.simplify_threshold = threshold
    bp = None
    for seed in range(900, 930):   # a batch none of whose pruning decisions sits on the threshold (the oracle reports the closest one)
        cand = random_batch(seed, B, O)
        os_ = [Oracle(params=po).set_problem(cand["q0"][b], cand["qd0"][b], cand["qdd0"][b], cand["q_des"][b], cand["obstacles"][b]) for b in range(B)]
        if min(o.min_margin() for o in os_) > 1e-9:
            bp = cand
            break
    assert bp is not None
    for build in (1, 2):   # per time step, time-vectorised
        nlp = ArmourNLP(params=pd).set_option(_lib.OPT_P1_BUILD, build).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
        assert nlp.build_info()["kernel"] == ("per_step" if build == 1 else "time_vectorised")
        _compare_tables(nlp, os_)
        nlp.close()


def test_two_handles_from_two_host_threads():
    """One handle = one stream; handles are independent (include/armour_hip.h): two host threads building and evaluating
    different worlds at the same time get the results of serial runs."""
    import threading
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_k, random_problem
    T = 50
    probs = [random_problem(60 + i, 3 + i) for i in range(2)]
    ks = random_k(8, 6)
    serial = []
    for p in probs:
        nlp = ArmourNLP(T=T).set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
        serial.append([nlp.eval_g_jac(k) for k in ks])
    out = [None, None]

    def work(i):
        nlp = ArmourNLP(T=T)
        res = []
        for rep in range(3):
            nlp.set_parameters(probs[i]["q0"], probs[i]["qd0"], probs[i]["qdd0"], probs[i]["q_des"], probs[i]["obstacles"])
            res = [nlp.eval_g_jac(k) for k in ks]
        out[i] = res

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    for i in range(2):
        for (g, j), (g0, j0) in zip(out[i], serial[i]):
            assert np.array_equal(g, g0) and np.array_equal(j, j0)


def test_non_finite_inputs_are_refused(sample_problem):
    """The reference parses whatever armour.in holds; here a NaN / inf state, goal or obstacle is an EINVAL, not NaN rows."""
    from armour_amd._lib import ArmourError
    from armour_amd.planner import ArmourNLP
    p = sample_problem
    nlp = ArmourNLP(T=100)
    for key, val in (("q0", np.nan), ("qd0", np.inf), ("q_des", -np.inf), ("obstacles", np.nan)):
        bad = {k: np.array(v, dtype=np.float64, copy=True) for k, v in p.items()}
        bad[key].reshape(-1)[3] = val
        with pytest.raises(ArmourError):
            nlp.set_parameters(bad["q0"], bad["qd0"], bad["qdd0"], bad["q_des"], bad["obstacles"])
    nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])   # the handle is still usable
    assert np.isfinite(nlp.eval_g(np.zeros(7))).all()


@pytest.mark.parametrize("B,O", [(1, 20), (16, 20), (48, 10)])
def test_builds_are_reproducible_from_launch_to_launch(B, O):
    """The same batch built five times on fresh handles gives the same bits every time, in the three launch shapes (three waves per
    item; one wave per item with three blocks per CU; the two-pass build).  A development build with two waves per SIMD failed
    exactly this, about every second launch (profiles/r02_p1_two_waves_per_simd.txt) -- the shipped object must never."""
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch, random_k
    bp = random_batch(7, B, O)
    ks = random_k(3, B)
    first = None
    for _ in range(5):
        nlp = ArmourNLP(T=100).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
        got = (nlp.torque_radius().copy(), nlp.link_generators().copy()) + tuple(a.copy() for a in nlp.eval_g_jac(ks))
        first = first or got
        for a, b in zip(got, first):
            assert np.array_equal(a, b)


def _tables_digest(B, options, T=100, robot=None):
    """Digest of everything a reach-set build leaves behind, for a handle with the given launch-shape options
    ({option id: value}); returns (sha256 hex, exact arrays, radii arrays)."""
    import hashlib
    from armour_amd.planner import ArmourNLP, default_params
    from armour_amd.worlds import random_batch, random_k
    O = 3
    bp = random_batch(900, B, O)
    bp["qd0"][B - 1] = 0.9 * np.array([1.3963, 1.3963, 1.3963, 1.3963, 1.2218, 1.2218, 1.2218])
    if robot is not None:   # a shorter chain: the first joints' states
        for k in ("q0", "qd0", "qdd0", "q_des"):
            bp[k] = np.ascontiguousarray(bp[k][:, :robot.num_factors])
    nlp = ArmourNLP(T=T) if robot is None else ArmourNLP(robot=robot, params=default_params(T))
    for opt, val in options.items():
        nlp.set_option(opt, val)
    nlp.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    h = hashlib.sha256()
    g, jac = nlp.eval_g_jac(random_k(4, B)[:, :nlp.n])
    for a in (g, jac, nlp.torque_radius(), nlp.link_generators()):
        h.update(np.ascontiguousarray(a).tobytes())
    exact, radii = [], [nlp.torque_radius().ravel(), nlp.link_generators().ravel(), g.ravel(), jac.ravel()]
    for b in (0, B - 1):
        for which, cnt in (("link", nlp.J), ("torque", nlp.n)):
            for i in range(cnt):
                for t in sorted({0, 37 % T, T // 2, T - 1}):
                    cen, ind, keys, co = nlp.pz(which, i, t, b=b)
                    for a in (cen, ind, keys, co):
                        h.update(np.ascontiguousarray(a).tobytes())
                    exact += [cen.ravel(), keys.astype(np.float64).ravel(), co.ravel()]
                    radii.append(ind.ravel())
    info = nlp.build_info()
    nlp.close()
    return h.hexdigest(), np.concatenate(exact), np.concatenate(radii), info


def _shape(**kw):
    """{ARMOUR_OPT_* id: value} from short names: build, step_waves, step_free, step_split_fk, tv_waves, tv_free, tv_helpers."""
    from armour_amd import _lib
    names = {"build": _lib.OPT_P1_BUILD, "step_waves": _lib.OPT_P1_STEP_WAVES, "step_free": _lib.OPT_P1_STEP_FREE, "step_split_fk": _lib.OPT_P1_STEP_SPLIT_FK,
             "tv_waves": _lib.OPT_P1_TV_WAVES, "tv_free": _lib.OPT_P1_TV_FREE, "tv_helpers": _lib.OPT_P1_TV_HELPERS,
             "step_queue": _lib.OPT_P1_STEP_QUEUE, "step_pairs": _lib.OPT_P1_STEP_PAIRS, "step_tail_cross": _lib.OPT_P1_STEP_TAIL_CROSS, "tv_tail_cross": _lib.OPT_P1_TV_TAIL_CROSS}
    return {names[k]: v for k, v in kw.items()}


@pytest.mark.parametrize("B,settings", [
    (1, [dict(build=1, step_free=1), dict(build=1, step_free=0), dict(build=1, step_waves=1), dict(build=1, step_free=1, step_split_fk=0),
         dict(build=1, step_pairs=0), dict(build=1, step_tail_cross=12)]),
    (2, [dict(build=1, step_free=1), dict(build=1, step_free=0), dict(build=1, step_waves=1), dict(build=1, step_pairs=0)]),
    (3, [dict(build=1, step_waves=4, step_queue=0), dict(build=1, step_waves=4), dict(build=1, step_waves=4, step_queue=3), dict(build=1, step_waves=1, step_queue=2),
         dict(build=1, step_waves=4, step_pairs=2)]),
    (3, [dict(build=2, tv_waves=4, tv_helpers=0), dict(build=2, tv_waves=3), dict(build=2, tv_waves=1), dict(build=2, tv_waves=4, tv_free=0), dict(build=2, tv_waves=3, tv_free=0),
         dict(build=2, tv_waves=4, tv_helpers=0, tv_tail_cross=0), dict(build=2, tv_waves=4, tv_helpers=0, tv_tail_cross=13)]),
], ids=["per-step, one problem", "per-step, two problems", "per-step, three problems: more items than blocks", "time-vectorised"])
def test_wave_choreographies_leave_identical_tables(B, settings):
    """Which wave of a block computes what -- one wave playing every role, three roles with a barrier per joint, three or four
    free-running waves with the angular velocity, the torque tables and the helper products dealt out among them
    (p1_free.inc.h) -- must not show in the result: every shape runs the same operators on the same operands.  The digest of
    g, jac, torque radii, link generators and sampled link / torque PZs is equal across the shapes of one kernel, bit for bit.
    Round 4 adds two more ways of dealing the work out, with the same claim: two waves on every operator of the backward pass of a per-step
    four-wave block (ARMOUR_OPT_P1_STEP_PAIRS: a reduce pass adds its pruned amounts as the same two partial sums on one wave as on two), and
    w x (w_aux x com) of the last links built by a wave that is through with its recursion (ARMOUR_OPT_P1_*_TAIL_CROSS).
    And which block builds which item (ARMOUR_OPT_P1_STEP_QUEUE: dealt out by index, or drawn from a counter in one of three orders).
    Round 4: the shapes are per-handle options (include/armour_hip.h, ARMOUR_OPT_P1_*), so all of them run in THIS process, one
    handle each -- they were environment variables read once per process, and every shape needed a process of its own."""
    digests, shapes = [], []
    for kw in settings:
        d, _, _, info = _tables_digest(B, _shape(**kw))
        digests.append(d); shapes.append((info["kernel"], info["waves"]))
    assert len(set(digests)) == 1, list(zip(settings, digests))
    assert len(set(shapes)) > 1, shapes   # (the options did select different launch shapes)


@pytest.mark.parametrize("T", [100, 90, 128, 40])
def test_row_widths_leave_identical_tables(T):
    """Round 5: the time-vectorised kernel exists twice -- rows of 64 doubles (p1_reach.hip) and packed rows of 50 (p1_reach_tv50.hip: T = 100
    makes groups of 50 time steps) -- and armour_p1_build picks by the group size (ARMOUR_OPT_P1_TV_ROW_WIDTH holds a handle to 64).  A row's
    width changes addresses, never a value or an order: every bit of every table is the same, in four-, three- and one-wave blocks (the one-wave
    kernel with narrow rows is the configuration that faulted in every launch until the return-address guard of pz_wave.h: DESIGN.md 7); for groups
    that do not fill a row of 50 (T = 90: 45 steps; T = 40) -- lanes with a place in the rows but no step of their own shadow another step
    (pz_tv.h) -- and with T = 128 (groups of 64: only the wide rows fit, asking for 50 is refused by falling back to 64)."""
    from armour_amd import _lib
    digests = []
    for rows in (0, 64):
        for waves in (4, 3, 1):
            d, _, _, info = _tables_digest(3, {**_shape(build=2, tv_waves=waves, tv_helpers=0), _lib.OPT_P1_TV_ROW_WIDTH: rows}, T=T)
            assert info["kernel"] == "time_vectorised", info
            digests.append(d)
    assert len(set(digests)) == 1, digests
    # ... and both agree with the per-step kernel: exact parts bit for bit, radii to the build contract's 1e-12
    _, ex_tv, rad_tv, _ = _tables_digest(3, _shape(build=2, tv_waves=4), T=T)
    _, ex_ps, rad_ps, _ = _tables_digest(3, _shape(build=1), T=T)
    assert np.array_equal(ex_tv, ex_ps)
    assert np.abs(rad_tv - rad_ps).max() <= 1e-12


def test_tail_cross_options_on_a_three_joint_chain():
    """ARMOUR_OPT_P1_*_TAIL_CROSS moved w x (w_aux x com) of the last n links to another wave with n clamped to 4 but not to the
    chain: on a robot of J < 4 joints the loop started at link J - n + 1 <= 0 and read states that are not the links'.  The tail is now held to
    the links the tail moments may take (none for J < 4), and set_option refuses values without a meaning.  A three-joint chain (the first
    three joints of the Kinova): every accepted value leaves the tables of the default bit for bit, in both kernels."""
    from armour_amd import _lib
    from armour_amd.planner import ArmourNLP, kinova_robot
    rb = kinova_robot()
    rb.num_joints = 3; rb.num_factors = 3
    ref = {}
    for build, opt in ((1, "step_tail_cross"), (2, "tv_tail_cross")):
        for val in (0, 1, 4, 11, 14):
            kw = dict(build=build, step_waves=4) if build == 1 else dict(build=2, tv_waves=4, tv_helpers=0)
            kw[opt] = val
            d, _, _, _ = _tables_digest(2, _shape(**kw), robot=rb)
            ref.setdefault(build, d)
            assert d == ref[build], (build, val)
    nlp = ArmourNLP(T=100)
    for bad in (5, 9, 15, 99, 600):
        with pytest.raises(Exception):
            nlp.set_option(_lib.OPT_P1_TV_TAIL_CROSS, bad)
    nlp.set_option(_lib.OPT_P1_TV_TAIL_CROSS, 213)   # (the development form: 100 * (n + 1) + links)
    nlp.close()


def test_walk_helpers_only_reorder_the_sums_of_the_radii():
    """Round 3: in the backward pass of a four-wave time-vectorised block the two idle waves each walk the upper part of the
    f- / n-recursion's sorted raw terms (pz_tv.h "One walk on two waves").  Against the same build with every walk on its own wave:
    centres, monomial keys and coefficients equal bit for bit; the independent radii -- and torque radius, link generators, g, jac
    through them -- differ by the rounding of two partial sums instead of one running sum, <= 1e-12.  And the shared walk is
    deterministic: two handles give the same bits.  (Two handles of one process with different ARMOUR_OPT_P1_TV_HELPERS.)"""
    out = {}
    for tag, kw in (("off", dict(build=2, tv_waves=4, tv_helpers=0)), ("on", dict(build=2, tv_waves=4)), ("on2", dict(build=2, tv_waves=4))):
        _, exact, radii, _ = _tables_digest(3, _shape(**kw))
        out[tag] = {"exact": exact, "radii": radii}
    assert np.array_equal(out["on"]["exact"], out["off"]["exact"])
    assert np.abs(out["on"]["radii"] - out["off"]["radii"]).max() <= 1e-12
    assert not np.array_equal(out["on"]["radii"], out["off"]["radii"])        # (the helpers did take part)
    assert np.array_equal(out["on"]["radii"], out["on2"]["radii"]) and np.array_equal(out["on"]["exact"], out["on2"]["exact"])


def test_two_handles_with_different_options_in_one_process():
    """A stray setting of one handle must not reach another: two handles alive at the same time, one held to the per-step kernel with one-wave
    blocks, one to the time-vectorised kernel, built alternately -- each keeps reporting its own kernel, and each reproduces its own tables."""
    from armour_amd import _lib
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch
    bp = random_batch(77, 4, 2)
    a = ArmourNLP(T=100).set_option(_lib.OPT_P1_BUILD, 1).set_option(_lib.OPT_P1_STEP_WAVES, 1)
    b = ArmourNLP(T=100).set_option(_lib.OPT_P1_BUILD, 2)
    seen = {}
    for rnd in range(2):
        for name, nlp in (("a", a), ("b", b)):
            nlp.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
            info = nlp.build_info()
            assert info["kernel"] == ("per_step" if name == "a" else "time_vectorised") and (info["waves"] == 1 if name == "a" else info["waves"] >= 3), (name, info)
            tabs = (nlp.torque_radius().copy(), nlp.link_generators().copy())
            if name in seen:
                assert all(np.array_equal(x, y) for x, y in zip(tabs, seen[name]))
            seen[name] = tabs
    assert np.abs(seen["a"][0] - seen["b"][0]).max() <= 1e-12 and np.abs(seen["a"][1] - seen["b"][1]).max() <= 1e-12
    with pytest.raises(Exception):
        a.set_option(_lib.OPT_P1_TV_WAVES, 5.5)   # not an integer
    with pytest.raises(Exception):
        a.set_option(999, 1)
    a.close(); b.close()


def _exact_plane_skip(gens, obstacles):
    """numpy restatement of the row test behind plane_skip (p1_reach.hip planes_redundant): per row the 36 pair normals exactly as
    RT/CollisionChecking.cu:176-190 computes them (cross product, norm, three divisions -- IEEE operations, the same bits), a plane
    redundant iff its normal is zero or equals an earlier plane's up to the sign; AND over all rows."""
    T, J = gens.shape[0], gens.shape[1]
    pairs = [(a, b) for a in range(9) for b in range(a + 1, 9)]
    mask = (1 << 36) - 1
    for ob in np.asarray(obstacles).reshape(-1, 12):
        for t in range(T):
            for l in range(J):
                G = [ob[3 * (g + 1):3 * (g + 1) + 3] for g in range(3)] + [gens[t, l, :, g] for g in range(6)]
                seen, row = [], 0
                for p, (a, b) in enumerate(pairs):
                    ga, gb = G[a], G[b]
                    cr = np.array([ga[1] * gb[2] - ga[2] * gb[1], ga[2] * gb[0] - ga[0] * gb[2], ga[0] * gb[1] - ga[1] * gb[0]])
                    nrm = np.sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2])
                    C = cr / nrm if nrm > 0 else np.zeros(3)
                    C = np.where(C == 0, 0.0, C)
                    nz = [x for x in C if x != 0]
                    key = tuple(-C if (nz and nz[0] < 0) else C)
                    if not nz or key in seen:
                        row |= 1 << p
                    seen.append(key)
                mask &= row
    return mask


@pytest.mark.gpu
def test_plane_skip_masks_equal_the_exact_row_test():
    """The lean half-space table (round 3) derives plane_skip from a class pre-pass plus the exact test on sampled rows, and runs the
    full signature test only where those leave a plane open.  Whatever the path, the mask must be what the exact row test gives on
    the device's own link generators: box worlds (12 of 36 planes, every block takes the shortcut), a world with a rotated obstacle
    (the shortcut still settles it: fewer planes drop out), one with a flat box (a zero generator)."""
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_problem
    T = 20
    box = random_problem(91, 4)
    c, s = np.cos(0.4), np.sin(0.4)
    rot = random_problem(92, 3)
    Rz = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    rot["obstacles"][1, 3:] = (Rz @ rot["obstacles"][1, 3:].reshape(3, 3).T).T.reshape(-1)      # obstacle 1: generators rotated about z
    flat = random_problem(93, 3)
    flat["obstacles"][0, 9:12] = 0.0                                                             # obstacle 0: no extent along z
    expected_counts = {}
    for name, p in (("box", box), ("rotated", rot), ("flat", flat)):
        nlp = ArmourNLP(T=T).set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
        got = int(nlp.plane_skip()[0])
        want = _exact_plane_skip(nlp.link_generators()[0], p["obstacles"])
        assert got == want, (name, hex(got), hex(want))
        expected_counts[name] = bin(got).count("1")
        nlp.close()
    assert expected_counts["box"] == 12 and expected_counts["rotated"] < 12
    # a batch: one mask per problem, each equal to its single-problem handle's
    probs = [box, random_problem(94, 4), random_problem(95, 4)]
    st = {k: np.stack([q[k] for q in probs]) for k in box}
    nb = ArmourNLP(T=T).set_parameters(st["q0"], st["qd0"], st["qdd0"], st["q_des"], st["obstacles"])
    for b, q in enumerate(probs):
        assert int(nb.plane_skip()[b]) == _exact_plane_skip(nb.link_generators()[b], q["obstacles"])
    nb.close()
