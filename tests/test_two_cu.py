"""A lone problem's time step on TWO compute units (round 6; armour_amd/csrc/p1_free.inc.h, ARMOUR_OPT_P1_STEP_TWO_CU / _LEAN_BACK).

VERDICT round 5, item 1(a): the reach-set build of one problem runs T four-wave blocks on 256 CUs; a second block per time step -- the helper --
reruns the two angular-velocity recursions of RT/Dynamics.cu:96-121, builds the cross products that read nothing else, the step's forward
kinematics and (level 3) the f-recursion of the backward pass (:150-170), and hands its products to the main block through the L2 of the XCD
the two share.  The claim tested here: WHICH block builds a product does not show -- every level leaves the tables of the one-CU build bit
for bit (keys, coefficients, centres, radii, prune margin), on every robot, with and without paired waves, at every T for which 2 T + 7 blocks
fit the device -- and a helper that never delivers is noticed: the build starts again on one CU per step and the handle stays there."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_p1_parity import _tables_digest, _shape


def _opts(level, lean=1, **kw):
    from armour_amd import _lib
    return {**_shape(build=1, **kw), _lib.OPT_P1_STEP_TWO_CU: level, _lib.OPT_P1_STEP_LEAN_BACK: lean}


@pytest.mark.parametrize("T", [100, 128, 40])
def test_every_level_leaves_the_tables_of_the_one_cu_build(T):
    ref, ex0, rad0, info0 = _tables_digest(1, _opts(0), T=T)
    assert info0 == {"kernel": "per_step", "waves": 4, "sort_entries": info0["sort_entries"], "launches": 1}
    # (lean: bit 0 the f-recursion on the helper, bit 1 set = every joint's JRS before the roles begin; T = 128: 2 T = 256 blocks, the whole device)
    for level, lean, kw in [(1, 1, {}), (2, 1, {}), (3, 0, {}), (3, 1, {}), (3, 2, {}), (3, 3, {}), (3, 1, dict(step_pairs=0)), (3, 1, dict(step_tail_cross=300)), (2, 1, dict(step_pairs=0))]:
        d, ex, rad, info = _tables_digest(1, _opts(level, lean, **kw), T=T)
        assert info["launches"] == 1 and info["waves"] == 4, (level, lean, kw, info)   # (no fall-back: the helper delivered)
        assert np.array_equal(ex, ex0) and np.array_equal(rad, rad0) and d == ref, (T, level, lean, kw)


def test_two_cus_on_the_other_robots():
    """Kinova with the gripper, the Fetch preset (nine links: more than level 3 takes, the build holds itself to level 2) and a three-joint chain."""
    from armour_amd.planner import fetch_robot, kinova_gripper_robot, kinova_robot
    short = kinova_robot()
    short.num_joints = 3; short.num_factors = 3
    for name, robot in (("gripper", kinova_gripper_robot()), ("fetch", fetch_robot(0.5)), ("three joints", short)):
        digests = {lvl: _tables_digest(1, _opts(lvl), T=100, robot=robot)[0] for lvl in (0, 3)}
        assert digests[0] == digests[3], name


def test_the_prune_margin_does_not_depend_on_the_level():
    from armour_amd import _lib
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_problem, reference_sample_problem
    for p in (reference_sample_problem(), random_problem(7, 5)):
        margins = []
        for level in (0, 1, 2, 3):
            nlp = ArmourNLP(T=100)
            nlp.set_option(_lib.OPT_P1_STEP_TWO_CU, level)
            nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
            margins.append(float(nlp.prune_margin()[0]))
            nlp.close()
        assert len(set(margins)) == 1, margins


def test_two_cus_against_the_oracle():
    """The default build of a lone problem (two CUs, level 3, the f-recursion on the helper) against the CPU restatement: identical key sets of
    every link / torque PZ at every time step, tables, g and Jacobian to the tolerances of tests/test_p1_parity.py."""
    from test_p1_parity import _compare_tables, G_TOL, J_TOL
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_k, random_problem
    from oracle.cpu_oracle import Oracle
    T = 100
    for seed, O in ((21, 6), (22, 0)):
        p = random_problem(seed, O)
        nlp = ArmourNLP(T=T).set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
        o = Oracle(T=T).set_problem(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
        _compare_tables(nlp, [o])
        k = random_k(5, 1)
        g, jac = nlp.eval_g_jac(k)
        go, jo = o.eval_g_jac(k[0])
        assert np.abs(g[0] - go).max() <= G_TOL and np.abs(jac[0] - jo).max() <= J_TOL
        nlp.close()


def test_a_helper_that_never_delivers_is_noticed():
    """ARMOUR_OPT_P1_STEP_TWO_CU = 10 + level: the helper blocks agree to help and publish nothing.  Every take of the main blocks runs into its
    cut-off, the launch reports ERR_HELPER, armour_set_problems builds again on one CU per time step -- same tables -- and the handle stays there."""
    from armour_amd import _lib
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_problem
    p = random_problem(3, 4)
    ref = ArmourNLP(T=100)
    ref.set_option(_lib.OPT_P1_STEP_TWO_CU, 0)
    ref.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    for level in (13, 11):
        nlp = ArmourNLP(T=100)
        nlp.set_option(_lib.OPT_P1_STEP_TWO_CU, level)
        nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
        assert nlp.build_info()["launches"] == 2, nlp.build_info()
        assert np.array_equal(nlp.torque_radius(), ref.torque_radius()) and np.array_equal(nlp.link_generators(), ref.link_generators())
        for i in range(nlp.n):
            for a, b in zip(nlp.pz("torque", i, 50), ref.pz("torque", i, 50)):
                assert np.array_equal(a, b)
        assert nlp.get_option(_lib.OPT_P1_STEP_TWO_CU) == 0          # (the caller can see where the handle is)
        nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
        assert nlp.build_info()["launches"] == 1, nlp.build_info()   # (one CU per step from the start)
        assert np.array_equal(nlp.torque_radius(), ref.torque_radius())
        nlp.close()
    ref.close()


def test_a_helper_that_starts_late_costs_one_slow_build():
    """ARMOUR_OPT_P1_STEP_TWO_CU = 20 + level: the helper blocks never say they have started -- what a main block sees when the device is shared or
    holds fewer CUs than it reports and its helper is queued behind other work.  Each main block gives up after ~1.5 ms and builds its item alone
    (the helper, whenever it runs, still does the forward kinematics): ONE launch, good tables, and the handle stays on one CU per step until the
    option is set again."""
    from armour_amd import _lib
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_problem
    p = random_problem(4, 6)
    ref = ArmourNLP(T=100)
    ref.set_option(_lib.OPT_P1_STEP_TWO_CU, 0)
    ref.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    nlp = ArmourNLP(T=100)
    nlp.set_option(_lib.OPT_P1_STEP_TWO_CU, 23)
    nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    assert nlp.build_info()["launches"] == 1 and nlp.build_ms > 1.5, (nlp.build_info(), nlp.build_ms)
    assert np.array_equal(nlp.torque_radius(), ref.torque_radius()) and np.array_equal(nlp.link_generators(), ref.link_generators())
    assert nlp.get_option(_lib.OPT_P1_STEP_TWO_CU) == 0
    nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    assert nlp.build_ms < 1.5 and np.array_equal(nlp.torque_radius(), ref.torque_radius())
    nlp.set_option(_lib.OPT_P1_STEP_TWO_CU, 3)                       # two CUs again
    nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    assert nlp.get_option(_lib.OPT_P1_STEP_TWO_CU) == 3 and nlp.build_ms < 1.5
    assert np.array_equal(nlp.torque_radius(), ref.torque_radius()) and np.array_equal(nlp.link_generators(), ref.link_generators())
    nlp.close(); ref.close()



def test_blocks_of_an_item_on_different_xcds_build_alone():
    """ARMOUR_OPT_P1_STEP_TWO_CU = 30 + level: block helper0 + k helps item k + 1, so the two blocks of every item sit on different XCDs -- a placement HIP
    is free to choose, under which a plain store of one block is not visible to the other through an L2.  Both publish their XCC_ID at the start of the item;
    every main block must decide against two CUs and build its item alone (the helper keeps the forward kinematics, and with level 3 the joints past the
    fourth get their JRS the way a one-block build deals them): ONE launch, the one-CU tables bit for bit, and no fall-back of the handle -- the placement of
    the next launch may be another."""
    from armour_amd import _lib
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_problem
    p = random_problem(6, 8)
    ref = ArmourNLP(T=100)
    ref.set_option(_lib.OPT_P1_STEP_TWO_CU, 0)
    ref.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    for level in (33, 32, 31):
        nlp = ArmourNLP(T=100)
        nlp.set_option(_lib.OPT_P1_STEP_TWO_CU, level)
        for _ in range(2):
            nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
            assert nlp.build_info()["launches"] == 1 and nlp.get_option(_lib.OPT_P1_STEP_TWO_CU) == level
            assert np.array_equal(nlp.torque_radius(), ref.torque_radius()) and np.array_equal(nlp.link_generators(), ref.link_generators())
            for i in range(nlp.n):
                for a, b in zip(nlp.pz("torque", i, 63), ref.pz("torque", i, 63)):
                    assert np.array_equal(a, b)
            for i in range(nlp.J):
                for a, b in zip(nlp.pz("link", i, 17), ref.pz("link", i, 17)):
                    assert np.array_equal(a, b)
        assert float(nlp.prune_margin()[0]) == float(ref.prune_margin()[0])
        nlp.close()
    ref.close()
