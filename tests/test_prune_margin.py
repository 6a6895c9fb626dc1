"""armour_get_prune_margin: how close a reach-set build came to a prune flip, reported by the device.

Every stated tolerance against the reference's arithmetic is conditional on "no monomial norm within ~1e-9 of SIMPLIFY_THRESHOLD"
(RT/PZsparse.cu:305-316), and until round 5 only the CPU oracle could tell (Oracle.min_margin()).  Both reach-set kernels now
keep, per lane, the largest pruned and the smallest kept squared norm of every simplify() verdict and reduce them per problem; here the
device's figure is held against the oracle's on the same problems -- per-step kernel, time-vectorised kernel, both planner modes, the
reference's own worlds -- and a batch gets its figure for EVERY world, not for the six the oracle is run on."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_pz_ops import margin_agrees   # dev in [kept-side figure of ref - 2e-10, ref + 2e-10]: the device's coefficients differ from the oracle's by
                                        # <= 3e-15 (summation order of equal keys), i.e. 3e-15 / 5e-4 on the margin; include/armour_hip.h says what else differs


def _oracle_margin(T, p, **kw):
    from oracle.cpu_oracle import Oracle
    return Oracle(T=T, **kw).set_problem(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"]).min_margin()


@pytest.mark.parametrize("T", [100, 128, 40])
def test_device_margin_equals_the_oracles_on_single_problems(T):
    """One problem per handle: the per-step kernel (four-wave blocks, pairs in the backward pass, forward kinematics as items of their own)."""
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_problem, reference_sample_problem
    nlp = ArmourNLP(T=T)
    for p in [reference_sample_problem()] + [random_problem(s, 3) for s in (1, 2, 3, 4)]:
        nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
        assert nlp.build_info()["kernel"] == "per_step"
        m = nlp.prune_margin()
        ref = _oracle_margin(T, p)
        assert m.shape == (1,) and margin_agrees(m[0], ref), (T, m[0], ref)
    nlp.close()


@pytest.mark.parametrize("build,waves", [(1, 0), (1, 1), (1, 3), (2, 0), (2, 1)], ids=["per_step", "per_step_1wave", "per_step_3waves", "time_vectorised", "time_vectorised_1wave"])
def test_device_margin_per_world_of_a_batch_in_every_launch_shape(build, waves):
    """24 worlds through both kernels and their block shapes: every world's margin equals the oracle's (and so does not depend on the shape)."""
    from armour_amd import _lib
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch
    T, B, O = 100, 24, 4
    bp = random_batch(400, B, O)
    nlp = ArmourNLP(T=T)
    nlp.set_option(_lib.OPT_P1_BUILD, build)
    if waves:
        nlp.set_option(_lib.OPT_P1_STEP_WAVES if build == 1 else _lib.OPT_P1_TV_WAVES, waves)
    nlp.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    assert nlp.build_info()["kernel"] == ("per_step" if build == 1 else "time_vectorised")
    m = nlp.prune_margin()
    for b in range(B):
        ref = _oracle_margin(T, {k: bp[k][b] for k in ("q0", "qd0", "qdd0", "q_des", "obstacles")})
        assert margin_agrees(m[b], ref), (b, m[b], ref)
    nlp.close()


def test_margin_of_the_reference_worlds_and_of_the_other_robots():
    from armour_amd.planner import ArmourNLP, default_params, fetch_robot, kinova_gripper_robot
    from armour_amd.scenes import as_batch, reference_worlds
    from armour_amd.worlds import random_fetch_problem, random_problem
    from oracle.cpu_oracle import Oracle
    from oracle.cpu_oracle import default_params as oparams
    from oracle.cpu_oracle import fetch_robot as ofetch
    from oracle.cpu_oracle import kinova_gripper_robot as ogrip
    T = 100
    ws = reference_worlds()
    bp = as_batch(ws)
    nlp = ArmourNLP(T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    m = nlp.prune_margin()
    assert m.shape == (107,) and np.all(m > 1e-9) and np.all(m < 0.4143)
    for b in range(0, 107, 9):
        ref = _oracle_margin(T, {k: bp[k][b] for k in ("q0", "qd0", "qdd0", "q_des", "obstacles")})
        assert margin_agrees(m[b], ref), (ws[b][0], m[b], ref)
    nlp.close()
    # the Fetch preset (mixed joint axes, 9 links, payload uncertainty) and the Kinova with its gripper (a fixed eighth joint)
    for robot, orobot, prob in ((fetch_robot(0.5), ofetch(0.5), random_fetch_problem(11, 5)), (kinova_gripper_robot(), ogrip(), random_problem(6, 5))):
        nlp = ArmourNLP(robot=robot, params=default_params(T)).set_parameters(prob["q0"], prob["qd0"], prob["qdd0"], prob["q_des"], prob["obstacles"])
        ref = Oracle(robot=orobot, params=oparams(T)).set_problem(prob["q0"], prob["qd0"], prob["qdd0"], prob["q_des"], prob["obstacles"]).min_margin()
        assert margin_agrees(nlp.prune_margin()[0], ref)
        nlp.close()


def test_margin_in_the_modes_without_torque_rows():
    """ARMTD comparison mode and TURN_OFF_INPUT_CONSTRAINTS: forward kinematics only -- fewer verdicts, a margin of their own."""
    from armour_amd.planner import ArmourNLP, default_params
    from armour_amd.worlds import random_problem, synthetic_offline_jrs
    from oracle.cpu_oracle import Oracle
    from oracle.cpu_oracle import default_params as oparams
    T = 100
    p = random_problem(21, 4)
    jrs, k_range = synthetic_offline_jrs(p["qd0"], T)
    nlp = ArmourNLP(T=T).set_parameters_armtd(p["q0"], p["qd0"], p["q_des"], jrs, k_range, p["obstacles"])
    ref = Oracle(T=T).set_problem_armtd(p["q0"], p["qd0"], p["q_des"], jrs, k_range, p["obstacles"]).min_margin()
    assert margin_agrees(nlp.prune_margin()[0], ref)
    nlp.close()
    pr, po = default_params(T), oparams(T)
    pr.input_constraints_off = 1; po.input_constraints_off = 1
    nlp = ArmourNLP(params=pr).set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    ref = Oracle(params=po).set_problem(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"]).min_margin()
    assert margin_agrees(nlp.prune_margin()[0], ref)
    nlp.close()


def test_loaded_tables_have_no_margin():
    from armour_amd._lib import ArmourError
    from armour_amd.planner import ArmourNLP
    from helpers import SAMPLE_PROBLEM, oracle_tables
    from oracle.cpu_oracle import Oracle
    p = SAMPLE_PROBLEM
    o = Oracle(T=10).set_problem(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    nlp = ArmourNLP(T=10)
    nlp.debug_load_tables(p["q0"], p["qd0"], p["qdd0"], p["q_des"], oracle_tables([o]))
    with pytest.raises(ArmourError):
        nlp.prune_margin()
    nlp.close()
