"""The product's robot presets (include/armour_robot_kinova.h, include/armour_robot_fetch.h, hand-written) against the
oracle's (oracle/robot_tables.hpp, GENERATED from the reference headers by oracle/gen_robot_tables.py) -- field by field,
exact -- and, where /root/reference exists (the build container), both against a fresh parse of
RT/KinovaWithoutGripperInfo.h, RT/KinovaInfo.h, CMP/FetchInfo.h and RT/Parameters.h.  Closes the round-1 hole where the
oracle included the product's own constant header."""
import ctypes as C
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))
PRESETS = {"kinova_gen3_no_gripper": "oracle_fill_kinova", "kinova_gen3_gripper": "oracle_fill_kinova_gripper", "fetch": "oracle_fill_fetch"}


def _product(name):
    from armour_amd import _lib
    r = _lib.ArmourRobot()
    getattr(_lib.load(), f"armour_robot_{name}")(C.byref(r))
    return r


def _oracle(name):
    from oracle import cpu_oracle
    r = cpu_oracle.ArmourRobot()
    getattr(cpu_oracle.lib(), PRESETS[name])(C.byref(r))
    return r


@pytest.mark.parametrize("name", list(PRESETS))
def test_product_preset_equals_oracle_table(name):
    import gen_robot_tables as g
    a, b = g.robot_from_struct(_product(name)), g.robot_from_struct(_oracle(name))
    assert g.diff(a, b) == []
    # the whole struct, padding included (both sides memset first): nothing beyond the compared fields differs either
    assert bytes(_product(name)) == bytes(_oracle(name))


def test_default_parameters_equal():
    from armour_amd import planner
    from oracle import cpu_oracle
    for T in (100, 128):
        assert bytes(planner.default_params(T)) == bytes(cpu_oracle.default_params(T))


@pytest.mark.skipif(not os.path.exists("/root/reference/kinova_src"), reason="reference checkout not present (GPU box)")
def test_both_against_the_reference_headers():
    import gen_robot_tables as g
    robots = {name: g.robot_from_header(name) for name in PRESETS}
    assert open(os.path.join(ROOT, "oracle", "robot_tables.hpp")).read() == g.emit_header(robots), "re-run oracle/gen_robot_tables.py"
    for name, ref in robots.items():
        assert g.diff(ref, g.robot_from_struct(_product(name))) == [], name
        assert g.diff(ref, g.robot_from_struct(_oracle(name))) == [], name
    # the only mixed-axis chain of the reference
    assert robots["fetch"]["axes"] == [3, 2, 1, 2, 1, 2, 1, 0, 0] and robots["fetch"]["num_joints"] == 9
    p = g.params_from_reference()
    from armour_amd import planner
    d = planner.default_params(p["num_time_steps_reference"])
    assert list(d.k_range) == p["k_range"] and d.duration == p["duration"] and d.simplify_threshold == p["simplify_threshold"]
    assert d.t_plan == p["t_plan"] and d.cost_scale == p["cost_scale"]
    assert d.collision_violation_threshold == p["collision_violation_threshold"] and d.torque_violation_threshold == p["torque_violation_threshold"]


def test_per_link_uncertainty_overrides_the_scalar():
    """BASELINE configs[4], payload-mass uncertainty: +-50 % on the last link only widens the torque radius (oracle)."""
    import numpy as np
    from oracle.cpu_oracle import Oracle, default_params, fetch_robot
    q0 = np.array([0.3, -0.4, 0.5, 0.9, -0.2, 0.6, 0.1])
    qd0, qdd0 = np.full(7, 0.2), np.zeros(7)
    base = Oracle(robot=fetch_robot(), params=default_params(10)).set_problem(q0, qd0, qdd0, q0, np.zeros((0, 12)))
    pay = Oracle(robot=fetch_robot(0.5), params=default_params(10)).set_problem(q0, qd0, qdd0, q0, np.zeros((0, 12)))
    rb, rp = base.torque_radius(), pay.torque_radius()
    assert np.isfinite(rb).all() and (rp >= rb - 1e-15).all() and (rp > rb + 1e-6).any()
    # same polynomials, different radii only (the uncertainty never feeds a centre or coefficient)
    for j in range(7):
        c0, _, k0, co0 = base.pz("torque", j, 4)
        c1, _, k1, co1 = pay.pz("torque", j, 4)
        assert np.array_equal(k0, k1) and np.array_equal(co0, co1) and np.array_equal(c0, c1)
