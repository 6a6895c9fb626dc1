"""CPU-side checks of the drop-in boundary: libarmour_hip.so loads, exports every symbol include/armour_hip.h
declares, and refuses to run without a device (there is no CPU path)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "armour_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(armour_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from armour_amd import _lib
    L = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), f"libarmour_hip.so does not export {name}"
    assert sorted(_lib.EXPORTS) == declared


def test_struct_layout_matches_header():
    from armour_amd import _lib
    # sizes as the C compiler lays them out: 2+7+7 int32, then doubles
    J, F = 9, 7   # ARMOUR_MAX_JOINTS, ARMOUR_MAX_FACTORS
    n_doubles = (J + 1) * 3 + J * 3 + J + 1 + J * 3 + J * 9 + 1 + J * 3 + F * 4 + 1 + J * 3 + J * 3 + 5 + 2 * J   # ... + the two per-link uncertainty arrays
    assert C.sizeof(_lib.ArmourRobot) == (2 + J + F) * 4 + n_doubles * 8
    assert C.sizeof(_lib.ArmourParams) == 8 + 8 * (1 + 7 + 5)
    assert C.sizeof(_lib.ArmourLimits) == 24


def test_presets_match_reference_constants():
    from armour_amd import planner
    r = planner.kinova_robot()
    assert (r.num_joints, r.num_factors) == (7, 7)
    assert list(r.axes)[:7] == [3] * 7
    assert abs(r.armature[1] - 11.9962024615303644) < 1e-15
    assert abs(r.trans[2] - 0.15643) < 1e-15 and abs(r.trans[3 * 6 + 1] + 0.10593) < 1e-15
    assert r.torque_limits[4] == 29.4 and r.state_limits_ub[3] == 2.66
    rg = planner.kinova_gripper_robot()   # RT/KinovaInfo.h
    assert (rg.num_joints, rg.num_factors, rg.axes[7]) == (8, 7, 0) and rg.mass[7] == 1.72 and rg.K == 10.0
    assert abs(rg.trans[7 * 3 + 2] + 0.163075) < 1e-15 and rg.link_zonotope_generators[7 * 3 + 1] == 0.09
    p = planner.default_params(100)
    assert p.num_time_steps == 100 and p.simplify_threshold == 5e-4 and p.t_plan == 0.5


def test_no_cpu_path_without_device():
    from armour_amd import _lib, planner
    L = _lib.load()
    if L.armour_device_available():
        pytest.skip("a GPU is visible here")
    with pytest.raises(_lib.ArmourError) as ei:
        planner.ArmourNLP(T=100)
    assert ei.value.code == _lib.EDEVICE
    assert "no CPU path" in str(ei.value)
