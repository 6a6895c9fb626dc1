"""Parity fuzz on the GPU: N random planning problems (both modes), device against the CPU oracle -- monomial key sets,
table values, g and jac at random k -- with the worst deviations and the smallest prune margin seen.  Development /
evidence tool (tests/ holds the fixed cases); the numbers quoted in DESIGN.md section 2 come from here.

    python tools/fuzz_parity.py [N=48] [first_seed=5000]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from armour_amd.planner import ArmourNLP  # noqa: E402
from armour_amd.worlds import SPEED, random_k, random_problem, synthetic_offline_jrs  # noqa: E402
from oracle.cpu_oracle import Oracle  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
S0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
T = int(sys.argv[3]) if len(sys.argv) > 3 else 100
GRIPPER = len(sys.argv) > 4 and sys.argv[4] == "gripper"   # the 8-link arm (fixed gripper joint, RT/KinovaInfo.h)
worst = dict(coef=0.0, cen=0.0, gens=0.0, radius=0.0, planes=0.0, g=0.0, jac=0.0)
min_margin, key_mismatch, low_margin_cases = 1.0, 0, 0
margin_mismatch, flagged = 0, []
t_start = time.time()
if GRIPPER:
    from armour_amd.planner import kinova_gripper_robot
    from oracle.cpu_oracle import kinova_gripper_robot as oracle_gripper
nlp = ArmourNLP(robot=kinova_gripper_robot(), T=T) if GRIPPER else ArmourNLP(T=T)
mk_oracle = (lambda: Oracle(robot=oracle_gripper(), T=T)) if GRIPPER else (lambda: Oracle(T=T))
for s in range(N):
    rng = np.random.default_rng(S0 + s)
    O = int(rng.choice([0, 1, 3, 10, 20, 40]))
    p = random_problem(S0 + s, O)
    p["qd0"] = p["qd0"] * rng.choice([0.0, 0.2, 1.0, 1.9])          # rest ... close to the speed limits
    p["qdd0"] = p["qdd0"] * rng.choice([0.0, 1.0, 5.0])
    p["qd0"] = np.clip(p["qd0"], -0.98 * SPEED, 0.98 * SPEED)
    armtd = s % 3 == 2
    if armtd:
        jrs, kr = synthetic_offline_jrs(p["qd0"], T)
        nlp.set_parameters_armtd(p["q0"], p["qd0"], p["q_des"], jrs, kr, p["obstacles"])
        o = mk_oracle().set_problem_armtd(p["q0"], p["qd0"], p["q_des"], jrs, kr, p["obstacles"])
    else:
        nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
        o = mk_oracle().set_problem(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    m = o.min_margin()
    min_margin = min(min_margin, m)
    md = float(nlp.prune_margin()[0])       # the device's own figure (armour_get_prune_margin): what a call without an oracle has
    r = min(m, 1.0)
    if not (np.sqrt(1.0 + 2.0 * r - r * r) - 1.0 - 2e-10 <= md <= m + 2e-10):
        margin_mismatch += 1
        print(f"seed {S0 + s}: device prune margin {md:.6e}, oracle {m:.6e}", flush=True)
    if md < 1e-9:
        flagged.append(S0 + s)
        print(f"seed {S0 + s}: DEVICE prune margin {md:.3e} < 1e-9 (oracle {m:.3e}): a verdict of this build may flip", flush=True)
    bad_keys = 0
    for which, cnt in (("link", o.J), ("torque", o.n)):
        for i in range(cnt):
            for t in range(T):
                c, ind, keys, co = o.pz(which, i, t)
                c2, ind2, keys2, co2 = nlp.pz(which, i, t)
                if not np.array_equal(keys, keys2):
                    bad_keys += 1
                    continue
                if len(keys):
                    worst["coef"] = max(worst["coef"], np.abs(co - co2).max())
                worst["cen"] = max(worst["cen"], np.abs(c - c2).max(), np.abs(ind - ind2).max())
    if bad_keys:
        key_mismatch += 1
        low_margin_cases += m < 1e-9
        print(f"seed {S0 + s}: {bad_keys} PZs with different key sets (oracle min_margin {m:.2e}, device {md:.2e}; flagged by the device: {md < 1e-9})", flush=True)
        continue
    worst["gens"] = max(worst["gens"], np.abs(nlp.link_generators()[0] - o.link_generators()).max())
    if not armtd:
        worst["radius"] = max(worst["radius"], np.abs(nlp.torque_radius()[0] - o.torque_radius()).max())
    if O:
        A2, d2, dl2 = nlp.hyperplanes()
        A, d, dl = o.hyperplanes()
        worst["planes"] = max(worst["planes"], np.abs(A - A2[0]).max(), np.abs(d - d2[0]).max(), np.abs(dl - dl2[0]).max())
    for k in random_k(S0 + s, 3):
        g, jac = nlp.eval_g_jac(k)
        gr, jr = o.eval_g_jac(k)
        worst["g"] = max(worst["g"], np.abs(g[0] - gr).max())
        worst["jac"] = max(worst["jac"], np.abs(jac[0] - jr).max())
print(f"{N} problems (seeds {S0}..{S0 + N - 1}, every third in ARMTD mode, O in {{0,1,3,10,20,40}}, T={T}{', 8-link arm with gripper' if GRIPPER else ''}) in {time.time() - t_start:.0f} s")
print("worst absolute deviations device vs oracle:", {k: float(f"{v:.3g}") for k, v in worst.items()})
print(f"smallest prune margin seen by the oracle: {min_margin:.3g}; problems with a key-set difference: {key_mismatch} (of which margin < 1e-9: {low_margin_cases})")
print(f"device prune margin against the oracle's: {margin_mismatch} disagreement(s); seeds the DEVICE flags below 1e-9: {flagged if flagged else 'none'}")
