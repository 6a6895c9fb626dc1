"""dev: more soundness stress of the row lists -- (a) armour_eval_violations culled against full over many worlds and points, (b) armour_solve culled
against the host form in the modes without torque rows (ARMTD comparison mode, TURN_OFF_INPUT_CONSTRAINTS) and on the Kinova with gripper.
    python tools/dev/cull_stress.py [SEEDS=20]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from armour_amd import _lib
from armour_amd.planner import ArmourNLP, default_params, kinova_gripper_robot
from armour_amd.worlds import random_batch, random_k, synthetic_offline_jrs
S = int(sys.argv[1]) if len(sys.argv) > 1 else 20
def key(res): return [(tuple(r["k_opt"]), r["cost"], r["max_violation"], r["feasible"], r["iterations"], r["evaluations"], r["status"]) for r in res]
bad = 0
# (a) violation records
full, cul = ArmourNLP(T=100), ArmourNLP(T=100).set_option(_lib.OPT_CULL_ROWS, 1)
nrec = 0
for seed in range(S):
    for B, O in ((8, 20), (4, 50), (3, 3)):
        bp = random_batch(20000 + 7 * seed + O, B, O)
        for h in (full, cul): h.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
        for it in range(6):
            k = random_k(seed * 10 + it, B) * (1.0 if it % 2 else 0.3)
            if it == 5: k = np.sign(k)
            a, c = full.eval_violations(k), cul.eval_violations(k)
            for x, y in zip(a, c):
                nrec += 1
                if not (x["l1_violation"] == y["l1_violation"] and x["n_violated"] == y["n_violated"] and x["n_outside_slack"] == y["n_outside_slack"] and x["feasible"] == y["feasible"]):
                    bad += 1; print("RECORD MISMATCH", seed, B, O, it, x, y, flush=True)
print(f"(a) {nrec} violation records compared, {bad} differ", flush=True)
# (b) solver in the other modes
nb = 0
pr = default_params(100); pr.input_constraints_off = 1
modes = {"no input constraints": ArmourNLP(params=pr), "gripper": ArmourNLP(robot=kinova_gripper_robot(), T=100), "armtd": ArmourNLP(T=100)}
for seed in range(S):
    for name, nlp in modes.items():
        B, O = 12, (20, 10, 40)[seed % 3]
        bp = random_batch(30000 + 11 * seed, B, O)
        if seed % 2: bp["q_des"] = bp["q0"] + 0.05 * (bp["q_des"] - bp["q0"])
        if name == "armtd":
            jrs = np.stack([synthetic_offline_jrs(bp["qd0"][b], T=100)[0] for b in range(B)]); kr = np.stack([synthetic_offline_jrs(bp["qd0"][b], T=100)[1] for b in range(B)])
            nlp.set_parameters_armtd(bp["q0"], bp["qd0"], bp["q_des"], jrs, kr, bp["obstacles"])
        else:
            nlp.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
        nlp.set_option(_lib.OPT_SOLVE_CULL, 1); c = nlp.solve(device_qp=True)
        nlp.set_option(_lib.OPT_SOLVE_CULL, 0); h = nlp.solve(host_qp=True)
        nb += 1
        if key(c) != key(h):
            bad += 1; print("SOLVE MISMATCH", name, seed, [i for i, (x, y) in enumerate(zip(key(c), key(h))) if x != y], flush=True)
        if not all(np.all(np.abs(r["k_opt"]) <= 1 + 1e-6) for r in c): bad += 1; print("OUT OF BOX", name, seed, flush=True)
print(f"(b) {nb} batches of 12 solved in three modes, culled device form against host form; differences so far {bad}")
sys.exit(1 if bad else 0)
