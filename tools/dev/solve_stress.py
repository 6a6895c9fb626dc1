"""dev: soundness stress of the culled device solve -- many random batches, the persistent kernel on the solver's row lists against the full device
form and the host-driven form: k_opt, cost, violation, counts, status of every problem must be equal bit for bit.
    python tools/dev/solve_stress.py [SEEDS=20] [B=32]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from armour_amd import _lib
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_batch
S = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
def key(res): return [(tuple(r["k_opt"]), r["cost"], r["max_violation"], r["feasible"], r["iterations"], r["evaluations"], r["status"]) for r in res]
bad = tot = feas = 0
t0 = time.time()
nlp = ArmourNLP(T=100)
for seed in range(S):
    for O in (3, 10, 20, 50):
        bp = random_batch(9000 + 31 * seed + O, B, O)
        if seed % 3 == 1:   # easier worlds: goals near the start, so that some problems are feasible and converge
            bp["q_des"] = bp["q0"] + 0.05 * (bp["q_des"] - bp["q0"])
        nlp.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
        nlp.set_option(_lib.OPT_SOLVE_CULL, 1); cul = nlp.solve(device_qp=True)
        nlp.set_option(_lib.OPT_SOLVE_CULL, 0); full = nlp.solve(device_qp=True)
        host = nlp.solve(host_qp=True) if seed % 4 == 0 else full
        ok = key(cul) == key(full) == key(host)
        tot += B; feas += sum(r["feasible"] for r in cul); bad += 0 if ok else 1
        if not ok:
            print("MISMATCH seed", seed, "O", O, [i for i, (a, b2, c) in enumerate(zip(key(cul), key(full), key(host))) if not (a == b2 == c)][:5], flush=True)
    if seed % 5 == 4: print(f"seed {seed}: {tot} problems so far, {feas} feasible, {bad} mismatching batches, {time.time() - t0:.0f} s", flush=True)
print(f"DONE: {tot} problems ({feas} feasible) in {4 * S} batches of {B}: {bad} batches with a difference")
sys.exit(1 if bad else 0)
