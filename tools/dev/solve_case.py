"""dev: one batch of tools/dev/solve_stress.py in detail:  python tools/dev/solve_case.py seed O [B=32] [problem]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from armour_amd import _lib
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_batch
seed, O = int(sys.argv[1]), int(sys.argv[2]); B = int(sys.argv[3]) if len(sys.argv) > 3 else 32; pb = int(sys.argv[4]) if len(sys.argv) > 4 else None
bp = random_batch(9000 + 31 * seed + O, B, O)
if seed % 3 == 1: bp["q_des"] = bp["q0"] + 0.05 * (bp["q_des"] - bp["q0"])
nlp = ArmourNLP(T=100).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
out = {}
for name, cull, kw in (("culled device", 1, dict(device_qp=True)), ("full device", 0, dict(device_qp=True)), ("host", 0, dict(host_qp=True))):
    nlp.set_option(_lib.OPT_SOLVE_CULL, cull); out[name] = nlp.solve(**kw)
for b in ([pb] if pb is not None else range(B)):
    ks = [tuple(out[n][b]["k_opt"]) for n in out]
    if pb is not None or len(set(ks)) > 1 or len({(out[n][b]["iterations"], out[n][b]["evaluations"], out[n][b]["status"]) for n in out}) > 1:
        for n in out:
            r = out[n][b]
            print(b, n, "status", r["status"], "it", r["iterations"], "ev", r["evaluations"], "feas", r["feasible"], "cost", repr(r["cost"]), "viol", repr(r["max_violation"]), "k", np.array2string(np.array(r["k_opt"]), precision=17))
# single-problem handles for the same world
if pb is not None:
    one = ArmourNLP(T=100).set_parameters(bp["q0"][pb], bp["qd0"][pb], bp["qdd0"][pb], bp["q_des"][pb], bp["obstacles"][pb])
    for name, kw in (("single host", dict(host_qp=True)), ("single device", dict(device_qp=True))):
        r = one.solve(**kw)[0]
        print(name, "status", r["status"], "it", r["iterations"], "ev", r["evaluations"], "feas", r["feasible"], "cost", repr(r["cost"]), "k", np.array2string(np.array(r["k_opt"]), precision=17))
    mask, cnt, tq, ms = nlp.solver_rows()
    print("solver rows of the problem:", cnt[pb], tq[pb])
