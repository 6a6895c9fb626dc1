"""dev: host = full device = culled device form on the reference's worlds (batches of them, shuffled, with perturbed waypoints) -- where the box-clipped first try fires."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from armour_amd import _lib
from armour_amd.planner import ArmourNLP
from armour_amd.scenes import reference_worlds, as_batch
def key(res): return [(tuple(r["k_opt"]), r["cost"], r["max_violation"], r["feasible"], r["iterations"], r["evaluations"], r["status"]) for r in res]
ws = reference_worlds(); rng = np.random.default_rng(1)
nlp = ArmourNLP(T=100); bad = tot = feas = 0
for rep in range(12):
    pick = rng.permutation(107)[: int(rng.choice([1, 2, 5, 16, 40]))]
    sub = [ws[i] for i in pick]
    bp = as_batch(sub)
    scale = float(rng.choice([1.0, 0.3, 0.05, 0.01]))          # shorter waypoints: interior optima, QPs that do NOT end in a corner
    bp["q_des"] = bp["q0"] + scale * (bp["q_des"] - bp["q0"])
    bp["qd0"] = bp["qd0"] + (rng.uniform(-0.3, 0.3, bp["qd0"].shape) if rep % 3 == 2 else 0.0)
    nlp.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    nlp.set_option(_lib.OPT_SOLVE_CULL, 1); cul = nlp.solve(device_qp=True)
    nlp.set_option(_lib.OPT_SOLVE_CULL, 0); full = nlp.solve(device_qp=True)
    host = nlp.solve(host_qp=True)
    ok = key(cul) == key(full) == key(host)
    tot += len(sub); feas += sum(r["feasible"] for r in cul); bad += 0 if ok else 1
    print(f"rep {rep}: B {len(sub)} scale {scale} feasible {sum(r['feasible'] for r in cul)} iterations {max(r['iterations'] for r in cul)} equal {ok}", flush=True)
print(f"DONE {tot} problems, {feas} feasible, {bad} mismatching batches"); sys.exit(1 if bad else 0)
