import os, sys
sys.path.insert(0, os.getcwd())
os.environ["ARMOUR_SOLVE_TIMING"] = "1"
import numpy as np
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import reference_sample_problem
from armour_amd.scenes import reference_worlds
p = reference_sample_problem()
nlp = ArmourNLP(T=100).set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
for _ in range(5): nlp.solve(device_qp=True)
print("=== sample, device form", flush=True)
r = nlp.solve(device_qp=True)[0]; print(r, flush=True)
print("=== sample, host form", flush=True)
r = nlp.solve(host_qp=True)[0]; print(r, flush=True)
for idx in (0, 50, 103):
    name, w = reference_worlds()[idx]
    nlp.set_parameters(w["q0"], w["qd0"], w["qdd0"], w["q_des"], w["obstacles"])
    for _ in range(3): nlp.solve(device_qp=True)
    print("===", name, "device form", flush=True)
    print(nlp.solve(device_qp=True)[0], flush=True)
    print("===", name, "host form", flush=True)
    print(nlp.solve(host_qp=True)[0], flush=True)
