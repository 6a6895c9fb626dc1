import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from helpers import SAMPLE_PROBLEM as p
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_problem, random_batch
nlp = ArmourNLP(T=128).set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
for i in range(2):
    t0 = time.time(); s = nlp.solve()[0]; print("sample T=128 O=10: wall ms", (time.time()-t0)*1e3, {k: s[k] for k in ("cost","feasible","iterations","evaluations","status","time_ms")})
for seed in range(4):
    q = random_problem(seed, 20)
    nlp = ArmourNLP(T=100).set_parameters(q["q0"], q["qd0"], q["qdd0"], q["q_des"], q["obstacles"])
    t0 = time.time(); s = nlp.solve()[0]; print("rand", seed, "O=20: P1 ms", round(nlp.build_ms,1), "solve ms", round((time.time()-t0)*1e3,1), {k: s[k] for k in ("cost","feasible","iterations","evaluations","status","max_violation")})
bp = random_batch(0, 32, 20)
nlp = ArmourNLP(T=100).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
t0 = time.time(); ss = nlp.solve(); dt = time.time()-t0
print("batch 32: P1 ms", round(nlp.build_ms,1), "solve ms", round(dt*1e3,1), "feasible", sum(s["feasible"] for s in ss), "status", [s["status"] for s in ss], "iters", [s["iterations"] for s in ss])
