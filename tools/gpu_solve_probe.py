"""Dev probe: armour_solve wall time, single problem and a batch."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_problem, random_batch
sys.path.insert(0, 'tests')
from helpers import SAMPLE_PROBLEM as SP
for B, O in ((1, 10), (16, 10), (128, 10)):
    pb = {k: np.stack([np.asarray(SP[k], dtype=float) + (0.01 * b if k == 'q_des' else 0.0) for b in range(B)]) for k in ('q0', 'qd0', 'qdd0', 'q_des', 'obstacles')}
    nlp = ArmourNLP(T=100).set_parameters(pb['q0'], pb['qd0'], pb['qdd0'], pb['q_des'], pb['obstacles'])
    for rep in range(3):
        t0 = time.perf_counter(); sol = nlp.solve(); dt = (time.perf_counter() - t0) * 1e3
    print(f"B={B} O={O}: solve {dt:.2f} ms wall, P1 {nlp.build_ms:.2f} ms; iterations {[s['iterations'] for s in sol][:4]} evaluations {[s['evaluations'] for s in sol][:4]} feasible {[bool(s['feasible']) for s in sol][:4]}")
    nlp.close()
