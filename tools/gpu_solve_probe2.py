"""Timing of the two forms of armour_solve (device-resident vs host-driven) on the reference's sample problem and batches."""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from helpers import SAMPLE_PROBLEM as p
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_batch

def t_solve(nlp, reps=20, **kw):
    nlp.solve(**kw)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); s = nlp.solve(**kw); ts.append((time.perf_counter() - t0) * 1e3)
    return np.median(ts), min(ts), s

for T in (100, 128):
    nlp = ArmourNLP(T=T).set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    for name, kw in (("device", {}), ("host-qp", dict(host_qp=True))):
        med, mn, s = t_solve(nlp, **kw)
        print(f"sample problem T={T} {name:8s}: median {med:.3f} ms, min {mn:.3f} ms | evaluations {s[0]['evaluations']} iterations {s[0]['iterations']} status {s[0]['status']} feasible {s[0]['feasible']}")
for B, O in ((1, 20), (16, 20), (128, 20), (128, 50)):
    bp = random_batch(0, B, O)
    nlp = ArmourNLP(T=100).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    for name, kw in (("device", {}), ("host-qp", dict(host_qp=True))):
        med, mn, s = t_solve(nlp, reps=5, **kw)
        print(f"random batch B={B} O={O} {name:8s}: median {med:.3f} ms ({med / B * 1e3:.1f} us/problem) | evaluations {[x['evaluations'] for x in s][:8]} feasible {sum(x['feasible'] for x in s)}/{B}")
