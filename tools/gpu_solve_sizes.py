import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_batch
for B, O in ((1, 20), (16, 20), (128, 20), (128, 50)):
    bp = random_batch(11, B, O)
    nlp = ArmourNLP(T=100).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    out = {}
    for host in (False, True):
        nlp.solve(host_qp=host)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); r = nlp.solve(host_qp=host); ts.append((time.perf_counter() - t0) * 1e3)
        out["host" if host else "device"] = min(ts)
    print(f"B={B} O={O}: device form {out['device']:.3f} ms, host form {out['host']:.3f} ms, feasible {sum(x['feasible'] for x in r)}/{B}", flush=True)
    nlp.close()
