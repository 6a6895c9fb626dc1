"""dev: armour_solve on small batches of random worlds (O = 20): device-resident form against host-driven form, mean wall ms over 12 batches."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_batch
for B in (1, 2, 3, 4, 6, 8, 12, 16):
    td, th = [], []
    nlp = ArmourNLP(T=100)
    for seed in range(12):
        bp = random_batch(9500 + seed, B, 20)
        nlp.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
        for host, acc in ((False, td), (True, th)):
            nlp.solve(host_qp=host)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); nlp.solve(host_qp=host); ts.append((time.perf_counter() - t0) * 1e3)
            acc.append(min(ts))
    td, th = np.array(td), np.array(th)
    print(f"B={B}: device form mean {td.mean():.3f} ms (median {np.median(td):.3f}), host form mean {th.mean():.3f} ms (median {np.median(th):.3f}); device faster in {int((td < th).sum())} of {len(td)} batches", flush=True)
    nlp.close()
