"""dev: armour_solve of ONE problem, device-resident form against host-driven form, over random worlds (O = 10 and 20): wall ms by outcome."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_problem
for O in (10, 20):
    rows = []
    nlp = ArmourNLP(T=100)
    for seed in range(40):
        p = random_problem(9000 + seed, O)
        nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
        t = {}
        for host in (False, True):
            nlp.solve(host_qp=host)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); r = nlp.solve(host_qp=host)[0]; ts.append((time.perf_counter() - t0) * 1e3)
            t[host] = min(ts)
        rows.append((r["feasible"], r["iterations"], r["evaluations"], t[False], t[True]))
    for feas in (True, False):
        sel = [x for x in rows if x[0] == feas]
        if sel:
            d = np.array([x[3] for x in sel]); h = np.array([x[4] for x in sel])
            print(f"O={O} feasible={feas}: {len(sel)} problems; device form median {np.median(d):.3f} ms (max {d.max():.3f}), host form median {np.median(h):.3f} ms (max {h.max():.3f}); device faster in {int((d < h).sum())}", flush=True)
    d = np.array([x[3] for x in rows]); h = np.array([x[4] for x in rows])
    print(f"O={O} all: mean device {d.mean():.3f} ms, mean host {h.mean():.3f} ms", flush=True)
    nlp.close()
