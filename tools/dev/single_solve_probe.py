"""dev: armour_solve of N single random problems (O = 20, T = 100), each in a handle of its own: median / mean wall time of a warm call, both forms."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_problem
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
O = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nlp = ArmourNLP(T=100)
res = {"host": [], "device": []}
feas = 0
for seed in range(N):
    p = random_problem(7000 + seed, O)
    nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    for name, kw in (("host", dict(host_qp=True)), ("device", dict(device_qp=True))):
        nlp.solve(**kw)
        gc.collect(); gc.disable()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); r = nlp.solve(**kw)[0]; ts.append(time.perf_counter() - t0)
        gc.enable()
        res[name].append(min(ts) * 1e3)
    feas += int(r["feasible"])
for name, v in res.items():
    v = np.array(v)
    print(f"{name} form, {N} single problems at O = {O}: median {np.median(v):.3f} ms, mean {v.mean():.3f} ms, max {v.max():.3f} ms", flush=True)
print("feasible:", feas, "of", N, "; device form faster in", int((np.array(res['device']) < np.array(res['host'])).sum()))
