"""dev: armour_solve of the reference's sample problem (RT/armour_main.cu:18-33), both forms: best / median wall time of 200 calls."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import reference_sample_problem
p = reference_sample_problem()
nlp = ArmourNLP(T=100).set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
for name, kw in (("host form", dict(host_qp=True)), ("device form", dict(device_qp=True)), ("automatic", {})):
    for _ in range(20): r = nlp.solve(**kw)[0]
    gc.collect(); gc.disable()
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); nlp.solve(**kw); ts.append(time.perf_counter() - t0)
    gc.enable()
    ts.sort()
    print(f"{name}: best {ts[0] * 1e3:.4f} ms, median {ts[100] * 1e3:.4f} ms; feasible {r['feasible']}, iterations {r['iterations']}, evaluations {r['evaluations']}", flush=True)
