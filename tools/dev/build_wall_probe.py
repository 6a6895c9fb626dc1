"""dev: wall time of armour_set_problems for the reference's sample problem (one problem, T = 100), per call, beside the device time of its kernels.
    python tools/dev/build_wall_probe.py [id=value ...]        GPU box"""
import gc, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import reference_sample_problem
p = reference_sample_problem()
nlp = ArmourNLP(T=100)
for kv in sys.argv[1:]:
    nlp.set_option(int(kv.split("=")[0]), float(kv.split("=")[1]))
gc.collect(); gc.disable()
wall, dev = [], []
for i in range(40):
    t0 = time.perf_counter()
    nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    wall.append((time.perf_counter() - t0) * 1e3); dev.append(nlp.build_ms)
wall, dev = np.array(wall[5:]), np.array(dev[5:])
print("set_problems wall ms: median %.3f min %.3f | device (events) ms: median %.3f min %.3f | options %s" % (np.median(wall), wall.min(), np.median(dev), dev.min(), sys.argv[1:]), flush=True)
nlp.close()
