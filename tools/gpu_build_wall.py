"""dev: wall time of armour_set_problems (one problem, O = 20) next to its device time: median of 50 calls, interpreter GC off."""
import gc, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_batch
for B in (1, 4):
    bp = random_batch(5, B, 20)
    nlp = ArmourNLP(T=100)
    for _ in range(3): nlp.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    gc.disable()
    wall, dev = [], []
    for _ in range(50):
        t0 = time.perf_counter(); nlp.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"]); wall.append((time.perf_counter() - t0) * 1e3); dev.append(nlp.build_ms)
    gc.enable()
    print(f"B={B}: wall median {np.median(wall):.3f} ms (min {min(wall):.3f}), device median {np.median(dev):.3f} ms", flush=True)
    nlp.close()
