"""dev: wall time of armour_set_problems for a batch with the work arena released after every build (default) and kept (ARMOUR_OPT_P1_KEEP_WORK_MEMORY)."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from armour_amd import _lib
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_batch
for B in [int(x) for x in sys.argv[1:]] or (24, 64, 128):
    bp = random_batch(5, B, 20)
    for keep in (0, 1):
        nlp = ArmourNLP(T=100).set_option(_lib.OPT_P1_KEEP_WORK_MEMORY, keep)
        wall, dev = [], []
        for _ in range(6):
            t0 = time.perf_counter()
            nlp.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
            wall.append((time.perf_counter() - t0) * 1e3); dev.append(nlp.build_ms)
        print(f"B={B} keep={keep}: wall ms {' '.join(f'{w:.2f}' for w in wall)} | device ms (reach sets + half-spaces) {min(dev):.2f}", flush=True)
        nlp.close()
