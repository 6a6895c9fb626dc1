"""dev: ARMTD comparison mode -- break-even batch size between the per-step and the time-vectorised reach-set kernel (ARMOUR_OPT_P1_BUILD 1 / 2), build ms (best of 3), O = 20, T = 100."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from armour_amd import _lib
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_batch, synthetic_offline_jrs
for B in (16, 20, 24, 28, 32, 36, 40, 48, 64):
    bp = random_batch(5, B, 20)
    jk = [synthetic_offline_jrs(bp["qd0"][b], T=100) for b in range(B)]
    jrs = np.stack([x[0] for x in jk]); kr = np.stack([x[1] for x in jk])
    out = {}
    for opt in (1, 2):
        nlp = ArmourNLP(T=100).set_option(_lib.OPT_P1_BUILD, opt)
        ms = []
        for _ in range(3):
            nlp.set_parameters_armtd(bp["q0"], bp["qd0"], bp["q_des"], jrs, kr, bp["obstacles"]); ms.append(nlp.build_ms)
        out[opt] = min(ms); nlp.close()
    print(f"B={B}: per-step {out[1]:.2f} ms, time-vectorised {out[2]:.2f} ms", flush=True)
