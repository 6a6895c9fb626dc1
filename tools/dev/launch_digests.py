"""Dev: launch-to-launch reproducibility of the shipped reach-set build in every launch shape -- the same batch built REPS times,
sha1 of all tables (torque radii, link generators, link / torque PZ tables via g and jac at one k) compared with the first build.
    python tools/dev/launch_digests.py [REPS=12]   (the digests profiles/r04_fuzz.txt records: a value-preserving change reproduces them)"""
import hashlib
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_batch, random_k
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
total_bad = 0
for B, O in ((1, 20), (2, 20), (3, 20), (5, 20), (9, 20), (12, 20), (14, 20), (16, 20), (40, 10), (64, 20), (128, 20), (150, 5), (300, 3)):
    bp = random_batch(7, B, O)
    ks = random_k(3, B)
    first = None
    bad = 0
    for rep in range(REPS):
        nlp = ArmourNLP(T=100).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
        g, jac = nlp.eval_g_jac(ks)
        h = hashlib.sha1(np.ascontiguousarray(nlp.torque_radius()).tobytes() + np.ascontiguousarray(nlp.link_generators()).tobytes()
                         + np.ascontiguousarray(g).tobytes() + np.ascontiguousarray(jac).tobytes()).hexdigest()
        first = first or h
        info = nlp.build_info()
        bad += h != first
        del nlp
    total_bad += bad
    print(f"B={B:4d} O={O:3d}: {REPS} builds, {bad} differ from the first ({first[:12]})  [{info['kernel']}, {info['waves']} wave(s) per block, {info['launches']} launch(es)]", flush=True)
print("TOTAL differing builds:", total_bad)
