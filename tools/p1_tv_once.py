"""dev: a few time-vectorised builds of B problems (argv[1]); further arguments: per-handle options opt=val"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pb = random_batch(0, B, 20)
nlp = ArmourNLP(T=100)
for kv in sys.argv[2:]:
    k, v = kv.split("="); nlp.set_option(int(k), float(v))
for _ in range(4): nlp.set_parameters(pb["q0"], pb["qd0"], pb["qdd0"], pb["q_des"], pb["obstacles"])
print("B", B, "build ms", nlp.build_ms)
