import os, sys, re, io
sys.path.insert(0, os.getcwd())
os.environ["ARMOUR_P1_TRACE"] = "1"
from armour_amd.planner import ArmourNLP
from armour_amd.scenes import reference_worlds
ws = reference_worlds()
nlp = ArmourNLP(T=100)
for name, p in ws:
    for _ in range(2): nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    print("WORLD", name, "%.3f" % nlp.build_ms, flush=True)
nlp.close()
