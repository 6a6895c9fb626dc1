"""dev: ARMOUR_P1_TRACE's phase lines of a lone problem's reach-set build (release library): where the main and the helper blocks' cycles go.
    python tools/dev/phase_probe.py [seed] [id=value ...]      GPU box"""
import os, sys
sys.path.insert(0, os.getcwd())
os.environ["ARMOUR_P1_TRACE"] = "1"
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_batch, reference_sample_problem
seed = int(sys.argv[1]) if len(sys.argv) > 1 and "=" not in sys.argv[1] else -1
p = reference_sample_problem() if seed < 0 else {k: v[0] for k, v in random_batch(seed, 1, 20).items()}
nlp = ArmourNLP(T=100)
for kv in sys.argv[1:]:
    if "=" in kv: nlp.set_option(int(kv.split("=")[0]), float(kv.split("=")[1]))
for _ in range(4):
    nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
nlp.close()
