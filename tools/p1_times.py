"""Dev: reach-set build times (device) for one problem and a batch."""
import sys
sys.path.insert(0, '.')
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_problem, random_batch
p = random_problem(0, 20)
nlp = ArmourNLP(T=100)
for _ in range(3):
    nlp.set_parameters(p['q0'], p['qd0'], p['qdd0'], p['q_des'], p['obstacles'])
    print("B=1 build ms", round(nlp.build_ms, 3), flush=True)
nlp.close()
for B in (16, 128):
    pb = random_batch(0, B, 20)
    nlp = ArmourNLP(T=100)
    for _ in range(2):
        nlp.set_parameters(pb['q0'], pb['qd0'], pb['qdd0'], pb['q_des'], pb['obstacles'])
        print("B=%d build ms" % B, round(nlp.build_ms, 2), "per problem", round(nlp.build_ms / B, 3), flush=True)
    nlp.close()
