import sys; sys.path.insert(0,'.')
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_problem
p = random_problem(0, 20)
nlp = ArmourNLP(T=100).set_parameters(p['q0'], p['qd0'], p['qdd0'], p['q_des'], p['obstacles'])
print("build ms", nlp.build_ms)
