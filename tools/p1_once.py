"""Dev: run the reach-set build (P1) a few times, B problems (argv[1], default 1), O = 20."""
import sys
sys.path.insert(0, '/root/repo')
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
pb = random_batch(0, B, 20)
nlp = ArmourNLP(T=100)
import os
for kv in os.environ.get('P1_ONCE_OPTS', '').split(';'):   # dev: 'id=value;id=value' per-handle options
    if kv: nlp.set_option(int(kv.split('=')[0]), float(kv.split('=')[1]))
for _ in range(3):
    nlp.set_parameters(pb['q0'], pb['qd0'], pb['qdd0'], pb['q_des'], pb['obstacles'])
print("B", B, "build ms", nlp.build_ms)
nlp.close()
