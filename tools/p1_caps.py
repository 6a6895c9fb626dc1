"""Dev: reach-set build time against the raw-term capacity (LDS per wave -> waves per CU)."""
import sys
sys.path.insert(0, '/root/repo')
from armour_amd.planner import ArmourNLP
from armour_amd._lib import ArmourLimits
from armour_amd.worlds import random_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for seed in (0, 1000):
    pb = random_batch(seed, B, 20)
    for raw in (4096, 2048, 1024):
        lim = ArmourLimits(); lim.raw_terms = raw
        nlp = ArmourNLP(T=100, limits=lim)
        for _ in range(2):
            nlp.set_parameters(pb['q0'], pb['qd0'], pb['qdd0'], pb['q_des'], pb['obstacles'])
        print("seed", seed, "B", B, "raw_terms", raw, "build ms", round(nlp.build_ms, 2), flush=True)
        nlp.close()
