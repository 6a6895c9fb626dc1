"""dev: per-wave stamps (variant library built with -DP1_STAMPS) of one problem's per-step build, pairs off / on"""
import sys
sys.path.insert(0, '/root/repo')
from armour_amd import _lib
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_batch
bp = random_batch(5, 1, 20)
for pairs in (0, 1):
    print("== pairs", pairs, flush=True)
    nlp = ArmourNLP(T=100).set_option(_lib.OPT_P1_BUILD, 1).set_option(_lib.OPT_P1_STEP_PAIRS, pairs)
    for _ in range(2):
        nlp.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    print("build ms", nlp.build_ms, flush=True)
    nlp.close()
