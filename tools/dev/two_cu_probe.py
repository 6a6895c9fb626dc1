"""dev: a lone problem's reach-set build with a time step on one CU (ARMOUR_OPT_P1_STEP_TWO_CU = 0) and on two (1): build times of a few
fresh builds each, the tables compared bit for bit, the device's trace lines (ARMOUR_P1_TRACE) for which shape ran.  GPU box."""
import os, sys, hashlib
sys.path.insert(0, os.getcwd())
if os.environ.get("TWO_CU_TRACE"): os.environ["ARMOUR_P1_TRACE"] = "1"
import numpy as np
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_batch, random_k

def tables(nlp, ks):
    g, jac = nlp.eval_g_jac(ks)
    h = hashlib.sha1(b"".join(np.ascontiguousarray(a).tobytes() for a in (nlp.torque_radius(), nlp.link_generators(), g, jac)))
    for which, cnt in (("link", nlp.J), ("torque", nlp.n)):
        for i in range(cnt):
            for t in range(0, nlp.T, 7):
                h.update(b"".join(np.ascontiguousarray(a).tobytes() for a in nlp.pz(which, i, t, b=0)))
    return h.hexdigest()[:16]

T = 100
variants = [[kv for kv in v.split(",") if kv] for v in sys.argv[1:]] or [["123=0"], ["123=1"], ["123=2"], ["123=3"]]
res = {}
for seed in (5, 11):
    bp = random_batch(seed, 1, 20); ks = random_k(3, 1)
    for two in variants + variants:
        nlp = ArmourNLP(T=T)
        for kv in two: nlp.set_option(int(kv.split('=')[0]), float(kv.split('=')[1]))
        ms = []
        for _ in range(6):
            nlp.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
            ms.append(nlp.build_ms)
        d = tables(nlp, ks)
        print("TWO_CU", two, "seed", seed, "build ms best %.3f |" % min(ms), " ".join("%.3f" % m for m in ms), "digest", d, "margin", nlp.prune_margin(), flush=True)
        res.setdefault(seed, set()).add(d)
        nlp.close()
print("digests equal:", all(len(v) == 1 for v in res.values()))
