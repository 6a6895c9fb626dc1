"""Dev: a batch larger than the resident blocks of the time-vectorised build (blocks loop over several items), sampled worlds against the CPU oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_batch, random_k
from oracle.cpu_oracle import Oracle
B, T, O = int(sys.argv[1]) if len(sys.argv) > 1 else 300, 100, 3
bp = random_batch(7000, B, O)
nlp = ArmourNLP(T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
print("B", B, "build ms", round(nlp.build_ms, 2), "per problem", round(nlp.build_ms / B, 4))
ks = random_k(1, B)
g, jac = nlp.eval_g_jac(ks)
for b in (0, B // 2 + 1, B - 1):
    o = Oracle(T=T).set_problem(bp["q0"][b], bp["qd0"][b], bp["qdd0"][b], bp["q_des"][b], bp["obstacles"][b])
    gr, jr = o.eval_g_jac(ks[b])
    bad = 0
    for which, cnt in (("link", o.J), ("torque", o.n)):
        for i in range(cnt):
            for t in range(0, T, 7):
                c, ind, keys, co = o.pz(which, i, t)
                c2, ind2, keys2, co2 = nlp.pz(which, i, t, b=b)
                if not np.array_equal(keys, keys2): bad += 1
    print("world", b, "|dg|", np.abs(g[b] - gr).max(), "|djac|", np.abs(jac[b] - jr).max(), "key mismatches", bad, "tr diff", np.abs(nlp.torque_radius()[b] - o.torque_radius()).max())
    assert bad == 0 and np.abs(g[b] - gr).max() <= 1e-9 and np.abs(jac[b] - jr).max() <= 1e-8
print("ok")
