"""Ad-hoc development check of the device reach-set build (P1) against the oracle."""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from helpers import PZ_TESTS_K, DEBUG_STATE
from oracle.cpu_oracle import Oracle
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_problem, random_k

def compare(T, p, label):
    o = Oracle(T=T).set_problem(p['q0'], p['qd0'], p['qdd0'], p['q_des'], p['obstacles'])
    nlp = ArmourNLP(T=T)
    t0 = time.time(); nlp.set_parameters(p['q0'], p['qd0'], p['qdd0'], p['q_des'], p['obstacles']); t1 = time.time()
    print(label, "oracle build ms", round(o.build_ms, 1), "gpu build ms", round(nlp.build_ms, 2), "wall", round((t1 - t0) * 1e3, 1), o.table_sizes(), nlp.table_sizes())
    bad = 0; maxc = 0.0
    for which, cnt in (("link", o.J), ("torque", o.n)):
        for i in range(cnt):
            for t in range(T):
                c, ind, keys, co = o.pz(which, i, t)
                c2, ind2, keys2, co2 = nlp.pz(which, i, t)
                if len(keys) != len(keys2) or not np.array_equal(keys, keys2):
                    bad += 1
                    if bad < 4: print("  key mismatch", which, i, t, len(keys), len(keys2))
                    continue
                maxc = max(maxc, np.abs(co - co2).max() if len(keys) else 0, np.abs(c - c2).max(), np.abs(ind - ind2).max())
    print("  key mismatches", bad, "max coeff/center/indep diff", maxc)
    print("  torque radius diff", np.abs(nlp.torque_radius()[0] - o.torque_radius()).max(), "link gens diff", np.abs(nlp.link_generators()[0] - o.link_generators()).max())
    if o.O:
        A, d, dl = o.hyperplanes(); A2, d2, dl2 = nlp.hyperplanes()
        print("  planes diff", np.abs(A - A2[0]).max(), np.abs(d - d2[0]).max(), np.abs(dl - dl2[0]).max())
    for k in (PZ_TESTS_K, random_k(1, 1)[0]):
        g, jac = nlp.eval_g_jac(k[None]); gr, jr = o.eval_g_jac(k)
        print("  |dg|", np.abs(g[0] - gr).max(), "|djac|", np.abs(jac[0] - jr).max())
    return nlp

p = random_problem(0, 20)
compare(100, p, "rand0")
p2 = random_problem(1, 5); p2.update(DEBUG_STATE)
compare(128, p2, "debug-state T=128")
# batch timing
from armour_amd.worlds import random_batch
from armour_amd._lib import ArmourLimits
bp = random_batch(0, 32, 20)
nlp = ArmourNLP(T=100)
t0 = time.time(); nlp.set_parameters(bp['q0'], bp['qd0'], bp['qdd0'], bp['q_des'], bp['obstacles']); t1 = time.time()
print("batch 32: gpu build ms", nlp.build_ms, "wall", (t1 - t0) * 1e3)
