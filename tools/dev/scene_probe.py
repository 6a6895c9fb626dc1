"""Where a reference world's device rows differ from the oracle's: python tools/dev/scene_probe.py <world index> [pad]   (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from armour_amd.planner import ArmourNLP
from armour_amd.scenes import reference_worlds, pad_obstacles
from oracle.cpu_oracle import Oracle
sys.path.insert(0, "tests")
from helpers import PZ_TESTS_K

idx = int(sys.argv[1]); pad = int(sys.argv[2]) if len(sys.argv) > 2 else 0
name, p = reference_worlds()[idx]
obs = pad_obstacles(p["obstacles"], pad) if pad else p["obstacles"]
T = 100
nlp = ArmourNLP(T=T).set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], obs)
o = Oracle(T=T).set_problem(p["q0"], p["qd0"], p["qdd0"], p["q_des"], obs)
O = obs.shape[0]; n = J = 7
print(name, "O", O, "skip mask %x" % int(nlp.plane_skip()[0]))
for k in (PZ_TESTS_K, np.zeros(7)):
    g, jac = nlp.eval_g_jac(k)
    gr, jr = o.eval_g_jac(k)
    d = np.abs(g[0] - gr)
    bad = np.nonzero(d > 1e-9)[0]
    print("k", k, "rows off:", bad.size, "max", d.max(), "jac max", np.abs(jac[0] - jr).max())
    for r in bad[:12]:
        q = r - n * T
        l, t, ob = q // (T * O), (q // O) % T, q % O
        print("  row", r, "link", l, "t", t, "obs", ob, "device", g[0, r], "oracle", gr[r])
A, d, dl = o.hyperplanes()
A2, d2, dl2 = nlp.hyperplanes()
print("planes: |dA| %.3e |dd| %.3e |ddelta| %.3e" % (np.abs(A - A2[0]).max(), np.abs(d - d2[0]).max(), np.abs(dl - dl2[0]).max()))
if bad.size:
    r = bad[0]; q = r - n * T; l, t, ob = q // (T * O), (q // O) % T, q % O
    print("obstacle", obs[ob])
    print("A oracle\n", A[t, l, ob], "\ndelta", dl[t, l, ob], "\nd", d[t, l, ob])
    cen = o.slice_links(np.zeros(7))[t, l]
    v = A[t, l, ob] @ cen
    print("pos", v - d[t, l, ob] - dl[t, l, ob]); print("neg", -v + d[t, l, ob] - dl[t, l, ob])
    print("gens", nlp.link_generators()[0][t, l])
