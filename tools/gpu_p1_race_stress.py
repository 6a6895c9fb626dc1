"""Dev: repeat 3-wave (single-problem) reach-set builds and compare them bit for bit with a 1-wave batched build of the same
problems -- a missing barrier between roles would show up as an intermittent mismatch."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_batch, random_k
T, O, B, REPS = 100, 3, 10, int(sys.argv[1]) if len(sys.argv) > 1 else 30
bp = random_batch(300, B, O)
bp["qd0"][3] = 0.9 * np.array([1.3963, 1.3963, 1.3963, 1.3963, 1.2218, 1.2218, 1.2218])
ks = random_k(1, B)
ref = ArmourNLP(T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
g, jac = ref.eval_g_jac(ks)
tr, lg = ref.torque_radius(), ref.link_generators()
one = ArmourNLP(T=T)
bad = 0
for rep in range(REPS):
    for b in range(B):
        one.set_parameters(bp["q0"][b], bp["qd0"][b], bp["qdd0"][b], bp["q_des"][b], bp["obstacles"][b])
        g1, j1 = one.eval_g_jac(ks[b])
        ok = np.array_equal(g1[0], g[b]) and np.array_equal(j1[0], jac[b]) and np.array_equal(one.torque_radius()[0], tr[b]) and np.array_equal(one.link_generators()[0], lg[b])
        if not ok:
            bad += 1
            print("MISMATCH rep", rep, "problem", b, flush=True)
print(f"{REPS * B} three-wave builds compared with the one-wave batch: {bad} mismatches")
