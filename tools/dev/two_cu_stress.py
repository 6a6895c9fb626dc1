"""dev / evidence: a lone problem's reach-set build on two CUs per time step, many times, every table of every build against the one-CU build of the
same problem -- alone on the device, and beside a second process that keeps launching the fused evaluation of another handle (uneven load, the
L2 and the CUs shared: what a cross-CU hand-off must survive).
    python tools/dev/two_cu_stress.py [problems=16] [builds per problem=120] [load: 0 | 1]       GPU box"""
import hashlib, os, subprocess, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np

def digest(nlp, ks):
    g, jac = nlp.eval_g_jac(ks)
    return hashlib.sha1(b"".join(np.ascontiguousarray(a).tobytes() for a in (nlp.torque_radius(), nlp.link_generators(), g, jac))).hexdigest()[:16]

if len(sys.argv) > 1 and sys.argv[1] == "load":   # the child: evaluations of a 40-world handle until killed
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch, random_k
    bp = random_batch(9, 128, 50)   # (a launch of this handle fills the device: the two-CU builds meet it at random points of their time steps)
    nlp = ArmourNLP(T=100).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    ks = random_k(1, 128)
    t_end = time.time() + float(sys.argv[2])
    n = 0
    while time.time() < t_end:
        nlp.eval_g_jac(ks); n += 1
    print("load child:", n, "evaluations", flush=True)
    nlp.close(); sys.exit(0)

from armour_amd import _lib
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_k, random_problem
NP = int(sys.argv[1]) if len(sys.argv) > 1 else 16
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 120
LOAD = len(sys.argv) > 3 and sys.argv[3] == "1"
child = subprocess.Popen([sys.executable, __file__, "load", str(10 + 0.02 * NP * NB)]) if LOAD else None
if child: time.sleep(4.0)
ks = random_k(2, 1)
bad = total = fell_back = 0
t0 = time.time()
for s in range(NP):
    rng = np.random.default_rng(8800 + s)
    p = random_problem(8800 + s, int(rng.choice([0, 3, 10, 20])))
    p["qd0"] = p["qd0"] * rng.choice([0.0, 0.5, 1.0])
    ref = ArmourNLP(T=100); ref.set_option(_lib.OPT_P1_STEP_TWO_CU, 0)
    ref.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    d0 = digest(ref, ks); ref.close()
    nlp = ArmourNLP(T=100)
    for i in range(NB):
        if i % 40 == 39:   # a fresh handle now and then (fresh arena, fresh exchange area, epoch 1)
            nlp.close(); nlp = ArmourNLP(T=100)
        nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
        total += 1
        if nlp.get_option(_lib.OPT_P1_STEP_TWO_CU) == 0:   # (a helper that started late under load: legitimate, counted, switched back on)
            fell_back += 1; nlp.set_option(_lib.OPT_P1_STEP_TWO_CU, 3)
        if digest(nlp, ks) != d0:
            bad += 1; print("MISMATCH problem", s, "build", i, flush=True)
    nlp.close()
print("two-CU stress%s: %d builds of %d problems in %.0f s, %d differ from the one-CU tables, %d fell back to one CU (late helper)" % (" beside a second process" if LOAD else "", total, NP, time.time() - t0, bad, fell_back), flush=True)
if child: child.wait()
sys.exit(1 if bad else 0)
