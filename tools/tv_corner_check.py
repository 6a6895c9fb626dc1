"""Dev: corner configurations of the time-vectorised build against the CPU oracle: few time steps (partly filled lanes), the 9-link Fetch chain
(mixed joint axes, two fixed links: the largest slot demand), the 8-link gripper arm, T = 128 (full lanes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from armour_amd import planner as P
from armour_amd.worlds import random_batch, random_k
import oracle.cpu_oracle as OC

def check(label, T, B, robot=None, orobot=None, O=2, seed=8100):
    n = 7
    bp = random_batch(seed, B, O)
    nlp = (P.ArmourNLP(robot=robot, T=T) if robot is not None else P.ArmourNLP(T=T)).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    ks = random_k(2, B)
    g, jac = nlp.eval_g_jac(ks)
    worst = 0.0
    for b in (0, B - 1):
        o = (OC.Oracle(robot=orobot, T=T) if orobot is not None else OC.Oracle(T=T)).set_problem(bp["q0"][b], bp["qd0"][b], bp["qdd0"][b], bp["q_des"][b], bp["obstacles"][b])
        gr, jr = o.eval_g_jac(ks[b])
        bad = 0
        for which, cnt in (("link", o.J), ("torque", o.n)):
            for i in range(cnt):
                for t in range(0, T, 5):
                    if not np.array_equal(o.pz(which, i, t)[2], nlp.pz(which, i, t, b=b)[2]): bad += 1
        assert bad == 0, (label, b, bad)
        worst = max(worst, np.abs(g[b] - gr).max(), np.abs(jac[b] - jr).max())
    assert worst <= 1e-8, (label, worst)
    print(label, "B", B, "T", T, "build ms", round(nlp.build_ms, 2), "worst |dg|,|djac|", worst, flush=True)
    nlp.close()

check("few time steps", 20, 64)
check("T = 128", 128, 40)
check("gripper arm", 100, 40, P.kinova_gripper_robot(), OC.kinova_gripper_robot())
if hasattr(P, "fetch_robot"):
    check("fetch", 100, 32, P.fetch_robot(), OC.fetch_robot(), O=1)
print("ok")
