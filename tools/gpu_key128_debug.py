"""dev (run with ARMOUR_KEY128=1): where does the 8-factor arm's device build differ from the 128-bit oracle?"""
import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from armour_amd import _lib
from armour_amd.planner import ArmourNLP, default_params, kinova_robot
from oracle import cpu_oracle as orc
import test_key128 as t
def kr(p):
    p.k_range[7 if orc.MAXF == 8 else 6] = p.k_range[0]
    return p
T, O, n = int(os.environ.get("DBG_T", "100")), 3, int(os.environ.get("DBG_N", "8"))
mk = (lambda r: t.make_eight_factor_arm(r)) if n == 8 else (lambda r: r)
rng = np.random.default_rng(2024)
lb = (np.array([-np.pi, -2.41, -np.pi, -2.66, -np.pi, -2.23, -np.pi, -2.2]) + 0.3)[:n]
q0 = rng.uniform(lb, -lb)
speed = np.array([1.3963, 1.3963, 1.3963, 1.3963, 1.2218, 1.2218, 1.2218, 1.2218])[:n]
qd0, qdd0 = rng.uniform(-0.5, 0.5, n) * speed, rng.uniform(-1.0, 1.0, n)
q_des = q0 + 0.2
obs = np.zeros((O, 12)); obs[:, 0:3] = rng.uniform([-0.8, -0.8, 0.05], [0.8, 0.8, 1.2], (O, 3)); obs[:, 3] = obs[:, 7] = obs[:, 11] = 0.1
oracle = orc.Oracle(robot=mk(orc.kinova_robot()), params=kr(orc.default_params(T))).set_problem(q0, qd0, qdd0, q_des, obs)
print("oracle sizes", oracle.table_sizes() if hasattr(oracle, "table_sizes") else None)
for build in (1, 2):
    nlp = ArmourNLP(robot=mk(kinova_robot()), params=kr(default_params(T))).set_option(_lib.OPT_P1_BUILD, build)
    B = 1 if build == 1 else 2
    rep = lambda a: np.repeat(np.asarray(a)[None], B, axis=0)
    nlp.set_parameters(rep(q0), rep(qd0), rep(qdd0), rep(q_des), rep(obs))
    print("build", build, nlp.build_info(), nlp.table_sizes())
    tr, tro = nlp.torque_radius()[0], oracle.torque_radius()
    print(" torque radius shapes", tr.shape, tro.shape, "max diff", np.abs(tr - tro).max(), "per joint", np.abs(tr - tro).max(axis=1))
    lg, lgo = nlp.link_generators()[0], oracle.link_generators()
    print(" link gens max diff", np.abs(lg - lgo).max(), "per link", np.abs(lg - lgo).reshape(T, n, -1).max(axis=(0, 2)))
    for which in ("link", "torque"):
        for i in range(n):
            bad = []
            for tt in range(0, T, max(1, T // 10)):
                c1, r1, k1, co1 = nlp.pz(which, i, tt)
                c2, r2, k2, co2 = oracle.pz(which, i, tt)
                same = np.array_equal(k1, k2)
                bad.append((tt, len(k1), len(k2), same, float(np.abs(c1 - c2).max()), float(np.abs(r1 - r2).max()), float(np.abs(co1 - co2).max()) if same and len(k1) else -1))
            worst = max(bad, key=lambda x: (not x[3], x[4] + x[5]))
            print(f"  {which} {i}: worst sample (t, n_dev, n_orc, keys equal, dcen, drad, dcoef) = {worst}")
    nlp.close()
