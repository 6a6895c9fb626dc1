"""dev (ARMOUR_KEY128=1): the "Fetch 8-DOF" preset, payload +-50 %, O obstacles, T = 100: reach-set build ms (per-step kernel, B = 1), eval us,
optional oracle parity.   python tools/dev/fetch8_probe.py [O] [check] [option_id=value ...]"""
import os, sys, time
os.environ.setdefault("ARMOUR_KEY128", "1")
sys.path.insert(0, '/root/repo')
import numpy as np
from armour_amd import _lib
from armour_amd.planner import ArmourNLP, default_params, fetch8_robot
from armour_amd.worlds import random_fetch8_problem
O = int(sys.argv[1]) if len(sys.argv) > 1 else 100
check = len(sys.argv) > 2 and sys.argv[2] == "check"
T = 100
p = random_fetch8_problem(11, O)
pr = default_params(T); pr.k_range[7] = pr.k_range[6]
nlp = ArmourNLP(robot=fetch8_robot(0.5), params=pr)
for kv in sys.argv[3:]:
    nlp.set_option(int(kv.split("=")[0]), float(kv.split("=")[1]))
ms = []
for _ in range(4):
    nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"]); ms.append(nlp.build_ms)
print("fetch8 O =", O, "m =", nlp.m, "build ms", [round(x, 3) for x in ms], nlp.build_info(), nlp.table_sizes(), flush=True)
k = np.random.default_rng(3).uniform(-1, 1, (1, 8))
g, jac = nlp.eval_g_jac(k)
t0 = time.perf_counter()
for _ in range(50): nlp.eval_g_jac(k)
print("sync eval us", (time.perf_counter() - t0) / 50 * 1e6)
if check:
    from oracle import cpu_oracle as orc
    po = orc.default_params(T); po.k_range[7] = po.k_range[6]
    o = orc.Oracle(robot=orc.fetch8_robot(0.5), params=po).set_problem(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    g_ref, jac_ref = o.eval_g_jac(k[0])
    print("oracle build ms", round(o.build_ms, 1), "|dg|", np.abs(g[0] - g_ref).max(), "|djac|", np.abs(jac[0] - jac_ref).max(), "stats", o.stats(), "min margin", o.min_margin())
