import os, sys
sys.path.insert(0, os.getcwd())
os.environ["ARMOUR_P1_TRACE"] = "1" if len(sys.argv) > 1 else ""
if not os.environ["ARMOUR_P1_TRACE"]: del os.environ["ARMOUR_P1_TRACE"]
from armour_amd.planner import ArmourNLP, default_params, fetch_robot
from armour_amd.worlds import random_fetch_problem
p = random_fetch_problem(11, 100)
for lvl in (0, 1, 2, 3):
    nlp = ArmourNLP(robot=fetch_robot(0.5), params=default_params(100)); nlp.set_option(123, lvl)
    ms = []
    for _ in range(6):
        nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"]); ms.append(nlp.build_ms)
    print("Fetch preset (J = %d, n = %d), O = 100: ARMOUR_OPT_P1_STEP_TWO_CU = %d: build %.3f ms, planes %.3f" % (nlp.J, nlp.n, lvl, min(ms), 0.0), flush=True)
    nlp.close()
