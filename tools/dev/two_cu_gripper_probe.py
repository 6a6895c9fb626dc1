import os, sys
sys.path.insert(0, os.getcwd())
from armour_amd.planner import ArmourNLP, default_params, kinova_gripper_robot
from armour_amd.worlds import random_problem
p = random_problem(5, 20)
for lvl, lean in ((0, 1), (2, 1), (3, 0), (3, 1), (3, 2), (3, 3), (0, 1), (2, 1), (3, 1)):
    nlp = ArmourNLP(robot=kinova_gripper_robot(), params=default_params(100)); nlp.set_option(123, lvl); nlp.set_option(124, lean)
    ms = []
    for _ in range(6):
        nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"]); ms.append(nlp.build_ms)
    print("Kinova with gripper (J = %d): ARMOUR_OPT_P1_STEP_TWO_CU = %d, _LEAN_BACK = %d: build %.3f ms" % (nlp.J, lvl, lean, min(ms)), flush=True)
    nlp.close()
