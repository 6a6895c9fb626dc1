import os, sys
sys.path.insert(0, os.getcwd())
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import reference_sample_problem, random_problem
for name, p in (("sample problem", reference_sample_problem()), ("random world, seed 5", random_problem(5, 20))):
    for T in (100, 128):
        for lvl in (0, 3):
            nlp = ArmourNLP(T=T); nlp.set_option(123, lvl)
            ms = []
            for _ in range(8):
                nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"]); ms.append(nlp.build_ms)
            print("%s, T = %d, ARMOUR_OPT_P1_STEP_TWO_CU = %d: build %.3f ms (best of 8)" % (name, T, lvl, min(ms)), flush=True)
            nlp.close()
