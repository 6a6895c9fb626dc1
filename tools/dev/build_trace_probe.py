import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ["ARMOUR_P1_TRACE"] = "1"
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import reference_sample_problem
p = reference_sample_problem()
nlp = ArmourNLP(T=100)
for i in range(6):
    t0 = time.perf_counter()
    nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    print("python wall %.1f us, device %.1f us" % ((time.perf_counter() - t0) * 1e6, nlp.build_ms * 1e3), flush=True)
