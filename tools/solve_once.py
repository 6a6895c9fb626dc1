"""One process that solves the reference's sample problem N times with the device-resident armour_solve (for rocprofv3)."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from helpers import SAMPLE_PROBLEM as p
from armour_amd.planner import ArmourNLP
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nlp = ArmourNLP(T=100).set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
for _ in range(N):
    s = nlp.solve(device_qp=True)[0]   # (one problem would take the host-QP form by itself: the persistent kernel is what is profiled here)
print("solved", N, "times:", {k: s[k] for k in ("feasible", "iterations", "evaluations", "status")})
nlp.close()   # free the handle before interpreter exit (under rocprofv3 a handle freed from the exit handlers crashed inside the tool library)
