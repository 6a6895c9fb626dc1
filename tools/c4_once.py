"""One process that evaluates BASELINE configs[4] (Fetch preset, last link +-50 % payload, O = 100, T = 100, one problem) K times at fresh k
through the graph entry bench.py times (for rocprofv3: every armour_p2_eval_kernel dispatch of this process is a configs[4] launch)."""
import sys
sys.path.insert(0, '/root/repo')
import torch
from armour_amd.planner import ArmourNLP, default_params, fetch_robot
from armour_amd.worlds import random_fetch_problem, random_k
K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
p = random_fetch_problem(11, 100)
nlp = ArmourNLP(robot=fetch_robot(0.5), params=default_params(100))
nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
dev = torch.device("cuda", 0)
ks = torch.tensor(random_k(311, K).reshape(K, 1, nlp.n), device=dev)
d_g = torch.empty((1, nlp.m), device=dev, dtype=torch.float64)
d_j = torch.empty((1, nlp.m, nlp.n), device=dev, dtype=torch.float64)
st = torch.cuda.Stream(device=dev)
for _ in range(3):
    nlp.eval_g_jac_device_steps(ks.data_ptr(), K, d_g.data_ptr(), d_j.data_ptr(), st.cuda_stream)
st.synchronize()
print("configs[4]:", K, "steps x 3, m =", nlp.m, "finite:", bool(torch.isfinite(d_g).all()))
nlp.close()
