"""dev (run with ARMOUR_KEY128=1): a few fused evaluations + reach-set builds of the configs[4] 8-factor workload, for rocprofv3 passes."""
import os, sys
os.environ.setdefault("ARMOUR_KEY128", "1")
sys.path.insert(0, '/root/repo')
import numpy as np
import torch
from armour_amd.planner import ArmourNLP, default_params, fetch8_robot
from armour_amd.worlds import random_fetch8_problem, random_k
T, O = 100, 100
p = random_fetch8_problem(11, O)
pr = default_params(T); pr.k_range[7] = pr.k_range[6]
nlp = ArmourNLP(robot=fetch8_robot(0.5), params=pr)
for _ in range(3):
    nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
dev = torch.device("cuda", 0)
ks = torch.tensor(random_k(5, 40, n=8).reshape(40, 1, 8), device=dev)
d_g = torch.empty((1, nlp.m), device=dev, dtype=torch.float64)
d_j = torch.empty((1, nlp.m, 8), device=dev, dtype=torch.float64)
st = torch.cuda.Stream(device=dev)
for i in range(40):
    nlp.eval_g_jac_device(ks[i].data_ptr(), d_g.data_ptr(), d_j.data_ptr(), st.cuda_stream)
torch.cuda.synchronize()
print("build ms", nlp.build_ms, "m", nlp.m)
