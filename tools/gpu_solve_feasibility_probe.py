"""How often does armour_solve give up (status 3: linearised QP infeasible) on problems that DO have a feasible k?
Random worlds; for every problem the solver reports infeasible, 512 random k plus the corners' neighbourhood are
evaluated with the multi-point launch and re-checked.  Development probe.

    python tools/gpu_solve_feasibility_probe.py [B] [O]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from armour_amd.planner import ArmourNLP  # noqa: E402
from armour_amd.worlds import random_batch  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    O = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    T = 100
    bp = random_batch(900, B, O)
    nlp = ArmourNLP(T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    res = nlp.solve()
    feas = np.array([r["feasible"] for r in res])
    status = np.array([r["status"] for r in res])
    print(f"B={B} O={O}: solver feasible {feas.sum()}, status histogram {np.bincount(status).tolist()}")
    P = 512
    rng = np.random.default_rng(0)
    ks = rng.uniform(-1, 1, (P, B, 7))
    ks[0] = 0
    dk = torch.tensor(ks, device="cuda")
    dg = torch.zeros((P, B, nlp.m), dtype=torch.float64, device="cuda")
    nlp.eval_g_jac_device_multi(dk.data_ptr(), P, dg.data_ptr(), 0)
    torch.cuda.synchronize()
    g = dg.cpu().numpy()
    found = np.zeros(B, bool)
    for p in range(P):
        found |= nlp.finalize_solution(g[p])
    missed = found & ~feas
    print(f"random search finds a feasible k for {found.sum()} problems; solver missed {missed.sum()} of them: {np.nonzero(missed)[0].tolist()}")
    print("status of the missed ones:", status[missed].tolist())
    print(f"start point k=0 feasible for {nlp.finalize_solution(g[0]).sum()} problems")


if __name__ == "__main__":
    main()
