"""Dev probe: WHAT in bench.py's process makes the page-locked synchronous call slow (0.8-1.0 ms there, 35-55 us in a bare process)?
usage: gpu_pinned_probe2.py <stage>   0 bare | 1 + import torch, torch.cuda.init | 2 + a torch stream and a device tensor |
                                      3 + the graph of 20 steps (armour_prepare_steps) | 4 + 20 graph launches | 5 + 20 pageable calls first"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
stage = int(sys.argv[1])
if stage >= 1:
    import torch
    torch.cuda.init(); torch.cuda.set_device(0)
from armour_amd.planner import ArmourNLP, _dp
from armour_amd.worlds import random_problem, random_k
T, O = 100, 20
p = random_problem(0, O)
nlp = ArmourNLP(T=T).set_parameters(p['q0'], p['qd0'], p['qdd0'], p['q_des'], p['obstacles'])
L, h = nlp.L, nlp.h
if stage >= 2:
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    ks = torch.tensor(random_k(1, 25).reshape(25, 1, 7), device=dev)
    dg = torch.empty((1, nlp.m), device=dev, dtype=torch.float64); dj = torch.empty((1, nlp.m, 7), device=dev, dtype=torch.float64)
if stage >= 3:
    nlp.prepare_steps(ks[5:].data_ptr(), 20, dg.data_ptr(), dj.data_ptr())
if stage >= 4:
    for _ in range(20):
        nlp.eval_g_jac_device_steps(ks[5:].data_ptr(), 20, dg.data_ptr(), dj.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
kk = random_k(0, 1)
if stage >= 5:
    for _ in range(20):
        nlp.eval_g_jac(kk)
ts = []
for i in range(40):
    t1 = time.perf_counter(); nlp.eval_g_jac(kk, pinned=True); ts.append((time.perf_counter() - t1) * 1e6)
tp = []
for i in range(40):
    t1 = time.perf_counter(); nlp.eval_g_jac(kk); tp.append((time.perf_counter() - t1) * 1e6)
print(f"stage {stage}: pinned median {np.median(ts[2:]):.1f} us (calls 1-4: " + " ".join(f"{x:.0f}" for x in ts[:4]) + f"), pageable median {np.median(tp):.1f} us")
