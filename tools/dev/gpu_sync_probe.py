"""Dev probe: where does the synchronous armour_eval_g_jac call spend its time?"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from armour_amd.planner import ArmourNLP, _dp
from armour_amd.worlds import random_problem, random_k
T, O = 100, 20
p = random_problem(0, O)
nlp = ArmourNLP(T=T).set_parameters(p['q0'], p['qd0'], p['qdd0'], p['q_des'], p['obstacles'])
L, h = nlp.L, nlp.h
k = nlp._pinned("k", (1, 7)); k[...] = random_k(0, 1)
g = nlp._pinned("g", (1, nlp.m)); jac = nlp._pinned("jac", (1, nlp.m, 7))
kp, gp, jp = _dp(k), _dp(g), _dp(jac)
def t(label, fn, n=200):
    for _ in range(20): fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    print(f"{label:50s} {(time.perf_counter()-t0)/n*1e6:8.2f} us")
t("pinned g+jac (zero-copy)", lambda: L.armour_eval_g_jac(h, kp, gp, jp))
t("pinned g only", lambda: L.armour_eval_g_jac(h, kp, gp, None))
t("pinned jac only", lambda: L.armour_eval_g_jac(h, kp, None, jp))
kk = np.array(k); gg = np.zeros((1, nlp.m)); jj = np.zeros((1, nlp.m, 7))
t("pageable g+jac", lambda: L.armour_eval_g_jac(h, _dp(kk), _dp(gg), _dp(jj)))
t("pageable g only", lambda: L.armour_eval_g_jac(h, _dp(kk), _dp(gg), None))
dev = torch.device('cuda:0')
dk = torch.tensor(kk, device=dev); dg = torch.empty(nlp.m, device=dev, dtype=torch.float64); dj = torch.empty(nlp.m * 7, device=dev, dtype=torch.float64)
st = torch.cuda.Stream()
def devsync():
    L.armour_eval_g_jac_device(h, dk.data_ptr(), dg.data_ptr(), dj.data_ptr(), st.cuda_stream); st.synchronize()
t("device buffers + stream.synchronize()", devsync)
def devspin():
    L.armour_eval_g_jac_device(h, dk.data_ptr(), dg.data_ptr(), dj.data_ptr(), st.cuda_stream)
    while not st.query(): pass
t("device buffers + spin on query", devspin)
t("ctypes call overhead (armour_get_build_ms)", lambda: nlp.build_ms)
