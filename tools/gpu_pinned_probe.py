"""Dev probe: the synchronous host call armour_eval_g_jac with page-locked buffers, call by call from a fresh
process -- first allocation, first 50 calls -- in the mode given by ARMOUR_PINNED_MODE (0 staged DMA copies [default], 1 zero copy,
2 zero-copy k only), next to pageable buffers.  Run once per mode: the mode is read once per process."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from armour_amd.planner import ArmourNLP, _dp
from armour_amd.worlds import random_problem, random_k
T, O = 100, 20
p = random_problem(0, O)
nlp = ArmourNLP(T=T).set_parameters(p['q0'], p['qd0'], p['qdd0'], p['q_des'], p['obstacles'])
L, h = nlp.L, nlp.h
t0 = time.perf_counter()
k = nlp._pinned("k", (1, 7)); k[...] = random_k(0, 1)
g = nlp._pinned("g", (1, nlp.m)); jac = nlp._pinned("jac", (1, nlp.m, 7))
alloc_us = (time.perf_counter() - t0) * 1e6
kp, gp, jp = _dp(k), _dp(g), _dp(jac)
ts = []
for i in range(50):
    t1 = time.perf_counter(); L.armour_eval_g_jac(h, kp, gp, jp); ts.append((time.perf_counter() - t1) * 1e6)
kk = np.array(k); gg = np.zeros((1, nlp.m)); jj = np.zeros((1, nlp.m, 7))
tp = []
for i in range(50):
    t1 = time.perf_counter(); L.armour_eval_g_jac(h, _dp(kk), _dp(gg), _dp(jj)); tp.append((time.perf_counter() - t1) * 1e6)
assert np.array_equal(g, gg) and np.array_equal(jac, jj)
print(f"ARMOUR_PINNED_MODE={os.environ.get('ARMOUR_PINNED_MODE', '0')}: alloc {alloc_us:.0f} us; pinned calls 1-5 " + " ".join(f"{x:.0f}" for x in ts[:5])
      + f" | median of 50 {np.median(ts):.1f} us, max {max(ts):.0f} us || pageable median {np.median(tp):.1f} us, max {max(tp):.0f} us")
