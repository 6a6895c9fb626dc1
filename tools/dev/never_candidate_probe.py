"""dev: how many collision rows can EVER pass the solver's candidate filter (slv::row_upper_candidate: g_i + 2 |J_i|_1 > u_i)?  Sampled lower bound:
max over N random k (and the box's corners) of g_i(k) + 2 |J_i(k)|_1 per row, against the relevance mask (rows that can be violated).
    python tools/dev/never_candidate_probe.py B O [N=200]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_batch
B, O = int(sys.argv[1]), int(sys.argv[2]); N = int(sys.argv[3]) if len(sys.argv) > 3 else 200
bp = random_batch(1000, B, O)
nlp = ArmourNLP(T=100).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
xl, xu, lo, hi = nlp.get_bounds_info()
rng = np.random.default_rng(0)
nT = nlp.n * nlp.T; nC = nlp.J * nlp.T * O
worst = np.full((B, nlp.m), -np.inf); worst_g = np.full((B, nlp.m), -np.inf)
for s in range(N):
    k = rng.uniform(-1, 1, (B, nlp.n)) if s % 4 else rng.choice([-1.0, 1.0], (B, nlp.n))
    g, jac = nlp.eval_g_jac(k)
    l1 = np.abs(jac).sum(axis=2)
    worst = np.maximum(worst, g + 2 * l1 - hi.reshape(B, -1)); worst_g = np.maximum(worst_g, g - hi.reshape(B, -1))
rel, cnt, ms = nlp.row_relevance()
col = slice(nT, nT + nC)
print(f"B={B} O={O} N={N}: collision rows {nC}; can be violated (device mask) {rel[:, col].mean():.4f}; sampled violated {np.mean(worst_g[:, col] > 0):.4f}; "
      f"sampled CANDIDATE at some k {np.mean(worst[:, col] > 0):.4f}")
c = (worst[:, col] > 0).reshape(B, nlp.J, nlp.T, O)
print("candidate fraction by link:", np.round(c.mean(axis=(0, 2, 3)), 3))
print("64-row tiles with no candidate row:", np.mean(~(worst[:, col] > 0).reshape(B, -1)[:, : (nC // 64) * 64].reshape(B, -1, 64).any(axis=2)))
tq = worst[:, :nT] > 0
print("torque rows candidate:", tq.mean())
