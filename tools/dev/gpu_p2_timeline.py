"""Dev probe: build with EXTRA="-DP2_ABLATE=6 -DP2_TIMELINE"; prints phase stamps (10 ns ticks) of three collision blocks."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_problem, random_k
T, O = 100, 20
p = random_problem(0, O)
nlp = ArmourNLP(T=T).set_parameters(p['q0'], p['qd0'], p['qdd0'], p['q_des'], p['obstacles'])
dev = torch.device('cuda:0')
K = 64
ks = torch.tensor(random_k(0, K), device=dev)
dg = torch.zeros((K, nlp.m), device=dev, dtype=torch.float64); dj = torch.zeros((K, nlp.m, 7), device=dev, dtype=torch.float64)
st = torch.cuda.Stream()
for rep in range(2):
    for s in range(K):
        nlp.eval_g_jac_device(ks[s].data_ptr(), dg[s].data_ptr(), dj[s].data_ptr(), st.cuda_stream)
    st.synchronize()
gfull = dg.cpu().numpy()
g = gfull[:, 7 * T + 7 * T * O:]
raw = g[:, :24].reshape(K, 3, 8)
stamps = raw[:, :, :7]
print('shader cycles per block / (end-start) ns -> GHz:', np.median(raw[8:, :, 7] / ((raw[8:, :, 6] - raw[8:, :, 0]) * 10.0), axis=0))
names = ["start", "planes issued", "k+tables+barrier", "mono+barrier", "reduce+barrier", "scan+barrier", "end"]
base = stamps[:, :, 0].min(axis=1)            # earliest of the three blocks, per launch
rel = (stamps - base[:, None, None]) * 10.0   # ns
print("median ns since the earliest block start, per block (first / middle / last collision block):")
for i, nme in enumerate(names):
    print(f"  {nme:18s}", np.median(rel[8:, :, i], axis=0))
prev_end = stamps[:-1, :, 6].max(axis=1)
print("gap from previous launch's last end stamp to this launch's first start: median ns", np.median((base[1:] - prev_end)[8:]) * 10.0)
print("launch-to-launch (start to start) median ns", np.median(np.diff(base)[8:]) * 10.0)

sl = gfull[:, :24].reshape(K, 3, 8)[:, :, :7]
if sl[8:].any():
    rel2 = (sl - base[:, None, None]) * 10.0
    print("slicing-wave view (split kernel):")
    for i, nme in enumerate(names):
        print(f"  {nme:18s}", np.median(rel2[8:, :, i], axis=0))
