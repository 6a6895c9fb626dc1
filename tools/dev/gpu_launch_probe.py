"""Dev probe: launch-rate of the P2 kernel on different streams / via hipGraph."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_problem, random_k
T, O = 100, 20
p = random_problem(0, O)
nlp = ArmourNLP(T=T).set_parameters(p['q0'], p['qd0'], p['qdd0'], p['q_des'], p['obstacles'])
dev = torch.device('cuda:0')
K = 2000
ks = torch.tensor(random_k(0, K), device=dev)
dg = torch.empty(nlp.m, device=dev, dtype=torch.float64); dj = torch.empty(nlp.m * 7, device=dev, dtype=torch.float64)
def run(stream, label):
    sh = stream.cuda_stream
    with torch.cuda.stream(stream):
        nlp.eval_g_jac_device_steps(ks.data_ptr(), 200, dg.data_ptr(), dj.data_ptr(), sh)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record(stream)
        nlp.eval_g_jac_device_steps(ks.data_ptr(), K, dg.data_ptr(), dj.data_ptr(), sh)
        t_enq = time.perf_counter() - t0
        e1.record(stream); torch.cuda.synchronize(); t1 = time.perf_counter() - t0
    print(f"{label}: enqueue {t_enq*1e6/K:.2f} us/launch, wall {t1*1e6/K:.2f} us/step, events {e0.elapsed_time(e1)*1e3/K:.2f} us/step", flush=True)
run(torch.cuda.current_stream(), "default stream")
s = torch.cuda.Stream()
run(s, "torch side stream")
run(s, "torch side stream (2nd)")
# own stream of the handle (stream=0 -> handle stream)
t0 = time.perf_counter()
nlp.eval_g_jac_device_steps(ks.data_ptr(), K, dg.data_ptr(), dj.data_ptr(), 0)
t_enq = time.perf_counter() - t0
nlp.eval_g_jac(np.zeros(7))  # syncs the handle stream
print(f"handle stream: enqueue {t_enq*1e6/K:.2f} us/launch, wall incl sync {(time.perf_counter()-t0)*1e6/K:.2f}", flush=True)
# graph capture
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    nlp.eval_g_jac_device_steps(ks.data_ptr(), K, dg.data_ptr(), dj.data_ptr(), s.cuda_stream)
torch.cuda.synchronize()
for rep in range(3):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    with torch.cuda.stream(s):
        e0.record(s); g.replay(); e1.record(s)
    torch.cuda.synchronize(); t1 = time.perf_counter() - t0
    print(f"graph replay: wall {t1*1e6/K:.2f} us/step, events {e0.elapsed_time(e1)*1e3/K:.2f} us/step", flush=True)
