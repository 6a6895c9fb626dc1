"""Dev: wall time of handle creation and of the first / second set_parameters for a few batch sizes."""
import sys, time
sys.path.insert(0, '/root/repo')
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_batch
for B in (1, 9, 128):
    pb = random_batch(0, B, 6)
    t0 = time.perf_counter(); nlp = ArmourNLP(T=100); t1 = time.perf_counter()
    nlp.set_parameters(pb['q0'], pb['qd0'], pb['qdd0'], pb['q_des'], pb['obstacles']); t2 = time.perf_counter()
    nlp.set_parameters(pb['q0'], pb['qd0'], pb['qdd0'], pb['q_des'], pb['obstacles']); t3 = time.perf_counter()
    print(f"B={B}: create {1e3*(t1-t0):.1f} ms, first set_parameters {1e3*(t2-t1):.1f} ms, second {1e3*(t3-t2):.1f} ms (device {nlp.build_ms:.1f} ms)", flush=True)
    t4 = time.perf_counter(); nlp.close(); print(f"      close {1e3*(time.perf_counter()-t4):.1f} ms")
