"""Dev: throughput of the batched tracking controller (includes H2D/D2H and the per-call model preparation)."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from armour_amd.controller import kinova_controller
rng = np.random.default_rng(0)
for B in (1, 1000, 100000, 1000000):
    q = rng.uniform(-np.pi, np.pi, (B, 7)); qd = rng.uniform(-1, 1, (B, 7))
    a = (q, qd, q + 0.01, qd + 0.02, rng.uniform(-2, 2, (B, 7)))
    kinova_controller(10.0, 1.0, 1e-2, 1e-10, *a)
    t0 = time.perf_counter(); kinova_controller(10.0, 1.0, 1e-2, 1e-10, *a); dt = time.perf_counter() - t0
    print(f"B={B}: {dt*1e3:.3f} ms per call, {B/dt:.3g} states/s")
