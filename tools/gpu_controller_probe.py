"""Dev: the tracking controller through the host-pointer ABI -- per-call time for one state (the way the simulator calls
the MEX it replaces) and throughput for batches (H2D / D2H included)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from armour_amd import _lib  # noqa: E402
from armour_amd.controller import kinova_controller  # noqa: E402
from armour_amd.planner import kinova_robot  # noqa: E402

rng = np.random.default_rng(0)
L = _lib.load()
rb = kinova_robot()
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
for B in [int(x) for x in sys.argv[1:]] or (1, 64, 1000, 4096, 16384, 65536, 100000, 1000000):
    q = rng.uniform(-np.pi, np.pi, (B, 7)); qd = rng.uniform(-1, 1, (B, 7))
    a = [np.ascontiguousarray(x) for x in (q, qd, q + 0.01, qd + 0.02, rng.uniform(-2, 2, (B, 7)))]
    kr = np.full(7, 10.0)
    u, tau, v = np.zeros((B, 7)), np.zeros((B, 7)), np.zeros((B, 7))
    call = lambda: L.armour_robust_controller(C.byref(rb), 0.03, dp(kr), 1.0, 1e-2, 1e-10, B, *[dp(x) for x in a], dp(u), dp(tau), dp(v))
    assert call() == 0
    reps = 200 if B <= 1000 else 5
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter(); kinova_controller(10.0, 1.0, 1e-2, 1e-10, *a); dw = time.perf_counter() - t0
    print(f"B={B}: {dt * 1e6:.1f} us per call through the C ABI ({B / dt:.3g} states/s); {dw * 1e6:.1f} us through the Python wrapper")
