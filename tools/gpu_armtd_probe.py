"""ARMTD comparison mode on the GPU: reach-set build time, fused-eval launch time and solve time (development probe).

    python tools/gpu_armtd_probe.py [B ...]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from armour_amd.planner import ArmourNLP  # noqa: E402
from armour_amd.worlds import random_k, random_problem, synthetic_offline_jrs  # noqa: E402

T, O = 100, 20


def main():
    for B in [int(a) for a in sys.argv[1:]] or [1, 16, 128]:
        ps = [random_problem(500 + b, O) for b in range(B)]
        tabs = [synthetic_offline_jrs(p["qd0"], T) for p in ps]
        st = lambda key: np.stack([p[key] for p in ps])
        nlp = ArmourNLP(T=T)
        args = (st("q0"), st("qd0"), st("q_des"), np.stack([t[0] for t in tabs]), np.stack([t[1] for t in tabs]), st("obstacles"))
        nlp.set_parameters_armtd(*args)
        t0 = time.perf_counter()
        nlp.set_parameters_armtd(*args)
        wall = (time.perf_counter() - t0) * 1e3
        dk = torch.tensor(random_k(1, B), device="cuda")
        dg = torch.zeros((B, nlp.m), dtype=torch.float64, device="cuda")
        dj = torch.zeros((B, nlp.m, 7), dtype=torch.float64, device="cuda")
        stream = torch.cuda.Stream()   # a real stream: 0 would select the handle's own stream, which torch events do not see
        s = stream.cuda_stream
        torch.cuda.synchronize()
        torch.cuda.set_stream(stream)
        for _ in range(50):
            nlp.eval_g_jac_device(dk.data_ptr(), dg.data_ptr(), dj.data_ptr(), s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(500):
            nlp.eval_g_jac_device(dk.data_ptr(), dg.data_ptr(), dj.data_ptr(), s)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 500
        nlp.solve()
        t0 = time.perf_counter()
        res = nlp.solve()
        solve_ms = (time.perf_counter() - t0) * 1e3
        ts = nlp.table_sizes()
        print(f"B={B}: build {nlp.build_ms:.3f} ms device / {wall:.3f} ms wall; eval {us:.2f} us/launch ({us / B:.2f} us/problem); "
              f"solve {solve_ms:.2f} ms ({sum(r['feasible'] for r in res)}/{B} feasible, {np.mean([r['iterations'] for r in res]):.1f} iterations); "
              f"status {np.bincount([r['status'] for r in res]).tolist()}; max_link {ts['max_link']}, m {nlp.m}", flush=True)


if __name__ == "__main__":
    main()
