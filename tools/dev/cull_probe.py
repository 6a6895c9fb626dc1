"""dev: armour_eval_violations_device over every row against the culled form (ARMOUR_OPT_CULL_ROWS = 1): us per call, identical records.
    python tools/dev/cull_probe.py [B] [O]"""
import sys, time
sys.path.insert(0, '/root/repo')
import ctypes as C
import numpy as np
import torch
from armour_amd import _lib
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_batch, random_k
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
O = int(sys.argv[2]) if len(sys.argv) > 2 else 50
bp = random_batch(1000, B, O)
dev = torch.device("cuda", 0)
out = {}
for tag, cull in (("full", 0), ("culled", 1)):
    nlp = ArmourNLP(T=100).set_option(_lib.OPT_CULL_ROWS, cull).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    ks = torch.tensor(random_k(3, 40 * B).reshape(40, B, nlp.n), device=dev)
    rec = torch.zeros((B, 32), dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream(device=dev)
    for i in range(5):
        nlp.eval_violations_device(ks[i].data_ptr(), rec.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for i in range(40):
        nlp.eval_violations_device(ks[i].data_ptr(), rec.data_ptr(), st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 40
    host = nlp.eval_violations(ks[-1].cpu().numpy())
    rel, cnt, ms = nlp.row_relevance()
    out[tag] = (us, host)
    print(f"{tag}: {us:.1f} us per call (B = {B}, O = {O}); relevant collision rows mean {cnt.mean():.0f} of {nlp.J * nlp.T * O}, relevance test {ms:.3f} ms, build {nlp.build_ms:.2f} ms", flush=True)
    nlp.close()
same = all(a["l1_violation"] == c["l1_violation"] and a["n_violated"] == c["n_violated"] and a["feasible"] == c["feasible"] for a, c in zip(out["full"][1], out["culled"][1]))
print("records identical:", same, "speed-up", round(out["full"][0] / out["culled"][0], 2))
