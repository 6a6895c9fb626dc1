"""dev: reach-set builds of ONE process under several per-handle option settings: best build ms of several, and whether every table
(keys / coefficients / centres AND radii) equals the first setting's bit for bit.
    python tools/gpu_opt_ab.py B[,B...] "opt=val[;opt=val]" "opt=val" ...        ("-" = defaults)"""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_batch

def tables(nlp, B):
    out = [nlp.torque_radius().ravel(), nlp.link_generators().ravel()]
    for b in sorted({0, B - 1}):
        for which, cnt in (("link", nlp.J), ("torque", nlp.n)):
            for i in range(cnt):
                for t in range(0, nlp.T, 3):
                    cen, ind, keys, co = nlp.pz(which, i, t, b=b)
                    out += [cen.ravel(), keys.astype(np.float64).ravel(), co.ravel(), ind.ravel()]
    return np.concatenate(out)

Bs = [int(x) for x in sys.argv[1].split(",")]
settings = sys.argv[2:] or ["-"]
for B in Bs:
    bp = random_batch(5, B, 20)
    base = None
    best = {s: 1e9 for s in settings}
    same = {}
    keeper = ArmourNLP(T=100)   # holds the device's work arena, so that every setting builds in the same memory
    keeper.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    for rnd in range(3):
        for st in (settings if rnd % 2 == 0 else settings[::-1]):
            nlp = ArmourNLP(T=100)
            if st != "-":
                for kv in st.split(";"):
                    k, v = kv.split("="); nlp.set_option(int(k), float(v))
            for _ in range(4 if B <= 16 else 3):
                nlp.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"]); best[st] = min(best[st], nlp.build_ms)
            if rnd == 0:
                tb = tables(nlp, B)
                if base is None: base = tb
                same[st] = bool(np.array_equal(tb, base))
            info = nlp.build_info()
            nlp.close()
    keeper.close()
    print(f"B={B} ({info['kernel']}, {info['waves']} waves): " + " | ".join(f"[{s}] {best[s]:.3f} ms{'' if same[s] else ' DIFFERENT TABLES'}" for s in settings), flush=True)
