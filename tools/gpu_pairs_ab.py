"""dev: the per-step reach-set build with two waves per operator in the backward pass (ARMOUR_OPT_P1_STEP_PAIRS) against the same build with
one -- in ONE process, two handles: best build ms of several, keys / coefficients / centres bit for bit, radii within 1e-12.
    python tools/gpu_pairs_ab.py [B ...]"""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from armour_amd import _lib
from armour_amd.planner import ArmourNLP
from armour_amd.worlds import random_batch

def tables(nlp, B):
    exact, radii = [], [nlp.torque_radius().ravel(), nlp.link_generators().ravel()]
    for b in sorted({0, B - 1}):
        for which, cnt in (("link", nlp.J), ("torque", nlp.n)):
            for i in range(cnt):
                for t in range(0, nlp.T, 3):
                    cen, ind, keys, co = nlp.pz(which, i, t, b=b)
                    exact += [cen.ravel(), keys.astype(np.float64).ravel(), co.ravel()]
                    radii.append(ind.ravel())
    return np.concatenate(exact), np.concatenate(radii)

for B in [int(x) for x in sys.argv[1:]] or [1, 2, 4]:
    bp = random_batch(5, B, 20)
    res = {}
    for pairs in (0, 1, 1):
        nlp = ArmourNLP(T=100).set_option(_lib.OPT_P1_BUILD, 1).set_option(_lib.OPT_P1_STEP_PAIRS, pairs)
        ms = []
        for _ in range(5):
            nlp.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"]); ms.append(nlp.build_ms)
        ex, ra = tables(nlp, B)
        info = nlp.build_info()
        key = pairs if pairs not in res else 2
        res[key] = (min(ms), ex, ra, info)
        nlp.close()
    a, b, c = res[0], res[1], res[2]
    print(f"B={B}: one wave per operator {a[0]:.3f} ms, pairs {b[0]:.3f} ms ({b[3]['kernel']}, {b[3]['waves']} waves) | exact tables equal: {np.array_equal(a[1], b[1])} "
          f"| radii max diff {np.abs(a[2] - b[2]).max():.3e} (bit-identical: {np.array_equal(a[2], b[2])}) | pairs reproducible: {np.array_equal(b[1], c[1]) and np.array_equal(b[2], c[2])}", flush=True)
