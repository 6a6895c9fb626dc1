"""Shader-clock cost of single PZ operators on the device (armour_debug_pz_op reports the operator's own cycles):
cycles against raw terms for the operators the RNEA chain is made of.  Development tool.

    python tools/dev/gpu_pzop_cost.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from armour_amd.planner import ArmourNLP  # noqa: E402

rng = np.random.default_rng(0)
n = 7


def rand_keys(M, with_err=True):
    """M distinct sorted keys: products of k-powers, some carrying an error variable (bits above 2n)."""
    ks = set()
    while len(ks) < M:
        key = 0
        for i in rng.choice(n, size=rng.integers(1, 4), replace=False):
            key += int(rng.integers(1, 3)) << (2 * int(i))
        if with_err and rng.random() < 0.5:
            key += 1 << (2 * n + int(rng.integers(0, 3 * n)))
        ks.add(key)
    return np.array(sorted(ks), dtype=np.uint64)


def vec(M, sz=3):
    return dict(sz=sz, keys=rand_keys(M), coef=rng.normal(size=(M, sz)) * 0.1, cen=rng.normal(size=sz), ind=np.abs(rng.normal(size=sz)) * 1e-3)


def vec_low(M, sz=3, nvar=4):
    """M distinct keys over few variables with degree <= 1 each plus one error variable: products of two such operands
    collide often (runs of equal keys in the sorted raw terms, as in the RNEA chain)."""
    ks = set()
    while len(ks) < M:
        key = 0
        for i in range(nvar):
            if rng.random() < 0.5:
                key += 1 << (2 * i)
        if rng.random() < 0.7:
            key += 1 << (2 * n + int(rng.integers(0, 8)))
        if key:
            ks.add(key)
    return dict(sz=sz, keys=np.array(sorted(ks), dtype=np.uint64), coef=rng.normal(size=(M, sz)) * 0.1, cen=rng.normal(size=sz), ind=np.abs(rng.normal(size=sz)) * 1e-3)


def rot(i=2):
    keys = np.array(sorted([1 << (2 * i), 1 << (5 * n + 2 * i), 1 << (7 * n + 2 * i)]), dtype=np.uint64)
    return dict(sz=9, keys=keys, coef=rng.normal(size=(3, 9)) * 0.05, cen=rng.normal(size=9), ind=np.zeros(9))


nlp = ArmourNLP(T=100)


def run(name, op, operands, consts=None, reps=5):
    best = None
    for _ in range(reps):
        r = nlp.debug_pz_op(op, operands, consts=consts)
        best = r if best is None or r["cycles"] < best["cycles"] else best
    raw = max(best["raw_terms"], 1)
    print(f"{name:34s} raw terms {best['raw_terms']:5d} -> {len(best['keys']):4d} monomials: {best['cycles']:9.0f} cycles, {best['cycles'] / raw:7.1f} per raw term, "
          f"{best['cycles'] / ((raw + 63) // 64):8.0f} per 64-term pass", flush=True)
    pr = best["prof"]
    if pr.any():   # -DP1_PROFILE build: where the cycles went (slots of pz_wave.h)
        names = {0: "fill", 1: "sort", 2: "emit", 3: "abs_sum", 15: "rank", 16: "bitonic", 17: "lin-merge", 18: "mul-merge",
                 19: "emit:head", 20: "emit:coef", 21: "emit:run", 22: "emit:prune", 23: "emit:store"}
        # (hash-classified products re-use the slots: fill = entry, rank = table clear, lin-merge = key loads of pass A, mul-merge = pass A,
        #  emit:head = operand loads of pass B, emit:coef = pass B, bitonic = pass D, emit:run = sort of the final list, emit:store = pass E,
        #  emit:prune = centre / radii)
        acc = sum(pr[i] for i in (0, 1, 2, 3))
        inside = pr[8] + pr[9] + pr[10]   # whole-call cycles measured inside the operator function(s)
        print("      " + ", ".join(f"{nm} {pr[i]:.0f}" for i, nm in names.items() if pr[i]) + f" | inside the operator functions {inside:.0f} "
              f"(unattributed there {inside - acc:.0f}), outside (call, slot allocation, views) {best['cycles'] - inside:.0f}", flush=True)


for M in (4, 15, 40, 100, 200, 400):
    run(f"mulMV  R(3) x v({M})", 0, [rot(), vec(M)])
for M in (4, 15, 40, 100, 200, 400):
    run(f"add    v({M}) + v({M})", 4, [vec(M), vec(M)])
for M in (15, 100, 400):
    run(f"crossPzMat v({M}) x const", 8, [vec(M)], consts=[0.1, -0.2, 0.3])
for Ma, Mb in ((3, 3), (7, 7), (7, 60), (15, 15), (30, 30), (40, 40), (20, 80), (80, 20), (300, 2), (2, 300)):
    run(f"crossPzPz v({Ma}) x v({Mb})", 10, [vec(Ma), vec(Mb)])
for Ma, Mb in ((30, 30), (40, 40)):
    run(f"crossPzPz colliding v({Ma}) x v({Mb})", 10, [vec_low(Ma), vec_low(Mb)])
for M in (40, 100):
    run(f"add colliding v({M}) + v({M})", 4, [vec_low(M, nvar=5), vec_low(M, nvar=5)])
for M in (15, 100, 400):
    run(f"mulSV  s(0) x v({M})  (presorted)", 3, [dict(sz=1, keys=np.zeros(0, np.uint64), coef=np.zeros((0, 1)), cen=[2.0], ind=[0.0]), vec(M)])
