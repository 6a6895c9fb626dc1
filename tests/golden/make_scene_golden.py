"""Oracle goldens on the reference's own worlds (armour_amd/scenes.py): first planning iteration of every tenth saved random
scene and of the hard scenarios 4 and 6 (the two with the most boxes), at T = 100 and T = 128 -- run from the repo root:

    python tests/golden/make_scene_golden.py

The reference holds the INPUTS (kinova_src/saved_worlds/random/*.csv, KSI/kinova_scenarios/get_kinova_scenario_info.m) and no
recorded answers, so these fixtures pin the oracle's outputs on them (PARITY UNPINNED), as make_golden.py
does for the sample problem.  Per case: torque radii, link generators of every eighth time step, monomial counts and a digest of the key lists, bounds,
g at k = 0; g and Jacobian rows (torque block, limit block, 200 seeded collision rows) at the slice point of
RT/PZ_tests.cu:198, cost and gradient, and min_margin.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from armour_amd.scenes import hard_scenarios, saved_scenes  # noqa: E402
from helpers import PZ_TESTS_K  # noqa: E402
from oracle.cpu_oracle import Oracle  # noqa: E402


def chosen_worlds():
    s, h = saved_scenes(), hard_scenarios()
    return s[::10] + [h[3], h[5]]


def jac_rows(m, T, n=7, count=200):
    rng = np.random.default_rng(1234)
    first, last = n * T, m - 4 * n
    col = np.sort(rng.choice(np.arange(first, last), size=min(count, last - first), replace=False))
    return np.concatenate([np.arange(0, first), col, np.arange(last, m)]).astype(np.int64)


def key_digest(o, T):
    """(link counts, torque counts, sha256 of all key lists in (kind, joint, time step) order)."""
    h = hashlib.sha256()
    lc, tc = [], []
    for kind, cnt in (("link", lc), ("torque", tc)):
        for i in range(7):
            for t in range(T):
                keys = np.ascontiguousarray(o.pz(kind, i, t)[2], dtype=np.uint64)
                cnt.append(len(keys))
                h.update(keys.tobytes())
    return np.array(lc, np.int32), np.array(tc, np.int32), h.hexdigest()


def record(p, T):
    o = Oracle(T=T).set_problem(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    rows = jac_rows(o.m, T)
    lc, tc, dig = key_digest(o, T)
    _, _, gl, gu = o.bounds()
    g0, _ = o.eval_g_jac(np.zeros(7))
    gt, jt = o.eval_g_jac(PZ_TESTS_K)
    return dict(torque_radius=o.torque_radius(), link_gens=o.link_generators()[::8], link_count=lc, torque_count=tc, key_digest=np.array(dig),
                g_l=gl[:7 * T], g_u=gu[:7 * T], g_k0=g0, g_kt=gt[rows], jac_rows=rows, jac_kt=jt[rows], f_kt=o.eval_f(PZ_TESTS_K),
                gradf_kt=o.eval_grad_f(PZ_TESTS_K), min_margin=o.min_margin(), m=o.m)


def main():
    out = {}
    names = []
    for name, p in chosen_worlds():
        names.append(name)
        for k in ("q0", "q_des", "obstacles"):
            out[f"{name}/{k}"] = p[k]
        for T in (100, 128):
            r = record(p, T)
            for k, v in r.items():
                out[f"{name}/T{T}/{k}"] = v
            print(name, T, "m =", r["m"], "min_margin = %.3e" % r["min_margin"], flush=True)
    out["names"] = np.array(names)
    path = os.path.join(HERE, "reference_scenes.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
