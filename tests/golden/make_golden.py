"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/) -- run from the repo root:

    python tests/golden/make_golden.py

The reference (roahmlab/armour) ships no golden vectors for this path and cannot be built or imported here
(SURVEY.md 8c), so these fixtures pin the ORACLE's own outputs on the reference's known inputs:
  sample   : the commented sample problem of RT/armour_main.cu:18-33 (10 obstacles)
  debug    : the initial state of RT/debug_script.m:29-31, no obstacles
  scene013 : kinova_src/saved_worlds/random/scene_013_001.csv (start row as q0, goal row as q_des, 6 boxes);
             the csv itself is committed next to this script as data
each at T = 100 (BASELINE configs) and T = 128 (RT/Parameters.h:17), evaluated at k = 0 and at the slice point
of RT/PZ_tests.cu:198.  Stored per case: torque_radius, link generators, per-(l,t)/(j,t) monomial counts and
keys, bounds, g, and the Jacobian rows of the torque block, the limit block and 400 seeded collision rows;
plus min_margin = the closest any monomial norm came to SIMPLIFY_THRESHOLD (relative), so that a prune flip is
detectable rather than silent.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from armour_amd.worlds import load_scene_csv  # noqa: E402
from helpers import DEBUG_STATE, PZ_TESTS_K, SAMPLE_PROBLEM  # noqa: E402
from oracle.cpu_oracle import Oracle  # noqa: E402


def cases():
    z = np.zeros(7)
    yield "sample", SAMPLE_PROBLEM
    yield "debug", dict(q0=DEBUG_STATE["q0"], qd0=DEBUG_STATE["qd0"], qdd0=DEBUG_STATE["qdd0"], q_des=DEBUG_STATE["q0"] + 0.1, obstacles=np.zeros((0, 12)))
    qs, qg, obs = load_scene_csv(os.path.join(HERE, "scene_013_001.csv"))
    yield "scene013", dict(q0=qs, qd0=z, qdd0=z, q_des=qg, obstacles=obs)


def jac_rows(m, T, n=7, count=400):
    rng = np.random.default_rng(1234)
    first, last = n * T, m - 4 * n
    col = np.sort(rng.choice(np.arange(first, last), size=min(count, last - first), replace=False)) if last > first else np.zeros(0, int)
    return np.concatenate([np.arange(0, first), col, np.arange(last, m)]).astype(np.int64)


def main():
    for name, p in cases():
        for T in (100, 128):
            o = Oracle(T=T).set_problem(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
            rows = jac_rows(o.m, T)
            out = dict(T=T, q0=p["q0"], qd0=p["qd0"], qdd0=p["qdd0"], q_des=p["q_des"], obstacles=p["obstacles"],
                       torque_radius=o.torque_radius(), link_gens=o.link_generators(), jac_rows=rows,
                       min_margin=o.min_margin())
            xl, xu, gl, gu = o.bounds()
            out.update(g_l=gl, g_u=gu)
            lc, lk, tc, tk = [], [], [], []
            for l in range(7):
                for t in range(T):
                    keys = o.pz("link", l, t)[2]
                    lc.append(len(keys)); lk.append(keys)
                    keys = o.pz("torque", l, t)[2]
                    tc.append(len(keys)); tk.append(keys)
            out.update(link_count=np.array(lc, np.int32), link_keys=np.concatenate(lk).astype(np.uint32),
                       torque_count=np.array(tc, np.int32), torque_keys=np.concatenate(tk).astype(np.uint32))
            for tag, k in (("k0", np.zeros(7)), ("kt", PZ_TESTS_K)):
                g, jac = o.eval_g_jac(k)
                out[f"g_{tag}"] = g
                out[f"jac_{tag}"] = jac[rows]
                out[f"f_{tag}"] = o.eval_f(k)
                out[f"gradf_{tag}"] = o.eval_grad_f(k)
            path = os.path.join(HERE, f"{name}_T{T}.npz")
            np.savez_compressed(path, **out)
            print(path, os.path.getsize(path) // 1024, "KiB", "m =", o.m, "min_margin =", out["min_margin"])


def main_armtd():
    """ARMTD comparison mode (CMP/): the sample problem of CMP/armtd_main.cu:17-33 (the same numbers as RT's) with the
    stand-in offline tables of armour_amd.worlds.synthetic_offline_jrs (the reference's .mat tables are not in its
    checkout); the tables are stored in the fixture, so it does not depend on that generator staying as it is."""
    from armour_amd.worlds import synthetic_offline_jrs
    T = 100
    p = SAMPLE_PROBLEM
    jrs, k_range = synthetic_offline_jrs(p["qd0"], T)
    o = Oracle(T=T).set_problem_armtd(p["q0"], p["qd0"], p["q_des"], jrs, k_range, p["obstacles"])
    rng = np.random.default_rng(4321)
    Q = o.m - 28
    rows = np.concatenate([np.sort(rng.choice(Q, size=400, replace=False)), np.arange(Q, o.m)]).astype(np.int64)
    out = dict(T=T, q0=p["q0"], qd0=p["qd0"], q_des=p["q_des"], obstacles=p["obstacles"], jrs=jrs, k_range=k_range,
               link_gens=o.link_generators(), jac_rows=rows, min_margin=o.min_margin())
    xl, xu, gl, gu = o.bounds()
    out.update(g_l=gl, g_u=gu)
    lc, lk = [], []
    for l in range(7):
        for t in range(T):
            keys = o.pz("link", l, t)[2]
            lc.append(len(keys)); lk.append(keys)
    out.update(link_count=np.array(lc, np.int32), link_keys=np.concatenate(lk).astype(np.uint32))
    for tag, k in (("k0", np.zeros(7)), ("kt", PZ_TESTS_K)):
        g, jac = o.eval_g_jac(k)
        out[f"g_{tag}"] = g
        out[f"jac_{tag}"] = jac[rows]
        out[f"f_{tag}"] = o.eval_f(k)
        out[f"gradf_{tag}"] = o.eval_grad_f(k)
    path = os.path.join(HERE, f"armtd_sample_T{T}.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB", "m =", o.m, "min_margin =", out["min_margin"])


if __name__ == "__main__":
    if "--armtd-only" not in sys.argv:
        main()
    main_armtd()
