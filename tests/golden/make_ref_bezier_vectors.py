#!/usr/bin/env python3
"""Known-answer vectors of the REFERENCE's own scalar Bezier functions (build container only; never runs on the GPU box).

oracle/_ref/libref_bezier.so is the reference's source text RT/Trajectory.cu:542-822 compiled as it lies under
/root/reference (recipe: `make -C oracle ref`): q_des_func, qd_des_func, qdd_des_func, the four MATLAB-generated
q*_des_extrema{2,3}_k_derivative helpers and the three *_k_indep polynomials.  This script calls that library on seeded
random inputs and records inputs + outputs as data:

  tests/golden/ref_bezier_vectors.npz   x[N,5] = (q0, Tqd0, TTqdd0, k_actual, t)  ->  y[N,10], one column per function
  tests/golden/ref_limit_rows.npz       for the reference's known inputs (sample problem of RT/armour_main.cu:18-33, the
                                        RT/debug_script.m:29-31 state) and 12 points k each: the 4n joint-limit rows of
                                        eval_g and the diagonal of their Jacobian block, assembled from the reference's
                                        functions by the selection logic of RT/Trajectory.cu:256-540 (restated below)

They pin oracle/ and the device's closed forms (armour_amd/csrc/bezier.h) to the reference itself for SURVEY.md 8 row a13.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

NAMES5 = ["q_des_func", "qd_des_func", "qdd_des_func"]                       # (q0, Tqd0, TTqdd0, k, t)
NAMES4 = ["q_des_extrema2_k_derivative", "q_des_extrema3_k_derivative", "qd_des_extrema2_k_derivative", "qd_des_extrema3_k_derivative",
          "q_des_k_indep", "qd_des_k_indep", "qdd_des_k_indep"]              # (q0, Tqd0, TTqdd0, k | s)


def load_ref(path=None):
    """The compiled reference functions as Python callables (C++ symbols: _Z<len><name>d...d)."""
    path = path or os.path.join(ROOT, "oracle", "_ref", "libref_bezier.so")
    L = C.CDLL(path)
    fns = {}
    for name, nargs in [(n, 5) for n in NAMES5] + [(n, 4) for n in NAMES4]:
        f = getattr(L, f"_Z{len(name)}{name}" + "d" * nargs)
        f.restype = C.c_double
        f.argtypes = [C.c_double] * nargs
        fns[name] = f
    return fns


def ref_limit_rows(fns, q0, Tqd0, TTqdd0, k, k_range, duration):
    """returnJoint{Position,Velocity}Extremum and ...Gradient (RT/Trajectory.cu:256-540) on the reference's functions:
    rows [q_min(n), q_max(n), qd_min(n), qd_max(n)] and the diagonal entries of the 4n x n Jacobian block."""
    n = len(q0)
    rows, diag = np.zeros(4 * n), np.zeros(4 * n)
    for i in range(n):
        ka = k_range[i] * k[i]
        a, b = Tqd0[i], TTqdd0[i]
        for vel in (False, True):
            with np.errstate(invalid="ignore", divide="ignore"):
                if not vel:
                    sq = np.sqrt(64 * a ** 2 + 14 * a * b - 120 * ka * a + b ** 2)
                    e2, e3 = (2 * a + b + sq) / (5 * (6 * a - 12 * ka + b)), (2 * a + b - sq) / (5 * (6 * a - 12 * ka + b))
                else:
                    sq = np.sqrt(6 * (150 * ka ** 2 - 180 * ka * a - 20 * ka * b + 54 * a ** 2 + 14 * a * b + b ** 2))
                    e2, e3 = (18 * a - 30 * ka + 4 * b + sq) / (10 * (6 * a - 12 * ka + b)), (18 * a - 30 * ka + 4 * b - sq) / (10 * (6 * a - 12 * ka + b))
            f = fns["qd_des_func" if vel else "q_des_func"]
            v1, v2, v3, v4 = (f(q0[i], a, b, ka, t) for t in (0.0, e2, e3, 1.0))
            if v1 < v4:
                mn, mn_id, mx, mx_id = v1, 1, v4, 4
            else:
                mn, mn_id, mx, mx_id = v4, 4, v1, 1
            if 0 <= e2 <= 1:
                if v2 < mn:
                    mn, mn_id = v2, 2
                if mx < v2:
                    mx, mx_id = v2, 2
            if 0 <= e3 <= 1:
                if v3 < mn:
                    mn, mn_id = v3, 3
                if mx < v3:
                    mx, mx_id = v3, 3
            pre = "qd" if vel else "q"

            def grad(idn):
                if idn == 1:
                    return 0.0
                if idn == 4:
                    return 1.0           # also for the velocity rows (RT/Trajectory.cu:503,521)
                return fns[f"{pre}_des_extrema{idn}_k_derivative"](q0[i], a, b, ka)
            sc = k_range[i] / duration if vel else k_range[i]
            off = 2 * n if vel else 0
            rows[off + i], rows[off + n + i] = (mn / duration, mx / duration) if vel else (mn, mx)
            diag[off + i], diag[off + n + i] = grad(mn_id) * sc, grad(mx_id) * sc
    return rows, diag


def main():
    from helpers import DEBUG_STATE, SAMPLE_PROBLEM
    fns = load_ref()
    rng = np.random.default_rng(20261003)
    N = 600
    # operating range of the planner: |q0| <= pi, Tqd0 = qd0*DURATION within the speed limits, TTqdd0 a few rad/s^2,
    # k_actual = k_range*k with k_range up to pi/24 (RT/Parameters.h:20-22), t and s in [0, 1]
    x = np.column_stack([rng.uniform(-np.pi, np.pi, N), rng.uniform(-1.4, 1.4, N), rng.uniform(-3, 3, N),
                         rng.uniform(-np.pi / 24, np.pi / 24, N), rng.uniform(0, 1, N)])
    x[:8, 1:3] = 0.0          # a start at rest (the sample problem): the stationary-point formulas degenerate (0/0 -> NaN)
    y = np.zeros((N, 10))
    for r in range(N):
        q0, a, b, k, t = x[r]
        y[r, 0:3] = [fns[nm](q0, a, b, k, t) for nm in NAMES5]
        y[r, 3:7] = [fns[nm](q0, a, b, k) for nm in NAMES4[:4]]
        y[r, 7:10] = [fns[nm](q0, a, b, t) for nm in NAMES4[4:]]
    np.savez_compressed(os.path.join(HERE, "ref_bezier_vectors.npz"), x=x, y=y, columns=np.array(NAMES5 + NAMES4))
    # joint-limit rows of the reference's known inputs
    kr, D = np.full(7, np.pi / 48), 1.0
    cases = {"sample": (SAMPLE_PROBLEM["q0"], SAMPLE_PROBLEM["qd0"], SAMPLE_PROBLEM["qdd0"]),
             "debug": (DEBUG_STATE["q0"], DEBUG_STATE["qd0"], DEBUG_STATE["qdd0"])}
    for s in range(3):   # plus three seeded random states of armour_amd.worlds.random_problem
        from armour_amd.worlds import random_problem
        p = random_problem(40 + s, 0)
        cases[f"random{40 + s}"] = (p["q0"], p["qd0"], p["qdd0"])
    out = {}
    for name, (q0, qd0, qdd0) in cases.items():
        ks = np.vstack([np.zeros(7), rng.uniform(-1, 1, (11, 7))])
        rows, diag = zip(*[ref_limit_rows(fns, q0, qd0 * D, qdd0 * D * D, k, kr, D) for k in ks])
        out[f"{name}_state"] = np.stack([q0, qd0, qdd0])
        out[f"{name}_k"], out[f"{name}_rows"], out[f"{name}_diag"] = ks, np.stack(rows), np.stack(diag)
    np.savez_compressed(os.path.join(HERE, "ref_limit_rows.npz"), cases=np.array(list(cases)), **out)
    print("wrote ref_bezier_vectors.npz", x.shape, y.shape, "NaN rows:", int(np.isnan(y).any(axis=1).sum()),
          "and ref_limit_rows.npz", list(cases))


if __name__ == "__main__":
    main()
