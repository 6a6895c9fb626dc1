"""The dense QP of armour_solve (Goldfarb-Idnani dual active set, armour_amd/csrc/solver.hip) checked on the CPU
through its test hook by its optimality certificate: for a strictly convex QP the KKT conditions are sufficient,
so no second solver is needed."""
import ctypes as C

import numpy as np
import pytest


def _qp(Gd, g0, A, lo, hi):
    from armour_amd import _lib
    L = _lib.load()
    n, m = len(Gd), A.shape[0]
    dp = C.POINTER(C.c_double)
    x = np.zeros(n)
    feas = C.c_int32(0)
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (Gd, g0, A.reshape(-1) if m else np.zeros(1), lo if m else np.zeros(1), hi if m else np.zeros(1))]
    _lib.check(L.armour_debug_qp(n, *[a.ctypes.data_as(dp) for a in arrs[:2]], m, *[a.ctypes.data_as(dp) for a in arrs[2:]], x.ctypes.data_as(dp), C.byref(feas)))
    return x, bool(feas.value)


def _kkt_certificate(Gd, g0, A, lo, hi, x, tol=1e-7):
    """x is optimal iff it is feasible and -(G x + g0) lies in the cone of the active constraint normals."""
    Ax = A @ x
    assert np.all(Ax <= hi + tol) and np.all(Ax >= lo - tol), "primal infeasible"
    grad = Gd * x + g0
    act_hi = np.abs(Ax - hi) <= 1e-6
    act_lo = np.abs(Ax - lo) <= 1e-6
    N = np.vstack([A[act_hi], -A[act_lo]]) if (act_hi.any() or act_lo.any()) else np.zeros((0, len(x)))
    # stationarity: grad + N' u = 0 with u >= 0  -> NNLS residual
    from scipy.optimize import nnls
    if N.shape[0] == 0:
        assert np.abs(grad).max() <= 1e-6
        return
    u, res = nnls(N.T, -grad)
    assert res <= 1e-6 * (1 + np.abs(grad).max()), f"not stationary: residual {res}"


@pytest.mark.parametrize("seed", range(12))
def test_random_qps_satisfy_kkt(seed):
    rng = np.random.default_rng(seed)
    n = 7
    m = int(rng.integers(5, 4000))
    Gd = rng.uniform(0.05, 3.0, n)
    g0 = rng.normal(size=n) * 3
    A = rng.normal(size=(m, n))
    x_feas = rng.uniform(-0.5, 0.5, n)            # guarantees a non-empty feasible set
    slack = rng.uniform(0.0, 1.0, m)
    hi = A @ x_feas + slack
    lo = np.where(rng.random(m) < 0.3, A @ x_feas - rng.uniform(0.0, 1.0, m), -1e19)
    # box |x| <= 1 as extra rows, like the planner's variable bounds
    A = np.vstack([A, np.eye(n)]); hi = np.concatenate([hi, np.ones(n)]); lo = np.concatenate([lo, -np.ones(n)])
    x, feasible = _qp(Gd, g0, A, lo, hi)
    assert feasible
    _kkt_certificate(Gd, g0, A, lo, hi, x)


def test_unconstrained_and_bound_only():
    Gd = np.array([2.0, 4.0, 1.0]); g0 = np.array([-2.0, 8.0, -10.0])
    x, ok = _qp(Gd, g0, np.zeros((0, 3)), np.zeros(0), np.zeros(0))
    assert ok and np.allclose(x, -g0 / Gd)
    A = np.eye(3); x, ok = _qp(Gd, g0, A, -np.ones(3), np.ones(3))
    assert ok and np.allclose(x, np.clip(-g0 / Gd, -1, 1))     # separable: the clipped minimiser


def test_nearly_parallel_rows_like_adjacent_time_steps():
    """collision rows of neighbouring time intervals are almost parallel: the active set must not cycle"""
    rng = np.random.default_rng(3)
    n = 7
    base = rng.normal(size=n)
    A = np.vstack([base + 1e-6 * rng.normal(size=n) for _ in range(200)] + [np.eye(n)])
    hi = np.concatenate([np.full(200, -0.3) + 1e-7 * rng.normal(size=200), np.ones(n)])
    lo = np.concatenate([np.full(200, -1e19), -np.ones(n)])
    Gd = np.ones(n); g0 = np.zeros(n)
    x, ok = _qp(Gd, g0, A, lo, hi)
    assert ok
    _kkt_certificate(Gd, g0, A, lo, hi, x, tol=1e-6)


def test_inconsistent_constraints_are_reported():
    A = np.array([[1.0, 0, 0], [-1.0, 0, 0]]); hi = np.array([-1.0, -1.0]); lo = np.full(2, -1e19)   # x0 <= -1 and x0 >= 1
    x, ok = _qp(np.ones(3), np.zeros(3), A, lo, hi)
    assert not ok


def _qp_box(Gd, g0, A, lo, hi, x_lo, x_hi, first_try):
    from armour_amd import _lib
    L = _lib.load()
    n, m = len(Gd), A.shape[0]
    dp = C.POINTER(C.c_double)
    x = np.zeros(n)
    feas, steps, mult = C.c_int32(0), C.c_int32(-1), C.c_double(0)
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (Gd, g0, A.reshape(-1) if m else np.zeros(1), lo if m else np.zeros(1), hi if m else np.zeros(1), x_lo, x_hi)]
    p = [a.ctypes.data_as(dp) for a in arrs]
    _lib.check(L.armour_debug_qp_box(n, p[0], p[1], m, p[2], p[3], p[4], p[5], p[6], int(first_try), x.ctypes.data_as(dp), C.byref(feas), C.byref(steps), C.byref(mult)))
    return x, bool(feas.value), steps.value, mult.value


@pytest.mark.parametrize("seed", range(16))
def test_box_clipped_first_try_gives_the_active_set_methods_answer(seed):
    """armour_solve's first try (solver_common.h box_clipped_step): the unconstrained minimiser clipped to the variables' box is the QP's solution whenever
    it satisfies every row.  Against the active-set method alone on the same QP: the same point (1e-9) and the same largest multiplier (1e-7) -- with
    zero steps when no row is in the way (the reference's worlds), and the usual steps when one is."""
    rng = np.random.default_rng(100 + seed)
    n = 7
    Gd = rng.uniform(0.05, 30.0, n)
    g0 = rng.normal(size=n) * (8 if seed % 2 else 0.5)         # odd seeds: the minimiser is far outside the box, most variables end at a bound
    xc = rng.uniform(-0.6, 0.6, n)                             # the current iterate: the box of the step is [-1 - xc, 1 - xc]
    x_lo, x_hi = -1.0 - xc, 1.0 - xc
    m = int(rng.integers(0, 300))
    A = rng.normal(size=(m, n))
    d_box = np.clip(-g0 / Gd, x_lo, x_hi)
    far = seed % 4 != 3                                        # every fourth case: some rows cut the clipped point off
    hi = A @ d_box + (rng.uniform(0.05, 2.0, m) if far else rng.uniform(-0.3, 1.0, m))
    lo = np.full(m, -1e19)
    x1, ok1, steps1, mult1 = _qp_box(Gd, g0, A, lo, hi, x_lo, x_hi, True)
    x0, ok0, steps0, mult0 = _qp_box(Gd, g0, A, lo, hi, x_lo, x_hi, False)
    assert ok1 == ok0
    if not ok0:
        return
    assert np.abs(x1 - x0).max() <= 1e-9 and abs(mult1 - mult0) <= 1e-7 * (1 + mult0), (x1, x0, mult1, mult0)
    violated = m > 0 and (A @ d_box - hi).max() > 1e-7
    assert (steps1 == 0) == (not violated)
    if not violated:
        assert np.abs(x1 - d_box).max() <= 1e-15 and steps0 >= int(np.sum((d_box == x_lo) | (d_box == x_hi)))   # (the library divides by G through its reciprocal) at least one dual step per clipped variable saved
    Afull = np.vstack([A, np.eye(n)]) if m else np.eye(n)
    _kkt_certificate(Gd, g0, Afull, np.concatenate([lo, x_lo]), np.concatenate([hi, x_hi]), x1)
