"""The CPU oracle against the committed golden fixtures (tests/golden/*.npz, made by make_golden.py) and the
row-order / bounds contract of the reference (RT/NLPclass.cu:46-49,87-165)."""
import numpy as np
import pytest

from helpers import PZ_TESTS_K, load_golden

CASES = [f"{n}_T{T}" for n in ("sample", "debug", "scene013") for T in (100, 128)]


def _solve(gd):
    from oracle.cpu_oracle import Oracle
    return Oracle(T=int(gd["T"])).set_problem(gd["q0"], gd["qd0"], gd["qdd0"], gd["q_des"], gd["obstacles"])


@pytest.mark.parametrize("case", CASES)
def test_oracle_reproduces_golden(case):
    gd = load_golden(case)
    o = _solve(gd)
    T = int(gd["T"])
    assert abs(o.min_margin() - float(gd["min_margin"])) <= 1e-6 * float(gd["min_margin"]) + 1e-15
    assert np.abs(o.torque_radius() - gd["torque_radius"]).max() <= 1e-12
    assert np.abs(o.link_generators() - gd["link_gens"]).max() <= 1e-13
    lc, lk, tc, tk = [], [], [], []
    for i in range(7):
        for t in range(T):
            k1 = o.pz("link", i, t)[2]; lc.append(len(k1)); lk.append(k1)
            k2 = o.pz("torque", i, t)[2]; tc.append(len(k2)); tk.append(k2)
    assert np.array_equal(np.array(lc), gd["link_count"]) and np.array_equal(np.concatenate(lk), gd["link_keys"])
    assert np.array_equal(np.array(tc), gd["torque_count"]) and np.array_equal(np.concatenate(tk), gd["torque_keys"])
    _, _, gl, gu = o.bounds()
    assert np.abs(gl - gd["g_l"]).max() <= 1e-12 and np.abs(gu - gd["g_u"]).max() <= 1e-12
    for tag, k in (("k0", np.zeros(7)), ("kt", PZ_TESTS_K)):
        g, jac = o.eval_g_jac(k)
        assert np.abs(g - gd[f"g_{tag}"]).max() <= 1e-12
        assert np.abs(jac[gd["jac_rows"]] - gd[f"jac_{tag}"]).max() <= 1e-12
        assert abs(o.eval_f(k) - float(gd[f"f_{tag}"])) <= 1e-13
        assert np.abs(o.eval_grad_f(k) - gd[f"gradf_{tag}"]).max() <= 1e-13


def test_row_order_and_bounds_contract():
    """m = n*T + J*T*O + 4n; torque bounds = +-(limit - radius); collision (-1e19, 0]; position rows use qe,
    velocity rows use qde (RT/NLPclass.cu:117-164; RT/KinovaWithoutGripperInfo.h:76-112)."""
    gd = load_golden("sample_T100")
    o = _solve(gd)
    T, n, J, O = 100, 7, 7, 10
    assert o.m == n * T + J * T * O + 4 * n
    xl, xu, gl, gu = o.bounds()
    assert np.all(xl == -1) and np.all(xu == 1)
    tr = o.torque_radius()
    lim = np.array([56.7, 56.7, 56.7, 56.7, 29.4, 29.4, 29.4])
    for t in (0, 57, 99):
        assert np.allclose(gu[t * n:(t + 1) * n], lim - tr[:, t], rtol=0, atol=1e-13)
        assert np.allclose(gl[t * n:(t + 1) * n], -lim + tr[:, t], rtol=0, atol=1e-13)
    assert np.all(gl[n * T:n * T + J * T * O] == -1e19) and np.all(gu[n * T:n * T + J * T * O] == 0)
    eps = np.sqrt(2 * 1e-2 / 5.095620491878957)
    assert abs(gu[-4 * n + 1] - (2.41 - eps / 5)) < 1e-14 and abs(gl[-3 * n + 3] - (-2.66 + eps / 5)) < 1e-14
    assert abs(gu[-2 * n] - (1.3963 - 2 * eps)) < 1e-14 and abs(gl[-1] - (-1.2218 + 2 * eps)) < 1e-14
    # collision row order is link-major: g[nT + (l*T + t)*O + o]  (RT/CollisionChecking.cu:127-132)
    k = PZ_TESTS_K
    g, _ = o.eval_g_jac(k)
    A, d, delta = o.hyperplanes()
    cen = o.slice_links(k)
    for (l, t, ob) in ((0, 0, 0), (3, 41, 7), (6, 99, 9)):
        v = A[t, l, ob] @ cen[t, l]
        nz = np.linalg.norm(A[t, l, ob], axis=1) > 0
        expect = -max(np.max((v - d[t, l, ob] - delta[t, l, ob])[nz]), np.max((-v + d[t, l, ob] - delta[t, l, ob])[nz]))
        assert abs(g[n * T + (l * T + t) * O + ob] - expect) <= 1e-14
    # duplicated obstacles (the sample lists each box twice) give identical rows
    blk = g[n * T:n * T + J * T * O].reshape(J, T, O)
    assert np.array_equal(blk[:, :, :5], blk[:, :, 5:])
