"""The CORA-semantics mode (armour_amd/cora_mode.py: the reference's MATLAB path restated -- exponent-matrix polynomial
zonotopes, Girard order reduction, Taylor cos / sin with interval remainder, pzfk, polytope_PH constraints).

CPU: every operator against dense sampling of the sets it acts on (a polynomial zonotope is a function of its factors: evaluate
both sides at random factor values), enclosure of cos / sin, soundness of the Girard reduction, and the pipeline's invariant --
the true swept volume of every link lies inside its forward-occupancy set for every k, every time inside the interval and
every tracking error within the ultimate bound (the MATLAB counterpart of RT/debug_script.m:96-123).
GPU: cross-validation of the two reference paths through this repository's two implementations."""
import numpy as np
import pytest

from armour_amd import cora_mode as cm


def _rand_pz(rng, dim, ids, P, Q, maxdeg=2):
    E = rng.integers(0, maxdeg + 1, size=(len(ids), P))
    E[:, E.sum(axis=0) == 0] = 1
    return cm.PZ(rng.normal(size=dim), rng.normal(size=(dim, P)), rng.normal(size=(dim, Q)) * 0.1, E, ids)


def _eval(p, x, beta):
    """point of the set: factors x (dict id -> value in [-1, 1]) for the dependent part, beta in [-1,1]^Q for Grest"""
    xv = np.array([x[i] for i in p.id.tolist()]) if p.id.size else np.zeros(0)
    dep = (p.G * np.prod(xv[:, None] ** p.E, axis=0)).sum(axis=1) if p.G.shape[1] else 0.0
    return p.c + dep + p.Grest @ beta


def test_plus_times_mtimes_are_exact_in_the_dependent_factors():
    rng = np.random.default_rng(0)
    a = _rand_pz(rng, 1, [1, 2], 3, 0)
    b = _rand_pz(rng, 1, [2, 3], 4, 0)
    A = cm.MatPZ(rng.normal(size=(3, 3)), rng.normal(size=(3, 3, 2)), None, [[1, 0], [0, 2]], [1, 4])
    v = _rand_pz(rng, 3, [2, 4], 3, 0)
    B = cm.MatPZ(rng.normal(size=(3, 3)), rng.normal(size=(3, 3, 3)), None, rng.integers(0, 3, size=(2, 3)) + np.eye(2, 3, dtype=int), [3, 4])
    for _ in range(20):
        x = {i: rng.uniform(-1, 1) for i in range(1, 6)}
        z = np.zeros(0)
        assert np.allclose(_eval(a + b, x, z), _eval(a, x, z) + _eval(b, x, z))
        assert np.allclose(_eval(a * b, x, z), _eval(a, x, z) * _eval(b, x, z))
        assert np.allclose(_eval(2.5 * a - 1.0, x, z), 2.5 * _eval(a, x, z) - 1.0)
        Ax = A.C + (A.G * np.prod(np.array([x[i] for i in A.id])[:, None] ** A.E, axis=0)).sum(axis=2)
        Bx = B.C + (B.G * np.prod(np.array([x[i] for i in B.id])[:, None] ** B.E, axis=0)).sum(axis=2)
        assert np.allclose(_eval(A @ v, x, z), Ax @ _eval(v, x, z))
        AB = A @ B
        ABx = AB.C + (AB.G * np.prod(np.array([x[i] for i in AB.id])[:, None] ** AB.E, axis=0)).sum(axis=2)
        assert np.allclose(ABx, Ax @ Bx)
    # equal monomials merge (removeRedundantExponents), the merged set is the same function
    s = a + a
    assert s.G.shape[1] == a.G.shape[1] and np.allclose(s.G, 2 * a.G)


def test_independent_generators_are_enclosed_by_products():
    """with Grest the operators over-approximate: every product of points lies in the interval hull of the product set"""
    rng = np.random.default_rng(1)
    a = _rand_pz(rng, 1, [1], 2, 2)
    b = _rand_pz(rng, 1, [1, 2], 2, 1)
    lo, hi = cm.interval(a * b)
    A = cm.MatPZ(rng.normal(size=(3, 3)), rng.normal(size=(3, 3, 1)), 0.1 * rng.normal(size=(3, 3, 2)), [[1]], [1])
    v = _rand_pz(rng, 3, [2], 2, 2)
    lo3, hi3 = cm.interval(A @ v)
    for _ in range(300):
        x = {1: rng.uniform(-1, 1), 2: rng.uniform(-1, 1)}
        pa, pb = _eval(a, x, rng.uniform(-1, 1, 2)), _eval(b, x, rng.uniform(-1, 1, 1))
        assert lo[0] - 1e-12 <= pa[0] * pb[0] <= hi[0] + 1e-12
        Ax = A.C + A.G[:, :, 0] * x[1] + (A.Grest * rng.uniform(-1, 1, 2)).sum(axis=2)
        w = Ax @ _eval(v, x, rng.uniform(-1, 1, 2))
        assert np.all(lo3 - 1e-12 <= w) and np.all(w <= hi3 + 1e-12)


@pytest.mark.parametrize("order", [1, 2, 6])
def test_cos_sin_enclose_the_function(order):
    rng = np.random.default_rng(2)
    for c0 in (-2.0, 0.3, 1.5, 3.0):
        q = cm.PZ(c0, [[0.2, 0.05]], [[0.01]], [[1, 0], [0, 1]], [1, 2])
        cq, sq = cm.cos(q, order), cm.sin(q, order)
        for _ in range(200):
            x = {1: rng.uniform(-1, 1), 2: rng.uniform(-1, 1)}
            qv = _eval(q, x, rng.uniform(-1, 1, 1))[0]
            for f, val in ((cq, np.cos(qv)), (sq, np.sin(qv))):
                centre = _eval(f, x, np.zeros(1))[0]
                assert abs(val - centre) <= f.Grest.sum() + 0.01 * 1.0 + 1e-12   # |Grest| bounds the remainder AND the unmodelled Grest of q


def test_girard_reduction_is_sound_and_reaches_the_order():
    rng = np.random.default_rng(3)
    p = _rand_pz(rng, 3, [1, 2, 3], 40, 30)
    r = cm.reduce(p, 5)
    assert r.G.shape[1] + r.Grest.shape[1] <= 3 * 5
    lo, hi = cm.interval(r)
    for _ in range(300):
        x = {i: rng.uniform(-1, 1) for i in (1, 2, 3)}
        pt = _eval(p, x, rng.uniform(-1, 1, 30))
        # the kept dependent generators stay exact in x; everything removed is boxed: the point lies in slice(r, x) + box
        kept = _eval(r, x, np.zeros(r.Grest.shape[1]))
        assert np.all(np.abs(pt - kept) <= np.abs(r.Grest).sum(axis=1) + 1e-12)
        assert np.all(lo - 1e-12 <= pt) and np.all(pt <= hi + 1e-12)
    A = cm.MatPZ(rng.normal(size=(3, 3)), rng.normal(size=(3, 3, 30)), rng.normal(size=(3, 3, 20)), rng.integers(1, 3, size=(2, 30)), [1, 2])
    Ar = cm.reduce_mat(A, 4)
    assert Ar.G.shape[2] + Ar.Grest.shape[2] <= 9 * 4
    for _ in range(100):
        x = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1)])
        Ax = A.C + (A.G * np.prod(x[:, None] ** A.E, axis=0)).sum(axis=2) + (A.Grest * rng.uniform(-1, 1, 20)).sum(axis=2)
        xr = np.array([x[i - 1] for i in Ar.id]) if Ar.id.size else np.zeros(0)
        Akept = Ar.C + ((Ar.G * np.prod(xr[:, None] ** Ar.E, axis=0)).sum(axis=2) if Ar.G.shape[2] else 0.0)
        assert np.all(np.abs(Ax - Akept) <= np.abs(Ar.Grest).sum(axis=2) + 1e-12)


def _bezier(q0, qd0, qdd0, ka, s):
    return (s**3 * (6 * s**2 - 15 * s + 10)) * ka + q0 + qd0 * s - 6 * qd0 * s**3 + 8 * qd0 * s**4 - 3 * qd0 * s**5 \
        + qdd0 * s**2 / 2 - 3 * qdd0 * s**3 / 2 + 3 * qdd0 * s**4 / 2 - qdd0 * s**5 / 2


STATE = dict(q0=np.array([0.6, -0.5, 0.4, -1.2, 0.9, -0.7, 0.3]), qd0=np.array([0.4, -0.3, 0.5, -0.4, 0.3, -0.5, 0.2]), qdd0=np.array([0.5, -1, 0.5, 1, -0.5, 1, -1.0]))


def test_forward_occupancy_contains_the_true_swept_volume():
    from armour_amd.robot_geometry import joint_frames, link_frames
    from oracle.cpu_oracle import kinova_robot
    rb = kinova_robot()
    n_t, ub, kr, krange = 20, 0.0191, 10.0, np.pi / 36
    FO = cm.forward_occupancy(STATE["q0"], STATE["qd0"], STATE["qdd0"], rb, n_t=n_t, ultimate_bound=ub, k_r=kr, k_range=krange)
    _, _, _, centers, half = joint_frames(rb)
    rng = np.random.default_rng(4)
    assert max(fo.G.shape[1] + fo.Grest.shape[1] for row in FO for fo in row) <= 3 * 40
    for _ in range(6):
        k = rng.uniform(-1, 1, 7)
        for i in range(0, n_t, 3):
            t = (i + rng.uniform(0, 1)) / n_t
            q = _bezier(STATE["q0"], STATE["qd0"], STATE["qdd0"], krange * k, t) + rng.uniform(-1, 1, 7) * ub / kr
            for j, (Rw, pw) in enumerate(link_frames(rb, q)):
                fo = FO[i][j]
                mid = cm.slice_pz(fo, k)
                box = np.abs(fo.Grest).sum(axis=1)
                for corner in ((-1, -1, -1), (1, 1, 1), (1, -1, 1), (-1, 1, -1), (0, 0, 0)):
                    pt = Rw @ (centers[j] + np.array(corner) * half[j]) + pw
                    assert np.all(np.abs(pt - mid) <= box + 1e-9), (i, j, corner, np.abs(pt - mid) - box)


def test_obstacle_constraints_separate_hit_from_miss():
    """KSI/uarmtd_planner.m:572-609,698-709: an obstacle placed on a link at some (k, t) gives h > 0 there; one far away
    produces no constraint at all (its buffered polytope does not contain the occupancy's centre)."""
    from armour_amd.robot_geometry import link_frames
    from oracle.cpu_oracle import kinova_robot
    rb = kinova_robot()
    n_t, krange = 10, np.pi / 36
    FO = cm.forward_occupancy(STATE["q0"], STATE["qd0"], STATE["qdd0"], rb, n_t=n_t, k_range=krange)
    k = np.array([0.3, -0.2, 0.5, 0.1, -0.4, 0.2, 0.0])
    i, j = 6, 4
    q = _bezier(STATE["q0"], STATE["qd0"], STATE["qdd0"], krange * k, (i + 0.5) / n_t)
    Rw, pw = link_frames(rb, q)[j]
    box = lambda c: np.concatenate([c, [0.02, 0, 0], [0, 0.02, 0], [0, 0, 0.02]])
    cons = cm.obstacle_constraints(FO, np.stack([box(pw), box(np.array([3.0, 3.0, 3.0]))]), 7)
    assert cons and all(c[2] == 0 for c in cons)          # only the obstacle on the arm produces constraints
    hit = [c for c in cons if (c[0], c[1]) == (i, j)]
    assert hit
    h, grad = cm.eval_obstacle_constraint(hit[0], k, 7)
    assert h > 0 and grad.shape == (7,)                    # in collision at this k: h <= 0 is what feasible means
    # gradient against central differences
    for c in range(7):
        kp, km = k.copy(), k.copy()
        kp[c] += 1e-6; km[c] -= 1e-6
        fd = (cm.eval_obstacle_constraint(hit[0], kp, 7)[0] - cm.eval_obstacle_constraint(hit[0], km, 7)[0]) / 2e-6
        assert abs(fd - grad[c]) <= 1e-5 * max(1.0, abs(fd))


@pytest.mark.gpu
def test_cross_validation_against_the_hip_path():
    """The two reference paths agree only as over-approximations of the same sets.  With the HIP planner set to the MATLAB
    path's parameters (T = n_t = 100, k_range = pi/36), for sampled k and times: the true link-frame origin lies inside BOTH
    sliced reach sets; an obstacle sitting on the arm is infeasible in both (h > 0 / g > 0), a distant one feasible in both."""
    from armour_amd.planner import ArmourNLP, default_params, kinova_robot
    from armour_amd.robot_geometry import joint_frames, link_frames
    T, krange = 100, np.pi / 36
    rb = kinova_robot()
    pr = default_params(T)
    for c in range(7):
        pr.k_range[c] = krange
    k = np.array([0.3, -0.2, 0.5, 0.1, -0.4, 0.2, 0.0])
    i, j = 60, 4
    q = _bezier(STATE["q0"], STATE["qd0"], STATE["qdd0"], krange * k, (i + 0.5) / T)
    Rw, pw = link_frames(rb, q)[j]
    box = lambda c: np.concatenate([c, [0.02, 0, 0], [0, 0.02, 0], [0, 0, 0.02]])
    obs = np.stack([box(pw), box(np.array([3.0, 3.0, 3.0]))])
    nlp = ArmourNLP(robot=rb, params=pr).set_parameters(STATE["q0"], STATE["qd0"], STATE["qdd0"], STATE["q0"], obs)
    g, _ = nlp.eval_g_jac(k)
    col = g[0, 7 * T:7 * T + 7 * T * 2].reshape(7, T, 2)             # g[nT + (l*T + t)*O + o]
    assert col[j, i, 0] > 0 and (col[:, :, 1] < 0).all()              # HIP path: hit / miss
    FO = cm.forward_occupancy(STATE["q0"], STATE["qd0"], STATE["qdd0"], rb, n_t=T, k_range=krange)
    cons = cm.obstacle_constraints([FO[i]], obs, 7)                   # the MATLAB path at that time interval
    hit = [c for c in cons if c[1] == j and c[2] == 0]
    assert hit and cm.eval_obstacle_constraint(hit[0], k, 7)[0] > 0 and not [c for c in cons if c[2] == 1]
    # containment of the truth in both paths' sliced link sets
    cen = nlp.link_centers(k)[0]                                      # [T, J, 3] sliced centres of the HIP path
    gens = nlp.link_generators()[0]                                   # [T, J, 3, 6]
    _, _, _, centers, _ = joint_frames(rb)
    rng = np.random.default_rng(5)
    for ti in range(0, T, 7):
        qt = _bezier(STATE["q0"], STATE["qd0"], STATE["qdd0"], krange * k, (ti + rng.uniform(0, 1)) / T)
        for l, (Rl, pl) in enumerate(link_frames(rb, qt)):
            pt = Rl @ centers[l] + pl
            assert np.all(np.abs(pt - cen[ti, l]) <= np.abs(gens[ti, l]).sum(axis=1) + 1e-9)
            fo = FO[ti][l]
            assert np.all(np.abs(pt - cm.slice_pz(fo, k)) <= np.abs(fo.Grest).sum(axis=1) + 1e-9)
    nlp.close()
