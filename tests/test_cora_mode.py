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


# ------------------------------------------------------------------------------------------------ the dynamics half (round 3)
def _bezier_all(q0, qd0, qdd0, ka, s):
    """position, velocity, acceleration of the degree-5 Bezier curve (duration 1)"""
    q = _bezier(q0, qd0, qdd0, ka, s)
    qd = 30 * s**2 * (s - 1)**2 * ka + ((s - 1)**2 * (2 * qd0 + 4 * qd0 * s + 2 * qdd0 * s - 30 * qd0 * s**2 - 5 * qdd0 * s**2)) / 2
    qdd = 60 * s * (2 * s**2 - 3 * s + 1) * ka - (s - 1) * (qdd0 - 36 * qd0 * s - 8 * qdd0 * s + 60 * qd0 * s**2 + 10 * qdd0 * s**2)
    return q, qd, qdd


def _scalar_passivity_rnea(rb, q, qd, qda, qdda, mass_scale=None):
    """PZM/utility/poly_zonotope_rnea.m on numbers (= SIM/dynamics/rnea.m): no armature / damping terms, optional per-link mass and
    inertia scaling (a model inside the uncertainty set)."""
    from armour_amd.robot_geometry import joint_frames, rot_axis
    T0, P, axes, _, _ = joint_frames(rb)
    n = len(axes)
    sc = np.ones(n) if mass_scale is None else np.asarray(mass_scale)
    Pn = np.array(rb.trans)[:3 * (n + 1)].reshape(n + 1, 3)
    R = [T0[i] @ rot_axis(axes[i], q[i]) for i in range(n)] + [np.eye(3)]
    w = np.zeros(3); wd = np.zeros(3); wa = np.zeros(3); la = np.array([0, 0, rb.gravity])
    F, N = [], []
    for i in range(n):
        Rt, z = R[i].T, axes[i]
        la = Rt @ (la + np.cross(wd, Pn[i]) + np.cross(w, np.cross(wa, Pn[i])))
        wd = Rt @ wd + np.cross(Rt @ wa, qd[i] * z) + qdda[i] * z
        w = Rt @ w + qd[i] * z
        wa = Rt @ wa + qda[i] * z
        cm_ = np.array(rb.com[3 * i:3 * i + 3])
        I = np.array(rb.inertia[9 * i:9 * i + 9]).reshape(3, 3) * sc[i]
        F.append(sc[i] * rb.mass[i] * (la + np.cross(wd, cm_) + np.cross(w, np.cross(wa, cm_))))
        N.append(I @ wd + np.cross(wa, I @ w))
    f = np.zeros(3); nn = np.zeros(3); u = np.zeros(n)
    for i in range(n - 1, -1, -1):
        cm_ = np.array(rb.com[3 * i:3 * i + 3])
        nn = N[i] + R[i + 1] @ nn + np.cross(cm_, F[i]) + np.cross(Pn[i + 1], R[i + 1] @ f)
        f = R[i + 1] @ f + F[i]
        u[i] = axes[i] @ nn
    return u


def test_jrs_velocity_and_acceleration_sets_contain_the_curve():
    """create_jrs_online.m:150-178: Qd, Qd_a, Qdd_a and R_t of every interval enclose the curve's velocity / auxiliary velocity /
    auxiliary acceleration / transposed rotation for sampled k, time and tracking errors."""
    from armour_amd.robot_geometry import joint_frames, rot_axis
    from oracle.cpu_oracle import kinova_robot
    rb = kinova_robot()
    axes = joint_frames(rb)[2]
    n_t, ub, kr, krange = 10, 0.0191, 10.0, np.pi / 36
    jrs = cm.create_jrs_online(STATE["q0"], STATE["qd0"], STATE["qdd0"], axes.T, ultimate_bound=ub, k_r=kr, k_range=krange, n_t=n_t, full=True)
    rng = np.random.default_rng(9)
    inside = lambda p, k, v: np.all(np.abs(v - cm.slice_pz(p, k)) <= np.abs(p.Grest).sum(axis=1) + 1e-9)
    for _ in range(8):
        k = rng.uniform(-1, 1, 7)
        for i in range(n_t):
            s = (i + rng.uniform(0, 1)) / n_t
            q, qd, qdd = _bezier_all(STATE["q0"], STATE["qd0"], STATE["qdd0"], krange * k, s)
            ep, ev = rng.uniform(-1, 1, 7) * ub / kr, rng.uniform(-1, 1, 7) * 2 * ub
            for j in range(7):
                assert inside(jrs["Q"][i][j], k, q[j] + ep[j])
                assert inside(jrs["Qd"][i][j], k, qd[j] + ev[j])
                assert inside(jrs["Qd_a"][i][j], k, qd[j] + kr * ep[j])
                assert inside(jrs["Qdd_a"][i][j], k, qdd[j] + kr * ev[j])
                Rt = jrs["R_t"][i][j]
                true = rot_axis(axes[j], q[j] + ep[j]).T
                mid = Rt.C + (Rt.G * np.prod(np.array([k[j]])[:, None] ** Rt.E, axis=0)).sum(axis=2)
                assert np.all(np.abs(true - mid) <= np.abs(Rt.Grest).sum(axis=2) + 1e-9)


def test_cross_is_the_cross_product_of_the_sets():
    rng = np.random.default_rng(12)
    a, b = _rand_pz(rng, 3, [1, 2], 3, 0), _rand_pz(rng, 3, [2, 3], 2, 0)
    v = rng.normal(size=3)
    for _ in range(10):
        x = {i: rng.uniform(-1, 1) for i in range(1, 4)}
        z = np.zeros(0)
        assert np.allclose(_eval(cm.cross(a, b), x, z), np.cross(_eval(a, x, z), _eval(b, x, z)))
        assert np.allclose(_eval(cm.cross(a, v), x, z), np.cross(_eval(a, x, z), v))
        assert np.allclose(_eval(cm.cross(v, b), x, z), np.cross(v, _eval(b, x, z)))
    s = _rand_pz(rng, 1, [1], 2, 1)
    sv = cm.scalar_times_axis(s, [0, 0, 1.0])
    assert sv.dim == 3 and np.allclose(sv.G[2], s.G[0]) and np.all(sv.G[:2] == 0) and np.allclose(sv.Grest[2], s.Grest[0])


def test_pz_rnea_torque_sets_contain_the_true_torques():
    """PZM/utility/poly_zonotope_rnea.m: for sampled k, time inside the interval, tracking errors within the ultimate bound and
    (interval parameters) link masses / inertias within +-3 %, the scalar passivity RNEA torque of every joint lies inside the
    sliced torque set; the nominal sets are tighter than the interval ones; set sizes stay within zono_order."""
    from oracle.cpu_oracle import kinova_robot
    rb = kinova_robot()
    n_t, ub, kr, krange = 20, 0.0191, 10.0, np.pi / 36
    idx = [0, 7, 13, 19]
    nom, itv = cm.inertial_params(rb, False), cm.inertial_params(rb, True)
    jrs = cm.create_jrs_online(STATE["q0"], STATE["qd0"], STATE["qdd0"], nom["axes"].T, ultimate_bound=ub, k_r=kr, k_range=krange, n_t=n_t,
                               full=True, time_indices=set(idx))
    rng = np.random.default_rng(21)
    for i in idx:
        args = (jrs["R"][i], jrs["R_t"][i], jrs["Qd"][i], jrs["Qd_a"][i], jrs["Qdd_a"][i], True)
        u_nom, f_nom, n_nom = cm.poly_zonotope_rnea(*args, nom)
        u_int, _, _ = cm.poly_zonotope_rnea(*args, itv)
        assert max(p.G.shape[1] + p.Grest.shape[1] for p in f_nom + n_nom) <= 3 * 40
        for j in range(7):
            (l0, h0), (l1, h1) = cm.interval(u_nom[j]), cm.interval(u_int[j])
            assert l1[0] <= l0[0] + 1e-9 and h1[0] >= h0[0] - 1e-9          # the interval-parameter set encloses the nominal one
        for _ in range(6):
            k = rng.uniform(-1, 1, 7)
            s = (i + rng.uniform(0, 1)) / n_t
            q, qd, qdd = _bezier_all(STATE["q0"], STATE["qd0"], STATE["qdd0"], krange * k, s)
            ep, ev = rng.uniform(-1, 1, 7) * ub / kr, rng.uniform(-1, 1, 7) * 2 * ub
            qs, qds, qda, qdda = q + ep, qd + ev, qd + kr * ep, qdd + kr * ev
            u = _scalar_passivity_rnea(rb, qs, qds, qda, qdda)
            um = _scalar_passivity_rnea(rb, qs, qds, qda, qdda, mass_scale=1 + 0.03 * rng.uniform(-1, 1, 7))
            for j in range(7):
                mid, rad = cm.slice_pz(u_nom[j], k)[0], np.abs(u_nom[j].Grest).sum()
                assert abs(u[j] - mid) <= rad + 1e-9, (i, j, u[j] - mid, rad)
                mid, rad = cm.slice_pz(u_int[j], k)[0], np.abs(u_int[j].Grest).sum()
                assert abs(um[j] - mid) <= rad + 1e-9, (i, j, um[j] - mid, rad)


def test_input_and_joint_limit_constraints_follow_the_planner():
    """KSI/uarmtd_planner.m:471-559,562-576,622-690: v_norm = alpha V_diff.sup / ultimate_bound + rho_max; u_ub / u_lb bracket the
    true torque by at least v_norm; the pruning keeps exactly the constraints whose interval reaches 0; gradients against central
    differences; a torque limit below the gravity load makes a constraint both `needed` and violated."""
    from oracle.cpu_oracle import kinova_robot
    rb = kinova_robot()
    n_t, ub, kr, krange = 20, 0.0191, 10.0, np.pi / 36
    lim = np.array(rb.torque_limits)
    res = cm.input_constraints(STATE["q0"], STATE["qd0"], STATE["qdd0"], rb, lim, [11], n_t=n_t, ultimate_bound=ub, k_r=kr, k_range=krange)[11]
    assert res["v_norm"] >= res["rho_max"] > 0 and res["V_sup"] > 0
    assert abs(res["v_norm"] - (10.0 * res["V_sup"] / ub + res["rho_max"])) <= 1e-12
    rng = np.random.default_rng(3)
    k = rng.uniform(-1, 1, 7)
    q, qd, qdd = _bezier_all(STATE["q0"], STATE["qd0"], STATE["qdd0"], krange * k, 11.5 / n_t)
    u = _scalar_passivity_rnea(rb, q, qd, qd, qdd)
    for j in range(7):
        ub_v, g_ub = cm.eval_constraint(res["u_ub"][j], k, 7)      # (tau + v_norm + buffer) - u_max
        lb_v, g_lb = cm.eval_constraint(res["u_lb"][j], k, 7)      # -(tau - v_norm - buffer) - u_max
        assert u[j] + res["v_norm"] - lim[j] <= ub_v + 1e-9 and -(u[j] - res["v_norm"]) - lim[j] <= lb_v + 1e-9
        assert not cm.needed(res["u_ub"][j]) and not cm.needed(res["u_lb"][j])          # the real limits are far away: pruned
        for c in range(7):
            kp, km = k.copy(), k.copy()
            kp[c] += 1e-6; km[c] -= 1e-6
            fd = (cm.eval_constraint(res["u_ub"][j], kp, 7)[0] - cm.eval_constraint(res["u_ub"][j], km, 7)[0]) / 2e-6
            assert abs(fd - g_ub[c]) <= 1e-5 * max(1.0, abs(fd)) and abs(g_ub[c] + g_lb[c]) <= 1e-12
    tight = lim.copy(); tight[1] = 5.0                            # joint 2 carries ~15 N m of gravity load here
    res2 = cm.input_constraints(STATE["q0"], STATE["qd0"], STATE["qdd0"], rb, tight, [11], n_t=n_t, ultimate_bound=ub, k_r=kr, k_range=krange)[11]
    hot = [j for j in range(7) if cm.needed(res2["u_ub"][j]) or cm.needed(res2["u_lb"][j])]
    assert hot == [1]
    assert max(cm.eval_constraint(res2["u_ub"][1], k, 7)[0], cm.eval_constraint(res2["u_lb"][1], k, 7)[0]) > 0
    # joint limits: a continuous joint is never constrained; the speed limit of a joint driven at its limit is
    jl = cm.joint_limit_constraints(STATE["q0"], STATE["qd0"], STATE["qdd0"], rb, [0, 11], n_t=n_t, ultimate_bound=ub, k_r=kr, k_range=krange)
    assert not cm.needed(jl[11][0]["q_ub"]) and not cm.needed(jl[11][0]["q_lb"])
    fast = STATE["qd0"].copy(); fast[4] = rb.speed_limits[4] * 0.999
    jl2 = cm.joint_limit_constraints(STATE["q0"], fast, STATE["qdd0"], rb, [0], n_t=n_t, ultimate_bound=ub, k_r=kr, k_range=krange)
    assert cm.needed(jl2[0][4]["dq_ub"]) and not cm.needed(jl2[0][4]["dq_lb"])


def _torque_cross_validation(make_planner):
    """Both reference paths bound the same applied torque.  `make_planner(robot, params)` returns an object with eval_g(k) [m],
    bounds() -> (g_l, g_u) and torque_radius() [n, T] -- the HIP path on the GPU, the CPU oracle of the same C++ path here.  With the
    planner at the MATLAB path's parameters (T = 100, k_range = pi/36) and the CORA-mode JRS at the robot's ultimate bound: for
    sampled (k, t) and tracking errors inside the bound, the scalar passivity-RNEA torque of every joint lies inside the C++ path's
    sliced u_nom +- torque_radius AND inside the CORA-mode bracket [tau - v_norm - buffer, tau + v_norm + buffer]; with the real
    torque limits both paths call every sampled point feasible, with joint 2's limit below its gravity load both infeasible."""
    from oracle.cpu_oracle import default_params, kinova_robot
    T, krange = 100, np.pi / 36
    rb = kinova_robot()
    pr = default_params(T)
    for c in range(7):
        pr.k_range[c] = krange
    eps = np.sqrt(2 * rb.V_m / rb.M_min)                           # RT/KinovaWithoutGripperInfo.h:108-112 = LLC.ultimate_bound
    idx = [3, 38, 71, 96]
    rng = np.random.default_rng(8)
    for limits, expect_feasible in ((None, True), (5.0, False)):
        if limits is not None:
            rb.torque_limits[1] = limits
        pl = make_planner(rb, pr)
        cora = cm.input_constraints(STATE["q0"], STATE["qd0"], STATE["qdd0"], rb, np.array(rb.torque_limits), idx, n_t=T,
                                    alpha_constant=rb.alpha, ultimate_bound=eps, k_r=rb.K, k_range=krange)
        gl, gu = pl.bounds()
        radius = pl.torque_radius()                                # [n, T]
        for _ in range(4):
            k = rng.uniform(-1, 1, 7)
            g = pl.eval_g(k)
            for i in idx:
                s = (i + rng.uniform(0, 1)) / T
                q, qd, qdd = _bezier_all(STATE["q0"], STATE["qd0"], STATE["qdd0"], krange * k, s)
                ep, ev = rng.uniform(-1, 1, 7) * eps / rb.K, rng.uniform(-1, 1, 7) * 2 * eps
                u = _scalar_passivity_rnea(rb, q + ep, qd + ev, qd + rb.K * ep, qdd + rb.K * ev)
                for j in range(7):
                    cpp = g[i * 7 + j]                             # sliced centre of u_nom(j, i) (row t*n + j), incl. armature / damping terms
                    extra = rb.armature[j] * (qdd[j] + rb.K * ev[j]) + rb.damping[j] * (qd[j] + ev[j])   # RT/Dynamics.cu:175-179 adds them, the MATLAB RNEA does not
                    assert abs(u[j] + extra - cpp) <= radius[j, i] + 1e-9, (i, j)
                    ub_v = cm.eval_constraint(cora[i]["u_ub"][j], k, 7)[0] + rb.torque_limits[j]
                    lb_v = -(cm.eval_constraint(cora[i]["u_lb"][j], k, 7)[0] + rb.torque_limits[j])
                    assert lb_v - 1e-9 <= u[j] <= ub_v + 1e-9, (i, j, lb_v, u[j], ub_v)
                cpp_ok = bool(np.all((g[i * 7:i * 7 + 7] >= gl[i * 7:i * 7 + 7]) & (g[i * 7:i * 7 + 7] <= gu[i * 7:i * 7 + 7])))
                cora_ok = all(cm.eval_constraint(cora[i]["u_ub"][j], k, 7)[0] <= 0 and cm.eval_constraint(cora[i]["u_lb"][j], k, 7)[0] <= 0 for j in range(7))
                assert cpp_ok == cora_ok == expect_feasible, (i, cpp_ok, cora_ok)


def test_torque_bounds_cross_validation_of_the_two_reference_paths_on_the_cpu():
    """the C++ path as the CPU oracle restates it against the MATLAB path as cora_mode restates it"""
    from oracle.cpu_oracle import Oracle

    class P:
        def __init__(self, rb, pr):
            self.o = Oracle(robot=rb, params=pr).set_problem(STATE["q0"], STATE["qd0"], STATE["qdd0"], STATE["q0"], np.zeros((0, 12)))
        def eval_g(self, k):
            return self.o.eval_g_jac(k, want_jac=False)[0]
        def bounds(self):
            return self.o.bounds()[2:]
        def torque_radius(self):
            return self.o.torque_radius()
    _torque_cross_validation(P)


@pytest.mark.gpu
def test_torque_bounds_cross_validation_against_the_hip_path():
    """the same with the HIP path (libarmour_hip.so through the C ABI) in place of the oracle"""
    from armour_amd import _lib
    from armour_amd.planner import ArmourNLP
    import ctypes as C

    class P:
        def __init__(self, rb, pr):
            # (the oracle's and the product's robot / parameter structs have the same layout: include/armour_types.h)
            hr, hp = _lib.ArmourRobot(), _lib.ArmourParams()
            C.memmove(C.byref(hr), C.byref(rb), C.sizeof(hr)); C.memmove(C.byref(hp), C.byref(pr), C.sizeof(hp))
            self.nlp = ArmourNLP(robot=hr, params=hp).set_parameters(STATE["q0"], STATE["qd0"], STATE["qdd0"], STATE["q0"], np.zeros((0, 12)))
        def eval_g(self, k):
            return self.nlp.eval_g(k)[0]
        def bounds(self):
            _, _, gl, gu = self.nlp.get_bounds_info()
            return gl[0], gu[0]
        def torque_radius(self):
            return self.nlp.torque_radius()[0]
    _torque_cross_validation(P)
