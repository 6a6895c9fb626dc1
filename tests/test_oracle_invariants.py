"""Invariants that pin the CPU oracle in the absence of reference golden vectors:
  * the reference's own sanity check (RT/debug_script.m:96-123): scalar passivity-RNEA torques and forward
    kinematics at a random time inside every interval lie inside the sliced reach sets;
  * IPOPT-style first-order derivative test of eval_jac_g (RT/armour_main.cu:255-259: perturbation 1e-8, tol 1e-6);
  * closed-form joint-limit rows against dense sampling of the Bezier curve;
  * cost gradient against finite differences."""
import numpy as np
import pytest

from helpers import DEBUG_STATE, PZ_TESTS_K


def rot_rpy(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cp * cy, -cp * sy, sp],
                     [cr * sy + cy * sp * sr, cr * cy - sp * sr * sy, -cp * sr],
                     [sr * sy - cr * cy * sp, cy * sr + cr * sp * sy, cp * cr]])


def rot_z(q):
    c, s = np.cos(q), np.sin(q)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])


def robot_arrays():
    from oracle.cpu_oracle import kinova_robot
    r = kinova_robot()
    tr = np.array(r.trans)[:24].reshape(8, 3)
    rots = np.array(r.rots)[:21].reshape(7, 3)
    return dict(trans=tr, rots=rots, mass=np.array(r.mass)[:7], com=np.array(r.com)[:21].reshape(7, 3),
                inertia=np.array(r.inertia)[:63].reshape(7, 3, 3), armature=np.array(r.armature)[:7], gravity=r.gravity,
                zc=np.array(r.link_zonotope_center)[:21].reshape(7, 3))


def bezier(q0, qd0, qdd0, k_actual, s):
    """position / velocity / acceleration of the degree-5 Bezier trajectory (RT/Trajectory.h:10-31), duration 1."""
    q = (s**3 * (6 * s**2 - 15 * s + 10)) * k_actual + q0 + qd0 * s - 6 * qd0 * s**3 + 8 * qd0 * s**4 - 3 * qd0 * s**5 \
        + qdd0 * s**2 / 2 - 3 * qdd0 * s**3 / 2 + 3 * qdd0 * s**4 / 2 - qdd0 * s**5 / 2
    qd = 30 * s**2 * (s - 1)**2 * k_actual + ((s - 1)**2 * (2 * qd0 + 4 * qd0 * s + 2 * qdd0 * s - 30 * qd0 * s**2 - 5 * qdd0 * s**2)) / 2
    qdd = 60 * s * (2 * s**2 - 3 * s + 1) * k_actual - (s - 1) * (qdd0 - 36 * qd0 * s - 8 * qdd0 * s + 60 * qd0 * s**2 + 10 * qdd0 * s**2)
    return q, qd, qdd


def scalar_rnea(rb, q, qd, qda, qdda):
    """Passivity RNEA with auxiliary velocity (structure of SIM/dynamics/rnea.m / RT/Dynamics.cu:83-181), scalars."""
    J = 7
    z = np.array([0, 0, 1.0])
    R = [rot_rpy(*rb["rots"][i]) @ rot_z(q[i]) for i in range(J)] + [np.eye(3)]
    w = np.zeros(3); wdot = np.zeros(3); waux = np.zeros(3); lacc = np.array([0, 0, rb["gravity"]])
    F, N = [], []
    for i in range(J):
        Rt = R[i].T
        tr, cm = rb["trans"][i], rb["com"][i]
        lacc = Rt @ (lacc + np.cross(wdot, tr) + np.cross(w, np.cross(waux, tr)))
        w = Rt @ w + qd[i] * z
        waux = Rt @ waux
        wdot = Rt @ wdot + np.cross(waux, qd[i] * z) + qdda[i] * z
        waux = waux + qda[i] * z
        F.append(rb["mass"][i] * (lacc + np.cross(wdot, cm) + np.cross(w, np.cross(waux, cm))))
        N.append(rb["inertia"][i] @ wdot + np.cross(waux, rb["inertia"][i] @ w))
    f = np.zeros(3); n = np.zeros(3); u = np.zeros(J)
    for i in range(J - 1, -1, -1):
        n = N[i] + R[i + 1] @ n + np.cross(rb["com"][i], F[i]) + np.cross(rb["trans"][i + 1], R[i + 1] @ f)
        f = R[i + 1] @ f + F[i]
        u[i] = n[2] + rb["armature"][i] * qdda[i]
    return u


def link_center_fk(rb, q):
    Rw = np.eye(3); pw = np.zeros(3); out = []
    for i in range(7):
        pw = pw + Rw @ rb["trans"][i]
        Rw = Rw @ (rot_rpy(*rb["rots"][i]) @ rot_z(q[i]))
        out.append(Rw @ rb["zc"][i] + pw)
    return np.array(out)


@pytest.fixture(scope="module")
def debug_oracle():
    from oracle.cpu_oracle import Oracle
    s = DEBUG_STATE
    return Oracle(T=128).set_problem(s["q0"], s["qd0"], s["qdd0"], s["q0"], np.zeros((0, 12)))


def test_scalar_rnea_and_fk_inside_reach_sets(debug_oracle):
    """RT/debug_script.m:96-123 made into an assertion: with zero tracking error the nominal torque at any time
    of interval t lies in slice(u_nom)(k) = centre +- independent radius, and the link-box centre lies in the
    interval hull of the sliced link zonotope."""
    o = debug_oracle
    rb = robot_arrays()
    s = DEBUG_STATE
    k = PZ_TESTS_K
    ka = k * np.pi / 48
    tq_c = o.slice_torque(k)
    ln_c = o.slice_links(k)
    gens = o.link_generators()
    rng = np.random.default_rng(0)
    worst_t, worst_l = -np.inf, -np.inf
    for t in range(0, 128):
        tt = (t + rng.uniform(0.02, 0.98)) / 128
        q, qd, qdd = bezier(s["q0"], s["qd0"], s["qdd0"], ka, tt)
        u = scalar_rnea(rb, q, qd, qd, qdd)
        rad = np.array([o.pz("torque", j, t)[1][0] for j in range(7)])
        worst_t = max(worst_t, np.max(np.abs(u - tq_c[t]) - rad))
        assert np.all(np.abs(u - tq_c[t]) <= rad + 1e-9), (t, u - tq_c[t], rad)
        p = link_center_fk(rb, q)
        hull = np.abs(gens[t]).sum(axis=2)
        worst_l = max(worst_l, np.max(np.abs(p - ln_c[t]) - hull))
        assert np.all(np.abs(p - ln_c[t]) <= hull + 1e-9)
    assert worst_t < 0 and worst_l < 0


def test_torque_radius_bounds_disturbance(debug_oracle):
    """radius >= ultimate-bound term + half the disturbance hull (RT/armour_main.cu:172-205)."""
    o = debug_oracle
    tr = o.torque_radius()
    eps = np.sqrt(2 * 1e-2 / 5.095620491878957)
    base = 10.0 * (15.79635774 - 5.095620491878957) * eps
    assert np.all(tr > base)
    for t in (0, 64, 127):
        for j in range(7):
            c, ind, keys, co = o.pz("disturbance", j, t)
            hull = abs(c[0]) + ind[0] + np.abs(co).sum()
            assert tr[j, t] >= base + 0.5 * hull - 1e-12


def test_ipopt_style_derivative_check():
    from armour_amd.worlds import random_problem
    from oracle.cpu_oracle import Oracle
    p = random_problem(5, 6)
    o = Oracle(T=100).set_problem(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    k = np.array([0.31, -0.42, 0.13, 0.77, -0.58, 0.21, -0.09])
    g0, jac = o.eval_g_jac(k)
    h = 1e-7
    T, n = 100, 7
    fd = np.zeros_like(jac)
    for j in range(n):
        kp, km = k.copy(), k.copy()
        kp[j] += h; km[j] -= h
        fd[:, j] = (o.eval_g_jac(kp, want_jac=False)[0] - o.eval_g_jac(km, want_jac=False)[0]) / (2 * h)
    err = np.abs(fd - jac) / np.maximum(1.0, np.abs(jac))
    assert err[:n * T].max() < 1e-6                      # torque rows: smooth polynomials
    col = err[n * T:-4 * n]
    assert np.mean(col.max(axis=1) < 1e-6) > 0.995       # collision rows: max of planes, kinks are measure-zero
    lim = err[-4 * n:]
    # velocity rows whose extremum sits at t = 1 carry the reference's slope-1.0 quirk (RT/Trajectory.cu:503,521)
    ok = lim.max(axis=1) < 1e-6
    quirk = np.isclose(np.abs(jac[-4 * n:]).max(axis=1), (np.pi / 48), rtol=1e-12)
    assert np.all(ok | quirk)


def test_joint_limit_rows_match_dense_sampling():
    from armour_amd.worlds import random_problem
    from oracle.cpu_oracle import Oracle
    for seed in (1, 2, 3):
        p = random_problem(seed, 0)
        o = Oracle(T=100).set_problem(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
        k = np.random.default_rng(seed).uniform(-1, 1, 7)
        g, _ = o.eval_g_jac(k, want_jac=False)
        s = np.linspace(0, 1, 200001)[:, None]
        q, qd, _ = bezier(p["q0"], p["qd0"], p["qdd0"], k * np.pi / 48, s)
        assert np.abs(g[-28:-21] - q.min(axis=0)).max() < 1e-8 and np.abs(g[-21:-14] - q.max(axis=0)).max() < 1e-8
        assert np.abs(g[-14:-7] - qd.min(axis=0)).max() < 1e-8 and np.abs(g[-7:] - qd.max(axis=0)).max() < 1e-8


def test_cost_gradient():
    from oracle.cpu_oracle import Oracle
    from helpers import SAMPLE_PROBLEM as p
    o = Oracle(T=100).set_problem(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    k = PZ_TESTS_K
    gf = o.eval_grad_f(k)
    for j in range(7):
        kp, km = k.copy(), k.copy()
        kp[j] += 1e-6; km[j] -= 1e-6
        assert abs((o.eval_f(kp) - o.eval_f(km)) / 2e-6 - gf[j]) < 1e-7
    # q(t_plan = 0.5) = q0 + k_range*k/2 for zero initial velocity/acceleration -> closed-form cost
    q_plan = p["q0"] + 0.5 * k * np.pi / 48
    e = p["q_des"] - q_plan
    assert abs(o.eval_f(k) - 10.0 * np.sum(e**2)) < 1e-12


# ---------------------------------------------------------------------------------------------------------------------
# The reference's only mixed-axis chain: Fetch (CMP/FetchInfo.h:16, axes {z,y,x,y,x,y,x,fixed,fixed}).  Same containment
# check as RT/debug_script.m:96-123 with rotations about x / y / z: pins the oracle's handling of axes 1 and 2 (JRS
# rotation matrices RT/PZsparse.cu:211-250, the joint-axis entries of RNEA RT/Dynamics.cu:111-131,166) to rigid-body
# physics before the device is compared with it.
def _rot_axis(axis, q):
    c, s = np.cos(q), np.sin(q)
    if axis == 1:
        return np.array([[1.0, 0, 0], [0, c, -s], [0, s, c]])
    if axis == 2:
        return np.array([[c, 0, s], [0, 1.0, 0], [-s, 0, c]])
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])


def _general_rnea_and_fk(r, q, qd, qda, qdda):
    J, n = r.num_joints, r.num_factors
    axes = list(r.axes)[:J]
    tr = np.array(r.trans)[:3 * (J + 1)].reshape(J + 1, 3)
    rots = np.array(r.rots)[:3 * J].reshape(J, 3)
    com = np.array(r.com)[:3 * J].reshape(J, 3)
    inertia = np.array(r.inertia)[:9 * J].reshape(J, 3, 3)
    mass, arm, damp = np.array(r.mass)[:J], np.array(r.armature)[:J], np.array(r.damping)[:J]
    zc = np.array(r.link_zonotope_center)[:3 * J].reshape(J, 3)
    R = [rot_rpy(*rots[i]) @ (_rot_axis(axes[i], q[i]) if axes[i] else np.eye(3)) for i in range(J)] + [np.eye(3)]
    w = np.zeros(3); wdot = np.zeros(3); waux = np.zeros(3); lacc = np.array([0, 0, r.gravity])
    F, N = [], []
    for i in range(J):
        Rt = R[i].T
        z = np.zeros(3)
        if axes[i]:
            z[abs(axes[i]) - 1] = 1.0
        v, va, a = (qd[i], qda[i], qdda[i]) if axes[i] else (0.0, 0.0, 0.0)
        lacc = Rt @ (lacc + np.cross(wdot, tr[i]) + np.cross(w, np.cross(waux, tr[i])))
        w = Rt @ w + v * z
        waux = Rt @ waux
        wdot = Rt @ wdot + np.cross(waux, v * z) + a * z
        waux = waux + va * z
        F.append(mass[i] * (lacc + np.cross(wdot, com[i]) + np.cross(w, np.cross(waux, com[i]))))
        N.append(inertia[i] @ wdot + np.cross(waux, inertia[i] @ w))
    f = np.zeros(3); nn = np.zeros(3); u = np.zeros(n)
    for i in range(J - 1, -1, -1):
        nn = N[i] + R[i + 1] @ nn + np.cross(com[i], F[i]) + np.cross(tr[i + 1], R[i + 1] @ f)
        f = R[i + 1] @ f + F[i]
        if axes[i]:
            u[i] = nn[abs(axes[i]) - 1] + arm[i] * qdda[i] + damp[i] * qd[i]
    Rw = np.eye(3); pw = np.zeros(3); links = []
    for i in range(J):
        pw = pw + Rw @ tr[i]
        Rw = Rw @ R[i]
        links.append(Rw @ zc[i] + pw)
    return u, np.array(links)


def test_fetch_mixed_axes_inside_reach_sets():
    from oracle.cpu_oracle import Oracle, default_params, fetch_robot
    r = fetch_robot()
    T = 32
    q0 = np.array([0.4, -0.5, 0.8, 1.1, -0.7, 0.9, 0.3]); qd0 = np.array([0.5, -0.4, 0.6, -0.5, 0.7, -0.6, 0.4]); qdd0 = np.array([1.0, -1, 0.5, 1, -0.5, 1, -1])
    o = Oracle(robot=r, params=default_params(T)).set_problem(q0, qd0, qdd0, q0, np.zeros((0, 12)))
    assert (o.J, o.n) == (9, 7)
    rng = np.random.default_rng(3)
    for k in (np.zeros(7), PZ_TESTS_K, rng.uniform(-1, 1, 7)):
        ka = k * np.pi / 48
        tq_c, ln_c, gens = o.slice_torque(k), o.slice_links(k), o.link_generators()
        for t in range(T):
            tt = (t + rng.uniform(0.02, 0.98)) / T
            q, qd, qdd = bezier(q0, qd0, qdd0, ka, tt)
            u, links = _general_rnea_and_fk(r, q, qd, qd, qdd)
            rad = np.array([o.pz("torque", j, t)[1][0] for j in range(7)])
            assert np.all(np.abs(u - tq_c[t]) <= rad + 1e-9), (t, u - tq_c[t], rad)
            hull = np.abs(gens[t]).sum(axis=2)
            assert np.all(np.abs(links - ln_c[t]) <= hull + 1e-9), t
    # and the joint axes matter: with all-z axes the same state gives different link positions
    rz = fetch_robot()
    for i in range(7):
        rz.axes[i] = 3
    oz = Oracle(robot=rz, params=default_params(T)).set_problem(q0, qd0, qdd0, q0, np.zeros((0, 12)))
    assert np.abs(oz.slice_links(np.zeros(7)) - o.slice_links(np.zeros(7))).max() > 1e-2
