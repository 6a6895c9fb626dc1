"""Shared test helpers: build device-table inputs from the CPU oracle, PZ_tests slice point, etc."""
import numpy as np

# slice point of the reference's only numeric sanity program (RT/PZ_tests.cu:198)
PZ_TESTS_K = np.array([0.5, 0.6, 0.7, 0.0, -0.5, -0.6, -0.7])
# initial state of RT/debug_script.m:29-31
DEBUG_STATE = dict(q0=np.array([-1.0, -1, -1, -1, 1, 1, 1]), qd0=np.array([1.0, 1, 1, -1, -1, -1, -1]), qdd0=np.full(7, 2.0))


_FIVE = np.array([
    [-0.28239, -0.33281, 0.88069, 0.069825, 0, 0, 0, 0.09508, 0, 0, 0, 0.016624],
    [-0.19033, 0.035391, 1.3032, 0.11024, 0, 0, 0, 0.025188, 0, 0, 0, 0.014342],
    [0.67593, -0.085841, 0.43572, 0.17408, 0, 0, 0, 0.07951, 0, 0, 0, 0.18012],
    [0.75382, 0.51895, 0.4731, 0.030969, 0, 0, 0, 0.22312, 0, 0, 0, 0.22981],
    [0.75382, 0.51895, 0.4731, 0.030969, 0, 0, 0, 0.22312, 0, 0, 0, 0.22981]])
# the reference's commented sample input (RT/armour_main.cu:18-33): a known input without a known answer
SAMPLE_PROBLEM = dict(q0=np.array([0.6543, -0.0876, -0.4837, -1.2278, -1.5735, -1.0720, 0]), qd0=np.zeros(7), qdd0=np.zeros(7),
                      q_des=np.array([0.6831, 0.009488, -0.2471, -0.9777, -1.414, -0.9958, 0]), obstacles=np.vstack([_FIVE, _FIVE]))


def load_golden(name):
    import os
    return dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz")))


def oracle_tables(oracles, cap_l=32, cap_t=128):
    """Pack the reach-set tables of a list of solved oracle problems in the layout armour_debug_load_tables takes."""
    B = len(oracles)
    o0 = oracles[0]
    T, J, n, O = o0.T, o0.J, o0.n, o0.O
    t = dict(
        link_count=np.zeros((B, J, T), np.int32), link_center=np.zeros((B, J, T, 2, 3)),
        link_keys=np.zeros((B, J, T, cap_l), np.uint64), link_coeffs=np.zeros((B, J, T, cap_l, 3)),
        torque_count=np.zeros((B, n, T), np.int32), torque_center=np.zeros((B, n, T, 2)),
        torque_keys=np.zeros((B, n, T, cap_t), np.uint64), torque_coeffs=np.zeros((B, n, T, cap_t)),
        A=np.zeros((B, T, J, O, 36, 3)), d=np.zeros((B, T, J, O, 36)), delta=np.zeros((B, T, J, O, 36)),
        torque_radius=np.zeros((B, n, T)))
    for b, o in enumerate(oracles):
        for l in range(J):
            for tt in range(T):
                c, ind, keys, co = o.pz("link", l, tt)
                M = len(keys)
                t["link_count"][b, l, tt] = M
                t["link_center"][b, l, tt, 0] = c
                t["link_center"][b, l, tt, 1] = ind
                t["link_keys"][b, l, tt, :M] = keys
                t["link_coeffs"][b, l, tt, :M] = co
        for j in range(n):
            for tt in range(T):
                c, ind, keys, co = o.pz("torque", j, tt)
                M = len(keys)
                t["torque_count"][b, j, tt] = M
                t["torque_center"][b, j, tt] = [c[0], ind[0]]
                t["torque_keys"][b, j, tt, :M] = keys
                t["torque_coeffs"][b, j, tt, :M] = co[:, 0]
        if O > 0:
            A, d, delta = o.hyperplanes()
            t["A"][b], t["d"][b], t["delta"][b] = A, d, delta
        t["torque_radius"][b] = o.torque_radius()
    return t


def plane_pairs():
    """(a, b) generator indices of the 36 half-space pairs, RT/CollisionChecking.cu:26-39."""
    out, a, b = [], 0, 1
    for _ in range(36):
        out.append((a, b))
        if b < 8:
            b += 1
        else:
            a += 1
            b = a + 1
    return out


def noise_planes(obstacle, link_gens, rel=1e-9):
    """bool[36]: planes whose normal is rounding noise.  The reference normalises cross(g_a, g_b) whenever its norm is > 0
    (RT/CollisionChecking.cu:177-189); for two generators that are parallel in exact arithmetic the computed cross product is
    ~1e-20 instead of 0 and its DIRECTION is decided by the last bits of the generators -- any two correct implementations
    differ there (found on the reference's hard scenario 3, where the arm starts at multiples of pi/4).  obstacle: [12]
    column-major [c g1 g2 g3]; link_gens: [3][6]."""
    G = np.concatenate([np.asarray(obstacle, dtype=np.float64).reshape(4, 3)[1:], np.asarray(link_gens, dtype=np.float64).T])
    out = np.zeros(36, bool)
    for p, (a, b) in enumerate(plane_pairs()):
        c = np.cross(G[a], G[b])
        nn = np.linalg.norm(G[a]) * np.linalg.norm(G[b])
        out[p] = 0 < np.linalg.norm(c) <= rel * nn
    return out
