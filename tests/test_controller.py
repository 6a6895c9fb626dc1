"""The tracking controller: kinova_controller(Kr, alpha, V_max, r_thr, q, qd, q_des, qd_des, qdd_des, eps).

The reference records no outputs of its controller (parity unpinned), so the CPU restatement (oracle/controller_oracle.cpp)
is pinned by invariants -- an independent formulation of the same dynamics, interval enclosure, the control-barrier
logic -- and the device implementation is then compared with it."""
import numpy as np
import pytest

from test_oracle_invariants import robot_arrays, scalar_rnea

KR, ALPHA, V_MAX, R_THR = 10.0, 1.0, 1e-2, 1e-10   # gains of kinova_src/kinova_simulator_interfaces/uarmtd_robust_CBF_MEX_LLC.m defaults


def _states(seed, count):
    rng = np.random.default_rng(seed)
    q = rng.uniform(-np.pi, np.pi, (count, 7))
    qd = rng.uniform(-1, 1, (count, 7))
    q_des = q + rng.uniform(-0.02, 0.02, (count, 7))
    qd_des = qd + rng.uniform(-0.05, 0.05, (count, 7))
    qdd_des = rng.uniform(-2, 2, (count, 7))
    return q, qd, q_des, qd_des, qdd_des


def test_spatial_passivity_rnea_equals_the_planners_vector_formulation():
    """The controller's Featherstone-style passRNEA (CoM frames, twists) and the planner's 3-vector passivity RNEA
    (RT/Dynamics.cu:83-181, written out in numpy in test_oracle_invariants.scalar_rnea) are two formulations of
    M(q) qdd_a + C(q, qd) qd_a + g(q): their torques must agree for arbitrary (q, qd, qd_a, qdd_a)."""
    from oracle.cpu_oracle import pass_rnea_scaled
    rb = robot_arrays()
    rng = np.random.default_rng(1)
    for _ in range(20):
        q, qd, qda, qdda = rng.uniform(-np.pi, np.pi, 7), rng.uniform(-2, 2, 7), rng.uniform(-2, 2, 7), rng.uniform(-3, 3, 7)
        tau = pass_rnea_scaled(np.zeros(7), np.zeros(7), q, qd, qda, qdda)
        assert np.abs(tau - scalar_rnea(rb, q, qd, qda, qdda)).max() <= 1e-10


def test_interval_rnea_encloses_every_model_in_the_uncertainty_set():
    from oracle.cpu_oracle import pass_rnea_scaled, robust_controller
    eps = 0.03
    q, qd, q_des, qd_des, qdd_des = _states(2, 6)
    rng = np.random.default_rng(3)
    for s in range(6):
        out = robust_controller(KR, ALPHA, V_MAX, R_THR, q[s], qd[s], q_des[s], qd_des[s], qdd_des[s], eps=eps)
        lo, hi = out["tau_interval"][:, 0], out["tau_interval"][:, 1]
        assert out["inside"] and np.all(lo <= out["tau"]) and np.all(out["tau"] <= hi) and np.all(hi - lo > 0)
        # the reference inputs of the RNEA calls (robust_controller.cpp:70-80)
        e = (q_des[s] - q[s] + np.pi) % (2 * np.pi) - np.pi
        qa_d, qa_dd = qd_des[s] + KR * e, qdd_des[s] + KR * (qd_des[s] - qd[s])
        for _ in range(25):
            corner = rng.random() < 0.5
            s_m = eps * (rng.choice([-1.0, 1.0], 7) if corner else rng.uniform(-1, 1, 7))
            s_I = eps * (rng.choice([-1.0, 1.0], 7) if corner else rng.uniform(-1, 1, 7))
            tau = pass_rnea_scaled(s_m, s_I, q[s], qd[s], qa_d, qa_dd)
            assert np.all(tau >= lo - 1e-12) and np.all(tau <= hi + 1e-12)


def test_robust_input_logic():
    """v = 0 when the tracking error r vanishes; otherwise v is anti-parallel to r with the gain of robust_controller.cpp:150-163,
    u = tau - v, and a larger uncertainty never shrinks the gain."""
    from oracle.cpu_oracle import robust_controller
    q, qd, q_des, qd_des, qdd_des = _states(4, 4)
    out0 = robust_controller(KR, ALPHA, V_MAX, R_THR, q[0], qd[0], q[0], qd[0], qdd_des[0])
    assert not out0["v"].any() and np.array_equal(out0["u"], out0["tau"])
    for s in range(4):
        e = (q_des[s] - q[s] + np.pi) % (2 * np.pi) - np.pi
        r = (qd_des[s] - qd[s]) + KR * e
        small = robust_controller(KR, ALPHA, V_MAX, R_THR, q[s], qd[s], q_des[s], qd_des[s], qdd_des[s], eps=0.01)
        big = robust_controller(KR, ALPHA, V_MAX, R_THR, q[s], qd[s], q_des[s], qd_des[s], qdd_des[s], eps=0.10)
        for out in (small, big):
            assert np.abs(out["u"] - (out["tau"] - out["v"])).max() == 0.0
            lam = np.linalg.norm(out["v"])
            assert np.abs(out["v"] + lam * r / np.linalg.norm(r)).max() <= 1e-12 * max(1.0, lam)
        assert np.linalg.norm(big["v"]) >= np.linalg.norm(small["v"]) - 1e-12


@pytest.mark.gpu
def test_device_controller_matches_the_cpu_restatement():
    """One device thread per state against the CPU restatement: same operation order, same outward rounding -> agreement
    to the last bits (1e-12 relative asserted); single-state call = the MEX shape."""
    from armour_amd.controller import kinova_controller
    from oracle.cpu_oracle import robust_controller
    B = 200
    q, qd, q_des, qd_des, qdd_des = _states(7, B)
    q_des[5], qd_des[5] = q[5], qd[5]                         # r = 0: no robust input
    u, tau, v = kinova_controller(KR, ALPHA, V_MAX, R_THR, q, qd, q_des, qd_des, qdd_des, eps=0.03)
    for s in list(range(0, B, 17)) + [5]:
        ref = robust_controller(KR, ALPHA, V_MAX, R_THR, q[s], qd[s], q_des[s], qd_des[s], qdd_des[s], eps=0.03)
        for got, key in ((u[s], "u"), (tau[s], "tau"), (v[s], "v")):
            assert np.abs(got - ref[key]).max() <= 1e-12 * max(1.0, np.abs(ref[key]).max())
    assert not v[5].any()
    u1, tau1, v1 = kinova_controller(KR, ALPHA, V_MAX, R_THR, q[3], qd[3], q_des[3], qd_des[3], qdd_des[3])
    assert np.array_equal(u1, u[3]) and np.array_equal(tau1, tau[3]) and np.array_equal(v1, v[3])


@pytest.mark.gpu
def test_latency_and_throughput_kernels_agree_bit_for_bit():
    """Few states run the three RNEA passes of an update on three waves (controller.hip, the latency kernel), many run one lane per state:
    the same states give the same bits either way, with and without a robust input."""
    from armour_amd.controller import kinova_controller
    B = 64 * 256 + 1000                                       # past the latency kernel's limit
    q, qd, q_des, qd_des, qdd_des = _states(11, B)
    q_des[70], qd_des[70] = q[70], qd[70]                     # r = 0: the third pass is skipped
    bulk = kinova_controller(KR, ALPHA, V_MAX, R_THR, q, qd, q_des, qd_des, qdd_des, eps=0.03)
    for lo, hi in ((0, 1), (64, 130), (16000, 16200)):
        few = kinova_controller(KR, ALPHA, V_MAX, R_THR, q[lo:hi], qd[lo:hi], q_des[lo:hi], qd_des[lo:hi], qdd_des[lo:hi], eps=0.03)
        for a, b in zip(few, bulk):
            assert np.array_equal(np.asarray(a).reshape(hi - lo, 7), b[lo:hi])
    assert not bulk[2][70].any()


@pytest.mark.gpu
def test_controller_kernel_choice_is_an_api_call_and_changes_no_bit():
    """armour_controller_set_kernel (round 4: it was the environment variable ARMOUR_CTL_SPLIT): the same 300 states through the one-lane-per-state
    kernel (0), the four-wave latency kernel (1) and the automatic choice (-1) -- identical outputs; values outside -1..1 are refused."""
    from armour_amd import _lib
    from armour_amd.controller import kinova_controller
    L = _lib.load()
    q, qd, q_des, qd_des, qdd_des = _states(23, 300)
    got = []
    try:
        for which in (0, 1, -1):
            _lib.check(L.armour_controller_set_kernel(which))
            got.append(kinova_controller(KR, ALPHA, V_MAX, R_THR, q, qd, q_des, qd_des, qdd_des, eps=0.03))
        assert L.armour_controller_set_kernel(2) < 0
    finally:
        L.armour_controller_set_kernel(-1)
    for other in got[1:]:
        for a, b in zip(got[0], other):
            assert np.array_equal(a, b)
