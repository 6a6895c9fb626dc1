"""The reference's tracking controller behind its MEX signature, for a batch of states.

    [u, tau, v] = kinova_controller(Kr, alpha, V_max, r_norm_threshold, q, qd, q_des, qd_des, qdd_des[, eps])
    (kinova_src/kinova_simulator_interfaces/kinova_robust_controllers_mex/kinova_controller.cpp:19-84)

All arithmetic happens in libarmour_hip.so (armour_robust_controller, one device thread per state); q .. qdd_des may be
[n] (one state, as the MEX is called from the simulator) or [B, n].
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check
from .planner import kinova_robot


def kinova_controller(Kr, alpha, V_max, r_norm_threshold, q, qd, q_des, qd_des, qdd_des, eps=0.03, robot=None):
    L = _lib.load()
    rb = robot if robot is not None else kinova_robot()
    n = rb.num_factors
    arrs = [np.ascontiguousarray(np.atleast_2d(np.asarray(a, dtype=np.float64))) for a in (q, qd, q_des, qd_des, qdd_des)]
    B = arrs[0].shape[0]
    for a in arrs:
        if a.shape != (B, n):
            raise ValueError(f"expected shape ({B},{n}), got {a.shape}")
    kr = np.ascontiguousarray(np.broadcast_to(np.asarray(Kr, dtype=np.float64).ravel(), (n,)) if np.size(Kr) in (1, n) else np.diag(np.asarray(Kr, dtype=np.float64)))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    u, tau, v = np.zeros((B, n)), np.zeros((B, n)), np.zeros((B, n))
    check(L.armour_robust_controller(C.byref(rb), float(eps), dp(kr), float(alpha), float(V_max), float(r_norm_threshold), B,
                                     *[dp(a) for a in arrs], dp(u), dp(tau), dp(v)))
    single = np.ndim(q) == 1
    return (u[0], tau[0], v[0]) if single else (u, tau, v)
