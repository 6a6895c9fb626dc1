"""ctypes wrapper of oracle/liboracle.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The wrapped library is the CPU restatement of the reference path (see armour_oracle.cpp header).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ARMOUR_KEY128=1 in the environment selects the second ABI of include/armour_types.h for the whole process -- 128-bit monomial keys, room for
# 8 factors: liboracle_k128.so here, libarmour_hip_k128.so in armour_amd/_lib.py (tests/test_key128.py runs in a process of its own)
KEY128 = os.environ.get("ARMOUR_KEY128", "0") not in ("", "0")
_LIB_PATH = os.path.join(_HERE, "liboracle_k128.so" if KEY128 else "liboracle.so")

MAXJ = 9                    # ARMOUR_MAX_JOINTS
MAXF = 8 if KEY128 else 7   # ARMOUR_MAX_FACTORS


class ArmourRobot(C.Structure):
    _fields_ = [
        ("num_joints", C.c_int32), ("num_factors", C.c_int32),
        ("axes", C.c_int32 * MAXJ), ("continuous", C.c_int32 * MAXF),
        ("trans", C.c_double * ((MAXJ + 1) * 3)), ("rots", C.c_double * (MAXJ * 3)),
        ("mass", C.c_double * MAXJ), ("mass_uncertainty", C.c_double),
        ("com", C.c_double * (MAXJ * 3)),
        ("inertia", C.c_double * (MAXJ * 9)), ("inertia_uncertainty", C.c_double),
        ("friction", C.c_double * MAXJ), ("damping", C.c_double * MAXJ), ("armature", C.c_double * MAXJ),
        ("state_limits_lb", C.c_double * MAXF), ("state_limits_ub", C.c_double * MAXF),
        ("speed_limits", C.c_double * MAXF), ("torque_limits", C.c_double * MAXF),
        ("gravity", C.c_double),
        ("link_zonotope_center", C.c_double * (MAXJ * 3)), ("link_zonotope_generators", C.c_double * (MAXJ * 3)),
        ("alpha", C.c_double), ("V_m", C.c_double), ("M_max", C.c_double), ("M_min", C.c_double), ("K", C.c_double),
        ("mass_uncertainty_link", C.c_double * MAXJ), ("inertia_uncertainty_link", C.c_double * MAXJ),
    ]


class ArmourParams(C.Structure):
    _fields_ = [
        ("num_time_steps", C.c_int32), ("input_constraints_off", C.c_int32),
        ("duration", C.c_double), ("k_range", C.c_double * MAXF),
        ("simplify_threshold", C.c_double), ("t_plan", C.c_double), ("cost_scale", C.c_double),
        ("collision_violation_threshold", C.c_double), ("torque_violation_threshold", C.c_double),
    ]


def build(force=False):
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.POINTER(ArmourRobot), C.POINTER(ArmourParams)]
        L.oracle_destroy.argtypes = [C.c_void_p]
        L.oracle_set_problem.argtypes = [C.c_void_p, dp, dp, dp, dp, C.c_int, dp, C.c_int]
        L.oracle_set_problem_armtd.argtypes = [C.c_void_p, dp, dp, dp, dp, dp, C.c_int, dp, C.c_int]
        L.oracle_num_constraints.argtypes = [C.c_void_p]
        L.oracle_build_ms.argtypes = [C.c_void_p]
        L.oracle_build_ms.restype = C.c_double
        L.oracle_eval_g_jac.argtypes = [C.c_void_p, dp, dp, dp, C.c_int]
        L.oracle_time_eval.argtypes = [C.c_void_p, dp, C.c_int, C.c_int, dp, dp, C.c_int]
        L.oracle_time_eval.restype = C.c_double
        L.oracle_get_bounds.argtypes = [C.c_void_p, dp, dp, dp, dp]
        L.oracle_eval_f.argtypes = [C.c_void_p, dp]
        L.oracle_eval_f.restype = C.c_double
        L.oracle_eval_grad_f.argtypes = [C.c_void_p, dp, dp]
        L.oracle_get_torque_radius.argtypes = [C.c_void_p, dp]
        L.oracle_get_link_generators.argtypes = [C.c_void_p, dp]
        L.oracle_get_hyperplanes.argtypes = [C.c_void_p, dp, dp, dp]
        L.oracle_pz_size.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.oracle_pz_get.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, dp, dp, C.POINTER(C.c_uint64), dp]
        L.oracle_table_sizes.argtypes = [C.c_void_p] + [C.POINTER(C.c_int64)] * 4
        L.oracle_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.oracle_slice_torque.argtypes = [C.c_void_p, dp, dp]
        L.oracle_slice_links.argtypes = [C.c_void_p, dp, dp]
        L.oracle_max_threads.restype = C.c_int
        L.oracle_pz_op.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(dp),
                                   dp, dp, dp, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_uint64), dp, dp]
        L.oracle_min_margin.argtypes = [C.c_void_p]
        L.oracle_min_margin.restype = C.c_double
        assert L.oracle_abi_max_factors() == MAXF, "oracle library and struct mirrors disagree on ARMOUR_MAX_FACTORS"
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def kinova_robot():
    """RT/KinovaWithoutGripperInfo.h constants from the oracle's own generated table (oracle/robot_tables.hpp)."""
    r = ArmourRobot()
    lib().oracle_fill_kinova(C.byref(r))
    return r


def kinova_gripper_robot():
    """RT/KinovaInfo.h: the arm with the 1.72 kg gripper as a fixed 8th joint."""
    r = ArmourRobot()
    lib().oracle_fill_kinova_gripper(C.byref(r))
    return r


def fetch_robot(payload_mass_uncertainty=0.0):
    """CMP/FetchInfo.h (oracle/robot_tables.hpp); payload_mass_uncertainty widens the last link's mass / inertia interval."""
    r = ArmourRobot()
    lib().oracle_fill_fetch(C.byref(r))
    if payload_mass_uncertainty:
        r.mass_uncertainty_link[r.num_joints - 1] = payload_mass_uncertainty
        r.inertia_uncertainty_link[r.num_joints - 1] = payload_mass_uncertainty
    return r


def fetch8_robot(payload_mass_uncertainty=0.0):
    """The oracle's own derivation of the "Fetch 8-DOF" preset (armour_oracle.cpp: oracle_fill_fetch8; 128-bit-key build only)."""
    r = ArmourRobot()
    lib().oracle_fill_fetch8(C.byref(r))
    if payload_mass_uncertainty:
        r.mass_uncertainty_link[r.num_joints - 1] = payload_mass_uncertainty
        r.inertia_uncertainty_link[r.num_joints - 1] = payload_mass_uncertainty
    return r


def default_params(T=128):
    p = ArmourParams()
    lib().oracle_fill_default_params(C.byref(p), T)
    return p


class Oracle:
    """One planning problem on the CPU restatement: set_problem (P1) then eval_g_jac (P2)."""

    def __init__(self, robot=None, params=None, T=128):
        self.robot = robot if robot is not None else kinova_robot()
        self.params = params if params is not None else default_params(T)
        self.T = self.params.num_time_steps
        self.J = self.robot.num_joints
        self.n = self.robot.num_factors
        self.h = lib().oracle_create(C.byref(self.robot), C.byref(self.params))
        self.O = 0

    def __del__(self):
        if getattr(self, "h", None):
            lib().oracle_destroy(self.h)
            self.h = None

    def set_problem(self, q0, qd0, qdd0, q_des, obstacles, threads=0):
        q0, qd0, qdd0, q_des = [np.ascontiguousarray(a, dtype=np.float64) for a in (q0, qd0, qdd0, q_des)]
        obs = np.ascontiguousarray(obstacles, dtype=np.float64).reshape(-1, 12)
        self.O = obs.shape[0]
        if self.O == 0:
            obs = np.zeros((1, 12))
        rc = lib().oracle_set_problem(self.h, _dp(q0), _dp(qd0), _dp(qdd0), _dp(q_des), self.O, _dp(obs), threads)
        if rc != 0:
            raise RuntimeError("oracle_set_problem failed")
        self.m = lib().oracle_num_constraints(self.h)
        return self

    def set_problem_armtd(self, q0, qd0, q_des, jrs, k_range, obstacles, threads=0):
        """ARMTD comparison mode (CMP/armtd_main.cu:36-156): jrs [n,6,T], k_range [n]."""
        q0, qd0, q_des, k_range = [np.ascontiguousarray(a, dtype=np.float64) for a in (q0, qd0, q_des, k_range)]
        jrs = np.ascontiguousarray(np.asarray(jrs, dtype=np.float64).reshape(self.n, 6, self.T))
        obs = np.ascontiguousarray(obstacles, dtype=np.float64).reshape(-1, 12)
        self.O = obs.shape[0]
        if self.O == 0:
            obs = np.zeros((1, 12))
        if lib().oracle_set_problem_armtd(self.h, _dp(q0), _dp(qd0), _dp(q_des), _dp(jrs), _dp(k_range), self.O, _dp(obs), threads) != 0:
            raise RuntimeError("oracle_set_problem_armtd failed")
        self.m = lib().oracle_num_constraints(self.h)
        return self

    @property
    def build_ms(self):
        return lib().oracle_build_ms(self.h)

    def eval_g_jac(self, k, want_g=True, want_jac=True, threads=0):
        k = np.ascontiguousarray(k, dtype=np.float64)
        g = np.zeros(self.m) if want_g else None
        jac = np.zeros((self.m, self.n)) if want_jac else None
        lib().oracle_eval_g_jac(self.h, _dp(k), _dp(g) if want_g else None, _dp(jac) if want_jac else None, threads)
        return g, jac

    def time_eval(self, ks, reps, threads=0):
        ks = np.ascontiguousarray(ks, dtype=np.float64).reshape(-1, self.n)
        g = np.zeros(self.m)
        jac = np.zeros((self.m, self.n))
        return lib().oracle_time_eval(self.h, _dp(ks), ks.shape[0], reps, _dp(g), _dp(jac), threads)

    def bounds(self):
        xl, xu = np.zeros(self.n), np.zeros(self.n)
        gl, gu = np.zeros(self.m), np.zeros(self.m)
        lib().oracle_get_bounds(self.h, _dp(xl), _dp(xu), _dp(gl), _dp(gu))
        return xl, xu, gl, gu

    def eval_f(self, k):
        k = np.ascontiguousarray(k, dtype=np.float64)
        return lib().oracle_eval_f(self.h, _dp(k))

    def eval_grad_f(self, k):
        k = np.ascontiguousarray(k, dtype=np.float64)
        out = np.zeros(self.n)
        lib().oracle_eval_grad_f(self.h, _dp(k), _dp(out))
        return out

    def torque_radius(self):
        out = np.zeros((self.n, self.T))
        lib().oracle_get_torque_radius(self.h, _dp(out))
        return out

    def link_generators(self):
        out = np.zeros((self.T, self.J, 3, 6))
        lib().oracle_get_link_generators(self.h, _dp(out))
        return out

    def hyperplanes(self):
        A = np.zeros((self.T, self.J, self.O, 36, 3))
        d = np.zeros((self.T, self.J, self.O, 36))
        delta = np.zeros((self.T, self.J, self.O, 36))
        lib().oracle_get_hyperplanes(self.h, _dp(A), _dp(d), _dp(delta))
        return A, d, delta

    def pz(self, which, i, t):
        """which: 'link' | 'torque' | 'disturbance' -> (center, indep, keys[M], coeffs[M, sz])."""
        w = {"link": 0, "torque": 1, "disturbance": 2}[which]
        sz = 3 if w == 0 else 1
        M = lib().oracle_pz_size(self.h, w, i, t)
        center, indep = np.zeros(sz), np.zeros(sz)
        keys = np.zeros(max(M, 1), dtype=np.uint64)
        coeffs = np.zeros((max(M, 1), sz))
        lib().oracle_pz_get(self.h, w, i, t, _dp(center), _dp(indep), keys.ctypes.data_as(C.POINTER(C.c_uint64)), _dp(coeffs))
        return center, indep, keys[:M], coeffs[:M]

    def table_sizes(self):
        v = [C.c_int64() for _ in range(4)]
        lib().oracle_table_sizes(self.h, *[C.byref(x) for x in v])
        return dict(sum_link=v[0].value, sum_torque=v[1].value, max_link=v[2].value, max_torque=v[3].value)

    def stats(self):
        out = (C.c_uint64 * 6)()
        lib().oracle_stats(self.h, out)
        keys = ["mul_calls", "mul_pairs", "simplify_calls", "simplify_terms", "max_raw_terms", "max_out_terms"]
        return dict(zip(keys, [int(x) for x in out]))

    def min_margin(self):
        """Smallest relative distance of any summed monomial norm to SIMPLIFY_THRESHOLD during the build: a value
        below ~1e-9 means a last-bit difference could flip a monomial between 'kept' and 'pruned'."""
        return lib().oracle_min_margin(self.h)

    def slice_torque(self, k):
        k = np.ascontiguousarray(k, dtype=np.float64)
        out = np.zeros((self.T, self.n))
        lib().oracle_slice_torque(self.h, _dp(k), _dp(out))
        return out

    def slice_links(self, k):
        k = np.ascontiguousarray(k, dtype=np.float64)
        out = np.zeros((self.T, self.J, 3))
        lib().oracle_slice_links(self.h, _dp(k), _dp(out))
        return out


def max_threads():
    return lib().oracle_max_threads()


def pz_op(op, operands, consts=None, r=0, threshold=5e-4, out_cap=1 << 16):
    """One PZ operator of the restatement (op codes of armour_debug_pz_op).  operands: list of dicts
    {sz, keys[cnt] u64, coef[cnt, sz], cen[sz], ind[sz]}.  Returns dict(keys, coef, cen, ind, min_margin)."""
    n = len(operands)
    sz = (C.c_int * n)(*[o["sz"] for o in operands])
    cnt = (C.c_int * n)(*[len(o["keys"]) for o in operands])
    ks = [np.ascontiguousarray(o["keys"], dtype=np.uint64) if len(o["keys"]) else np.zeros(1, np.uint64) for o in operands]
    cs = [np.ascontiguousarray(o["coef"], dtype=np.float64).reshape(-1) if len(o["keys"]) else np.zeros(1) for o in operands]
    kp = (C.POINTER(C.c_uint64) * n)(*[k.ctypes.data_as(C.POINTER(C.c_uint64)) for k in ks])
    cp = (C.POINTER(C.c_double) * n)(*[_dp(c) for c in cs])
    cen, ind = np.zeros((n, 9)), np.zeros((n, 9))
    for i, o in enumerate(operands):
        cen[i, :o["sz"]] = o["cen"]; ind[i, :o["sz"]] = o["ind"]
    cst = np.zeros(4) if consts is None else np.ascontiguousarray(np.concatenate([np.asarray(consts, dtype=np.float64), np.zeros(4)])[:4])
    ok, oc, misc = np.zeros(out_cap, np.uint64), np.zeros(out_cap * 9), np.zeros(32)
    rc = lib().oracle_pz_op(op, n, sz, cnt, kp, cp, _dp(cen), _dp(ind), _dp(cst), r, threshold, out_cap,
                            ok.ctypes.data_as(C.POINTER(C.c_uint64)), _dp(oc), _dp(misc))
    if rc != 0:
        raise RuntimeError("oracle_pz_op overflow")
    m, osz = int(misc[0]), int(misc[1])
    return dict(keys=ok[:m].copy(), coef=oc[:m * osz].reshape(m, osz).copy(), cen=misc[3:3 + osz].copy(), ind=misc[12:12 + osz].copy(), min_margin=misc[2])


def robust_controller(Kr, alpha, V_max, r_thr, q, qd, q_des, qd_des, qdd_des, eps=0.03, robot=None):
    """CPU restatement of the reference's kinova_controller MEX for ONE state (controller_oracle.cpp).
    Returns dict(u, tau, v, tau_interval [n,2], inside)."""
    L = lib()
    rb = robot if robot is not None else kinova_robot()
    n = rb.num_factors
    a = [np.ascontiguousarray(x, dtype=np.float64).ravel() for x in (q, qd, q_des, qd_des, qdd_des)]
    kr = np.ascontiguousarray(np.broadcast_to(np.asarray(Kr, dtype=np.float64).ravel(), (n,)))
    u, tau, v, ti = np.zeros(n), np.zeros(n), np.zeros(n), np.zeros((n, 2))
    L.oracle_robust_controller.restype = C.c_int
    ok = L.oracle_robust_controller(C.byref(rb), C.c_double(eps), _dp(kr), C.c_double(alpha), C.c_double(V_max), C.c_double(r_thr),
                                    *[_dp(x) for x in a], _dp(u), _dp(tau), _dp(v), _dp(ti))
    return dict(u=u, tau=tau, v=v, tau_interval=ti, inside=bool(ok))


def pass_rnea_scaled(s_m, s_I, q, qd, qda, qdd, gravity=True, robot=None):
    """Nominal passivity RNEA of the controller with masses / CoM-frame inertias scaled by (1 + s_m[i]) / (1 + s_I[i])."""
    L = lib()
    rb = robot if robot is not None else kinova_robot()
    n = rb.num_factors
    a = [np.ascontiguousarray(x, dtype=np.float64).ravel() for x in (s_m, s_I, q, qd, qda, qdd)]
    tau = np.zeros(n)
    L.oracle_pass_rnea_scaled.restype = None
    L.oracle_pass_rnea_scaled(C.byref(rb), *[_dp(x) for x in a], C.c_int(1 if gravity else 0), _dp(tau))
    return tau
