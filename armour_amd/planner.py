"""Host-side mirror of the reference's planner interface for the hot path.

`ArmourNLP` keeps the method names and argument meaning of the reference's `armtd_NLP : Ipopt::TNLP`
(RT/NLPclass.h; RT/ = kinova_src/kinova_simulator_interfaces/kinova_planner_realtime/) so that the parity
tests read like calls into the reference:

    nlp = ArmourNLP(T=100)                      # Obstacles ctor + KinematicsDynamics ctor
    nlp.set_parameters(q0, qd0, qdd0, q_des, obstacles)   # P1: armour_main.cu:86-216 + set_parameters
    n, m, nnz = nlp.get_nlp_info()
    g = nlp.eval_g(x); J = nlp.eval_jac_g(x)    # NLPclass.cu:272-396 (one fused device launch each)

and `eval_constraint(k)` returns the MATLAB callback shape `[h, heq, grad_h, grad_heq]` of
KSI/uarmtd_planner.m:776-796.  All arithmetic happens in libarmour_hip.so on the MI355X; this file only
marshals numpy arrays across the C ABI.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import ArmourLimits, ArmourParams, ArmourRobot, check


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def kinova_robot():
    r = ArmourRobot()
    _lib.load().armour_robot_kinova_gen3_no_gripper(C.byref(r))
    return r


def kinova_gripper_robot():
    r = ArmourRobot()
    _lib.load().armour_robot_kinova_gen3_gripper(C.byref(r))
    return r


def fetch_robot(payload_mass_uncertainty=0.0):
    """CMP/FetchInfo.h (9 links, 7 factors, mixed joint axes; link boxes / M_max are stand-ins, include/armour_robot_fetch.h).
    payload_mass_uncertainty > 0 widens the mass (and inertia) interval of the last link to +-that fraction: BASELINE
    configs[4], "payload-mass uncertainty"."""
    r = ArmourRobot()
    _lib.load().armour_robot_fetch(C.byref(r))
    if payload_mass_uncertainty:
        r.mass_uncertainty_link[r.num_joints - 1] = payload_mass_uncertainty
        r.inertia_uncertainty_link[r.num_joints - 1] = payload_mass_uncertainty
    return r


def fetch8_robot(payload_mass_uncertainty=0.0):
    """"Fetch 8-DOF" of BASELINE configs[4]: the Fetch arm behind a torso yaw joint -- 9 links, 8 factors (include/armour_robot_fetch.h says what is
    assumed; only the 128-bit-key library fills it: ARMOUR_KEY128=1).  payload_mass_uncertainty widens the last (gripper) link's intervals."""
    r = ArmourRobot()
    check(_lib.load().armour_robot_fetch8(C.byref(r)))
    if payload_mass_uncertainty:
        r.mass_uncertainty_link[r.num_joints - 1] = payload_mass_uncertainty
        r.inertia_uncertainty_link[r.num_joints - 1] = payload_mass_uncertainty
    return r


def default_params(T=128):
    p = ArmourParams()
    _lib.load().armour_params_default(C.byref(p), T)
    return p


def desired_trajectory(q0, qd0, qdd0, k, t, k_range=None, duration=1.0, t_plan=0.5, previous=None):
    """uarmtd_planner.desired_trajectory for traj_type 'bernstein' (KSI/uarmtd_planner.m:846-925).

    k finite: (q, qd, qdd) at time t of the planned degree-5 Bezier curve (armour_desired_trajectory).
    k is None / NaN (no plan found): braking -- if t <= t_plan and the arm is moving, evaluate `previous`, the
    desired-trajectory callable of the previous plan, at t + t_plan (:909-911); otherwise hold q0 with zero
    velocity and acceleration (:912-916).  Returns three [n] arrays."""
    q0 = np.ascontiguousarray(q0, dtype=np.float64).ravel()
    qd0 = np.ascontiguousarray(qd0, dtype=np.float64).ravel()
    qdd0 = np.ascontiguousarray(qdd0, dtype=np.float64).ravel()
    n = q0.size
    if k is None or np.any(np.isnan(np.asarray(k, dtype=np.float64))):
        if t <= t_plan and np.linalg.norm(qd0) > 1e-8 and previous is not None:
            return previous(t + t_plan)
        return q0.copy(), np.zeros(n), np.zeros(n)
    L = _lib.load()
    kr = np.full(n, np.pi / 48) if k_range is None else np.ascontiguousarray(k_range, dtype=np.float64).ravel()
    kk = np.ascontiguousarray(k, dtype=np.float64).ravel()
    q, qd, qdd = np.zeros(n), np.zeros(n), np.zeros(n)
    check(L.armour_desired_trajectory(n, _dp(q0), _dp(qd0), _dp(qdd0), _dp(kr), float(duration), _dp(kk), float(t), _dp(q), _dp(qd), _dp(qdd)))
    return q, qd, qdd


def _violation_dict(v):
    return dict(l1_violation=v.l1_violation, worst=v.worst, worst_row=v.worst_row, n_violated=v.n_violated,
                n_outside_slack=v.n_outside_slack, feasible=bool(v.feasible))


def _solve_options(L, max_iterations, tolerance, max_wall_time_s, host_qp, device_qp=False):
    opt = _lib.ArmourSolveOptions()
    L.armour_solve_options_default(C.byref(opt))
    if max_iterations is not None:
        opt.max_iterations = max_iterations
    if tolerance is not None:
        opt.tolerance = tolerance
    if max_wall_time_s is not None:
        opt.max_wall_time_s = max_wall_time_s
    if host_qp:   # the host-driven form (one launch per evaluation, QPs on the host) whatever the batch size
        opt.force_host_qp = 1.0
    elif device_qp:   # the persistent-kernel form whatever the batch size (automatic since round 6, for every batch size)
        opt.force_host_qp = -1.0
    return opt


def _solve_dicts(res, n):
    return [dict(k_opt=np.array(r.k_opt[:n]), cost=r.cost, max_violation=r.max_violation, feasible=bool(r.feasible),
                 iterations=r.iterations, evaluations=r.evaluations, status=r.status, time_ms=r.time_ms) for r in res]


def batch_partition(B, n_slots):
    """first[d] of armour_batch_*: slot d owns problems [first[d], first[d+1]) (pure host arithmetic in the library)."""
    first = (C.c_int32 * (n_slots + 1))()
    check(_lib.load().armour_batch_partition(B, n_slots, first))
    return list(first)


class ArmourBatchNLP:
    """B independent planning problems dealt to several MI355X from ONE host thread (include/armour_hip.h, armour_batch_*):
    one handle + stream + host thread per entry of `devices`, contiguous blocks of problems, results gathered in place."""

    def __init__(self, devices, robot=None, params=None, T=128, limits=None):
        self.L = _lib.load()
        self.robot = robot if robot is not None else kinova_robot()
        self.params = params if params is not None else default_params(T)
        self.T, self.J, self.n = self.params.num_time_steps, self.robot.num_joints, self.robot.num_factors
        self.limits = limits if limits is not None else ArmourLimits()
        self.devices = list(devices)
        dv = (C.c_int32 * len(self.devices))(*self.devices)
        h = C.c_void_p()
        check(self.L.armour_batch_create(C.byref(self.robot), C.byref(self.params), C.byref(self.limits), dv, len(self.devices), C.byref(h)))
        self.h = h
        self.B = self.O = self.m = 0

    def close(self):
        if getattr(self, "h", None):
            self.L.armour_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, option, value):
        check(self.L.armour_batch_set_option(self.h, option, float(value)))
        return self

    def set_parameters(self, q0, qd0, qdd0, q_des, obstacles):
        q0, qd0, qdd0, q_des = [np.ascontiguousarray(np.atleast_2d(np.asarray(a, dtype=np.float64))) for a in (q0, qd0, qdd0, q_des)]
        B = q0.shape[0]
        obs = np.asarray(obstacles, dtype=np.float64)
        obs = np.ascontiguousarray(obs.reshape(B, -1, 12)) if obs.size else np.zeros((B, 0, 12))
        O = obs.shape[1]
        check(self.L.armour_batch_set_problems(self.h, B, O, _dp(q0), _dp(qd0), _dp(qdd0), _dp(q_des), _dp(obs) if O else None))
        b, n, m, g = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        check(self.L.armour_batch_get_sizes(self.h, C.byref(b), C.byref(n), C.byref(m), C.byref(g)))
        self.B, self.O, self.m = b.value, O, m.value
        return self

    def _k(self, x):
        return np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(self.B, self.n))

    def get_bounds_info(self):
        xl, xu = np.zeros(self.n), np.zeros(self.n)
        gl, gu = np.zeros((self.B, self.m)), np.zeros((self.B, self.m))
        check(self.L.armour_batch_get_bounds(self.h, _dp(xl), _dp(xu), _dp(gl), _dp(gu)))
        return xl, xu, gl, gu

    def eval_g_jac(self, x):
        g, jac = np.zeros((self.B, self.m)), np.zeros((self.B, self.m, self.n))
        check(self.L.armour_batch_eval_g_jac(self.h, _dp(self._k(x)), _dp(g), _dp(jac)))
        return g, jac

    def eval_violations(self, x):
        out = (_lib.ArmourViolation * self.B)()
        check(self.L.armour_batch_eval_violations(self.h, _dp(self._k(x)), out))
        return [_violation_dict(v) for v in out]

    def solve(self, max_iterations=None, tolerance=None, max_wall_time_s=None, host_qp=False, device_qp=False):
        opt = _solve_options(self.L, max_iterations, tolerance, max_wall_time_s, host_qp, device_qp)
        res = (_lib.ArmourSolveResult * self.B)()
        check(self.L.armour_batch_solve(self.h, C.byref(opt), res))
        return _solve_dicts(res, self.n)

    @property
    def build_ms(self):
        v = C.c_double()
        per = (C.c_double * len(self.devices))()
        check(self.L.armour_batch_get_build_ms(self.h, C.byref(v), per))
        return v.value

    def prune_margin(self):
        """[B]: armour_batch_get_prune_margin -- every problem's prune margin, whichever slot built it (ArmourNLP.prune_margin)."""
        out = np.zeros(self.B)
        check(self.L.armour_batch_get_prune_margin(self.h, _dp(out)))
        return out

    def build_info(self):
        """armour_batch_get_build_info: per slot, as ArmourNLP.build_info() (None for a slot without problems)."""
        out = (C.c_int32 * (4 * len(self.devices)))()
        check(self.L.armour_batch_get_build_info(self.h, out))
        kn = {0: None, 1: "per_step", 2: "time_vectorised"}
        return [None if out[4 * d] == 0 else {"kernel": kn[out[4 * d]], "waves": out[4 * d + 1], "sort_entries": out[4 * d + 2], "launches": out[4 * d + 3]}
                for d in range(len(self.devices))]


class ArmourNLP:
    """B independent planning problems on one MI355X (B = 1 is the reference's use)."""

    def __init__(self, robot=None, params=None, T=128, device=0, limits=None):
        self.L = _lib.load()
        self.robot = robot if robot is not None else kinova_robot()
        self.params = params if params is not None else default_params(T)
        self.T = self.params.num_time_steps
        self.J = self.robot.num_joints
        self.n = self.robot.num_factors
        self.limits = limits if limits is not None else ArmourLimits()
        h = C.c_void_p()
        check(self.L.armour_create(C.byref(self.robot), C.byref(self.params), C.byref(self.limits), device, C.byref(h)))
        self.h = h
        self.B = 0
        self.O = 0
        self.m = 0

    def _pinned(self, name, shape):
        """numpy view of a page-locked buffer owned by this object (re-used across calls of the same shape)."""
        cache = self.__dict__.setdefault("_pin", {})
        ent = cache.get(name)
        if ent is None or ent[1].shape != tuple(shape):
            if ent is not None:
                self.L.armour_free_pinned(ent[0])
            nbytes = int(np.prod(shape)) * 8
            ptr = C.c_void_p()
            check(self.L.armour_alloc_pinned(nbytes, C.byref(ptr)))
            arr = np.ctypeslib.as_array((C.c_double * (nbytes // 8)).from_address(ptr.value)).reshape(shape)
            ent = (ptr, arr)
            cache[name] = ent
        return ent[1]

    def close(self):
        for ptr, _ in self.__dict__.get("_pin", {}).values():
            self.L.armour_free_pinned(ptr)
        self.__dict__["_pin"] = {}
        if getattr(self, "h", None):
            self.L.armour_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ P1
    def set_parameters(self, q0, qd0, qdd0, q_des, obstacles):
        """Build the reach sets (RT/armour_main.cu:86-216).  Single problem: q0.. are [n], obstacles [O,12];
        batch: q0.. are [B,n], obstacles [B,O,12]."""
        q0, qd0, qdd0, q_des = [np.ascontiguousarray(np.atleast_2d(np.asarray(a, dtype=np.float64))) for a in (q0, qd0, qdd0, q_des)]
        obs = np.asarray(obstacles, dtype=np.float64)
        B = q0.shape[0]
        obs = np.ascontiguousarray(obs.reshape(B, -1, 12)) if obs.size else np.zeros((B, 0, 12))
        O = obs.shape[1]
        for a in (q0, qd0, qdd0, q_des):
            if a.shape != (B, self.n):
                raise ValueError(f"expected shape ({B},{self.n}), got {a.shape}")
        self._obs = obs
        check(self.L.armour_set_problems(self.h, B, O, _dp(q0), _dp(qd0), _dp(qdd0), _dp(q_des), _dp(obs) if O else None))
        self._after_set(B, O)
        return self

    def set_parameters_armtd(self, q0, qd0, q_des, jrs, k_range, obstacles):
        """ARMTD comparison mode (CMP/armtd_main.cu:36-216, the reference's second planner): constant-acceleration
        trajectory, cos/sin JRS from offline tables, no torque rows (m = J*T*O + 4n).  jrs: [n,6,T] (or [B,n,6,T]) --
        per joint the rows c_cos, g_cos, r_cos, c_sin, g_sin, r_sin of armtd.in; k_range: [n] (or [B,n]).  Every other
        method of this class then follows CMP/NLPclass.cu."""
        q0, qd0, q_des, k_range = [np.ascontiguousarray(np.atleast_2d(np.asarray(a, dtype=np.float64))) for a in (q0, qd0, q_des, k_range)]
        B = q0.shape[0]
        jrs = np.ascontiguousarray(np.asarray(jrs, dtype=np.float64).reshape(B, self.n, 6, self.T))
        obs = np.asarray(obstacles, dtype=np.float64)
        obs = np.ascontiguousarray(obs.reshape(B, -1, 12)) if obs.size else np.zeros((B, 0, 12))
        O = obs.shape[1]
        for a in (q0, qd0, q_des, k_range):
            if a.shape != (B, self.n):
                raise ValueError(f"expected shape ({B},{self.n}), got {a.shape}")
        self._obs = obs
        check(self.L.armour_set_problems_armtd(self.h, B, O, _dp(q0), _dp(qd0), _dp(q_des), _dp(jrs), _dp(k_range), _dp(obs) if O else None))
        self._after_set(B, O)
        return self

    def _after_set(self, B, O):
        b, n, m = C.c_int32(), C.c_int32(), C.c_int32()
        check(self.L.armour_get_sizes(self.h, C.byref(b), C.byref(n), C.byref(m)))
        self.B, self.O, self.m = b.value, O, m.value

    def debug_load_tables(self, q0, qd0, qdd0, q_des, tables):
        """Test hook: load reach-set tables built elsewhere (see include/armour_hip.h)."""
        q0, qd0, qdd0, q_des = [np.ascontiguousarray(np.atleast_2d(np.asarray(a, dtype=np.float64))) for a in (q0, qd0, qdd0, q_des)]
        t = {k: np.ascontiguousarray(v) for k, v in tables.items()}
        B, O = q0.shape[0], t["A"].shape[3]
        u64 = C.POINTER(C.c_uint64)
        i32 = C.POINTER(C.c_int32)
        check(self.L.armour_debug_load_tables(
            self.h, B, O, _dp(q0), _dp(qd0), _dp(qdd0), _dp(q_des),
            t["link_count"].ctypes.data_as(i32), _dp(t["link_center"]), t["link_keys"].ctypes.data_as(u64), _dp(t["link_coeffs"]), t["link_keys"].shape[-1],
            t["torque_count"].ctypes.data_as(i32), _dp(t["torque_center"]), t["torque_keys"].ctypes.data_as(u64), _dp(t["torque_coeffs"]), t["torque_keys"].shape[-1],
            _dp(t["A"]), _dp(t["d"]), _dp(t["delta"]), _dp(t["torque_radius"])))
        self._after_set(B, O)
        return self

    # ------------------------------------------------------------------ TNLP surface
    def get_nlp_info(self):
        """(n, m, nnz_jac_g) -- RT/NLPclass.cu:62-82 (dense Jacobian)."""
        return self.n, self.m, self.m * self.n

    def get_bounds_info(self):
        """x_l, x_u [n]; g_l, g_u [B,m] -- RT/NLPclass.cu:87-165."""
        xl, xu = np.zeros(self.n), np.zeros(self.n)
        gl, gu = np.zeros((self.B, self.m)), np.zeros((self.B, self.m))
        check(self.L.armour_get_bounds(self.h, _dp(xl), _dp(xu), _dp(gl), _dp(gu)))
        return xl, xu, gl, gu

    def get_starting_point(self):
        """x = 0 -- RT/NLPclass.cu:170-202."""
        return np.zeros((self.B, self.n))

    def _k(self, x):
        k = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(self.B, self.n))
        return k

    def eval_f(self, x):
        f = np.zeros(self.B)
        check(self.L.armour_eval_f(self.h, _dp(self._k(x)), _dp(f)))
        return f

    def eval_grad_f(self, x):
        gf = np.zeros((self.B, self.n))
        check(self.L.armour_eval_grad_f(self.h, _dp(self._k(x)), _dp(gf)))
        return gf

    def eval_g(self, x):
        g = np.zeros((self.B, self.m))
        check(self.L.armour_eval_g_jac(self.h, _dp(self._k(x)), _dp(g), None))
        return g

    def eval_jac_g(self, x):
        """values[B, m, n] (row-major, row = constraint) -- RT/NLPclass.cu:330-396."""
        jac = np.zeros((self.B, self.m, self.n))
        check(self.L.armour_eval_g_jac(self.h, _dp(self._k(x)), None, _dp(jac)))
        return jac

    def eval_g_jac(self, x, pinned=False, out=None):
        """pinned=True returns views of page-locked buffers owned by this object (valid until the next call); out=(g, jac) writes into the
        caller's own C-contiguous float64 arrays (an IPOPT TNLP's `g` and `values`, reused call after call)."""
        if out is not None:
            k, (g, jac) = self._k(x), out
            assert g.dtype == np.float64 and jac.dtype == np.float64 and g.flags.c_contiguous and jac.flags.c_contiguous
            assert g.size == self.B * self.m and jac.size == self.B * self.m * self.n
        elif pinned:
            k = self._pinned("k", (self.B, self.n)); k[...] = np.asarray(x, dtype=np.float64).reshape(self.B, self.n)
            # g and jac in ONE page-locked block, jac directly behind g: armour_eval_g_jac then brings both back in one transfer
            gj = self._pinned("gjac", (self.B * self.m * (1 + self.n),))
            g, jac = gj[:self.B * self.m].reshape(self.B, self.m), gj[self.B * self.m:].reshape(self.B, self.m, self.n)
        else:
            k, g, jac = self._k(x), np.zeros((self.B, self.m)), np.zeros((self.B, self.m, self.n))
        check(self.L.armour_eval_g_jac(self.h, _dp(k), _dp(g), _dp(jac)))
        return g, jac

    def eval_g_jac_device(self, d_k, d_g, d_jac, stream=0):
        """Asynchronous, device pointers (ints), e.g. torch tensors' .data_ptr(); stream = hipStream_t as int."""
        check(self.L.armour_eval_g_jac_device(self.h, d_k, d_g, d_jac, stream))

    def eval_g_jac_device_steps(self, d_k, steps, d_g, d_jac, stream=0):
        """`steps` back-to-back fused evaluations (d_k holds [steps][B][n]); asynchronous."""
        check(self.L.armour_eval_g_jac_device_steps(self.h, d_k, steps, d_g, d_jac, stream))

    def prepare_steps(self, d_k, steps, d_g, d_jac):
        """Build (without launching) the graph eval_g_jac_device_steps uses for these arguments."""
        check(self.L.armour_prepare_steps(self.h, d_k, steps, d_g, d_jac))

    def eval_g_jac_device_multi(self, d_k, points, d_g, d_jac, stream=0):
        """`points` evaluations of the same problems in one launch: d_k [points][B][n] -> d_g [points][B][m],
        d_jac [points][B][m][n]; asynchronous.  Bit-identical to `points` single launches."""
        check(self.L.armour_eval_g_jac_device_multi(self.h, d_k, points, d_g, d_jac, stream))

    def finalize_solution(self, g):
        """feasible[B] from g[B,m] with the reference's slack thresholds -- RT/NLPclass.cu:422-538."""
        g = np.ascontiguousarray(np.asarray(g, dtype=np.float64).reshape(self.B, self.m))
        feas = np.zeros(self.B, dtype=np.int32)
        check(self.L.armour_check_feasible(self.h, _dp(g), feas.ctypes.data_as(C.POINTER(C.c_int32))))
        return feas.astype(bool)

    def set_option(self, option, value):
        """Per-handle option (include/armour_hip.h): e.g. set_option(_lib.OPT_P1_BUILD, 1) holds the reach-set build to the
        per-time-step kernel (bit-identical tables for a problem whatever the batch size)."""
        check(self.L.armour_set_option(self.h, option, float(value)))
        return self

    def get_option(self, option):
        """armour_get_option: the handle's current value (a fall-back the library took by itself shows here, e.g. OPT_P1_STEP_TWO_CU = 0)."""
        v = C.c_double()
        check(self.L.armour_get_option(self.h, option, C.byref(v)))
        return v.value

    def eval_violations(self, x):
        """Reduced outputs: one record per problem -- the row test of finalize_solution (RT/NLPclass.cu:422-538) applied on
        the device, g never leaves it.  Returns a list of dicts."""
        out = (_lib.ArmourViolation * self.B)()
        check(self.L.armour_eval_violations(self.h, _dp(self._k(x)), out))
        return [_violation_dict(v) for v in out]

    def row_relevance(self):
        """(relevant [B, m] bool, relevant collision rows per problem [B], device ms of the test) -- armour_get_row_relevance: the pruned
        constraint list of KSI/uarmtd_planner.m:577-583,628-690 (a row marked False cannot be violated for any k in [-1, 1]^n)."""
        rel = np.zeros((self.B, self.m), dtype=np.uint8)
        cnt = np.zeros(self.B, dtype=np.int32)
        ms = C.c_double()
        check(self.L.armour_get_row_relevance(self.h, rel.ctypes.data_as(C.POINTER(C.c_uint8)), cnt.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(ms)))
        return rel.astype(bool), cnt, ms.value

    def solver_rows(self):
        """(mask [B, m] bool, listed collision rows [B], listed torque rows [B], device ms) -- armour_get_solver_rows: the rows that can pass
        armour_solve's candidate filter for some k (what its culled device form walks)."""
        mask = np.zeros((self.B, self.m), dtype=np.uint8)
        cnt, tq = np.zeros(self.B, dtype=np.int32), np.zeros(self.B, dtype=np.int32)
        ms = C.c_double()
        i32 = C.POINTER(C.c_int32)
        check(self.L.armour_get_solver_rows(self.h, mask.ctypes.data_as(C.POINTER(C.c_uint8)), cnt.ctypes.data_as(i32), tq.ctypes.data_as(i32), C.byref(ms)))
        return mask.astype(bool), cnt, tq, ms.value

    def eval_violations_device(self, d_k, d_out, stream=0):
        """Asynchronous: d_k [B][n] doubles, d_out [B] ArmourViolation records (32 B each), device pointers (ints)."""
        check(self.L.armour_eval_violations_device(self.h, d_k, d_out, stream))

    def solve(self, max_iterations=None, tolerance=None, max_wall_time_s=None, host_qp=False, device_qp=False):
        """OptimizeTNLP + finalize_solution for all B problems (RT/armour_main.cu:237-304): returns a list of dicts
        (k_opt, cost, feasible, iterations, evaluations, status, time_ms).  host_qp / device_qp hold the solver to one of its two forms
        (same iterates; automatic: the persistent kernel, for every batch size since round 6)."""
        opt = _solve_options(self.L, max_iterations, tolerance, max_wall_time_s, host_qp, device_qp)
        res = (_lib.ArmourSolveResult * self.B)()
        check(self.L.armour_solve(self.h, C.byref(opt), res))
        return [dict(k_opt=np.array(r.k_opt[:self.n]), cost=r.cost, max_violation=r.max_violation, feasible=bool(r.feasible),
                     iterations=r.iterations, evaluations=r.evaluations, status=r.status, time_ms=r.time_ms) for r in res]

    # ------------------------------------------------------------------ MATLAB callback shape
    def eval_constraint(self, k, b=0):
        """[h, heq, grad_h, grad_heq] with h <= 0 feasible and grad_h sized n_k x n_constraints
        (KSI/uarmtd_planner.m:776-796).  Two-sided rows g_l <= g <= g_u are split into g - g_u <= 0 and
        g_l - g <= 0; collision rows (g_l = -1e19) contribute only g <= 0."""
        x = np.zeros((self.B, self.n))
        x[b] = k
        g, jac = self.eval_g_jac(x)
        _, _, gl, gu = self.get_bounds_info()
        g, jac, gl, gu = g[b], jac[b], gl[b], gu[b]
        two_sided = gl > -1e18
        h = np.concatenate([g - gu, (gl - g)[two_sided]])
        grad_h = np.concatenate([jac, -jac[two_sided]], axis=0).T
        return h, np.zeros(0), grad_h, np.zeros((self.n, 0))

    def debug_pz_op(self, op, operands, consts=None, r=0, out_cap=4096):
        """Test hook: one device PZ operator (include/armour_hip.h: armour_debug_pz_op).  operands: list of dicts
        {sz, keys[cnt], coef[cnt, sz], cen[sz], ind[sz], ind2[sz] (optional)}."""
        n = len(operands)
        i32 = C.c_int32 * n
        sz = i32(*[o["sz"] for o in operands])
        cnt = i32(*[len(o["keys"]) for o in operands])
        ks = [np.ascontiguousarray(o["keys"], dtype=np.uint64) if len(o["keys"]) else np.zeros(1, np.uint64) for o in operands]
        cs = [np.ascontiguousarray(o["coef"], dtype=np.float64).reshape(-1) if len(o["keys"]) else np.zeros(1) for o in operands]
        kp = (C.POINTER(C.c_uint64) * n)(*[k.ctypes.data_as(C.POINTER(C.c_uint64)) for k in ks])
        cp = (C.POINTER(C.c_double) * n)(*[_dp(c) for c in cs])
        cen, ind, ind2 = np.zeros((n, 9)), np.zeros((n, 9)), np.zeros((n, 9))
        for i, o in enumerate(operands):
            cen[i, :o["sz"]] = o["cen"]; ind[i, :o["sz"]] = o["ind"]; ind2[i, :o["sz"]] = o.get("ind2", o["ind"])
        cst = np.zeros(4) if consts is None else np.ascontiguousarray(np.concatenate([np.asarray(consts, dtype=np.float64), np.zeros(4)])[:4])
        ok, oc, misc = np.zeros(out_cap, np.uint64), np.zeros(out_cap * 9), np.zeros(64)
        check(self.L.armour_debug_pz_op(self.h, op, n, sz, cnt, kp, cp, _dp(cen), _dp(ind), _dp(ind2), _dp(cst), r, out_cap,
                                        ok.ctypes.data_as(C.POINTER(C.c_uint64)), _dp(oc), _dp(misc)))
        m, osz = int(misc[0]), int(misc[1])
        return dict(keys=ok[:m].copy(), coef=oc[:m * osz].reshape(m, osz).copy(), cen=misc[3:3 + osz].copy(), ind=misc[12:12 + osz].copy(),
                    ind2=misc[21:21 + osz].copy(), flags=int(misc[2]), cycles=float(misc[30]), raw_terms=int(misc[31]), prof=misc[32:60].copy(),
                    margin_sq=float(misc[60]), thr_sq=float(misc[61]))

    # ------------------------------------------------------------------ diagnostics / tables
    def torque_radius(self):
        out = np.zeros((self.B, self.n, self.T))
        check(self.L.armour_get_torque_radius(self.h, _dp(out)))
        return out

    def link_generators(self):
        out = np.zeros((self.B, self.T, self.J, 3, 6))
        check(self.L.armour_get_link_generators(self.h, _dp(out)))
        return out

    def link_centers(self, x):
        out = np.zeros((self.B, self.T, self.J, 3))
        check(self.L.armour_get_link_centers(self.h, _dp(self._k(x)), _dp(out)))
        return out

    def pz(self, which, i, t, b=0):
        """which: 'link' | 'torque' -> (center, indep, keys[M], coeffs[M, sz])."""
        w = {"link": 0, "torque": 1}[which]
        sz = 3 if w == 0 else 1
        cap = 4096
        cen = np.zeros(2 * sz)
        keys = np.zeros(cap, dtype=np.uint64)
        co = np.zeros((cap, sz))
        cnt = check(self.L.armour_get_pz(self.h, b, w, i, t, _dp(cen), keys.ctypes.data_as(C.POINTER(C.c_uint64)), _dp(co), cap))
        return cen[:sz], cen[sz:], keys[:cnt], co[:cnt]

    def polyzonotope(self, which, i, t, b=0):
        """The same PZ as CORA polyZonotope fields (armour_amd/cora.py): dict(c, G, Grest, expMat, id)."""
        from . import cora
        c, ind, keys, co = self.pz(which, i, t, b=b)
        return cora.to_polyzonotope(c, ind, keys, co, self.n)

    def table_sizes(self):
        out = (C.c_int64 * 4)()
        check(self.L.armour_get_table_sizes(self.h, out))
        return dict(sum_link=out[0], sum_torque=out[1], max_link=out[2], max_torque=out[3])

    def hyperplanes(self):
        A = np.zeros((self.B, self.T, self.J, self.O, 36, 3))
        d = np.zeros((self.B, self.T, self.J, self.O, 36))
        delta = np.zeros((self.B, self.T, self.J, self.O, 36))
        check(self.L.armour_get_hyperplanes(self.h, _dp(A), _dp(d), _dp(delta)))
        return A, d, delta

    @property
    def build_ms(self):
        v = C.c_double()
        check(self.L.armour_get_build_ms(self.h, C.byref(v)))
        return v.value

    def build_info(self):
        """How the last set_parameters built its tables (armour_get_build_info): kernel "per_step" | "time_vectorised",
        waves per block and sort-buffer entries of its last launch, reach-set launches in all."""
        out = (C.c_int32 * 4)()
        check(self.L.armour_get_build_info(self.h, out))
        return {"kernel": {0: None, 1: "per_step", 2: "time_vectorised"}[out[0]], "waves": out[1], "sort_entries": out[2], "launches": out[3]}

    def plane_skip(self):
        """[B] uint64: bit p set = half-space p is never needed by any collision row of the problem (armour_get_plane_skip)."""
        out = np.zeros(self.B, dtype=np.uint64)
        check(self.L.armour_get_plane_skip(self.h, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out

    def prune_margin(self):
        """[B]: min |norm - SIMPLIFY_THRESHOLD| / threshold over every simplify() verdict of the last reach-set build (armour_get_prune_margin):
        how close the build came to a prune flip.  The oracle's counterpart is Oracle.min_margin()."""
        out = np.zeros(self.B)
        check(self.L.armour_get_prune_margin(self.h, _dp(out)))
        return out

    def effective_bytes(self):
        """Bytes ONE fused evaluation of all B problems moves by the layout it actually reads (tables built by P1; common.h
        armour_plane_index, p2_tiles.h collision_block, p1_reach.hip planes_of_group), counted in whole 128-B lines: rows are
        padded to a multiple of 16, so per collision row and live plane below 21 the pairs {Ax,Ay} and {Az,delta} (32 B), per
        PAIR of link x link planes with a live member the pair of deltas (16 B), plus d (8 B per live plane) for the launches
        that read it instead of recomputing it (fewer than 8 problems and fewer than 32 768 rows); the compact normals as the
        three lines of each (link, time step) record; the obstacle centres; the link / torque PZ tables with their counts,
        centres and radii; k in, g and the dense Jacobian out.  Unlike algorithmic_bytes() -- defined by the reference's
        formulation, 1440 B per row -- a fraction of the HBM peak computed from this number cannot exceed 1."""
        ts = self.table_sizes()
        JT, Q, nT = self.J * self.T, self.J * self.T * self.O, self.n * self.T
        Qs = (Q + 15) // 16 * 16
        dfc = self.B >= 8 or self.B * Q >= 32768    # p1_reach.hip store_d / armour_make_tables (api.hip)
        line = lambda nbytes: (nbytes + 127) // 128 * 128
        total = 0
        for sk in self.plane_skip():
            live = [p for p in range(36) if not (int(sk) >> p) & 1]
            lo, hi = sum(1 for p in live if p < 21), sum(1 for p in live if p >= 21)
            hi_pairs = len({(p - 21) >> 1 for p in live if p >= 21})
            per_row = 32 * lo + 16 * hi_pairs + (0 if dfc else 8 * (lo + hi))
            total += Qs * per_row + (JT * 384 if hi else 0) + (line(24 * self.O) if dfc else 0)
        total += 32 * ts["sum_link"] + 16 * ts["sum_torque"] + self.B * (JT * 52 + nT * 20)
        total += self.B * (line(8 * self.m * (1 + self.n)) + line(8 * self.n))
        return total

    def algorithmic_bytes(self):
        """B_alg of one fused eval over all B problems:
        1440*T*J*O per problem + 32*SumM_link + 16*SumM_torque + 8*m*(1+n) per problem."""
        ts = self.table_sizes()
        return (1440 * self.T * self.J * self.O * self.B + 32 * ts["sum_link"] + 16 * ts["sum_torque"]
                + 8 * self.m * (1 + self.n) * self.B)
