"""ctypes binding of armour_amd/lib/libarmour_hip.so (the C ABI in include/armour_hip.h).

There is no CPU path: if the shared library is missing or fails to load this module raises, and
`armour_create` fails with ARMOUR_EDEVICE when no MI355X is visible.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ARMOUR_KEY128=1 selects the second ABI of include/armour_types.h for the whole process: libarmour_hip_k128.so (128-bit monomial keys, 8 factors)
KEY128 = os.environ.get("ARMOUR_KEY128", "0") not in ("", "0")
LIB_PATH = os.environ.get("ARMOUR_HIP_LIB") or os.path.join(_HERE, "lib", "libarmour_hip_k128.so" if KEY128 else "libarmour_hip.so")  # env override: A/B builds

MAXJ = 9                    # ARMOUR_MAX_JOINTS
MAXF = 8 if KEY128 else 7   # ARMOUR_MAX_FACTORS (checked against armour_abi_max_factors() of the library that was loaded)

OK, EINVAL, EDEVICE, ECAPACITY, ESTATE = 0, -1, -2, -3, -4


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


class ArmourLimits(C.Structure):
    _fields_ = [
        ("max_batch", C.c_int32), ("max_obstacles", C.c_int32), ("link_monomials", C.c_int32),
        ("torque_monomials", C.c_int32), ("work_monomials", C.c_int32), ("raw_terms", C.c_int32),
    ]


class ArmourSolveOptions(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("max_line_search", C.c_int32), ("tolerance", C.c_double),
                ("max_wall_time_s", C.c_double), ("force_host_qp", C.c_double), ("reserved", C.c_double * 3)]


class ArmourSolveResult(C.Structure):
    _fields_ = [("k_opt", C.c_double * MAXF), ("cost", C.c_double), ("max_violation", C.c_double), ("feasible", C.c_int32),
                ("iterations", C.c_int32), ("evaluations", C.c_int32), ("status", C.c_int32), ("time_ms", C.c_double)]


class ArmourViolation(C.Structure):
    _fields_ = [("l1_violation", C.c_double), ("worst", C.c_double), ("worst_row", C.c_int32), ("n_violated", C.c_int32),
                ("n_outside_slack", C.c_int32), ("feasible", C.c_int32)]


OPT_P1_BUILD = 1   # ARMOUR_OPT_P1_BUILD: 0 automatic, 1 per time step, 2 time-vectorised
OPT_P1_WORK_MEMORY_MB = 2   # ARMOUR_OPT_P1_WORK_MEMORY_MB: cap on the time-vectorised build's work memory, MiB (0: none)
# the other per-handle options of include/armour_hip.h (launch shapes: bit-identical keys / coefficients / centres for every value)
OPT_P1_KEEP_WORK_MEMORY = 3
OPT_P1_STEP_WAVES = 101
OPT_P1_STEP_FREE = 102
OPT_P1_STEP_SPLIT_FK = 103
OPT_P1_STEP_AUX3 = 104
OPT_P1_MAX_WAVES_PER_CU = 105
OPT_P1_TWO_PASS = 106
OPT_P1_STEP_PAIRS = 107
OPT_P1_STEP_QUEUE = 109
OPT_P1_STEP_TAIL_CROSS = 108
OPT_P1_STEP_TWO_CU = 123     # 3 (default) | 0..2: a lone problem's time step on two compute units (include/armour_hip.h); 10 + level: the fall-back's test hook
OPT_P1_STEP_LEAN_BACK = 124  # 1 (default) | 0: with level 3, the backward pass's f-recursion on the helper block as well
OPT_P1_TV_TAIL_CROSS = 121
OPT_P1_TV_MIN_GROUPS = 110
OPT_P1_TV_WAVES = 111
OPT_P1_TV_FREE = 112
OPT_P1_TV_SPLIT_FK = 113
OPT_P1_TV_DEDICATED = 114
OPT_P1_TV_HELP_SHIFT = 115
OPT_P1_TV_HELPERS = 116
OPT_P1_TV_HELP_MIN = 117
OPT_P1_TV_HELP_N = 118
OPT_P1_TV_AUX3 = 119
OPT_P1_FULL_PLANES = 120
OPT_P1_TV_ROW_WIDTH = 122   # 0 automatic | 50 | 64 doubles per row of the time-vectorised kernel's work slots
OPT_P2_EX = 130
OPT_STEPS_GRAPH_MIN = 131
OPT_PINNED_MODE = 132
OPT_CULL_ROWS = 133   # 1: armour_eval_violations* evaluate only the rows that can be violated for some k (armour_get_row_relevance)
OPT_SOLVE_SUB_TILES = 140
OPT_SOLVE_DEVICE = 141
OPT_SOLVE_CUT_TILES = 142
OPT_SOLVE_BLOCKS = 143
OPT_SOLVE_SUB_BATCH = 144
OPT_SOLVE_ROW_CAP = 145
OPT_SOLVE_HARD_CAP_S = 146
OPT_SOLVE_WAVES_PER_SIMD = 147
OPT_SOLVE_CULL = 148   # -1 automatic | 0 | 1: the device-resident solve walks only the rows that can ever be QP candidates (same iterates)

# every symbol include/armour_hip.h declares (tests check the .so exports all of them)
EXPORTS = [
    "armour_robot_kinova_gen3_no_gripper", "armour_robot_kinova_gen3_gripper", "armour_robot_fetch", "armour_robot_fetch8", "armour_params_default", "armour_create", "armour_destroy",
    "armour_last_error", "armour_device_available", "armour_alloc_pinned", "armour_free_pinned", "armour_set_problems", "armour_set_problems_armtd", "armour_get_sizes",
    "armour_get_bounds", "armour_eval_f", "armour_eval_grad_f", "armour_eval_g_jac",
    "armour_eval_g_jac_device", "armour_eval_g_jac_device_steps", "armour_prepare_steps", "armour_eval_g_jac_device_multi", "armour_desired_trajectory", "armour_robust_controller", "armour_check_feasible", "armour_get_torque_radius",
    "armour_get_link_generators", "armour_get_link_centers", "armour_get_pz", "armour_get_table_sizes",
    "armour_solve_options_default", "armour_solve", "armour_debug_qp", "armour_debug_qp_box", "armour_debug_pz_op",
    "armour_get_hyperplanes", "armour_get_build_ms", "armour_get_build_info", "armour_p2_kernel_name", "armour_debug_load_tables",
    "armour_get_plane_skip", "armour_get_prune_margin", "armour_batch_get_prune_margin", "armour_set_option", "armour_get_option", "armour_controller_set_kernel", "armour_device_memory", "armour_abi_max_factors", "armour_eval_violations_device", "armour_eval_violations", "armour_get_row_relevance", "armour_get_solver_rows",
    "armour_batch_partition", "armour_batch_create", "armour_batch_destroy", "armour_batch_set_option", "armour_batch_set_problems",
    "armour_batch_get_sizes", "armour_batch_get_bounds", "armour_batch_eval_g_jac", "armour_batch_eval_violations", "armour_batch_solve",
    "armour_batch_get_build_ms", "armour_batch_get_build_info",
]

_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  The PyTorch-ROCm wheel bundles its own libamdhip64.so (SONAME libamdhip64.so.7,
    the name libarmour_hip.so links against) but refers to it by file name, so if libarmour_hip.so pulled in
    /opt/rocm's copy first a later `import torch` would load a second runtime next to it, and device discovery in
    the second one fails ("no ROCm-capable device is detected").  When torch is installed, map its copy first (no
    torch import needed): the dynamic linker then binds libarmour_hip.so to it by SONAME and torch finds it already
    loaded.  Without torch nothing happens and /opt/rocm's runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load libarmour_hip.so (raises OSError with build instructions if it is absent)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(
            f"{LIB_PATH} not found: build it with `make -C armour_amd/csrc` (hipcc, gfx950) or "
            "`python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU path.")
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    if hasattr(L, "armour_abi_max_factors") and L.armour_abi_max_factors() != MAXF:
        raise RuntimeError(f"{LIB_PATH} is built for {L.armour_abi_max_factors()} factors, this process mirrors its structs for {MAXF} (ARMOUR_KEY128)")
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    vp = C.c_void_p
    L.armour_robot_kinova_gen3_no_gripper.argtypes = [C.POINTER(ArmourRobot)]
    L.armour_robot_kinova_gen3_no_gripper.restype = None
    L.armour_robot_kinova_gen3_gripper.argtypes = [C.POINTER(ArmourRobot)]
    L.armour_robot_fetch.argtypes = [C.POINTER(ArmourRobot)]
    L.armour_robot_fetch.restype = None
    L.armour_robot_fetch8.argtypes = [C.POINTER(ArmourRobot)]
    L.armour_robot_fetch8.restype = C.c_int
    L.armour_robot_kinova_gen3_gripper.restype = None
    L.armour_params_default.argtypes = [C.POINTER(ArmourParams), C.c_int32]
    L.armour_params_default.restype = None
    L.armour_create.argtypes = [C.POINTER(ArmourRobot), C.POINTER(ArmourParams), C.POINTER(ArmourLimits), C.c_int32, C.POINTER(vp)]
    L.armour_destroy.argtypes = [vp]
    L.armour_destroy.restype = None
    L.armour_last_error.restype = C.c_char_p
    L.armour_alloc_pinned.argtypes = [C.c_uint64, C.POINTER(vp)]
    L.armour_free_pinned.argtypes = [vp]
    L.armour_free_pinned.restype = None
    L.armour_set_problems.argtypes = [vp, C.c_int32, C.c_int32, dp, dp, dp, dp, dp]
    L.armour_set_problems_armtd.argtypes = [vp, C.c_int32, C.c_int32, dp, dp, dp, dp, dp, dp]
    L.armour_get_sizes.argtypes = [vp, ip, ip, ip]
    L.armour_get_bounds.argtypes = [vp, dp, dp, dp, dp]
    L.armour_eval_f.argtypes = [vp, dp, dp]
    L.armour_eval_grad_f.argtypes = [vp, dp, dp]
    L.armour_eval_g_jac.argtypes = [vp, dp, dp, dp]
    L.armour_eval_g_jac_device.argtypes = [vp, vp, vp, vp, vp]
    L.armour_eval_g_jac_device_steps.argtypes = [vp, vp, C.c_int32, vp, vp, vp]
    L.armour_prepare_steps.argtypes = [vp, vp, C.c_int32, vp, vp]
    L.armour_eval_g_jac_device_multi.argtypes = [vp, vp, C.c_int32, vp, vp, vp]
    L.armour_desired_trajectory.argtypes = [C.c_int32, dp, dp, dp, dp, C.c_double, dp, C.c_double, dp, dp, dp]
    L.armour_robust_controller.argtypes = [C.POINTER(ArmourRobot), C.c_double, dp, C.c_double, C.c_double, C.c_double, C.c_int32, dp, dp, dp, dp, dp, dp, dp, dp]
    L.armour_check_feasible.argtypes = [vp, dp, ip]
    L.armour_solve_options_default.argtypes = [C.POINTER(ArmourSolveOptions)]
    L.armour_solve_options_default.restype = None
    L.armour_solve.argtypes = [vp, C.POINTER(ArmourSolveOptions), C.POINTER(ArmourSolveResult)]
    L.armour_debug_qp.argtypes = [C.c_int32, dp, dp, C.c_int32, dp, dp, dp, dp, ip]
    L.armour_debug_qp_box.argtypes = [C.c_int32, dp, dp, C.c_int32, dp, dp, dp, dp, dp, C.c_int32, dp, ip, ip, dp]
    L.armour_debug_pz_op.argtypes = [vp, C.c_int32, C.c_int32, ip, ip, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(dp), dp, dp, dp, dp,
                                     C.c_int32, C.c_int32, C.POINTER(C.c_uint64), dp, dp]
    L.armour_get_torque_radius.argtypes = [vp, dp]
    L.armour_get_link_generators.argtypes = [vp, dp]
    L.armour_get_link_centers.argtypes = [vp, dp, dp]
    L.armour_get_pz.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, dp, C.POINTER(C.c_uint64), dp, C.c_int32]
    L.armour_get_table_sizes.argtypes = [vp, C.POINTER(C.c_int64)]
    L.armour_get_hyperplanes.argtypes = [vp, dp, dp, dp]
    L.armour_get_build_ms.argtypes = [vp, dp]
    L.armour_get_build_info.argtypes = [vp, ip]
    L.armour_p2_kernel_name.restype = C.c_char_p
    L.armour_debug_load_tables.argtypes = [vp, C.c_int32, C.c_int32, dp, dp, dp, dp, ip, dp, C.POINTER(C.c_uint64), dp,
                                           C.c_int32, ip, dp, C.POINTER(C.c_uint64), dp, C.c_int32, dp, dp, dp, dp]
    L.armour_get_plane_skip.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.armour_get_prune_margin.argtypes = [vp, dp]
    L.armour_batch_get_prune_margin.argtypes = [vp, dp]
    L.armour_set_option.argtypes = [vp, C.c_int32, C.c_double]
    L.armour_get_option.argtypes = [vp, C.c_int32, dp]
    L.armour_device_memory.argtypes = [C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.armour_controller_set_kernel.argtypes = [C.c_int32]
    L.armour_eval_violations_device.argtypes = [vp, vp, vp, vp]
    L.armour_eval_violations.argtypes = [vp, dp, C.POINTER(ArmourViolation)]
    L.armour_get_row_relevance.argtypes = [vp, C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    L.armour_get_solver_rows.argtypes = [vp, C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    L.armour_batch_partition.argtypes = [C.c_int32, C.c_int32, ip]
    L.armour_batch_create.argtypes = [C.POINTER(ArmourRobot), C.POINTER(ArmourParams), C.POINTER(ArmourLimits), ip, C.c_int32, C.POINTER(vp)]
    L.armour_batch_destroy.argtypes = [vp]
    L.armour_batch_destroy.restype = None
    L.armour_batch_set_option.argtypes = [vp, C.c_int32, C.c_double]
    L.armour_batch_set_problems.argtypes = [vp, C.c_int32, C.c_int32, dp, dp, dp, dp, dp]
    L.armour_batch_get_sizes.argtypes = [vp, ip, ip, ip, ip]
    L.armour_batch_get_bounds.argtypes = [vp, dp, dp, dp, dp]
    L.armour_batch_eval_g_jac.argtypes = [vp, dp, dp, dp]
    L.armour_batch_eval_violations.argtypes = [vp, dp, C.POINTER(ArmourViolation)]
    L.armour_batch_solve.argtypes = [vp, C.POINTER(ArmourSolveOptions), C.POINTER(ArmourSolveResult)]
    L.armour_batch_get_build_ms.argtypes = [vp, dp, dp]
    L.armour_batch_get_build_info.argtypes = [vp, ip]
    _lib = L
    return L


class ArmourError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"armour error {code}: {msg}")
        self.code = code


def check(rc):
    if rc < 0:
        raise ArmourError(rc, load().armour_last_error().decode())
    return rc
