"""The MEX gateway (armour_amd/mex/armour_hip_mex.cpp) compiled and RUN without MATLAB.

The reference's MEX precedent is kinova_robust_controllers_mex/kinova_controller.cpp:19-84, built by `mex` with
CXXFLAGS -std=c++14 -O2 (compile.m:1-10).  MATLAB is absent here, so the gateway is compiled with g++ against
tests/stubs/mex.h -- a functional test double of the MEX C API -- and tests/stubs/mex_harness.cpp plays MATLAB: it
builds prhs[], calls mexFunction and hands plhs[] back.  CPU: the gateway compiles with the reference's own flags,
links against libarmour_hip.so, rejects bad calls through mexErrMsgTxt, and its stateless 'traj' command equals the
ctypes path.  GPU: every command against armour_amd.planner.ArmourNLP (the ctypes binding of the same C ABI)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def mexlib(tmp_path_factory):
    out = tmp_path_factory.mktemp("mex") / "armour_hip_mex_test.so"
    libdir = os.path.join(ROOT, "armour_amd", "lib")
    if not os.path.exists(os.path.join(libdir, "libarmour_hip.so")):
        pytest.skip("libarmour_hip.so not built")
    # -std=c++14 -O2: MEXC/compile.m:1-10 (the stub header itself needs nothing newer)
    cmd = ["g++", "-std=c++14", "-O2", "-Wall", "-Werror", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "tests", "stubs"), "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "armour_amd", "mex", "armour_hip_mex.cpp"), os.path.join(ROOT, "tests", "stubs", "mex_harness.cpp"),
           "-L" + libdir, "-larmour_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    from armour_amd import _lib
    _lib.load()   # maps the HIP runtime the way every other test does, before the gateway pulls libarmour_hip.so in
    L = C.CDLL(str(out))
    L.mexh_call.restype = C.c_int
    L.mexh_error.restype = C.c_char_p
    yield L
    L.mexh_exit()


class Mex:
    """armour_hip_mex(...) as MATLAB would call it: strings and real double matrices in, a list of arrays out."""

    def __init__(self, L):
        self.L = L

    def __call__(self, nlhs, *args):
        n = len(args)
        strs = (C.c_char_p * n)()
        data = (C.POINTER(C.c_double) * n)()
        rows, cols = (C.c_int * n)(), (C.c_int * n)()
        keep = []
        for i, a in enumerate(args):
            if isinstance(a, str):
                strs[i] = a.encode()
            else:
                arr = np.asfortranarray(np.atleast_2d(np.asarray(a, dtype=np.float64)))
                if np.asarray(a).ndim <= 1:
                    arr = np.asfortranarray(np.asarray(a, dtype=np.float64).reshape(-1, 1))   # vectors are columns
                keep.append(arr)
                strs[i] = None
                data[i] = arr.ctypes.data_as(C.POINTER(C.c_double))
                rows[i], cols[i] = arr.shape
        rc = self.L.mexh_call(nlhs, n, strs, data, rows, cols)
        if rc < 0:
            raise RuntimeError(self.L.mexh_error().decode())
        outs = []
        for i in range(rc):
            r, c, lg = C.c_int(), C.c_int(), C.c_int()
            if self.L.mexh_out_dims(i, C.byref(r), C.byref(c), C.byref(lg)) != 0:
                outs.append(None)
                continue
            buf = np.zeros(r.value * c.value)
            self.L.mexh_out_copy(i, buf.ctypes.data_as(C.POINTER(C.c_double)))
            outs.append(buf.reshape((r.value, c.value), order="F"))
        return outs


def test_gateway_builds_and_checks_its_arguments(mexlib):
    mex = Mex(mexlib)
    with pytest.raises(RuntimeError, match="usage"):
        mex(0)
    with pytest.raises(RuntimeError, match="usage"):
        mex(0, np.zeros(3))                       # first argument must be the command string
    with pytest.raises(RuntimeError, match="create"):
        mex(2, "eval", np.zeros(7))               # no handle yet
    with pytest.raises(RuntimeError, match="create"):
        mex(3, "traj", np.zeros(7), np.zeros(7), np.zeros(7), np.zeros(7), 0.5)   # k_range / duration come from 'create'
    with pytest.raises(RuntimeError, match="time steps"):
        mex(0, "create")
    assert mexlib.mexh_is_locked() == 0


def test_create_fails_loudly_without_a_device(mexlib):
    """The library has no CPU path, and the gateway says so through mexErrMsgTxt (MATLAB would show the message)."""
    from armour_amd import _lib
    if _lib.load().armour_device_available():
        pytest.skip("a HIP device is visible")
    with pytest.raises(RuntimeError, match="no HIP device"):
        Mex(mexlib)(0, "create", 100)
    assert mexlib.mexh_is_locked() == 0


@pytest.mark.gpu
def test_every_command_against_the_ctypes_binding(mexlib, sample_problem):
    from armour_amd.planner import ArmourNLP, desired_trajectory
    from helpers import PZ_TESTS_K
    mex = Mex(mexlib)
    sp = sample_problem
    T = 100
    mex(0, "create", T)
    assert mexlib.mexh_is_locked() == 1
    Z = sp["obstacles"].T                                        # 12 x nObs, columns = Z(:) of each obstacle
    margin = mex(1, "set_problem", sp["q0"], sp["qd0"], sp["qdd0"], sp["q_des"], Z)[0]    # (optional output: the build's prune margin)
    nlp = ArmourNLP(T=T).set_parameters(sp["q0"], sp["qd0"], sp["qdd0"], sp["q_des"], sp["obstacles"])
    n, m = nlp.n, nlp.m
    assert margin.shape == (1, 1) and margin[0, 0] == nlp.prune_margin()[0] and 1e-9 < margin[0, 0] < 0.4143
    assert mex(1, "prune_margin")[0][0, 0] == margin[0, 0]
    k = PZ_TESTS_K
    g, jac = mex(2, "eval", k)
    g_ref, jac_ref = nlp.eval_g_jac(k)
    assert g.shape == (m, 1) and jac.shape == (n, m)
    assert np.array_equal(g[:, 0], g_ref[0]) and np.array_equal(jac.T, jac_ref[0])
    xl, xu, gl, gu = mex(4, "bounds")
    rxl, rxu, rgl, rgu = nlp.get_bounds_info()
    assert np.array_equal(xl[:, 0], rxl) and np.array_equal(xu[:, 0], rxu) and np.array_equal(gl[:, 0], rgl[0]) and np.array_equal(gu[:, 0], rgu[0])
    f, gf = mex(2, "cost", k)
    assert f[0, 0] == nlp.eval_f(k)[0] and np.array_equal(gf[:, 0], nlp.eval_grad_f(k)[0])
    # the fmincon nonlcon shape (KSI/uarmtd_planner.m:776-796)
    h, heq, grad_h, grad_heq = mex(4, "constraints", k)
    rh, rheq, rgh, rgheq = nlp.eval_constraint(k)
    assert np.array_equal(h[:, 0], rh) and np.array_equal(grad_h, rgh) and heq.size == 0 and grad_heq.shape == (n, 0)
    # ... and with the pruned list (KSI/uarmtd_planner.m:577-583,628-690: only rows that can be violated for some k): the kept rows of the full list
    rel = mex(1, "relevance")[0][:, 0].astype(bool)
    rrel, _, _ = nlp.row_relevance()
    assert np.array_equal(rel, rrel[0]) and rel[-4 * n:].all() and rel.sum() < 0.5 * m
    hp, _, ghp, _ = mex(4, "constraints", k, "pruned")
    two = rgl[0] > -1e18                                  # rows with a lower bound come a second time, behind the m upper-bound rows
    keep = np.concatenate([rel, rel[two]])
    assert np.array_equal(hp[:, 0], rh[keep]) and np.array_equal(ghp, rgh[:, keep])
    # whole NLP
    k_opt, feas, info = mex(3, "solve")
    sol = nlp.solve()[0]
    assert np.array_equal(k_opt[:, 0], sol["k_opt"]) and bool(feas[0, 0]) == sol["feasible"]
    assert info[1, 0] == sol["iterations"] and info[2, 0] == sol["evaluations"] and info[3, 0] == sol["status"]
    # reach sets as CORA polyZonotope fields, 1-based indices
    for which, i, t in (("link", 7, 60), ("torque", 3, 17)):
        c, G, Grest, expMat, pid = mex(5, "pz", which, i, t)
        pz = nlp.polyzonotope(which, i - 1, t - 1)
        assert np.array_equal(c[:, 0], pz["c"].ravel()) and np.array_equal(G, pz["G"]) and np.array_equal(Grest, pz["Grest"])
        assert np.array_equal(expMat, pz["expMat"]) and np.array_equal(pid[:, 0], pz["id"].ravel())
    # desired trajectory of the plan
    q, qd, qdd = mex(3, "traj", sp["q0"], sp["qd0"], sp["qdd0"], k, 0.37)
    rq, rqd, rqdd = desired_trajectory(sp["q0"], sp["qd0"], sp["qdd0"], k, 0.37)
    assert np.array_equal(q[:, 0], rq) and np.array_equal(qd[:, 0], rqd) and np.array_equal(qdd[:, 0], rqdd)
    # errors come back through mexErrMsgTxt with the library's message, and the handle survives them
    with pytest.raises(RuntimeError, match="k"):
        mex(2, "eval", np.zeros(5))
    with pytest.raises(RuntimeError, match="unknown"):
        mex(0, "nonsense")
    g2, = mex(1, "eval", k)
    assert np.array_equal(g2, g)
    mex(0, "destroy")
    assert mexlib.mexh_is_locked() == 0
    with pytest.raises(RuntimeError, match="create"):
        mex(1, "eval", k)
    nlp.close()


@pytest.mark.gpu
def test_batch_commands_drive_device_slots_from_one_thread(mexlib):
    """The multi-device batch through the gateway (armour_batch_*): two device slots on the one GPU of this box, every result
    against ArmourBatchNLP / a single handle over the same problems."""
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch, random_k
    mex = Mex(mexlib)
    T, B, O = 100, 5, 6
    bp = random_batch(77, B, O)
    with pytest.raises(RuntimeError, match="batch_create"):
        mex(1, "batch_violations", np.zeros((7, B)))
    mex(0, "batch_create", T, np.array([0.0, 0.0]))
    assert mexlib.mexh_is_locked() == 1
    Z = bp["obstacles"].reshape(B, O * 12).T                      # (12 nObs) x B
    mex(0, "batch_set_problems", bp["q0"].T, bp["qd0"].T, bp["qdd0"].T, bp["q_des"].T, Z)
    one = ArmourNLP(T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    K = random_k(9, B)
    V, = mex(1, "batch_violations", K.T)
    ref = one.eval_violations(K)
    for b in range(B):
        assert (V[0, b], V[1, b], int(V[2, b]) - 1, int(V[3, b]), int(V[4, b]), bool(V[5, b])) == \
               (ref[b]["l1_violation"], ref[b]["worst"], ref[b]["worst_row"], ref[b]["n_violated"], ref[b]["n_outside_slack"], ref[b]["feasible"])
    G, JAC = mex(2, "batch_eval", K.T)
    g_ref, jac_ref = one.eval_g_jac(K)
    assert np.array_equal(G.T, g_ref) and np.array_equal(JAC.T.reshape(B, one.m, one.n), jac_ref)
    K_opt, feas, info = mex(3, "batch_solve")
    sol = one.solve()
    for b in range(B):
        assert np.array_equal(K_opt[:, b], sol[b]["k_opt"]) and bool(feas[0, b]) == sol[b]["feasible"] and info[3, b] == sol[b]["status"]
    # the single-handle 'violations' command next to it
    mex(0, "create", T)
    mex(0, "set_problem", bp["q0"][2], bp["qd0"][2], bp["qdd0"][2], bp["q_des"][2], bp["obstacles"][2].T)
    v, = mex(1, "violations", K[2])
    assert (v[0, 0], v[1, 0], int(v[2, 0]) - 1, bool(v[5, 0])) == (ref[2]["l1_violation"], ref[2]["worst"], ref[2]["worst_row"], ref[2]["feasible"])
    mex(0, "destroy")
    assert mexlib.mexh_is_locked() == 1                           # the batch object still holds the gateway
    mex(0, "batch_destroy")
    assert mexlib.mexh_is_locked() == 0
    one.close()
