"""Device-side row relevance -- the pruned constraint list of the reference's MATLAB path (KSI/uarmtd_planner.m:577-583, 628-690: a constraint
is kept only if it can be violated), which its C++ path lacks (RT/NLPclass.cu:272-396 evaluates every row every time).  relevance.hip."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _nlp(B, O, T=100, seed=40):
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch
    bp = random_batch(seed, B, O)
    return ArmourNLP(T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"]), bp


@pytest.mark.parametrize("B,O", [(1, 20), (6, 50), (3, 0)])
def test_the_mask_contains_every_row_that_is_ever_violated(B, O):
    """Soundness: over 1 000 sampled k (uniform in the box, plus its corners' neighbourhood) no row the mask drops is ever > 0 (collision rows)
    or outside its bounds (torque rows); and the mask is not trivial: most collision rows are dropped, the 4n limit rows are always kept."""
    from armour_amd.worlds import random_k
    nlp, _ = _nlp(B, O)
    rel, cnt, ms = nlp.row_relevance()
    n, T, J = nlp.n, nlp.T, nlp.J
    Q = J * T * O
    assert rel.shape == (B, nlp.m) and rel[:, n * T + Q:].all()
    assert (rel[:, n * T:n * T + Q].sum(1) == cnt).all()
    if O:
        assert 0 < cnt.max() and cnt.mean() < 0.25 * Q, (cnt, Q)
    xl, xu, gl, gu = nlp.get_bounds_info()
    ever = np.zeros((B, nlp.m), bool)
    rng = np.random.default_rng(5)
    for it in range(1000 // 8):
        ks = random_k(it, B * 8).reshape(8, B, n)
        ks[0] = np.sign(ks[0]) * (1.0 - 0.05 * rng.random((B, n)))   # near the corners of the box
        for k in ks:
            g = nlp.eval_g(k)
            ever |= (g > gu) | (g < gl)
    assert not (ever & ~rel).any(), np.argwhere(ever & ~rel)[:5]
    print(f"B={B} O={O}: relevant collision rows {cnt.tolist()} of {Q}; rows ever violated {ever.sum(1).tolist()}; test {ms:.3f} ms")
    nlp.close()


@pytest.mark.parametrize("B,O", [(1, 20), (8, 50), (2, 0)])
def test_culled_violation_records_equal_the_full_ones(B, O):
    """ARMOUR_OPT_CULL_ROWS = 1: armour_eval_violations over the relevant rows only -- L1 violation, rows violated, rows outside the slacks and the
    verdict are the full evaluation's bit for bit at every tested k (an unlisted row adds exactly 0); `worst` and its row whenever any row is
    violated.  Includes k = 0, feasible-looking and badly infeasible points."""
    from armour_amd import _lib
    from armour_amd.worlds import random_k
    nlp, bp = _nlp(B, O, seed=90)
    from armour_amd.planner import ArmourNLP
    culled = ArmourNLP(T=100).set_option(_lib.OPT_CULL_ROWS, 1).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    some_violation = False
    for it in range(12):
        k = np.zeros((B, nlp.n)) if it == 0 else random_k(100 + it, B) * (1.0 if it % 2 else 0.2)
        full, cul = nlp.eval_violations(k), culled.eval_violations(k)
        for a, c in zip(full, cul):
            assert a["l1_violation"] == c["l1_violation"] and a["n_violated"] == c["n_violated"] and a["n_outside_slack"] == c["n_outside_slack"] and a["feasible"] == c["feasible"]
            if a["n_violated"] > 0:
                some_violation = True
                assert a["worst"] == c["worst"] and a["worst_row"] == c["worst_row"]
    assert some_violation
    nlp.close(); culled.close()


@pytest.mark.parametrize("B,O", [(1, 20), (6, 50), (3, 0)])
def test_the_solver_mask_contains_every_row_that_is_ever_a_qp_candidate(B, O):
    """The second mask (armour_get_solver_rows): armour_solve takes row i into a QP when g_i + 2 |J_i|_1 > u_i or g_i - 2 |J_i|_1 < l_i
    (solver_common.h); a row the mask drops must fail that test at every k.  Sampled over 600 points (box, corners' neighbourhood); the mask
    contains the relevance mask, keeps the limit rows, counts its torque rows, and is not trivial."""
    from armour_amd.worlds import random_k
    nlp, _ = _nlp(B, O)
    mask, cnt, tq, ms = nlp.solver_rows()
    rel = nlp.row_relevance()[0]
    n, T, J = nlp.n, nlp.T, nlp.J
    Q, nT = J * T * O, n * T
    assert mask.shape == (B, nlp.m) and mask[:, nT + Q:].all() and not (rel & ~mask).any()
    assert (mask[:, nT:nT + Q].sum(1) == cnt).all()
    assert (mask[:, :nT].sum(1) == tq).all(), (mask[:, :nT].sum(1), tq)
    if O:
        assert cnt.mean() < 0.4 * Q, (cnt, Q)
    xl, xu, gl, gu = nlp.get_bounds_info()
    ever = np.zeros((B, nlp.m), bool)
    rng = np.random.default_rng(6)
    for it in range(600 // 8):
        ks = random_k(1000 + it, B * 8).reshape(8, B, n)
        ks[0] = np.sign(ks[0]) * (1.0 - 0.05 * rng.random((B, n)))
        for k in ks:
            g, jac = nlp.eval_g_jac(k)
            l1 = np.abs(jac).sum(axis=2)
            ever |= ((gu < 1e18) & (g + 2.0 * l1 > gu)) | ((gl > -1e18) & (g - 2.0 * l1 < gl))
    assert not (ever & ~mask).any(), np.argwhere(ever & ~mask)[:5]
    print(f"B={B} O={O}: solver rows {cnt.tolist()} of {Q} collision rows, torque rows {tq.tolist()} of {nT}; rows ever candidates (sampled) {ever[:, nT:nT + Q].sum(1).tolist()}; {ms:.3f} ms")
    nlp.close()


def test_a_point_outside_the_box_is_not_answered_from_the_lists():
    """the mask is a statement about k in [-1, 1]^n.  With ARMOUR_OPT_CULL_ROWS = 1 the host entry takes every row when a component of k
    lies outside the box -- its records equal the full evaluation's bit for bit there too -- and the device entry marks such a problem's record
    (feasible = -1, worst_row = -2) while the in-box problems of the same launch keep the full evaluation's records."""
    import ctypes as C
    import torch
    from armour_amd import _lib
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_k
    B, O = 4, 20
    nlp, bp = _nlp(B, O, seed=91)
    culled = ArmourNLP(T=100).set_option(_lib.OPT_CULL_ROWS, 1).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    k = random_k(300, B)
    k[1, 2] = 1.7; k[3, 0] = -2.5          # problems 1 and 3 leave the box
    full, cul = nlp.eval_violations(k), culled.eval_violations(k)
    for a, c in zip(full, cul):
        assert a == c, (a, c)
    assert any(a["n_violated"] > 0 for a in full)
    # the device entry
    d_k = torch.tensor(k, device="cuda", dtype=torch.float64)
    d_out = torch.zeros(B * C.sizeof(_lib.ArmourViolation), device="cuda", dtype=torch.uint8)
    culled.eval_violations_device(d_k.data_ptr(), d_out.data_ptr())
    torch.cuda.synchronize()
    rec = (_lib.ArmourViolation * B).from_buffer_copy(d_out.cpu().numpy().tobytes())
    kin = k.copy(); kin[1, 2] = 0.7; kin[3, 0] = -0.5
    full_in = nlp.eval_violations(kin)
    for b in range(B):
        if b in (1, 3):
            assert rec[b].feasible == -1 and rec[b].worst_row == -2
        else:
            assert rec[b].feasible == full[b]["feasible"] and rec[b].l1_violation == full[b]["l1_violation"] and rec[b].n_violated == full[b]["n_violated"]
            assert full_in[b]["l1_violation"] == full[b]["l1_violation"]      # (problems are independent)
    nlp.close(); culled.close()


@pytest.mark.parametrize("world", ["random, 12 boxes", "reference scene"])
def test_the_device_list_contains_the_list_of_the_matlab_rule(world):
    """the device keeps a collision row unless ONE half-space of its 36-plane table separates the link's hull over k from
    the obstacle -- a sufficient test -- where KSI/uarmtd_planner.m:577-583 tests the zonotope itself: the obstacle buffered by EVERY generator of the
    link's occupancy (sliceable and not), turned into its full half-space form (PZM/utility/polytope_PH.m: the normals of all generator pairs), must
    contain the occupancy's centre.  That rule, restated in numpy (armour_amd/cora_mode.py: polytope_PH) and applied to the DEVICE's own reach sets,
    is the oracle here: every (link, time step, obstacle) it keeps the device keeps too (the device's normals are a subset of its normals and the bound
    over k is the same sum of |A . generator|), and the device's list is not much longer."""
    from armour_amd.cora_mode import polytope_PH
    from armour_amd.planner import ArmourNLP
    if world.startswith("random"):
        from armour_amd.worlds import random_problem
        p = random_problem(77, 12)
    else:
        from armour_amd.scenes import reference_worlds
        p = dict(reference_worlds())["scene_037_004"]
    T = 40
    nlp = ArmourNLP(T=T).set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    rel, cnt, _ = nlp.row_relevance()
    n, J = nlp.n, nlp.J
    obs = np.asarray(p["obstacles"], dtype=float).reshape(-1, 12)
    O = len(obs)
    gens = nlp.link_generators()[0]   # [T][J][3][6]: the link box's three generators | the independent radius per axis
    keep_matlab = np.zeros((J, T, O), bool)
    for l in range(J):
        for t in range(T):
            cen, _, _, co = nlp.pz("link", l, t)
            G = np.hstack([np.asarray(co, dtype=float).reshape(-1, 3).T, gens[t, l]])   # sliceable generators (one per k-monomial) | the rest
            for o in range(O):
                Zo = obs[o].reshape(4, 3).T
                A, b = polytope_PH(np.hstack([Zo, G]))
                keep_matlab[l, t, o] = bool(np.all(A @ cen - b <= 0))
    keep_device = rel[0, n * T:n * T + J * T * O].reshape(J, T, O)
    assert not (keep_matlab & ~keep_device).any(), np.argwhere(keep_matlab & ~keep_device)[:5]
    nm, nd = int(keep_matlab.sum()), int(keep_device.sum())
    print(f"{world}: the MATLAB rule keeps {nm} of {J * T * O} collision rows, the device {nd}")
    assert nd == int(cnt[0]) and nd <= max(3 * nm, nm + 0.02 * J * T * O), (nm, nd)
    nlp.close()
