"""GPU parity AT THE SIZES BASELINE.json quotes its numbers on (the other parity tests use sizes the oracle finishes in a
second or two):

    configs[2]   Kinova 7-DOF, 50 obstacles, batch of 128 random worlds on one MI355X, T = 100
    configs[3]   1024 worlds over 8 GPUs = 128 worlds per GPU at 20 obstacles, T = 100 (the per-GPU shard)

At B = 128 the reach-set build runs its two-pass shape (2048-entry sort buffers first, the overflowing items again with
the full buffers) at four waves per CU, and the fused kernel runs the <DFC, LL, 6 slots, EX> instantiation (d recomputed
from the obstacle centres, compact link x link normals, fixed-count loads) at full occupancy -- paths that the small
cases reach only in part.  Checked here, all through the C ABI:

  * tables (key sets, coefficients, centres, radii, torque radius, link generators) and g / jac of six sampled worlds
    against the live CPU oracle, tolerances of tests/test_p1_parity.py;
  * the same six worlds bit-equal to single-problem handles (one-pass build, 3-wave blocks, d read from the table);
  * the one-point device entry, the graph of back-to-back steps and the multi-point entry agree bit for bit with the
    host entry for ALL 128 worlds;
  * size-independent properties over all 128 worlds: outputs finite, collision rows of a duplicated obstacle equal,
    bounds consistent with the torque radius."""
import numpy as np
import pytest

from test_pz_ops import margin_agrees

pytestmark = pytest.mark.gpu

C_TOL, R_TOL, G_TOL, J_TOL = 1e-11, 1e-10, 1e-9, 1e-8
SAMPLE = (0, 17, 42, 63, 100, 127)


def _oracle(T, bp, b):
    from oracle.cpu_oracle import Oracle
    return Oracle(T=T).set_problem(bp["q0"][b], bp["qd0"][b], bp["qdd0"][b], bp["q_des"][b], bp["obstacles"][b])


def _tables_equal_oracle(nlp, o, b):
    assert o.min_margin() > 1e-9
    for which, cnt in (("link", o.J), ("torque", o.n)):
        for i in range(cnt):
            for t in range(0, o.T, 3):   # every third time step of every link / joint
                c, ind, keys, co = o.pz(which, i, t)
                c2, ind2, keys2, co2 = nlp.pz(which, i, t, b=b)
                assert np.array_equal(keys, keys2), (b, which, i, t)
                if len(keys):
                    assert np.abs(co - co2).max() <= C_TOL
                assert np.abs(c - c2).max() <= C_TOL and np.abs(ind - ind2).max() <= C_TOL


@pytest.mark.parametrize("O,first_seed", [(50, 1000), (20, 3000)], ids=["configs2_O50_B128", "configs3_shard_O20_B128"])
def test_batch_of_128_worlds(O, first_seed):
    import torch
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch, random_k
    T, B, P = 100, 128, 2
    bp = random_batch(first_seed, B, O)
    bp["obstacles"][5, 1] = bp["obstacles"][5, 0]     # world 5: obstacle 1 duplicates obstacle 0
    nlp = ArmourNLP(T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    n, m = nlp.n, nlp.m
    assert m == n * T + nlp.J * T * O + 4 * n
    ks = np.stack([random_k(500 + s, B) for s in range(P)])                 # [P, B, n]
    g, jac = nlp.eval_g_jac(ks[0])                                          # host entry (staging copies)
    assert np.isfinite(g).all() and np.isfinite(jac).all()
    tr, gens = nlp.torque_radius(), nlp.link_generators()
    assert np.isfinite(tr).all() and (tr > 0).all() and np.isfinite(gens).all()
    # the prune margin of EVERY world, from the device (armour_get_prune_margin): no simplify() verdict of the batch came within 1e-9 of flipping, which
    # is what the tolerances below are conditional on -- until round 6 only the six sampled worlds could be asked, through the oracle
    margins = nlp.prune_margin()
    assert margins.shape == (B,) and (margins > 1e-9).all(), (margins.min(), int(margins.argmin()))

    # ---- device entries, all 128 worlds: one-point launch, graph of steps, multi-point launch == host entry
    dev = torch.device("cuda:0")
    st = torch.cuda.Stream()
    d_k = torch.from_numpy(ks).to(dev)
    d_g = torch.full((B, m), float("nan"), dtype=torch.float64, device=dev)
    d_j = torch.full((B, m, n), float("nan"), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    nlp.eval_g_jac_device(d_k[0].data_ptr(), d_g.data_ptr(), d_j.data_ptr(), st.cuda_stream)
    st.synchronize()
    assert np.array_equal(d_g.cpu().numpy(), g) and np.array_equal(d_j.cpu().numpy(), jac)
    g1, j1 = nlp.eval_g_jac(ks[1])
    nlp.prepare_steps(d_k.data_ptr(), P, d_g.data_ptr(), d_j.data_ptr())
    nlp.eval_g_jac_device_steps(d_k.data_ptr(), P, d_g.data_ptr(), d_j.data_ptr(), st.cuda_stream)
    st.synchronize()
    assert np.array_equal(d_g.cpu().numpy(), g1) and np.array_equal(d_j.cpu().numpy(), j1)
    d_gm = torch.full((P, B, m), float("nan"), dtype=torch.float64, device=dev)
    d_jm = torch.full((P, B, m, n), float("nan"), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()   # the fills run on torch's stream, the launch below on `st`
    nlp.eval_g_jac_device_multi(d_k.data_ptr(), P, d_gm.data_ptr(), d_jm.data_ptr(), st.cuda_stream)
    st.synchronize()
    assert np.array_equal(d_gm[0].cpu().numpy(), g) and np.array_equal(d_gm[1].cpu().numpy(), g1)
    assert np.array_equal(d_jm[0].cpu().numpy(), jac) and np.array_equal(d_jm[1].cpu().numpy(), j1)
    del d_gm, d_jm

    # ---- properties over all worlds
    row0 = n * T
    col = g[5, row0:row0 + nlp.J * T * O].reshape(nlp.J * T, O)
    assert np.array_equal(col[:, 0], col[:, 1])                              # duplicated obstacle -> identical rows
    _, _, gl, gu = nlp.get_bounds_info()
    tq_lim = np.array(nlp.robot.torque_limits[:n])
    assert np.array_equal(gu[:, :row0].reshape(B, T, n), tq_lim[None, None, :] - tr.transpose(0, 2, 1))
    assert (gl[:, row0:row0 + nlp.J * T * O] == -1e19).all() and (gu[:, row0:row0 + nlp.J * T * O] == 0).all()

    # ---- six sampled worlds: the live oracle, and single-problem handles bit for bit
    for b in SAMPLE:
        o = _oracle(T, bp, b)
        _tables_equal_oracle(nlp, o, b)
        assert margin_agrees(margins[b], o.min_margin()), (b, margins[b], o.min_margin())
        assert np.abs(tr[b] - o.torque_radius()).max() <= R_TOL
        assert np.abs(gens[b] - o.link_generators()).max() <= C_TOL
        for s, (gs, js) in enumerate(((g, jac), (g1, j1))):
            gr, jr = o.eval_g_jac(ks[s, b])
            assert np.abs(gs[b] - gr).max() <= G_TOL and np.abs(js[b] - jr).max() <= J_TOL, (b, s)
        # A single-problem handle builds its reach sets step by step (one wave per time step), a batch of this size
        # time-vectorised (one wave per 64 time steps, pz_tv.h).  The two add the same coefficient terms in the same
        # order -- keys, coefficients and centres are equal bit for bit -- and the pruned-radius sums in a different
        # one: radii, and through them g, agree to rounding.
        one = ArmourNLP(T=T).set_parameters(bp["q0"][b], bp["qd0"][b], bp["qdd0"][b], bp["q_des"][b], bp["obstacles"][b])
        go, jo = one.eval_g_jac(ks[0, b])
        assert np.abs(g[b] - go[0]).max() <= 1e-12 * max(1.0, np.abs(go[0]).max()) and np.abs(jac[b] - jo[0]).max() <= 1e-12 * max(1.0, np.abs(jo[0]).max()), b
        assert np.abs(tr[b] - one.torque_radius()[0]).max() <= 1e-12 and np.abs(gens[b] - one.link_generators()[0]).max() <= 1e-12
        for which, cnt in (("link", nlp.J), ("torque", n)):
            for i in range(cnt):
                for t in (0, 33, 64, 99):
                    c1, i1, k1, co1 = nlp.pz(which, i, t, b=b)
                    c2, i2, k2, co2 = one.pz(which, i, t)
                    assert np.array_equal(k1, k2) and np.array_equal(co1, co2) and np.array_equal(c1, c2), (b, which, i, t)
                    assert np.abs(i1 - i2).max() <= 1e-12
        if b == SAMPLE[0]:   # the half-space table of one world in the reference's layout against the oracle's
            A2, d2, dl2 = one.hyperplanes()
            A, d, dl = o.hyperplanes()
            assert np.abs(A - A2[0]).max() <= C_TOL and np.abs(d - d2[0]).max() <= C_TOL and np.abs(dl - dl2[0]).max() <= C_TOL
        one.close()
    # ---- a world's tables do not depend on its batch mates: the same worlds in reverse order, bit for bit
    rev = ArmourNLP(T=T).set_parameters(bp["q0"][::-1].copy(), bp["qd0"][::-1].copy(), bp["qdd0"][::-1].copy(), bp["q_des"][::-1].copy(), bp["obstacles"][::-1].copy())
    gr_, jr_ = rev.eval_g_jac(ks[0][::-1].copy())
    assert np.array_equal(gr_[::-1], g) and np.array_equal(jr_[::-1], jac)
    assert np.array_equal(rev.torque_radius()[::-1], tr) and np.array_equal(rev.link_generators()[::-1], gens)
    rev.close()
    nlp.close()


def test_whole_planning_iterations_at_batch_128():
    """armour_solve over a configs[3] shard: every world's verdict equals the host re-check of its final g
    (finalize_solution, RT/NLPclass.cu:422-538) and sampled worlds reproduce their single-problem solves."""
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch
    T, B, O = 100, 128, 20
    bp = random_batch(3000, B, O)
    nlp = ArmourNLP(T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    sols = nlp.solve(max_iterations=8)
    kopt = np.stack([s["k_opt"] for s in sols])
    assert np.isfinite(kopt).all() and np.abs(kopt).max() <= 1.0 + 1e-9, np.abs(kopt).max()   # x_l <= k <= x_u up to the QP's active-set residual
    g = nlp.eval_g(kopt)
    feas = nlp.finalize_solution(g)
    assert [bool(s["feasible"]) for s in sols] == [bool(f) for f in feas]
    for b in (0, 64, 127):
        one = ArmourNLP(T=T).set_parameters(bp["q0"][b], bp["qd0"][b], bp["qdd0"][b], bp["q_des"][b], bp["obstacles"][b])
        s1 = one.solve(max_iterations=8)[0]
        # (the batch's reach sets are built time-vectorised, the single problem's step by step: radii equal to rounding)
        assert np.abs(s1["k_opt"] - sols[b]["k_opt"]).max() <= 1e-9 and s1["feasible"] == sols[b]["feasible"], b
        one.close()
    nlp.close()


def test_batch_with_more_groups_than_compute_units():
    """A batch of 150 worlds has 300 groups of 50 time steps for 256 CUs: the four-wave blocks of the time-vectorised build loop over the
    groups (round 3; before, such batches took the one-wave-per-group shape, still reachable through ARMOUR_P1_TV_WAVES=1).  Sampled
    worlds -- first and second round of the loop -- against single-problem handles on the per-step kernel: keys, coefficients and
    centres bit for bit, radii to the stated 1e-12; and one against the live oracle."""
    from armour_amd import _lib
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch, random_k
    T, B, O = 100, 150, 4
    bp = random_batch(8100, B, O)
    nlp = ArmourNLP(T=T).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
    ks = random_k(17, B)
    g, jac = nlp.eval_g_jac(ks)
    for b in (0, 77, 127, 128, 149):
        one = ArmourNLP(T=T).set_option(_lib.OPT_P1_BUILD, 1).set_parameters(bp["q0"][b], bp["qd0"][b], bp["qdd0"][b], bp["q_des"][b], bp["obstacles"][b])
        for which, cnt in (("link", nlp.J), ("torque", nlp.n)):
            for i in range(cnt):
                for t in range(0, T, 9):
                    c1, r1, k1, co1 = one.pz(which, i, t)
                    c2, r2, k2, co2 = nlp.pz(which, i, t, b=b)
                    assert np.array_equal(k1, k2) and np.array_equal(co1, co2) and np.array_equal(c1, c2), (b, which, i, t)
                    assert np.abs(r1 - r2).max() <= 1e-12
        assert np.abs(one.torque_radius()[0] - nlp.torque_radius()[b]).max() <= 1e-12
        g1, j1 = one.eval_g_jac(ks[b])
        assert np.abs(g1[0] - g[b]).max() <= 1e-11 and np.abs(j1[0] - jac[b]).max() <= 1e-11
        one.close()
    o = _oracle(T, bp, 149)
    _tables_equal_oracle(nlp, o, 149)
    nlp.close()
