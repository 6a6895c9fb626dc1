"""The reach-set PZs as CORA polyZonotope fields (armour_amd/cora.py): what the MATLAB side of the reference holds after
`remove_dependence_and_compress` (PZM/create_jrs_online.m:236-244) -- c, G, Grest, expMat, id with ids 1..n_q -- and its
`slice` (PZM/@polyZonotope_ROAHM/slice.m)."""
import numpy as np
import pytest

from helpers import PZ_TESTS_K, SAMPLE_PROBLEM


def test_roundtrip_and_slice_semantics():
    from armour_amd import cora
    rng = np.random.default_rng(0)
    exp = rng.integers(0, 4, (7, 12))
    exp[:, 0] = [1, 0, 0, 0, 0, 0, 0]
    keys = cora.keys_from_exponents(exp)
    assert keys[0] == 1 and np.array_equal(cora.exponents_from_keys(keys), exp)
    co, c, ind = rng.normal(size=(12, 3)), rng.normal(size=3), rng.uniform(0, 1, 3)
    pz = cora.to_polyzonotope(c, ind, keys, co)
    assert pz["c"].shape == (3, 1) and pz["G"].shape == (3, 12) and pz["Grest"].shape == (3, 3) and pz["expMat"].shape == (7, 12)
    assert np.array_equal(pz["id"].ravel(), np.arange(1, 8))
    c2, ind2, keys2, co2 = cora.from_polyzonotope(pz)
    assert np.array_equal(c2, c) and np.array_equal(ind2, ind) and np.array_equal(keys2, keys) and np.array_equal(co2, co)
    x = rng.uniform(-1, 1, 7)
    direct = c + sum(co[j] * np.prod(x ** exp[:, j]) for j in range(12))
    assert np.allclose(cora.slice_polyzonotope(pz, x), direct, rtol=0, atol=1e-14)
    with pytest.raises(ValueError):
        cora.exponents_from_keys(np.array([1 << 14], np.uint64))       # a non-k variable: not a reduced PZ
    empty = cora.to_polyzonotope(c, ind, np.zeros(0, np.uint64), np.zeros((0, 3)))
    assert np.array_equal(cora.slice_polyzonotope(empty, x), c)


def test_oracle_tables_slice_the_same_through_the_cora_form(tmp_path):
    """slice.m on the exported fields == the reference's own slice of the same PZ (RT/PZsparse.cu:404-470)."""
    from armour_amd import cora
    from oracle.cpu_oracle import Oracle
    from scipy.io import loadmat
    p = SAMPLE_PROBLEM
    o = Oracle(T=100).set_problem(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    links, torques = o.slice_links(PZ_TESTS_K), o.slice_torque(PZ_TESTS_K)
    out = {}
    for t in (0, 37, 99):
        for i in range(7):
            pl = cora.to_polyzonotope(*o.pz("link", i, t))
            assert np.allclose(cora.slice_polyzonotope(pl, PZ_TESTS_K), links[t, i], rtol=0, atol=1e-13)
            pt = cora.to_polyzonotope(*o.pz("torque", i, t))
            assert abs(cora.slice_polyzonotope(pt, PZ_TESTS_K)[0] - torques[t, i]) <= 1e-12
            out[f"link_{i}_{t}"] = pl
    cora.save_mat(tmp_path / "reach.mat", out)
    back = loadmat(tmp_path / "reach.mat", simplify_cells=True)["link_3_37"]
    assert np.array_equal(np.atleast_2d(back["G"]), out["link_3_37"]["G"]) and np.array_equal(back["expMat"], out["link_3_37"]["expMat"])


@pytest.mark.gpu
def test_device_tables_through_the_cora_form():
    from armour_amd import cora
    from armour_amd.planner import ArmourNLP
    p = SAMPLE_PROBLEM
    nlp = ArmourNLP(T=100).set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    cen = nlp.link_centers(PZ_TESTS_K)[0]
    for t in (0, 50, 99):
        for i in range(7):
            pz = nlp.polyzonotope("link", i, t)
            assert np.allclose(cora.slice_polyzonotope(pz, PZ_TESTS_K), cen[t, i], rtol=0, atol=1e-13)
            assert np.all(np.diag(pz["Grest"]) >= 0) and (pz["expMat"].size == 0 or pz["expMat"].max() <= 3)
    g = nlp.eval_g(PZ_TESTS_K)[0]
    for t in (0, 99):
        for j in range(7):
            pz = nlp.polyzonotope("torque", j, t)
            assert abs(cora.slice_polyzonotope(pz, PZ_TESTS_K)[0] - g[t * 7 + j]) <= 1e-11
