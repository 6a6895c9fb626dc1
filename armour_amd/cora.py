"""The reach-set tables in the form the MATLAB side of the reference holds them: CORA `polyZonotope` fields.

`polyZonotope_ROAHM(c, G, Grest, expMat, id)` (PZM/@polyZonotope_ROAHM/polyZonotope_ROAHM.m:6-24; PZM =
kinova_src/kinova_simulator_interfaces/polynomial_zonotope_matlab) is   c + sum_j G(:,j) prod_i x_id(i)^expMat(i,j) + Grest [-1,1]^M.
After `remove_dependence_and_compress` (PZM/create_jrs_online.m:236-244) the MATLAB planner keeps exactly what the device
tables keep: the k-dependent monomials with ids 1..n_q (`jrs_info.k_id = (1:n_q)'`, :225) and everything else folded into an
interval radius.  A device PZ (centre, independent radius, u64 keys, coefficients; include/armour_hip.h armour_get_pz)
maps onto those fields directly: the key packs the degree of k_i in bits 2i, 2i+1 (RT/PZsparse.h:8-21).

Nothing here runs on the GPU: it re-labels what `ArmourNLP.pz()` returns, so that MATLAB tooling written against the
CORA surface (slice.m, plotting, the fmincon constraint builder of KSI/uarmtd_planner.m:433-696) can consume the HIP
planner's reach sets -- `save_mat` writes the struct a MATLAB caller turns back into `polyZonotope_ROAHM(s.c, s.G,
s.Grest, s.expMat, s.id)`.
"""
import numpy as np


def exponents_from_keys(keys, n=7):
    """[n, M] degrees of k_1..k_n in each monomial (RT/PZsparse.h:8-21: 2 bits per trajectory parameter)."""
    keys = np.asarray(keys, dtype=np.uint64).reshape(-1)
    if np.any(keys >> np.uint64(2 * n)):
        raise ValueError("key holds variables other than the trajectory parameters: not a final (reduced) reach-set PZ")
    return np.stack([((keys >> np.uint64(2 * i)) & np.uint64(3)).astype(np.int64) for i in range(n)]) if keys.size else np.zeros((n, 0), np.int64)


def keys_from_exponents(exp_mat):
    exp_mat = np.asarray(exp_mat, dtype=np.int64)
    if np.any(exp_mat < 0) or np.any(exp_mat > 3):
        raise ValueError("degree outside the 2-bit field of the packed key")
    keys = np.zeros(exp_mat.shape[1], np.uint64)
    for i in range(exp_mat.shape[0]):
        keys |= exp_mat[i].astype(np.uint64) << np.uint64(2 * i)
    return keys


def to_polyzonotope(center, indep, keys, coeffs, n=7):
    """Device / oracle PZ -> dict(c [d,1], G [d,M], Grest [d,d], expMat [n,M], id [n,1]) with MATLAB's shapes.
    Grest is diag(independent radius): the interval hull the reference's `reduce` leaves (RT/PZsparse.cu:352-402)."""
    center = np.asarray(center, dtype=np.float64).reshape(-1)
    indep = np.asarray(indep, dtype=np.float64).reshape(-1)
    coeffs = np.asarray(coeffs, dtype=np.float64).reshape(-1, center.size)
    return dict(c=center.reshape(-1, 1).copy(), G=coeffs.T.copy(), Grest=np.diag(indep), expMat=exponents_from_keys(keys, n),
                id=np.arange(1, n + 1, dtype=np.int64).reshape(-1, 1))


def from_polyzonotope(pz):
    """Inverse of to_polyzonotope for polyZonotopes over ids 1..n with degrees <= 3 and an axis-aligned Grest:
    (center [d], indep [d], keys [M], coeffs [M,d])."""
    ids = np.asarray(pz["id"]).reshape(-1)
    if not np.array_equal(ids, np.arange(1, ids.size + 1)):
        raise ValueError("ids must be 1..n (jrs_info.k_id)")
    grest = np.atleast_2d(np.asarray(pz["Grest"], dtype=np.float64))
    indep = np.abs(grest).sum(axis=1) if grest.size else np.zeros(np.asarray(pz["c"]).size)
    return np.asarray(pz["c"], dtype=np.float64).reshape(-1), indep, keys_from_exponents(pz["expMat"]), np.asarray(pz["G"], dtype=np.float64).T.copy()


def slice_polyzonotope(pz, x):
    """PZM/@polyZonotope_ROAHM/slice.m:1-17: evaluate the dependent part at x (ids 1..max_id); Grest is ignored there."""
    x = np.asarray(x, dtype=np.float64).reshape(-1)[np.asarray(pz["id"]).reshape(-1) - 1]
    if pz["G"].size == 0:
        return pz["c"].reshape(-1).copy()
    monom = np.prod(x[:, None] ** pz["expMat"], axis=0)
    return pz["c"].reshape(-1) + (pz["G"] * monom).sum(axis=1)


def save_mat(path, pzs):
    """Write {name: polyzonotope dict} as MATLAB structs (scipy.io.savemat)."""
    from scipy.io import savemat
    savemat(path, {name: {k: (v.astype(np.float64) if k in ("expMat", "id") else v) for k, v in pz.items()} for name, pz in pzs.items()})
