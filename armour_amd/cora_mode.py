"""CORA-semantics polynomial-zonotope mode (second half): the MATLAB path of the reference, restated.

The reference has two implementations of its reach sets.  The C++ one (RT/, what libarmour_hip.so replaces) keeps sparse
monomial lists and prunes by a coefficient threshold.  The MATLAB one -- `uarmtd_planner` with `use_cuda = false`
(KSI/uarmtd_planner.m:433-696) on the classes of PZM/ = kinova_src/kinova_simulator_interfaces/polynomial_zonotope_matlab,
which derive from CORA 2021's `polyZonotope` -- keeps an EXPONENT MATRIX per set, never prunes, and bounds the representation
size by Girard order reduction (`zono_order` 40, then 3 before the half-space construction).  The two agree only as
over-approximations of the same sets, so this mode is a second arithmetic for CROSS-VALIDATION: both must
contain the true swept volume, and their collision constraints must agree in sign wherever the truth is unambiguous.

What is restated (file:line of the MATLAB source each piece follows):
  PZ / MatPZ              PZM/@polyZonotope_ROAHM/polyZonotope_ROAHM.m, PZM/@matPolyZonotope_ROAHM/matPolyZonotope_ROAHM.m
                          (fields c, G, Grest, expMat, id); CORA's mergeExpMatrix / removeRedundantExponents semantics
  plus, times, mtimes     @polyZonotope_ROAHM/plus.m:8-30, times.m:6-100, @matPolyZonotope_ROAHM/mtimes.m:1-232
  cos, sin                @polyZonotope_ROAHM/cos.m:1-53, sin.m (Taylor polynomial + interval Lagrange remainder)
  reduce                  @polyZonotope_ROAHM/reduce.m:40-108, @matPolyZonotope_ROAHM/reduce.m (Girard: keep the K longest
                          generators, box the rest -- CORA zonotope/reduce 'girard' with order 1)
  zonotope / interval     CORA polyZonotope/zonotope.m (all-even exponents: half into the centre, half as generator)
  create_jrs_online       PZM/create_jrs_online.m:1-254, traj_type 'bernstein', taylor_degree 1 (KSI/uarmtd_planner.m:41)
  pzfk, forward occupancy simulator/dynamics/pzfk.m; KSI/uarmtd_planner.m:461-469
  obstacle constraints    KSI/uarmtd_planner.m:572-609,698-709 with PZM/utility/polytope_PH.m

It is host arithmetic (numpy): the MATLAB path is interpreter code, there is no device counterpart to be faithful to, and its
job here is to check the HIP path, not to be fast.  tests/test_cora_mode.py: the algebra against sampling, enclosure of the
true forward occupancy, and (on the GPU) sign agreement of the two paths' collision constraints.
"""
import itertools
from math import comb

import numpy as np


# ------------------------------------------------------------------------------------------------ sets
class PZ:
    """c + sum_j G[:, j] prod_i x_{id[i]}^{E[i, j]} + Grest [-1, 1]^Q   (CORA polyZonotope fields)"""

    def __init__(self, c, G=None, Grest=None, E=None, ids=None):
        self.c = np.atleast_1d(np.asarray(c, dtype=float)).reshape(-1)
        d = self.c.size
        self.G = np.zeros((d, 0)) if G is None or np.size(G) == 0 else np.asarray(G, dtype=float).reshape(d, -1)
        self.Grest = np.zeros((d, 0)) if Grest is None or np.size(Grest) == 0 else np.asarray(Grest, dtype=float).reshape(d, -1)
        if self.G.shape[1]:
            E = np.asarray(E, dtype=np.int64).reshape(-1, self.G.shape[1])
            ids = np.asarray(ids, dtype=np.int64).reshape(-1)
            self.E, self.G = _remove_redundant_exponents(E, self.G)   # the CORA constructor merges equal exponent vectors
            self.id = ids
        else:
            self.E, self.id = np.zeros((0, 0), np.int64), np.zeros(0, np.int64)

    @property
    def dim(self):
        return self.c.size

    def __add__(self, o):
        return plus(self, o)

    __radd__ = __add__

    def __sub__(self, o):
        return plus(self, times(-1.0, o) if isinstance(o, PZ) else -np.asarray(o, dtype=float))

    def __mul__(self, o):
        return times(self, o)

    __rmul__ = __mul__


class MatPZ:
    """matrix-valued: C [r, c], G [r, c, P], Grest [r, c, Q] (PZM/@matPolyZonotope_ROAHM)"""

    def __init__(self, C, G=None, Grest=None, E=None, ids=None):
        self.C = np.asarray(C, dtype=float)
        r, c = self.C.shape
        self.G = np.zeros((r, c, 0)) if G is None or np.size(G) == 0 else np.asarray(G, dtype=float).reshape(r, c, -1)
        self.Grest = np.zeros((r, c, 0)) if Grest is None or np.size(Grest) == 0 else np.asarray(Grest, dtype=float).reshape(r, c, -1)
        if self.G.shape[2]:
            E = np.asarray(E, dtype=np.int64).reshape(-1, self.G.shape[2])
            E2, G2 = _remove_redundant_exponents(E, self.G.reshape(r * c, -1))
            self.E, self.G = E2, G2.reshape(r, c, -1)
            self.id = np.asarray(ids, dtype=np.int64).reshape(-1)
        else:
            self.E, self.id = np.zeros((0, 0), np.int64), np.zeros(0, np.int64)

    def __matmul__(self, o):
        return mtimes(self, o)


def _remove_redundant_exponents(E, G):
    """CORA removeRedundantExponents: drop all-zero generators, add the generators of identical exponent vectors."""
    keep = np.any(G != 0, axis=0)
    E, G = E[:, keep], G[:, keep]
    if E.shape[1] <= 1:
        return E, G
    _, first, inv = np.unique(E.T, axis=0, return_index=True, return_inverse=True)
    inv = inv.reshape(-1)
    order = np.argsort(first)                     # keep the order of first occurrence
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    Gn = np.zeros((G.shape[0], order.size))
    np.add.at(Gn.T, rank[inv], G.T)
    return E[:, np.sort(first)], Gn


def _merge_exp_matrix(id1, id2, E1, E2):
    """CORA mergeExpMatrix: a common id vector (id1, then the new ids of id2) and both exponent matrices over it."""
    if id1.size == id2.size and np.array_equal(id1, id2):
        return id1, E1, E2
    new = [i for i in id2 if i not in set(id1.tolist())]
    ids = np.concatenate([id1, np.asarray(new, dtype=np.int64)]) if new else id1.copy()
    pos = {v: k for k, v in enumerate(ids.tolist())}
    A = np.zeros((ids.size, E1.shape[1]), np.int64)
    if E1.size:
        A[:id1.size] = E1
    B = np.zeros((ids.size, E2.shape[1]), np.int64)
    for r, v in enumerate(id2.tolist()):
        if E2.size:
            B[pos[v]] = E2[r]
    return ids, A, B


# ------------------------------------------------------------------------------------------------ arithmetic
def plus(a, b):
    """@polyZonotope_ROAHM/plus.m:8-30 (exact addition: equal monomials merge)"""
    if not isinstance(a, PZ):
        a, b = b, a
    if not isinstance(b, PZ):
        return PZ(a.c + np.asarray(b, dtype=float), a.G, a.Grest, a.E, a.id)
    if a.G.shape[1] == 0 and b.G.shape[1] == 0:
        return PZ(a.c + b.c, None, np.hstack([a.Grest, b.Grest]))
    ids, E1, E2 = _merge_exp_matrix(a.id, b.id, a.E, b.E)
    return PZ(a.c + b.c, np.hstack([a.G, b.G]), np.hstack([a.Grest, b.Grest]), np.hstack([E1, E2]), ids)


def times(a, b):
    """@polyZonotope_ROAHM/times.m:6-100: product of two 1-D sets, or a number times a set"""
    if not isinstance(a, PZ):
        a, b = b, a
    if not isinstance(b, PZ):
        f = float(b)
        return PZ(f * a.c, f * a.G, f * a.Grest, a.E, a.id)
    assert a.dim == 1 and b.dim == 1, "both sets must be one-dimensional (times.m:11-13)"
    ids, E1, E2 = _merge_exp_matrix(a.id, b.id, a.E, b.E)
    G, E, R = [], [], []
    if b.G.shape[1]:
        G.append(a.c[0] * b.G); E.append(E2)
    if a.G.shape[1]:
        G.append(a.G * b.c[0]); E.append(E1)
    if a.G.shape[1] and b.G.shape[1]:
        G.append(np.outer(a.G[0], b.G[0]).reshape(1, -1))
        E.append(np.hstack([E1[:, [i]] + E2 for i in range(E1.shape[1])]))
    if b.Grest.shape[1]:
        R.append(a.c[0] * b.Grest)
    if a.Grest.shape[1]:
        R.append(a.Grest * b.c[0])
    if a.Grest.shape[1] and b.Grest.shape[1]:
        R.append(np.outer(a.Grest[0], b.Grest[0]).reshape(1, -1))
    if a.G.shape[1] and b.Grest.shape[1]:
        R.append(np.outer(a.G[0], b.Grest[0]).reshape(1, -1))
    if a.Grest.shape[1] and b.G.shape[1]:
        R.append(np.outer(a.Grest[0], b.G[0]).reshape(1, -1))
    G = np.hstack(G) if G else None
    return PZ(a.c * b.c, G, np.hstack(R) if R else None, np.hstack(E) if E else None, ids)


def power(a, n):
    """@polyZonotope_ROAHM/power.m"""
    if n == 0:
        return 1.0
    out = a
    for _ in range(2, n + 1):
        out = times(out, a)
    return out


def mtimes(A, b):
    """@matPolyZonotope_ROAHM/mtimes.m: MatPZ * (numeric vector | PZ | MatPZ); numeric matrix * PZ"""
    if not isinstance(A, MatPZ):
        M = np.asarray(A, dtype=float)
        return PZ(M @ b.c, M @ b.G, M @ b.Grest, b.E, b.id)
    pm = lambda X, Y: np.einsum("ijp,jk->ikp", X, Y)
    if not isinstance(b, (PZ, MatPZ)):
        v = np.asarray(b, dtype=float).reshape(-1)
        G = np.einsum("ijp,j->ip", A.G, v) if A.G.shape[2] else None
        R = np.einsum("ijp,j->ip", A.Grest, v) if A.Grest.shape[2] else None
        return PZ(A.C @ v, G, R, A.E, A.id)
    ids, E1, E2 = _merge_exp_matrix(A.id, b.id, A.E, b.E)
    if isinstance(b, PZ):
        G, E, R = [], [], []
        if b.G.shape[1]:
            G.append(A.C @ b.G); E.append(E2)
        if A.G.shape[2]:
            G.append(np.einsum("ijp,j->ip", A.G, b.c)); E.append(E1)
        if A.G.shape[2] and b.G.shape[1]:
            G.append(np.einsum("ijp,jq->ipq", A.G, b.G).reshape(A.C.shape[0], -1))          # order: for each generator of A, all of b
            E.append(np.hstack([E1[:, [i]] + E2 for i in range(E1.shape[1])]))
        if b.Grest.shape[1]:
            R.append(A.C @ b.Grest)
        if A.Grest.shape[2]:
            R.append(np.einsum("ijp,j->ip", A.Grest, b.c))
        if A.Grest.shape[2] and b.Grest.shape[1]:
            R.append(np.einsum("ijp,jq->ipq", A.Grest, b.Grest).reshape(A.C.shape[0], -1))
        if A.G.shape[2] and b.Grest.shape[1]:
            R.append(np.einsum("ijp,jq->ipq", A.G, b.Grest).reshape(A.C.shape[0], -1))
        if A.Grest.shape[2] and b.G.shape[1]:
            R.append(np.einsum("ijp,jq->ipq", A.Grest, b.G).reshape(A.C.shape[0], -1))
        return PZ(A.C @ b.c, np.hstack(G) if G else None, np.hstack(R) if R else None, np.hstack(E) if E else None, ids)
    # MatPZ * MatPZ
    G, E, R = [], [], []
    cat = lambda L: np.concatenate(L, axis=2) if L else None
    pp = lambda X, Y: np.einsum("ijp,jkq->ikpq", X, Y).reshape(X.shape[0], Y.shape[1], -1)
    if b.G.shape[2]:
        G.append(np.einsum("ij,jkq->ikq", A.C, b.G)); E.append(E2)
    if A.G.shape[2]:
        G.append(pm(A.G, b.C)); E.append(E1)
    if A.G.shape[2] and b.G.shape[2]:
        G.append(pp(A.G, b.G)); E.append(np.hstack([E1[:, [i]] + E2 for i in range(E1.shape[1])]))
    if b.Grest.shape[2]:
        R.append(np.einsum("ij,jkq->ikq", A.C, b.Grest))
    if A.Grest.shape[2]:
        R.append(pm(A.Grest, b.C))
    if A.Grest.shape[2] and b.Grest.shape[2]:
        R.append(pp(A.Grest, b.Grest))
    if A.G.shape[2] and b.Grest.shape[2]:
        R.append(pp(A.G, b.Grest))
    if A.Grest.shape[2] and b.G.shape[2]:
        R.append(pp(A.Grest, b.G))
    return MatPZ(A.C @ b.C, cat(G), cat(R), np.hstack(E) if E else None, ids)


def transpose(A):
    return MatPZ(A.C.T, A.G.transpose(1, 0, 2), A.Grest.transpose(1, 0, 2), A.E, A.id)


# ------------------------------------------------------------------------------------------------ enclosures
def to_zonotope(p):
    """CORA polyZonotope/zonotope.m: (centre, generators); a monomial with all-even exponents ranges over [0, 1]"""
    if p.G.shape[1] == 0:
        return p.c.copy(), p.Grest.copy()
    even = np.all(p.E % 2 == 0, axis=0)
    c = p.c + 0.5 * p.G[:, even].sum(axis=1)
    return c, np.hstack([p.G[:, ~even], 0.5 * p.G[:, even], p.Grest])


def interval(p):
    """CORA interval(polyZonotope) = interval(zonotope(pZ)): (lower, upper)"""
    c, G = to_zonotope(p)
    r = np.abs(G).sum(axis=1)
    return c - r, c + r


def _imul(a, b):
    v = [a[0] * b[0], a[0] * b[1], a[1] * b[0], a[1] * b[1]]
    return min(v), max(v)


def _icos(a):
    lo, hi = a
    if hi - lo >= 2 * np.pi:
        return -1.0, 1.0
    vals = [np.cos(lo), np.cos(hi)]
    k0 = np.ceil(lo / np.pi)
    for k in np.arange(k0, np.floor(hi / np.pi) + 1):
        vals.append(np.cos(k * np.pi))
    return min(vals), max(vals)


def _isin(a):
    return _icos((a[0] - np.pi / 2, a[1] - np.pi / 2))


def _sgn_cs(n):
    return 1.0 if n % 4 in (0, 3) else -1.0


def _sgn_sn(n):
    return 1.0 if n % 4 in (0, 1) else -1.0


def _trig(p, order, which):
    """@polyZonotope_ROAHM/cos.m:1-53 and sin.m: Taylor polynomial of degree `order` about the centre plus the Lagrange
    remainder as an interval, folded into Grest"""
    assert p.dim == 1
    c0 = p.c[0]
    cs, sn = np.cos(c0), np.sin(c0)
    out = PZ(cs if which == "cos" else sn)
    nb = p - c0
    T, factor = 1.0, 1.0
    for i in range(1, order + 1):
        factor *= i
        T = times(T, nb) if isinstance(T, PZ) else nb
        if which == "cos":
            coef = _sgn_cs(i) * (cs if i % 2 == 0 else sn) / factor
        else:
            coef = _sgn_sn(i) * (sn if i % 2 == 0 else cs) / factor
        out = out + times(coef, T)
    rl, ru = interval(nb)
    rem = (rl[0], ru[0])
    pl, pu = interval(times(T, nb))
    rem_pow = (pl[0], pu[0])
    arg = _imul((0.0, 1.0), rem)
    arg = (c0 + arg[0], c0 + arg[1])
    if which == "cos":
        J0 = _isin(arg) if (order + 1) % 2 == 1 else _icos(arg)
        neg = order % 4 in (0, 1)
    else:
        J0 = _isin(arg) if (order + 1) % 2 == 0 else _icos(arg)
        neg = order % 4 in (1, 2)
    J = (-J0[1], -J0[0]) if neg else J0
    r = _imul(rem_pow, J)
    s = 1.0 / (factor * (order + 1))
    r = (s * r[0], s * r[1])
    out = PZ(out.c + 0.5 * (r[0] + r[1]), out.G, np.hstack([out.Grest, [[0.5 * (r[1] - r[0])]]]), out.E, out.id)
    return PZ(out.c, out.G, np.abs(out.Grest).sum(axis=1, keepdims=True), out.E, out.id)


def cos(p, order=6):
    return _trig(p, order, "cos")


def sin(p, order=6):
    return _trig(p, order, "sin")


def _girard_split(Gall, E, P, K):
    """indices (dependent, independent) of the generators that reduce() removes: all but the K longest, all-even ones halved"""
    G = Gall.copy()
    if P:
        G[:, np.where(np.all(E % 2 == 0, axis=0))[0]] *= 0.5   # (the first P columns are the dependent generators)
    length = (G ** 2).sum(axis=0)
    order = np.argsort(-length, kind="stable")
    rem = order[K:]
    return rem[rem < P], rem[rem >= P] - P


def reduce(p, order):
    """@polyZonotope_ROAHM/reduce.m:40-108 with option 'girard'"""
    N, P, Q = p.dim, p.G.shape[1], p.Grest.shape[1]
    K = N * order - N
    c, G, Grest, E = p.c, p.G, p.Grest, p.E
    if P + Q > N * order and K >= 0:
        dep, ind = _girard_split(np.hstack([G, Grest]), E, P, K)
        zc, zG = to_zonotope(PZ(np.zeros(N), G[:, dep], Grest[:, ind], E[:, dep], p.id))
        box = np.diag(np.abs(zG).sum(axis=1))      # zonotope/reduce 'girard', order 1: the interval hull
        c = c + zc
        G, E = np.delete(G, dep, axis=1), np.delete(E, dep, axis=1)
        Grest = np.hstack([np.delete(Grest, ind, axis=1), box])
    ids = p.id
    if E.size:
        used = E.sum(axis=1) > 0
        E, ids = E[used], ids[used]
    if N == 1:
        Grest = np.abs(Grest).sum(axis=1, keepdims=True)
    return PZ(c, G, Grest, E, ids)


def reduce_mat(A, order):
    """@matPolyZonotope_ROAHM/reduce.m"""
    N, N2 = A.C.shape
    P, Q = A.G.shape[2], A.Grest.shape[2]
    K = N * N2 * order - N * N2
    C, G, Grest, E = A.C, A.G, A.Grest, A.E
    if P + Q > N * order and K >= 0:
        flat = lambda X: X.transpose(1, 0, 2).reshape(N * N2, -1)   # column-major vec of every page (polyZonotope_ROAHM(matPZ))
        Gv, Rv = flat(G), flat(Grest)
        dep, ind = _girard_split(np.hstack([Gv, Rv]), E, P, K)
        zc, zG = to_zonotope(PZ(np.zeros(N * N2), Gv[:, dep], Rv[:, ind], E[:, dep], A.id))
        box = np.diag(np.abs(zG).sum(axis=1))
        unflat = lambda V: V.reshape(N2, N, -1).transpose(1, 0, 2)
        C = C + zc.reshape(N2, N).T
        G, E = np.delete(G, dep, axis=2), np.delete(E, dep, axis=1)
        Grest = np.concatenate([np.delete(Grest, ind, axis=2), unflat(box)], axis=2)
    ids = A.id
    if E.size:
        used = E.sum(axis=1) > 0
        E, ids = E[used], ids[used]
    return MatPZ(C, G, Grest, E, ids)


def remove_dependence(p, max_id):
    """@polyZonotope_ROAHM/remove_dependence.m: monomials that involve an id > max_id become independent generators"""
    hi = p.id > max_id
    gen = np.any(p.E[hi] != 0, axis=0) if p.E.size else np.zeros(p.G.shape[1], bool)
    return PZ(p.c, p.G[:, ~gen], np.hstack([p.Grest, p.G[:, gen]]), p.E[~hi][:, ~gen], p.id[~hi])


def slice_pz(p, x):
    """@polyZonotope_ROAHM/slice.m: evaluate the dependent part at x (x indexed by id, 1-based ids)"""
    if p.G.shape[1] == 0:
        return p.c.copy()
    xv = np.asarray(x, dtype=float)[p.id - 1]
    return p.c + (p.G * np.prod(xv[:, None] ** p.E, axis=0)).sum(axis=1)


def grad_slice(p, x, n):
    """@polyZonotope_ROAHM/grad.m + slice: d slice / d x_i for i = 1..n, as an [n, dim] array"""
    out = np.zeros((n, p.dim))
    if p.G.shape[1] == 0:
        return out
    xv = np.asarray(x, dtype=float)[p.id - 1]
    for r, i in enumerate(p.id.tolist()):
        if i > n:
            continue
        e = p.E[r]
        Ed = p.E.copy()
        Ed[r] = np.maximum(e - 1, 0)
        mon = np.prod(xv[:, None] ** Ed, axis=0) * e
        out[i - 1] = (p.G * mon).sum(axis=1)
    return out


# ------------------------------------------------------------------------------------------------ the planner's pipeline
def _rotations_from_q(q, axis, deg):
    """PZM/utility/get_pz_rotations_from_q.m: Rodrigues' formula on the cos / sin sets of q"""
    cq, sq = cos(q, deg), sin(q, deg)
    # exactCartProd(cos_q, sin_q): one 2-D set over the merged ids
    ids, E1, E2 = _merge_exp_matrix(cq.id, sq.id, cq.E, sq.E)
    G = np.vstack([np.hstack([cq.G, np.zeros((1, sq.G.shape[1]))]), np.hstack([np.zeros((1, cq.G.shape[1])), sq.G])])
    R = np.vstack([np.hstack([cq.Grest, np.zeros((1, sq.Grest.shape[1]))]), np.hstack([np.zeros((1, cq.Grest.shape[1])), sq.Grest])])
    cs = PZ(np.array([cq.c[0], sq.c[0]]), G, R, np.hstack([E1, E2]), ids)
    e = np.asarray(axis, dtype=float) / np.linalg.norm(axis)
    U = np.array([[0, -e[2], e[1]], [e[2], 0, -e[0]], [-e[1], e[0], 0]])
    U2 = U @ U
    C = np.eye(3) + cs.c[1] * U + (1 - cs.c[0]) * U2
    gen = lambda M: np.stack([M[1, j] * U - M[0, j] * U2 for j in range(M.shape[1])], axis=2) if M.shape[1] else None
    return MatPZ(C, gen(cs.G), gen(cs.Grest), cs.E, cs.id)


def _compress(p, k_id):
    """create_jrs_online.m:236-244 remove_dependence_and_compress: keep the generator that depends on k alone"""
    row = p.id == k_id
    other = ~row
    slc = (p.E[row] != 0).all(axis=0) & (p.E[other] == 0).all(axis=0) if p.E.size else np.zeros(0, bool)
    if slc.sum() > 1:
        raise ValueError("There should only be one fully-k-sliceable generator")
    rest = np.abs(p.G[:, ~slc]).sum() + np.abs(p.Grest).sum()
    return PZ(p.c, p.G[:, slc], [[rest]], p.E[row][:, slc], [k_id])


def _compress_mat(A, k_id):
    row = A.id == k_id
    other = ~row
    slc = (A.E[row] != 0).all(axis=0) & (A.E[other] == 0).all(axis=0) if A.E.size else np.zeros(0, bool)
    return MatPZ(A.C, A.G[:, :, slc], np.concatenate([A.G[:, :, ~slc], A.Grest], axis=2), A.E[row][:, slc], [k_id])


def create_jrs_online(q, dq, ddq, joint_axes, taylor_degree=1, ultimate_bound=0.0191, k_r=10.0, k_range=np.pi / 36, n_t=100,
                      add_ultimate_bound=True, full=False, time_indices=None):
    """PZM/create_jrs_online.m, traj_type 'bernstein': per time interval and joint the position set Q (with tracking error)
    and its rotation matrix set R, in terms of the trajectory parameters k_1..k_n (ids 1..n) alone.
    full=True returns a dict with everything the dynamics half of the planner reads (:165-178): Q, Qd (velocity with the velocity
    error E_v), Qd_a (auxiliary velocity: desired + K_r E_p), Qdd_a (auxiliary acceleration: desired + K_r E_v), R, R_t.
    time_indices: build only these intervals (the others are None) -- the intervals are independent of each other."""
    q, dq, ddq = (np.asarray(v, dtype=float).reshape(-1) for v in (q, dq, ddq))
    n_q = q.size
    dt = 1.0 / n_t
    next_id = n_q + 1
    K = [PZ(0.0, [[1.0]], None, [[1]], [j + 1]) for j in range(n_q)]
    t_ids = list(range(next_id, next_id + n_t)); next_id += n_t
    Tm = [PZ(dt * i + dt / 2, [[dt / 2]], None, [[1]], [t_ids[i]]) for i in range(n_t)]
    E_p, E_v = [], []
    for j in range(n_q):
        E_p.append(PZ(0.0, [[ultimate_bound / k_r]], None, [[1]], [next_id]) if add_ultimate_bound else 0.0)
        E_v.append(PZ(0.0, [[2 * ultimate_bound]], None, [[1]], [next_id + 1]) if add_ultimate_bound else 0.0)
        next_id += 2          # (e1, e2: position and velocity error each get an id, create_jrs_online.m:125-133)
    alpha = []
    for j in range(n_q):
        q1 = q[j] + times(k_range, K[j])
        beta = [q[j], q[j] + dq[j] / 5, ddq[j] / 20 + 2 * dq[j] / 5 + q[j], q1, q1, q1]   # match_deg5_bernstein_coefficients.m, T = 1
        al = []
        for i in range(6):
            acc = 0.0
            for jj in range(i + 1):
                term = ((-1) ** (i - jj) * comb(5, i) * comb(i, jj)) * beta[jj] if not isinstance(beta[jj], PZ) \
                    else times(float((-1) ** (i - jj) * comb(5, i) * comb(i, jj)), beta[jj])
                acc = acc + term
            al.append(acc)
        alpha.append(al)
    sc = lambda f, x: times(float(f), x) if isinstance(x, PZ) else f * x
    mul = lambda a, b: times(a, b) if isinstance(a, PZ) or isinstance(b, PZ) else a * b
    Q, R = [], []
    out = dict(Q=Q, Qd=[], Qd_a=[], Qdd_a=[], R=R, R_t=[])
    for i in range(n_t):
        if time_indices is not None and i not in time_indices:
            for v in out.values():
                v.append(None)
            continue
        Qi, Ri, Qdi, Qdai, Qddai, Rti = [], [], [], [], [], []
        for j in range(n_q):
            Qd_, Qv_, Qa_ = 0.0, 0.0, 0.0
            for k in range(6):
                Qd_ = Qd_ + mul(alpha[j][k], power(Tm[i], k))
                if full and k > 0:
                    Qv_ = Qv_ + mul(sc(k, alpha[j][k]), power(Tm[i], k - 1))             # :156-158
                if full and k > 1:
                    Qa_ = Qa_ + mul(sc(k * (k - 1), alpha[j][k]), power(Tm[i], k - 2))   # :159-161
            Qe = Qd_ + E_p[j]
            Rj = _rotations_from_q(Qe, joint_axes[:, j], taylor_degree)
            Qi.append(_compress(Qe, j + 1))
            Ri.append(_compress_mat(Rj, j + 1))
            if full:
                Qdi.append(_compress(Qv_ + E_v[j], j + 1))                                # Qd   = Qd_des + E_v          (:167)
                Qdai.append(_compress(Qv_ + sc(k_r, E_p[j]), j + 1))                      # Qd_a = Qd_des + k_r E_p      (:168)
                Qddai.append(_compress(Qa_ + sc(k_r, E_v[j]), j + 1))                     # Qdd_a = Qdd_des + k_r E_v    (:169)
                Rti.append(_compress_mat(transpose(Rj), j + 1))                           # get_pz_rotations_from_q.m: every page transposed
        Q.append(Qi); R.append(Ri)
        out["Qd"].append(Qdi); out["Qd_a"].append(Qdai); out["Qdd_a"].append(Qddai); out["R_t"].append(Rti)
    return out if full else (Q, R)


def pzfk(R_in, T0, P, zono_order=40):
    """simulator/dynamics/pzfk.m: world rotation and position sets of every joint frame (all joints revolute here)"""
    R_out, p_out = [], []
    for i in range(len(R_in)):
        Ri = mtimes(MatPZ(T0[i]), R_in[i])
        if i == 0:
            p_out.append(PZ(P[:, 0]))
            R_out.append(mtimes(MatPZ(np.eye(3)), Ri))
        else:
            p = p_out[i - 1] + mtimes(R_out[i - 1], P[:, i])
            p_out.append(reduce(p, zono_order))
            R_out.append(reduce_mat(mtimes(R_out[i - 1], Ri), zono_order))
    return R_out, p_out


def forward_occupancy(q0, qd0, qdd0, robot, n_t=100, zono_order=40, **jrs_kw):
    """KSI/uarmtd_planner.m:446-469: FO[i][j] for time interval i and link j.  `robot`: an ArmourRobot-like object (trans, rots,
    axes, link boxes); the link sets are boxes of independent generators (PZM/create_pz_bounding_boxes.m)."""
    from .robot_geometry import joint_frames
    T0, P, axes, centers, half = joint_frames(robot)
    n = len(axes)
    Q, R = create_jrs_online(q0, qd0, qdd0, axes.T, n_t=n_t, **jrs_kw)
    FO = []
    for i in range(n_t):
        R_w, p_w = pzfk(R[i], T0, P, zono_order)
        row = []
        for j in range(n):
            link = PZ(centers[j], None, np.diag(half[j]))
            fo = mtimes(R_w[j], link) + p_w[j]
            fo = reduce(fo, zono_order)
            row.append(remove_dependence(fo, n))
        FO.append(row)
    return FO


def polytope_PH(Z):
    """PZM/utility/polytope_PH.m for 3-D zonotopes Z = [c, G]: half-space form A x <= b"""
    c, G = Z[:, 0], Z[:, 1:]
    G = G[:, np.linalg.norm(G, axis=0) >= 1e-6]
    combs = np.array(list(itertools.combinations(range(G.shape[1]), 2)))
    a, b = G[:, combs[:, 0]], G[:, combs[:, 1]]
    C = np.cross(a.T, b.T)
    with np.errstate(invalid="ignore", divide="ignore"):
        C = C / np.linalg.norm(C, axis=1, keepdims=True)
    C = C[~np.isnan(C).any(axis=1)]
    delta = np.abs(C @ G).sum(axis=1)
    d = C @ c
    return np.vstack([C, -C]), np.concatenate([d + delta, -d + delta])


def obstacle_constraints(FO, obstacles, n_q):
    """KSI/uarmtd_planner.m:572-609: one constraint per (time interval, link, obstacle) whose buffered obstacle contains the
    centre of the link's occupancy.  obstacles: [O, 12] column-major Z = [c g1 g2 g3].  Returns a list of
    (i, j, o, A, b, FO_reduced); evaluate with eval_obstacle_constraint."""
    out = []
    for i, row in enumerate(FO):
        for j, fo in enumerate(row):
            fo3 = None
            for o, ob in enumerate(np.asarray(obstacles, dtype=float).reshape(-1, 12)):
                Zo = ob.reshape(4, 3).T
                A, b = polytope_PH(np.hstack([Zo, fo.G, fo.Grest]))
                if not np.all(A @ fo.c - b <= 0):
                    continue
                if fo3 is None:
                    fo3 = reduce(fo, 3)          # "reduce FO so that polytope_PH has fewer directions to consider" (:587)
                A, b = polytope_PH(np.hstack([Zo, fo3.Grest]))
                out.append((i, j, o, A, b, PZ(fo3.c, fo3.G, None, fo3.E, fo3.id)))
    return out


def eval_obstacle_constraint(con, k, n_q):
    """KSI/uarmtd_planner.m:698-709: h = -max(A * slice(FO, k) - b) and its gradient (feasible: h <= 0)"""
    _, _, _, A, b, fo = con
    v = A @ slice_pz(fo, k) - b
    m = int(np.argmax(v))
    return -v[m], -(grad_slice(fo, k, n_q) @ A[m])


# ------------------------------------------------------------------------------------------------ the dynamics half
# KSI/uarmtd_planner.m:471-559 (input constraints) and :562-576,622-690 (joint limits, pruning) on PZM/utility/poly_zonotope_rnea.m.
def _vec(p):
    return p if isinstance(p, PZ) else PZ(np.asarray(p, dtype=float).reshape(-1))


def scalar_times_axis(p, z):
    """a 1-D set times a numeric column: `joint_vel{j} * z(:, i)` (poly_zonotope_rnea.m:92; CORA mtimes with a numeric matrix)"""
    z = np.asarray(z, dtype=float).reshape(-1, 1)
    if not isinstance(p, PZ):
        return PZ(float(p) * z[:, 0])
    return PZ(p.c[0] * z[:, 0], z @ p.G if p.G.shape[1] else None, z @ p.Grest if p.Grest.shape[1] else None, p.E, p.id)


def _skew(z):
    return np.array([[0.0, -z[2], z[1]], [z[2], 0.0, -z[0]], [-z[1], z[0], 0.0]])


def cross(a, b):
    """@polyZonotope_ROAHM/cross.m: the skew matrix (set) of a, times b; either operand may be a numeric 3-vector"""
    if not isinstance(a, PZ):
        M = _skew(np.asarray(a, dtype=float).reshape(-1))
        return mtimes(M, b) if isinstance(b, PZ) else PZ(M @ np.asarray(b, dtype=float).reshape(-1))
    G = np.stack([_skew(a.G[:, j]) for j in range(a.G.shape[1])], axis=2) if a.G.shape[1] else None
    R = np.stack([_skew(a.Grest[:, j]) for j in range(a.Grest.shape[1])], axis=2) if a.Grest.shape[1] else None
    A = MatPZ(_skew(a.c), G, R, a.E, a.id)
    return mtimes(A, b if isinstance(b, PZ) else np.asarray(b, dtype=float).reshape(-1))


def inertial_params(robot, uncertain, zono_order=40):
    """urdfs/urdf_utils/get_inertial_params.m:114-196, set_type 'polynomial_zonotope', track_inertial_generators false: per link the
    mass as a 3 x 3 matrix set m I_3 (+ one independent page dm I_3), the centre of mass as a point set (com_range [1, 1]) and the
    inertia about it with one independent page per distinct symmetric entry (d = relative uncertainty x |entry|).  `uncertain` False:
    the nominal parameters (pz_nominal), True: mass / inertia within +- the robot's uncertainty (pz_interval; the reference's
    uncertain_mass_range [0.97, 1.03], urdfs/urdf_utils/load_robot_params.m:9)."""
    from .robot_geometry import joint_frames
    T0, P, axes, _, _ = joint_frames(robot)
    n = len(axes)
    mass, com, I = [], [], []
    for i in range(n):
        um = (robot.mass_uncertainty_link[i] or robot.mass_uncertainty) if uncertain else 0.0
        ui = (robot.inertia_uncertainty_link[i] or robot.inertia_uncertainty) if uncertain else 0.0
        m = robot.mass[i]
        mass.append(MatPZ(m * np.eye(3), None, (um * abs(m) * np.eye(3))[:, :, None] if um else None))
        com.append(np.array(robot.com[3 * i:3 * i + 3]))
        Ic = np.array(robot.inertia[9 * i:9 * i + 9]).reshape(3, 3)
        pages = []
        for a in range(3):
            for b in range(a, 3):
                if ui and Ic[a, b] != 0.0:
                    Gp = np.zeros((3, 3)); Gp[a, b] = Gp[b, a] = ui * abs(Ic[a, b])
                    pages.append(Gp)
        I.append(MatPZ(Ic, None, np.stack(pages, axis=2) if pages else None))
    Pn = np.array(robot.trans)[:3 * (n + 1)].reshape(n + 1, 3).T     # (the last column: the frame after the last joint)
    return dict(mass=mass, com=com, I=I, T0=T0, P=Pn, axes=axes, n=n, zono_order=zono_order, gravity=robot.gravity)


def poly_zonotope_rnea(R_in, R_t_in, qd, qd_aux, qdd, use_gravity, prm):
    """PZM/utility/poly_zonotope_rnea.m:1-239 (all joints revolute): passivity RNEA on sets, Girard reduction to `zono_order`
    after every operation the reference reduces after.  R_in / R_t_in: MatPZ per joint; qd, qd_aux, qdd: 1-D PZ (or numbers)
    per joint.  Returns (u, f, n): the joint torque sets and the link force / moment sets."""
    n, zo = prm["n"], prm["zono_order"]
    red = lambda x: reduce(x, zo)
    T0, P, z = prm["T0"], prm["P"], prm["axes"]
    R = [mtimes(MatPZ(T0[i]), R_in[i]) for i in range(n)]                       # :49
    R_t = [mtimes(R_t_in[i], MatPZ(T0[i].T)) for i in range(n)]                 # :50
    w_p, wa_p, wd_p = np.zeros(3), np.zeros(3), np.zeros(3)                     # base frame (:69-72)
    la_p = np.array([0.0, 0.0, prm["gravity"]]) if use_gravity else np.zeros(3)   # linear_acc0 = -gravity' (:75-77)
    F, N = [], []
    for i in range(n):
        jv, ja, jva = scalar_times_axis(qd[i], z[i]), scalar_times_axis(qdd[i], z[i]), scalar_times_axis(qd_aux[i], z[i])
        Rt = R_t[i]
        w = red(plus(mtimes(Rt, w_p), jv))                                        # (6.45) :92,139-140
        wa = red(plus(mtimes(Rt, wa_p), jva))                                     # :95,143-144
        prod1 = red(mtimes(Rt, wa_p))                                             # :147-150
        prod2 = red(jv)
        wd = red(plus(plus(mtimes(Rt, wd_p), cross(prod1, prod2)), ja))           # (6.46) :151-154
        if i == 0:   # (6.47) :103-105 -- the reference's first joint uses cross(w0, cross(w0, P)), both zero at the base
            inner = plus(plus(_vec(la_p), cross(wd_p, P[:, i])), cross(w_p, cross(w_p, P[:, i])))
        else:        # :157-159
            inner = plus(plus(la_p, cross(wd_p, P[:, i])), cross(w_p, cross(wa_p, P[:, i])))
        la = red(mtimes(Rt, inner))
        com = prm["com"][i]
        p0 = red(cross(wd, com))                                                  # (6.48) :187-194
        p1 = red(cross(wa, com))
        lac = red(plus(plus(la, p0), cross(w, p1)))
        F.append(red(mtimes(prm["mass"][i], lac)))                                # (6.49) :197-198
        q0 = red(mtimes(prm["I"][i], wd))                                         # (6.50) :201-207
        q1 = red(mtimes(prm["I"][i], w))
        N.append(red(plus(q0, cross(wa, q1))))
        w_p, wa_p, wd_p, la_p = w, wa, wd, la
    f = [None] * n + [PZ(np.zeros(3))]
    nn = [None] * n + [PZ(np.zeros(3))]
    R.append(MatPZ(np.eye(3)))                                                    # :213 (the frame after the last joint: no rotation here)
    for i in range(n - 1, -1, -1):
        f[i] = red(plus(mtimes(R[i + 1], f[i + 1]), F[i]))                        # (6.51) :218-219
        p0 = red(mtimes(R[i + 1], nn[i + 1]))                                     # (6.52) :222-230
        p1 = red(mtimes(R[i + 1], f[i + 1]))
        nn[i] = red(plus(plus(plus(N[i], p0), cross(prm["com"][i], F[i])), cross(P[:, i + 1], p1)))
    u = [mtimes(z[i].reshape(1, 3), nn[i]) for i in range(n)]                     # (6.53) :237
    return u, f[:n], nn[:n]


def _buffered(p, sign):
    """c +- sum |Grest| with the independent part removed (KSI/uarmtd_planner.m:541-547): a PZ in k alone"""
    buf = np.abs(p.Grest).sum(axis=1)
    return PZ(p.c + sign * buf, p.G, None, p.E, p.id)


def input_constraints(q0, qd0, qdd0, robot, torque_limits, time_indices, n_t=100, zono_order=40, alpha_constant=10.0,
                      ultimate_bound=0.0191, k_r=10.0, use_robust_input=True, **jrs_kw):
    """KSI/uarmtd_planner.m:471-549: for the chosen time intervals the nominal torque sets tau_nom (PZ-RNEA with the nominal
    parameters), the disturbance w = tau_int - tau_nom against the interval parameters, the Lyapunov-function bound V (PZ-RNEA
    of the tracking-error set r without gravity), rho_max = || max(|w.inf|, |w.sup|) ||, v_norm = alpha V_diff.sup / ultimate_bound
    + rho_max, and the two constraint sets per joint
        u_ub = (tau_nom + v_norm, buffered upwards by its independent part) - u_max  <= 0
        u_lb = -(tau_nom - v_norm, buffered downwards)                   + u_min  <= 0.
    Returns {i: dict(tau_nom, v_norm, rho_max, V_sup, u_ub, u_lb)}; the reference's data-dependent pruning is `needed` below."""
    jrs = create_jrs_online(q0, qd0, qdd0, inertial_params(robot, False)["axes"].T, ultimate_bound=ultimate_bound, k_r=k_r, n_t=n_t,
                            full=True, time_indices=set(time_indices), **jrs_kw)
    nom, itv = inertial_params(robot, False, zono_order), inertial_params(robot, True, zono_order)
    n = nom["n"]
    lim = np.asarray(torque_limits, dtype=float).reshape(-1)
    out = {}
    for i in time_indices:
        R, R_t, dq, dqa, ddqa = jrs["R"][i], jrs["R_t"][i], jrs["Qd"][i], jrs["Qd_a"][i], jrs["Qdd_a"][i]
        tau_nom, _, _ = poly_zonotope_rnea(R, R_t, dq, dqa, ddqa, True, nom)
        v_norm, rho_max, V_sup = 0.0, 0.0, 0.0
        if use_robust_input:
            tau_int, _, _ = poly_zonotope_rnea(R, R_t, dq, dqa, ddqa, True, itv)
            w = [reduce(tau_int[j] - tau_nom[j], zono_order) for j in range(n)]                        # :478-481
            r = [PZ(0.0, None, [[ultimate_bound]]) for _ in range(n)]                                   # :455-457
            zero = [PZ(0.0) for _ in range(n)]
            V_cell, _, _ = poly_zonotope_rnea(R, R_t, zero, zero, r, False, itv)                       # :482
            V = 0.0
            for j in range(n):
                V = reduce(plus(V, times(times(0.5, r[j]), V_cell[j])), zono_order) if isinstance(V, PZ) else reduce(times(times(0.5, r[j]), V_cell[j]), zono_order)
            V_diff = reduce(V - V, zono_order)                                                          # :488-490
            V_sup = float(interval(V_diff)[1][0])
            wl = np.array([interval(w[j])[0][0] for j in range(n)]); wu = np.array([interval(w[j])[1][0] for j in range(n)])
            rho_max = float(np.linalg.norm(np.maximum(np.abs(wl), np.abs(wu))))                         # :507-512
            v_norm = alpha_constant * V_sup / ultimate_bound + rho_max                                  # :516
        u_ub, u_lb = [], []
        for j in range(n):
            ub = remove_dependence(tau_nom[j] + v_norm, n)                                              # :531-536
            lb = remove_dependence(tau_nom[j] - v_norm, n)
            u_ub.append(_buffered(ub, +1.0) - lim[j])                                                   # :537-540 (limits(2, j) = +u_max)
            u_lb.append(times(-1.0, _buffered(lb, -1.0)) + (-lim[j]))                                   # limits(1, j) = -u_max
        out[i] = dict(tau_nom=tau_nom, v_norm=v_norm, rho_max=rho_max, V_sup=V_sup, u_ub=u_ub, u_lb=u_lb)
    return out


def joint_limit_constraints(q0, qd0, qdd0, robot, time_indices, n_t=100, **jrs_kw):
    """KSI/uarmtd_planner.m:562-576: position / velocity sets buffered by their independent part against the joint limits.
    (The reference adds the buffer with the same sign on the lower bound, `-1*(c + buf) + lower` -- kept; a continuous joint has
    limits +-1000 here as in RT/KinovaWithoutGripperInfo.h:76-77, +-inf in the MATLAB agent: never active either way.)"""
    from .robot_geometry import joint_frames
    axes = joint_frames(robot)[2]
    n = len(axes)
    jrs = create_jrs_online(q0, qd0, qdd0, axes.T, n_t=n_t, full=True, time_indices=set(time_indices), **jrs_kw)
    out = {}
    for i in time_indices:
        row = []
        for j in range(n):
            qs, vs = remove_dependence(jrs["Q"][i][j], n), remove_dependence(jrs["Qd"][i][j], n)
            qb, vb = _buffered(qs, +1.0), _buffered(vs, +1.0)
            row.append(dict(q_ub=qb - robot.state_limits_ub[j], q_lb=times(-1.0, qb) + robot.state_limits_lb[j],
                            dq_ub=vb - robot.speed_limits[j], dq_lb=times(-1.0, vb) + (-robot.speed_limits[j])))
        out[i] = row
    return out


def needed(con):
    """the reference keeps a constraint only if it can be violated: `~(interval(con).sup < 0)` (KSI/uarmtd_planner.m:628,640,659,...)"""
    return not (interval(con)[1][0] < 0)


def eval_constraint(con, k, n_q):
    """slice and gradient of a 1-D constraint set at k (KSI/uarmtd_planner.m:632-634): (value, d value / d k [n_q])"""
    return float(slice_pz(con, k)[0]), grad_slice(con, k, n_q)[:, 0]
