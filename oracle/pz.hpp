/*
 * oracle/pz.hpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's sparse polynomial-zonotope arithmetic
 * (RT/PZsparse.h, RT/PZsparse.cu of roahmlab/armour; RT/ =
 * kinova_src/kinova_simulator_interfaces/kinova_planner_realtime/).  Each function cites the
 * reference lines it follows.  Plain C++17, no Eigen / Boost: coefficients are fixed 3x3 (max)
 * row-major arrays and intervals are a two-double struct with outward nudging.
 *
 * PARITY UNPINNED: the reference ships no golden vectors for this path and cannot be compiled
 * in this image (needs CUDA, Eigen 3.3.7, Boost.Interval, IPOPT/HSL -- none present), so this
 * restatement is pinned only by the invariants in tests/ (scalar-RNEA containment,
 * finite-difference Jacobian, algebraic identities), not by recorded reference outputs.
 *
 * Known, documented ulp-level differences from the reference build:
 *   - Eigen's vectorised redux order for squaredNorm of a 3x3 (pairs of lanes) vs. the
 *     left-to-right sum here (affects only the "<= threshold" test at the 1e-16 level);
 *   - Boost.Interval directed rounding vs. one-ulp outward nudging here.
 */
#ifndef ORACLE_PZ_HPP
#define ORACLE_PZ_HPP

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace oracle {

/* monomial key: the reference's u64 (RT/PZsparse.h:23-35); a -DARMOUR_KEY128 build widens it to 128 bits for an 8-factor arm, which the
 * reference itself cannot represent -- same packing, same plain integer addition of keys (RT/PZsparse.cu:938-940) */
#if defined(ARMOUR_KEY128)
typedef unsigned __int128 okey_t;
#else
typedef uint64_t okey_t;
#endif

/* ---- key layout: RT/PZsparse.h:23-40 (MOVE_BIT_INC / DEGREE_MASK), generalised to n factors ---- */
struct KeyLayout {
    int n = 7;
    /* field groups in LSB-first order: k (2 bits), qde (1), qdae (1), qddae (1), cosqe (2), sinqe (2) */
    int shift_k(int j) const { return 2 * j; }
    int shift_qde(int j) const { return 2 * n + j; }
    int shift_qdae(int j) const { return 3 * n + j; }
    int shift_qddae(int j) const { return 4 * n + j; }
    int shift_cosqe(int j) const { return 5 * n + 2 * j; }
    int shift_sinqe(int j) const { return 7 * n + 2 * j; }
    okey_t max_hash_dependent_k_only() const { return (okey_t)1 << (2 * n); }       /* PZsparse.h:37 */
    okey_t max_hash_dependent_k_links_only() const { return (okey_t)1 << (5 * n); } /* PZsparse.h:39 */
    okey_t dependent_k_mask() const { return max_hash_dependent_k_only() - 1; }       /* PZsparse.h:40 */
    int k_degree(okey_t key, int j) const { return (int)((key >> (2 * j)) & 3); }     /* PZsparse.cu:578-585 */
};

struct Mono {
    okey_t key;
    double c[9];
};

/* counters used by tests / bench to report the measured table sizes */
struct Stats {
    uint64_t mul_calls = 0, mul_pairs = 0, simplify_calls = 0, simplify_terms = 0;
    uint64_t max_raw_terms = 0, max_out_terms = 0;
    double min_margin = 1e300; /* min over all simplify() decisions of |norm - threshold| / threshold */
    void merge(const Stats& o) {
        min_margin = std::min(min_margin, o.min_margin);
        mul_calls += o.mul_calls; mul_pairs += o.mul_pairs;
        simplify_calls += o.simplify_calls; simplify_terms += o.simplify_terms;
        max_raw_terms = std::max(max_raw_terms, o.max_raw_terms);
        max_out_terms = std::max(max_out_terms, o.max_out_terms);
    }
};

struct Ctx {
    KeyLayout kl;
    double threshold = 5e-4; /* SIMPLIFY_THRESHOLD, RT/Parameters.h:10 */
    Stats st;
};

struct PZ {
    int R = 0, C = 0;
    double center[9] = {0};
    std::vector<Mono> poly;
    double indep[9] = {0};

    PZ() {}
    PZ(int r, int c) : R(r), C(c) {}            /* PZsparse.cu:50-55 */
    int sz() const { return R * C; }
};

inline PZ pz_scalar(double v) { PZ p(1, 1); p.center[0] = v; return p; }  /* PZsparse.cu:66-72 */

inline PZ pz_matrix(int r, int c, const double* v) {                      /* PZsparse.cu:75-80 */
    PZ p(r, c);
    for (int i = 0; i < r * c; i++) p.center[i] = v[i];
    return p;
}

inline PZ pz_matrix_uncertain(int r, int c, const double* v, double pct) { /* PZsparse.cu:93-98 */
    PZ p = pz_matrix(r, c, v);
    for (int i = 0; i < r * c; i++) p.indep[i] = pct * std::fabs(v[i]);
    return p;
}

inline double frob(const double* v, int n) {
    double s = 0;
    for (int i = 0; i < n; i++) s += v[i] * v[i];
    return std::sqrt(s);
}

/* PZsparse.cu:284-350: sort by key, sum equal keys, move small coefficients to the independent part */
inline void simplify(Ctx& cx, PZ& p) {
    const int n = p.sz();
    cx.st.simplify_calls++;
    cx.st.simplify_terms += p.poly.size();
    cx.st.max_raw_terms = std::max<uint64_t>(cx.st.max_raw_terms, p.poly.size());
    std::sort(p.poly.begin(), p.poly.end(), [](const Mono& a, const Mono& b) { return a.key < b.key; });
    double reduce_amount[9] = {0};
    std::vector<Mono> out;
    out.reserve(p.poly.size());
    size_t i = 0;
    while (i < p.poly.size()) {
        size_t j;
        const okey_t key = p.poly[i].key;
        for (j = i + 1; j < p.poly.size(); j++) {
            if (p.poly[j].key != key) break;
            for (int e = 0; e < n; e++) p.poly[i].c[e] += p.poly[j].c[e];
        }
        const double nrm = frob(p.poly[i].c, n);
        if (nrm > 0) cx.st.min_margin = std::min(cx.st.min_margin, std::fabs(nrm - cx.threshold) / cx.threshold);
        if (nrm <= cx.threshold) {
            for (int e = 0; e < n; e++) reduce_amount[e] += std::fabs(p.poly[i].c[e]);
        } else {
            out.push_back(p.poly[i]);
        }
        i = j;
    }
    p.poly.swap(out);
    cx.st.max_out_terms = std::max<uint64_t>(cx.st.max_out_terms, p.poly.size());
    if (frob(reduce_amount, n) != 0) {
        for (int e = 0; e < n; e++) p.indep[e] = p.indep[e] + reduce_amount[e];
    }
}

/* PZsparse.cu:120-136: 1x1 PZ from (center, coeffs, keys) */
inline PZ pz_scalar_poly(Ctx& cx, double center, const double* coeff, const okey_t* keys, int m) {
    PZ p(1, 1);
    p.center[0] = center;
    for (int i = 0; i < m; i++) {
        Mono mo{};
        mo.key = keys[i];
        mo.c[0] = coeff[i];
        p.poly.push_back(mo);
    }
    simplify(cx, p);
    return p;
}

/* PZsparse.cu:160-176: rotation from roll/pitch/yaw, centre only */
inline PZ pz_rpy(double roll, double pitch, double yaw) {
    PZ p(3, 3);
    double* c = p.center;
    c[0] = cos(pitch) * cos(yaw);
    c[1] = -cos(pitch) * sin(yaw);
    c[2] = sin(pitch);
    c[3] = cos(roll) * sin(yaw) + cos(yaw) * sin(pitch) * sin(roll);
    c[4] = cos(roll) * cos(yaw) - sin(pitch) * sin(roll) * sin(yaw);
    c[5] = -cos(pitch) * sin(roll);
    c[6] = sin(roll) * sin(yaw) - cos(roll) * cos(yaw) * sin(pitch);
    c[7] = cos(yaw) * sin(roll) + cos(roll) * sin(pitch) * sin(yaw);
    c[8] = cos(pitch) * cos(roll);
    return p;
}

/* PZsparse.cu:211-250 */
inline void make_rotation(double* Rm, double cosE, double sinE, int axis, bool from_zero) {
    for (int i = 0; i < 9; i++) Rm[i] = 0;
    if (!from_zero) Rm[0] = Rm[4] = Rm[8] = 1;
    const double negSin = -1.0 * sinE;
    switch (axis) {
        case 0: return;
        case 1: Rm[4] = cosE; Rm[5] = negSin; Rm[7] = sinE; Rm[8] = cosE; break;
        case 2: Rm[0] = cosE; Rm[2] = sinE; Rm[6] = negSin; Rm[8] = cosE; break;
        case 3: Rm[0] = cosE; Rm[1] = negSin; Rm[3] = sinE; Rm[4] = cosE; break;
        default: assert(false);
    }
}

/* PZsparse.cu:179-205: 3x3 rotation PZ about `axis` from the cos / sin 1x1 polynomial data */
inline PZ pz_rotation(Ctx& cx, double cos_c, const double* cos_coeff, const okey_t* cos_keys, int cm,
                      double sin_c, const double* sin_coeff, const okey_t* sin_keys, int sm, int axis) {
    PZ p(3, 3);
    make_rotation(p.center, cos_c, sin_c, axis, false);
    for (int i = 0; i < cm; i++) {
        Mono mo{};
        mo.key = cos_keys[i];
        make_rotation(mo.c, cos_coeff[i], 0, axis, true);
        p.poly.push_back(mo);
    }
    for (int i = 0; i < sm; i++) {
        Mono mo{};
        mo.key = sin_keys[i];
        make_rotation(mo.c, 0, sin_coeff[i], axis, true);
        p.poly.push_back(mo);
    }
    simplify(cx, p);
    return p;
}

/* PZsparse.cu:352-368 */
inline void reduce(Ctx& cx, PZ& p) {
    const int n = p.sz();
    std::vector<Mono> out;
    for (const Mono& it : p.poly) {
        if (it.key < cx.kl.max_hash_dependent_k_only()) out.push_back(it);
        else for (int e = 0; e < n; e++) p.indep[e] += std::fabs(it.c[e]);
    }
    p.poly.swap(out);
}

/* PZsparse.cu:370-402: returns the 3x6 matrix row-major in out18 */
inline void reduce_link_PZ(Ctx& cx, PZ& p, double* out18) {
    assert(p.R == 3 && p.C == 1);
    for (int i = 0; i < 18; i++) out18[i] = 0;
    std::vector<Mono> out;
    int j = 0;
    for (const Mono& it : p.poly) {
        if (it.key < cx.kl.max_hash_dependent_k_only()) {
            out.push_back(it);
        } else if (it.key < cx.kl.max_hash_dependent_k_links_only() && (it.key & cx.kl.dependent_k_mask()) == 0) {
            assert(j < 3);
            for (int r = 0; r < 3; r++) out18[r * 6 + j] = it.c[r];
            j++;
        } else {
            for (int e = 0; e < 3; e++) p.indep[e] += std::fabs(it.c[e]);
        }
    }
    p.poly.swap(out);
    out18[0 * 6 + 3] = p.indep[0];
    out18[1 * 6 + 4] = p.indep[1];
    out18[2 * 6 + 5] = p.indep[2];
}

/* PZsparse.cu:404-435 -- slice value: centre of the result and its radius */
inline void slice_value(const Ctx& cx, const PZ& p, const double* factor, double* center_out, double* radius_out) {
    const int n = p.sz();
    double cen[9], rad[9];
    for (int e = 0; e < n; e++) { cen[e] = p.center[e]; rad[e] = p.indep[e]; }
    for (const Mono& it : p.poly) {
        double t[9];
        for (int e = 0; e < n; e++) t[e] = it.c[e];
        if (it.key < ((okey_t)1 << (2 * cx.kl.n))) {
            for (int j = 0; j < cx.kl.n; j++) {
                const double pw = std::pow(factor[j], (double)cx.kl.k_degree(it.key, j));
                for (int e = 0; e < n; e++) t[e] *= pw;
            }
            for (int e = 0; e < n; e++) cen[e] += t[e];
        } else {
            for (int e = 0; e < n; e++) rad[e] += std::fabs(t[e]);
        }
    }
    /* res = Interval(c - r, c + r); getCenter = (lo + hi) * 0.5 (PZsparse.cu:10-12,427-432) */
    for (int e = 0; e < n; e++) {
        const double lo = cen[e] - rad[e], hi = cen[e] + rad[e];
        center_out[e] = (lo + hi) * 0.5;
        if (radius_out) radius_out[e] = (hi - lo) * 0.5;
    }
}

/* PZsparse.cu:437-555 (all three overloads share this body): grad_out[k*sz + e] */
inline void slice_gradient(const Ctx& cx, const PZ& p, const double* factor, double* grad_out) {
    const int n = p.sz(), nf = cx.kl.n;
    for (int i = 0; i < nf * n; i++) grad_out[i] = 0;
    for (const Mono& it : p.poly) {
        if (it.key <= ((okey_t)1 << (2 * nf))) { /* "<=" as in PZsparse.cu:447,488,527 */
            double t[ARMOUR_MAX_FACTORS][9];
            for (int k = 0; k < nf; k++) for (int e = 0; e < n; e++) t[k][e] = it.c[e];
            for (int j = 0; j < nf; j++) {
                const int dj = cx.kl.k_degree(it.key, j);
                for (int k = 0; k < nf; k++) {
                    if (j == k) {
                        if (dj == 0) { for (int e = 0; e < n; e++) t[k][e] = 0; }
                        else {
                            const double f = (double)dj * std::pow(factor[j], (double)(dj - 1));
                            for (int e = 0; e < n; e++) t[k][e] *= f;
                        }
                    } else {
                        const double f = std::pow(factor[j], (double)dj);
                        for (int e = 0; e < n; e++) t[k][e] *= f;
                    }
                }
            }
            for (int k = 0; k < nf; k++) for (int e = 0; e < n; e++) grad_out[k * n + e] += t[k][e];
        }
    }
}

/* PZsparse.cu:557-576 : centre and radius of the interval hull */
inline void to_interval(const PZ& p, double* lo, double* hi) {
    const int n = p.sz();
    for (int e = 0; e < n; e++) {
        double rad = p.indep[e];
        for (const Mono& it : p.poly) rad += std::fabs(it.c[e]);
        lo[e] = p.center[e] - rad;
        hi[e] = p.center[e] + rad;
    }
}
/* NB: the reference accumulates res_radius monomial-major (all entries per monomial); for one
 * entry the order of additions is the same, so the loop interchange above is value-identical. */

/* PZsparse.cu:678-697 */
inline PZ elem(const PZ& p, int r, int c) {
    PZ o(1, 1);
    const int idx = r * p.C + c;
    o.center[0] = p.center[idx];
    o.poly.reserve(p.poly.size());
    for (const Mono& it : p.poly) {
        Mono mo{};
        mo.key = it.key;
        mo.c[0] = it.c[idx];
        o.poly.push_back(mo);
    }
    o.indep[0] = p.indep[idx];
    return o;
}

/* PZsparse.cu:743-764 */
inline PZ add(Ctx& cx, const PZ& a, const PZ& b) {
    PZ o(a.R, a.C);
    const int n = a.sz();
    for (int e = 0; e < n; e++) o.center[e] = a.center[e] + b.center[e];
    o.poly.reserve(a.poly.size() + b.poly.size());
    o.poly.insert(o.poly.end(), a.poly.begin(), a.poly.end());
    o.poly.insert(o.poly.end(), b.poly.begin(), b.poly.end());
    for (int e = 0; e < n; e++) o.indep[e] = a.indep[e] + b.indep[e];
    simplify(cx, o);
    return o;
}

/* PZsparse.cu:813-834 */
inline PZ sub(Ctx& cx, const PZ& a, const PZ& b) {
    PZ o(a.R, a.C);
    const int n = a.sz();
    for (int e = 0; e < n; e++) o.center[e] = a.center[e] - b.center[e];
    o.poly.reserve(a.poly.size() + b.poly.size());
    o.poly.insert(o.poly.end(), a.poly.begin(), a.poly.end());
    for (const Mono& it : b.poly) {
        Mono mo = it;
        for (int e = 0; e < n; e++) mo.c[e] = -it.c[e];
        o.poly.push_back(mo);
    }
    for (int e = 0; e < n; e++) o.indep[e] = a.indep[e] + b.indep[e];
    simplify(cx, o);
    return o;
}

/* PZsparse.cu:996-1030 (PZ * double and double * PZ are the same arithmetic) */
inline PZ scale(const PZ& a, double s) {
    PZ o(a.R, a.C);
    const int n = a.sz();
    for (int e = 0; e < n; e++) o.center[e] = a.center[e] * s;
    o.poly.reserve(a.poly.size());
    for (const Mono& it : a.poly) {
        Mono mo = it;
        for (int e = 0; e < n; e++) mo.c[e] = s * it.c[e];
        o.poly.push_back(mo);
    }
    for (int e = 0; e < n; e++) o.indep[e] = a.indep[e] * std::fabs(s);
    return o;
}

/* PZsparse.cu:1050-1066 */
inline PZ transpose(const PZ& a) {
    PZ o(a.C, a.R);
    auto tr = [&](const double* s, double* d) {
        for (int r = 0; r < a.R; r++) for (int c = 0; c < a.C; c++) d[c * a.R + r] = s[r * a.C + c];
    };
    tr(a.center, o.center);
    for (const Mono& it : a.poly) {
        Mono mo{};
        mo.key = it.key;
        tr(it.c, mo.c);
        o.poly.push_back(mo);
    }
    tr(a.indep, o.indep);
    return o;
}

inline void matmul(const double* A, int ar, int ac, const double* B, int bc, double* O) {
    for (int r = 0; r < ar; r++)
        for (int c = 0; c < bc; c++) {
            double s = 0;
            for (int k = 0; k < ac; k++) s += A[r * ac + k] * B[k * bc + c];
            O[r * bc + c] = s;
        }
}

/* generic "coefficient product" honouring the 1x1 broadcast rules of PZsparse.cu:870-919 */
inline void coeff_mul(const double* A, int ar, int ac, const double* B, int br, int bc, double* O) {
    if (ar == 1 && ac == 1) { for (int e = 0; e < br * bc; e++) O[e] = A[0] * B[e]; }
    else if (br == 1 && bc == 1) { for (int e = 0; e < ar * ac; e++) O[e] = A[e] * B[0]; }
    else matmul(A, ar, ac, B, bc, O);
}

/* PZsparse.cu:864-994 */
inline PZ mul(Ctx& cx, const PZ& a, const PZ& b) {
    PZ o;
    const bool a11 = (a.R == 1 && a.C == 1), b11 = (b.R == 1 && b.C == 1);
    if (a11) { o.R = b.R; o.C = b.C; }
    else if (b11) { o.R = a.R; o.C = a.C; }
    else { assert(a.C == b.R); o.R = a.R; o.C = b.C; }
    const int n = o.sz();
    cx.st.mul_calls++;
    cx.st.mul_pairs += (uint64_t)(a.poly.size() + 1) * (b.poly.size() + 1);

    coeff_mul(a.center, a.R, a.C, b.center, b.R, b.C, o.center);
    o.poly.reserve(a.poly.size() + b.poly.size() + a.poly.size() * b.poly.size());
    for (const Mono& it : a.poly) {            /* polynomial * a.center  (:896-906) */
        Mono mo{};
        mo.key = it.key;
        coeff_mul(it.c, a.R, a.C, b.center, b.R, b.C, mo.c);
        o.poly.push_back(mo);
    }
    for (const Mono& it : b.poly) {            /* center * a.polynomial  (:909-919) */
        Mono mo{};
        mo.key = it.key;
        coeff_mul(a.center, a.R, a.C, it.c, b.R, b.C, mo.c);
        o.poly.push_back(mo);
    }
    for (const Mono& i1 : a.poly)              /* all pairs, plain u64 key add (:924-942) */
        for (const Mono& i2 : b.poly) {
            Mono mo{};
            mo.key = i1.key + i2.key;
            coeff_mul(i1.c, a.R, a.C, i2.c, b.R, b.C, mo.c);
            o.poly.push_back(mo);
        }

    /* independent part (:944-989) */
    double r2[9] = {0}, r3[9] = {0}, t2[9], t3[9], ii[9];
    for (int e = 0; e < a.sz(); e++) r2[e] = std::fabs(a.center[e]);
    for (const Mono& it : a.poly) for (int e = 0; e < a.sz(); e++) r2[e] += std::fabs(it.c[e]);
    coeff_mul(r2, a.R, a.C, b.indep, b.R, b.C, t2);
    for (int e = 0; e < b.sz(); e++) r3[e] = std::fabs(b.center[e]);
    for (const Mono& it : b.poly) for (int e = 0; e < b.sz(); e++) r3[e] += std::fabs(it.c[e]);
    coeff_mul(a.indep, a.R, a.C, r3, b.R, b.C, t3);
    coeff_mul(a.indep, a.R, a.C, b.indep, b.R, b.C, ii);
    for (int e = 0; e < n; e++) o.indep[e] = ii[e] + (t2[e] + t3[e]);
    simplify(cx, o);
    return o;
}

/* PZsparse.cu:1068-1085 */
inline void add_one_dim(Ctx& cx, PZ& p, const PZ& a, int r, int c) {
    assert(a.R == 1 && a.C == 1);
    const int idx = r * p.C + c;
    p.center[idx] += a.center[0];
    for (const Mono& it : a.poly) {
        Mono mo{};
        mo.key = it.key;
        mo.c[idx] = it.c[0];
        p.poly.push_back(mo);
    }
    p.indep[idx] += a.indep[0];
    simplify(cx, p);
}

/* PZsparse.cu:1087-1116 */
inline PZ stack3(Ctx& cx, const PZ& a0, const PZ& a1, const PZ& a2) {
    const PZ* a[3] = {&a0, &a1, &a2};
    PZ o(3, 1);
    for (int i = 0; i < 3; i++) o.center[i] = a[i]->center[0];
    for (int i = 0; i < 3; i++)
        for (const Mono& it : a[i]->poly) {
            Mono mo{};
            mo.key = it.key;
            mo.c[i] = it.c[0];
            o.poly.push_back(mo);
        }
    for (int i = 0; i < 3; i++) o.indep[i] = a[i]->indep[0];
    simplify(cx, o);
    return o;
}

/* PZsparse.cu:1118-1132 : constant vector x PZ */
inline PZ cross_mat_pz(Ctx& cx, const double* a, const PZ& b) {
    PZ b0 = elem(b, 0, 0), b1 = elem(b, 1, 0), b2 = elem(b, 2, 0);
    PZ r0 = sub(cx, scale(b2, a[1]), scale(b1, a[2]));
    PZ r1 = sub(cx, scale(b0, a[2]), scale(b2, a[0]));
    PZ r2 = sub(cx, scale(b1, a[0]), scale(b0, a[1]));
    return stack3(cx, r0, r1, r2);
}

/* PZsparse.cu:1134-1151 : PZ x PZ */
inline PZ cross_pz_pz(Ctx& cx, const PZ& a, const PZ& b) {
    PZ a0 = elem(a, 0, 0), a1 = elem(a, 1, 0), a2 = elem(a, 2, 0);
    PZ b0 = elem(b, 0, 0), b1 = elem(b, 1, 0), b2 = elem(b, 2, 0);
    PZ r0 = sub(cx, mul(cx, a1, b2), mul(cx, a2, b1));
    PZ r1 = sub(cx, mul(cx, a2, b0), mul(cx, a0, b2));
    PZ r2 = sub(cx, mul(cx, a0, b1), mul(cx, a1, b0));
    return stack3(cx, r0, r1, r2);
}

/* PZsparse.cu:1153-1167 : PZ x constant vector */
inline PZ cross_pz_mat(Ctx& cx, const PZ& a, const double* b) {
    PZ a0 = elem(a, 0, 0), a1 = elem(a, 1, 0), a2 = elem(a, 2, 0);
    PZ r0 = sub(cx, scale(a1, b[2]), scale(a2, b[1]));
    PZ r1 = sub(cx, scale(a2, b[0]), scale(a0, b[2]));
    PZ r2 = sub(cx, scale(a0, b[1]), scale(a1, b[0]));
    return stack3(cx, r0, r1, r2);
}

}  // namespace oracle
#endif
