// Products of polynomial zonotopes by HASH CLASSIFICATION of the raw terms (included by pz_wave.h inside namespace pzw).
//
// What the reference does (RT/PZsparse.cu:864-994 + simplify, :284-350): form all (na+1)(nb+1)-1 raw terms, sort them by
// key, add terms of equal key, drop every sum whose norm is <= SIMPLIFY_THRESHOLD into the independent radius.  Measured
// on the reference's own operator sequence (oracle statistics, one Kinova problem, T = 100): 92 % of the raw terms of a
// product have a key no other raw term shares, and 89 % of the raw terms end up pruned -- so sorting ALL raw terms (rounds
// 1-2: 40 % of the chain kernel in the sort, 35 % in the ordered sums that gather operand coefficients from memory) is work
// spent on terms whose fate needs no neighbour.  Here a product is four passes over data that stays on the CU:
//
//   A  mark      every raw term's key hash h(ka + kb) = h(ka) + h(kb) (multiplicative hash: linear, two adds per term) is
//                offered to a small LDS table of 32-bit fingerprints (compare-and-swap, at most four probes).  A term that
//                meets its own fingerprint again -- or finds no free slot -- sets a bit, indexed by other bits of the same
//                hash, in an 8-kbit filter.  Afterwards: filter bit clear  =>  the term's key is unique among the raw terms.
//                (False positives -- equal fingerprints of different keys, a shared filter bit, a full probe sequence --
//                only send a term the slow way.  There are no false negatives: terms of equal key have equal hashes, walk
//                the same probe sequence, and the table only fills up.)
//   B  classify  the operands sit in registers (one operand's terms across the lanes, the other's broadcast row by row with
//                v_readlane): coefficient product, then
//                  unique key   -> simplify()'s verdict at once: pruned into the lane's radius sum, or appended to the
//                                  FINAL list (key, finished coefficient) in LDS;
//                  shared key   -> appended to the SLOW list (key, raw coefficient product) in LDS.
//                The loops run in an order that leaves terms of equal key in generation order on the slow list.
//   D  the slow list (~8 % of the raw terms) is sorted by (key, position), runs of equal keys are summed in generation
//      order exactly as rounds 1-2 did for all terms, and simplify()'s verdict moves each sum to the radius or the final list;
//   E  the final list (unique keys, ~11 % of the raw terms) is sorted by key and written to the result slot.
//
// Coefficients are the same sums in the same order as before (generation order: a-major), so tables keep their keys and
// coefficients bit for bit; the pruned-radius sums add the same numbers in a different (fixed) order, i.e. they move in the
// last bits.  Everything is still a function of the operands alone -- one wave, fixed loop orders -- so a (problem, time
// step) item gives the same tables in every launch shape.
//
// LDS: the wave's sort block (skey | sidx of pz_wave.h) is re-used as  table | filter | list keys | list indices | values.
// The table is as large as the block allows; a product with more raw terms than half the table is marked in several
// passes over disjoint hash classes, so the raw-term count itself never overflows anything.  What can overflow is the two
// lists (ERR_RAW_OVERFLOW -> the host retries with a larger block, as before).

#ifdef P1_PROFILE
#define H_LAP(slot) { const long long n__ = clock64(); if (lane == 0) w.prof[slot] += (unsigned long long)(n__ - h_t); h_t = n__; }
#define H_LAP0 long long h_t = clock64();
#else
#define H_LAP(slot)
#define H_LAP0
#endif
constexpr unsigned long long kHashMul = 0x9E3779B97F4A7C15ull;  // 2^64 / golden ratio, odd
constexpr int kBloomWords = 256;                                // 8192 filter bits
constexpr int kHashProbes = 4;

struct HashLds {
    LDS_AS uint32_t* htab; int shift; unsigned mask;   // mask + 1 slots, slot = h >> shift
    LDS_AS uint32_t* bloom;
    LDS_AS uint64_t* ckey; LDS_AS uint16_t* cidx; int fcap, scap;  // final list in [0, fcap), slow list in [fcap, fcap + scap)
    LDS_AS double* cval; int nval;                     // final values from the bottom, slow values from the top
};

// A value every lane holds alike, moved to a scalar register (see the note above hash_mark).
__device__ inline int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ inline HashLds hash_lds(const Wave& w) {
    HashLds H;
    const int bytes = uni(w.cap_key) * 8 + uni(w.cap_raw) * 2;   // skey and sidx are contiguous (p1_reach.hip)
    int ts = 256, lg = 8;
    while ((ts * 2) * 4 * 5 <= bytes) { ts *= 2; lg++; }          // table <= 20 % of the block: 2048 slots in 40 KB, 1024 in 28 KB
    // list capacities (powers of two, the bitonic network pads to one): the final list is bounded by the result slot's
    // capacity (work_monomials, 1024) anyway; the slow list holds the few terms whose key is shared
    const int fcap = bytes >= 36 * 1024 ? 1024 : bytes >= 18 * 1024 ? 512 : 256;
    const int scap = bytes >= 18 * 1024 ? 256 : 128;
    const int pk = fcap + scap;
    LDS_AS unsigned char* p = (LDS_AS unsigned char*)w.skey;
    H.htab = (LDS_AS uint32_t*)p; p += (size_t)ts * 4;
    H.bloom = (LDS_AS uint32_t*)p; p += kBloomWords * 4;
    H.ckey = (LDS_AS uint64_t*)p; p += (size_t)pk * 8;
    H.cidx = (LDS_AS uint16_t*)p; p += (size_t)pk * 2;
    H.cval = (LDS_AS double*)p;
    H.nval = (bytes - (ts * 4 + kBloomWords * 4 + pk * 10)) / 8;
    H.fcap = fcap; H.scap = scap; H.mask = (unsigned)(ts - 1); H.shift = 64 - lg;
    return H;
}
// smallest block the products can work in (the debug hook and the host check against it)
constexpr int kHashMinBlockBytes = 256 * 4 + kBloomWords * 4 + (256 + 128) * 10 + 64 * 9 * 8 * 2;

__device__ inline uint64_t readlane_u64(uint64_t v, int l) {
    return ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)v, l);
}
__device__ inline double readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ inline unsigned bloom_index(uint64_t h) { return (unsigned)(h >> 38) & (unsigned)(kBloomWords * 32 - 1); }

// term t of a view in "centre first" numbering: t = 0 is the centre (key 0), t >= 1 monomial t - 1
__device__ inline uint64_t term_key(const GLB_AS uint64_t* keys, int t, bool ok) { return (ok && t > 0) ? keys[t - 1] : 0ull; }
// A value every lane holds alike, moved to a scalar register: loop bounds and v_readlane selectors must be scalar, and what
// arrives through a struct reference is a (flat) load the compiler has to treat as divergent.
// NOTE for everything below: `w`, the views and the result slot arrive BY REFERENCE, i.e. as pointers into the caller's
// stack frame; every `w.field` inside a loop is a flat load with a full wait behind it.  The hot loops therefore work on
// local copies taken once at function entry.

// Pass A.  `sk` / `lk`: key lists of the operand walked row by row and of the operand spread over the lanes (which is which
// does not matter here); ns1 / nl1 = their term counts, centre included (scalar).
__device__ PZW_NOINLINE void hash_mark(Wave& w, HashLds H, const GLB_AS uint64_t* sk, int ns1_, const GLB_AS uint64_t* lk, int nl1_, int N_) {
    PROF_T0
    const int lane = w.lane;
    const int ns1 = uni(ns1_), nl1 = uni(nl1_), N = uni(N_);
    const unsigned mask = (unsigned)uni((int)H.mask);
    const int shift = uni(H.shift);
    LDS_AS uint32_t* const htab = H.htab;
    LDS_AS uint32_t* const bloom = H.bloom;
    const int ts = (int)mask + 1;
    int npass = 1;
    while (N > npass * (ts >> 1)) npass <<= 1;
    H_LAP0
    for (int i = lane; i < kBloomWords; i += WAVE) bloom[i] = 0u;
    for (int pass = 0; pass < npass; pass++) {
        for (int i = lane; i < ts; i += WAVE) htab[i] = 0u;
        H_LAP(PR_S_RANK)
        for (int lc = 0; lc < nl1; lc += WAVE) {
            const int lt = lc + lane;
            const bool lvalid = lt < nl1;
            const uint64_t hL = term_key(lk, lt, lvalid) * kHashMul;
            for (int sc = 0; sc < ns1; sc += WAVE) {
                const int st = sc + lane;
                const uint64_t hS = term_key(sk, st, st < ns1) * kHashMul;
                const int rows = min(WAVE, ns1 - sc);
                H_LAP(PR_S_LINMERGE)
                for (int r = 0; r < rows; r++) {
                    const uint64_t h = readlane_u64(hS, r) + hL;
                    bool pend = lvalid && !(sc + r == 0 && lt == 0) && (int)((unsigned)(h >> 34) & (unsigned)(npass - 1)) == pass;
                    unsigned slot = (unsigned)(h >> shift);
                    const unsigned fp = (unsigned)(h >> 20) | 1u;
                    bool mark = false;
                    for (int k = 0; k < kHashProbes; k++) {
                        if (pend) {
                            unsigned seen = 0u;
                            __hip_atomic_compare_exchange_strong(&htab[slot], &seen, fp, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            if (seen == 0u) pend = false;
                            else if (seen == fp) { pend = false; mark = true; }
                            else slot = (slot + 1u) & mask;
                        }
                        if (__ballot(pend) == 0ull) break;
                    }
                    if (mark || pend) {
                        const unsigned bi = bloom_index(h);
                        __hip_atomic_fetch_or(&bloom[bi >> 5], 1u << (bi & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    }
                }
            }
        }
    }
    WSYNC();
    PROF_ADD(PR_SORT) PROF_ADD(PR_S_MULMERGE)
}

// Sort list entries [0, M) of (K, I) ascending by (key, index value); I[p] == p on entry.  Afterwards K is sorted and I[p]
// is the original position of the p-th entry.  M <= 64: ranks by counting in registers; otherwise the bitonic network.
__device__ inline void sort_list(int lane, LDS_AS uint64_t* K, LDS_AS uint16_t* I, int M) {
    if (M <= 1) return;
    if (M <= WAVE) {
        const uint64_t key = lane < M ? K[lane] : ~0ull;
        const unsigned klo = (unsigned)key, khi = (unsigned)(key >> 32);
        int rank = 0;
        for (int l = 0; l < M; l++) {
            const uint64_t kl = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)khi, l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)klo, l);
            rank += (kl < key || (kl == key && l < lane)) ? 1 : 0;
        }
        WSYNC();
        if (lane < M) { K[rank] = key; I[rank] = (uint16_t)lane; }
        WSYNC();
        return;
    }
    const int P = next_pow2(M);
    for (int p = M + lane; p < P; p += WAVE) { K[p] = ~0ull; I[p] = (uint16_t)p; }
    WSYNC();
    Wave t;
    t.skey = K; t.sidx = I; t.lane = lane;
    bitonic_sort(t, P);
}

// ---- what a product shape brings: operand entry sizes, the raw coefficient product of one term pair, simplify()'s
// verdict on a finished sum, and the assembly of centre / radii.
template <int AR, int AC, int BR, int BC>
struct MulPol {
    typedef MulShape<AR, AC, BR, BC> SH;
    static constexpr int ASZ = SH::ASZ, BSZ = SH::BSZ, RAW = SH::SZ, OSZ = SH::SZ, NR = SH::SZ;
    __device__ static inline void raw(const double* ca, const double* cb, double* v) { SH::mul(ca, cb, v); }
    // RT/PZsparse.cu:327-341: keep the monomial, or add |coefficient| to the radius
    __device__ static inline bool finalize(double thr, double thr_sq, const double* v, double* u, double* rad) {
        bool keep;
        if constexpr (OSZ == 1) keep = !norm1_le(v[0], thr);
        else {
            double sq = 0.0;
#pragma unroll
            for (int e = 0; e < OSZ; e++) sq += v[e] * v[e];
            keep = !(sq <= thr_sq);
        }
#pragma unroll
        for (int e = 0; e < OSZ; e++) { u[e] = v[e]; if (!keep) rad[e] += fabs(v[e]); }
        return keep;
    }
    // centre and independent radii (RT/PZsparse.cu:868,944-989); r2 / r3 = |centre| + sum |coefficients| of a / b
    __device__ static inline void finish(const Wave& w, const PZ& out, const View& a, const View& b, const double* r2, const double* r3, const double* rad) {
        double ia[ASZ], ib[BSZ], ia2[ASZ], ib2[BSZ], ca[ASZ], cb[BSZ];
#pragma unroll
        for (int e = 0; e < ASZ; e++) { ia[e] = a.ind[a.off + e]; ia2[e] = a.ind2[a.off + e]; ca[e] = a.cen[a.off + e]; }
#pragma unroll
        for (int e = 0; e < BSZ; e++) { ib[e] = b.ind[b.off + e]; ib2[e] = b.ind2[b.off + e]; cb[e] = b.cen[b.off + e]; }
        double t2[OSZ], t3[OSZ], ii[OSZ], cen[OSZ], base[OSZ], base2[OSZ];
        SH::mul(r2, ib, t2);
        SH::mul(ia, r3, t3);
        SH::mul(ia, ib, ii);
        SH::mul(ca, cb, cen);
#pragma unroll
        for (int e = 0; e < OSZ; e++) base[e] = ii[e] + (t2[e] + t3[e]);
        SH::mul(r2, ib2, t2);
        SH::mul(ia2, r3, t3);
        SH::mul(ia2, ib2, ii);
#pragma unroll
        for (int e = 0; e < OSZ; e++) base2[e] = ii[e] + (t2[e] + t3[e]);
        WSYNC();
        if (w.lane == 0) {
#pragma unroll
            for (int e = 0; e < OSZ; e++) { out.cen[e] = cen[e]; out.ind[e] = base[e] + rad[e]; out.ind2[e] = base2[e] + rad[e]; }
        }
    }
};

// cross(a, b) of 3x1 operands as the reference composes it (RT/PZsparse.cu:1134-1151): six 1x1 products, three differences,
// a stack -- each ending in simplify().  RAW = the six coefficient products of a term pair; the verdict replays the three
// simplify() stages on the six sums of a key (see the note in pz_wave.h above the sorted variant).
struct CrossPol {
    static constexpr int ASZ = 3, BSZ = 3, RAW = 6, OSZ = 3, NR = 12;  // radii: 6 products | 3 differences | 3 stack
    __device__ static inline void raw(const double* ca, const double* cb, double* p6) {
        p6[0] = ca[1] * cb[2]; p6[1] = ca[2] * cb[1];
        p6[2] = ca[2] * cb[0]; p6[3] = ca[0] * cb[2];
        p6[4] = ca[0] * cb[1]; p6[5] = ca[1] * cb[0];
    }
    __device__ static inline bool finalize(double thr, double thr_sq, const double* acc, double* u, double* rad) {
        bool anyc = false, keep = false;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            u[c] = 0.0;
            const double v0 = acc[2 * c], v1 = acc[2 * c + 1];
            const bool h0 = !norm1_le(v0, thr), h1 = !norm1_le(v1, thr);
            if (!h0) rad[2 * c] += fabs(v0);
            if (!h1) rad[2 * c + 1] += fabs(v1);
            if (h0 || h1) {
                double wv = h0 ? 1.0 * v0 : -1.0 * v1;
                if (h0 && h1) wv += -1.0 * v1;
                if (norm1_le(wv, thr)) rad[6 + c] += fabs(wv);
                else { u[c] = wv; anyc = true; }
            }
        }
        if (anyc) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < 3; c++) s += u[c] * u[c];
            keep = !(s <= thr_sq);
            if (!keep) {
#pragma unroll
                for (int c = 0; c < 3; c++) rad[9 + c] += fabs(u[c]);
            }
        }
        return keep;
    }
    __device__ static inline void finish(const Wave& w, const PZ& out, const View& a, const View& b, const double* r2, const double* r3, const double* rad) {
        double cenP[6], baseP[6], base2P[6];
        const int ia_[6] = {1, 2, 2, 0, 0, 1}, ib_[6] = {2, 1, 0, 2, 1, 0};
#pragma unroll
        for (int e = 0; e < 6; e++) {
            const int i = ia_[e], j = ib_[e];
            const double ia = a.ind[i], ib = b.ind[j], ia2 = a.ind2[i], ib2 = b.ind2[j];
            cenP[e] = a.cen[i] * b.cen[j];
            baseP[e] = ia * ib + (r2[i] * ib + ia * r3[j]);
            base2P[e] = ia2 * ib2 + (r2[i] * ib2 + ia2 * r3[j]);
        }
        WSYNC();
        if (w.lane == 0) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const double i0 = baseP[2 * c] + rad[2 * c], i1 = baseP[2 * c + 1] + rad[2 * c + 1];
                const double j0 = base2P[2 * c] + rad[2 * c], j1 = base2P[2 * c + 1] + rad[2 * c + 1];
                const double ir = (i0 * 1.0 + i1 * 1.0) + rad[6 + c], jr = (j0 * 1.0 + j1 * 1.0) + rad[6 + c];
                out.cen[c] = 0.0 + (1.0 * cenP[2 * c] + -1.0 * cenP[2 * c + 1]);
                out.ind[c] = (0.0 + ir) + rad[9 + c];
                out.ind2[c] = (0.0 + jr) + rad[9 + c];
            }
        }
    }
};

// One product.  ROWS_A: a's terms are walked row by row and b's sit in the lanes; otherwise the other way round.  Either
// orientation is correct for any operands; the caller picks the one that fills the lanes (the longer operand in the lanes).
template <class Pol, bool ROWS_A>
__device__ PZW_NOINLINE void prod_hash(Wave& w, const PZ& out, const View& a, const View& b) {
    PROF_CALL_T0
    constexpr int SSZ = ROWS_A ? Pol::ASZ : Pol::BSZ, LSZ = ROWS_A ? Pol::BSZ : Pol::ASZ;
    constexpr int RAW = Pol::RAW, OSZ = Pol::OSZ, NR = Pol::NR;
    // local copies of everything the loops touch (see the note above hash_mark)
    const int lane = w.lane;
    const double thr = w.thr, thr_sq = w.thr_sq;
    const GLB_AS uint64_t* const skeys = ROWS_A ? a.keys : b.keys;
    const GLB_AS uint64_t* const lkeys = ROWS_A ? b.keys : a.keys;
    const GLB_AS double* const scoef = ROWS_A ? a.coef + a.off : b.coef + b.off;
    const GLB_AS double* const lcoef = ROWS_A ? b.coef + b.off : a.coef + a.off;
    const LDS_AS double* const scen = ROWS_A ? a.cen + a.off : b.cen + b.off;
    const LDS_AS double* const lcen = ROWS_A ? b.cen + b.off : a.cen + a.off;
    const int sstride = uni(ROWS_A ? a.stride : b.stride), lstride = uni(ROWS_A ? b.stride : a.stride);
    const int ns1 = uni(ROWS_A ? a.cnt : b.cnt) + 1, nl1 = uni(ROWS_A ? b.cnt : a.cnt) + 1, N = ns1 * nl1 - 1;
    H_LAP0
    const HashLds H = hash_lds(w);
    if (lane == 0 && N > w.lstat[ST_MAX_RAW]) w.lstat[ST_MAX_RAW] = N;
    H_LAP(PR_FILL)
#ifdef P1_PROFILE
    if (lane == 0) { w.prof[PR_CALLS] += 1; w.prof[PR_TERMS] += N; if (N <= 64) w.prof[PR_SMALL] += 1; }
#endif
    hash_mark(w, H, skeys, ns1, lkeys, nl1, N);

    double rad[NR], absS[SSZ], absL[LSZ];
#pragma unroll
    for (int e = 0; e < NR; e++) rad[e] = 0.0;
#pragma unroll
    for (int e = 0; e < SSZ; e++) absS[e] = 0.0;
#pragma unroll
    for (int e = 0; e < LSZ; e++) absL[e] = 0.0;
    const int fcap = uni(H.fcap), scap = uni(H.scap), nval = uni(H.nval);
    LDS_AS uint32_t* const bloom = H.bloom;
    LDS_AS double* const cval = H.cval;
    LDS_AS uint64_t* const FK = H.ckey; LDS_AS uint16_t* const FI = H.cidx;
    LDS_AS uint64_t* const SK = H.ckey + fcap; LDS_AS uint16_t* const SI = H.cidx + fcap;
    int M1 = 0, M2 = 0;
    bool ovf = false;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    { PROF_T0
    // ---- pass B.  Terms of equal key (i1, j1), (i2, j2) with i1 < i2 have j1 > j2 (both key lists ascend), and the loop
    // order below meets (i1, j1) first: rows = a: b's chunks descending, a ascending; rows = b: a's chunks ascending, b descending.
    const int nlc = (nl1 + WAVE - 1) / WAVE, nsc = (ns1 + WAVE - 1) / WAVE;
    for (int lci = 0; lci < nlc && !ovf; lci++) {
        const int lc = (ROWS_A ? nlc - 1 - lci : lci) * WAVE;
        const int lt = lc + lane;
        const bool lvalid = lt < nl1;
        const uint64_t kL = term_key(lkeys, lt, lvalid);
        const uint64_t hL = kL * kHashMul;
        double cL[LSZ];
        if (lvalid && lt > 0) {
            const GLB_AS double* pc = lcoef + (size_t)(lt - 1) * lstride;
#pragma unroll
            for (int e = 0; e < LSZ; e++) { cL[e] = pc[e]; absL[e] += fabs(cL[e]); }
        } else {
#pragma unroll
            for (int e = 0; e < LSZ; e++) cL[e] = lvalid ? lcen[e] : 0.0;
        }
        for (int sci = 0; sci < nsc && !ovf; sci++) {
            const int sc = (ROWS_A ? sci : nsc - 1 - sci) * WAVE;
            const int st = sc + lane;
            const bool svalid = st < ns1;
            const uint64_t kS = term_key(skeys, st, svalid);
            const uint64_t hS = kS * kHashMul;
            double cS[SSZ];
            if (svalid && st > 0) {
                const GLB_AS double* pc = scoef + (size_t)(st - 1) * sstride;
#pragma unroll
                for (int e = 0; e < SSZ; e++) { cS[e] = pc[e]; if (lci == 0) absS[e] += fabs(cS[e]); }
            } else {
#pragma unroll
                for (int e = 0; e < SSZ; e++) cS[e] = svalid ? scen[e] : 0.0;
            }
            const int rows = min(WAVE, ns1 - sc);
#ifdef P1_PROFILE
            { h_t = prof_t0__; H_LAP(PR_E_HEAD) }
#endif
            for (int ri = 0; ri < rows; ri++) {
                const int r = ROWS_A ? ri : rows - 1 - ri;
                const uint64_t key = readlane_u64(kS, r) + kL;   // plain u64 add (RT/PZsparse.cu:938-940)
                const uint64_t h = readlane_u64(hS, r) + hL;
                double cR[SSZ];
#pragma unroll
                for (int e = 0; e < SSZ; e++) cR[e] = readlane_f64(cS[e], r);
                const bool valid = lvalid && !(sc + r == 0 && lt == 0);
                const unsigned bi = bloom_index(h);
                const bool slow = valid && ((bloom[bi >> 5] >> (bi & 31u)) & 1u) != 0u;
                double v[RAW], u[OSZ];
                if constexpr (ROWS_A) Pol::raw(cR, cL, v); else Pol::raw(cL, cR, v);
                bool keep = false;
                if (valid && !slow) keep = Pol::finalize(thr, thr_sq, v, u, rad);
                const unsigned long long mk = __ballot(keep), ms = __ballot(slow);
                const int n1 = __popcll(mk), n2 = __popcll(ms);
                if (M1 + n1 > fcap || M2 + n2 > scap || (M1 + n1) * OSZ + (M2 + n2) * RAW > nval) { ovf = true; break; }
                if (keep) {
                    const int pos = M1 + __popcll(mk & lt_mask);
                    FK[pos] = key; FI[pos] = (uint16_t)pos;
#pragma unroll
                    for (int e = 0; e < OSZ; e++) cval[pos * OSZ + e] = u[e];
                }
                if (slow) {
                    const int q = M2 + __popcll(ms & lt_mask);
                    SK[q] = key; SI[q] = (uint16_t)q;
#pragma unroll
                    for (int e = 0; e < RAW; e++) cval[nval - (q + 1) * RAW + e] = v[e];
                }
                M1 += n1; M2 += n2;
            }
        }
    }
    WSYNC();
    PROF_ADD(PR_EMIT) PROF_ADD(PR_E_COEF) }

    // ---- pass D: the slow list.  Sorted by (key, position) = (key, generation order); the head lane of each run of equal
    // keys adds the members in that order (every lane fetches its own entry, members are shifted down the wave by DPP
    // moves; only the tail of a run that crosses the end of a 64-entry chunk is fetched directly), then simplify()'s verdict.
    if (M2 > 0 && !ovf) {
        PROF_T0
        sort_list(lane, SK, SI, M2);
        for (int base = 0; base < M2 && !ovf; base += WAVE) {
            const int p = base + lane;
            bool head = false;
            uint64_t key = 0;
            double acc[RAW];
#pragma unroll
            for (int e = 0; e < RAW; e++) acc[e] = 0.0;
            if (p < M2) {
                key = SK[p];
                head = (p == 0) || (SK[p - 1] != key);
                const int q = SI[p];
#pragma unroll
                for (int e = 0; e < RAW; e++) acc[e] = cval[nval - (q + 1) * RAW + e];
            }
            {
                double sh[RAW];
#pragma unroll
                for (int e = 0; e < RAW; e++) sh[e] = acc[e];
                for (int j = 1;; j++) {
                    const int pj = p + j;
                    const bool more = head && pj < M2 && SK[pj] == key;
                    if (__ballot(more) == 0ull) break;
#pragma unroll
                    for (int e = 0; e < RAW; e++) sh[e] = dpp_take<0x130, 0xf>(sh[e]);  // wave_shl:1 -- lane l now holds the entry of lane l + j
                    if (more) {
                        double c[RAW];
#pragma unroll
                        for (int e = 0; e < RAW; e++) c[e] = sh[e];
                        if (lane + j >= WAVE) {
                            const int q = SI[pj];
#pragma unroll
                            for (int e = 0; e < RAW; e++) c[e] = cval[nval - (q + 1) * RAW + e];
                        }
#pragma unroll
                        for (int e = 0; e < RAW; e++) acc[e] += c[e];
                    }
                }
            }
            double u[OSZ];
            bool keep = false;
            if (head) keep = Pol::finalize(thr, thr_sq, acc, u, rad);
            const unsigned long long mk = __ballot(keep);
            const int n1 = __popcll(mk);
            if (M1 + n1 > fcap || (M1 + n1) * OSZ + M2 * RAW > nval) { ovf = true; break; }
            if (keep) {
                const int pos = M1 + __popcll(mk & lt_mask);
                FK[pos] = key; FI[pos] = (uint16_t)pos;
#pragma unroll
                for (int e = 0; e < OSZ; e++) cval[pos * OSZ + e] = u[e];
            }
            M1 += n1;
        }
        WSYNC();
        PROF_ADD(PR_SORT) PROF_ADD(PR_S_BITONIC)
    }

    // ---- pass E: the final list in key order -> the result slot
    if (ovf) { flag(w, ERR_RAW_OVERFLOW); M1 = 0; }
    { PROF_T0
#ifdef P1_PROFILE
    h_t = clock64();
#endif
    sort_list(lane, FK, FI, M1);
    H_LAP(PR_E_RUN)
    const int ocap = uni(out.cap);
    if (M1 > ocap) { flag(w, ERR_SLOT_OVERFLOW); M1 = ocap; }
    GLB_AS uint64_t* const okeys = out.keys;
    GLB_AS double* const ocoef = out.coef;
    for (int p = lane; p < M1; p += WAVE) {
        const int src = FI[p];
        okeys[p] = FK[p];
#pragma unroll
        for (int e = 0; e < OSZ; e++) ocoef[(size_t)p * OSZ + e] = cval[src * OSZ + e];
    }
    PROF_ADD(PR_EMIT) PROF_ADD(PR_E_STORE) }

    // ---- centre and radii
    double r2[Pol::ASZ], r3[Pol::BSZ];
    { PROF_T0
#pragma unroll
    for (int e = 0; e < NR; e++) rad[e] = wave_sum(rad[e]);
    double* absA = ROWS_A ? absS : absL;
    double* absB = ROWS_A ? absL : absS;
#pragma unroll
    for (int e = 0; e < Pol::ASZ; e++) r2[e] = fabs(a.cen[a.off + e]) + wave_sum(absA[e]);
#pragma unroll
    for (int e = 0; e < Pol::BSZ; e++) r3[e] = fabs(b.cen[b.off + e]) + wave_sum(absB[e]);
    PROF_ADD(PR_ABS) }
#ifdef P1_PROFILE
    h_t = clock64();
#endif
    Pol::finish(w, out, a, b, r2, r3, rad);
    H_LAP(PR_E_PRUNE)
    if (lane == 0) {
        w.cnt[out.id] = M1;
        if (M1 > w.lstat[ST_MAX_OUT]) w.lstat[ST_MAX_OUT] = M1;
    }
    WSYNC();
    PROF_CALL_END(N)
}

// A product whose left operand has no monomials (mass, inertia, the fixed rpy rotation times a PZ): term m is
// `centre_a * coef_b[m]` under b's m-th key -- already in key order with unique keys, so one ordered pass.
template <int AR, int AC, int BR, int BC>
__device__ PZW_NOINLINE void mul_const_left(Wave& w, const PZ& out, const View& a, const View& b) {
    PROF_CALL_T0
    typedef MulPol<AR, AC, BR, BC> Pol;
    constexpr int OSZ = Pol::OSZ;
    const int N = uni(b.cnt), lane = w.lane, ocap = uni(out.cap), bstride = uni(b.stride);
    const double thr = w.thr, thr_sq = w.thr_sq;
    const GLB_AS uint64_t* const bkeys = b.keys;
    const GLB_AS double* const bcoef = b.coef + b.off;
    GLB_AS uint64_t* const okeys = out.keys;
    GLB_AS double* const ocoef = out.coef;
    if (lane == 0 && N > w.lstat[ST_MAX_RAW]) w.lstat[ST_MAX_RAW] = N;
#ifdef P1_PROFILE
    if (lane == 0) { w.prof[PR_CALLS] += 1; w.prof[PR_TERMS] += N; if (N <= 64) w.prof[PR_SMALL] += 1; }
#endif
    double ca[Pol::ASZ], rad[OSZ], absB[Pol::BSZ], r2[Pol::ASZ], r3[Pol::BSZ];
#pragma unroll
    for (int e = 0; e < Pol::ASZ; e++) { ca[e] = a.cen[a.off + e]; r2[e] = fabs(ca[e]); }
#pragma unroll
    for (int e = 0; e < OSZ; e++) rad[e] = 0.0;
#pragma unroll
    for (int e = 0; e < Pol::BSZ; e++) absB[e] = 0.0;
    int emitted = 0;
    for (int base = 0; base < N; base += WAVE) {
        const int m = base + lane;
        bool keep = false;
        uint64_t key = 0;
        double u[OSZ];
        if (m < N) {
            key = bkeys[m];
            const GLB_AS double* pc = bcoef + (size_t)m * bstride;
            double cb[Pol::BSZ], v[OSZ];
#pragma unroll
            for (int e = 0; e < Pol::BSZ; e++) { cb[e] = pc[e]; absB[e] += fabs(cb[e]); }
            Pol::raw(ca, cb, v);
            keep = Pol::finalize(thr, thr_sq, v, u, rad);
        }
        const unsigned long long mk = __ballot(keep);
        if (keep) {
            const int pos = emitted + __popcll(mk & ((1ull << lane) - 1ull));
            if (pos < ocap) {
                okeys[pos] = key;
#pragma unroll
                for (int e = 0; e < OSZ; e++) ocoef[(size_t)pos * OSZ + e] = u[e];
            }
        }
        emitted += __popcll(mk);
    }
    if (emitted > ocap) { flag(w, ERR_SLOT_OVERFLOW); emitted = ocap; }
#pragma unroll
    for (int e = 0; e < OSZ; e++) rad[e] = wave_sum(rad[e]);
#pragma unroll
    for (int e = 0; e < Pol::BSZ; e++) r3[e] = fabs(b.cen[b.off + e]) + wave_sum(absB[e]);
    Pol::finish(w, out, a, b, r2, r3, rad);
    if (lane == 0) {
        w.cnt[out.id] = emitted;
        if (emitted > w.lstat[ST_MAX_OUT]) w.lstat[ST_MAX_OUT] = emitted;
    }
    WSYNC();
    PROF_CALL_END(N)
}

// operator* (RT/PZsparse.cu:864-994).  PREFER_ROWS_A picks the orientation the chain's operands of this shape have
// (the rotation / the joint's own factor is the short one); it only matters for speed.
template <int AR, int AC, int BR, int BC>
__device__ inline void mul(Wave& w, const PZ& out, const View& a, const View& b) {
    if (a.cnt == 0) { mul_const_left<AR, AC, BR, BC>(w, out, a, b); return; }
    constexpr bool kRowsA = !(AR == 3 && AC == 3 && BR == 3 && BC == 3);  // 3x3 * 3x3: the accumulated rotation (a) grows, the joint's rotation (b) is short
    prod_hash<MulPol<AR, AC, BR, BC>, kRowsA>(w, out, a, b);
}

__device__ inline void cross_pzpz(Wave& w, const PZ& out, const View& a, const View& b) {
    if (a.cnt <= b.cnt) prod_hash<CrossPol, true>(w, out, a, b);
    else prod_hash<CrossPol, false>(w, out, a, b);
}
