// Wave-level sparse polynomial-zonotope arithmetic for gfx950 (one 64-lane wavefront = one PZ "thread of
// control").  MI355X-native counterpart of the reference's host class PZsparse (RT/PZsparse.h:50-183,
// RT/PZsparse.cu:284-1167), designed around the wavefront instead of std::vector<Monomial>:
//
//   * a PZ lives in a per-wave slot of a global-memory arena as structure-of-arrays
//     {keys u64[cap], coef f64[cap][sz], centre[sz], indep[sz]} with its monomials kept sorted by key and
//     unique (the invariant PZsparse::simplify establishes, RT/PZsparse.cu:284-350);
//   * every operator that the reference ends with simplify() (operator*, +, -, +=, addOneDimPZ, stack,
//     constructors; RT/PZsparse.cu:135,204,761,808,831,991,1084,1113) is ONE pass of
//         generate (key, term-index) pairs into LDS -> 64-lane bitonic sort on (key, index) ->
//         segmented sum of equal keys in index order -> Frobenius-norm prune against SIMPLIFY_THRESHOLD ->
//         ballot/popcount compaction into the destination slot;
//     coefficients of raw terms are never materialised: the segmented sum recomputes coeff_i * coeff_j (or the
//     scaled / embedded source coefficient) from the operands, so only 10 B per raw term touch LDS;
//   * summation of equal keys is in generation order (a stable order); the reference's std::sort leaves that
//     order unspecified, so last-bit differences against any CPU build are inherent (hard parts).
//
// Control flow is wave-uniform everywhere; the kernels that use this header launch 64-thread workgroups, so
// __syncthreads() is a wave-local fence + barrier.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pzw {

constexpr int WAVE = 64;

// status words (global, one array per launch)
enum { ST_ERR = 0, ST_MAX_RAW = 1, ST_MAX_OUT = 2, ST_WORDS = 4 };
enum { ERR_RAW_OVERFLOW = 1, ERR_SLOT_OVERFLOW = 2, ERR_TABLE_OVERFLOW = 4, ERR_LINK_GENS = 8 };

// A PZ slot (all fields wave-uniform).  id indexes the per-wave LDS count table.
struct PZ {
    uint64_t* keys;
    double* coef;  // [cap][sz]
    double* cen;   // [sz]
    double* ind;   // [sz]
    int sz, cap, id;
};

// Read view of a PZ or of one entry of it (RT/PZsparse.cu:678-697 operator()(r,c) without the copy).
struct View {
    const uint64_t* keys;
    const double* coef;
    const double* cen;
    const double* ind;
    int cnt, stride, off, sz;
};

struct Wave {
    uint64_t* skey;   // LDS [cap_raw]
    uint16_t* sidx;   // LDS [cap_raw]
    int* cnt;         // LDS per-slot monomial counts
    int cap_raw;
    double thr;       // SIMPLIFY_THRESHOLD
    int* lstat;       // LDS [ST_WORDS]: error bits / max raw terms / max monomials of this wave (flushed once per launch)
    int lane;
};

__device__ inline View view(const Wave& w, const PZ& p) { return View{p.keys, p.coef, p.cen, p.ind, w.cnt[p.id], p.sz, 0, p.sz}; }
__device__ inline View elem(const Wave& w, const PZ& p, int r) { return View{p.keys, p.coef, p.cen, p.ind, w.cnt[p.id], p.sz, r, 1}; }

__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ inline int next_pow2(int v) { int p = 64; while (p < v) p <<= 1; return p; }

__device__ inline void flag(const Wave& w, int bit) { if (w.lane == 0) w.lstat[ST_ERR] |= bit; }

// 64-lane bitonic sort of (skey, sidx)[0, P) ascending by (key, idx); P is a power of two >= 64.
__device__ inline void bitonic_sort(const Wave& w, int P) {
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = w.lane; t < (P >> 1); t += WAVE) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const uint64_t ka = w.skey[i], kb = w.skey[l];
                const uint16_t ia = w.sidx[i], ib = w.sidx[l];
                const bool gt = (ka > kb) || (ka == kb && ia > ib);
                const bool up = (i & k) == 0;
                if (gt == up) { w.skey[i] = kb; w.skey[l] = ka; w.sidx[i] = ib; w.sidx[l] = ia; }
            }
            __syncthreads();
        }
    }
}

// Sort the N raw terms described by `ev`, sum equal keys, prune small coefficients into the independent part
// and write the result to `out` (RT/PZsparse.cu:284-350).  base_ind = independent part before pruning.
// Eval: uint64_t key(int idx) const; void coef(int idx, double* c /*[SZ]*/) const.
template <int SZ, class Eval>
__device__ inline void sort_reduce_emit(Wave& w, int N, const Eval& ev, const PZ& out, const double* base_ind) {
    int emitted = 0;
    double ra[SZ];
#pragma unroll
    for (int e = 0; e < SZ; e++) ra[e] = 0.0;
    if (N > w.cap_raw) { flag(w, ERR_RAW_OVERFLOW); N = 0; }
    if (w.lane == 0 && N > w.lstat[ST_MAX_RAW]) w.lstat[ST_MAX_RAW] = N;
    if (N > 0) {
        const int P = next_pow2(N);
        for (int p = w.lane; p < P; p += WAVE) {
            w.skey[p] = p < N ? ev.key(p) : ~0ull;
            w.sidx[p] = (uint16_t)p;
        }
        __syncthreads();
        bitonic_sort(w, P);
        for (int base = 0; base < N; base += WAVE) {
            const int p = base + w.lane;
            bool head = false, keep = false;
            uint64_t key = 0;
            double acc[SZ];
            if (p < N) {
                key = w.skey[p];
                head = (p == 0) || (w.skey[p - 1] != key);
            }
            if (head) {
                ev.coef(w.sidx[p], acc);
                for (int q = p + 1; q < N && w.skey[q] == key; q++) {
                    double c[SZ];
                    ev.coef(w.sidx[q], c);
#pragma unroll
                    for (int e = 0; e < SZ; e++) acc[e] += c[e];
                }
                double s = 0.0;
#pragma unroll
                for (int e = 0; e < SZ; e++) s += acc[e] * acc[e];
                keep = !(sqrt(s) <= w.thr);
                if (!keep) {
#pragma unroll
                    for (int e = 0; e < SZ; e++) ra[e] += fabs(acc[e]);
                }
            }
            const unsigned long long m = __ballot(keep);
            if (keep) {
                const int pos = emitted + __popcll(m & ((1ull << w.lane) - 1ull));
                if (pos < out.cap) {
                    out.keys[pos] = key;
#pragma unroll
                    for (int e = 0; e < SZ; e++) out.coef[(size_t)pos * SZ + e] = acc[e];
                }
            }
            emitted += __popcll(m);
        }
    }
    if (emitted > out.cap) { flag(w, ERR_SLOT_OVERFLOW); emitted = out.cap; }
#pragma unroll
    for (int e = 0; e < SZ; e++) ra[e] = wave_sum(ra[e]);
    if (w.lane == 0) {
#pragma unroll
        for (int e = 0; e < SZ; e++) out.ind[e] = base_ind[e] + ra[e];
        w.cnt[out.id] = emitted;
        if (emitted > w.lstat[ST_MAX_OUT]) w.lstat[ST_MAX_OUT] = emitted;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------
// linear combination: out = sum_s scale_s * embed(src_s).  Covers operator+, operator-, operator+=,
// double*PZ followed by +/-, addOneDimPZ and stack (RT/PZsparse.cu:743-834,996-1030,1068-1116).
// A source with sz == 1 and comp >= 0 is a 1x1 PZ embedded into entry `comp` of the result.
struct Seg {
    View v;
    double scale;
    int comp;  // -1: same shape as the result
};

template <int SZ, int NS>
struct LinEval {
    Seg s[NS];
    int off[NS + 1];
    __device__ inline int seg_of(int idx) const {
        int k = 0;
#pragma unroll
        for (int i = 1; i < NS; i++) k += (idx >= off[i]) ? 1 : 0;
        return k;
    }
    __device__ inline uint64_t key(int idx) const {
        const int k = seg_of(idx);
        return s[k].v.keys[idx - off[k]];
    }
    __device__ inline void coef(int idx, double* c) const {
        const int k = seg_of(idx);
        const Seg& g = s[k];
        const double* src = g.v.coef + (size_t)(idx - off[k]) * g.v.stride + g.v.off;
        if (g.comp < 0) {
#pragma unroll
            for (int e = 0; e < SZ; e++) c[e] = g.scale * src[e];
        } else {
            const double v = g.scale * src[0];
#pragma unroll
            for (int e = 0; e < SZ; e++) c[e] = (e == g.comp) ? v : 0.0;
        }
    }
};

template <int SZ, int NS>
__device__ inline void lincomb(Wave& w, const PZ& out, const Seg* segs) {
    LinEval<SZ, NS> ev;
    int N = 0;
    double cen[SZ], ind[SZ];
#pragma unroll
    for (int e = 0; e < SZ; e++) { cen[e] = 0.0; ind[e] = 0.0; }
#pragma unroll
    for (int k = 0; k < NS; k++) {
        ev.s[k] = segs[k];
        ev.off[k] = N;
        N += segs[k].v.cnt;
        const View& v = segs[k].v;
        const double sc = segs[k].scale, asc = fabs(sc);
        if (segs[k].comp < 0) {
#pragma unroll
            for (int e = 0; e < SZ; e++) {
                const double c = sc * v.cen[v.off + e], i = v.ind[v.off + e] * asc;
                cen[e] = (k == 0) ? c : cen[e] + c;
                ind[e] = (k == 0) ? i : ind[e] + i;
            }
        } else {
            const double c = sc * v.cen[v.off], i = v.ind[v.off] * asc;
#pragma unroll
            for (int e = 0; e < SZ; e++)
                if (e == segs[k].comp) { cen[e] = cen[e] + c; ind[e] = ind[e] + i; }
        }
    }
    ev.off[NS] = N;
    __syncthreads();  // all lanes have read the sources' centre / indep before `out` (possibly aliasing) is written
    if (w.lane == 0) {
#pragma unroll
        for (int e = 0; e < SZ; e++) out.cen[e] = cen[e];
    }
    sort_reduce_emit<SZ>(w, N, ev, out, ind);
}

// ---------------------------------------------------------------------------------------------------
// product (RT/PZsparse.cu:864-994).  Shapes: A is AR x AC (AR = AC = 1: scalar broadcast), B is AC x BC
// (or any shape when A is 1x1); the result has SZ = (A scalar ? BSZ : AR*BC) entries.
template <int AR, int AC, int BR, int BC>
struct MulShape {
    static constexpr bool a11 = (AR == 1 && AC == 1);
    static constexpr int ASZ = AR * AC, BSZ = BR * BC;
    static constexpr int OR_ = a11 ? BR : AR, OC = a11 ? BC : BC, SZ = OR_ * OC;
    __device__ static inline void mul(const double* A, const double* B, double* O) {
        if (a11) {
#pragma unroll
            for (int e = 0; e < BSZ; e++) O[e] = A[0] * B[e];
        } else {
#pragma unroll
            for (int r = 0; r < AR; r++)
#pragma unroll
                for (int c = 0; c < BC; c++) {
                    double s = 0.0;
#pragma unroll
                    for (int k = 0; k < AC; k++) s += A[r * AC + k] * B[k * BC + c];
                    O[r * BC + c] = s;
                }
        }
    }
};

template <class SH>
struct MulEval {
    View a, b;
    int mb1;
    __device__ inline void split(int idx, int& i, int& j) const {
        const int t = idx + 1;  // (0,0) = centre*centre is not a monomial
        i = t / mb1;
        j = t - i * mb1;
    }
    __device__ inline uint64_t key(int idx) const {
        int i, j;
        split(idx, i, j);
        return (i ? a.keys[i - 1] : 0ull) + (j ? b.keys[j - 1] : 0ull);  // plain u64 add (RT/PZsparse.cu:938-940)
    }
    __device__ inline void coef(int idx, double* c) const {
        int i, j;
        split(idx, i, j);
        double ca[SH::ASZ], cb[SH::BSZ];
        const double* pa = i ? a.coef + (size_t)(i - 1) * a.stride + a.off : a.cen + a.off;
        const double* pb = j ? b.coef + (size_t)(j - 1) * b.stride + b.off : b.cen + b.off;
#pragma unroll
        for (int e = 0; e < SH::ASZ; e++) ca[e] = pa[e];
#pragma unroll
        for (int e = 0; e < SH::BSZ; e++) cb[e] = pb[e];
        SH::mul(ca, cb, c);
    }
};

// |centre| + sum |coef| over a view, entry-wise (RT/PZsparse.cu:945-949,962-966)
template <int SZ>
__device__ inline void abs_sum(const Wave& w, const View& v, double* r) {
#pragma unroll
    for (int e = 0; e < SZ; e++) r[e] = 0.0;
    for (int m = w.lane; m < v.cnt; m += WAVE) {
        const double* c = v.coef + (size_t)m * v.stride + v.off;
#pragma unroll
        for (int e = 0; e < SZ; e++) r[e] += fabs(c[e]);
    }
#pragma unroll
    for (int e = 0; e < SZ; e++) r[e] = fabs(v.cen[v.off + e]) + wave_sum(r[e]);
}

template <int AR, int AC, int BR, int BC>
__device__ inline void mul(Wave& w, const PZ& out, const View& a, const View& b) {
    typedef MulShape<AR, AC, BR, BC> SH;
    MulEval<SH> ev;
    ev.a = a; ev.b = b; ev.mb1 = b.cnt + 1;
    const int N = (a.cnt + 1) * (b.cnt + 1) - 1;
    double r2[SH::ASZ], r3[SH::BSZ], ia[SH::ASZ], ib[SH::BSZ], ca[SH::ASZ], cb[SH::BSZ];
    abs_sum<SH::ASZ>(w, a, r2);
    abs_sum<SH::BSZ>(w, b, r3);
#pragma unroll
    for (int e = 0; e < SH::ASZ; e++) { ia[e] = a.ind[a.off + e]; ca[e] = a.cen[a.off + e]; }
#pragma unroll
    for (int e = 0; e < SH::BSZ; e++) { ib[e] = b.ind[b.off + e]; cb[e] = b.cen[b.off + e]; }
    double t2[SH::SZ], t3[SH::SZ], ii[SH::SZ], cen[SH::SZ], base[SH::SZ];
    SH::mul(r2, ib, t2);   // (|c_a| + sum|coef_a|) * indep_b
    SH::mul(ia, r3, t3);   // indep_a * (|c_b| + sum|coef_b|)
    SH::mul(ia, ib, ii);
    SH::mul(ca, cb, cen);
#pragma unroll
    for (int e = 0; e < SH::SZ; e++) base[e] = ii[e] + (t2[e] + t3[e]);
    __syncthreads();
    if (w.lane == 0) {
#pragma unroll
        for (int e = 0; e < SH::SZ; e++) out.cen[e] = cen[e];
    }
    sort_reduce_emit<SH::SZ>(w, N, ev, out, base);
}

// out = a^T for 3x3 (RT/PZsparse.cu:1050-1066); keys unchanged, no simplify.
__device__ inline void transpose33(Wave& w, const PZ& out, const PZ& a) {
    const int n = w.cnt[a.id];
    for (int t = w.lane; t < n * 9; t += WAVE) {
        const int m = t / 9, e = t - m * 9, r = e / 3, c = e - r * 3;
        out.coef[(size_t)m * 9 + c * 3 + r] = a.coef[(size_t)m * 9 + e];
    }
    for (int m = w.lane; m < n; m += WAVE) out.keys[m] = a.keys[m];
    if (w.lane < 9) {
        const int r = w.lane / 3, c = w.lane - r * 3;
        out.cen[c * 3 + r] = a.cen[w.lane];
        out.ind[c * 3 + r] = a.ind[w.lane];
    }
    if (w.lane == 0) w.cnt[out.id] = n;
    __syncthreads();
}

// constant PZ (centre + independent radius, no monomials): RT/PZsparse.cu:66-98
__device__ inline void set_const(Wave& w, const PZ& out, const double* cen, const double* ind) {
    if (w.lane < out.sz) {
        out.cen[w.lane] = cen ? cen[w.lane] : 0.0;
        out.ind[w.lane] = ind ? ind[w.lane] : 0.0;
    }
    if (w.lane == 0) w.cnt[out.id] = 0;
    __syncthreads();
}

// plain copy (operator=)
__device__ inline void copy(Wave& w, const PZ& out, const PZ& a) {
    const int n = w.cnt[a.id], sz = a.sz;
    for (int t = w.lane; t < n * sz; t += WAVE) out.coef[t] = a.coef[t];
    for (int m = w.lane; m < n; m += WAVE) out.keys[m] = a.keys[m];
    if (w.lane < sz) { out.cen[w.lane] = a.cen[w.lane]; out.ind[w.lane] = a.ind[w.lane]; }
    if (w.lane == 0) w.cnt[out.id] = n;
    __syncthreads();
}

}  // namespace pzw
