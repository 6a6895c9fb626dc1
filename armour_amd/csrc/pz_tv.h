// Time-vectorised polynomial-zonotope arithmetic for gfx950: one 64-lane wavefront carries 64 TIME STEPS of one planning
// problem through the reference's operator sequence at once (lane = time step).
//
// Why.  The reference (RT/armour_main.cu:96-216) runs the same operator chain for every time interval; so did rounds 1-2 of
// this repository, one wavefront per (problem, time step), with the 64 lanes sharing the SYMBOLIC work of one step (sorting
// raw monomial keys).  But the monomial sets of neighbouring time steps are nearly the same -- measured on the reference's
// own operator sequence, the union over 64 steps of an operator's result keys is 1.39 x the size of one step's -- while the
// symbolic work (sort, run detection: ~2000 wave instructions per 64 raw terms) dwarfs the arithmetic.  Here the symbolic
// work is done ONCE per 64 time steps on the union of the key sets, and the arithmetic runs in the lanes:
//
//   * a PZ is {keys u64[cap] (one sorted unique list for all lanes), coef f64[cap][sz][64], and per-lane header rows
//     centre[sz][64], indep[sz][64], indep2[sz][64], asum[sz][64]} -- every "row" is 64 consecutive doubles, one per time
//     step, so all arithmetic loads and stores are 512-byte coalesced and no value ever crosses lanes;
//   * a monomial that time step t does not have (never generated there, or pruned there by simplify()) simply carries
//     coefficient 0 in lane t.  That reproduces the per-step results EXACTLY: a zero factor makes a raw term 0, adding 0
//     changes no sum, a sum of zeros is pruned and adds 0 to the radius -- so every lane computes what RT/PZsparse.cu
//     computes for its time step, with the surviving terms added in the same (generation) order;
//   * simplify()'s verdict (RT/PZsparse.cu:327-341) is taken per lane; a key stays in the union while ANY lane keeps it;
//   * asum[e] = sum |coefficient| of entry e, accumulated by the producer in monomial order: the |centre| + sum |coef|
//     term of the product radius (RT/PZsparse.cu:945-966) is then four row loads instead of a pass over both operands.
//
// The sort of the raw keys is pz_wave.h's (sort_terms with its three sorters) on the union key lists; what follows it here
// is a serial, wave-uniform walk over the sorted raw terms whose loads are software-pipelined U entries ahead.
#pragma once
#include "pz_wave.h"

namespace tv {

using pzw::WAVE;
using pzw::Wave;

#define TV_NOINLINE __attribute__((noinline))
#if defined(TV_PROFILE_FULL) && !defined(TV_PROFILE)
#define TV_PROFILE   // -DTV_PROFILE: per-wave stamps and waits only; -DTV_PROFILE_FULL: also cycles per operator type (slows the walks)
#endif

struct TPZ {
    GLB_AS pzkey_t* keys;
    GLB_AS double* coef;  // [cap][sz][64]
    GLB_AS double* hdr;   // [4][sz][64]: indep, indep2, asum, centre
    int sz, cap, id;
};
struct TView {
    const GLB_AS pzkey_t* keys;
    const GLB_AS double* coef;
    const GLB_AS double* hdr;
    int cnt, stride, off, sz;  // stride = rows per monomial of the underlying PZ, off = first row of this view, sz = rows of the view
};
enum { H_IND = 0, H_IND2 = 1, H_ASUM = 2, H_CEN = 3 };  // the centre rows come LAST, i.e. directly before the coefficient rows: the centre is "monomial -1" (no branch in the row loads)

struct TW {
    Wave w;        // sort buffers, thresholds, lane, per-wave status, LDS count table (pz_wave.h)
    bool active;   // the lane takes part in the arithmetic: always, since round 5 (a lane without a step of its own shadows the group's last step)
    bool live;     // this lane's time step exists and is its own (lane < the group's step count): the lanes that write the final tables
    int rl;        // the lane's place in a row: the lane itself, or -- beyond the row's width GR -- the place of the lane it shadows
    LDS_AS double* stage;  // LDS staging area for the rows of a product's SHORT operand (stage_rows below)
    int stage_rows;        // its capacity in rows of 64 doubles
    // walk helper (below, "One walk on two waves"): the channel to a wave that has nothing of its own to do right now, or nullptr
    LDS_AS int* hch = nullptr;
    int hseq = 0;          // jobs posted (primary) / served (helper) on that channel so far
    int hnum = 16;         // the primary keeps hnum / 32 of a shared walk's terms
    int hmin = 192;        // walks with fewer sorted terms stay on one wave (two hand-overs cost more than half of such a walk)
    bool hded = false;     // the channel belongs to a DEDICATED helper wave (eight-wave blocks, below): it serves whatever is posted, nothing is counted
#ifdef TV_PROFILE  // development: cycles in the sorts, in the walks, raw terms walked, operator calls
    long long c_sort = 0, c_walk = 0, c_cc = 0, n_raw = 0, n_calls = 0, n_emit = 0, c_wait = 0, c_fwd = 0, c_wait_fwd = 0, c_hwait = 0, n_shared = 0, n_shared_terms = 0;
    long long c_type[3] = {0, 0, 0}, n_type[3] = {0, 0, 0};  // walk cycles / raw terms of mul, cross, sums
    long long c_fn[8] = {0, 0, 0, 0, 0, 0, 0, 0}, n_fn[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // whole-call cycles / calls: 0 sorted product, 1 product with a constant left operand, 2 cross, 3 sum, 4 constant cross, 5 helper service, 6 set / transpose, 7 link tables
#endif
};
#ifdef TV_PROFILE
struct TvFn {   // whole-call time of an operator, by kind (the kind may be set late)
    TW& t; int kind; long long t0;
    __device__ TvFn(TW& t_, int k) : t(t_), kind(k), t0(clock64()) {}
    __device__ ~TvFn() { t.c_fn[kind] += clock64() - t0; t.n_fn[kind] += 1; }
};
#define TVP_FN(t, k) ::tv::TvFn tvfn__(t, k);
#define TVP_FN_KIND(k) tvfn__.kind = (k);
#else
#define TVP_FN(t, k)
#define TVP_FN_KIND(k)
#endif
#ifdef TV_PROFILE
#define TVP_T0 const long long tvp0__ = clock64();
#define TVP_T1 const long long tvp1__ = clock64();
#define TVP_END(t, N, E, TY) { const long long tvp2__ = clock64(); (t).c_sort += tvp1__ - tvp0__; (t).c_walk += tvp2__ - tvp1__; (t).n_raw += (N); (t).n_calls += 1; (t).n_emit += (E); (t).c_type[TY] += tvp2__ - tvp1__; (t).n_type[TY] += (N); }
#else
#define TVP_T0
#define TVP_T1
#define TVP_END(t, N, E, TY)
#endif

__device__ inline int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Rows in global memory.  A row holds one double per time step of the group, GR doubles apart from the next row: GR = 64 (a row = the wave's
// 64 lanes, four 128-byte lines) or -- round 5 -- the group's own lane count rounded up to even (TV_GROW: T = 100 makes groups of 50 time steps,
// rows of 400 bytes packed back to back).  The fabric moves whole 128-byte lines: with 512-byte rows of which 400 bytes are used, every fourth
// line was fetched and written back for two lanes (tools/micro/row_gather.hip: 50 lanes of 64 cost a full row); packed, consecutive rows share
// their lines, so a pass over an operand in its own order moves 3.125 lines a row instead of 4.
// A lane without a time step of its own (lane >= the group's step count nl <= GR) SHADOWS step (lane mod nl) of its group: same inputs, same
// arithmetic, hence the same values -- stored to a place of its own while the row has one (lane < GR), to the shadowed lane's place otherwise
// (TW::rl; the same value to the same address from two lanes).  No lane is masked anywhere in the arithmetic, a shadow's votes in the ballots
// repeat the shadowed step's, and only the final tables are written by the lanes that own a step (TW::live).  (Rounds 3-4 kept such lanes
// inactive, with zeros in places of their own in every row.)  The rows staged in LDS keep the wave's stride.
#ifndef TV_GROW
#define TV_GROW 64
#endif
constexpr int GR = TV_GROW;
static_assert(GR >= 2 && GR <= WAVE, "a row holds at most one element per lane");

// ---------------------------------------------------------------------------------------------------------------------
// One walk on two waves.  The serial walk over the sorted raw terms is what an operator costs (the sort is 5 %), and in the
// backward pass of the RNEA two of a block's four waves have nothing to do while the f- and the n-recursion -- one product and
// one sum per joint each, strictly one after the other -- are what the pass takes.  There the wave that owns an operator (the
// PRIMARY) sorts as before, then hands the upper part [S, N) of the sorted terms to an idle HELPER wave through a channel of words
// in LDS and walks [0, S) itself.  S sits on a boundary between runs of equal keys, so no sum crosses it.  The helper walks its part
// with the same code on the primary's sorted arrays (and staged rows) in LDS, emits into a scratch slot of its own, and -- once the
// primary has posted how many terms IT kept -- moves its rows down behind them in the result; the primary then adds the helper's
// partial sums of the pruned amounts (and of |coefficient|) to its own.  Keys, coefficients and centres are exactly those of the
// one-wave walk; the radii add the same numbers as two partial sums instead of one running sum (<= 1e-12, the tolerance the
// time-vectorised build already states against the per-step one).
enum { HJ_SEQ = 0, HJ_WALKED, HJ_N0SEQ, HJ_DONE,           // progress words (sequence numbers)
       HJ_KIND, HJ_S, HJ_N, HJ_INDIRECT, HJ_N0, HJ_NH,
       HJ_SKEY, HJ_SIDX, HJ_STAGE,                          // the primary's LDS buffers (addresses)
       HJ_OUT_KEYS, HJ_OUT_COEF = HJ_OUT_KEYS + 2, HJ_OUT_CAP = HJ_OUT_COEF + 2,
       HJ_HDR = HJ_OUT_CAP + 1,                             // the helper's scratch slot: header rows (partial sums) and coefficient rows, written by the helper
       HJ_TMP_COEF = HJ_HDR + 2,
       HJ_SEG0 = 24, HJ_SEG_WORDS = 8,                      // per source: coef address (2), cnt, stride, off, comp, scale (2)
       HJ_WORDS = HJ_SEG0 + 4 * HJ_SEG_WORDS };
enum { HK_NONE = 0, HK_MUL_3331_A_STAGED = 1, HK_LIN2 = 2, HK_LIN4_CHAIN = 3,
       // kinds only a dedicated helper serves (its scratch slot has room for their partial sums)
       HK_MUL_3331_UNSTAGED = 4, HK_MUL_3331_B_STAGED = 5, HK_LIN3_CHAIN = 6, HK_CROSS_UNSTAGED = 7, HK_CROSS_A_STAGED = 8, HK_CROSS_B_STAGED = 9, HK_CROSS_CONST = 10,
       HK_BAR = 100, HK_EXIT = 101 };   // control: join the block barrier the primary is about to enter / this item is over
// Dedicated helpers.  A block of EIGHT waves -- two per SIMD, which the operators' 256 vector registers allow -- gives each of the four role
// waves a helper of its own for the whole item: the helper sits in serve_loop() on its primary's channel, walks the upper part of every
// operator large enough to share, mirrors the primary's block barriers (HK_BAR) and leaves at HK_EXIT.  What a lone wave cannot hide -- the
// latency between its own dependent instructions, an LDS read, a row load -- the SIMD fills with the other wave's instructions.

__device__ inline int lds_ld(LDS_AS int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void lds_st_ptr(LDS_AS int* p, const GLB_AS void* q) { const uint64_t v = (uint64_t)q; p[0] = (int)(unsigned)v; p[1] = (int)(unsigned)(v >> 32); }
template <class T>
__device__ inline T* lds_ld_ptr(LDS_AS int* p) { return (T*)(((uint64_t)(unsigned)p[1] << 32) | (unsigned)p[0]); }
// a pointer every lane holds alike, moved to scalar registers: row addresses are then scalar arithmetic plus a lane offset
template <class T>
__device__ inline T* uni_ptr(T* p) {
    const uint64_t v = (uint64_t)p;
    return (T*)(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v));
}
__device__ inline uint64_t readlane_u64(uint64_t v, int l) {
    return ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)v, l);
}

__device__ inline TView view(const TW& t, const TPZ& p) { return TView{p.keys, p.coef, p.hdr, t.w.cnt[p.id], p.sz, 0, p.sz}; }
__device__ inline TView elem(const TW& t, const TPZ& p, int r) { return TView{p.keys, p.coef, p.hdr, t.w.cnt[p.id], p.sz, r, 1}; }
__device__ inline pzw::View kview(const TView& v) {  // what the sorters need: keys and count
    pzw::View k;
    k.keys = v.keys; k.coef = nullptr; k.cen = nullptr; k.ind = nullptr; k.ind2 = nullptr;
    k.cnt = v.cnt; k.stride = v.stride; k.off = v.off; k.sz = v.sz;
    return k;
}
// row e of monomial m / of header block `which` of a view, this lane's element
// (`rl` = the lane's place in a row: TW::rl)
__device__ inline double ld_coef(const TView& v, int m, int e, int rl) { return v.coef[((size_t)m * v.stride + v.off + e) * GR + rl]; }
__device__ inline double ld_hdr(const TView& v, int which, int e, int rl) { return v.hdr[((size_t)which * v.stride + v.off + e) * GR + rl]; }
__device__ inline void st_hdr(const TPZ& p, int which, int e, int rl, double x) { p.hdr[((size_t)which * p.sz + e) * GR + rl] = x; }

// Result writer: keys by lane 0, coefficient rows by every lane (0 where the lane pruned the term), asum on the way.
template <int SZ>
struct Out {
    GLB_AS pzkey_t* keys;
    GLB_AS double* coef;
    int cap, n, lane;   // lane: the lane's place in a row (TW::rl)
    double asum[SZ];
    __device__ inline void init(const TPZ& o, int lane_) {
        keys = uni_ptr(o.keys); coef = uni_ptr(o.coef); cap = uni(o.cap); n = 0; lane = lane_;
#pragma unroll
        for (int e = 0; e < SZ; e++) asum[e] = 0.0;
    }
    __device__ inline void emit(pzkey_t key, const double* v) {
        // no branches here (each one costs the walk a block of register moves behind it): a term beyond the capacity lands in the
        // slot's spare row block (tv_slot_bytes) -- finish() flags the overflow -- and every lane stores the wave-uniform key
        const int at = n < cap ? n : cap;
        keys[at] = key;
#pragma unroll
        for (int e = 0; e < SZ; e++) coef[((size_t)at * SZ + e) * GR + lane] = v[e];
#pragma unroll
        for (int e = 0; e < SZ; e++) asum[e] += fabs(v[e]);
        n++;
    }
    __device__ inline void finish(TW& t, const TPZ& o) {
        if (n > cap) { pzw::flag(t.w, pzw::ERR_SLOT_OVERFLOW); n = cap; }
#pragma unroll
        for (int e = 0; e < SZ; e++) st_hdr(o, H_ASUM, e, lane, asum[e]);
        if (t.w.lane == 0) {
            t.w.cnt[o.id] = n;
            if (n > t.w.lstat[pzw::ST_MAX_OUT]) t.w.lstat[pzw::ST_MAX_OUT] = n;
        }
        WSYNC();
    }
};

// ---- prune margin (round 6; pz_wave.h Wave::mabs): every per-lane verdict of simplify() leaves |s - thr_sq| of the squared norm s it was taken on
// in its walk context (a subtraction and a minimum: nothing gated -- a lane without the monomial records thr_sq, a stage looking again at a dropped
// sum records that sum again); when the context ends the minimum is folded into the wave's register (TW::w.mabs, which lives as long as the work
// item), and the kernel reduces it over the lanes -- the group's time steps -- into the problem's word when the item ends (p1_tv.inc.h).
__device__ inline void mtrk(double& m, double thr_sq, double s) {
#ifndef PZW_NO_MARGIN   // (development: the build without the tracking, for its cost -- profiles/r06_prune_margin.txt)
    m = fmin(m, fabs(s - thr_sq));
#endif
}
template <class CX>
__device__ inline void mfold(TW& t, const CX& cx) { t.w.mabs = fmin(t.w.mabs, cx.mabs); }

// simplify()'s verdict on one finished sum, per lane (RT/PZsparse.cu:327-341): a pruned term goes to the lane's radius and
// its coefficient becomes 0.  Returns whether this lane keeps the term.
template <int SZ>
__device__ inline bool verdict(double thr, double thr_sq, bool active, double* acc, double* rad) {
    // (selects, not branches: a branch on a per-lane condition costs a dozen exec-mask instructions, and this runs once per
    //  distinct raw key; adding 0.0 to a non-negative radius changes nothing)
    bool small;
    if constexpr (SZ == 1) small = fabs(acc[0]) <= thr;
    else {
        double s = 0.0;
#pragma unroll
        for (int e = 0; e < SZ; e++) s += acc[e] * acc[e];
        small = s <= thr_sq;
    }
    const bool keep = !small && active;
#pragma unroll
    for (int e = 0; e < SZ; e++) { rad[e] += keep ? 0.0 : fabs(acc[e]); acc[e] = keep ? acc[e] : 0.0; }
    return keep;
}

// The serial walk over N raw terms sorted by (key, generation index).  `keyat(p)` = key of the p-th sorted term (LDS),
// `idxat(p)` = its generation index; `P` supplies
//     struct Regs;  void load(int idx, Regs&)   -- issue the row loads of one raw term (idx is wave-uniform)
//     void add(const Regs&, bool first)          -- accumulate it (first: start a new sum)
//     void close(pzkey_t key)                   -- the run of equal keys is complete: verdict + emit
// Loads run U terms ahead of their use.
#ifdef TV_PROFILE_FULL
__device__ long long g_tvprof[8];  // (round 3's in-walk stamps: load phase, process phase, chunk prologue, batches -- no longer filled: the walk is pipelined)
#endif
#ifndef TV_WALK_PIPELINE
#define TV_WALK_PIPELINE 1   // development: 0 = round 3's form (a batch of U terms is loaded, waited for and processed before the next is asked for)
#endif
// the serial part of a batch of HB terms whose loads have been issued: products (PB at a time, term-innermost), then run logic, sums, verdicts
template <int HB, class P>
__device__ inline void walk_consume(P& pol, const typename P::Regs* regs, pzkey_t key_v, int l0, int n, bool& have, pzkey_t& cur) {
    // The products of PB terms first, as straight-line code without a branch in it -- each is a chain of dependent fp64 operations
    // (18 cycles apiece for the one wave of its SIMD), independent of the other terms' -- and only then the serial part: the run
    // logic, the sums in generation order, the verdicts.  (Round 3's walk formed a term's product inside its guarded block, so
    // that nothing overlapped.)  Same operations on the same operands in the same order per sum.
    constexpr int PB = HB % 4 == 0 ? 4 : HB % 3 == 0 ? 3 : HB % 2 == 0 ? 2 : 1;
#pragma unroll
    for (int u0 = 0; u0 < HB; u0 += PB) {
        typename P::Prod pr[PB];
        pol.template prod_batch<PB>(&regs[u0], pr);   // (written term-innermost: the compiler keeps source order, a term after a term otherwise)
#pragma unroll
        for (int v = 0; v < PB; v++) pr[v].pin();   // (formed HERE: not sunk into the guarded blocks below)
#pragma unroll
        for (int v = 0; v < PB; v++) {
            const int u = u0 + v;
            if (l0 + u < n) {
                const pzkey_t key = pzkey_readlane(key_v, l0 + u);
                if (have && key != cur) { pol.close(cur); have = false; }
                pol.accum(regs[u], pr[v], !have);
                have = true; cur = key;
            }
        }
    }
}
// a chunk of <= 64 sorted terms, one per lane: key, generation index, load descriptor (pol.describe)
template <class P>
struct WalkChunk {
    pzkey_t key_v;
    int idx_v, n;
    typename P::Desc dv;
};
template <class P, class KeyAt, class IdxAt>
__device__ inline WalkChunk<P> walk_chunk(const P& pol, int lane, int base, int N, const KeyAt& keyat, const IdxAt& idxat) {
    WalkChunk<P> c;
    const int p = base + lane;
    const bool in = p < N;
    c.key_v = in ? keyat(p) : 0ull;
    c.idx_v = in ? idxat(p) : 0;
    c.n = min(WAVE, N - base);
    // what a term's loads need -- which source, where its rows start -- is worked out for the chunk's 64 terms at once, one term per
    // lane on the vector unit, and the serial loop picks a term's descriptor out with a few v_readlane instead of redoing
    // the source selection with scalar selects per term (measured on lincomb<3, 4>: 35 of the ~110 instructions per raw term)
    c.dv = pol.describe(c.idx_v);
    return c;
}
// the loads of terms [l0, l0 + HB) of a chunk (clamped to its last term: unconditional, so that the number of loads in flight is static)
template <int HB, class P>
__device__ inline void walk_issue(const P& pol, const WalkChunk<P>& c, int l0, typename P::Regs* r) {
#pragma unroll
    for (int u = 0; u < HB; u++) { const int l = min(l0 + u, c.n - 1); pol.load(c.dv, l, __builtin_amdgcn_readlane(c.idx_v, l), r[u]); }
}
template <int U, class P, class KeyAt, class IdxAt>
__device__ inline void walk_sorted(int lane, int N_, const KeyAt& keyat, const IdxAt& idxat, P& pol, int N0_ = 0) {
    // N arrives from the sorter as a value the compiler must assume differs between lanes; as a loop bound it would put the
    // whole walk under divergent control flow (every wave-uniform variable in it becomes a vector register with exec-mask
    // bookkeeping around each update: measured 230 instructions per raw term).  One readfirstlane makes the walk scalar.
    const int N = uni(N_), N0 = uni(N0_);   // the sorted terms [N0, N): a part of a walk shared with a helper wave starts at N0 > 0
    bool have = false;
    pzkey_t cur = 0;
    if (N0 >= N) return;
    if constexpr (TV_WALK_PIPELINE && U % 2 == 0) {
        // Two half-batches in flight alternately (round 4): while one half's terms are processed the other half's rows are on their
        // way -- round 3 asked for U terms' rows, waited for them, processed them, and only then asked for the next U: a walk waits
        // for rows, and during the processing nothing was in flight.  The same registers as before (U terms' rows);
        // every load is unconditional with a clamped term index, so that the compiler's vmcnt for a half is the exact number of
        // loads issued after it.  The pipeline runs ACROSS the 64-term chunks: the next chunk's keys and descriptors are worked out
        // while the current chunk's first loads are in flight, and the half that follows a chunk's last half is the next chunk's first
        // (picked with selects on the wave-uniform condition -- a branch would make the load count path-dependent).
        // (Measured against it, one box, B = 128: a ring of FOUR quarter-batches -- three in flight -- 9.28 against 9.12 ms: the products of a
        //  quarter interleave less; larger batches, U = 24 / 12: 9.57 against 9.06, the rows no longer fit the registers.)
        constexpr int HB = U / 2;
        typename P::Regs ra[HB], rb[HB];
        WalkChunk<P> c = walk_chunk(pol, lane, N0, N, keyat, idxat);
        walk_issue<HB>(pol, c, 0, ra);
        for (int base = N0; base < N; base += WAVE) {
            const bool more = base + WAVE < N;
            WalkChunk<P> nx = c;
            if (more) nx = walk_chunk(pol, lane, base + WAVE, N, keyat, idxat);   // (wave-uniform; no row load inside)
            for (int l0 = 0; l0 < c.n; l0 += U) {
                walk_issue<HB>(pol, c, l0 + HB, rb);
                walk_consume<HB>(pol, ra, c.key_v, l0, c.n, have, cur);
                {   // the next first half: this chunk's, or -- behind its last half -- the next chunk's
                    const bool here = l0 + U < c.n;
                    WalkChunk<P> w2;
                    w2.n = here ? c.n : nx.n;
                    w2.idx_v = here ? c.idx_v : nx.idx_v;
                    w2.key_v = 0;
                    w2.dv = P::select_desc(here, c.dv, nx.dv);
                    walk_issue<HB>(pol, w2, here ? l0 + U : 0, ra);
                }
                walk_consume<HB>(pol, rb, c.key_v, l0 + HB, c.n, have, cur);
            }
            c = nx;
        }
    } else {
        for (int base = N0; base < N; base += WAVE) {
            const WalkChunk<P> c = walk_chunk(pol, lane, base, N, keyat, idxat);
            for (int l0 = 0; l0 < c.n; l0 += U) {
                typename P::Regs regs[U];
                walk_issue<U>(pol, c, l0, regs);
                walk_consume<U>(pol, regs, c.key_v, l0, c.n, have, cur);
            }
        }
    }
    if (have) pol.close(cur);
}

// Rows of one operand copied to LDS, centre first: term i (0 = centre, i >= 1 monomial i - 1), entry e at row i * sz + e.
// The walk below visits the raw terms in key order, i.e. it jumps around in both operands; every jump into global memory
// costs a round trip to the Infinity Cache or HBM (2-4 k cycles with a per-wave working set of megabytes).  The short
// operand of a product -- a joint rotation, the joint's own velocity factor -- fits LDS, which leaves the walk ONE operand's
// rows to fetch per raw term, so four times as many terms can be in flight in the same registers.
__device__ inline bool stage_fits(const TW& t, const TView& v) { return (v.cnt + 1) * v.sz <= t.stage_rows; }
__device__ inline void stage_rows_of(const TW& t, const TView& v, int lane) {
    const int sz = uni(v.sz);
    const int rl = t.rl;
    for (int e = 0; e < sz; e++) t.stage[(size_t)e * WAVE + lane] = ld_hdr(v, H_CEN, e, rl);
    const int rows = uni(v.cnt * sz);   // (a scalar bound: the guarded stores below are scalar branches, not sixteen exec masks)
    const GLB_AS double* src = v.coef + (size_t)v.off * GR + rl;  // (whole-PZ views: off = 0, stride = sz, rows are consecutive)
    LDS_AS double* dst = t.stage + (size_t)sz * WAVE + lane;
    for (int r0 = 0; r0 < rows; r0 += 16) {
        double x[16];
#pragma unroll
        for (int u = 0; u < 16; u++) x[u] = src[(size_t)min(r0 + u, rows - 1) * GR];
#pragma unroll
        for (int u = 0; u < 16; u++) if (r0 + u < rows) dst[(size_t)(r0 + u) * WAVE] = x[u];
    }
    WSYNC();
}

// ---------------------------------------------------------------------------------------------------------------------
// product (RT/PZsparse.cu:864-994), shapes as in pz_wave.h.  STAGE: 0 both operands from global memory, 1 a's rows staged in
// LDS, 2 b's rows staged.
template <class SH, int STAGE>
struct MulCtx {
    TView a, b;
    int lane, mb1;
    unsigned long long mb1_magic;
    double thr, thr_sq;
    bool active;
    Out<SH::SZ>* o;
    const LDS_AS double* stage;
    double acc[SH::SZ], rad[SH::SZ];
    double mabs = __builtin_inf();   // prune margin of this walk (mtrk / mfold)
    struct Regs { double ca[STAGE == 1 ? 1 : SH::ASZ], cb[STAGE == 2 ? 1 : SH::BSZ]; int i, j; };   // i, j: the staged operand's term index (rows in LDS)
    static constexpr int kRegDoubles = (STAGE == 1 ? 0 : SH::ASZ) + (STAGE == 2 ? 0 : SH::BSZ);
    static constexpr int kU = kRegDoubles <= 3 ? 16 : kRegDoubles <= 6 ? 8 : kRegDoubles <= 9 ? 6 : 4;  // terms whose row loads are in flight together
    // per-lane descriptors of the chunk's terms (lane l describes the chunk's l-th sorted term): the index split and the row addresses are
    // worked out for 64 terms at once on the vector unit; per term the serial loop then needs a few v_readlane instead of a chain of
    // twenty dependent scalar instructions (split by magic multiply, 64-bit address arithmetic) in front of every term's loads
    struct Desc { unsigned alo, ahi, blo, bhi; int i, j; };
    __device__ static inline Desc select_desc(bool first, const Desc& x, const Desc& y) {   // (wave-uniform condition: selects, no branch)
        Desc d;
        d.alo = first ? x.alo : y.alo; d.ahi = first ? x.ahi : y.ahi; d.blo = first ? x.blo : y.blo; d.bhi = first ? x.bhi : y.bhi; d.i = first ? x.i : y.i; d.j = first ? x.j : y.j;
        return d;
    }
    __device__ inline Desc describe(int idx_lane) const {
        Desc d;
        const int t = idx_lane + 1;
        const int i = (int)(((unsigned long long)t * mb1_magic) >> 32), j = t - i * mb1;
        d.i = i; d.j = j;
        const uint64_t pa = (uint64_t)(a.coef + ((ptrdiff_t)(i - 1) * a.stride + a.off) * GR);   // i = 0: the centre rows, stored right before the coefficients
        const uint64_t pb = (uint64_t)(b.coef + ((ptrdiff_t)(j - 1) * b.stride + b.off) * GR);
        d.alo = (unsigned)pa; d.ahi = (unsigned)(pa >> 32); d.blo = (unsigned)pb; d.bhi = (unsigned)(pb >> 32);
        return d;
    }
    __device__ inline void load(const Desc& d, int l, int, Regs& r) const {
        if constexpr (STAGE == 1) r.i = __builtin_amdgcn_readlane(d.i, l);
        if constexpr (STAGE == 2) r.j = __builtin_amdgcn_readlane(d.j, l);
        if constexpr (STAGE != 1) {
            const uint64_t pa = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)d.ahi, l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)d.alo, l);
            const GLB_AS double* p = (const GLB_AS double*)pa + lane;
#pragma unroll
            for (int e = 0; e < SH::ASZ; e++) r.ca[e] = p[e * GR];
        }
        if constexpr (STAGE != 2) {
            const uint64_t pb = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)d.bhi, l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)d.blo, l);
            const GLB_AS double* p = (const GLB_AS double*)pb + lane;
#pragma unroll
            for (int e = 0; e < SH::BSZ; e++) r.cb[e] = p[e * GR];
        }
    }
    // (generation index known as a scalar: the ordered pass of a product with a constant left operand)
    __device__ inline void load(int idx, Regs& r) const {
        const int t = idx + 1;
        const int i = (int)(((unsigned long long)t * mb1_magic) >> 32), j = t - i * mb1;
        r.i = i; r.j = j;
        if constexpr (STAGE != 1) {
            const GLB_AS double* pa = a.coef + ((ptrdiff_t)(i - 1) * a.stride + a.off) * GR + lane;
#pragma unroll
            for (int e = 0; e < SH::ASZ; e++) r.ca[e] = pa[e * GR];
        }
        if constexpr (STAGE != 2) {
            const GLB_AS double* pb = b.coef + ((ptrdiff_t)(j - 1) * b.stride + b.off) * GR + lane;
#pragma unroll
            for (int e = 0; e < SH::BSZ; e++) r.cb[e] = pb[e * GR];
        }
    }
    // the term's coefficient product: pure (no state of the walk), so that the walk can form several terms' products at once
    struct Prod { double c[SH::SZ]; __device__ inline void pin() { for (int e = 0; e < SH::SZ; e++) __asm__ volatile("" : "+v"(c[e])); } };
    __device__ inline void prod(const Regs& r, Prod& p) const {
        double ca[SH::ASZ], cb[SH::BSZ];
        if constexpr (STAGE == 1) {
            const LDS_AS double* pa = stage + (size_t)r.i * SH::ASZ * WAVE + lane;
#pragma unroll
            for (int e = 0; e < SH::ASZ; e++) ca[e] = pa[e * WAVE];
        } else {
#pragma unroll
            for (int e = 0; e < SH::ASZ; e++) ca[e] = r.ca[e];
        }
        if constexpr (STAGE == 2) {
            const LDS_AS double* pb = stage + (size_t)r.j * SH::BSZ * WAVE + lane;
#pragma unroll
            for (int e = 0; e < SH::BSZ; e++) cb[e] = pb[e * WAVE];
        } else {
#pragma unroll
            for (int e = 0; e < SH::BSZ; e++) cb[e] = r.cb[e];
        }
        SH::mul(ca, cb, p.c);
    }
    // PB terms' products with the term index INNERMOST in every loop: the staged rows of all PB terms are requested first, and every step of
    // the dot products is taken for all PB terms before the next -- PB x rows independent chains in flight instead of one term's after the
    // other's (per element exactly SH::mul's operations in SH::mul's order)
    template <int PB>
    __device__ inline void prod_batch(const Regs* r, Prod* p) const {
        double ca[PB][SH::ASZ], cb[PB][SH::BSZ];
#pragma unroll
        for (int v = 0; v < PB; v++) {
            if constexpr (STAGE == 1) {
                const LDS_AS double* pa = stage + (size_t)r[v].i * SH::ASZ * WAVE + lane;
#pragma unroll
                for (int e = 0; e < SH::ASZ; e++) ca[v][e] = pa[e * WAVE];
            } else {
#pragma unroll
                for (int e = 0; e < SH::ASZ; e++) ca[v][e] = r[v].ca[e];
            }
            if constexpr (STAGE == 2) {
                const LDS_AS double* pb = stage + (size_t)r[v].j * SH::BSZ * WAVE + lane;
#pragma unroll
                for (int e = 0; e < SH::BSZ; e++) cb[v][e] = pb[e * WAVE];
            } else {
#pragma unroll
                for (int e = 0; e < SH::BSZ; e++) cb[v][e] = r[v].cb[e];
            }
        }
        if constexpr (SH::a11) {
#pragma unroll
            for (int e = 0; e < SH::BSZ; e++)
#pragma unroll
                for (int v = 0; v < PB; v++) p[v].c[e] = ca[v][0] * cb[v][e];
        } else {
            constexpr int AR = SH::OR_, BC = SH::OC, AC = SH::ASZ / SH::OR_;
#pragma unroll
            for (int rr = 0; rr < AR; rr++)
#pragma unroll
                for (int cc = 0; cc < BC; cc++) {
                    double sm[PB];
#pragma unroll
                    for (int v = 0; v < PB; v++) sm[v] = 0.0;
#pragma unroll
                    for (int k = 0; k < AC; k++)
#pragma unroll
                        for (int v = 0; v < PB; v++) sm[v] += ca[v][rr * AC + k] * cb[v][k * BC + cc];
#pragma unroll
                    for (int v = 0; v < PB; v++) p[v].c[rr * BC + cc] = sm[v];
                }
        }
    }
    __device__ inline void accum(const Regs&, const Prod& p, bool first) {
        // (selects on the wave-uniform `first`, not a branch: every branch of the walk ends in a block of register moves that merges
        //  the accumulators of its two sides -- about forty per raw term before this form)
#pragma unroll
        for (int e = 0; e < SH::SZ; e++) { const double sum = acc[e] + p.c[e]; acc[e] = first ? p.c[e] : sum; }
    }
    __device__ inline void close(pzkey_t key) {
        bool small;
        double q = 0.0;
#pragma unroll
        for (int e = 0; e < SH::SZ; e++) q += acc[e] * acc[e];
        if constexpr (SH::SZ == 1) small = fabs(acc[0]) <= thr;
        else small = q <= thr_sq;
        mtrk(mabs, thr_sq, q);
        const bool keep = !small && active;
        // the pruned amount goes to the radius in every lane that does not keep the term (all of them, nine times in ten) ...
#pragma unroll
        for (int e = 0; e < SH::SZ; e++) rad[e] += keep ? 0.0 : fabs(acc[e]);
        // ... and only a term some lane keeps is written
        if (__ballot(keep) != 0ull) {
            double v[SH::SZ];
#pragma unroll
            for (int e = 0; e < SH::SZ; e++) v[e] = keep ? acc[e] : 0.0;
            o->emit(key, v);
        }
    }
};

template <class SH, int STAGE>
__device__ inline void mul_ctx_init(MulCtx<SH, STAGE>& cx, const TW& t, const TView& a, const TView& b, Out<SH::SZ>* o) {
    cx.a = a; cx.b = b; cx.lane = t.rl; cx.mb1 = b.cnt + 1; cx.mb1_magic = pzw::magic_u32(cx.mb1);   // (cx.lane addresses rows, global and staged)
    cx.a.coef = uni_ptr(a.coef); cx.b.coef = uni_ptr(b.coef); cx.a.stride = uni(a.stride); cx.b.stride = uni(b.stride); cx.a.off = uni(a.off); cx.b.off = uni(b.off);
    cx.thr = t.w.thr; cx.thr_sq = t.w.thr_sq; cx.active = t.active; cx.o = o; cx.stage = t.stage;
#pragma unroll
    for (int e = 0; e < SH::SZ; e++) { cx.acc[e] = 0.0; cx.rad[e] = 0.0; }
}
// (sw: the wave whose LDS buffers hold the sorted terms -- this wave's own, or the primary's when a helper walks a part of them;
//  the sorted terms [N0, N))
template <class SH, int STAGE>
__device__ inline void mul_walk(TW& t, const Wave& sw, const LDS_AS double* stage, int N0, int N, bool indirect, const pzw::MulEval<SH>& ev, const TView& a, const TView& b,
                                Out<SH::SZ>* o, double* rad) {
    MulCtx<SH, STAGE> cx;
    mul_ctx_init(cx, t, a, b, o);
    cx.stage = stage;
    const Wave& w = sw;
    const int lane = t.w.lane;
    if (indirect) walk_sorted<MulCtx<SH, STAGE>::kU>(lane, N, [&](int p) { return ev.key_lds(w, w.sidx[p]); }, [&](int p) { return (int)w.sidx[p]; }, cx, N0);
    else walk_sorted<MulCtx<SH, STAGE>::kU>(lane, N, [&](int p) { return w.skey[p]; }, [&](int p) { return (int)w.sidx[p]; }, cx, N0);
#pragma unroll
    for (int e = 0; e < SH::SZ; e++) rad[e] = cx.rad[e];
    mfold(t, cx);
}

// rows [r_lo, r_hi) of a block of rows from src to dst (row 0 of each, WITHOUT the lane offset), 32 rows in flight
__device__ inline void move_rows(const GLB_AS double* src0, GLB_AS double* dst0, int rl, int r_lo, int r_hi) {
    const GLB_AS double* src = src0 + rl;
    GLB_AS double* dst = dst0 + rl;
    for (int r0 = r_lo; r0 < r_hi; r0 += 32) {
        double x[32];
#pragma unroll
        for (int u = 0; u < 32; u++) x[u] = src[(size_t)min(r0 + u, r_hi - 1) * GR];
#pragma unroll
        for (int u = 0; u < 32; u++) if (r0 + u < r_hi) dst[(size_t)(r0 + u) * GR] = x[u];
    }
}

// ---- the primary's side of a shared walk
// The split point: about `num / den` of the N sorted terms for the primary, moved up to the next boundary between runs of equal keys.
template <class KeyAt>
__device__ inline int split_point(int N, int num, int den, const KeyAt& keyat) {
    int S = uni((int)((long long)N * num / den));
    if (S < 1) S = 1;
    while (S < N && keyat(S) == keyat(S - 1)) S++;
    return uni(S);
}
__device__ inline void hj_signal(const TW& t, LDS_AS int* p, int v) {
    WSYNC();   // this wave's stores (rows, keys, channel words) are done
    if (t.w.lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ inline void hj_wait(TW& t, LDS_AS int* p, int v) {
#ifdef TV_PROFILE
    const long long hw0__ = clock64();
#endif
    int spins = 0;
    while (lds_ld(p) < v) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1 << 23)) { pzw::flag(t.w, pzw::ERR_SLOT_OVERFLOW); break; }   // never seen; ends the wait instead of the box
    }
    WSYNC();
#ifdef TV_PROFILE
    t.c_hwait += clock64() - hw0__;
#endif
}
// no helper work for this operator: the helper's job counter still advances (both sides count the operators of the pass)
// (the channel holds ONE job: the helper acknowledges every job, also an empty one, in HJ_DONE, and the primary does not write the
//  next job's words before the previous acknowledgement is in)
__device__ inline void hj_post_none(TW& t) {
    if (t.hded) return;   // (a dedicated helper counts nothing)
    hj_wait(t, &t.hch[HJ_DONE], t.hseq);
    t.hseq++;
    if (t.w.lane == 0) t.hch[HJ_KIND] = HK_NONE;
    hj_signal(t, &t.hch[HJ_SEQ], t.hseq);
}
// control job for a dedicated helper (HK_BAR, HK_EXIT)
__device__ inline void hj_post_ctl(TW& t, int kind) {
    hj_wait(t, &t.hch[HJ_DONE], t.hseq);
    t.hseq++;
    if (t.w.lane == 0) t.hch[HJ_KIND] = kind;
    hj_signal(t, &t.hch[HJ_SEQ], t.hseq);
}
// common part of a job: the range, the primary's sorted arrays, the result slot.  (Increments the job counter.)
__device__ inline void hj_post_common(TW& t, int kind, int S, int N, bool indirect, const TPZ& out) {
    hj_wait(t, &t.hch[HJ_DONE], t.hseq);
    t.hseq++;
    if (t.w.lane == 0) {
        LDS_AS int* ch = t.hch;
        ch[HJ_KIND] = kind; ch[HJ_S] = S; ch[HJ_N] = N; ch[HJ_INDIRECT] = indirect ? 1 : 0;
        ch[HJ_SKEY] = (int)(uintptr_t)t.w.skey; ch[HJ_SIDX] = (int)(uintptr_t)t.w.sidx; ch[HJ_STAGE] = (int)(uintptr_t)t.stage;
        lds_st_ptr(&ch[HJ_OUT_KEYS], out.keys); lds_st_ptr(&ch[HJ_OUT_COEF], out.coef); ch[HJ_OUT_CAP] = out.cap;
    }
}
__device__ inline void hj_post_seg(TW& t, int k, const TView& v, double scale, int comp) {
    if (t.w.lane == 0) {
        LDS_AS int* sg = t.hch + HJ_SEG0 + k * HJ_SEG_WORDS;
        lds_st_ptr(&sg[0], v.coef); sg[2] = v.cnt; sg[3] = v.stride; sg[4] = v.off; sg[5] = comp;
        const uint64_t sb = (uint64_t)__double_as_longlong(scale);
        sg[6] = (int)(unsigned)sb; sg[7] = (int)(unsigned)(sb >> 32);
    }
}
// after the primary's own part: post how many terms it kept, wait until the helper has moved its rows behind them, and return the
// helper's count; its partial sums are then in the header rows of its scratch slot (`hdr`: [4][3][64], rows H_IND.. as the caller laid out)
// (the move of the helper's rows behind the primary's is the one serial step of a shared walk: both waves take half of it)
__device__ inline int hj_collect(TW& t, int n0, const GLB_AS double*& hdr, const TPZ& out) {
    if (t.w.lane == 0) t.hch[HJ_N0] = n0;
    hj_signal(t, &t.hch[HJ_N0SEQ], t.hseq);
    hj_wait(t, &t.hch[HJ_WALKED], t.hseq);
    const int nh = uni(t.hch[HJ_NH]);
    const int room = out.cap - n0 > 0 ? out.cap - n0 : 0, ncopy = nh < room ? nh : room;
    const GLB_AS double* src = lds_ld_ptr<const GLB_AS double>(&t.hch[HJ_TMP_COEF]);
    move_rows(src, out.coef + (size_t)n0 * 3 * GR, t.rl, 0, (ncopy * 3) / 2);
    hj_wait(t, &t.hch[HJ_DONE], t.hseq);
    hdr = lds_ld_ptr<const GLB_AS double>(&t.hch[HJ_HDR]);
    return nh;
}

template <int AR, int AC, int BR, int BC>
__device__ TV_NOINLINE void mul(TW& t, const TPZ& out, const TView& a_, const TView& b_) {
    PZ_KEEP_RETURN_ADDRESS();
    typedef pzw::MulShape<AR, AC, BR, BC> SH;
    constexpr int SZ = SH::SZ;
    const int lane = t.w.lane;
    const int rl = t.rl;   // (the lane's place in a row)
    TView a = a_, b = b_;
    a.cnt = uni(a.cnt); b.cnt = uni(b.cnt);
    TVP_FN(t, a.cnt == 0 ? 1 : 0)
    int N = (a.cnt + 1) * (b.cnt + 1) - 1;
    // per-lane centre and radii (RT/PZsparse.cu:868,944-989)
    double ca[SH::ASZ], cb[SH::BSZ], ia[SH::ASZ], ib[SH::BSZ], ia2[SH::ASZ], ib2[SH::BSZ], r2[SH::ASZ], r3[SH::BSZ];
#pragma unroll
    for (int e = 0; e < SH::ASZ; e++) {
        ca[e] = ld_hdr(a, H_CEN, e, rl); ia[e] = ld_hdr(a, H_IND, e, rl); ia2[e] = ld_hdr(a, H_IND2, e, rl);
        r2[e] = fabs(ca[e]) + ld_hdr(a, H_ASUM, e, rl);
    }
#pragma unroll
    for (int e = 0; e < SH::BSZ; e++) {
        cb[e] = ld_hdr(b, H_CEN, e, rl); ib[e] = ld_hdr(b, H_IND, e, rl); ib2[e] = ld_hdr(b, H_IND2, e, rl);
        r3[e] = fabs(cb[e]) + ld_hdr(b, H_ASUM, e, rl);
    }
    double t2[SZ], t3[SZ], ii[SZ], cen[SZ], base[SZ], base2[SZ];
    SH::mul(r2, ib, t2);
    SH::mul(ia, r3, t3);
    SH::mul(ia, ib, ii);
    SH::mul(ca, cb, cen);
#pragma unroll
    for (int e = 0; e < SZ; e++) base[e] = ii[e] + (t2[e] + t3[e]);
    SH::mul(r2, ib2, t2);
    SH::mul(ia2, r3, t3);
    SH::mul(ia2, ib2, ii);
#pragma unroll
    for (int e = 0; e < SZ; e++) base2[e] = ii[e] + (t2[e] + t3[e]);
    WSYNC();

    Out<SZ> o;
    o.init(out, rl);
    double rad[SZ];
#pragma unroll
    for (int e = 0; e < SZ; e++) rad[e] = 0.0;
    if (lane == 0 && N > t.w.lstat[pzw::ST_MAX_RAW]) t.w.lstat[pzw::ST_MAX_RAW] = N;
    if (a.cnt == 0) {
        // constant left operand (mass, inertia, the fixed rpy rotation): b's keys in b's order, no sort
        if (t.hch != nullptr && AR == 3 && AC == 3 && BR == 3 && BC == 1) hj_post_none(t);
        MulCtx<SH, 0> cx;
        mul_ctx_init(cx, t, a, b, &o);
        // b's rows are walked in their own order (contiguous), two half-batches in flight alternately as in walk_sorted; the left operand is its
        // centre alone -- `ca`, loaded once above: round 3 fetched its rows again with every term (L1 hits, but 9 of a term's 12 row loads for an
        // inertia matrix).  Per term the same product and the same verdict in the same order.
        constexpr int HBc = SH::BSZ > 3 ? 2 : 4;
        double xa[HBc][SH::BSZ], xb[HBc][SH::BSZ];
        const int nb_ = b.cnt;
        auto issue_c = [&](double (*x)[SH::BSZ], int m) {
#pragma unroll
            for (int u = 0; u < HBc; u++) {
                const int mm = min(m + u, nb_ - 1);
#pragma unroll
                for (int e = 0; e < SH::BSZ; e++) x[u][e] = ld_coef(b, mm, e, rl);
            }
        };
        auto consume_c = [&](double (*x)[SH::BSZ], int m, pzkey_t key_v, int m0) {
            typename MulCtx<SH, 0>::Prod pr[HBc];
#pragma unroll
            for (int u = 0; u < HBc; u++) SH::mul(ca, x[u], pr[u].c);
#pragma unroll
            for (int u = 0; u < HBc; u++) pr[u].pin();
            typename MulCtx<SH, 0>::Regs dummy;
#pragma unroll
            for (int u = 0; u < HBc; u++)
                if (m + u < nb_) { cx.accum(dummy, pr[u], true); cx.close(pzkey_readlane(key_v, m + u - m0)); }
        };
        if (nb_ > 0) issue_c(xa, 0);
        for (int m0 = 0; m0 < nb_; m0 += WAVE) {
            const pzkey_t key_v = m0 + lane < nb_ ? b.keys[m0 + lane] : 0ull;
            const int n = min(WAVE, nb_ - m0);
            for (int l0 = 0; l0 < n; l0 += 2 * HBc) {
                issue_c(xb, m0 + l0 + HBc);
                consume_c(xa, m0 + l0, key_v, m0);
                issue_c(xa, m0 + l0 + 2 * HBc);
                consume_c(xb, m0 + l0 + HBc, key_v, m0);
            }
        }
#pragma unroll
        for (int e = 0; e < SZ; e++) rad[e] = cx.rad[e];
        mfold(t, cx);
    } else {
        pzw::MulEval<SH> ev;
        ev.a = kview(a); ev.set_b(kview(b));
        bool indirect = false;
        TVP_T0
        N = pzw::sort_terms(t.w, N, ev, indirect);
        TVP_T1
        // the shorter operand's rows go to LDS when they fit (whole-PZ views only)
        const bool a_short = a.cnt <= b.cnt;
        const bool can_a = a.off == 0 && a.sz == a.stride && stage_fits(t, a), can_b = b.off == 0 && b.sz == b.stride && stage_fits(t, b);
        constexpr bool kSplittable = (AR == 3 && AC == 3 && BR == 3 && BC == 1);   // rotation x vector: the products of the backward recursions
        // (wave-uniform in fact -- every lane holds the same counts and the same view -- and made so for the compiler, which otherwise builds the
        //  three walks as exec-masked branches of each other)
        const bool stage_a = uni(((a_short && can_a) || (!can_b && can_a)) ? 1 : 0) != 0;
        const int stage = uni(stage_a ? 1 : can_b ? 2 : 0);
        bool shared = false;
        if constexpr (kSplittable) {
            if (t.hch != nullptr) {
                if ((stage_a || t.hded) && N >= t.hmin) shared = true; else hj_post_none(t);   // (an opportunistic helper serves the staged-rotation form only)
            }
        }
        if (stage == 1) stage_rows_of(t, a, lane); else if (stage == 2) stage_rows_of(t, b, lane);
        int S = N;
        if constexpr (kSplittable) {
            if (shared) {
                const Wave& w = t.w;
                S = indirect ? split_point(N, t.hnum, 32, [&](int p) { return ev.key_lds(w, w.sidx[p]); }) : split_point(N, t.hnum, 32, [&](int p) { return w.skey[p]; });
                hj_post_common(t, stage == 1 ? HK_MUL_3331_A_STAGED : stage == 2 ? HK_MUL_3331_B_STAGED : HK_MUL_3331_UNSTAGED, S, N, indirect, out);
#ifdef TV_PROFILE
                t.n_shared += 1; t.n_shared_terms += N;
#endif
                hj_post_seg(t, 0, a, 1.0, -1); hj_post_seg(t, 1, b, 1.0, -1);
                hj_signal(t, &t.hch[HJ_SEQ], t.hseq);
            }
        }
        if (stage == 1) mul_walk<SH, 1>(t, t.w, t.stage, 0, S, indirect, ev, a, b, &o, rad);
        else if (stage == 2) mul_walk<SH, 2>(t, t.w, t.stage, 0, S, indirect, ev, a, b, &o, rad);
        else mul_walk<SH, 0>(t, t.w, t.stage, 0, S, indirect, ev, a, b, &o, rad);
        if constexpr (kSplittable) {
            if (shared) {
                const GLB_AS double* hdr;
                const int nh = hj_collect(t, o.n, hdr, out);
#pragma unroll
                for (int e = 0; e < SZ; e++) { rad[e] += hdr[((size_t)H_IND * SZ + e) * GR + rl]; o.asum[e] += hdr[((size_t)H_ASUM * SZ + e) * GR + rl]; }
                o.n += nh;
            }
        }
        TVP_END(t, N, o.n, 0)
    }
#pragma unroll
    for (int e = 0; e < SZ; e++) {
        st_hdr(out, H_CEN, e, rl, cen[e]);
        st_hdr(out, H_IND, e, rl, base[e] + rad[e]);
        st_hdr(out, H_IND2, e, rl, base2[e] + rad[e]);
    }
    o.finish(t, out);
}

// ---------------------------------------------------------------------------------------------------------------------
// cross(a, b) of 3x1 operands (RT/PZsparse.cu:1134-1151) with the three simplify() stages of the composition, per lane
template <int STAGE>  // 0: both operands from global memory, 1: a's rows staged in LDS, 2: b's
struct CrossCtx {
    TView a, b;
    int lane, mb1;
    unsigned long long mb1_magic;
    double thr, thr_sq;
    bool active;
    Out<3>* o;
    const LDS_AS double* stage;
    double acc[6], rad[12];  // radii: 6 products | 3 differences | 3 stack
    double mabs = __builtin_inf();   // prune margin of this walk (mtrk / mfold)
    struct Regs { double ca[STAGE == 1 ? 1 : 3], cb[STAGE == 2 ? 1 : 3]; int i, j; };
    static constexpr int kU = STAGE == 0 ? 8 : 16;
    struct Desc { unsigned alo, ahi, blo, bhi; int i, j; };   // (see MulCtx::Desc)
    __device__ static inline Desc select_desc(bool first, const Desc& x, const Desc& y) {
        Desc d;
        d.alo = first ? x.alo : y.alo; d.ahi = first ? x.ahi : y.ahi; d.blo = first ? x.blo : y.blo; d.bhi = first ? x.bhi : y.bhi; d.i = first ? x.i : y.i; d.j = first ? x.j : y.j;
        return d;
    }
    __device__ inline Desc describe(int idx_lane) const {
        Desc d;
        const int t = idx_lane + 1;
        const int i = (int)(((unsigned long long)t * mb1_magic) >> 32), j = t - i * mb1;
        d.i = i; d.j = j;
        const uint64_t pa = (uint64_t)(a.coef + (ptrdiff_t)(i - 1) * 3 * GR), pb = (uint64_t)(b.coef + (ptrdiff_t)(j - 1) * 3 * GR);   // i, j = 0: the centre rows
        d.alo = (unsigned)pa; d.ahi = (unsigned)(pa >> 32); d.blo = (unsigned)pb; d.bhi = (unsigned)(pb >> 32);
        return d;
    }
    __device__ inline void load(const Desc& d, int l, int, Regs& r) const {
        if constexpr (STAGE == 1) r.i = __builtin_amdgcn_readlane(d.i, l);
        if constexpr (STAGE == 2) r.j = __builtin_amdgcn_readlane(d.j, l);
        if constexpr (STAGE != 1) {
            const uint64_t pa = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)d.ahi, l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)d.alo, l);
            const GLB_AS double* p = (const GLB_AS double*)pa + lane;
#pragma unroll
            for (int e = 0; e < 3; e++) r.ca[e] = p[e * GR];
        }
        if constexpr (STAGE != 2) {
            const uint64_t pb = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)d.bhi, l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)d.blo, l);
            const GLB_AS double* p = (const GLB_AS double*)pb + lane;
#pragma unroll
            for (int e = 0; e < 3; e++) r.cb[e] = p[e * GR];
        }
    }
    struct Prod { double p6[6]; __device__ inline void pin() { for (int e = 0; e < 6; e++) __asm__ volatile("" : "+v"(p6[e])); } };
    __device__ inline void prod(const Regs& r, Prod& p) const {
        double ca[3], cb[3];
#pragma unroll
        for (int e = 0; e < 3; e++) {
            if constexpr (STAGE == 1) ca[e] = stage[((size_t)r.i * 3 + e) * WAVE + lane]; else ca[e] = r.ca[e];
            if constexpr (STAGE == 2) cb[e] = stage[((size_t)r.j * 3 + e) * WAVE + lane]; else cb[e] = r.cb[e];
        }
        p.p6[0] = ca[1] * cb[2]; p.p6[1] = ca[2] * cb[1];
        p.p6[2] = ca[2] * cb[0]; p.p6[3] = ca[0] * cb[2];
        p.p6[4] = ca[0] * cb[1]; p.p6[5] = ca[1] * cb[0];
    }
    template <int PB>
    __device__ inline void prod_batch(const Regs* r, Prod* p) const {   // (the staged rows of all PB terms first: see MulCtx::prod_batch)
        double ca[PB][3], cb[PB][3];
#pragma unroll
        for (int v = 0; v < PB; v++)
#pragma unroll
            for (int e = 0; e < 3; e++) {
                if constexpr (STAGE == 1) ca[v][e] = stage[((size_t)r[v].i * 3 + e) * WAVE + lane]; else ca[v][e] = r[v].ca[e];
                if constexpr (STAGE == 2) cb[v][e] = stage[((size_t)r[v].j * 3 + e) * WAVE + lane]; else cb[v][e] = r[v].cb[e];
            }
#pragma unroll
        for (int v = 0; v < PB; v++) {
            p[v].p6[0] = ca[v][1] * cb[v][2]; p[v].p6[1] = ca[v][2] * cb[v][1];
            p[v].p6[2] = ca[v][2] * cb[v][0]; p[v].p6[3] = ca[v][0] * cb[v][2];
            p[v].p6[4] = ca[v][0] * cb[v][1]; p[v].p6[5] = ca[v][1] * cb[v][0];
        }
    }
    __device__ inline void accum(const Regs&, const Prod& p, bool first) {
        // (selects on the wave-uniform `first`: see MulCtx::accum)
#pragma unroll
        for (int e = 0; e < 6; e++) { const double sum = acc[e] + p.p6[e]; acc[e] = first ? p.p6[e] : sum; }
    }
    __device__ inline void close(pzkey_t key) {
        // first stage for everybody: a product below the threshold goes to its product's radius.  In the common case that is all
        // six of them in every lane, and then nothing else happens (no difference, no stack entry)
        bool h[6], anyh = false;
#pragma unroll
        for (int e = 0; e < 6; e++) {
            h[e] = !(fabs(acc[e]) <= thr);
            mtrk(mabs, thr_sq, acc[e] * acc[e]);
            rad[e] += h[e] ? 0.0 : fabs(acc[e]);
            anyh = anyh || h[e];
        }
        if (__ballot(anyh) == 0ull) return;
        // the other two simplify() stages of the composed cross product, per lane, as selects (see verdict())
        double u[3];
        bool anyc = false;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double v0 = acc[2 * c], v1 = acc[2 * c + 1];
            const bool h0 = h[2 * c], h1 = h[2 * c + 1];
            double wv = h0 ? 1.0 * v0 : -1.0 * v1;
            const double wv2 = wv + -1.0 * v1;
            wv = (h0 && h1) ? wv2 : wv;
            const bool any01 = h0 || h1, smallw = fabs(wv) <= thr;
            mtrk(mabs, thr_sq, wv * wv);
            rad[6 + c] += (any01 && smallw) ? fabs(wv) : 0.0;
            const bool kc = any01 && !smallw;
            u[c] = kc ? wv : 0.0;
            anyc = anyc || kc;
        }
        const double s = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
        mtrk(mabs, thr_sq, s);
        const bool keep = anyc && !(s <= thr_sq) && active;
#pragma unroll
        for (int c = 0; c < 3; c++) { rad[9 + c] += (anyc && !keep) ? fabs(u[c]) : 0.0; u[c] = keep ? u[c] : 0.0; }
        if (__ballot(keep) != 0ull) o->emit(key, u);
    }
};

// (sw, stage, [N0, N): as mul_walk)
template <int STAGE>
__device__ inline void cross_walk(TW& t, const Wave& sw, const LDS_AS double* stage, int N0, int N, bool indirect, const pzw::MulEval<pzw::MulShape<1, 1, 1, 1>>& ev, const TView& a, const TView& b, Out<3>* o, double* rad) {
    CrossCtx<STAGE> cx;
    cx.a = a; cx.b = b; cx.lane = t.rl; cx.mb1 = b.cnt + 1; cx.mb1_magic = pzw::magic_u32(cx.mb1);
    cx.a.coef = uni_ptr(a.coef); cx.b.coef = uni_ptr(b.coef);
    cx.thr = t.w.thr; cx.thr_sq = t.w.thr_sq; cx.active = t.active; cx.o = o; cx.stage = stage;
#pragma unroll
    for (int e = 0; e < 6; e++) cx.acc[e] = 0.0;
#pragma unroll
    for (int e = 0; e < 12; e++) cx.rad[e] = 0.0;
    const Wave& w = sw;
    const int lane = t.w.lane;
    if (indirect) walk_sorted<CrossCtx<STAGE>::kU>(lane, N, [&](int p) { return ev.key_lds(w, w.sidx[p]); }, [&](int p) { return (int)w.sidx[p]; }, cx, N0);
    else walk_sorted<CrossCtx<STAGE>::kU>(lane, N, [&](int p) { return w.skey[p]; }, [&](int p) { return (int)w.sidx[p]; }, cx, N0);
#pragma unroll
    for (int e = 0; e < 12; e++) rad[e] = cx.rad[e];
    mfold(t, cx);
}

__device__ TV_NOINLINE void cross_pzpz(TW& t, const TPZ& out, const TView& a_, const TView& b_) {
    PZ_KEEP_RETURN_ADDRESS();
    TVP_FN(t, 2)
    typedef pzw::MulShape<1, 1, 1, 1> SH;
    const int lane = t.w.lane;
    const int rl = t.rl;   // (the lane's place in a row)
    TView a = a_, b = b_;
    a.cnt = uni(a.cnt); b.cnt = uni(b.cnt);
    int N = (a.cnt + 1) * (b.cnt + 1) - 1;
    double cenP[6], baseP[6], base2P[6];
    {
        double ca[3], cb[3], ia[3], ib[3], ia2[3], ib2[3], r2[3], r3[3];
#pragma unroll
        for (int e = 0; e < 3; e++) {
            ca[e] = ld_hdr(a, H_CEN, e, rl); ia[e] = ld_hdr(a, H_IND, e, rl); ia2[e] = ld_hdr(a, H_IND2, e, rl);
            r2[e] = fabs(ca[e]) + ld_hdr(a, H_ASUM, e, rl);
            cb[e] = ld_hdr(b, H_CEN, e, rl); ib[e] = ld_hdr(b, H_IND, e, rl); ib2[e] = ld_hdr(b, H_IND2, e, rl);
            r3[e] = fabs(cb[e]) + ld_hdr(b, H_ASUM, e, rl);
        }
        const int ia_[6] = {1, 2, 2, 0, 0, 1}, ib_[6] = {2, 1, 0, 2, 1, 0};
#pragma unroll
        for (int e = 0; e < 6; e++) {
            const int i = ia_[e], j = ib_[e];
            cenP[e] = ca[i] * cb[j];
            baseP[e] = ia[i] * ib[j] + (r2[i] * ib[j] + ia[i] * r3[j]);
            base2P[e] = ia2[i] * ib2[j] + (r2[i] * ib2[j] + ia2[i] * r3[j]);
        }
    }
    WSYNC();
    Out<3> o;
    o.init(out, rl);
    double rad[12];
    if (lane == 0 && N > t.w.lstat[pzw::ST_MAX_RAW]) t.w.lstat[pzw::ST_MAX_RAW] = N;
    pzw::MulEval<SH> ev;
    ev.a = kview(a); ev.set_b(kview(b));
    ev.a.stride = 1; ev.a.off = 0; ev.b.stride = 1; ev.b.off = 0;
    bool indirect = false;
    TVP_T0
    N = pzw::sort_terms(t.w, N, ev, indirect);
    TVP_T1
    {
        const bool a_short = a.cnt <= b.cnt;
        const bool can_a = stage_fits(t, a), can_b = stage_fits(t, b);
        const int stage = uni(((a_short && can_a) || (!can_b && can_a)) ? 1 : can_b ? 2 : 0);   // (wave-uniform, and made so for the compiler: see mul)
        if (stage == 1) stage_rows_of(t, a, lane); else if (stage == 2) stage_rows_of(t, b, lane);
        // a dedicated helper walks the upper part of the sorted terms (whole-PZ 3x1 operands)
        const bool shared = t.hch != nullptr && t.hded && N >= t.hmin;
        int S = N;
        if (shared) {
            const Wave& w = t.w;
            S = indirect ? split_point(N, t.hnum, 32, [&](int p) { return ev.key_lds(w, w.sidx[p]); }) : split_point(N, t.hnum, 32, [&](int p) { return w.skey[p]; });
            hj_post_common(t, stage == 1 ? HK_CROSS_A_STAGED : stage == 2 ? HK_CROSS_B_STAGED : HK_CROSS_UNSTAGED, S, N, indirect, out);
#ifdef TV_PROFILE
            t.n_shared += 1; t.n_shared_terms += N;
#endif
            hj_post_seg(t, 0, a, 1.0, -1); hj_post_seg(t, 1, b, 1.0, -1);
            hj_signal(t, &t.hch[HJ_SEQ], t.hseq);
        }
        if (stage == 1) cross_walk<1>(t, t.w, t.stage, 0, S, indirect, ev, a, b, &o, rad);
        else if (stage == 2) cross_walk<2>(t, t.w, t.stage, 0, S, indirect, ev, a, b, &o, rad);
        else cross_walk<0>(t, t.w, t.stage, 0, S, indirect, ev, a, b, &o, rad);
        if (shared) {   // the helper's partial radii: rows 0..11 of its partial-sum block, |coefficient| sums in rows 12..14
            const GLB_AS double* hdr;
            const int nh = hj_collect(t, o.n, hdr, out);
#pragma unroll
            for (int e = 0; e < 12; e++) rad[e] += hdr[(size_t)e * GR + rl];
#pragma unroll
            for (int e = 0; e < 3; e++) o.asum[e] += hdr[(size_t)(12 + e) * GR + rl];
            o.n += nh;
        }
    }
    TVP_END(t, N, o.n, 1)
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const double i0 = baseP[2 * c] + rad[2 * c], i1 = baseP[2 * c + 1] + rad[2 * c + 1];
        const double j0 = base2P[2 * c] + rad[2 * c], j1 = base2P[2 * c + 1] + rad[2 * c + 1];
        const double ir = (i0 * 1.0 + i1 * 1.0) + rad[6 + c], jr = (j0 * 1.0 + j1 * 1.0) + rad[6 + c];
        st_hdr(out, H_CEN, c, rl, 0.0 + (1.0 * cenP[2 * c] + -1.0 * cenP[2 * c + 1]));
        st_hdr(out, H_IND, c, rl, (0.0 + ir) + rad[9 + c]);
        st_hdr(out, H_IND2, c, rl, (0.0 + jr) + rad[9 + c]);
    }
    o.finish(t, out);
}

// ---------------------------------------------------------------------------------------------------------------------
// sums: out = ((s0*x0 + s1*x1) + s2*x2) + ..., simplify() after every `+` when CHAIN, once at the end otherwise
// (RT/PZsparse.cu:743-834,996-1030,1068-1116; the staging is described in pz_wave.h above lincomb_chain).
struct TSeg {
    TView v;
    double scale;
    int comp;  // -1: same shape as the result; otherwise a 1x1 source embedded into entry `comp`
};

// XK >= 0: source XK is not read as it is but as cross(source, b) with a constant vector b -- the constant cross product of pz_tv.h's
// cross_const (a x b: out[c] = sA[c] * x[cA[c]] + sB[c] * x[cB[c]], cA = {1, 2, 0}, cB = {2, 0, 1}) with its two simplify() stages, applied to
// every term of the source as the walk meets it instead of being written out as a PZ of its own first and read back here (its rows were two
// fifths of a block's row writes).  The source's keys are the operand's: a monomial the cross product would have pruned in every lane is a
// member with coefficient 0 in every lane, which changes no sum and no verdict.  The pruned amounts xr1 / xr2 accumulate in the operand's key
// order, as cross_const accumulates them.
template <int SZ, int NS, bool CHAIN, int XK = -1>
struct LinCtx {
    TSeg s[NS];
    int off[NS + 1];
    int lane;
    double thr, thr_sq;
    bool active;
    Out<SZ>* o;
    double acc[SZ], ra[NS][SZ];
    mutable double mabs = __builtin_inf();   // prune margin of this walk (mtrk / mfold)
    double xsA[XK >= 0 ? 3 : 1], xsB[XK >= 0 ? 3 : 1], xr1[XK >= 0 ? 3 : 1], xr2[XK >= 0 ? 3 : 1];   // XK >= 0: the constants and the two stages' pruned amounts
    bool present;   // per lane
    int last;       // source of the run's latest member (wave-uniform)
    struct Regs { double x[SZ]; int k; };   // k: the term's source (wave-uniform)
    __device__ inline int seg_of(int idx) const {
        int k = 0;
#pragma unroll
        for (int i = 1; i < NS; i++) k += (idx >= off[i]) ? 1 : 0;
        return k;
    }
    // row0[k] = address of source k's row of raw term 0 WITHOUT the lane offset, i.e. coef + (off - first * stride) rows, set by prepare():
    // a raw term's rows then start at row0 + idx * stride rows.  A 1x1 source loads its one row SZ times rather than branching (term()
    // reads x[0] only).
    const GLB_AS double* row0[NS];
    __device__ inline void prepare() {
#pragma unroll
        for (int k = 0; k < NS; k++) row0[k] = s[k].v.coef + ((ptrdiff_t)s[k].v.off - (ptrdiff_t)off[k] * s[k].v.stride) * GR;
    }
    // per-lane descriptors of the chunk's terms (lane l describes the chunk's l-th term): source, byte address of its first row, row step
    struct Desc { int k; unsigned lo, hi; int step; };
    __device__ static inline Desc select_desc(bool first, const Desc& x, const Desc& y) {
        Desc d;
        d.k = first ? x.k : y.k; d.lo = first ? x.lo : y.lo; d.hi = first ? x.hi : y.hi; d.step = first ? x.step : y.step;
        return d;
    }
    __device__ inline Desc describe(int idx_lane) const {
        Desc d;
        d.k = seg_of(idx_lane);
        const GLB_AS double* base = row0[0];
        int stride = s[0].v.stride, comp = s[0].comp;
#pragma unroll
        for (int q = 1; q < NS; q++) {
            const bool me = (d.k == q);
            base = me ? row0[q] : base;
            stride = me ? s[q].v.stride : stride; comp = me ? s[q].comp : comp;
        }
        const uint64_t a = (uint64_t)(base + (ptrdiff_t)idx_lane * stride * GR);
        d.lo = (unsigned)a; d.hi = (unsigned)(a >> 32);
        d.step = comp < 0 ? GR : 0;
        return d;
    }
    // Only the loads: what is loaded must not be touched here, or the wave would wait for it before issuing the next term's loads.
    __device__ inline void load(const Desc& d, int l, int, Regs& r) const {
        r.k = __builtin_amdgcn_readlane(d.k, l);
        const uint64_t a = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)d.hi, l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)d.lo, l);
        const int step = __builtin_amdgcn_readlane(d.step, l);
        const GLB_AS double* src = (const GLB_AS double*)a + lane;
#pragma unroll
        for (int e = 0; e < SZ; e++) r.x[e] = src[e * step];
    }
    // scale * embed(source entry): pure -- the walk forms several terms' at once (walk_sorted).  XK: the pruned amounts of the constant
    // cross product's two stages come back in x1 / x2 and join xr1 / xr2 in accum(), i.e. in the operand's key order as before.
    struct Prod {
        double c[SZ], x1[XK >= 0 ? 3 : 1], x2[XK >= 0 ? 3 : 1];
        __device__ inline void pin() { for (int e = 0; e < SZ; e++) __asm__ volatile("" : "+v"(c[e])); }
    };
    __device__ inline void prod(const Regs& r, Prod& p) const {
        const int k = r.k;
        double* c = p.c;
        if constexpr (XK >= 0) {
            static_assert(SZ == 3, "a constant cross product is a 3x1 source");
#pragma unroll
            for (int q = 0; q < 3; q++) { p.x1[q] = 0.0; p.x2[q] = 0.0; }
            if (k == XK) {   // (wave-uniform) cross_const's arithmetic on this term, in its order
                const double x0 = r.x[0], x1 = r.x[1], x2 = r.x[2];
                const double xa[3] = {x1, x2, x0}, xb[3] = {x2, x0, x1};
                double rr[3];
                bool anyc = false;
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    double v = xsA[q] * xa[q];
                    v += xsB[q] * xb[q];
                    const bool small = fabs(v) <= thr;
                    mtrk(mabs, thr_sq, v * v);
                    p.x1[q] = small ? fabs(v) : 0.0;
                    rr[q] = small ? 0.0 : v;
                    anyc = anyc || !small;
                }
                const double s3 = rr[0] * rr[0] + rr[1] * rr[1] + rr[2] * rr[2];
                mtrk(mabs, thr_sq, s3);
                const bool keep = anyc && !(s3 <= thr_sq) && active;
#pragma unroll
                for (int q = 0; q < 3; q++) { p.x2[q] = (anyc && !keep) ? fabs(rr[q]) : 0.0; c[q] = s[XK].scale * (keep ? rr[q] : 0.0); }
                return;
            }
        }
        int comp = s[0].comp;
        double scale = s[0].scale;
#pragma unroll
        for (int q = 1; q < NS; q++) { const bool me = (k == q); comp = me ? s[q].comp : comp; scale = me ? s[q].scale : scale; }
        // (no branch on the wave-uniform comp: an embedded 1x1 source has its one row in every x[e] -- load() reads it SZ times --
        //  so entry comp is scale * x[0] either way)
#pragma unroll
        for (int e = 0; e < SZ; e++) { const double x = scale * r.x[e]; c[e] = (comp < 0 || e == comp) ? x : 0.0; }
    }
    __device__ inline bool is_small() const {   // (a lane that holds no term holds zeros or a sum an earlier stage dropped: tracked again, harmlessly)
        double q = 0.0;
#pragma unroll
        for (int e = 0; e < SZ; e++) q += acc[e] * acc[e];
        bool small;
        if constexpr (SZ == 1) small = fabs(acc[0]) <= thr; else small = q <= thr_sq;
        mtrk(mabs, thr_sq, q);
        return small;
    }
    // simplify() of stage k (1 <= k < NS, wave-uniform) on what has been accumulated so far: a lane whose sum is small drops it into that
    // stage's radius (selects, not branches on the per-lane verdict; the stage index is wave-uniform and picks the radius registers)
    __device__ inline void stage_at(int k) {
        const bool sm = is_small();   // (every lane: no branch on the per-lane `present`)
        const bool drop = present && sm;
        double v[SZ];
#pragma unroll
        for (int e = 0; e < SZ; e++) v[e] = drop ? fabs(acc[e]) : 0.0;
#pragma unroll
        for (int kk = 1; kk < NS; kk++) {
            const bool me = (kk == k);
#pragma unroll
            for (int e = 0; e < SZ; e++) ra[kk][e] += me ? v[e] : 0.0;
        }
        present = present && !drop;
    }
    // The composed simplify() stages of a run of equal keys (pz_wave.h lincomb_chain): after source k (k >= 1) has been added -- or found
    // absent -- what is accumulated is tested.  Between two tests without a new member the sum does not change, so of the stages
    // (last, rk) that have no member of this key only the FIRST can prune (a sum it keeps passes the later ones unchanged, a sum it
    // drops is gone): one evaluation where the first version of this walk did NS - 1, identical results.
    template <int PB>
    __device__ inline void prod_batch(const Regs* r, Prod* p) const {   // (one multiplication deep: nothing to interleave)
#pragma unroll
        for (int v = 0; v < PB; v++) prod(r[v], p[v]);
    }
    __device__ inline void accum(const Regs& r, const Prod& p, bool first) {
        const double* c = p.c;
        const int rk = r.k;
        if constexpr (XK >= 0) {   // (zeros from every other source's terms: adding 0.0 to a non-negative sum changes nothing)
#pragma unroll
            for (int q = 0; q < 3; q++) { xr1[q] += p.x1[q]; xr2[q] += p.x2[q]; }
        }
        if constexpr (CHAIN) {
            const int kf = last + 1 > 1 ? last + 1 : 1;
            if (!first && kf < rk) stage_at(kf);   // (wave-uniform condition; `first`: nothing accumulated yet)
        }
#pragma unroll
        for (int e = 0; e < SZ; e++) acc[e] = (present && !first) ? acc[e] + c[e] : c[e];
        present = true;
        if constexpr (CHAIN) {
            if (rk >= 1) stage_at(rk);
        }
        last = rk;
    }
    __device__ inline void close(pzkey_t key) {
        if constexpr (CHAIN) {
            const int kf = last + 1 > 1 ? last + 1 : 1;
            if (kf < NS) stage_at(kf);
        } else {
            // one simplify() at the end: pruned terms go to ra[0]
            const bool sm = is_small();
            const bool drop = present && sm;
#pragma unroll
            for (int e = 0; e < SZ; e++) ra[0][e] += drop ? fabs(acc[e]) : 0.0;
            present = present && !drop;
        }
        const bool keep = present && active;
        double v[SZ];
#pragma unroll
        for (int e = 0; e < SZ; e++) v[e] = keep ? acc[e] : 0.0;
        if (__ballot(keep) != 0ull) o->emit(key, v);
    }
};

// xb (XK >= 0): the constant vector b of the cross product source XK stands for
template <int SZ, int NS, bool CHAIN, int XK = -1>
__device__ TV_NOINLINE void lincomb(TW& t, const TPZ& out, const TSeg* segs, const double* xb = nullptr) {
    PZ_KEEP_RETURN_ADDRESS();
    TVP_FN(t, 3)
    const int lane = t.w.lane;
    const int rl = t.rl;   // (the lane's place in a row)
    LinCtx<SZ, NS, CHAIN, XK> cx;
    pzw::LinEval<SZ, NS> ev;
    int N = 0;
    double cen[SZ], indk[NS][SZ], ind2k[NS][SZ];
    double xind[3] = {0.0, 0.0, 0.0}, xind2[3] = {0.0, 0.0, 0.0};   // XK >= 0: the cross product's radii before its pruned amounts
#pragma unroll
    for (int e = 0; e < SZ; e++) cen[e] = 0.0;
#pragma unroll
    for (int k = 0; k < NS; k++) {
        cx.s[k] = segs[k];
        cx.s[k].v.cnt = uni(cx.s[k].v.cnt); cx.s[k].v.stride = uni(cx.s[k].v.stride); cx.s[k].v.off = uni(cx.s[k].v.off); cx.s[k].comp = uni(cx.s[k].comp);
        cx.s[k].v.coef = uni_ptr(cx.s[k].v.coef);
        ev.s[k].v = kview(cx.s[k].v); ev.s[k].scale = segs[k].scale; ev.s[k].comp = segs[k].comp;
        cx.off[k] = N; ev.off[k] = N;
        N += cx.s[k].v.cnt;
        const TView& v = cx.s[k].v;
        const double sc = segs[k].scale, asc = fabs(sc);
#pragma unroll
        for (int e = 0; e < SZ; e++) { indk[k][e] = 0.0; ind2k[k][e] = 0.0; }
        if (XK >= 0 && k == XK) {
            if constexpr (XK >= 0) {
                // centre and radii of cross(source, b) as cross_const forms them; its pruned amounts join the radii after the walk
                const double b0 = xb[0], b1 = xb[1], b2 = xb[2];
                cx.xsA[0] = b2; cx.xsA[1] = b0; cx.xsA[2] = b1; cx.xsB[0] = -b1; cx.xsB[1] = -b2; cx.xsB[2] = -b0;
                double x0[3], i0[3], j0[3];
#pragma unroll
                for (int q = 0; q < 3; q++) { x0[q] = ld_hdr(v, H_CEN, q, rl); i0[q] = ld_hdr(v, H_IND, q, rl); j0[q] = ld_hdr(v, H_IND2, q, rl); cx.xr1[q] = 0.0; cx.xr2[q] = 0.0; }
                const int cA[3] = {1, 2, 0}, cB[3] = {2, 0, 1};
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    const double cq = sc * (cx.xsA[q] * x0[cA[q]] + cx.xsB[q] * x0[cB[q]]);
                    cen[q] = (k == 0) ? cq : cen[q] + cq;
                    xind[q] = i0[cA[q]] * fabs(cx.xsA[q]) + i0[cB[q]] * fabs(cx.xsB[q]);
                    xind2[q] = j0[cA[q]] * fabs(cx.xsA[q]) + j0[cB[q]] * fabs(cx.xsB[q]);
                }
            }
        } else if (segs[k].comp < 0) {
#pragma unroll
            for (int e = 0; e < SZ; e++) {
                const double c = sc * ld_hdr(v, H_CEN, e, rl);
                cen[e] = (k == 0) ? c : cen[e] + c;
                indk[k][e] = ld_hdr(v, H_IND, e, rl) * asc; ind2k[k][e] = ld_hdr(v, H_IND2, e, rl) * asc;
            }
        } else {
            const double c = sc * ld_hdr(v, H_CEN, 0, rl), i1 = ld_hdr(v, H_IND, 0, rl) * asc, i2 = ld_hdr(v, H_IND2, 0, rl) * asc;
#pragma unroll
            for (int e = 0; e < SZ; e++)
                if (e == segs[k].comp) { cen[e] = cen[e] + c; indk[k][e] = i1; ind2k[k][e] = i2; }
        }
    }
    cx.off[NS] = N; ev.off[NS] = N;
    cx.lane = rl;
    cx.prepare();
    WSYNC();  // every lane has read the sources' header rows before `out` (possibly one of them) is written
    Out<SZ> o;
    o.init(out, rl);
    cx.lane = rl; cx.thr = t.w.thr; cx.thr_sq = t.w.thr_sq; cx.active = t.active; cx.o = &o;
    cx.present = false; cx.last = -1;
#pragma unroll
    for (int e = 0; e < SZ; e++) cx.acc[e] = 0.0;
#pragma unroll
    for (int k = 0; k < NS; k++)
#pragma unroll
        for (int e = 0; e < SZ; e++) cx.ra[k][e] = 0.0;
    bool indirect = false;
    TVP_T0
    N = pzw::sort_terms(t.w, N, ev, indirect);
    TVP_T1
    const Wave& w = t.w;
    constexpr int kU = SZ <= 3 ? 8 : 4;
    // the sums of the backward recursions (f = R f + F; n = ((N + R n) + c x F) + p x R f) share their walk with an idle wave
    // (the three-term chained sums of the forward pass: with a dedicated helper only)
    constexpr bool kSplittable = XK < 0 && SZ == 3 && ((NS == 2 && !CHAIN) || (NS == 4 && CHAIN) || (NS == 3 && CHAIN));
    bool shared = false;
    int S = N;
    if constexpr (kSplittable) {
        if (t.hch != nullptr && (NS != 3 || t.hded)) {
            if (N >= t.hmin) {
                shared = true;
                S = indirect ? split_point(N, t.hnum, 32, [&](int p) { return ev.key_lds(w, w.sidx[p]); }) : split_point(N, t.hnum, 32, [&](int p) { return w.skey[p]; });
                hj_post_common(t, !CHAIN ? HK_LIN2 : NS == 4 ? HK_LIN4_CHAIN : HK_LIN3_CHAIN, S, N, indirect, out);
#ifdef TV_PROFILE
                t.n_shared += 1; t.n_shared_terms += N;
#endif
#pragma unroll
                for (int k = 0; k < NS; k++) hj_post_seg(t, k, cx.s[k].v, cx.s[k].scale, cx.s[k].comp);
                hj_signal(t, &t.hch[HJ_SEQ], t.hseq);
            } else hj_post_none(t);
        }
    }
    if (indirect) walk_sorted<kU>(lane, S, [&](int p) { return ev.key_lds(w, w.sidx[p]); }, [&](int p) { return (int)w.sidx[p]; }, cx);
    else walk_sorted<kU>(lane, S, [&](int p) { return w.skey[p]; }, [&](int p) { return (int)w.sidx[p]; }, cx);
    if constexpr (kSplittable) {
        if (shared) {   // the helper's rows are behind ours now; its partial sums of the pruned amounts, stage by stage, and of |coefficient|
            const GLB_AS double* hdr;
            const int nh = hj_collect(t, o.n, hdr, out);
#pragma unroll
            for (int e = 0; e < SZ; e++) {
                if constexpr (CHAIN) {
                    cx.ra[1][e] += hdr[((size_t)H_IND * SZ + e) * GR + rl];
                    cx.ra[2][e] += hdr[((size_t)H_IND2 * SZ + e) * GR + rl];
                    if constexpr (NS == 4) cx.ra[3][e] += hdr[((size_t)H_CEN * SZ + e) * GR + rl];
                } else cx.ra[0][e] += hdr[((size_t)H_IND * SZ + e) * GR + rl];
                o.asum[e] += hdr[((size_t)H_ASUM * SZ + e) * GR + rl];
            }
            o.n += nh;
        }
    }
    TVP_END(t, N, o.n, 2)
    mfold(t, cx);
    if constexpr (XK >= 0) {   // (cross_const: ind = (ind + ra1) + ra2, then this sum's |scale|)
        const double asc = fabs(cx.s[XK].scale);
#pragma unroll
        for (int q = 0; q < 3; q++) { indk[XK][q] = ((xind[q] + cx.xr1[q]) + cx.xr2[q]) * asc; ind2k[XK][q] = ((xind2[q] + cx.xr1[q]) + cx.xr2[q]) * asc; }
    }
#pragma unroll
    for (int e = 0; e < SZ; e++) {
        double r, r2;
        if constexpr (CHAIN) {
            // stage 1: (i0 + i1) + pruned; stage k: (previous + ik) + pruned
            r = indk[0][e]; r2 = ind2k[0][e];
#pragma unroll
            for (int k = 1; k < NS; k++) { r = (r + indk[k][e]) + cx.ra[k][e]; r2 = (r2 + ind2k[k][e]) + cx.ra[k][e]; }
        } else {
            r = indk[0][e]; r2 = ind2k[0][e];
#pragma unroll
            for (int k = 1; k < NS; k++) { r = r + indk[k][e]; r2 = r2 + ind2k[k][e]; }
            r = r + cx.ra[0][e]; r2 = r2 + cx.ra[0][e];
        }
        st_hdr(out, H_CEN, e, rl, cen[e]);
        st_hdr(out, H_IND, e, rl, r);
        st_hdr(out, H_IND2, e, rl, r2);
    }
    o.finish(t, out);
}

// ---------------------------------------------------------------------------------------------------------------------
// cross of a 3x1 PZ with a constant vector, either order (RT/PZsparse.cu:1118-1132, 1153-1167): out[c] = sA[c]*a[cA[c]] +
// sB[c]*a[cB[c]]; the key list is a's, so one ordered pass (see pz_wave.h cross_const for the two simplify() stages)
// (monomials [m_lo, m_hi) of a: the whole list, or the part of it one of two waves takes)
__device__ inline void cross_const_range(TW& t, const TView& a, int m_lo, int m_hi, const double* sA, const int* cA, const double* sB, const int* cB, Out<3>& o, double* ra1, double* ra2) {
    const int lane = t.w.lane;
    const int rl = t.rl;   // (the lane's place in a row)
    const double thr = t.w.thr, thr_sq = t.w.thr_sq;
    const bool active = t.active;
    double mabs = __builtin_inf();   // prune margin of this pass (mtrk), folded into t.w below
    // (a's rows in their own order, two half-batches of eight monomials in flight alternately: see walk_sorted)
    double xa[8][3], xb[8][3];
    auto issue_c = [&](double (*x)[3], int m) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int mm = min(m + u, m_hi - 1);
#pragma unroll
            for (int q = 0; q < 3; q++) x[u][q] = ld_coef(a, mm, q, rl);
        }
    };
    auto consume_c = [&](double (*x)[3], int m, pzkey_t key_v, int m0) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (m + u < m_hi) {
                double r[3];
                bool anyc = false;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    double xa_ = 0, xb_ = 0;
#pragma unroll
                    for (int q = 0; q < 3; q++) { if (q == cA[c]) xa_ = x[u][q]; if (q == cB[c]) xb_ = x[u][q]; }
                    double v = sA[c] * xa_;
                    v += sB[c] * xb_;
                    const bool small = fabs(v) <= thr;
                    mtrk(mabs, thr_sq, v * v);
                    ra1[c] += small ? fabs(v) : 0.0;
                    r[c] = small ? 0.0 : v;
                    anyc = anyc || !small;
                }
                const double s3 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
                mtrk(mabs, thr_sq, s3);
                const bool keep = anyc && !(s3 <= thr_sq) && active;
#pragma unroll
                for (int c = 0; c < 3; c++) { ra2[c] += (anyc && !keep) ? fabs(r[c]) : 0.0; r[c] = keep ? r[c] : 0.0; }
                if (__ballot(keep) != 0ull) o.emit(pzkey_readlane(key_v, m + u - m0), r);
            }
        }
    };
    if (m_lo < m_hi) issue_c(xa, m_lo);
    for (int m0 = m_lo; m0 < m_hi; m0 += WAVE) {
        const pzkey_t key_v = m0 + lane < m_hi ? a.keys[m0 + lane] : 0ull;
        const int n = min(WAVE, m_hi - m0);
        for (int l0 = 0; l0 < n; l0 += 16) {
            issue_c(xb, m0 + l0 + 8);
            consume_c(xa, m0 + l0, key_v, m0);
            issue_c(xa, m0 + l0 + 16);
            consume_c(xb, m0 + l0 + 8, key_v, m0);
        }
    }
    t.w.mabs = fmin(t.w.mabs, mabs);
}
__device__ TV_NOINLINE void cross_const(TW& t, const TPZ& out, const TView& a_, const double* sA, const int* cA, const double* sB, const int* cB) {
    PZ_KEEP_RETURN_ADDRESS();
    TVP_FN(t, 4)
    const int lane = t.w.lane;
    const int rl = t.rl;   // (the lane's place in a row)
    TView a = a_;
    a.cnt = uni(a.cnt);
    double x0[3], cen[3], ind[3], ind2[3], ra1[3] = {0, 0, 0}, ra2[3] = {0, 0, 0};
    double i0[3], j0[3];
#pragma unroll
    for (int c = 0; c < 3; c++) { x0[c] = ld_hdr(a, H_CEN, c, rl); i0[c] = ld_hdr(a, H_IND, c, rl); j0[c] = ld_hdr(a, H_IND2, c, rl); }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        double xa = 0, xb = 0, ia = 0, ib = 0, ja = 0, jb = 0;
#pragma unroll
        for (int q = 0; q < 3; q++) {
            if (q == cA[c]) { xa = x0[q]; ia = i0[q]; ja = j0[q]; }
            if (q == cB[c]) { xb = x0[q]; ib = i0[q]; jb = j0[q]; }
        }
        cen[c] = sA[c] * xa + sB[c] * xb;
        ind[c] = ia * fabs(sA[c]) + ib * fabs(sB[c]);
        ind2[c] = ja * fabs(sA[c]) + jb * fabs(sB[c]);
    }
    WSYNC();
    Out<3> o;
    o.init(out, rl);
#ifdef TV_PROFILE_FULL
    const long long cc0__ = clock64();
#endif
    // a dedicated helper takes the upper part of a's monomials (the key list is a's: no sort, the split is one of the list)
    const bool shared = t.hch != nullptr && t.hded && 2 * a.cnt >= t.hmin;
    int S = a.cnt;
    if (shared) {
        S = uni((int)((long long)a.cnt * t.hnum / 32));
        hj_post_common(t, HK_CROSS_CONST, S, a.cnt, false, out);
        hj_post_seg(t, 0, a, 1.0, -1);
        if (lane == 0) {
            LDS_AS int* sg = t.hch + HJ_SEG0 + HJ_SEG_WORDS;   // source 1: a's key list, the last component indices; sources 2, 3: the scales and the other indices
            lds_st_ptr(&sg[0], a.keys); sg[2] = cA[2]; sg[3] = cB[2];
            for (int q = 0; q < 3; q++) {
                const uint64_t ua = (uint64_t)__double_as_longlong(sA[q]), ub = (uint64_t)__double_as_longlong(sB[q]);
                sg[HJ_SEG_WORDS + 2 * q] = (int)(unsigned)ua; sg[HJ_SEG_WORDS + 2 * q + 1] = (int)(unsigned)(ua >> 32);
                sg[2 * HJ_SEG_WORDS + 2 * q] = (int)(unsigned)ub; sg[2 * HJ_SEG_WORDS + 2 * q + 1] = (int)(unsigned)(ub >> 32);
            }
            sg[HJ_SEG_WORDS + 6] = cA[0]; sg[HJ_SEG_WORDS + 7] = cA[1]; sg[2 * HJ_SEG_WORDS + 6] = cB[0]; sg[2 * HJ_SEG_WORDS + 7] = cB[1];
        }
        hj_signal(t, &t.hch[HJ_SEQ], t.hseq);
    }
    cross_const_range(t, a, 0, S, sA, cA, sB, cB, o, ra1, ra2);
    if (shared) {   // the helper's partial sums: rows 0..2 first stage, 3..5 second stage, 6..8 |coefficient|
        const GLB_AS double* hdr;
        const int nh = hj_collect(t, o.n, hdr, out);
#pragma unroll
        for (int c = 0; c < 3; c++) { ra1[c] += hdr[(size_t)c * GR + rl]; ra2[c] += hdr[(size_t)(3 + c) * GR + rl]; o.asum[c] += hdr[(size_t)(6 + c) * GR + rl]; }
        o.n += nh;
    }
#ifdef TV_PROFILE_FULL
    t.c_cc += clock64() - cc0__;
#endif
#pragma unroll
    for (int c = 0; c < 3; c++) {
        st_hdr(out, H_CEN, c, rl, cen[c]);
        st_hdr(out, H_IND, c, rl, (ind[c] + ra1[c]) + ra2[c]);
        st_hdr(out, H_IND2, c, rl, (ind2[c] + ra1[c]) + ra2[c]);
    }
    o.finish(t, out);
}

// ---------------------------------------------------------------------------------------------------------------------
// The helper's side of a shared walk ("One walk on two waves" above): wait for the primary's next operator on this wave's channel;
// if it shares its walk, walk the upper part of its sorted terms into `tmp` (a scratch slot of this wave), leave the partial sums in
// tmp's header rows, and move the rows down behind the primary's once it has posted its count.  Returns the job's kind.
// (tmp.hdr: 12 rows in a 3x1 work slot -- enough for the kinds an opportunistic helper serves -- and 16 in a dedicated helper's slot)
__device__ inline TView hj_seg_view(LDS_AS int* ch, int k, int sz) {
    LDS_AS int* sg = ch + HJ_SEG0 + k * HJ_SEG_WORDS;
    TView v;
    v.keys = nullptr; v.hdr = nullptr;
    v.coef = lds_ld_ptr<const GLB_AS double>(&sg[0]);
    v.cnt = uni(sg[2]); v.stride = uni(sg[3]); v.off = uni(sg[4]); v.sz = sz;
    return v;
}
__device__ inline double hj_seg_double(LDS_AS int* p) { return __longlong_as_double((long long)(((uint64_t)(unsigned)p[1] << 32) | (unsigned)p[0])); }
template <int NS, bool CHAIN>
__device__ inline void serve_lincomb(TW& t, LDS_AS int* ch, const Wave& sw, int S, int N, bool indirect, Out<3>& o, double (&ra)[4][3]) {
    const int lane = t.w.lane;
    const int rl = t.rl;   // (the lane's place in a row)
    LinCtx<3, NS, CHAIN> cx;
    pzw::LinEval<3, NS> ev;   // (key_lds reads the primary's key buffer only)
    int tot = 0;
#pragma unroll
    for (int k = 0; k < NS; k++) {
        LDS_AS int* sg = ch + HJ_SEG0 + k * HJ_SEG_WORDS;
        cx.s[k].v = hj_seg_view(ch, k, 3); cx.s[k].comp = uni(sg[5]);
        cx.s[k].scale = hj_seg_double(&sg[6]);
        cx.off[k] = tot; tot += cx.s[k].v.cnt;
    }
    cx.off[NS] = tot;
    cx.lane = rl; cx.prepare();
    cx.thr = t.w.thr; cx.thr_sq = t.w.thr_sq; cx.active = t.active; cx.o = &o; cx.present = false; cx.last = -1;
#pragma unroll
    for (int e = 0; e < 3; e++) cx.acc[e] = 0.0;
#pragma unroll
    for (int k = 0; k < NS; k++)
#pragma unroll
        for (int e = 0; e < 3; e++) cx.ra[k][e] = 0.0;
    if (indirect) walk_sorted<8>(lane, N, [&](int p) { return ev.key_lds(sw, sw.sidx[p]); }, [&](int p) { return (int)sw.sidx[p]; }, cx, S);
    else walk_sorted<8>(lane, N, [&](int p) { return sw.skey[p]; }, [&](int p) { return (int)sw.sidx[p]; }, cx, S);
#pragma unroll
    for (int k = 0; k < NS; k++)
#pragma unroll
        for (int e = 0; e < 3; e++) ra[k][e] = cx.ra[k][e];
    mfold(t, cx);
}
__device__ TV_NOINLINE int serve_walk(TW& t, const TPZ& tmp) {
    PZ_KEEP_RETURN_ADDRESS();
    TVP_FN(t, 5)
    LDS_AS int* ch = t.hch;
    const int lane = t.w.lane;
    const int rl = t.rl;   // (the lane's place in a row)
    t.hseq++;
    hj_wait(t, &ch[HJ_SEQ], t.hseq);
    const int kind = uni(ch[HJ_KIND]);
    if (kind == HK_NONE || kind == HK_BAR || kind == HK_EXIT) { hj_signal(t, &ch[HJ_DONE], t.hseq); return kind; }
    const int S = uni(ch[HJ_S]), N = uni(ch[HJ_N]);
    const bool indirect = uni(ch[HJ_INDIRECT]) != 0;
    Wave sw = t.w;   // the PRIMARY's sorted terms
    sw.skey = (LDS_AS pzkey_t*)(uintptr_t)(unsigned)uni(ch[HJ_SKEY]);
    sw.sidx = (LDS_AS uint16_t*)(uintptr_t)(unsigned)uni(ch[HJ_SIDX]);
    const LDS_AS double* stage = (const LDS_AS double*)(uintptr_t)(unsigned)uni(ch[HJ_STAGE]);
    Out<3> o;
    o.init(tmp, rl);
    GLB_AS double* pp = tmp.hdr + rl;   // partial sums, row q at pp[q * GR]
    if (kind == HK_MUL_3331_A_STAGED || kind == HK_MUL_3331_B_STAGED || kind == HK_MUL_3331_UNSTAGED) {
        typedef pzw::MulShape<3, 3, 3, 1> SH;
        const TView a = hj_seg_view(ch, 0, 9), b = hj_seg_view(ch, 1, 3);
        pzw::MulEval<SH> ev;
        ev.a = kview(a); ev.set_b(kview(b));
        double rad[3];
        if (kind == HK_MUL_3331_A_STAGED) mul_walk<SH, 1>(t, sw, stage, S, N, indirect, ev, a, b, &o, rad);
        else if (kind == HK_MUL_3331_B_STAGED) mul_walk<SH, 2>(t, sw, stage, S, N, indirect, ev, a, b, &o, rad);
        else mul_walk<SH, 0>(t, sw, stage, S, N, indirect, ev, a, b, &o, rad);
#pragma unroll
        for (int e = 0; e < 3; e++) { pp[(size_t)(H_IND * 3 + e) * GR] = rad[e]; pp[(size_t)(H_ASUM * 3 + e) * GR] = o.asum[e]; }
    } else if (kind == HK_LIN2 || kind == HK_LIN3_CHAIN || kind == HK_LIN4_CHAIN) {
        double ra[4][3];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int e = 0; e < 3; e++) ra[k][e] = 0.0;
        if (kind == HK_LIN2) { serve_lincomb<2, false>(t, ch, sw, S, N, indirect, o, ra); }
        else if (kind == HK_LIN3_CHAIN) serve_lincomb<3, true>(t, ch, sw, S, N, indirect, o, ra);
        else serve_lincomb<4, true>(t, ch, sw, S, N, indirect, o, ra);
        // rows: [H_IND] = the unchained sum's pruned amounts or stage 1's, [H_IND2] = stage 2's, [H_CEN] = stage 3's
#pragma unroll
        for (int e = 0; e < 3; e++) {
            pp[(size_t)(H_IND * 3 + e) * GR] = kind == HK_LIN2 ? ra[0][e] : ra[1][e];
            pp[(size_t)(H_IND2 * 3 + e) * GR] = ra[2][e];
            pp[(size_t)(H_CEN * 3 + e) * GR] = ra[3][e];
            pp[(size_t)(H_ASUM * 3 + e) * GR] = o.asum[e];
        }
    } else if (kind == HK_CROSS_A_STAGED || kind == HK_CROSS_B_STAGED || kind == HK_CROSS_UNSTAGED) {
        typedef pzw::MulShape<1, 1, 1, 1> SH;
        const TView a = hj_seg_view(ch, 0, 3), b = hj_seg_view(ch, 1, 3);
        pzw::MulEval<SH> ev;
        ev.a = kview(a); ev.set_b(kview(b));
        ev.a.stride = 1; ev.a.off = 0; ev.b.stride = 1; ev.b.off = 0;
        double rad[12];
        if (kind == HK_CROSS_A_STAGED) cross_walk<1>(t, sw, stage, S, N, indirect, ev, a, b, &o, rad);
        else if (kind == HK_CROSS_B_STAGED) cross_walk<2>(t, sw, stage, S, N, indirect, ev, a, b, &o, rad);
        else cross_walk<0>(t, sw, stage, S, N, indirect, ev, a, b, &o, rad);
#pragma unroll
        for (int e = 0; e < 12; e++) pp[(size_t)e * GR] = rad[e];
#pragma unroll
        for (int e = 0; e < 3; e++) pp[(size_t)(12 + e) * GR] = o.asum[e];
    } else {   // HK_CROSS_CONST: monomials [S, N) of the operand
        LDS_AS int* sg = ch + HJ_SEG0 + HJ_SEG_WORDS;
        TView a = hj_seg_view(ch, 0, 3);
        a.keys = lds_ld_ptr<const GLB_AS pzkey_t>(&sg[0]);
        double sA[3], sB[3];
        int cA[3], cB[3];
#pragma unroll
        for (int q = 0; q < 3; q++) { sA[q] = hj_seg_double(&sg[HJ_SEG_WORDS + 2 * q]); sB[q] = hj_seg_double(&sg[2 * HJ_SEG_WORDS + 2 * q]); }
        cA[0] = uni(sg[HJ_SEG_WORDS + 6]); cA[1] = uni(sg[HJ_SEG_WORDS + 7]); cA[2] = uni(sg[2]);
        cB[0] = uni(sg[2 * HJ_SEG_WORDS + 6]); cB[1] = uni(sg[2 * HJ_SEG_WORDS + 7]); cB[2] = uni(sg[3]);
        double ra1[3] = {0, 0, 0}, ra2[3] = {0, 0, 0};
        cross_const_range(t, a, S, N, sA, cA, sB, cB, o, ra1, ra2);
#pragma unroll
        for (int c = 0; c < 3; c++) { pp[(size_t)c * GR] = ra1[c]; pp[(size_t)(3 + c) * GR] = ra2[c]; pp[(size_t)(6 + c) * GR] = o.asum[c]; }
    }
    const int nh = uni(o.n < tmp.cap ? o.n : tmp.cap);
    if (o.n > tmp.cap) pzw::flag(t.w, pzw::ERR_SLOT_OVERFLOW);
    if (lane == 0) { ch[HJ_NH] = nh; lds_st_ptr(&ch[HJ_HDR], tmp.hdr); lds_st_ptr(&ch[HJ_TMP_COEF], tmp.coef); }
    hj_signal(t, &ch[HJ_WALKED], t.hseq);
    // ... and once the primary has finished its part: our rows behind its n0
    hj_wait(t, &ch[HJ_N0SEQ], t.hseq);
    const int n0 = uni(ch[HJ_N0]), cap = uni(ch[HJ_OUT_CAP]);
    GLB_AS pzkey_t* ok = lds_ld_ptr<GLB_AS pzkey_t>(&ch[HJ_OUT_KEYS]);
    GLB_AS double* oc = lds_ld_ptr<GLB_AS double>(&ch[HJ_OUT_COEF]);
    const int room = cap - n0 > 0 ? cap - n0 : 0, ncopy = nh < room ? nh : room;   // (an overflow of the result is flagged by the primary's finish())
    for (int m = lane; m < ncopy; m += WAVE) ok[n0 + m] = tmp.keys[m];
    move_rows(tmp.coef, oc + (size_t)n0 * 3 * GR, rl, (ncopy * 3) / 2, ncopy * 3);   // (the primary moves the lower half)
    hj_signal(t, &ch[HJ_DONE], t.hseq);
    return kind;
}
// a dedicated helper's item: serve until the primary says the item is over, joining every block barrier it enters
__device__ inline void serve_loop(TW& t, const TPZ& tmp) {
    for (;;) {
        const int kind = serve_walk(t, tmp);
        if (kind == HK_BAR) __syncthreads();
        if (kind == HK_EXIT) break;
    }
}

// out = a^T for 3x3 (RT/PZsparse.cu:1050-1066): keys unchanged, no simplify
__device__ TV_NOINLINE void transpose33(TW& t, const TPZ& out, const TPZ& a) {
    PZ_KEEP_RETURN_ADDRESS();
    TVP_FN(t, 6)
    const int lane = t.w.lane;
    const int rl = t.rl;   // (the lane's place in a row)
    const int n = uni(t.w.cnt[a.id]);
    for (int m = 0; m < n; m++)
#pragma unroll
        for (int e = 0; e < 9; e++) out.coef[((size_t)m * 9 + (e % 3) * 3 + e / 3) * GR + rl] = a.coef[((size_t)m * 9 + e) * GR + rl];
    for (int m = lane; m < n; m += WAVE) out.keys[m] = a.keys[m];
#pragma unroll
    for (int h = 0; h < 4; h++)
#pragma unroll
        for (int e = 0; e < 9; e++) out.hdr[((size_t)h * 9 + (e % 3) * 3 + e / 3) * GR + rl] = a.hdr[((size_t)h * 9 + e) * GR + rl];
    if (lane == 0) t.w.cnt[out.id] = n;
    WSYNC();
}

// constant PZ: the same centre / radii in every lane (RT/PZsparse.cu:66-98); ind2 == nullptr: equal to ind
__device__ TV_NOINLINE void set_const(TW& t, const TPZ& out, const double* cen, const double* ind, const double* ind2 = nullptr) {
    PZ_KEEP_RETURN_ADDRESS();
    TVP_FN(t, 6)
    const int lane = t.w.lane;
    const int rl = t.rl;   // (the lane's place in a row)
    for (int e = 0; e < out.sz; e++) {
        st_hdr(out, H_CEN, e, rl, cen ? cen[e] : 0.0);
        st_hdr(out, H_IND, e, rl, ind ? ind[e] : 0.0);
        st_hdr(out, H_IND2, e, rl, ind2 ? ind2[e] : (ind ? ind[e] : 0.0));
        st_hdr(out, H_ASUM, e, rl, 0.0);
    }
    if (lane == 0) t.w.cnt[out.id] = 0;
    WSYNC();
}

}  // namespace tv
