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

#include "pz_key.h"

// Synchronisation INSIDE an operator is wave-local: every operator runs on one 64-lane wave, whose LDS and memory
// operations execute in order, so draining the wave's outstanding loads / stores / LDS traffic is all that the lanes
// need to see each other's data.  (With one wave per block this is what __syncthreads() amounted to; written this way
// the operators can also run in blocks whose waves work on different operators at the same time.)
#define WSYNC() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
// FIRST STATEMENT OF EVERY NON-INLINED DEVICE FUNCTION.  The operator functions are large (tv::mul<3,3,3,3>: 24 000 instructions, 150 KB), and a
// branch across more than 128 KB is relaxed by the compiler (ROCm 7.2 clang) into s_getpc_b64 / s_add_u32 / s_addc_u32 / s_setpc_b64 on a
// SCAVENGED scalar pair -- for which it takes s[30:31], the RETURN ADDRESS of a function that makes no calls, without saving it: once such a
// branch has been taken the function "returns" into its own middle and runs on with whatever the registers hold.  That is the mechanism
// behind the memory-aperture violations of rounds 1-5 (profiles/r05_tv_one_wave_narrow_rows.txt: found with rocgdb on the one-wave kernel with
// narrow rows, whose mul<3,3,3,3> is the first shipped function to cross the limit).  Declaring s30 / s31 clobbered at the entry makes the
// compiler keep the return address elsewhere (a lane of a VGPR, as in functions that call), so that the scavenged pair is free to use.
// tools/check_long_branches.py verifies on every P1 object that no function relaxes a branch through an unsaved s[30:31].
#ifdef PZ_NO_RETURN_ADDRESS_GUARD   // (forensic builds only: the objects of rounds 1-4 as they were, for tools/check_long_branches.py to look at)
#define PZ_KEEP_RETURN_ADDRESS()
#else
#define PZ_KEEP_RETURN_ADDRESS() asm volatile("; return address kept out of s[30:31] (pz_wave.h)" ::: "s30", "s31")
#endif

namespace pzw {

constexpr int WAVE = 64;

// Explicit address spaces: the PZ slots are reached through structs of pointers, which the compiler would
// otherwise treat as generic and lower to flat_load / flat_store (slow for LDS, and unordered against ds_ / global_
// instructions).  LDS_AS -> ds_read / ds_write, GLB_AS -> global_load / global_store.
// The operators are real functions (one copy per shape), not inlined into every call site: the chain kernel issues
// ~1000 of them per time step and fully inlined it is a 300 KB instruction stream that thrashes the 64 KB I-cache.
#define PZW_NOINLINE __attribute__((noinline))
#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))

// status words (global, one array per launch)
enum { ST_ERR = 0, ST_MAX_RAW = 1, ST_MAX_OUT = 2, ST_WORDS = 4 };
#ifdef P1_PROFILE  // development only: per-wave cycle attribution, dumped after the status words
enum { PR_FILL = 0, PR_SORT = 1, PR_EMIT = 2, PR_ABS = 3, PR_CALLS = 4, PR_TERMS = 5, PR_SMALL = 6, PR_TOTAL = 7,
       PR_CYC64 = 8, PR_CYC512 = 9, PR_CYCBIG = 10, PR_N512 = 11, PR_NBIG = 12, PR_TERMS512 = 13, PR_TERMSBIG = 14, PR_S_RANK = 15, PR_S_BITONIC = 16, PR_S_LINMERGE = 17, PR_S_MULMERGE = 18, PR_E_HEAD = 19, PR_E_COEF = 20, PR_E_RUN = 21, PR_E_PRUNE = 22, PR_E_STORE = 23, PR_WORDS = 24 };
#define PROF_CALL_T0 const long long prof_c0__ = clock64();
#define PROF_CALL_END(N) if (w.lane == 0) { const unsigned long long d__ = (unsigned long long)(clock64() - prof_c0__); if ((N) <= 64) w.prof[PR_CYC64] += d__; else if ((N) <= 512) { w.prof[PR_CYC512] += d__; w.prof[PR_N512] += 1; w.prof[PR_TERMS512] += (N); } else { w.prof[PR_CYCBIG] += d__; w.prof[PR_NBIG] += 1; w.prof[PR_TERMSBIG] += (N); } }
#define PROF_T0 const long long prof_t0__ = clock64();
#define PROF_ADD(slot) if (w.lane == 0) w.prof[slot] += (unsigned long long)(clock64() - prof_t0__);
#else
#define PROF_T0
#define PROF_ADD(slot)
#define PROF_CALL_T0
#define PROF_CALL_END(N)
#endif
enum { ERR_RAW_OVERFLOW = 1, ERR_SLOT_OVERFLOW = 2, ERR_TABLE_OVERFLOW = 4, ERR_LINK_GENS = 8, ERR_PAIR = 32, ERR_DBG_BOUNDS = 128,
       ERR_HELPER = 256 /* a time step on two CUs (p1_free.inc.h): the helper block's results did not arrive -- the host builds again on one CU per step */,
       ERR_HELPER_LATE = 512 /* not an error: a main block gave up waiting for its helper to START (the device is shared, or holds fewer CUs than it reports) and built its item alone -- the tables are good, the handle goes back to one CU per step */ };
// -DDBG_BOUNDS (root-cause tooling, tools/dev/gpu_fault_hunt.py): every LDS / arena index of the product merge and of the reduce
// pass is range-checked BEFORE the access; a violation is recorded (flag 128, lstat[3] = code * 2^20 + the offending value's
// low 20 bits, first one wins) and the index clamped to 0, so that a genuine out-of-range index shows up as a report
// instead of a memory fault.
#ifdef DBG_BOUNDS
#define BCHK(w, ok, code, val) ((ok) ? true : (((w).lstat[ST_ERR] & ERR_DBG_BOUNDS) ? false : ((w).lstat[ST_ERR] |= ERR_DBG_BOUNDS, (w).lstat[3] = ((code) << 20) | ((int)(val) & 0xfffff), false)))
#define BIDX(w, idx, lim, code) (BCHK(w, (idx) >= 0 && (idx) < (lim), code, idx) ? (idx) : 0)
#else
#define BIDX(w, idx, lim, code) (idx)
#endif

// A PZ slot (all fields wave-uniform).  id indexes the per-wave LDS count table.
struct PZ {
    GLB_AS pzkey_t* keys;
    GLB_AS double* coef;  // [cap][sz]
    LDS_AS double* cen;   // [sz]
    LDS_AS double* ind;   // [sz] independent (interval) radius, nominal inertial parameters
    LDS_AS double* ind2;  // [sz] the same with the uncertain mass / inertia (see the note on the fused RNEA in p1_reach.hip)
    int sz, cap, id;
};

// Read view of a PZ or of one entry of it (RT/PZsparse.cu:678-697 operator()(r,c) without the copy).
struct View {
    const GLB_AS pzkey_t* keys;
    const GLB_AS double* coef;
    const LDS_AS double* cen;
    const LDS_AS double* ind;
    const LDS_AS double* ind2;
    int cnt, stride, off, sz;
};

struct Wave {
    LDS_AS pzkey_t* skey;   // LDS [cap_key]: raw keys (bitonic / linear-combination merge) or the operands' key lists (product merge)
    LDS_AS uint16_t* sidx;   // LDS [cap_raw]: sorted permutation of the raw terms
    LDS_AS int* cnt;         // LDS per-slot monomial counts
    int cap_raw, cap_key;
    double thr;       // SIMPLIFY_THRESHOLD
    double thr_sq;    // the largest double s with sqrt(s) <= thr (sq_threshold()): `s <= thr_sq` IS `sqrt(s) <= thr`, without the root
#ifdef P1_PROFILE
    unsigned long long prof[PR_WORDS];  // per-wave counters (lane 0's copy is the one that counts)
#endif
    LDS_AS int* lstat;       // LDS [ST_WORDS]: error bits / max raw terms / max monomials of this wave (flushed once per launch)
    // Prune margin (round 6): how close a simplify() verdict of this wave came to flipping.  Per lane, in the squared domain of the norm test: the
    // smallest |s - thr_sq| over the squared norms s the verdicts were taken on -- TWO instructions per verdict (a subtraction and a minimum with
    // the |.| modifier), which is what the time-vectorised walks can afford (a select-based form that kept the pruned and the kept side apart cost
    // the B = 128 build 5.5 %: profiles/r06_prune_margin.txt).  Nothing is gated: a verdict on an exact zero (a lane without the monomial) records
    // thr_sq, i.e. a margin of 1; a stage that looks again at a sum an earlier stage dropped records that sum's distance a second time.
    // The operators track in the register of their local copy (mtrack) and fold it into this wave's LDS row mg[WAVE] when they end (mflush); the
    // kernel reduces the row per work item into the problem's word (margin_item_end).
    double mabs;
    LDS_AS double* mg;       // LDS [WAVE] (nullptr: not tracked)
    int lane;
    // Two waves on one operator (psync() below; the backward pass of the per-step kernel's four-wave blocks, p1_free.inc.h): the operators
    // that support it stride their term loops by `nl` cooperating lanes from `lane2`; skey / sidx are then the FIRST wave's buffers on both.
    int lane2, nl, half;     // lane2 = lane + WAVE * half; nl = WAVE (one wave, the rule) or 2 * WAVE; half = 0 | 1
    LDS_AS int* pair;        // LDS words of the pair (PW_*), nullptr for a wave on its own
};
enum { PW_BAR = 0, PW_BROKEN = 1, PW_E0 = 2, PW_E1 = 3, PW_DBL = 8, PW_DBL_PER_HALF = 10, PW_WORDS = PW_DBL + 1 + 2 * PW_DBL_PER_HALF };
__device__ inline void solo(Wave& w) { w.lane2 = w.lane; w.nl = 64; w.half = 0; w.pair = nullptr; }

// Arguments of the (non-inlined) operators arrive by reference, i.e. as pointers into the caller's stack frame: every `w.field`,
// `a.cnt`, `out.keys` inside a loop is a flat load the compiler may not hoist across the loop's stores, and a count read that way
// is a per-lane value -- as a loop bound it puts the loop under divergent control flow.  The operators therefore start by
// taking local copies (PZW_LOCALS) with the counts moved to scalar registers.
__device__ inline int uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ inline View uni_view(const View& v) { View u = v; u.cnt = uni_i(v.cnt); u.stride = uni_i(v.stride); u.off = uni_i(v.off); return u; }
__device__ inline void uni_caps(struct Wave& w);
#ifdef P1_PROFILE
#define PZW_WAVE_LOCAL(w, w_) Wave& w = w_; w.mabs = __builtin_inf();
#else
#define PZW_WAVE_LOCAL(w, w_) Wave w = w_; uni_caps(w); w.mabs = __builtin_inf();
#endif
__device__ inline void uni_caps(Wave& w) { w.cap_raw = uni_i(w.cap_raw); w.cap_key = uni_i(w.cap_key); w.nl = uni_i(w.nl); w.half = uni_i(w.half); }
// Barrier of the lanes that work on one operator: the wave's own memory traffic drained (WSYNC) and, for a pair of waves, both of them here.
// The counter only grows: a wave that finds it even is the first of its round and waits for the next even value, one that finds it odd
// completes the round.  (A wave cannot be a whole round ahead: it needs the other one's arrival to leave the round it is in.)  A wait that
// never ends -- the two waves disagreeing about the number of barriers, which would be a bug -- is cut off, flagged, and disables the
// pair's barriers for the rest of the launch instead of hanging the chip.
__device__ inline void psync(const Wave& w) {
    WSYNC();
    if (w.nl == WAVE) return;
    if (__hip_atomic_load(&w.pair[PW_BROKEN], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) return;
    int old = 0;
    if (w.lane == 0) old = __hip_atomic_fetch_add(&w.pair[PW_BAR], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    old = __builtin_amdgcn_readfirstlane(old);
    const int target = (old | 1) + 1;
    int spins = 0;
    while (__hip_atomic_load(&w.pair[PW_BAR], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) {
            if (w.lane == 0) { w.lstat[ST_ERR] |= ERR_PAIR; __hip_atomic_store(&w.pair[PW_BROKEN], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
            break;
        }
    }
    asm volatile("" ::: "memory");
}
__device__ inline PZ uni_pz(const PZ& p) { PZ u = p; u.cap = uni_i(p.cap); u.id = uni_i(p.id); return u; }
__device__ inline View view(const Wave& w, const PZ& p) { return View{p.keys, p.coef, p.cen, p.ind, p.ind2, w.cnt[p.id], p.sz, 0, p.sz}; }
__device__ inline View elem(const Wave& w, const PZ& p, int r) { return View{p.keys, p.coef, p.cen, p.ind, p.ind2, w.cnt[p.id], p.sz, r, 1}; }

// Sum over the 64 lanes, returned in every lane.  Data-parallel-primitive moves inside the VALU (quad permutes, row
// mirrors, row broadcasts) instead of six rounds of cross-lane shuffles through the LDS crossbar, which cost ~3 k cycles
// per product operator in abs_sum alone (tools/dev/gpu_pzop_cost.py).  Fixed summation tree: the same result in every launch shape.
#ifdef NO_DPP  /* development: the same data movement through ds_bpermute instead of DPP moves */
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_take(double v) {
    const int lane = (int)(threadIdx.x & 63), row = lane >> 4, li = lane & 15;
    int src = lane; bool valid = true;
    if (CTRL == 0xB1) src = lane ^ 1;
    else if (CTRL == 0x4E) src = lane ^ 2;
    else if (CTRL == 0x141) src = (lane & ~7) | (7 - (lane & 7));
    else if (CTRL == 0x140) src = (lane & ~15) | (15 - li);
    else if (CTRL == 0x142) { src = row * 16 - 1; valid = row > 0; }
    else if (CTRL == 0x143) { src = 31; valid = row >= 2; }
    else if (CTRL == 0x130) { src = lane + 1; valid = lane < 63; }
    const double g = __shfl(v, valid ? src : lane, 64);
    return (valid && ((ROW_MASK >> row) & 1)) ? g : 0.0;
}
#else
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_take(double v) {  // the value of the lane selected by CTRL; 0.0 where no lane is selected / the row is masked
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
#endif
__device__ inline double wave_sum(double v) {
    v += dpp_take<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v += dpp_take<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v += dpp_take<0x141, 0xf>(v);  // row_half_mirror
    v += dpp_take<0x140, 0xf>(v);  // row_mirror: every lane holds its row's sum
    v += dpp_take<0x142, 0xa>(v);  // row_bcast15 into rows 1 and 3
    v += dpp_take<0x143, 0xc>(v);  // row_bcast31 into rows 2 and 3: lane 63 holds the total
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
// |v| <= thr for a 1x1 entry.  The reference tests the Euclidean norm sqrt(v*v); in binary floating point with a correctly
// rounded square root, sqrt(fl(v*v)) == |v| exactly unless v*v under- or overflows (and an underflowing v is far below
// any threshold either way), so this is the same decision bit for bit without the square root.
__device__ inline bool norm1_le(double v, double thr) { return fabs(v) <= thr; }
// The correctly rounded square root is monotone, so { s : sqrt(s) <= thr } is an interval [0, S] of doubles: with S found
// once per launch, the norm test of simplify() needs no root and decides exactly as before.
__device__ inline double sq_threshold(double thr) {
    double S = thr * thr;
    while (sqrt(nextafter(S, INFINITY)) <= thr) S = nextafter(S, INFINITY);
    while (S > 0.0 && sqrt(S) > thr) S = nextafter(S, -INFINITY);
    return S;
}
// ||acc|| <= thr as simplify() tests it (RT/PZsparse.cu:327-335): entries squared and summed in order, then the root
template <int SZ>
__device__ inline bool norm_le(const double* acc, const Wave& w) {
    if constexpr (SZ == 1) return norm1_le(acc[0], w.thr);
    double s = 0.0;
#pragma unroll
    for (int e = 0; e < SZ; e++) s += acc[e] * acc[e];
    return s <= w.thr_sq;
}
// ---- prune margin: the verdicts of simplify() with their distance to the threshold recorded (Wave::mabs) ----
// s = the squared norm the verdict was taken on (1x1: v * v, whose root IS |v|)
__device__ inline void mtrack(Wave& w, double s) {
#ifndef PZW_NO_MARGIN   // (development: the build without the tracking, for its cost -- profiles/r06_prune_margin.txt)
    w.mabs = fmin(w.mabs, fabs(s - w.thr_sq));
#endif
}
__device__ inline bool norm1_le_t(Wave& w, double v) { mtrack(w, v * v); return fabs(v) <= w.thr; }
template <int SZ>
__device__ inline bool norm_le_t(const double* acc, Wave& w) {
    if constexpr (SZ == 1) return norm1_le_t(w, acc[0]);
    double s = 0.0;
#pragma unroll
    for (int e = 0; e < SZ; e++) s += acc[e] * acc[e];
    mtrack(w, s);
    return s <= w.thr_sq;
}
// fold this operator's tracker into the wave's LDS row (every lane its own place: no reduction here)
__device__ inline void mflush(const Wave& w) {
#ifdef PZW_NO_MARGIN
    return;
#endif
    if (w.mg == nullptr) return;
    w.mg[w.lane] = fmin(w.mg[w.lane], w.mabs);
}
// the same for a verdict taken outside the operators (the closed-form JRS of p1_reach.hip): straight into the row
__device__ inline void mtrack_lds(const Wave& w, double s) {
    if (w.mg == nullptr) return;
    w.mg[w.lane] = fmin(w.mg[w.lane], fabs(s - w.thr_sq));
}
template <int SZ>
__device__ inline bool norm_le_lds(const double* acc, const Wave& w) {   // norm_le() + mtrack_lds()
    double s = 0.0;
#pragma unroll
    for (int e = 0; e < SZ; e++) s += acc[e] * acc[e];
    mtrack_lds(w, s);
    return SZ == 1 ? fabs(acc[0]) <= w.thr : s <= w.thr_sq;
}
// a work item's verdicts, reduced over the wave's lanes, into its problem's two words
__device__ inline void margin_reset(const Wave& w) { if (w.mg) w.mg[w.lane] = __builtin_inf(); }
// (m: a lane's smallest |squared norm - thr_sq|) -> the problem's word: (bits of +inf) - bits of the minimum over the lanes, by atomicMax, so that
// a cleared word means "no verdict" (non-negative doubles order as their bits)
__device__ inline void margin_reduce_store(double m, int lane, unsigned long long* dst) {
    if (dst == nullptr) return;
    unsigned long long v = 0x7ff0000000000000ull - (unsigned long long)__double_as_longlong(m);
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) { const unsigned long long v2 = __shfl_xor(v, k, WAVE); v = v2 > v ? v2 : v; }
    if (lane == 0 && v) atomicMax(dst, v);
}
__device__ inline void margin_item_end(const Wave& w, unsigned long long* dst) {
    if (w.mg == nullptr || dst == nullptr) return;
    margin_reduce_store(w.mg[w.lane], w.lane, dst);
}
__device__ inline int next_pow2(int v) { int p = 64; while (p < v) p <<= 1; return p; }

__device__ inline void flag(const Wave& w, int bit) { if (w.lane == 0) w.lstat[ST_ERR] |= bit; }

// 64-lane bitonic sort of (skey, sidx)[0, P) ascending by (key, idx); P is a power of two >= 64.
// (k_first > 2: the blocks of k_first / 2 entries are sorted already, alternately ascending and descending -- only the network's last stages run;
//  k_first = P is the merge of one ascending and one descending half: MulEval::split_merge)
__device__ inline void bitonic_sort(const Wave& w, int P, int k_first = 2) {
    const int half = P >> 1;
    for (int k = k_first; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            // four compare-exchanges per lane in flight: all LDS reads of a group are issued before the first compare
            // (the pairs of one pass are disjoint); the loop is LDS-latency bound otherwise
            for (int t0 = w.lane; t0 < half; t0 += 4 * WAVE) {
                int ii[4], ll[4];
                pzkey_t ka[4], kb[4];
                uint16_t ia[4], ib[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int t = t0 + u * WAVE;
                    const int tc = t < half ? t : t0;  // clamp: the spare slots re-read the first pair and do not write
                    ii[u] = ((tc & ~(j - 1)) << 1) | (tc & (j - 1));
                    ll[u] = ii[u] | j;
                    ka[u] = w.skey[ii[u]]; kb[u] = w.skey[ll[u]];
                    ia[u] = w.sidx[ii[u]]; ib[u] = w.sidx[ll[u]];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const bool gt = (ka[u] > kb[u]) || (ka[u] == kb[u] && ia[u] > ib[u]);
                    const bool up = (ii[u] & k) == 0;
                    if (t0 + u * WAVE < half && gt == up) { w.skey[ii[u]] = kb[u]; w.skey[ll[u]] = ka[u]; w.sidx[ii[u]] = ib[u]; w.sidx[ll[u]] = ia[u]; }
                }
            }
            WSYNC();
        }
    }
}

// number of a[0..n) strictly below t / not above t (a ascending, in LDS)
__device__ inline int lower_bound_lds(const LDS_AS pzkey_t* a, int n, pzkey_t t) {
    int lo = 0, len = n;
    while (len > 0) {
        const int half = len >> 1;
        if (a[lo + half] < t) { lo += half + 1; len -= half + 1; } else len = half;
    }
    return lo;
}
__device__ inline int upper_bound_lds(const LDS_AS pzkey_t* a, int n, pzkey_t t) {
    int lo = 0, len = n;
    while (len > 0) {
        const int half = len >> 1;
        if (a[lo + half] <= t) { lo += half + 1; len -= half + 1; } else len = half;
    }
    return lo;
}

// Merge path (round 4): places [c, c9) of the merge of the two adjacent sorted runs [ps, am) and [am, pe) of (K, V), written to (Ko, Vo) at the same
// places.  One search along the diagonal of the first place -- how many of the places before it the first run fills -- then a two-finger merge,
// one key read per place.  Ties: the first run's key first (first_wins), or the second's.  (No key is PZKEY_MAX: a run that is through never wins.)
template <bool FIRST_WINS>
__device__ inline void merge_span_t(const LDS_AS pzkey_t* K, const LDS_AS uint16_t* V, LDS_AS pzkey_t* Ko, LDS_AS uint16_t* Vo,
                                    int ps, int am, int pe, int c, int c9, int N) {
    const int la = am - ps, lb = pe - am;
    const int d0 = c - ps;
    int lo = max(0, d0 - lb), hi = min(d0, la);
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const pzkey_t ka = K[ps + mid], kq = K[am + d0 - 1 - mid];
        const bool first = FIRST_WINS ? (ka <= kq) : (ka < kq);   // does A[mid] go before B[d0 - 1 - mid]?
        lo = first ? mid + 1 : lo; hi = first ? hi : mid;
    }
    int pa = ps + lo, pb = am + d0 - lo;   // the two fingers
    pzkey_t ka = K[min(pa, N - 1)], kq = K[min(pb, N - 1)];
    ka = pa < am ? ka : PZKEY_MAX; kq = pb < pe ? kq : PZKEY_MAX;
    for (int o = c; o < c9; o++) {
        const bool first = FIRST_WINS ? (ka <= kq) : (ka < kq);
        const int src = first ? pa : pb;
        Ko[o] = first ? ka : kq;
        Vo[o] = V[src];
        const int nsrc = src + 1;   // the run that gave the key moves on
        const pzkey_t nx = K[min(nsrc, N - 1)];
        const pzkey_t nv = nsrc < (first ? am : pe) ? nx : PZKEY_MAX;
        pa = first ? nsrc : pa; pb = first ? pb : nsrc;
        ka = first ? nv : ka; kq = first ? kq : nv;
    }
}
__device__ inline void merge_span(const LDS_AS pzkey_t* K, const LDS_AS uint16_t* V, LDS_AS pzkey_t* Ko, LDS_AS uint16_t* Vo,
                                  int ps, int am, int pe, int c, int c9, bool first_wins, int N) {
    if (first_wins) merge_span_t<true>(K, V, Ko, Vo, ps, am, pe, c, c9, N);   // (wave-uniform)
    else merge_span_t<false>(K, V, Ko, Vo, ps, am, pe, c, c9, N);
}

// Sort step shared by every operator: checks the LDS capacity, then leaves the N raw terms of `ev` ordered by
// (key, generation index) -- sidx[p] = index of the p-th term; the p-th key is skey[p], or ev.key_lds(w, sidx[p]) when
// `indirect` is returned (merge paths that keep the operands' key lists in skey).  Returns the N to process (0 on overflow).
template <class Eval>
__device__ inline int sort_terms(Wave& w, int N, const Eval& ev, bool& indirect) {
    indirect = false;
    // LDS capacity: sidx holds every raw term; skey holds either the operands' key lists (merge paths) or, for the
    // bitonic path, all raw keys padded to a power of two.  On overflow the host retries with larger buffers.
    if (N > w.cap_raw || (N > Eval::kCountMax && !ev.can_merge(w, N) && next_pow2(N) > w.cap_key)) { flag(w, ERR_RAW_OVERFLOW); N = 0; }
    if (w.lane == 0 && N > w.lstat[ST_MAX_RAW]) w.lstat[ST_MAX_RAW] = N;
#ifdef P1_PROFILE
    if (w.lane == 0) { w.prof[PR_CALLS] += 1; w.prof[PR_TERMS] += N; if (N <= 64) w.prof[PR_SMALL] += 1; }
#endif
    if (N > 0) {
        if (N <= Eval::kCountMax) {
            // one term per lane: rank by counting (key, index) pairs that sort before this lane's, N broadcast steps
            // (Eval::kCountMax: up to a full wave for products, whose merge has per-level fixed costs; a handful of terms
            // for sums, whose one-search merge is cheaper than ~200 cycles per broadcast step from there on)
            PROF_T0
            if (w.half == 0) {   // (a pair: the first wave alone)
                const pzkey_t key = w.lane < N ? ev.key(w.lane) : PZKEY_MAX;
                int rank = 0;
                for (int l = 0; l < N; l++) {
                    const pzkey_t kl = pzkey_readlane(key, l);
                    rank += (kl < key || (kl == key && l < w.lane)) ? 1 : 0;
                }
                if (w.lane < N) { w.skey[rank] = key; w.sidx[rank] = (uint16_t)w.lane; }
            }
            psync(w);
            PROF_ADD(PR_SORT) PROF_ADD(PR_S_RANK)
        } else if (ev.try_merge(w, N, indirect)) {
            // the operands' sorted runs were merged: either skey / sidx hold the sorted keys and the permutation like the
            // bitonic path leaves them, or (`indirect`) sidx holds the permutation and the keys come from the LDS staging
#ifdef DBG_CHECK_MERGE
            {
                int bad = 0;
                for (int p = w.lane; p < N; p += WAVE) {
                    const pzkey_t kp = indirect ? ev.key_lds(w, w.sidx[p]) : w.skey[p];
                    if (p > 0 && (indirect ? ev.key_lds(w, w.sidx[p - 1]) : w.skey[p - 1]) > kp) bad = 1;
                    if (p > 0 && (indirect ? ev.key_lds(w, w.sidx[p - 1]) : w.skey[p - 1]) == kp && w.sidx[p - 1] > w.sidx[p]) bad |= 4;
                    if (kp != ev.key(w.sidx[p])) bad |= 2;
                }
                {
                    const unsigned long long any = __ballot(bad != 0);
                    if (any != 0ull && !(w.lstat[ST_ERR] & 64)) {
                        int bits = 0;
                        for (int b = 1; b <= 4; b <<= 1) if (__ballot((bad & b) != 0) != 0ull) bits |= b;
                        if (w.lane == 0) { w.lstat[ST_ERR] |= 64; w.lstat[3] = (N & 0xffff) | (bits << 16) | ((indirect ? 1 : 0) << 20) | (__popcll(any) << 24); }
                    }
                }
            }
#endif
        } else {
            const int P = next_pow2(N);
            if (w.half == 0) {   // (a pair: the first wave alone -- the network is the path of last resort)
            { PROF_T0
            for (int p0 = w.lane; p0 < P; p0 += 4 * WAVE) {   // (four keys per lane and pass in flight: see MulEval::tree_merge)
                pzkey_t kk[4];
#pragma unroll
                for (int u = 0; u < 4; u++) kk[u] = ev.key(min(p0 + u * WAVE, N - 1));
#pragma unroll
                for (int u = 0; u < 4; u++) { const int p = p0 + u * WAVE; if (p < P) { w.skey[p] = p < N ? kk[u] : PZKEY_MAX; w.sidx[p] = (uint16_t)p; } }
            }
            WSYNC();
            PROF_ADD(PR_FILL) }
            { PROF_T0
            bitonic_sort(w, P);
            PROF_ADD(PR_SORT) PROF_ADD(PR_S_BITONIC) }
            }
            psync(w);
        }
    }
    return N;
}

// ---- a pair of waves on one reduce pass (Wave::nl == 2 * WAVE) ------------------------------------------------------------------------
// The sorted terms are reduced in chunks of 64 (one per lane); a chunk never depends on another one's registers -- a run of equal keys that
// crosses a chunk's end is finished by its head lane from memory, and the lanes of the next chunk that continue it do nothing.  So the first
// wave takes the lower half of the chunks and the second the upper half, at the same time.  What the single wave carries from chunk to chunk
// is the number of monomials emitted so far: the second wave starts at the number of RUN HEADS below its first chunk (an upper bound of what
// the first wave will emit: counted from the keys alone), and once both are through it moves its rows down behind the first wave's
// (pair_finish).  And the pruned amounts: a reduce pass -- on one wave or two -- adds those of the chunks below reduce_mid(N) and those of
// the chunks from there on as TWO sums per lane, reduces each over the lanes and adds the two results; which wave works on a chunk does
// not enter.  So every bit of a result -- keys, coefficients, centres and radii -- is the same whether one wave or a pair produced it.
// (Round 4 changed this order for the single wave too, from one running sum: the per-step kernel's radii moved by a few 1e-16 against
// round 3's, inside the 1e-11 its parity with the CPU oracle is stated to.)
__device__ inline int reduce_mid(int N) { const int C = (N + WAVE - 1) / WAVE; return min(((C + 1) / 2) * WAVE, N); }
template <class KeyAt>
__device__ inline void pair_split(const Wave& w, int N, int cap, const KeyAt& keyat, int& b0, int& b1, int& pos0) {
    const int mid = reduce_mid(N);
    // run heads below and above the split (both waves count both: the keys are in LDS).  The second wave's rows start at `lo`, so lo + hi
    // rows must fit the slot; a result that close to the slot's capacity is left to the first wave alone (as a wave on its own would do it).
    int lo = 0, hi = 0;
    for (int base = 0; base < N; base += WAVE) {
        const int p = base + w.lane;
        const bool head = p < N && ((p == 0) || (keyat(p - 1) != keyat(p)));
        const int n = __popcll(__ballot(head));
        if (base < mid) lo += n; else hi += n;
    }
    const bool split = lo + hi <= cap;
    if (w.half == 0) { b0 = 0; b1 = split ? mid : N; pos0 = 0; }
    else { b0 = split ? mid : N; b1 = N; pos0 = split ? lo : 0; }
}
__device__ inline LDS_AS double* pair_doubles(const Wave& w) {
    LDS_AS int* q = w.pair + PW_DBL;
    if (((unsigned)(size_t)q & 4u) != 0u) q += 1;   // (8-byte aligned)
    return (LDS_AS double*)q;
}
// In: this wave's rows are out[pos0 .. emitted); rb[NR][SZ] = its sums (over its lanes) of the pruned amounts of the upper chunks.  Out, on
// both waves: emitted = the result's monomial count; on the first wave rb = its own + the second's (one of the two is 0: a chunk is one
// wave's); the second wave's rows sit behind the first's.  (The caller's closing psync() makes the moved rows and the header words visible.)
template <int SZ, int NR>
__device__ inline void pair_finish(const Wave& w, const PZ& out, int& emitted, int pos0, double (*rb)[SZ]) {
    static_assert(NR * SZ <= PW_DBL_PER_HALF, "pair exchange area");
    LDS_AS double* xd = pair_doubles(w);
    if (w.lane == 0) {
        w.pair[PW_E0 + w.half] = emitted - pos0;
        if (w.half == 1) {
            w.pair[PW_E1 + 1] = pos0;
#pragma unroll
            for (int r = 0; r < NR; r++)
#pragma unroll
                for (int e = 0; e < SZ; e++) xd[r * SZ + e] = rb[r][e];
        }
    }
    psync(w);
    const int e0 = uni_i(w.pair[PW_E0]), e1 = uni_i(w.pair[PW_E1]), p1 = uni_i(w.pair[PW_E1 + 1]);
    if (w.half == 0) {
#pragma unroll
        for (int r = 0; r < NR; r++)
#pragma unroll
            for (int e = 0; e < SZ; e++) rb[r][e] = rb[r][e] + xd[r * SZ + e];
    }
    if (w.half == 1 && e1 > 0 && p1 != e0) {   // (the first wave pruned something: p1 > e0)
        for (int r0 = 0; r0 < e1; r0 += 2 * WAVE) {
            pzkey_t kk[2];
            double cc[2][SZ];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int r = min(r0 + u * WAVE + w.lane, e1 - 1), src = p1 + r;
                kk[u] = out.keys[src];
#pragma unroll
                for (int e = 0; e < SZ; e++) cc[u][e] = out.coef[(size_t)src * SZ + e];
            }
            WSYNC();   // every row of this step is in registers before one of them is overwritten (the rows move DOWN: a step never writes a later step's sources)
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int r = r0 + u * WAVE + w.lane;
                if (r < e1) {
                    out.keys[e0 + r] = kk[u];
#pragma unroll
                    for (int e = 0; e < SZ; e++) out.coef[(size_t)(e0 + r) * SZ + e] = cc[u][e];
                }
            }
        }
    }
    emitted = e0 + e1;
}

// Sort the N raw terms described by `ev`, sum equal keys, prune small coefficients into the independent part
// and write the result to `out` (RT/PZsparse.cu:284-350).  base_ind / base_ind2 = independent parts before pruning.
// Eval: pzkey_t key(int idx) const; void coef(int idx, double* c /*[SZ]*/) const.
template <int SZ, class Eval>
__device__ inline void sort_reduce_emit(Wave& w, int N, const Eval& ev, const PZ& out, const double* base_ind, const double* base_ind2) {
    int emitted = 0;
    bool any_pruned = false, indirect = false;
    double ra[SZ], rb[SZ];   // pruned amounts of the chunks below / from reduce_mid(N) on
#pragma unroll
    for (int e = 0; e < SZ; e++) { ra[e] = 0.0; rb[e] = 0.0; }
    N = uni_i(sort_terms(w, N, ev, indirect));
    const bool paired = w.nl != WAVE;
    int pos0 = 0;
    if (N > 0) {
        PROF_T0
        int b0 = 0, b1 = N;
        const int mid = reduce_mid(N);
        if (paired) {
            if (indirect) pair_split(w, N, out.cap, [&](int p) { return ev.key_lds(w, w.sidx[p]); }, b0, b1, pos0);
            else pair_split(w, N, out.cap, [&](int p) { return w.skey[p]; }, b0, b1, pos0);
            b0 = uni_i(b0); b1 = uni_i(b1); pos0 = uni_i(pos0);
            emitted = pos0;
        }
        for (int base = b0; base < b1; base += WAVE) {
            const int p = base + w.lane;
            bool head = false, keep = false;
            pzkey_t key = 0;
            double acc[SZ];
#ifdef P1_PROFILE
            long long e_t = clock64();
#define E_LAP(slot) { const long long n__ = clock64(); if (w.lane == 0) w.prof[slot] += (unsigned long long)(n__ - e_t); e_t = n__; }
#else
#define E_LAP(slot)
#endif
            if (p < N) {
                key = indirect ? ev.key_lds(w, w.sidx[p]) : w.skey[p];
                head = (p == 0) || ((indirect ? ev.key_lds(w, w.sidx[p - 1]) : w.skey[p - 1]) != key);
            }
            E_LAP(PR_E_HEAD)
            // Every lane evaluates its own term (one batch of loads in flight for the whole wave); the head lane of a run of
            // equal keys then adds the members in generation order.  Member j of every run sits j lanes up: the terms are
            // shifted down the wave one lane per round (a DPP move, no memory traffic), and only the tail of a run that
            // crosses the end of this 64-term chunk is loaded directly.  (Before: each head walked its run with one dependent
            // round trip to memory per member -- 37 % of the chain's operator time was this pass.)  Same sums, same order.
#pragma unroll
            for (int e = 0; e < SZ; e++) acc[e] = 0.0;
            if (p < N) ev.coef(BIDX(w, (int)w.sidx[p], N, 5), acc);
            E_LAP(PR_E_COEF)
            {
                double sh[SZ];
#pragma unroll
                for (int e = 0; e < SZ; e++) sh[e] = acc[e];
                for (int j = 1;; j++) {
                    const int q = p + j;
                    const bool more = head && q < N && (indirect ? ev.key_lds(w, w.sidx[q]) : w.skey[q]) == key;
                    if (__ballot(more) == 0ull) break;
#pragma unroll
                    for (int e = 0; e < SZ; e++) sh[e] = dpp_take<0x130, 0xf>(sh[e]);  // wave_shl:1 -- lane l now holds the term of lane l + j
                    if (more) {
                        double c[SZ];
#pragma unroll
                        for (int e = 0; e < SZ; e++) c[e] = sh[e];
                        if (w.lane + j >= WAVE) ev.coef(w.sidx[q], c);
#pragma unroll
                        for (int e = 0; e < SZ; e++) acc[e] += c[e];
                    }
                }
            }
            E_LAP(PR_E_RUN)
            if (head) {
                keep = !norm_le_t<SZ>(acc, w);
                if (!keep) {
                    const bool upper = base >= mid;   // (wave-uniform; + 0.0 leaves a sum of absolute values as it is)
#pragma unroll
                    for (int e = 0; e < SZ; e++) { const double v = fabs(acc[e]); ra[e] += upper ? 0.0 : v; rb[e] += upper ? v : 0.0; }
                }
            }
            E_LAP(PR_E_PRUNE)
            const unsigned long long m = __ballot(keep);
            any_pruned = any_pruned || (__ballot(head && !keep) != 0ull);
            if (keep) {
                const int pos = emitted + __popcll(m & ((1ull << w.lane) - 1ull));
                if (pos < out.cap) {
                    out.keys[pos] = key;
#pragma unroll
                    for (int e = 0; e < SZ; e++) out.coef[(size_t)pos * SZ + e] = acc[e];
                }
            }
            emitted += __popcll(m);
            E_LAP(PR_E_STORE)
        }
        PROF_ADD(PR_EMIT)
    }
    if (any_pruned) {
#pragma unroll
        for (int e = 0; e < SZ; e++) { ra[e] = wave_sum(ra[e]); rb[e] = wave_sum(rb[e]); }
    }
    if (paired && N > 0) pair_finish<SZ, 1>(w, out, emitted, pos0, (double (*)[SZ])rb);
    if (emitted > out.cap) { flag(w, ERR_SLOT_OVERFLOW); emitted = out.cap; }
    if (w.lane2 == 0) {
#pragma unroll
        for (int e = 0; e < SZ; e++) { const double r = ra[e] + rb[e]; out.ind[e] = base_ind[e] + r; out.ind2[e] = base_ind2[e] + r; }
        w.cnt[out.id] = emitted;
        if (emitted > w.lstat[ST_MAX_OUT]) w.lstat[ST_MAX_OUT] = emitted;
    }
    mflush(w);
    psync(w);
}
#undef E_LAP

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
    static constexpr int kCountMax = 8;
    Seg s[NS];
    int off[NS + 1];
    __device__ inline int seg_of(int idx) const {
        int k = 0;
#pragma unroll
        for (int i = 1; i < NS; i++) k += (idx >= off[i]) ? 1 : 0;
        return k;
    }
    // (a source picked by a per-lane index is picked with selects: `s[k]` with a run-time k puts the whole array into scratch
    //  memory, and every access becomes a memory round trip)
    __device__ inline pzkey_t key(int idx) const {
        const int k = seg_of(idx);
        const GLB_AS pzkey_t* keys = s[0].v.keys;
        int first = off[0];
#pragma unroll
        for (int q = 1; q < NS; q++) { const bool me = (k == q); keys = me ? s[q].v.keys : keys; first = me ? off[q] : first; }
        return keys[idx - first];
    }
    template <int K>
    __device__ inline void coef_of(int idx, double* c) const {   // idx is a term of source K
        const Seg& g = s[K];
        const GLB_AS double* src = g.v.coef + (size_t)(idx - off[K]) * g.v.stride + g.v.off;
        if (g.comp < 0) {
#pragma unroll
            for (int e = 0; e < SZ; e++) c[e] = g.scale * src[e];
        } else {
            const double v = g.scale * src[0];
#pragma unroll
            for (int e = 0; e < SZ; e++) c[e] = (e == g.comp) ? v : 0.0;
        }
    }
    // every source is a simplified PZ (sorted, unique keys): merge the NS runs by ranking each term with binary
    // searches in the other runs; ties across runs go to the earlier run (= generation order)
    __device__ inline bool can_merge(const Wave& w, int N) const { return N <= w.cap_key; }
    __device__ inline bool try_merge(Wave& w, int N, bool& indirect) const {
        if (!can_merge(w, N)) return false;
        indirect = true;
#ifdef DBG_NO_MERGE_LIN
        return false;
#endif
        PROF_T0
        const int nl = w.nl;   // (cooperating lanes: one wave's, or a pair's)
#ifndef LIN_MERGE_BY_SEARCH
        // Merge path (round 4): the NS sorted runs are merged pairwise in ceil(log2 NS) levels between the two halves of the sort buffers, as
        // a product's runs are (MulEval::tree_merge) -- a lane fills consecutive places of a level with one diagonal search and a two-finger
        // merge -- instead of NS - 1 binary searches for every term (-DLIN_MERGE_BY_SEARCH: that form, also taken when the buffers cannot
        // hold two copies).  Ties go to the earlier run, which is generation order: the same order, and the keys come out sorted in skey.
        if (2 * N <= w.cap_key && 2 * N <= w.cap_raw) {
            constexpr int LV = NS <= 2 ? 1 : 2;   // (NS <= 4)
            static_assert(NS <= 4, "merge levels");
            indirect = false;
            int cur = LV & 1;   // the last level writes buffer 0 = (skey, sidx) as the reduce pass expects them
            LDS_AS pzkey_t* kb[2] = {w.skey, w.skey + N};
            LDS_AS uint16_t* vb[2] = {w.sidx, w.sidx + N};
            for (int i0 = w.lane2; i0 < N; i0 += 4 * nl) {   // (four keys per lane and pass in flight)
                pzkey_t kk[4];
#pragma unroll
                for (int u = 0; u < 4; u++) kk[u] = key(min(i0 + u * nl, N - 1));
#pragma unroll
                for (int u = 0; u < 4; u++) { const int idx = i0 + u * nl; if (idx < N) { kb[cur][idx] = kk[u]; vb[cur][idx] = (uint16_t)idx; } }
            }
            psync(w);
            const int E = ((N + nl - 1) / nl) | 1;
#pragma unroll
            for (int lv = 0; lv < LV; lv++) {
                // level 0: runs (0, 1) and (2, 3); level 1: (01, 23).  With two runs: one level, one pair.
                int bnd[5];   // pair p of this level: [bnd[2p], bnd[2p+1]) + [bnd[2p+1], bnd[2p+2])
                if (NS <= 2 || lv == 1) { bnd[0] = 0; bnd[1] = NS <= 2 ? off[1] : off[2 < NS ? 2 : NS]; bnd[2] = N; bnd[3] = N; bnd[4] = N; }
                else { bnd[0] = 0; bnd[1] = off[1]; bnd[2] = off[2]; bnd[3] = NS >= 4 ? off[3] : N; bnd[4] = N; }
                const LDS_AS pzkey_t* K = kb[cur];
                const LDS_AS uint16_t* V = vb[cur];
                int c = w.lane2 * E;
                const int cend = min(c + E, N);
                while (c < cend) {
                    const bool second = c >= bnd[2];
                    const int ps = second ? bnd[2] : bnd[0], am = second ? bnd[3] : bnd[1], pe = second ? bnd[4] : bnd[2];
                    const int c9 = min(cend, pe);
                    merge_span(K, V, kb[cur ^ 1], vb[cur ^ 1], ps, am, pe, c, c9, true, N);
                    c = c9;
                }
                psync(w);
                cur ^= 1;
            }
            PROF_ADD(PR_SORT) PROF_ADD(PR_S_LINMERGE)
            return true;
        }
#endif
        for (int i0 = w.lane2; i0 < N; i0 += 4 * nl) {   // (four keys per lane and pass in flight: see MulEval::tree_merge)
            pzkey_t kk[4];
#pragma unroll
            for (int u = 0; u < 4; u++) kk[u] = key(min(i0 + u * nl, N - 1));
#pragma unroll
            for (int u = 0; u < 4; u++) if (i0 + u * nl < N) w.skey[i0 + u * nl] = kk[u];
        }
        psync(w);
        for (int idx = w.lane2; idx < N; idx += nl) {
            const int k = seg_of(idx);
            const pzkey_t ky = w.skey[idx];
            int first = off[0];
#pragma unroll
            for (int q = 1; q < NS; q++) first = (k == q) ? off[q] : first;
            int rank = idx - first;
#pragma unroll
            for (int k2 = 0; k2 < NS; k2++) {
                if (k2 == k) continue;
                const LDS_AS pzkey_t* run = w.skey + off[k2];
                const int len = off[k2 + 1] - off[k2];
                rank += (k2 < k) ? upper_bound_lds(run, len, ky) : lower_bound_lds(run, len, ky);
            }
            w.sidx[rank] = (uint16_t)idx;
        }
        psync(w);
        PROF_ADD(PR_SORT) PROF_ADD(PR_S_LINMERGE)
        return true;
    }
    __device__ inline pzkey_t key_lds(const Wave& w, int idx) const { return w.skey[idx]; }
    __device__ inline void coef(int idx, double* c) const {
        const int k = seg_of(idx);
        const GLB_AS double* cf = s[0].v.coef;
        int first = off[0], stride = s[0].v.stride, voff = s[0].v.off, comp = s[0].comp;
        double scale = s[0].scale;
#pragma unroll
        for (int q = 1; q < NS; q++) {
            const bool me = (k == q);
            cf = me ? s[q].v.coef : cf; first = me ? off[q] : first; stride = me ? s[q].v.stride : stride; voff = me ? s[q].v.off : voff;
            comp = me ? s[q].comp : comp; scale = me ? s[q].scale : scale;
        }
        const GLB_AS double* src = cf + (size_t)(idx - first) * stride + voff;
        if (comp < 0) {
#pragma unroll
            for (int e = 0; e < SZ; e++) c[e] = scale * src[e];
        } else {
            const double v = scale * src[0];
#pragma unroll
            for (int e = 0; e < SZ; e++) c[e] = (e == comp) ? v : 0.0;
        }
    }
};

template <int SZ, int NS>
__device__ inline void coef_static(const LinEval<SZ, NS>& ev, int k, int idx, double* c) {
    if constexpr (NS >= 1) { if (k == 0) { ev.template coef_of<0>(idx, c); return; } }
    if constexpr (NS >= 2) { if (k == 1) { ev.template coef_of<1>(idx, c); return; } }
    if constexpr (NS >= 3) { if (k == 2) { ev.template coef_of<2>(idx, c); return; } }
    if constexpr (NS >= 4) { if (k == 3) { ev.template coef_of<3>(idx, c); return; } }
}

template <int SZ, int NS>
__device__ PZW_NOINLINE void lincomb(Wave& w_, const PZ& out_, const Seg* segs) {
    PZ_KEEP_RETURN_ADDRESS();
    PZW_WAVE_LOCAL(w, w_)
    const PZ out = uni_pz(out_);
    PROF_CALL_T0
    LinEval<SZ, NS> ev;
    int N = 0;
    double cen[SZ], ind[SZ], ind2[SZ];
#pragma unroll
    for (int e = 0; e < SZ; e++) { cen[e] = 0.0; ind[e] = 0.0; ind2[e] = 0.0; }
#pragma unroll
    for (int k = 0; k < NS; k++) {
        ev.s[k] = segs[k];
        ev.s[k].v = uni_view(segs[k].v);
        ev.off[k] = N;
        N += ev.s[k].v.cnt;
        const View& v = ev.s[k].v;
        const double sc = segs[k].scale, asc = fabs(sc);
        if (segs[k].comp < 0) {
#pragma unroll
            for (int e = 0; e < SZ; e++) {
                const double c = sc * v.cen[v.off + e], i = v.ind[v.off + e] * asc, i2 = v.ind2[v.off + e] * asc;
                cen[e] = (k == 0) ? c : cen[e] + c;
                ind[e] = (k == 0) ? i : ind[e] + i;
                ind2[e] = (k == 0) ? i2 : ind2[e] + i2;
            }
        } else {
            const double c = sc * v.cen[v.off], i = v.ind[v.off] * asc, i2 = v.ind2[v.off] * asc;
#pragma unroll
            for (int e = 0; e < SZ; e++)
                if (e == segs[k].comp) { cen[e] = cen[e] + c; ind[e] = ind[e] + i; ind2[e] = ind2[e] + i2; }
        }
    }
    ev.off[NS] = N;
    psync(w);  // all lanes have read the sources' centre / indep before `out` (possibly aliasing) is written
    if (w.lane2 == 0) {
#pragma unroll
        for (int e = 0; e < SZ; e++) out.cen[e] = cen[e];
    }
    sort_reduce_emit<SZ>(w, N, ev, out, ind, ind2);
    PROF_CALL_END(N)
}

// Chain of sums ((s0*x0 + s1*x1) + s2*x2) + ... where the reference calls simplify() after every `+`
// (e.g. linear_acc + cross(..) + cross(..), RT/Dynamics.cu:107-109): ONE sort over all sources, and the head lane of
// each run of equal keys replays the stages -- after source k (k >= 1) has been added, the term accumulated so far is
// pruned into that stage's radius if its norm is <= threshold and takes no part in the later stages, exactly as the
// intermediate PZ would have lost it.  Centre and radii are accumulated in the composed order.  NS = 2 is lincomb.
template <int SZ, int NS>
__device__ PZW_NOINLINE void lincomb_chain(Wave& w_, const PZ& out_, const Seg* segs) {
    PZ_KEEP_RETURN_ADDRESS();
    PZW_WAVE_LOCAL(w, w_)
    const PZ out = uni_pz(out_);
    PROF_CALL_T0
    LinEval<SZ, NS> ev;
    int N = 0;
    double cen[SZ], indk[NS][SZ], ind2k[NS][SZ];  // indk[k]: what source k adds to the radius at its stage
#pragma unroll
    for (int e = 0; e < SZ; e++) cen[e] = 0.0;
#pragma unroll
    for (int k = 0; k < NS; k++) {
        ev.s[k] = segs[k];
        ev.s[k].v = uni_view(segs[k].v);
        ev.off[k] = N;
        N += ev.s[k].v.cnt;
        const View& v = ev.s[k].v;
        const double sc = segs[k].scale, asc = fabs(sc);
#pragma unroll
        for (int e = 0; e < SZ; e++) { indk[k][e] = 0.0; ind2k[k][e] = 0.0; }
        if (segs[k].comp < 0) {
#pragma unroll
            for (int e = 0; e < SZ; e++) {
                const double c = sc * v.cen[v.off + e];
                cen[e] = (k == 0) ? c : cen[e] + c;
                indk[k][e] = v.ind[v.off + e] * asc; ind2k[k][e] = v.ind2[v.off + e] * asc;
            }
        } else {
            const double c = sc * v.cen[v.off];
#pragma unroll
            for (int e = 0; e < SZ; e++)
                if (e == segs[k].comp) { cen[e] = cen[e] + c; indk[k][e] = v.ind[v.off] * asc; ind2k[k][e] = v.ind2[v.off] * asc; }
        }
    }
    ev.off[NS] = N;
    [[maybe_unused]] const int N_in = N;
    psync(w);  // all lanes have read the sources' centre / indep before `out` (possibly aliasing) is written
    if (w.lane2 == 0) {
#pragma unroll
        for (int e = 0; e < SZ; e++) out.cen[e] = cen[e];
    }
    int emitted = 0;
    bool any_pruned = false, indirect = false;
    double ra[NS][SZ], rb[NS][SZ];  // ra[k] / rb[k]: pruned at stage k (k >= 1) in the chunks below / from reduce_mid(N) on
#pragma unroll
    for (int k = 0; k < NS; k++)
#pragma unroll
        for (int e = 0; e < SZ; e++) { ra[k][e] = 0.0; rb[k][e] = 0.0; }
    N = uni_i(sort_terms(w, N, ev, indirect));
    const bool paired = w.nl != WAVE;   // (two waves on this pass: see pair_split)
    int pos0 = 0;
    if (N > 0) {
        PROF_T0
        int b0 = 0, b1 = N;
        const int mid = reduce_mid(N);
        if (paired) {
            if (indirect) pair_split(w, N, out.cap, [&](int p) { return ev.key_lds(w, w.sidx[p]); }, b0, b1, pos0);
            else pair_split(w, N, out.cap, [&](int p) { return w.skey[p]; }, b0, b1, pos0);
            b0 = uni_i(b0); b1 = uni_i(b1); pos0 = uni_i(pos0);
            emitted = pos0;
        }
        for (int base = b0; base < b1; base += WAVE) {
            const int p = base + w.lane;
            bool head = false, keep = false, pruned = false;
            pzkey_t key = 0;
            double acc[SZ];
#pragma unroll
            for (int e = 0; e < SZ; e++) acc[e] = 0.0;
            if (p < N) {
                key = indirect ? ev.key_lds(w, w.sidx[p]) : w.skey[p];
                head = (p == 0) || ((indirect ? ev.key_lds(w, w.sidx[p - 1]) : w.skey[p - 1]) != key);
            }
            if (head) {
                int q = p;
                bool present = false;
#pragma unroll
                for (int k = 0; k < NS; k++) {
                    // the sources hold unique keys, so a run has at most one term per source, in source order
                    if (q < N && (indirect ? ev.key_lds(w, w.sidx[q]) : w.skey[q]) == key && ev.seg_of(w.sidx[q]) == k) {
                        double c[SZ];
                        coef_static<SZ, NS>(ev, k, w.sidx[q], c);   // (k is a constant once the loop is unrolled)
#pragma unroll
                        for (int e = 0; e < SZ; e++) acc[e] = present ? acc[e] + c[e] : c[e];
                        present = true;
                        q++;
                    }
                    if (k >= 1 && present) {  // simplify() of stage k
                        if (norm_le_t<SZ>(acc, w)) {
                            const bool upper = base >= mid;
#pragma unroll
                            for (int e = 0; e < SZ; e++) { const double v = fabs(acc[e]); ra[k][e] += upper ? 0.0 : v; rb[k][e] += upper ? v : 0.0; }
                            present = false;
                            pruned = true;
                        }
                    }
                }
                keep = present;
            }
            const unsigned long long m = __ballot(keep);
            any_pruned = any_pruned || (__ballot(pruned) != 0ull);
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
        PROF_ADD(PR_EMIT)
    }
    if (any_pruned) {
#pragma unroll
        for (int k = 1; k < NS; k++)
#pragma unroll
            for (int e = 0; e < SZ; e++) { ra[k][e] = wave_sum(ra[k][e]); rb[k][e] = wave_sum(rb[k][e]); }
    }
    // (pair_split has handed the upper chunks to the second wave by now: an instantiation whose partial radii do not fit the exchange area
    //  must not compile, it would publish the first wave's half alone)
    static_assert((NS - 1) * SZ <= PW_DBL_PER_HALF, "lincomb_chain: the pair's exchange area does not hold this instantiation's partial radii");
    if (paired && N > 0) pair_finish<SZ, NS - 1>(w, out, emitted, pos0, &rb[1]);
#pragma unroll
    for (int k = 1; k < NS; k++)
#pragma unroll
        for (int e = 0; e < SZ; e++) ra[k][e] = ra[k][e] + rb[k][e];
    if (emitted > out.cap) { flag(w, ERR_SLOT_OVERFLOW); emitted = out.cap; }
    if (w.lane2 == 0) {
#pragma unroll
        for (int e = 0; e < SZ; e++) {
            // stage 1: (i0 + i1) + pruned; stage k: (previous * 1.0 + ik) + pruned  -- as lincomb over two sources each time
            double r = indk[0][e], r2 = ind2k[0][e];
#pragma unroll
            for (int k = 1; k < NS; k++) { r = (r + indk[k][e]) + ra[k][e]; r2 = (r2 + ind2k[k][e]) + ra[k][e]; }
            out.ind[e] = r; out.ind2[e] = r2;
        }
        w.cnt[out.id] = emitted;
        if (emitted > w.lstat[ST_MAX_OUT]) w.lstat[ST_MAX_OUT] = emitted;
    }
    mflush(w);
    psync(w);
    PROF_CALL_END(N_in)
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

// floor(2^32 / d) + 1 for 1 <= d < 2^16, through one double division (a 64-bit integer division is ~100 instructions): the
// quotient is either an integer (d a power of two) or at least 2^-16 away from one, far more than the rounding error
__device__ inline unsigned long long magic_u32(int d) { return (unsigned long long)(4294967296.0 / (double)d) + 1ull; }


template <class SH>
struct MulEval {
    static constexpr int kCountMax = WAVE;
    View a, b;
    int mb1;
    unsigned long long mb1_magic;  // floor(2^32 / mb1) + 1: t / mb1 == (t * magic) >> 32 while t * mb1 < 2^32 (set_b())
    __device__ inline void set_b(const View& bv) { b = bv; mb1 = bv.cnt + 1; mb1_magic = magic_u32(mb1); }
    __device__ inline void split(int idx, int& i, int& j) const {
        const int t = idx + 1;  // (0,0) = centre*centre is not a monomial
        i = (int)(((unsigned long long)t * mb1_magic) >> 32);  // a division by a run-time value is ~30 instructions, per term
        j = t - i * mb1;
    }
    __device__ inline pzkey_t key(int idx) const {
        int i, j;
        split(idx, i, j);
        return (i ? a.keys[i - 1] : 0ull) + (j ? b.keys[j - 1] : 0ull);  // plain u64 add (RT/PZsparse.cu:938-940)
    }
    // When one operand has at most 7 monomials the raw terms are (its count + 1) sorted runs over the other operand's
    // keys [0, k_1, k_2, ...]: rank each term by binary searches with the target shifted by the run's own key.
    // LDS staging: skey[0 .. nl] = long operand's keys (0 for the centre), skey[nl+1 ..] = short operand's keys.
    static constexpr int MAX_RUNS = 8;
    __device__ inline bool can_rank(const Wave& w) const {
        const bool a_short = a.cnt <= b.cnt;
        const int ns = (a_short ? a.cnt : b.cnt) + 1, nl = (a_short ? b.cnt : a.cnt) + 1;
        return ns <= MAX_RUNS && nl + ns <= w.cap_key;
    }
    __device__ inline bool can_tree(const Wave& w, int N) const { return 2 * N <= w.cap_key && 2 * N <= w.cap_raw; }
    __device__ inline bool can_merge(const Wave& w, int N) const { return can_tree(w, N) || can_rank(w); }
    // The split of the runs at the tree's last level: runs [0, r0) and [r0, nr), r0 = the largest power of two below nr.  (round 6)
    struct Split { int levels, q0, N1, N2, P; bool ok; };
    __device__ inline Split split_of(const Wave& w, int N) const {
        Split sp;
        const int na1 = a.cnt + 1;
        const bool by_a = na1 <= mb1;
        const int nr = by_a ? na1 : mb1, d1 = by_a ? mb1 : na1;
        sp.levels = 0;
        while ((1 << sp.levels) < nr) sp.levels++;
        sp.P = next_pow2(N);
        const int C = min(w.cap_key, w.cap_raw);
        sp.q0 = sp.levels >= 1 ? (d1 << (sp.levels - 1)) - 1 : 0;
        sp.N1 = sp.q0; sp.N2 = N - sp.q0;
        sp.ok = sp.levels >= 1 && sp.N2 > 0 && sp.P <= C && sp.N1 <= sp.P / 2 && sp.N2 <= sp.P / 2 && 2 * sp.N1 <= C && sp.N1 + 2 * sp.N2 <= sp.P;
        return sp;
    }
    __device__ inline bool try_merge(Wave& w, int N, bool& indirect) const {
#ifdef DBG_NO_MERGE_MUL
        return false;
#endif
#ifndef DBG_NO_TREE
        if (can_tree(w, N)) { indirect = false; tree_merge(w, N); return true; }
#endif
#ifdef DBG_NO_RANK
        return false;
#endif
        if (!can_rank(w)) {   // too long for the tree's two buffers and too many runs to rank
#ifndef DBG_NO_SPLIT
            const Split sp = split_of(w, N);
            if (sp.ok) { indirect = false; split_merge(w, N, sp); return true; }
#endif
            return false;     // ... and no split that fits: the bitonic network
        }
        indirect = true;
        return rank_merge(w, N);
    }
    // The raw terms in generation order ARE a concatenation of sorted runs: for a fixed term i of a (centre first), the
    // keys KA[i] + KB[j] rise with j.  Merge those runs pairwise, level by level, between the two halves of the sort
    // buffers: an element's place in the merged pair is its offset in its own run plus the number of the sibling run's
    // elements that go before it -- ONE binary search per element and level (ties: the earlier run first, which is
    // generation order), log2(#runs) levels.  Against the bitonic network this replaces (72 % of a 40 x 40 cross product,
    // tools/dev/gpu_pzop_cost.py) that is ~50 search steps per element instead of 66 compare-exchange stages, and against the
    // one-search-per-run ranking below it is log2 instead of linear in the number of runs.  Eight elements per lane are
    // searched together, branch-free with a wave-uniform step count, so that their LDS reads overlap.
    // The runs are taken along the SHORTER operand (fewer runs = fewer levels): a's terms when a is the shorter one -- the
    // generation order itself -- and otherwise b's terms, with the raw terms laid out b-major for the merge and their
    // generation index carried as the payload.  Equal keys must come out in generation order (a-major): between two runs
    // of a's terms that is the earlier run first; between two runs of b's terms it is the LATER run first (equal sums,
    // so the larger b-term pairs with the smaller a-term).
    __device__ inline void tree_merge(Wave& w, int N) const {
        PROF_T0
        const int na1 = a.cnt + 1;
        const bool by_a = na1 <= mb1;
        const int nr = by_a ? na1 : mb1, d1 = by_a ? mb1 : na1;  // number of runs, terms per run (run 0 lacks the centre x centre term)
        int levels = 0;
        while ((1 << levels) < nr) levels++;
        int cur = levels & 1;  // the last level writes buffer 0 = (skey, sidx) as the reduce pass expects them
        LDS_AS pzkey_t* kb[2] = {w.skey, w.skey + N};
        LDS_AS uint16_t* vb[2] = {w.sidx, w.sidx + N};
        // t / d1 by multiplication: exact while t * d1 < 2^32 (both are below 2^13 here)
        const unsigned long long magic = magic_u32(d1);
        // (the raw keys come from the operands' key lists in global memory: four terms per lane and pass, all eight loads issued before the
        //  first sum -- one term per pass made every pass a load latency of its own, ~20 of them for a 1500-term product; round 4)
        constexpr int KF = 4;
        const int nl = w.nl;   // (cooperating lanes: one wave's, or a pair's -- every pass below is a loop over independent positions)
        if (by_a) {
            for (int i0 = w.lane2; i0 < N; i0 += KF * nl) {
                pzkey_t kk[KF];
#pragma unroll
                for (int u = 0; u < KF; u++) kk[u] = key(min(i0 + u * nl, N - 1));
#pragma unroll
                for (int u = 0; u < KF; u++) { const int idx = i0 + u * nl; if (idx < N) { kb[cur][idx] = kk[u]; vb[cur][idx] = (uint16_t)idx; } }
            }
        } else {
            for (int q0 = w.lane2; q0 < N; q0 += KF * nl) {  // position q of the b-major layout holds the pair (i, j): q + 1 = j * na1 + i
                pzkey_t kk[KF];
                int gi[KF];
#pragma unroll
                for (int u = 0; u < KF; u++) {
                    const int q = min(q0 + u * nl, N - 1);
                    const int j = (int)(((unsigned long long)(q + 1) * magic) >> 32), i = q + 1 - j * na1;
                    kk[u] = (i ? a.keys[i - 1] : 0ull) + (j ? b.keys[j - 1] : 0ull);
                    gi[u] = i * mb1 + j - 1;
                }
#pragma unroll
                for (int u = 0; u < KF; u++) { const int q = q0 + u * nl; if (q < N) { kb[cur][q] = kk[u]; vb[cur][q] = (uint16_t)gi[u]; } }
            }
        }
        psync(w);
        [[maybe_unused]] constexpr int U = 8;
        for (int lv = 0; lv < levels; lv++) {
            const int rl = d1 << lv;  // run r of this level holds the positions q with q + 1 in [r*rl, (r+1)*rl)
            [[maybe_unused]] int top = 1;
            while (top * 2 <= rl) top *= 2;
            const LDS_AS pzkey_t* K = kb[cur];
            const LDS_AS uint16_t* V = vb[cur];
            LDS_AS pzkey_t* Ko = kb[cur ^ 1];
            LDS_AS uint16_t* Vo = vb[cur ^ 1];
#ifndef TREE_MERGE_BY_SEARCH
            // Merge path (round 4; -DTREE_MERGE_BY_SEARCH builds the rounds 2-3 form below, one binary search per element and level).  Every
            // lane produces E CONSECUTIVE places of the level's output: one search along the diagonal of its first place -- how many of the
            // places before it come from the pair's first run -- and then a serial two-finger merge of E elements, one key read per element.
            // The search form spent 11 steps of ~8 instructions on every element at every level; this spends them once per lane and level.
            // A product's sort was half of the operator (1603 raw terms: 67 k of 119 k cycles; a 40 x 40 cross product: 168 k of 240 k).
            // Ties as before: between runs of a's terms the earlier run first, between runs of b's terms the later.  Same order, same bits.
            {
                const int pl = 2 * rl;                      // a pair of runs, in the q + 1 numbering
                const int E = ((N + nl - 1) / nl) | 1;      // (odd: the lanes' places start an odd number of keys apart -- not all in one LDS bank)
                int c = w.lane2 * E;
                const int cend = min(c + E, N);
                while (c < cend) {                          // (a lane's places may reach into the next pair)
                    const int pr = (int)(((unsigned long long)(c + 1) * magic) >> 32) >> (lv + 1);
                    const int ps = max(pr * pl - 1, 0);                  // the pair's first run: [ps, am), its second: [am, pe)
                    const int am = min(pr * pl + rl - 1, N), pe = min((pr + 1) * pl - 1, N);
                    const int c9 = min(cend, pe);                        // this lane's places of the pair: [c, c9)
                    merge_span(K, V, Ko, Vo, ps, am, pe, c, c9, by_a, N);
                    c = c9;
                }
            }
#else
            for (int p0 = w.lane2; p0 < N; p0 += nl * U) {
                int ss[U], len[U], cnt[U], dst[U];
                pzkey_t tg[U], ky[U];
                bool ok[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int p = p0 + nl * u;
                    ok[u] = p < N;
                    const int pc = ok[u] ? p : 0;
                    const int r = (int)(((unsigned long long)(pc + 1) * magic) >> 32) >> lv;
                    const int s0 = max(r * rl - 1, 0);
                    const int rs = r ^ 1;
                    const int sb = max(rs * rl - 1, 0), se = min((rs + 1) * rl - 1, N);
                    len[u] = max(se - sb, 0);  // 0: the last run of an odd count has no sibling
                    ss[u] = len[u] > 0 ? sb : 0;
                    ky[u] = K[BIDX(w, pc, N, 1)];
                    // count the sibling's keys below ky -- and its equal key too if the sibling's term goes first on a tie
                    tg[u] = ky[u] + (pzkey_t)(((r & 1) != 0) == by_a ? 1 : 0);
                    cnt[u] = 0;
                    dst[u] = min(s0, sb) + (pc - s0);
                }
                for (int step = top; step > 0; step >>= 1) {
                    pzkey_t v[U];
#pragma unroll
                    for (int u = 0; u < U; u++) v[u] = K[BIDX(w, ss[u] + max(min(cnt[u] + step, len[u]), 1) - 1, N, 2)];
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const int take = (int)(cnt[u] + step <= len[u]) & (int)(v[u] < tg[u]);  // (no short circuit: a branch would serialise the reads)
#ifdef TREE_MERGE_SELECT_FORM  /* the value-identical rewrite that faulted in round 1; kept for the root-cause tooling only:
                                  a bit mask over the product shapes (1: 3x3*3x1, 2: 3x3*3x3, 4: 1x1*1x1 incl. cross, 8: 1x1*3x1) */
                        if constexpr ((((TREE_MERGE_SELECT_FORM) >> (SH::ASZ == 9 ? (SH::BSZ == 3 ? 0 : 1) : (SH::BSZ == 1 ? 2 : 3))) & 1) != 0)
                            cnt[u] = take ? cnt[u] + step : cnt[u];
                        else
                            cnt[u] += take ? step : 0;
#else
                        cnt[u] += take ? step : 0;
#endif
                    }
                }
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (ok[u]) { const int o__ = BIDX(w, dst[u] + cnt[u], N, 3); Ko[o__] = ky[u]; Vo[o__] = V[BIDX(w, p0 + nl * u, N, 4)]; }
            }
#endif
            psync(w);
            cur ^= 1;
        }
        PROF_ADD(PR_SORT) PROF_ADD(PR_S_MULMERGE)
    }
    // Levels [0, nlev) of the tree over the positions [q_lo, q_hi) alone -- a group of runs that the first nlev levels never merge with a run
    // outside it -- between two buffers of their own: position q of buffer k sits at entry base[k] + q - q_lo.  The result is in buffer 0.
    __device__ inline void tree_group(Wave& w, int q_lo, int q_hi, int nlev, int base0, int base1) const {
        const int na1 = a.cnt + 1;
        const bool by_a = na1 <= mb1;
        const int d1 = by_a ? mb1 : na1;
        const unsigned long long magic = magic_u32(d1);
        const int nl = w.nl, Ng = q_hi - q_lo;
        int cur = nlev & 1;
        LDS_AS pzkey_t* kb[2] = {w.skey + base0 - q_lo, w.skey + base1 - q_lo};
        LDS_AS uint16_t* vb[2] = {w.sidx + base0 - q_lo, w.sidx + base1 - q_lo};
        constexpr int KF = 4;
        for (int q0 = q_lo + w.lane2; q0 < q_hi; q0 += KF * nl) {
            pzkey_t kk[KF];
            int gi[KF];
#pragma unroll
            for (int u = 0; u < KF; u++) {
                const int q = min(q0 + u * nl, q_hi - 1);
                if (by_a) { kk[u] = key(q); gi[u] = q; }
                else {
                    const int j = (int)(((unsigned long long)(q + 1) * magic) >> 32), i = q + 1 - j * na1;
                    kk[u] = (i ? a.keys[i - 1] : 0ull) + (j ? b.keys[j - 1] : 0ull);
                    gi[u] = i * mb1 + j - 1;
                }
            }
#pragma unroll
            for (int u = 0; u < KF; u++) { const int q = q0 + u * nl; if (q < q_hi) { kb[cur][q] = kk[u]; vb[cur][q] = (uint16_t)gi[u]; } }
        }
        psync(w);
        for (int lv = 0; lv < nlev; lv++) {
            const int rl = d1 << lv, pl = 2 * rl;
            const int E = ((Ng + nl - 1) / nl) | 1;
            int c = q_lo + w.lane2 * E;
            const int cend = min(c + E, q_hi);
            while (c < cend) {
                const int pr = (int)(((unsigned long long)(c + 1) * magic) >> 32) >> (lv + 1);
                const int ps = max(pr * pl - 1, 0);
                const int am = min(pr * pl + rl - 1, q_hi), pe = min((pr + 1) * pl - 1, q_hi);
                const int c9 = min(cend, pe);
                merge_span(kb[cur], vb[cur], kb[cur ^ 1], vb[cur ^ 1], ps, am, pe, c, c9, by_a, q_hi);
                c = c9;
            }
            psync(w);
            cur ^= 1;
        }
    }
    // A product a little too long for the tree's two buffers (2 N entries) with too many runs to rank went to the bitonic network over all
    // next_pow2(N) entries: 78 stages for 2 049 .. 4 096 raw terms, 0.2 M cycles and more where the tree takes 0.06 M -- and a cross product of the
    // last links is 2 050 .. 2 250 raw terms in a good part of the worlds (4 of the reference's 107: their builds took 1.0 ms instead of 0.8).
    // Here the two halves the tree's LAST level would merge are sorted by the tree's other levels, each between buffers that fit (the second
    // half's in the space the first leaves), laid out as one ascending and one descending half of next_pow2(N) entries with the padding in
    // the middle -- a bitonic sequence -- and merged by the LAST stage of the network alone: log2 P passes.  The order is the one every path
    // produces: by (key, generation index).
    __device__ inline void split_merge(Wave& w, int N, const Split& sp) const {
        PROF_T0
        const int nl = w.nl, P = sp.P, N1 = sp.N1, N2 = sp.N2;
        tree_group(w, 0, sp.q0, sp.levels - 1, 0, N1);                 // -> [0, N1) ascending
        tree_group(w, sp.q0, N, sp.levels - 1, N1, N1 + N2);           // -> [N1, N1 + N2) ascending
        for (int t = w.lane2; t < N2; t += nl) {                       // the second half reversed into the END of the array ([N1 + N2, P) holds nothing: N1 + 2 N2 <= P)
            w.skey[P - 1 - t] = w.skey[N1 + t];
            w.sidx[P - 1 - t] = w.sidx[N1 + t];
        }
        psync(w);
        for (int t = N1 + w.lane2; t < P - N2; t += nl) { w.skey[t] = PZKEY_MAX; w.sidx[t] = (uint16_t)0xffffu; }   // the padding, between the two
        psync(w);
        if (w.half == 0) bitonic_sort(w, P, P);
        psync(w);
        PROF_ADD(PR_SORT) PROF_ADD(PR_S_BITONIC)
    }
    __device__ inline bool rank_merge(Wave& w, int N) const {
        const bool a_short = a.cnt <= b.cnt;
        const int ns = (a_short ? a.cnt : b.cnt) + 1, nl = (a_short ? b.cnt : a.cnt) + 1;
        if (!can_rank(w)) return false;
#ifdef DBG_NO_MERGE_MUL
        return false;
#endif
        PROF_T0
        const GLB_AS pzkey_t* lk = a_short ? b.keys : a.keys;
        const GLB_AS pzkey_t* sk = a_short ? a.keys : b.keys;
        for (int t = w.lane2; t < nl; t += w.nl) w.skey[t] = t ? lk[t - 1] : 0ull;
        if (w.lane2 < ns) w.skey[nl + w.lane2] = w.lane2 ? sk[w.lane2 - 1] : 0ull;
        psync(w);
        const LDS_AS pzkey_t* L = w.skey;
        const LDS_AS pzkey_t* S = w.skey + nl;
        for (int idx = w.lane2; idx < N; idx += w.nl) {
            int i, j;
            split(idx, i, j);
            const int r = a_short ? i : j, pos = a_short ? j : i;
            const pzkey_t ky = S[r] + L[pos];
            int rank = pos - 1;  // the (centre, centre) term sorts first and is not a monomial
            for (int r2 = 0; r2 < ns; r2++) {
                if (r2 == r || ky < S[r2]) continue;
                const pzkey_t tgt = ky - S[r2];
                rank += (r2 < r) ? upper_bound_lds(L, nl, tgt) : lower_bound_lds(L, nl, tgt);
            }
            w.sidx[rank] = (uint16_t)idx;
        }
        psync(w);
        PROF_ADD(PR_SORT) PROF_ADD(PR_S_MULMERGE)
        return true;
    }
    __device__ inline pzkey_t key_lds(const Wave& w, int idx) const {
        int i, j;
        split(idx, i, j);
        const bool a_short = a.cnt <= b.cnt;
        const int nl = (a_short ? b.cnt : a.cnt) + 1;
        return w.skey[nl + (a_short ? i : j)] + w.skey[a_short ? j : i];
    }
    __device__ inline void coef(int idx, double* c) const {
        int i, j;
        split(idx, i, j);
#ifdef DBG_BOUNDS
        if (i < 0 || i > a.cnt || j < 0 || j >= mb1) { i = 0; j = 0; }   // (reported by the caller-side checks of sidx below)
#endif
        double ca[SH::ASZ], cb[SH::BSZ];
        if (i) {
            const GLB_AS double* pa = a.coef + (size_t)(i - 1) * a.stride + a.off;
#pragma unroll
            for (int e = 0; e < SH::ASZ; e++) ca[e] = pa[e];
        } else {
#pragma unroll
            for (int e = 0; e < SH::ASZ; e++) ca[e] = a.cen[a.off + e];
        }
        if (j) {
            const GLB_AS double* pb = b.coef + (size_t)(j - 1) * b.stride + b.off;
#pragma unroll
            for (int e = 0; e < SH::BSZ; e++) cb[e] = pb[e];
        } else {
#pragma unroll
            for (int e = 0; e < SH::BSZ; e++) cb[e] = b.cen[b.off + e];
        }
        SH::mul(ca, cb, c);
    }
};

// |centre| + sum |coef| over a view, entry-wise (RT/PZsparse.cu:945-949,962-966)
template <int SZ>
__device__ inline void abs_sum(const Wave& w, const View& v, double* r) {
#pragma unroll
    for (int e = 0; e < SZ; e++) r[e] = 0.0;
    for (int m = w.lane; m < v.cnt; m += WAVE) {
        const GLB_AS double* c = v.coef + (size_t)m * v.stride + v.off;
#pragma unroll
        for (int e = 0; e < SZ; e++) r[e] += fabs(c[e]);
    }
#pragma unroll
    for (int e = 0; e < SZ; e++) r[e] = fabs(v.cen[v.off + e]) + wave_sum(r[e]);
}

// The N raw terms of `ev` already come in key order with unique keys (a product whose left operand has no monomials:
// mass, inertia and the fixed rpy rotation times a PZ -- term m is `centre_a * coef_b[m]` under b's m-th key): the pass of
// sort_reduce_emit without the sort and without runs.  Same lane-to-term assignment, so the results are identical.
template <int SZ, class Eval>
__device__ inline void emit_presorted(Wave& w, int N, const Eval& ev, const PZ& out, const double* base_ind, const double* base_ind2) {
    int emitted = 0;
    bool any_pruned = false;
    double ra[SZ];
#pragma unroll
    for (int e = 0; e < SZ; e++) ra[e] = 0.0;
    if (w.lane == 0 && N > w.lstat[ST_MAX_RAW]) w.lstat[ST_MAX_RAW] = N;
#ifdef P1_PROFILE
    if (w.lane == 0) { w.prof[PR_CALLS] += 1; w.prof[PR_TERMS] += N; if (N <= 64) w.prof[PR_SMALL] += 1; }
#endif
    PROF_T0
    for (int base = 0; base < N; base += WAVE) {
        const int p = base + w.lane;
        bool keep = false, pruned = false;
        pzkey_t key = 0;
        double acc[SZ];
        if (p < N) {
            key = ev.key(p);
            ev.coef(p, acc);
            keep = !norm_le_t<SZ>(acc, w);
            if (!keep) {
#pragma unroll
                for (int e = 0; e < SZ; e++) ra[e] += fabs(acc[e]);
                pruned = true;
            }
        }
        const unsigned long long m = __ballot(keep);
        any_pruned = any_pruned || (__ballot(pruned) != 0ull);
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
    PROF_ADD(PR_EMIT)
    if (emitted > out.cap) { flag(w, ERR_SLOT_OVERFLOW); emitted = out.cap; }
    if (any_pruned) {
#pragma unroll
        for (int e = 0; e < SZ; e++) ra[e] = wave_sum(ra[e]);
    }
    if (w.lane == 0) {
#pragma unroll
        for (int e = 0; e < SZ; e++) { out.ind[e] = base_ind[e] + ra[e]; out.ind2[e] = base_ind2[e] + ra[e]; }
        w.cnt[out.id] = emitted;
        if (emitted > w.lstat[ST_MAX_OUT]) w.lstat[ST_MAX_OUT] = emitted;
    }
    mflush(w);
    WSYNC();
}

// The product sorts ALL raw terms.  (Rounds 2-3 carried a second, hash-classified product behind a build flag -- unique keys decided without
// the sort; measured slower on the chip and never shipped.  It is kept as a text, docs/experiments/pz_hash.h.txt, not as source.)
template <int AR, int AC, int BR, int BC>
__device__ PZW_NOINLINE void mul(Wave& w_, const PZ& out_, const View& a_, const View& b_) {
    PZ_KEEP_RETURN_ADDRESS();
    PZW_WAVE_LOCAL(w, w_)
    const PZ out = uni_pz(out_);
    const View a = uni_view(a_), b = uni_view(b_);
    PROF_CALL_T0
    typedef MulShape<AR, AC, BR, BC> SH;
    MulEval<SH> ev;
    ev.a = a; ev.set_b(b);
    const int N = (a.cnt + 1) * (b.cnt + 1) - 1;
    double r2[SH::ASZ], r3[SH::BSZ], ia[SH::ASZ], ib[SH::BSZ], ca[SH::ASZ], cb[SH::BSZ];
    { PROF_T0
    abs_sum<SH::ASZ>(w, a, r2);
    abs_sum<SH::BSZ>(w, b, r3);
    PROF_ADD(PR_ABS) }
    double ia2[SH::ASZ], ib2[SH::BSZ];
#pragma unroll
    for (int e = 0; e < SH::ASZ; e++) { ia[e] = a.ind[a.off + e]; ia2[e] = a.ind2[a.off + e]; ca[e] = a.cen[a.off + e]; }
#pragma unroll
    for (int e = 0; e < SH::BSZ; e++) { ib[e] = b.ind[b.off + e]; ib2[e] = b.ind2[b.off + e]; cb[e] = b.cen[b.off + e]; }
    double t2[SH::SZ], t3[SH::SZ], ii[SH::SZ], cen[SH::SZ], base[SH::SZ], base2[SH::SZ];
    SH::mul(r2, ib, t2);   // (|c_a| + sum|coef_a|) * indep_b
    SH::mul(ia, r3, t3);   // indep_a * (|c_b| + sum|coef_b|)
    SH::mul(ia, ib, ii);
    SH::mul(ca, cb, cen);
#pragma unroll
    for (int e = 0; e < SH::SZ; e++) base[e] = ii[e] + (t2[e] + t3[e]);
    SH::mul(r2, ib2, t2);  // the same with the second (uncertain-parameter) radii; r2, r3 are shared
    SH::mul(ia2, r3, t3);
    SH::mul(ia2, ib2, ii);
#pragma unroll
    for (int e = 0; e < SH::SZ; e++) base2[e] = ii[e] + (t2[e] + t3[e]);
    psync(w);
    if (w.lane2 == 0) {
#pragma unroll
        for (int e = 0; e < SH::SZ; e++) out.cen[e] = cen[e];
    }
    if (a.cnt == 0) {   // constant left operand: b's keys, in b's order  (a pair: the first wave alone)
        if (w.half == 0) emit_presorted<SH::SZ>(w, N, ev, out, base, base2);
        psync(w);
    } else sort_reduce_emit<SH::SZ>(w, N, ev, out, base, base2);
    PROF_CALL_END(N)
}

// ---------------------------------------------------------------------------------------------------
// cross(PZ a, PZ b) for 3x1 operands (RT/PZsparse.cu:1134-1151) in ONE pass.
// The reference composes it from 1x1 operators, each ending in simplify():
//     P(c,0) = a[c+1] * b[c+2],  P(c,1) = a[c+2] * b[c+1]      (6 scalar products, indices mod 3)
//     r_c    = P(c,0) - P(c,1)                                  (3 differences)
//     out    = stack(r_0, r_1, r_2)
// All six products run over the same raw terms -- the pairs (monomial or centre of a) x (monomial or centre of b), with
// the same key per pair -- so one sort of those pairs serves them all.  For every run of equal keys the head lane
// sums the six coefficient products in generation order and then applies the three simplify() stages in sequence,
// exactly as the composed operators would: |sum| <= threshold drops a product term into that product's radius; the
// surviving terms of a component are combined as 1.0*P0 + (-1.0)*P1 and pruned again into the difference's radius;
// the surviving components form the 3-vector that stack() prunes by its norm.  Radii are assembled in the order of the
// composed operators (product radius + pruned, summed over the two products, + pruned, + pruned by stack).
struct CrossEval : MulEval<MulShape<1, 1, 1, 1>> {
    // p6[2*c + s] = coefficient product of P(c,s) for raw pair idx
    __device__ inline void coef6(int idx, double* p6) const {
        int i, j;
        split(idx, i, j);
        double ca[3], cb[3];
        if (i) {
            const GLB_AS double* pa = a.coef + (size_t)(i - 1) * 3;
            ca[0] = pa[0]; ca[1] = pa[1]; ca[2] = pa[2];
        } else { ca[0] = a.cen[0]; ca[1] = a.cen[1]; ca[2] = a.cen[2]; }
        if (j) {
            const GLB_AS double* pb = b.coef + (size_t)(j - 1) * 3;
            cb[0] = pb[0]; cb[1] = pb[1]; cb[2] = pb[2];
        } else { cb[0] = b.cen[0]; cb[1] = b.cen[1]; cb[2] = b.cen[2]; }
        p6[0] = ca[1] * cb[2]; p6[1] = ca[2] * cb[1];
        p6[2] = ca[2] * cb[0]; p6[3] = ca[0] * cb[2];
        p6[4] = ca[0] * cb[1]; p6[5] = ca[1] * cb[0];
    }
};

__device__ PZW_NOINLINE void cross_pzpz(Wave& w_, const PZ& out_, const View& a_, const View& b_) {
    PZ_KEEP_RETURN_ADDRESS();
    PZW_WAVE_LOCAL(w, w_)
    const PZ out = uni_pz(out_);
    const View a = uni_view(a_), b = uni_view(b_);
    PROF_CALL_T0
    CrossEval ev;
    ev.a = a; ev.set_b(b);
    int N = (a.cnt + 1) * (b.cnt + 1) - 1;
    [[maybe_unused]] const int N_in = N;
    // centres and radii of the six products, as mul<1,1,1,1> forms them
    double r2[3], r3[3];
    { PROF_T0
    abs_sum<3>(w, a, r2);
    abs_sum<3>(w, b, r3);
    PROF_ADD(PR_ABS) }
    double cenP[6], baseP[6], base2P[6];
    {
        const int ia_[6] = {1, 2, 2, 0, 0, 1}, ib_[6] = {2, 1, 0, 2, 1, 0};
#pragma unroll
        for (int e = 0; e < 6; e++) {
            const int i = ia_[e], j = ib_[e];
            const double ia = a.ind[i], ib = b.ind[j], ia2 = a.ind2[i], ib2 = b.ind2[j];
            cenP[e] = a.cen[i] * b.cen[j];
            baseP[e] = ia * ib + (r2[i] * ib + ia * r3[j]);
            base2P[e] = ia2 * ib2 + (r2[i] * ib2 + ia2 * r3[j]);
        }
    }
    WSYNC();
    int emitted = 0;
    bool any_pruned = false, indirect = false;
    double raP[6] = {0, 0, 0, 0, 0, 0}, raR[3] = {0, 0, 0}, raS[3] = {0, 0, 0};
    N = uni_i(sort_terms(w, N, ev, indirect));
    if (N > 0) {
        PROF_T0
        for (int base = 0; base < N; base += WAVE) {
            const int p = base + w.lane;
            bool head = false, keep = false;
            pzkey_t key = 0;
            double u[3] = {0.0, 0.0, 0.0};
            if (p < N) {
                key = indirect ? ev.key_lds(w, w.sidx[p]) : w.skey[p];
                head = (p == 0) || ((indirect ? ev.key_lds(w, w.sidx[p - 1]) : w.skey[p - 1]) != key);
            }
            // (own term in every lane, run members shifted down the wave: see sort_reduce_emit)
            double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            if (p < N) ev.coef6(w.sidx[p], acc);
            {
                double sh[6];
#pragma unroll
                for (int e = 0; e < 6; e++) sh[e] = acc[e];
                for (int j = 1;; j++) {
                    const int q = p + j;
                    const bool more = head && q < N && (indirect ? ev.key_lds(w, w.sidx[q]) : w.skey[q]) == key;
                    if (__ballot(more) == 0ull) break;
#pragma unroll
                    for (int e = 0; e < 6; e++) sh[e] = dpp_take<0x130, 0xf>(sh[e]);
                    if (more) {
                        double c6[6];
#pragma unroll
                        for (int e = 0; e < 6; e++) c6[e] = sh[e];
                        if (w.lane + j >= WAVE) ev.coef6(w.sidx[q], c6);
#pragma unroll
                        for (int e = 0; e < 6; e++) acc[e] += c6[e];
                    }
                }
            }
            if (head) {
                bool anyc = false, pruned = false;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    // simplify() of each product: 1x1 norm
                    double v0 = acc[2 * c], v1 = acc[2 * c + 1];
                    bool h0 = !norm1_le_t(w, v0), h1 = !norm1_le_t(w, v1);
                    if (!h0) { raP[2 * c] += fabs(v0); pruned = true; }
                    if (!h1) { raP[2 * c + 1] += fabs(v1); pruned = true; }
                    // simplify() of the difference 1.0*P0 + (-1.0)*P1 over the surviving terms
                    if (h0 || h1) {
                        double wv = h0 ? 1.0 * v0 : -1.0 * v1;
                        if (h0 && h1) wv += -1.0 * v1;
                        if (norm1_le_t(w, wv)) { raR[c] += fabs(wv); pruned = true; }
                        else { u[c] = wv; anyc = true; }
                    }
                }
                // simplify() of stack(): 3-vector norm over the surviving components
                if (anyc) {
                    double s = 0.0;
#pragma unroll
                    for (int c = 0; c < 3; c++) s += u[c] * u[c];
                    keep = !(s <= w.thr_sq);
                    mtrack(w, s);
                    if (!keep) {
#pragma unroll
                        for (int c = 0; c < 3; c++) raS[c] += fabs(u[c]);
                        pruned = true;
                    }
                }
                head = pruned;
            }
            const unsigned long long m = __ballot(keep);
            any_pruned = any_pruned || (__ballot(head) != 0ull);
            if (keep) {
                const int pos = emitted + __popcll(m & ((1ull << w.lane) - 1ull));
                if (pos < out.cap) {
                    out.keys[pos] = key;
                    out.coef[(size_t)pos * 3 + 0] = u[0]; out.coef[(size_t)pos * 3 + 1] = u[1]; out.coef[(size_t)pos * 3 + 2] = u[2];
                }
            }
            emitted += __popcll(m);
        }
        PROF_ADD(PR_EMIT)
    }
    if (emitted > out.cap) { flag(w, ERR_SLOT_OVERFLOW); emitted = out.cap; }
    if (any_pruned) {
#pragma unroll
        for (int e = 0; e < 6; e++) raP[e] = wave_sum(raP[e]);
#pragma unroll
        for (int c = 0; c < 3; c++) { raR[c] = wave_sum(raR[c]); raS[c] = wave_sum(raS[c]); }
    }
    if (w.lane == 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            // product: base + pruned; difference: sum of the two (scales 1, |-1|) + pruned; stack: 0 + that, + pruned
            const double i0 = baseP[2 * c] + raP[2 * c], i1 = baseP[2 * c + 1] + raP[2 * c + 1];
            const double j0 = base2P[2 * c] + raP[2 * c], j1 = base2P[2 * c + 1] + raP[2 * c + 1];
            const double ir = (i0 * 1.0 + i1 * 1.0) + raR[c], jr = (j0 * 1.0 + j1 * 1.0) + raR[c];
            out.cen[c] = 0.0 + (1.0 * cenP[2 * c] + -1.0 * cenP[2 * c + 1]);
            out.ind[c] = (0.0 + ir) + raS[c];
            out.ind2[c] = (0.0 + jr) + raS[c];
        }
        w.cnt[out.id] = emitted;
        if (emitted > w.lstat[ST_MAX_OUT]) w.lstat[ST_MAX_OUT] = emitted;
    }
    mflush(w);
    psync(w);
    PROF_CALL_END(N_in)
}

// Cross product of a 3x1 PZ with a constant vector, either order (RT/PZsparse.cu:1118-1132, 1153-1167):
//     out[c] = sA[c] * a[cA[c]] + sB[c] * a[cB[c]],  c = 0..2.
// The reference builds each entry as (double * 1x1 PZ) - (double * 1x1 PZ) -- both operands carry a's key list, so
// simplify() just adds the two coefficients per key and prunes |.| <= threshold -- and then stack()s the three
// entries, whose simplify() merges equal keys with disjoint non-zero entries and prunes by the 3-vector norm.
// Because the key list never changes, the whole thing is one ordered pass over a's monomials: no sort.
__device__ PZW_NOINLINE void cross_const(Wave& w_, const PZ& out_, const View& a_, const double* sA, const int* cA,
                                   const double* sB, const int* cB) {
    PZ_KEEP_RETURN_ADDRESS();
    PZW_WAVE_LOCAL(w, w_)
    const PZ out = uni_pz(out_);
    const View a = uni_view(a_);
    double cen[3], ind[3], ind2[3], ra1[3] = {0, 0, 0}, ra2[3] = {0, 0, 0};
#pragma unroll
    for (int c = 0; c < 3; c++) {
        cen[c] = sA[c] * a.cen[cA[c]] + sB[c] * a.cen[cB[c]];
        ind[c] = a.ind[cA[c]] * fabs(sA[c]) + a.ind[cB[c]] * fabs(sB[c]);
        ind2[c] = a.ind2[cA[c]] * fabs(sA[c]) + a.ind2[cB[c]] * fabs(sB[c]);
    }
    int emitted = 0;
    bool any1 = false, any2 = false;
    for (int base = 0; base < a.cnt; base += WAVE) {
        const int m = base + w.lane;
        bool keep = false;
        double r[3] = {0, 0, 0};
        pzkey_t key = 0;
        if (m < a.cnt) {
            key = a.keys[m];
            const GLB_AS double* x = a.coef + (size_t)m * a.stride + a.off;
            const double x3[3] = {x[0], x[1], x[2]};
            bool anyc = false;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                double v = sA[c] * x3[cA[c]];
                v += sB[c] * x3[cB[c]];
                if (norm1_le_t(w, v)) { ra1[c] += fabs(v); any1 = true; v = 0.0; } else anyc = true;
                r[c] = v;
            }
            if (anyc) {
                const double s3 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
                keep = !(s3 <= w.thr_sq);
                mtrack(w, s3);
                if (!keep) { ra2[0] += fabs(r[0]); ra2[1] += fabs(r[1]); ra2[2] += fabs(r[2]); any2 = true; }
            }
        }
        const unsigned long long mk = __ballot(keep);
        if (keep) {
            const int pos = emitted + __popcll(mk & ((1ull << w.lane) - 1ull));
            if (pos < out.cap) {
                out.keys[pos] = key;
                out.coef[(size_t)pos * 3 + 0] = r[0]; out.coef[(size_t)pos * 3 + 1] = r[1]; out.coef[(size_t)pos * 3 + 2] = r[2];
            }
        }
        emitted += __popcll(mk);
    }
    if (emitted > out.cap) { flag(w, ERR_SLOT_OVERFLOW); emitted = out.cap; }
    if (__ballot(any1) != 0ull) { ra1[0] = wave_sum(ra1[0]); ra1[1] = wave_sum(ra1[1]); ra1[2] = wave_sum(ra1[2]); }
    if (__ballot(any2) != 0ull) { ra2[0] = wave_sum(ra2[0]); ra2[1] = wave_sum(ra2[1]); ra2[2] = wave_sum(ra2[2]); }
    WSYNC();
    if (w.lane == 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) { out.cen[c] = cen[c]; out.ind[c] = (ind[c] + ra1[c]) + ra2[c]; out.ind2[c] = (ind2[c] + ra1[c]) + ra2[c]; }
        w.cnt[out.id] = emitted;
        if (emitted > w.lstat[ST_MAX_OUT]) w.lstat[ST_MAX_OUT] = emitted;
    }
    mflush(w);
    WSYNC();
}

// out = a^T for 3x3 (RT/PZsparse.cu:1050-1066); keys unchanged, no simplify.
__device__ PZW_NOINLINE void transpose33(Wave& w, const PZ& out, const PZ& a) {
    PZ_KEEP_RETURN_ADDRESS();
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
        out.ind2[c * 3 + r] = a.ind2[w.lane];
    }
    if (w.lane == 0) w.cnt[out.id] = n;
    WSYNC();
}

// constant PZ (centre + independent radius, no monomials): RT/PZsparse.cu:66-98
// ind2 == nullptr: the second radius equals the first (everything except the mass / inertia PZs)
__device__ PZW_NOINLINE void set_const(Wave& w, const PZ& out, const double* cen, const double* ind, const double* ind2 = nullptr) {
    PZ_KEEP_RETURN_ADDRESS();
    if (w.lane < out.sz) {
        out.cen[w.lane] = cen ? cen[w.lane] : 0.0;
        out.ind[w.lane] = ind ? ind[w.lane] : 0.0;
        out.ind2[w.lane] = ind2 ? ind2[w.lane] : (ind ? ind[w.lane] : 0.0);
    }
    if (w.lane == 0) w.cnt[out.id] = 0;
    WSYNC();
}

// plain copy (operator=)
__device__ inline void copy(Wave& w, const PZ& out, const PZ& a) {
    const int n = w.cnt[a.id], sz = a.sz;
    for (int t = w.lane; t < n * sz; t += WAVE) out.coef[t] = a.coef[t];
    for (int m = w.lane; m < n; m += WAVE) out.keys[m] = a.keys[m];
    if (w.lane < sz) { out.cen[w.lane] = a.cen[w.lane]; out.ind[w.lane] = a.ind[w.lane]; out.ind2[w.lane] = a.ind2[w.lane]; }
    if (w.lane == 0) w.cnt[out.id] = n;
    WSYNC();
}

}  // namespace pzw
