// Device code of the fused eval_g + eval_jac_g evaluation (P2), shared by the stand-alone kernel (p2_eval.hip) and the
// on-device NLP solver (solver_device.hip): k-power tables, monomial slicing, and the three kinds of row tiles --
//   collision_block   64 consecutive collision rows (RT/CollisionChecking.cu:230-299 + RT/PZsparse.cu:404-516)
//   torque_block      8 torque rows x 32 monomial lanes (RT/NLPclass.cu:304-310,376-383)
//   limit_block       the 4n joint position / velocity limit rows (RT/Trajectory.cu:256-540)
// as functions of (problem b, tile index) instead of blockIdx, so that a persistent block can walk over several tiles.
// One source of arithmetic: whatever launches these produces bit-identical rows.
#pragma once
#include <algorithm>
#include <cstdlib>

#include "bezier.h"
#include "cacc.h"
#include "common.h"

#define P2_BLOCK 256
#define P2_ROWS 64        // collision rows per block (one per lane); the 4 waves split the 36 planes 9 each
#define P2_PPW 9          // planes per wave (36 / 4); the PPW template parameter is 9, or 6 when no problem has more than 24 live planes
#define P2_TASK_ROUNDS 2   // (monomial, axis) slicing tasks a thread preloads per LDS pass
#define P2_TQ_ROWS 8      // torque rows per block (32 monomial lanes each)
#define P2_TQ_ROUNDS 4    // monomials per lane of a torque row (strideT <= 128)
#define P2_SL (3 * (1 + ARMOUR_MAX_FACTORS))   // doubles of one sliced link monomial / one sliced link PZ: x[3], dx[ARMOUR_MAX_FACTORS][3] (24; 27 in the 8-factor build)
#define P2_TQW (1 + ARMOUR_MAX_FACTORS)        // doubles of one sliced torque monomial: value and ARMOUR_MAX_FACTORS partials (8; 9)

#ifndef P2_DFC_WAVES
#define P2_DFC_WAVES 3
#endif
// batched one-point launches (DFC) are occupancy-bound: hold the kernel to the 168 VGPRs of 3 waves per SIMD (4 was measured:
// 128 VGPRs + 120 B/lane of scratch, twice as slow)
#define P2_WPE(DFC, MULTI) ((DFC) && !(MULTI) ? P2_DFC_WAVES : 1)

namespace p2 {


#ifdef P2_TIMELINE  // development only: 100 MHz wall-clock stamps of the collision phases of three blocks, left in the limit rows of g
#define P2_STAMP(i) do { unsigned long long t__; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__) :: "memory"); stamp[i] = t__; } while (0)
#else
#define P2_STAMP(i)
#endif

// k-power tables in LDS: pw[j][d] = k_j^d (d = 0..3); df[j][d] = d * k_j^(d-1)
struct KPow {
    double pw[ARMOUR_MAX_FACTORS][4];
    double df[ARMOUR_MAX_FACTORS][4];
};

__device__ inline void fill_kpow(KPow& kp, double x, int n) {
    const int j = threadIdx.x;
    if (j < n) {
        kp.pw[j][0] = 1.0; kp.pw[j][1] = x; kp.pw[j][2] = x * x; kp.pw[j][3] = x * x * x;
        kp.df[j][0] = 0.0; kp.df[j][1] = 1.0; kp.df[j][2] = 2.0 * x; kp.df[j][3] = 3.0 * (x * x);
    }
}

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would wait
// for the plane loads a wave has in flight; the EX kernels keep those in flight across the slicing barriers.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// value of coeff * prod_j k_j^{d_j}, multiplying in factor order (RT/PZsparse.cu:416-418).  The 7 table reads are
// independent of the multiply chain (pw[j][0] = 1.0 makes the absent factors exact no-ops), so they issue together.
__device__ inline double mono_value(const KPow& kp, uint32_t key, double c, int n) {
    double f[ARMOUR_MAX_FACTORS];
#pragma unroll
    for (int j = 0; j < ARMOUR_MAX_FACTORS; j++) f[j] = (j < n) ? kp.pw[j][(key >> (2 * j)) & 3u] : 1.0;
    double v = c;
#pragma unroll
    for (int j = 0; j < ARMOUR_MAX_FACTORS; j++) v *= f[j];
    return v;
}
// d/dk_kk of the same monomial (RT/PZsparse.cu:454-468): the kk-th factor is d*k^(d-1) (0 when k_kk is absent)
__device__ inline double mono_grad(const KPow& kp, uint32_t key, double c, int n, int kk) {
    double f[ARMOUR_MAX_FACTORS];
#pragma unroll
    for (int j = 0; j < ARMOUR_MAX_FACTORS; j++) {
        const uint32_t d = (key >> (2 * j)) & 3u;
        f[j] = (j < n) ? (j == kk ? kp.df[j][d] : kp.pw[j][d]) : 1.0;
    }
    double v = c;
#pragma unroll
    for (int j = 0; j < ARMOUR_MAX_FACTORS; j++) v *= f[j];
    return v;
}
// value and all n partials of one monomial with ONE decode of the key and one set of table reads; each of the
// 8 products still multiplies coeff * f_0 * ... * f_{n-1} in factor order as the reference does.
template <bool WANT_J>
__device__ inline void mono_all(const KPow& kp, uint32_t key, double c, int n, double* o) {
    double pw[ARMOUR_MAX_FACTORS], df[ARMOUR_MAX_FACTORS];
#pragma unroll
    for (int j = 0; j < ARMOUR_MAX_FACTORS; j++) {
        const uint32_t d = (key >> (2 * j)) & 3u;
        pw[j] = (j < n) ? kp.pw[j][d] : 1.0;
        df[j] = (j < n) ? kp.df[j][d] : 0.0;
    }
    double v = c;
#pragma unroll
    for (int j = 0; j < ARMOUR_MAX_FACTORS; j++) v *= pw[j];
    o[0] = v;
    if (WANT_J) {
#pragma unroll
        for (int kk = 0; kk < ARMOUR_MAX_FACTORS; kk++) {
            double g = c;
#pragma unroll
            for (int j = 0; j < ARMOUR_MAX_FACTORS; j++) g *= (j == kk) ? df[j] : pw[j];
            o[1 + kk] = g;
        }
    }
}
// centre of Interval(c - r, c + r) as getCenter computes it (RT/PZsparse.cu:10-12,427-432)
__device__ inline double interval_center(double c, double r) {
    const double lo = c - r, hi = c + r;
    return (lo + hi) * 0.5;
}

// x / d for 0 <= x < 2^20 and 1 <= d < 2^20 with the host-computed m = ceil(2^40 / d): exact, one 64-bit multiply
// (the compiler's runtime-divisor sequence is ~30 instructions; the fused kernel divides by O and by 3*strideL per thread)
__host__ __device__ inline unsigned long long div_magic(int d) { return ((1ull << 40) + (unsigned long long)d - 1ull) / (unsigned long long)d; }
__device__ inline int fast_div(int x, unsigned long long m) { return (int)(((unsigned long long)(unsigned)x * m) >> 40); }

// 16-B table entries as a NATIVE vector: a load of HIP's double2 (a struct) is split into two 8-B loads by the optimiser before the
// two plane-loading paths of the EX kernels are merged, and the backend cannot pair them again afterwards (seen in the ISA)
typedef double v2d __attribute__((ext_vector_type(2)));

struct P2Launch {
    int nbc, nbt;       // collision / torque block counts
    int role0;          // first block role of the launch: 0, or nbc for a launch without the collision blocks (relevance.hip: the culled row test)
    int max_pairs;      // (l,t) pairs a collision block can touch
    int strideL;        // monomial stride of the LDS term buffer for link PZs (>= max link count)
    int strideT;        // same for torque PZs
    int pair_chunk;     // (l,t) pairs sliced per LDS pass
    int skip_by_value;  // 1: plane_skip of the (single) problem is in `skip0` (saves a dependent load at B = 1)
    unsigned long long skip0;
    // multi-point evaluation: the block keeps its share of the tables in registers and loops over `steps` points k;
    // point s reads k_all + s*k_stride and writes g_all + s*g_stride, jac_all + s*j_stride (strides in doubles; 0 = overwrite)
    int steps;
    long long k_stride, g_stride, j_stride;
    unsigned long long magic_O, magic_pp3;  // div_magic(O), div_magic(3 * strideL)
};

// registers of one slicing pass: the (monomial, axis) tasks of this thread and, for threads < pairs*24, the inputs of
// the ordered reduction.  None of it depends on k, so a block that evaluates several points loads it once.
struct PassRegs {
    uint32_t tkey[P2_TASK_ROUNDS];
    double tco[P2_TASK_ROUNDS];
    int tcnt[P2_TASK_ROUNDS], tdst[P2_TASK_ROUNDS];
    double rc_cen, rc_ind;
    int rc_cnt;
};

// (Measured at B = 1 against these predicated groups: unconditional clamped loads issued as one batch behind the planes,
// 3 % slower; the same plus a fixed 9 plane slots per wave, so that the slicing overlaps the planes in flight, 15 % slower.)
__device__ inline void load_pass(const P2Tables& tb, const P2Launch& lp, int b, int lt_first, int p0, int pc, PassRegs& pr) {
    const int tid = threadIdx.x;
    const int per_pair = lp.strideL * P2_SL, per_pair3 = lp.strideL * 3;
    const int ntask = pc * per_pair3;
#pragma unroll
    for (int r = 0; r < P2_TASK_ROUNDS; r++) {
        const int task = tid + r * P2_BLOCK;
        pr.tcnt[r] = -1; pr.tdst[r] = 0; pr.tkey[r] = 0; pr.tco[r] = 0.0;
        if (task < ntask) {
            const int pi = fast_div(task, lp.magic_pp3), rem = task - pi * per_pair3, mo = rem / 3, e = rem - mo * 3;
            const size_t idx = (size_t)b * tb.J * tb.T + (lt_first + p0 + pi);
            pr.tcnt[r] = tb.link_count[idx] - mo;  // > 0: live monomial
            pr.tdst[r] = pi * per_pair + mo * P2_SL + e;
            if (mo < tb.capL) {
                pr.tkey[r] = tb.link_keys[idx * tb.capL + mo];
                pr.tco[r] = tb.link_coeff[(idx * tb.capL + mo) * 3 + e];
            }
        }
    }
    pr.rc_cen = 0.0; pr.rc_ind = 0.0; pr.rc_cnt = 0;
    if (tid < pc * P2_SL) {
        const int pi = tid / P2_SL, c = tid - pi * P2_SL, e = c % 3;
        const size_t idx = (size_t)b * tb.J * tb.T + (lt_first + p0 + pi);
        pr.rc_cnt = min(tb.link_count[idx], lp.strideL);
        if (c < 3) { pr.rc_cen = tb.link_center[idx * 3 + e]; pr.rc_ind = tb.link_indep[idx * 3 + e]; }
    }
}

// DFC: d = A . c_obstacle is recomputed from tb.obs_center instead of read (batched launches of tables built by P1)
// LL: the link x link normals come from the compact tb.planes_ll (tables whose normals are obstacle-independent)
// PPW: plane slots per wave.  With axis-aligned box obstacles 12 of the 36 planes are skipped (armour_p1_planes_kernel), so a
// wave holds at most 6: the 6-slot instantiation drops a third of the slot loops (loads, d recomputation, scan, pick).
// The same registers through UNCONDITIONAL loads with clamped indices (threads without a task re-read task 0's entry) and no
// select on a loaded value (the compiler would sink the load into the branch): a fixed number of loads, so that loads issued
// after them can stay in flight while these are waited for (vmcnt counts in order and takes an immediate).
__device__ inline void load_pass_uncond(const P2Tables& tb, const P2Launch& lp, int b, int lt_first, int pc, PassRegs& pr) {
    const int tid = threadIdx.x;
    const int per_pair = lp.strideL * P2_SL, per_pair3 = lp.strideL * 3;
    const int ntask = pc * per_pair3;
    const size_t idx0 = (size_t)b * tb.J * tb.T + lt_first;
#pragma unroll
    for (int r = 0; r < P2_TASK_ROUNDS; r++) {
        const int task = tid + r * P2_BLOCK;
        const bool valid = task < ntask;
        const int tk = valid ? task : 0;
        const int pi = fast_div(tk, lp.magic_pp3), rem = tk - pi * per_pair3, mo = rem / 3, e = rem - mo * 3;
        const size_t idx = idx0 + pi;
        const int moc = min(mo, tb.capL - 1);
        const int cnt = tb.link_count[idx];
        pr.tkey[r] = tb.link_keys[idx * tb.capL + moc];
        pr.tco[r] = tb.link_coeff[(idx * tb.capL + moc) * 3 + e];
        pr.tcnt[r] = cnt - (valid ? mo : (1 << 20));  // > 0: live monomial (cnt <= capL, so a clamped mo is never live)
        pr.tdst[r] = pi * per_pair + mo * P2_SL + e;
    }
    const int tc = tid < pc * P2_SL ? tid : 0;
    const int pi = tc / P2_SL, c = tc - pi * P2_SL, e = c % 3;
    const size_t idx = idx0 + pi;
    pr.rc_cnt = min(tb.link_count[idx], lp.strideL);  // used by threads < pc*P2_SL only
    pr.rc_cen = tb.link_center[idx * 3 + e];            // used for the value column (c < 3) only
    pr.rc_ind = tb.link_indep[idx * 3 + e];
}

// EX: every wave holds exactly PPW live planes and the block slices in one pass: the PZ-table loads are issued first and
// the plane loads, a fixed number, after them, so the slicing (waiting for the tables with vmcnt(#plane loads)) runs while
// the planes are still in flight; the barriers before the scan wait for LDS traffic only.
template <bool WANT_G, bool WANT_J, bool MULTI, bool DFC, bool LL, int PPW, bool EX>
__device__ __forceinline__ void collision_block(const P2Tables& tb, const P2Launch& lp, const int b, const int role, const double* __restrict__ k0,
                                   const double k_first, double* __restrict__ g0_in, double* __restrict__ jac0_in, unsigned char* smem_raw) {
    KPow* kp2 = reinterpret_cast<KPow*>(smem_raw);  // double-buffered over the points
    double* lds = reinterpret_cast<double*>(smem_raw + 2 * sizeof(KPow));
    const int n = tb.n, T = tb.T, O = tb.O, Q = tb.Q, m = tb.m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave index, known uniform: the plane dealing below runs on the scalar unit
    // k0 / g0_in / jac0_in: THIS problem's k [n], g [m] and jac [m][n] (of the first point for multi-point launches);
    // k_first: this thread's component of the first point (tid < n), fetched by the caller as early as it can -- a fresh k
    // is an HBM miss, and at B = 1 the launch waits for it
    double* g0 = WANT_G ? g0_in : nullptr;
    double* jac0 = WANT_J ? jac0_in : nullptr;
    const int nsteps = MULTI ? lp.steps : 1;  // the one-point instantiation keeps the loop-free code of a single launch
    // this thread's component of the NEXT point's k is fetched one point ahead: a fresh k is an HBM miss
    double k_next = k_first;
    (void)lane; (void)wv; (void)T; (void)O; (void)Q; (void)m; (void)kp2; (void)lds; (void)g0; (void)jac0; (void)nsteps; (void)k_next;
#ifdef P2_TIMELINE
    unsigned long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    P2_STAMP(0);
#ifdef P2_TIMELINE
    const unsigned long long cyc0 = __builtin_amdgcn_s_memtime();
#endif
    // ------------------------------------------------------------------ collision rows
    double* sx = lds;                                               // [max_pairs][P2_SL]: x[3], dx[ARMOUR_MAX_FACTORS][3]
    double* terms = sx + (size_t)lp.max_pairs * P2_SL;                 // [pair_chunk][strideL][P2_SL]
    double* part = terms + (size_t)lp.pair_chunk * lp.strideL * P2_SL; // [4][64][4] partial scans
    int* pneg = reinterpret_cast<int*>(part + 4 * 64 * 4);          // [4][64]
    double* stage = part + 4 * 64 * 4 + 4 * 64 / 2;                 // [64][n] Jacobian staging tile

    const int q_begin = role * P2_ROWS;
    const int q_end = min(Q, q_begin + P2_ROWS);
    const int q = min(q_begin + lane, q_end - 1);
    // 1. issue this lane's share of the live planes x 5 components first: nothing below depends on them until
    //    step 3.  Planes flagged in plane_skip[b] (degenerate or exact duplicates in every row, see
    //    armour_p1_planes_kernel) are not fetched; the live ones are dealt to the 4 waves in ascending order.
    // layout: common.h armour_plane_index -- 16-B pairs {Ax,Ay}, {Az,delta}, {delta,delta} of two link x link planes; row stride Qs
    const size_t Qs = (size_t)armour_row_stride(Q);
    const double* pl0 = tb.planes + (size_t)b * armour_planes_per_problem(Q);
    const v2d* plAB = reinterpret_cast<const v2d*>(pl0) + q;
    const v2d* plCD = reinterpret_cast<const v2d*>(pl0 + armour_planes_cd_offset(Q)) + q;
    const v2d* plLL = reinterpret_cast<const v2d*>(pl0 + armour_planes_dll_offset(Q)) + q;
    const double* plD = pl0 + armour_planes_d_offset(Q) + q;
    const int JT = tb.J * T;
    const int q_lt = fast_div(q, lp.magic_O), q_o = q - q_lt * O;  // q = (l*T + t)*O + o
    // compact link x link normals: one 384-B record per (link, time step), entry 3*(p - 21) + c
    const double* pll = tb.planes_ll + (size_t)b * armour_planes_ll_per_problem(JT) + (size_t)q_lt * ARMOUR_LL_RECORD;
    unsigned long long live = ~(lp.skip_by_value ? lp.skip0 : tb.plane_skip[b]) & ((1ull << ARMOUR_NPLANES) - 1ull);
    const bool plane0_live = (live & 1ull) != 0;
    const int na = __popcll(live), base = na >> 2, rem = na & 3;
    const int my_cnt = base + (wv < rem ? 1 : 0), my_start = wv * base + min(wv, rem);
    for (int s = 0; s < my_start; s++) live &= live - 1ull;
    double a0[PPW], a1[PPW], a2[PPW], dd[PPW], dl[PPW];
    const int lt_first = q_begin / O;
    const int npairs = (q_end - 1) / O - lt_first + 1;
    const bool single_pass = npairs <= lp.pair_chunk;
    PassRegs pr;
    if (EX) load_pass_uncond(tb, lp, b, lt_first, npairs, pr);
    // EX launches (armour_p2_plan: ex_layout_ok): waves 0 and 1 hold six planes with an obstacle generator each -- {Ax,Ay}, {Az,delta}:
    // twelve 16-B loads; waves 2 and 3 six CONSECUTIVE link x link planes each, starting at an even p - 21: their 18 normal components
    // are 144 contiguous, 16-B aligned bytes of the (link, time step) record -- nine loads whose 64 lanes touch 64/O + 1 records -- and
    // their deltas three 16-B pairs.  Twelve loads [+ six of d] either way, issued by common code from per-wave ADDRESSES into the same
    // registers, so that the compiler's vmcnt for the slicing below is exact; what the registers mean is sorted out after the slicing.
    v2d raw[EX ? 12 : 1];
    bool ll_wave = false;
    if (EX) {
        static_assert(!EX || PPW == 6, "the EX kernels hold six planes per wave");
        ll_wave = wv >= 2;
        const v2d* ra[12];
        const double* da[6];
        if (ll_wave) {
            const int pll0 = __builtin_ctzll(live) - ARMOUR_FIRST_LL_PLANE;
            const v2d* nrm = reinterpret_cast<const v2d*>(pll + (size_t)pll0 * 3);
#pragma unroll
            for (int j = 0; j < 9; j++) ra[j] = nrm + j;
#pragma unroll
            for (int j = 0; j < 3; j++) ra[9 + j] = plLL + (size_t)((pll0 >> 1) + j) * Qs;
#pragma unroll
            for (int i = 0; i < 6; i++) da[i] = plD + (size_t)(ARMOUR_FIRST_LL_PLANE + pll0 + i) * Qs;
        } else {
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const size_t o = (size_t)__builtin_ctzll(live) * Qs;
                live &= live - 1ull;
                ra[2 * i] = plAB + o; ra[2 * i + 1] = plCD + o; da[i] = plD + o;
            }
        }
#pragma unroll
        for (int j = 0; j < 12; j++) raw[j] = *ra[j];
#pragma unroll
        for (int i = 0; i < 6; i++) { dd[i] = 0.0; if (!DFC) dd[i] = *da[i]; }
    } else {
#pragma unroll
    for (int i = 0; i < PPW; i++) {
        a0[i] = 0.0; a1[i] = 0.0; a2[i] = 0.0; dd[i] = 0.0; dl[i] = 0.0;
        if (i < my_cnt) {
            const int pidx = __builtin_ctzll(live);
            const size_t o = (size_t)pidx * Qs;
            live &= live - 1ull;
            // (non-temporal loads were measured 7 % slower at B=128, O=50: default cache policy kept)
            if (LL && pidx >= ARMOUR_FIRST_LL_PLANE) {
                // link x link plane: its normal is the same for the O obstacles of a (link, time step) and is read from
                // the compact copy -- a wave touches 64/O + 1 distinct records instead of 64 rows
                const double* al = pll + (size_t)(pidx - ARMOUR_FIRST_LL_PLANE) * 3;
                a0[i] = al[0]; a1[i] = al[1]; a2[i] = al[2];
                dl[i] = pl0[armour_plane_index(Q, q, pidx, 4)];
            } else {
                const v2d ab = plAB[o], cd = plCD[o];
                a0[i] = ab.x; a1[i] = ab.y; a2[i] = cd.x; dl[i] = cd.y;
                if (!LL && pidx >= ARMOUR_FIRST_LL_PLANE) dl[i] = pl0[armour_plane_index(Q, q, pidx, 4)];
            }
            if (!DFC) dd[i] = plD[o];
        }
#if defined(P2_ABLATE) && (P2_ABLATE & 8)
        a0[i] = 1.0 + i; a1[i] = 0.5; a2[i] = 0.25; dd[i] = 0.1; dl[i] = 0.2;
#endif
    }
    }
    // d = A . c_obstacle (RT/CollisionChecking.cu:200-202): with the obstacle centres at hand it is recomputed, in the
    // expression of armour_p1_planes_kernel, instead of read -- 8 B less per plane and row
    double oc0 = 0.0, oc1 = 0.0, oc2 = 0.0;
    if (DFC) {
        const double* oc = tb.obs_center + (size_t)b * 3 * O + q_o;
        oc0 = oc[0]; oc1 = oc[O]; oc2 = oc[2 * (size_t)O];
    }
    P2_STAMP(1);
    // 2. per point: slice the (l,t) link PZs this block's rows touch (one thread per (pair, monomial, axis)), then
    //    scan the planes.  The link-PZ table entries do not depend on k: when one pass covers all pairs they are
    //    loaded once, before the k-power table of the first point is waited for, so the fresh k (an HBM miss), the
    //    PZ tables (L2) and the planes are all in flight together.
    const int per_pair = lp.strideL * P2_SL;
    if (MULTI && single_pass) load_pass(tb, lp, b, lt_first, 0, npairs, pr);
    const double* xs = sx + (q_lt - lt_first) * P2_SL;
    for (int s = 0; s < nsteps; s++) {
        KPow& kp = kp2[s & 1];
        const double k_cur = k_next;
        if (tid < n && s + 1 < nsteps) k_next = k0[(size_t)(s + 1) * lp.k_stride + tid];
        fill_kpow(kp, k_cur, n);
        double* g = WANT_G ? g0 + (size_t)s * lp.g_stride : nullptr;
        double* jac = WANT_J ? jac0 + (size_t)s * lp.j_stride : nullptr;
        for (int p0 = 0; p0 < npairs; p0 += lp.pair_chunk) {
            const int pc = min(lp.pair_chunk, npairs - p0);
            if (!EX && (!MULTI || !single_pass)) load_pass(tb, lp, b, lt_first, p0, pc, pr);
            if (EX) lds_barrier(); else __syncthreads();  // k-power table ready / previous pass or point done with `terms`
            P2_STAMP(2);
#pragma unroll
            for (int r = 0; r < P2_TASK_ROUNDS; r++) {
                if (pr.tcnt[r] > 0) {
                    double o8[P2_TQW];
                    mono_all<WANT_J>(kp, pr.tkey[r], pr.tco[r], n, o8);
                    double* dst = terms + pr.tdst[r];
                    dst[0] = o8[0];
                    if (WANT_J) {
#pragma unroll
                        for (int kk = 0; kk < ARMOUR_MAX_FACTORS; kk++) dst[3 + kk * 3] = o8[1 + kk];
                    }
                }
            }
            if (EX) lds_barrier(); else __syncthreads();
            P2_STAMP(3);
            // ordered sum over monomials (the reference's accumulation order, RT/PZsparse.cu:420,470-472)
            if (tid < pc * P2_SL) {
                const int pi = tid / P2_SL, c = tid - pi * P2_SL, out = c / 3, e2 = c - out * 3;
                double acc = (!EX || out == 0) ? pr.rc_cen : 0.0;  // (the unconditional loader fills rc_cen for every column)
                const double* tp = terms + (size_t)pi * per_pair + c;
#pragma unroll 4
                for (int mo = 0; mo < pr.rc_cnt; mo++) acc += tp[mo * P2_SL];
                if (out == 0) acc = interval_center(acc, pr.rc_ind);
                sx[(p0 + pi) * P2_SL + (out == 0 ? e2 : 3 + (out - 1) * 3 + e2)] = acc;
            }
            if (EX) lds_barrier(); else __syncthreads();
        }
        P2_STAMP(4);
        // 3. this wave's planes in the reference's scan order (pos_p before neg_p, strict >), branch-free: `best` is
        //    2*slot + (1 if the negative side won); the winner's normal is picked out of the registers afterwards.
        //    max_id defaults to plane 0 (RT/CollisionChecking.cu:262): its normal if it is live (then it is slot 0 of
        //    wave 0), zero if it was skipped (best = -2 matches no slot).
        const double x0 = xs[0], x1 = xs[1], x2 = xs[2];
        if (EX) {  // what the raw 16-B registers of this wave hold (see the loads above)
            if (ll_wave) {
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    const int e = 3 * i;  // components e, e + 1, e + 2 of the 18 consecutive normal components
                    a0[i] = (e & 1) ? raw[e >> 1].y : raw[e >> 1].x;
                    a1[i] = ((e + 1) & 1) ? raw[(e + 1) >> 1].y : raw[(e + 1) >> 1].x;
                    a2[i] = ((e + 2) & 1) ? raw[(e + 2) >> 1].y : raw[(e + 2) >> 1].x;
                    dl[i] = (i & 1) ? raw[9 + (i >> 1)].y : raw[9 + (i >> 1)].x;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 6; i++) { a0[i] = raw[2 * i].x; a1[i] = raw[2 * i].y; a2[i] = raw[2 * i + 1].x; dl[i] = raw[2 * i + 1].y; }
            }
        }
        if (DFC) {
#pragma unroll
            for (int i = 0; i < PPW; i++) dd[i] = a0[i] * oc0 + a1[i] * oc1 + a2[i] * oc2;
        }
        double max_elt = -100000000.0;
        int best = (wv == 0 && plane0_live) ? 0 : -2;
#pragma unroll
        for (int i = 0; i < PPW; i++) {
            const bool nz = (a0[i] != 0.0) | (a1[i] != 0.0) | (a2[i] != 0.0);  // A_elt.norm() > 0 (RT/CollisionChecking.cu:252)
            const double dot = a0[i] * x0 + a1[i] * x1 + a2[i] * x2;
            const double pos_res = nz ? dot - (dd[i] + dl[i]) : -100000000.0;
            const double neg_res = nz ? -dot - (-dd[i] + dl[i]) : -100000000.0;
            const bool c1 = pos_res > max_elt;
            max_elt = c1 ? pos_res : max_elt; best = c1 ? 2 * i : best;
            const bool c2 = neg_res > max_elt;
            max_elt = c2 ? neg_res : max_elt; best = c2 ? 2 * i + 1 : best;
        }
        {
            double mA0 = 0.0, mA1 = 0.0, mA2 = 0.0;
            const int bs = best >> 1;
#pragma unroll
            for (int i = 0; i < PPW; i++) {
                const bool hit = bs == i;
                mA0 = hit ? a0[i] : mA0; mA1 = hit ? a1[i] : mA1; mA2 = hit ? a2[i] : mA2;
            }
            double* my = part + ((size_t)wv * 64 + lane) * 4;
            my[0] = max_elt; my[1] = mA0; my[2] = mA1; my[3] = mA2;
            pneg[wv * 64 + lane] = best & 1;  // (-2 & 1) = 0
        }
        __syncthreads();
        P2_STAMP(5);
        // 4. every wave merges the four partial scans in plane order (the reference's serial winner), then takes its
        //    share of the row's outputs: wave 0 the value, wave w the Jacobian columns w and w + 4
        {
            const double* c0 = part + (size_t)lane * 4;
            double m_elt = c0[0], mA0 = c0[1], mA1 = c0[2], mA2 = c0[3];
            int neg = pneg[lane];
#pragma unroll
            for (int w2 = 1; w2 < 4; w2++) {
                const double* o = part + ((size_t)w2 * 64 + lane) * 4;
                const double oe = o[0], o1 = o[1], o2 = o[2], o3 = o[3];
                const int on = pneg[w2 * 64 + lane];
                const bool c = oe > m_elt;
                m_elt = c ? oe : m_elt; mA0 = c ? o1 : mA0; mA1 = c ? o2 : mA1; mA2 = c ? o3 : mA2; neg = c ? on : neg;
            }
#if defined(P2_ABLATE) && (P2_ABLATE & 32)
            if (WANT_G && wv == 0 && m_elt == 12345.678) g[(size_t)tb.row0 + q_begin + lane] = -m_elt;
#else
            if (WANT_G && wv == 0 && q_begin + lane < q_end) g[(size_t)tb.row0 + q_begin + lane] = -m_elt;
#endif
            if (WANT_J) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int kk = wv + 4 * h;
                    if (kk < n) {
                        const double* dx = xs + 3 + kk * 3;
                        const double dot = mA0 * dx[0] + mA1 * dx[1] + mA2 * dx[2];
                        stage[lane * n + kk] = neg ? dot : -dot;
                    }
                }
            }
        }
        if (WANT_J) {
            __syncthreads();
            const int total = (q_end - q_begin) * n;
            double* jrow = jac + ((size_t)tb.row0 + q_begin) * n;
#if defined(P2_ABLATE) && (P2_ABLATE & 32)
            for (int i = tid; i < total; i += P2_BLOCK) if (stage[i] == 12345.678) jrow[i] = stage[i];
#else
            for (int i = tid; i < total; i += P2_BLOCK) jrow[i] = stage[i];
#endif
        }
    }
#ifdef P2_TIMELINE
    P2_STAMP(6);
    stamp[7] = __builtin_amdgcn_s_memtime() - cyc0;
    if (tid == 0 && (role == 0 || role == lp.nbc / 2 || role == lp.nbc - 1)) {
        const int slot = role == 0 ? 0 : role == lp.nbc - 1 ? 2 : 1;
        for (int i2 = 0; i2 < 8; i2++) g0_in[(size_t)tb.row0 + Q + slot * 8 + i2] = (double)(stamp[i2] & 0xffffffffffffull);
    }
#endif
}

template <bool WANT_G, bool WANT_J, bool MULTI, bool DFC, bool LL, int PPW, bool EX>
__device__ __forceinline__ void torque_block(const P2Tables& tb, const P2Launch& lp, const int b, const int role, const double* __restrict__ k0,
                                   const double k_first, double* __restrict__ g0_in, double* __restrict__ jac0_in, unsigned char* smem_raw) {
    KPow* kp2 = reinterpret_cast<KPow*>(smem_raw);  // double-buffered over the points
    double* lds = reinterpret_cast<double*>(smem_raw + 2 * sizeof(KPow));
    const int n = tb.n, T = tb.T, O = tb.O, Q = tb.Q, m = tb.m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave index, known uniform: the plane dealing below runs on the scalar unit
    // k0 / g0_in / jac0_in: THIS problem's k [n], g [m] and jac [m][n] (of the first point for multi-point launches);
    // k_first: this thread's component of the first point (tid < n), fetched by the caller as early as it can -- a fresh k
    // is an HBM miss, and at B = 1 the launch waits for it
    double* g0 = WANT_G ? g0_in : nullptr;
    double* jac0 = WANT_J ? jac0_in : nullptr;
    const int nsteps = MULTI ? lp.steps : 1;  // the one-point instantiation keeps the loop-free code of a single launch
    // this thread's component of the NEXT point's k is fetched one point ahead: a fresh k is an HBM miss
    double k_next = k_first;
    (void)lane; (void)wv; (void)T; (void)O; (void)Q; (void)m; (void)kp2; (void)lds; (void)g0; (void)jac0; (void)nsteps; (void)k_next;
    // ------------------------------------------------------------------ torque rows (row = t*n + j)
    double* terms = lds;  // [P2_TQ_ROWS][strideT][P2_TQW]
    const int r = tid >> 5, ml = tid & 31;
    const int row = (role - lp.nbc) * P2_TQ_ROWS + r;
    const bool live = row < n * T;
    int t = 0, j = 0, cnt = 0;
    size_t idx = 0;
    uint32_t tkey[P2_TQ_ROUNDS];
    double tco[P2_TQ_ROUNDS];
    double cen0 = 0.0, ind0 = 0.0;
    if (live) {
        t = row / n; j = row - t * n;
        idx = ((size_t)b * n + j) * T + t;
        cnt = min(tb.tq_count[idx], lp.strideT);
        if (ml == 0) { cen0 = tb.tq_center[idx]; ind0 = tb.tq_indep[idx]; }
#pragma unroll
        for (int rr = 0; rr < P2_TQ_ROUNDS; rr++) {
            const int mo = ml + rr * 32;
            tkey[rr] = 0; tco[rr] = 0.0;
            if (mo < lp.strideT && mo < tb.capT) { tkey[rr] = tb.tq_keys[idx * tb.capT + mo]; tco[rr] = tb.tq_coeff[idx * tb.capT + mo]; }
        }
    }
    for (int s = 0; s < nsteps; s++) {
        KPow& kp = kp2[s & 1];
        const double k_cur = k_next;
        if (tid < n && s + 1 < nsteps) k_next = k0[(size_t)(s + 1) * lp.k_stride + tid];
        fill_kpow(kp, k_cur, n);
        double* g = WANT_G ? g0 + (size_t)s * lp.g_stride : nullptr;
        double* jac = WANT_J ? jac0 + (size_t)s * lp.j_stride : nullptr;
        __syncthreads();  // k-power table ready; every thread is done with the previous point's `terms`
        if (live) {
#pragma unroll
            for (int rr = 0; rr < P2_TQ_ROUNDS; rr++) {
                const int mo = ml + rr * 32;
                if (mo < cnt) {
                    double* tp = terms + ((size_t)r * lp.strideT + mo) * P2_TQW;
                    double o8[P2_TQW];
                    mono_all<WANT_J>(kp, tkey[rr], tco[rr], n, o8);
                    tp[0] = o8[0];
                    if (WANT_J) {
#pragma unroll
                        for (int kk = 0; kk < ARMOUR_MAX_FACTORS; kk++) tp[1 + kk] = o8[1 + kk];
                    }
                }
            }
        }
        __syncthreads();
        if (live && ml <= n) {
            const double* tp = terms + (size_t)r * lp.strideT * P2_TQW + ml;
            if (ml == 0) {
                if (WANT_G) {
                    double cen = cen0;
#pragma unroll 4
                    for (int mo = 0; mo < cnt; mo++) cen += tp[mo * P2_TQW];
                    g[row] = interval_center(cen, ind0);
                }
            } else if (WANT_J) {
                double gr = 0.0;
#pragma unroll 4
                for (int mo = 0; mo < cnt; mo++) gr += tp[mo * P2_TQW];
                jac[(size_t)row * n + (ml - 1)] = gr;
            }
        }
    }
}

template <bool WANT_G, bool WANT_J, bool MULTI, bool DFC, bool LL, int PPW, bool EX>
__device__ __forceinline__ void limit_block(const P2Tables& tb, const P2Launch& lp, const int b, const int role, const double* __restrict__ k0,
                                   const double k_first, double* __restrict__ g0_in, double* __restrict__ jac0_in, unsigned char* smem_raw) {
    KPow* kp2 = reinterpret_cast<KPow*>(smem_raw);  // double-buffered over the points
    double* lds = reinterpret_cast<double*>(smem_raw + 2 * sizeof(KPow));
    const int n = tb.n, T = tb.T, O = tb.O, Q = tb.Q, m = tb.m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave index, known uniform: the plane dealing below runs on the scalar unit
    // k0 / g0_in / jac0_in: THIS problem's k [n], g [m] and jac [m][n] (of the first point for multi-point launches);
    // k_first: this thread's component of the first point (tid < n), fetched by the caller as early as it can -- a fresh k
    // is an HBM miss, and at B = 1 the launch waits for it
    double* g0 = WANT_G ? g0_in : nullptr;
    double* jac0 = WANT_J ? jac0_in : nullptr;
    const int nsteps = MULTI ? lp.steps : 1;  // the one-point instantiation keeps the loop-free code of a single launch
    // this thread's component of the NEXT point's k is fetched one point ahead: a fresh k is an HBM miss
    double k_next = k_first;
    (void)lane; (void)wv; (void)T; (void)O; (void)Q; (void)m; (void)kp2; (void)lds; (void)g0; (void)jac0; (void)nsteps; (void)k_next;
    // ------------------------------------------------------------------ joint limit rows
    // RT/Trajectory.cu:256-540.  One thread per (joint, position|velocity, piece): pieces 0-3 evaluate the curve
    // at t = 0, the two stationary points and t = 1, pieces 4-5 the d/dk of the two interior extrema; a second
    // step per (joint, position|velocity) selects min / max exactly as bez::joint_extremum does.
    double* pv = lds;  // [2n][8]
    const int jv = tid >> 3, piece = tid & 7;
    const double* bz = tb.bez + (size_t)b * 3 * n;
    if (tb.mode == ARMOUR_MODE_ARMTD) {
        // ARMTD comparison mode: constant-acceleration curve, CMP/Trajectory.cu:83-383 (cacc.h); bez = q0, qd0, k_range.
        // One thread per joint; rows q_min, q_max, qd_min, qd_max (n each), Jacobian diagonal = d/d(k_range*k) as the
        // reference stores it, every other entry of the rows 0.
        for (int s = 0; s < nsteps; s++) {
            const double* k = k0 + (size_t)s * lp.k_stride;
            double* g = WANT_G ? g0 + (size_t)s * lp.g_stride : nullptr;
            double* jac = WANT_J ? jac0 + (size_t)s * lp.j_stride : nullptr;
            if (tid < n) {
                const cacc::Extrema e = cacc::joint_extrema(bz[tid], bz[n + tid], bz[2 * n + tid] * k[tid]);
                const cacc::Cand c4[4] = {e.q_min, e.q_max, e.qd_min, e.qd_max};
                const size_t off = (size_t)tb.row0 + Q;
                for (int r = 0; r < 4; r++) {
                    const size_t row = off + (size_t)r * n + tid;
                    if (WANT_G) g[row] = c4[r].v;
                    if (WANT_J) for (int c = 0; c < n; c++) jac[row * n + c] = (c == tid) ? c4[r].d : 0.0;
                }
            }
        }
        return;
    }
    for (int s = 0; s < nsteps; s++) {
        const double* k = k0 + (size_t)s * lp.k_stride;
        double* g = WANT_G ? g0 + (size_t)s * lp.g_stride : nullptr;
        double* jac = WANT_J ? jac0 + (size_t)s * lp.j_stride : nullptr;
        if (jv < 2 * n && piece < 6) {
            const int i = jv % n;
            const bool vel = jv >= n;
            const double q0 = bz[i], a = bz[n + i], bb = bz[2 * n + i], ka = tb.k_range[i] * k[i];
            double e2, e3, v;
            if (!vel) bez::q_stationary(a, bb, ka, &e2, &e3); else bez::qd_stationary(a, bb, ka, &e2, &e3);
            if (piece < 4) {
                const double tt = piece == 0 ? 0.0 : piece == 1 ? e2 : piece == 2 ? e3 : 1.0;
                v = vel ? bez::qd_des(q0, a, bb, ka, tt) : bez::q_des(q0, a, bb, ka, tt);
            } else {
                const int sg = piece == 4 ? +1 : -1;
                v = vel ? bez::qd_extremum_dk(q0, a, bb, ka, sg) : bez::q_extremum_dk(q0, a, bb, ka, sg);
            }
            pv[jv * 8 + piece] = v;
            if (piece == 1) pv[jv * 8 + 6] = e2;
            if (piece == 2) pv[jv * 8 + 7] = e3;
        }
        __syncthreads();
        if (tid < 2 * n) {
            const int i = tid % n;
            const bool vel = tid >= n;
            const double* v = pv + tid * 8;
            const double v1 = v[0], v2 = v[1], v3 = v[2], v4 = v[3], e2 = v[6], e3 = v[7];
            double mn, mx;
            int mnId, mxId;
            if (v1 < v4) { mn = v1; mnId = 1; mx = v4; mxId = 4; } else { mn = v4; mnId = 4; mx = v1; mxId = 1; }
            if (0 <= e2 && e2 <= 1) { if (v2 < mn) { mn = v2; mnId = 2; } if (mx < v2) { mx = v2; mxId = 2; } }
            if (0 <= e3 && e3 <= 1) { if (v3 < mn) { mn = v3; mnId = 3; } if (mx < v3) { mx = v3; mxId = 3; } }
            const double sc = vel ? tb.k_range[i] / tb.duration : tb.k_range[i];
            const double dmn = (mnId == 1 ? 0.0 : mnId == 2 ? v[4] : mnId == 3 ? v[5] : 1.0) * sc;
            const double dmx = (mxId == 1 ? 0.0 : mxId == 2 ? v[4] : mxId == 3 ? v[5] : 1.0) * sc;
            const size_t off = (size_t)tb.row0 + Q;
            const size_t r_mn = off + (vel ? 2 * n : 0) + i, r_mx = r_mn + n;
            if (WANT_G) { g[r_mn] = vel ? mn / tb.duration : mn; g[r_mx] = vel ? mx / tb.duration : mx; }
            if (WANT_J) {
                for (int c = 0; c < n; c++) {
                    jac[r_mn * n + c] = (c == i) ? dmn : 0.0;
                    jac[r_mx * n + c] = (c == i) ? dmx : 0.0;
                }
            }
        }
        __syncthreads();  // `pv` is rewritten by the next point
    }
}

}  // namespace p2
