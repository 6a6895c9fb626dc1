// armour_solve, device-resident form: the whole SQP iterate of RT/armour_main.cu:237-304 (OptimizeTNLP) and
// armtd_NLP::finalize_solution (RT/NLPclass.cu:422-538) in ONE persistent kernel launch -- SURVEY.md 8f rank 1.
//
// The host-driven form (solver.hip) pays a launch, a scan launch and a host wake-up per evaluation (4 evaluations of the
// reference's sample problem: ~0.3 ms, of which the GPU works ~20 us) and solves the 7-variable QPs on the host.  Here:
//
//   grid   = B groups of `nb` blocks (all co-resident: cooperative launch); group b owns problem b, block j of a group owns a
//            fixed, contiguous range of the problem's row tiles -- torque tiles, collision tiles, limit rows, in ROW ORDER -- for
//            the whole solve.  The same tile always runs on the same CU, so after the first evaluation its share of the plane
//            / PZ tables (12.6 MB at configs[1], 1.6 MB per XCD) is served by that XCD's L2 instead of HBM.
//   phase  = one evaluation.  Every block evaluates its tiles at the point x the leader published (the tile code of
//            p2_tiles.h, i.e. bit-identical rows to armour_eval_g_jac), scans its own rows -- L1 violation in fixed point
//            (order-independent, solver_common.h), candidate rows of the QP compacted in row order into the block's slot, or the
//            finalize_solution verdict -- and arrives at the group's barrier (one atomic per block).
//   leader = block 0 of the group.  After the barrier it gathers the candidate rows in block (= row) order, runs the
//            Goldfarb-Idnani QP (<= 7 active rows; the search for the most violated row uses the whole block, the 7 x 7
//            algebra runs on one lane with the active normals in LDS), advances the SQP / line-search state machine and
//            publishes the next point.  The other blocks of the group poll one word meanwhile.
//
// The arithmetic of the QP, of the merit function and of the cost is the host form's, operation for operation, on the same
// candidate rows in the same order: both forms produce the same iterates (tests/test_solve.py compares k_opt bit for bit).
// Problems of a batch no longer run in lock step: each group proceeds, converges and leaves on its own.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>

#include "bezier.h"
#include "cacc.h"
#include "common.h"
#include "p2_tiles.h"
#include "solver_common.h"
#include "solver_device.h"

using namespace p2;
using namespace slv;

namespace {

enum { CMD_EVAL = 1, CMD_DONE = 4 };
enum { ST_AFTER_FIRST = 0, ST_AFTER_TRIAL = 1, ST_AFTER_RELIN = 2, ST_AFTER_VERDICT = 3 };
constexpr double kInf = 1e300;
// lanes of ONE wave exchanging data through LDS: its LDS instructions execute in order, so only the compiler has to be held
#define WAVE_LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// Communication between the blocks of a group.  MI355X has eight XCDs with an L2 each; L2s are not coherent with one another for
// ordinary accesses, so an agent-scope release / acquire FENCE writes back / invalidates a whole L2 -- tens of microseconds per
// phase when first tried here, and it throws away the very tables this kernel wants to keep cached.  Instead every word that
// crosses blocks (the point x, the command, the candidate rows and their counts) is written and read with RELAXED agent-scope
// atomic accesses -- the compiler emits them as coherent (sc1) loads / stores that go to the memory side and touch no other cache
// line -- and ordering is by completion: a block waits for its own stores (s_waitcnt vmcnt(0)) before it signals.
__device__ inline unsigned ld_coh(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline int ld_coh(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline long long ld_coh(const long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline double ld_coh(const double* p) { return __longlong_as_double(__hip_atomic_load(reinterpret_cast<const long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ inline void st_coh(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_coh(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_coh(long long* p, long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_coh(double* p, double v) { __hip_atomic_store(reinterpret_cast<long long*>(p), __double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void wait_my_memory_ops() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
// a candidate row, word by word (9 x 8 bytes)
__device__ inline void st_coh_row(SolveRow* dst, const SolveRow& r) {
    long long* d = reinterpret_cast<long long*>(dst);
    st_coh(d, ((long long)(unsigned)r.side << 32) | (unsigned)r.idx);
    st_coh(reinterpret_cast<double*>(d + 1), r.v);
#pragma unroll
    for (int j = 0; j < NV; j++) st_coh(reinterpret_cast<double*>(d + 2 + j), r.a[j]);
}
__device__ inline SolveRow ld_coh_row(const SolveRow* src) {
    const long long* d = reinterpret_cast<const long long*>(src);
    SolveRow r;
    const long long w0 = ld_coh(d);
    r.idx = (int)(unsigned)(w0 & 0xffffffffll); r.side = (int)(w0 >> 32);
    r.v = ld_coh(reinterpret_cast<const double*>(d + 1));
#pragma unroll
    for (int j = 0; j < NV; j++) r.a[j] = ld_coh(reinterpret_cast<const double*>(d + 2 + j));
    return r;
}

// tile t of a problem in row order -> the role index the tile functions use (collision tiles first) and its rows [r0, r1)
__device__ inline void tile_info(const SolveArgs& a, int t, int& role, int& r0, int& r1) {
    const int nbt = a.lp.nbt, nbc = a.lp.nbc, row0 = a.tb.row0;
    if (t < nbt) { role = nbc + t; r0 = t * P2_TQ_ROWS; r1 = min(row0, r0 + P2_TQ_ROWS); }
    else if (t < nbt + nbc) { role = t - nbt; r0 = row0 + (t - nbt) * P2_ROWS; r1 = min(row0 + a.tb.Q, r0 + P2_ROWS); }
    else { role = nbc + nbt; r0 = row0 + a.tb.Q; r1 = a.tb.m; }
}

__device__ inline double wrap_to_pi_dev(double x) {  // RT/NLPclass.cu:6-15
    const double pi = 3.14159265358979323846;
    while (x < -pi) x += 2 * pi;
    while (x > pi) x -= 2 * pi;
    return x;
}
// eval_f / eval_grad_f (RT/NLPclass.cu:207-267, CMP/NLPclass.cu:183-243): the expressions of api.hip armour_eval_f / _grad_f
__device__ inline double plan_point(const SolveArgs& a, const double* bz, int i, double k) {
    const int n = a.tb.n;
    return a.tb.mode == ARMOUR_MODE_ARMTD ? cacc::q_plan(bz[i], bz[n + i], bz[2 * n + i], k)
                                          : bez::q_des(bz[i], bz[n + i], bz[2 * n + i], a.tb.k_range[i] * k, a.t_plan);
}
// eval_f from the joints' squared errors, added in armour_eval_f's order (continuous joints first)
__device__ inline double cost_sum(const SolveArgs& a, const double* sq) {
    double obj = 0;
    for (int pass = 0; pass < 2; pass++)
        for (int i = 0; i < a.tb.n; i++)
            if ((((a.continuous_mask >> i) & 1) != 0) == (pass == 0)) obj += sq[i];
    return obj * a.cost_scale;
}
__device__ inline double plan_dk(const SolveArgs& a, const double* bz, int i) {
    const double tp = a.t_plan;
    return a.tb.mode == ARMOUR_MODE_ARMTD ? cacc::q_plan_dk(bz[2 * a.tb.n + i]) : (tp * tp * tp) * (6 * tp * tp - 15 * tp + 10) * a.tb.k_range[i];
}
// Block-wide integer sums and scans with one barrier instead of a tree of eight: inside each wave by shuffles, the four wave results through LDS.
// (Integer addition: any order gives the same number.)  `sh4` / `si4`: LDS arrays of >= 4 entries that nobody else is using.
__device__ inline void block_sum(long long& v, int& i, long long* sh4, int* si4) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { v += __shfl_xor(v, off, 64); i += __shfl_xor(i, off, 64); }
    __syncthreads();   // (the arrays may still be read from an earlier call)
    if ((threadIdx.x & 63) == 0) { sh4[threadIdx.x >> 6] = v; si4[threadIdx.x >> 6] = i; }
    __syncthreads();
    v = sh4[0] + sh4[1] + sh4[2] + sh4[3];
    i = si4[0] + si4[1] + si4[2] + si4[3];
}
// exclusive prefix sum of one int per thread over the block (and the block total)
__device__ inline int block_exclusive_scan(int v, int* si4, int& total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(inc, off, 64); if (lane >= off) inc += o; }
    __syncthreads();
    if (lane == 63) si4[wv] = inc;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) if (w < wv) base += si4[w];
    total = si4[0] + si4[1] + si4[2] + si4[3];
    return base + inc - v;
}

// ------------------------------------------------------------------------------------------------ per-block scan
// The block's rows after an evaluation: their L1 violation (fixed point), the candidate rows of the QP compacted in row order
// into `rows`, and the number of rows outside [g_l - slack, g_u + slack] (finalize_solution).  The row tests are the host
// form's (solver_common.h; MODE is kept for the two tests' placement only: every phase runs 3 = all of it).
struct ScanShared {
    int wave_tot[4];
    long long red[256];
    int redi[256];
};
template <int MODE>
__device__ inline void scan_rows(const SolveArgs& a, int b, int r0, int r1, SolveRow* __restrict__ rows, int cap, ScanShared& sh,
                                 long long& viol_out, int& count_out, int& bad_out) {
    const int n = a.tb.n, m = a.tb.m, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double* g = a.g + (size_t)b * m;
    const double* jac = a.jac + (size_t)b * m * n;
    const double* lo = a.lo + (size_t)b * m;
    const double* hi = a.hi + (size_t)b * m;
    const int n_unchecked = a.tb.Q - a.n_checked_collision;
    int base = 0, bad = 0;
    long long vsum = 0;
    for (int i0 = r0; i0 < r1; i0 += 256) {
        const int i = i0 + tid;
        const bool in = i < r1;
        const double gi = in ? g[i] : 0.0, li = in ? lo[i] : -1e300, ui = in ? hi[i] : 1e300;
        vsum += row_violation(gi, li, ui);
        if ((MODE & 2) && in) {
            const int ic = i - a.tb.row0 - a.n_checked_collision;  // >= 0: behind the re-checked collision rows
            const double slack = i < a.tb.row0 ? a.torque_slack : ic < 0 ? a.collision_slack : 0.0;
            if ((ic < 0 || ic >= n_unchecked) && (gi < li - slack || gi > ui + slack)) bad++;
        }
        if (MODE & 1) {
            double J[NV], l1 = 0.0;
#pragma unroll
            for (int j = 0; j < NV; j++) { J[j] = (in && j < n) ? jac[(size_t)i * n + j] : 0.0; l1 += fabs(J[j]); }
            const bool fh = in && row_upper_candidate(gi, ui, l1);
            const bool fl = in && row_lower_candidate(gi, li, l1);
            const unsigned long long bh = __ballot(fh), bl = __ballot(fl);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (lane == 0) sh.wave_tot[wv] = __popcll(bh) + __popcll(bl);
            __syncthreads();
            int pos = base + __popcll(bh & below) + __popcll(bl & below);
            for (int w2 = 0; w2 < wv; w2++) pos += sh.wave_tot[w2];
            if (fh) {
                if (pos < cap) {
                    SolveRow r; r.idx = i; r.side = 0; r.v = gi - ui;
#pragma unroll
                    for (int j = 0; j < NV; j++) r.a[j] = -J[j];
                    st_coh_row(rows + pos, r);
                }
                pos++;
            }
            if (fl && pos < cap) {
                SolveRow r; r.idx = i; r.side = 1; r.v = li - gi;
#pragma unroll
                for (int j = 0; j < NV; j++) r.a[j] = J[j];
                st_coh_row(rows + pos, r);
            }
            base += sh.wave_tot[0] + sh.wave_tot[1] + sh.wave_tot[2] + sh.wave_tot[3];
            __syncthreads();
        }
    }
    block_sum(vsum, bad, sh.red, sh.redi);
    viol_out = vsum;
    bad_out = bad;
    count_out = base;
    __syncthreads();
}

// The culled form's scan: the block's rows are those of its tiles [t0, t1) of the problem's LISTED rows -- tiles of P2_BLOCK listed torque rows, then
// tiles of P2_BLOCK listed collision rows, then the limit rows -- visited tile by tile in row order, one thread per row of a tile; the tests and the
// compaction are scan_rows' (candidates of the block in ascending row order).
struct TileSet { int ntq, nsp, cnt2, cntq; };   // tiles of listed torque rows, tiles of listed collision rows, listed collision rows, listed torque rows
__device__ inline TileSet problem_tiles(const SolveArgs& a, int b) {
    TileSet ts;
    ts.cntq = a.tq_count[b]; ts.ntq = (ts.cntq + P2_BLOCK - 1) / P2_BLOCK;
    ts.cnt2 = a.sl.count[b]; ts.nsp = (ts.cnt2 + P2_BLOCK - 1) / P2_BLOCK;
    return ts;
}
// row of thread `tid` in listed tile t (or -1)
__device__ inline int tile_row(const SolveArgs& a, int b, const TileSet& ts, int t, int tid) {
    if (t < ts.ntq) {
        const int i = t * P2_BLOCK + tid;
        return i < ts.cntq ? a.tq_tiles[(size_t)b * a.tq_cap + i] : -1;
    }
    if (t < ts.ntq + ts.nsp) {
        const int i = (t - ts.ntq) * P2_BLOCK + tid;
        return i < ts.cnt2 ? a.tb.row0 + a.sl.rows[(size_t)b * a.tb.Q + i] : -1;
    }
    const int r = a.tb.row0 + a.tb.Q + tid;
    return r < a.tb.m ? r : -1;
}
__device__ inline void scan_tiles(const SolveArgs& a, int b, const TileSet& ts, int t0, int t1, SolveRow* __restrict__ rows, int cap, ScanShared& sh,
                                  long long& viol_out, int& count_out, int& bad_out) {
    const int n = a.tb.n, m = a.tb.m, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double* g = a.g + (size_t)b * m;
    const double* jac = a.jac + (size_t)b * m * n;
    const double* lo = a.lo + (size_t)b * m;
    const double* hi = a.hi + (size_t)b * m;
    const int n_unchecked = a.tb.Q - a.n_checked_collision;
    int base = 0, bad = 0;
    long long vsum = 0;
    for (int t = t0; t < t1; t++) {
        const int i = tile_row(a, b, ts, t, tid);
        const bool in = i >= 0;
        const double gi = in ? g[i] : 0.0, li = in ? lo[i] : -1e300, ui = in ? hi[i] : 1e300;
        vsum += row_violation(gi, li, ui);
        if (in) {
            const int ic = i - a.tb.row0 - a.n_checked_collision;  // >= 0: behind the re-checked collision rows
            const double slack = i < a.tb.row0 ? a.torque_slack : ic < 0 ? a.collision_slack : 0.0;
            if ((ic < 0 || ic >= n_unchecked) && (gi < li - slack || gi > ui + slack)) bad++;
        }
        double J[NV], l1 = 0.0;
#pragma unroll
        for (int j = 0; j < NV; j++) { J[j] = (in && j < n) ? jac[(size_t)i * n + j] : 0.0; l1 += fabs(J[j]); }
        const bool fh = in && row_upper_candidate(gi, ui, l1);
        const bool fl = in && row_lower_candidate(gi, li, l1);
        const unsigned long long bh = __ballot(fh), bl = __ballot(fl);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (lane == 0) sh.wave_tot[wv] = __popcll(bh) + __popcll(bl);
        __syncthreads();
        int pos = base + __popcll(bh & below) + __popcll(bl & below);
        for (int w2 = 0; w2 < wv; w2++) pos += sh.wave_tot[w2];
        if (fh) {
            if (pos < cap) {
                SolveRow r; r.idx = i; r.side = 0; r.v = gi - ui;
#pragma unroll
                for (int j = 0; j < NV; j++) r.a[j] = -J[j];
                st_coh_row(rows + pos, r);
            }
            pos++;
        }
        if (fl && pos < cap) {
            SolveRow r; r.idx = i; r.side = 1; r.v = li - gi;
#pragma unroll
            for (int j = 0; j < NV; j++) r.a[j] = J[j];
            st_coh_row(rows + pos, r);
        }
        base += sh.wave_tot[0] + sh.wave_tot[1] + sh.wave_tot[2] + sh.wave_tot[3];
        __syncthreads();
    }
    block_sum(vsum, bad, sh.red, sh.redi);
    viol_out = vsum;
    bad_out = bad;
    count_out = base;
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------ the leader
constexpr int kFlagsInLds = 1024;   // QP rows whose active / excluded flags fit in LDS (more: the global byte arrays)
struct Leader {
    // SQP state (ProblemState + the line-search variables of solver.hip)
    double x[NV], d[NV], xt[NV], gradf[NV], Hd[NV];
    double f, viol, mu, alpha, phi0, dphi;
    int it, ls, iters, evals, status, stage, ncand, bad;
    long long t_start;
    // the problem's constants, staged once (global loads from a single lane cost ~1 us each)
    double bz[3 * NV], qdes[NV];
    // QP state and temporaries (thread 0 works on them with run-time indices: LDS, not scratch)
    double qx[NV], u[NV + 1], An[NV][NV], np[NV];
    double M[NV * NV], Lc[NV][NV], Lci[NV], invHd[NV], rhs[NV], r[NV], z[NV];   // Lci[i] = 1 / Lc[i][i], invHd[j] = 1 / Hd[j] (solver.hip spd_solve)
    double bp, up, max_mult, sigma;
    int A[NV], q, p, qp_iter, feasible, stop, added, chol_rows;
    unsigned char act[kFlagsInLds], exc[kFlagsInLds];
    // scratch of the block-wide reductions
    double rs[256];
    int ri[256];
    int scan[1024];
};

// QP rows: [0, ncand) the gathered candidates (b = v - sigma * max(v, 0)), then 2n bound rows  x_l <= x + d <= x_u
__device__ inline void qp_row(const SolveArgs& a, const Leader& L, const SolveRow* cand, int i, double* arow, double& brow) {
    if (i < L.ncand) {
        const SolveRow& r = cand[i];
#pragma unroll
        for (int j = 0; j < NV; j++) arow[j] = r.a[j];
        const double v = r.v;
        brow = v - (v > 0 ? L.sigma * v : 0.0);
    } else {
        const int e = i - L.ncand, j = e >> 1;
        const double sg = (e & 1) == 0 ? 1.0 : -1.0;
#pragma unroll
        for (int jj = 0; jj < NV; jj++) arow[jj] = jj == j ? sg : 0.0;    // (no run-time register index: that would be scratch)
        brow = (e & 1) == 0 ? -1.0 - L.x[j] : -(1.0 - L.x[j]);            // r.b = xl[j] - s.x[j];  r2.b = -(xu[j] - s.x[j])
    }
}

// Goldfarb-Idnani for  min 1/2 d'Gd + g0'd  s.t.  a_i'd >= b_i,  G = diag(Hd) > 0: solve_qp of solver.hip, same arithmetic, arranged
// for a GPU block.  All 256 threads call it; results in L.qx / L.max_mult / L.feasible.
//   * the search for the most violated row uses the whole block; with at most 256 rows every thread keeps its row in registers
//     across the steps, and the active / excluded flags live in LDS while they fit;
//   * the step runs on wave 0, all 64 lanes in lockstep on the same LDS state (redundant, hence free).  The loops with a division
//     per term -- M = N'G^-1 N, N'G^-1 np and z -- are dealt one entry per lane, each entry summed in the host form's order;
//   * the Cholesky factor of M and the two triangular solves run in registers (fully unrolled, guarded by the active count) and
//     the factor is extended row by row as rows enter the active set: row i of the factor depends on rows <= i of M only, so the
//     numbers are those of the from-scratch factorisation the host form does at every step.
template <int WPS>   // (one copy per compiled occupancy, see leader_step)
__device__ inline void solve_qp_device(const SolveArgs& a, Leader& L, const SolveRow* cand, unsigned char* is_active_g, unsigned char* excluded_g,
                                       int max_iter = kQpMaxSteps) {
    const int n = a.tb.n, tid = threadIdx.x;
    const int mrows = L.ncand + 2 * n;
    unsigned char* is_active = mrows <= kFlagsInLds ? L.act : is_active_g;
    unsigned char* excluded = mrows <= kFlagsInLds ? L.exc : excluded_g;
    if (tid == 0) {
        for (int j = 0; j < NV; j++) { L.invHd[j] = 1.0 / L.Hd[j]; L.qx[j] = j < n ? -L.gradf[j] * L.invHd[j] : 0.0; }
        L.q = 0; L.qp_iter = 0; L.feasible = 1; L.max_mult = 0; L.stop = 0; L.chol_rows = 0;
    }
    for (int i = tid; i < mrows; i += 256) { is_active[i] = 0; excluded[i] = 0; }
    const bool one_row = mrows <= 256;
    double my_a[NV], my_b = 0.0;
#pragma unroll
    for (int j = 0; j < NV; j++) my_a[j] = 0.0;
    if (one_row && tid < mrows) qp_row(a, L, cand, tid, my_a, my_b);
    __syncthreads();
    // wave 0's QP state (the same values in all of its lanes): iterate, multipliers, active count, valid rows of the Cholesky factor
    double qx[NV], u[NV + 1];
#pragma unroll
    for (int j = 0; j < NV; j++) { qx[j] = j < n ? L.qx[j] : 0.0; u[j] = 0.0; }
    u[NV] = 0.0;
    int q = 0, cv = 0;
    long long* qst = a.stamps ? a.stamps + (size_t)(blockIdx.x / a.nb + a.b0) * 64 + 44 : nullptr;   // ARMOUR_SOLVE_TIMING: ticks per part of the QP step, summed over the steps
    long long q_t = qst ? wall_clock64() : 0;
#define QP_LAP(slot) if (qst && tid == 0) { const long long n__ = wall_clock64(); qst[slot] += n__ - q_t; q_t = n__; }
    for (;;) {
        // most violated inactive row: smallest s = a_i'x - b_i below -1e-10, the first such row on ties
        double best = -1e-10;
        int bi = -1;
        if (one_row) {
            if (tid < mrows && !is_active[tid] && !excluded[tid]) {
                double s = -my_b;
#pragma unroll
                for (int j = 0; j < NV; j++) if (j < n) s += my_a[j] * L.qx[j];
                if (s < best) { best = s; bi = tid; }
            }
        } else {
            for (int i = tid; i < mrows; i += 256) {
                if (is_active[i] || excluded[i]) continue;
                double ar[NV], br;
                qp_row(a, L, cand, i, ar, br);
                double s = -br;
#pragma unroll
                for (int j = 0; j < NV; j++) if (j < n) s += ar[j] * L.qx[j];   // (unrolled: `ar` stays in registers)
                if (s < best) { best = s; bi = i; }
            }
        }
        // lexicographic (value, index) minimum: the sequential scan keeps the FIRST row attaining the minimum.
        // Inside each wave by shuffles, then the four waves' results through LDS: two barriers instead of nine.
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (oi >= 0 && (bi < 0 || o < best || (o == best && oi < bi))) { best = o; bi = oi; }
        }
        if ((tid & 63) == 0) { L.rs[tid >> 6] = best; L.ri[tid >> 6] = bi; }
        __syncthreads();
        {
            best = L.rs[0]; bi = L.ri[0];
#pragma unroll
            for (int w2 = 1; w2 < 4; w2++) {
                const double o = L.rs[w2];
                const int oi = L.ri[w2];
                if (oi >= 0 && (bi < 0 || o < best || (o == best && oi < bi))) { best = o; bi = oi; }
            }
        }
        const int p = bi;
        __syncthreads();
        QP_LAP(0)
        if (p < 0) break;
        if (one_row && tid == p) {   // the entering row's normal and right-hand side: from the thread that holds them
#pragma unroll
            for (int j = 0; j < NV; j++) L.np[j] = my_a[j];
            L.bp = my_b;
        }
        __syncthreads();
        QP_LAP(1)
        if (tid < 64) {
            // Wave 0, all lanes in lockstep.  The small vectors (entering normal, iterate, multipliers, z, r) live in REGISTERS, the same
            // values in every lane; LDS carries only what lanes exchange (M, rhs, the z entries) and what the other waves read (qx, flags).
            // (With everything in LDS a step was a chain of ~150 dependent LDS round trips: 6 us.)
            L.p = p;
            if (++L.qp_iter > max_iter) { L.feasible = 0; L.stop = 1; }
            else {
                if (!one_row) {
                    double npr0[NV], bpr;
                    qp_row(a, L, cand, p, npr0, bpr);
#pragma unroll
                    for (int j = 0; j < NV; j++) L.np[j] = npr0[j];
                    L.bp = bpr;
                    WAVE_LDS_SYNC();
                }
                double npr[NV], ih[NV];
#pragma unroll
                for (int j = 0; j < NV; j++) { npr[j] = L.np[j]; ih[j] = L.invHd[j]; }
                const double bp = L.bp;
                double up = 0.0;
                bool added = false;
                for (int guard = 0; guard < 4 * NV + 8 && !added; guard++) {
                    // r = N* np,  z = G^-1 (np - N r)
                    double rr[NV];
#pragma unroll
                    for (int i = 0; i < NV; i++) rr[i] = 0.0;
                    WAVE_LDS_SYNC();
                    if (q > 0) {
                        if (tid < NV * NV) {
                            const int i = tid / NV, k = tid - i * NV;
                            if (i < q && k <= i) {
                                double s = 0;
#pragma unroll
                                for (int j = 0; j < NV; j++) if (j < n) s += L.An[i][j] * L.An[k][j] * ih[j];
                                L.M[i * NV + k] = s; L.M[k * NV + i] = s;
                            }
                        } else if (tid < NV * NV + NV) {
                            const int i = tid - NV * NV;
                            if (i < q) {
                                double s = 0;
#pragma unroll
                                for (int j = 0; j < NV; j++) if (j < n) s += L.An[i][j] * npr[j] * ih[j];
                                L.rhs[i] = s;
                            }
                        }
                        WAVE_LDS_SYNC();
                        QP_LAP(2)
                        // Cholesky M = Lc Lc' (spd_solve of solver.hip), rows [cv, q) new, in registers
                        double Lr[NV][NV], Li[NV];
#pragma unroll
                        for (int i = 0; i < NV; i++) {
                            Li[i] = i < cv ? L.Lci[i] : 0.0;
#pragma unroll
                            for (int k = 0; k <= i; k++) Lr[i][k] = i < cv ? L.Lc[i][k] : 0.0;
                        }
                        bool spd = true;
#pragma unroll
                        for (int i = 0; i < NV; i++) {
                            if (i >= cv && i < q && spd) {
#pragma unroll
                                for (int jj = 0; jj <= i; jj++) {
                                    if (spd) {
                                        double s = L.M[i * NV + jj];
#pragma unroll
                                        for (int k = 0; k < jj; k++) s -= Lr[i][k] * Lr[jj][k];
                                        if (jj == i) {
                                            if (s <= 1e-14 * fabs(L.M[i * NV + i]) || s <= 0) spd = false;
                                            else { Lr[i][i] = sqrt(s); Li[i] = 1.0 / Lr[i][i]; }
                                        } else {
                                            Lr[i][jj] = s * Li[jj];
                                        }
                                    }
                                }
                                if (spd) {
#pragma unroll
                                    for (int k = 0; k <= i; k++) L.Lc[i][k] = Lr[i][k];
                                    L.Lci[i] = Li[i];
                                    cv = i + 1;
                                }
                            }
                        }
                        if (!spd) { excluded[p] = 1; break; }  // dependent active set: skip this row
                        double tt[NV];
#pragma unroll
                        for (int i = 0; i < NV; i++) {
                            tt[i] = 0.0;
                            if (i < q) {
                                double s = L.rhs[i];
#pragma unroll
                                for (int k = 0; k < i; k++) s -= Lr[i][k] * tt[k];
                                tt[i] = s * Li[i];
                            }
                        }
#pragma unroll
                        for (int i = NV - 1; i >= 0; i--) {
                            if (i < q) {
                                double s = tt[i];
#pragma unroll
                                for (int k = i + 1; k < NV; k++) if (k < q) s -= Lr[k][i] * rr[k];
                                rr[i] = s * Li[i];
                            }
                        }
                    }
                    QP_LAP(3)
                    // z: lane j forms entry j (np_j - sum_i An[i][j] r_i in row order, times 1/Hd_j), then every lane takes all of them
                    double zmine = 0.0;
                    {
                        const int j = tid < NV ? tid : 0;
                        double s = 0.0;
#pragma unroll
                        for (int jj = 0; jj < NV; jj++) if (jj == j) s = npr[jj];
#pragma unroll
                        for (int i = 0; i < NV; i++) if (i < q) s -= L.An[i][j] * rr[i];
                        double ihj = 0.0;
#pragma unroll
                        for (int jj = 0; jj < NV; jj++) if (jj == j) ihj = ih[jj];
                        zmine = s * ihj;
                    }
                    double z[NV];
#pragma unroll
                    for (int j = 0; j < NV; j++) z[j] = j < n ? __shfl(zmine, j, 64) : 0.0;
                    double zz = 0, znp = 0;
#pragma unroll
                    for (int j = 0; j < NV; j++) if (j < n) { zz += z[j] * z[j]; znp += z[j] * npr[j]; }
                    // step lengths
                    double t1 = kInf;
                    int l = -1;
#pragma unroll
                    for (int i = 0; i < NV; i++)
                        if (i < q && rr[i] > 1e-14) { const double ur = u[i] / rr[i]; if (ur < t1) { t1 = ur; l = i; } }
                    double sp = -bp;
#pragma unroll
                    for (int j = 0; j < NV; j++) if (j < n) sp += npr[j] * qx[j];
                    double t2 = kInf;
                    if (zz > 1e-24 && znp > 1e-16) t2 = -sp / znp;
                    if (t2 < 0) t2 = 0;
                    const double t = t1 < t2 ? t1 : t2;
                    if (t >= kInf) { L.feasible = 0; break; }
                    QP_LAP(4)
                    const bool dual_only = t2 >= kInf;
                    if (!dual_only) {
#pragma unroll
                        for (int j = 0; j < NV; j++) if (j < n) qx[j] += t * z[j];
                    }
#pragma unroll
                    for (int i = 0; i < NV; i++) if (i < q) u[i] -= t * rr[i];
                    up += t;
                    if (!dual_only && t == t2) {  // full step: the row becomes active
                        if (q >= n) { L.feasible = 0; break; }
#pragma unroll
                        for (int i = 0; i < NV; i++) if (i == q) u[i] = up;
                        if (tid < NV) L.An[q][tid] = L.np[tid];
                        if (tid == 0) { L.A[q] = p; is_active[p] = 1; }
                        q++;
                        added = true;
                    } else {        // dual step only, or a partial step: drop the blocking row (and try again)
                        if (tid == 0) is_active[L.A[l]] = 0;
                        WAVE_LDS_SYNC();
                        for (int i = l; i < q - 1; i++) {   // (rare; rows move up one by one, every lane a column)
                            if (tid < NV) L.An[i][tid] = L.An[i + 1][tid];
                            if (tid == 0) L.A[i] = L.A[i + 1];
                            WAVE_LDS_SYNC();
                        }
#pragma unroll
                        for (int i = 0; i < NV; i++) if (i >= l && i < q - 1) u[i] = u[i + 1];
                        q--;
                        if (cv > l) cv = l;
                    }
                }
                if (tid < NV) L.qx[tid] = 0.0;
#pragma unroll
                for (int j = 0; j < NV; j++) if (tid == j) L.qx[j] = qx[j];
                L.added = added ? 1 : 0;
                if (!L.feasible) L.stop = 1;
                else if (!added && !excluded[p]) excluded[p] = 1;  // could not make progress on this row
            }
        }
        QP_LAP(5)
        __threadfence_block();
        __syncthreads();
        QP_LAP(6)
        if (L.stop) break;
    }
#undef QP_LAP
    if (tid == 0) {
        L.q = q; L.chol_rows = cv;
#pragma unroll
        for (int i = 0; i < NV; i++) L.u[i] = u[i];
    }
    __syncthreads();
    if (tid == 0)
        for (int i = 0; i < L.q; i++) if (L.u[i] > L.max_mult) L.max_mult = L.u[i];
    // excluded rows that remain violated mean the linearisation is inconsistent
    int viol = 0;
    if (L.feasible)
        for (int i = tid; i < mrows; i += 256) {
            double ar[NV], br;   // (every row, active ones included)
            qp_row(a, L, cand, i, ar, br);
            double s = -br;
#pragma unroll
            for (int j = 0; j < NV; j++) if (j < n) s += ar[j] * L.qx[j];   // (unrolled: `ar` stays in registers)
            if (s < -1e-7) viol = 1;
        }
    L.ri[tid] = viol;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (tid < s2) L.ri[tid] |= L.ri[tid + s2];
        __syncthreads();
    }
    if (tid == 0 && L.ri[0]) L.feasible = 0;
    __syncthreads();
}

__device__ inline int ld_lds_int(const int* p) { return *reinterpret_cast<const volatile int*>(p); }
// ---- the four elastic attempts of one QP side by side, one per wave ------------------------------------------------------------------
// solver.hip tries sigma = 0, 0.5, 0.9, 0.99 one after the other and keeps the first attempt that is feasible; the attempts share nothing but
// their input (each starts from the unconstrained minimiser with an empty active set), so here wave w runs attempt w on a state of its own
// and the leader takes the lowest-numbered feasible one: the same numbers as the sequential form, in the time of the longest attempt that
// matters instead of the sum (an infeasible problem pays all four at every SQP iteration -- after the row culling that was what a batch waited
// for).  A wave that finishes feasible tells the attempts above it to stop.  Inside an attempt nothing crosses waves: the search for the most
// violated row runs on the wave's 64 lanes (with at most 64 * kRegRows rows every lane keeps its rows in registers across the steps; more: it
// reads them four at a time), the step is the lockstep code of solve_qp_device on the wave's own LDS state, and there is no block barrier
// until all four are done.
struct QpWave {
    double qx[NV], An[NV][NV], np[NV], M[NV * NV], Lc[NV][NV], Lci[NV], rhs[NV];
    double bp, max_mult;
    int A[NV], q, qp_iter, feasible;
    unsigned char act[kFlagsInLds], exc[kFlagsInLds];
};
struct QpShared {
    QpWave w[4];
    int first_ok;     // lowest attempt that has finished feasible so far (4: none)
};
__device__ inline double attempt_sigma(int attempt) { return attempt == 0 ? 0.0 : attempt == 1 ? 0.5 : attempt == 2 ? 0.9 : 0.99; }
// QP row i of the attempt with elasticity sigma (qp_row with the attempt's own sigma)
__device__ inline void qp_row_s(const SolveArgs& a, const Leader& L, const SolveRow* cand, int i, double sigma, double* arow, double& brow) {
    if (i < L.ncand) {
        const SolveRow& r = cand[i];
#pragma unroll
        for (int j = 0; j < NV; j++) arow[j] = r.a[j];
        const double v = r.v;
        brow = v - (v > 0 ? sigma * v : 0.0);
    } else {
        const int e = i - L.ncand, j = e >> 1;
        const double sg = (e & 1) == 0 ? 1.0 : -1.0;
#pragma unroll
        for (int jj = 0; jj < NV; jj++) arow[jj] = jj == j ? sg : 0.0;
        brow = (e & 1) == 0 ? -1.0 - L.x[j] : -(1.0 - L.x[j]);
    }
}
// ---- wave-level exchanges without LDS: a value of a lane whose index is wave-uniform (v_readlane), and the lexicographic minimum of
// (value, index) pairs over the wave by DPP moves (row shifts inside the rows of 16, then the two row broadcasts of gfx9: the total ends up in lane 63)
__device__ inline double readlane_f64(double v, int src_lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src_lane), __builtin_amdgcn_readlane(__double2loint(v), src_lane));
}
template <int CTRL, int ROW_MASK>
__device__ inline void min_pair_dpp(double& best, int& bi) {
    // a lane without a source lane (or masked off) gets its own value back: combining a pair with itself changes nothing
    const int hi = __double2hiint(best), lo = __double2loint(best);
    const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const int oi = __builtin_amdgcn_update_dpp(bi, bi, CTRL, ROW_MASK, 0xf, false);
    const double o = __hiloint2double(ohi, olo);
    if (oi >= 0 && (bi < 0 || o < best || (o == best && oi < bi))) { best = o; bi = oi; }
}
// the first row attaining the smallest value (the sequential scan's choice), as its index, in every lane
__device__ inline int wave_argmin_first(double best, int bi) {
    min_pair_dpp<0x111, 0xf>(best, bi);   // row_shr:1
    min_pair_dpp<0x112, 0xf>(best, bi);   // row_shr:2
    min_pair_dpp<0x114, 0xf>(best, bi);   // row_shr:4
    min_pair_dpp<0x118, 0xf>(best, bi);   // row_shr:8   -> lane 15 of every row holds its row's minimum
    min_pair_dpp<0x142, 0xa>(best, bi);   // row_bcast:15 into rows 1 and 3
    min_pair_dpp<0x143, 0xc>(best, bi);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's
    return __builtin_amdgcn_readlane(bi, 63);
}

template <int WPS>
__device__ inline void solve_qp_wave(const SolveArgs& a, const Leader& L, QpShared& S, const SolveRow* cand, const double* lds_rows, unsigned char* flags_g, int max_iter = kQpMaxSteps) {
    constexpr int kRegRows = WPS == 1 ? 4 : 2;   // rows a lane keeps in registers (the two-waves-per-SIMD build has 256 registers for everything)
    const int n = a.tb.n, lane = threadIdx.x & 63, attempt = threadIdx.x >> 6;
    QpWave& W = S.w[attempt];
    const double sigma = attempt_sigma(attempt);
    const int mrows = L.ncand + 2 * n;
    unsigned char* is_active = mrows <= kFlagsInLds ? W.act : flags_g + (size_t)(2 * attempt) * (a.cap_rows + 2 * NV);
    unsigned char* excluded = mrows <= kFlagsInLds ? W.exc : flags_g + (size_t)(2 * attempt + 1) * (a.cap_rows + 2 * NV);
    for (int i = lane; i < mrows; i += 64) { is_active[i] = 0; excluded[i] = 0; }
    const bool reg_rows = mrows <= 64 * kRegRows;
    // a row of the QP: from the candidates staged in LDS (component-major: lanes read consecutive words) when they fit, from global memory otherwise
    auto load_row = [&](int i, double* arow, double& brow) {
        if (lds_rows && i < L.ncand) {
#pragma unroll
            for (int j = 0; j < NV; j++) arow[j] = lds_rows[(size_t)j * a.lds_rows + i];
            const double v = lds_rows[(size_t)NV * a.lds_rows + i];
            brow = v - (v > 0 ? sigma * v : 0.0);
        } else qp_row_s(a, L, cand, i, sigma, arow, brow);
    };
    double my_a[kRegRows][NV], my_b[kRegRows];
#pragma unroll
    for (int r = 0; r < kRegRows; r++) {
        my_b[r] = 0.0;
#pragma unroll
        for (int j = 0; j < NV; j++) my_a[r][j] = 0.0;
        if (reg_rows && lane + 64 * r < mrows) load_row(lane + 64 * r, my_a[r], my_b[r]);
    }
    // the wave's QP state, the same values in all of its lanes: iterate, multipliers, active count, valid rows of the Cholesky factor
    double qx[NV], u[NV + 1], ih[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) { ih[j] = L.invHd[j]; qx[j] = j < n ? -L.gradf[j] * ih[j] : 0.0; u[j] = 0.0; }
    u[NV] = 0.0;
    int q = 0, cv = 0, qp_iter = 0;
    bool feasible = true, stop = false;
    unsigned act_bits = 0, exc_bits = 0;   // reg_rows: the active / excluded flags of this lane's rows (bit r: row lane + 64 r), instead of the LDS bytes
    long long* qst = a.stamps && attempt == 0 ? a.stamps + (size_t)(blockIdx.x / a.nb + a.b0) * 64 + 50 : nullptr;   // ARMOUR_SOLVE_TIMING: ticks per part of attempt 0's steps
    long long q_t = qst ? wall_clock64() : 0;
#define QPW_LAP(slot) if (qst && lane == 0) { const long long n__ = wall_clock64(); qst[slot] += n__ - q_t; q_t = n__; }
    WAVE_LDS_SYNC();
    {   // the box-clipped minimiser first (solver_common.h box_clipped_step: the host form's arithmetic), verified against every row like any result
        double lo[NV], hi[NV], hd[NV], g0[NV], d[NV];
#pragma unroll
        for (int j = 0; j < NV; j++) { lo[j] = -1.0 - L.x[j]; hi[j] = 1.0 - L.x[j]; hd[j] = L.Hd[j]; g0[j] = L.gradf[j]; d[j] = 0.0; }
        const double mmc = box_clipped_step(n, hd, ih, g0, lo, hi, d);
        int viol = 0;
        for (int i = lane; i < mrows; i += 64) {
            double ar[NV], br;
            load_row(i, ar, br);
            double s = -br;
#pragma unroll
            for (int j = 0; j < NV; j++) if (j < n) s += ar[j] * d[j];
            if (s < -1e-7) viol = 1;
        }
        if (__ballot(viol != 0) == 0ull) {
            if (lane == 0) {
                W.q = 0; W.qp_iter = 0; W.feasible = 1; W.max_mult = mmc;
                atomicMin(&S.first_ok, attempt);
            }
#pragma unroll
            for (int j = 0; j < NV; j++) if (lane == j) W.qx[j] = d[j];
            return;
        }
    }
    while (!stop) {
        if (ld_lds_int(&S.first_ok) < attempt) { feasible = false; break; }   // a lower attempt is feasible: this one is not needed
        // most violated inactive row: smallest s = a_i'x - b_i below -1e-10, the first such row on ties
        double best = -1e-10;
        int bi = -1;
        if (reg_rows) {
#pragma unroll
            for (int r = 0; r < kRegRows; r++) {
                const int i = lane + 64 * r;
                if (i < mrows && !(((act_bits | exc_bits) >> r) & 1u)) {
                    double s = -my_b[r];
#pragma unroll
                    for (int j = 0; j < NV; j++) if (j < n) s += my_a[r][j] * qx[j];
                    if (s < best) { best = s; bi = i; }
                }
            }
        } else {
            for (int i0 = lane; i0 < mrows; i0 += 4 * 64) {   // four rows of a lane requested together (a row is 72 bytes of global memory)
                double ar[4][NV], br[4];
#pragma unroll
                for (int e = 0; e < 4; e++) load_row(min(i0 + 64 * e, mrows - 1), ar[e], br[e]);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int i = i0 + 64 * e;
                    if (i < mrows && !is_active[i] && !excluded[i]) {
                        double s = -br[e];
#pragma unroll
                        for (int j = 0; j < NV; j++) if (j < n) s += ar[e][j] * qx[j];
                        if (s < best) { best = s; bi = i; }
                    }
                }
            }
        }
        // lexicographic (value, index) minimum: the sequential scan keeps the FIRST row attaining the minimum
        const int p = wave_argmin_first(best, bi);
        QPW_LAP(0)
        if (p < 0) break;
        if (++qp_iter > max_iter) { feasible = false; break; }
        // the entering row's normal and right-hand side, in every lane
        double npr[NV], bp;
        if (reg_rows) {
            const int slot = p >> 6, src = p & 63;
            double v[NV + 1];
#pragma unroll
            for (int j = 0; j <= NV; j++) v[j] = 0.0;
#pragma unroll
            for (int r = 0; r < kRegRows; r++) {
                if (slot == r) {
#pragma unroll
                    for (int j = 0; j < NV; j++) v[j] = my_a[r][j];
                    v[NV] = my_b[r];
                }
            }
#pragma unroll
            for (int j = 0; j < NV; j++) npr[j] = readlane_f64(v[j], src);
            bp = readlane_f64(v[NV], src);
        } else {
            load_row(p, npr, bp);
        }
        if (lane < NV) W.np[lane] = 0.0;
#pragma unroll
        for (int j = 0; j < NV; j++) if (lane == j) W.np[j] = npr[j];
        double up = 0.0;
        bool added = false;
        QPW_LAP(1)
        for (int guard = 0; guard < 4 * NV + 8 && !added; guard++) {
            // r = N* np,  z = G^-1 (np - N r)
            double rr[NV];
#pragma unroll
            for (int i = 0; i < NV; i++) rr[i] = 0.0;
            WAVE_LDS_SYNC();
            if (q > 0) {
                if (lane < NV * NV) {
                    const int i = lane / NV, k = lane - i * NV;
                    if (i < q && k <= i) {
                        double s = 0;
#pragma unroll
                        for (int j = 0; j < NV; j++) if (j < n) s += W.An[i][j] * W.An[k][j] * ih[j];
                        W.M[i * NV + k] = s; W.M[k * NV + i] = s;
                    }
                } else if (lane < NV * NV + NV) {
                    const int i = lane - NV * NV;
                    if (i < q) {
                        double s = 0;
#pragma unroll
                        for (int j = 0; j < NV; j++) if (j < n) s += W.An[i][j] * npr[j] * ih[j];
                        W.rhs[i] = s;
                    }
                }
                WAVE_LDS_SYNC();
                QPW_LAP(2)
                // Cholesky M = Lc Lc' (spd_solve of solver.hip), rows [cv, q) new, in registers
                double Lr[NV][NV], Li[NV];
#pragma unroll
                for (int i = 0; i < NV; i++) {
                    Li[i] = i < cv ? W.Lci[i] : 0.0;
#pragma unroll
                    for (int k = 0; k <= i; k++) Lr[i][k] = i < cv ? W.Lc[i][k] : 0.0;
                }
                bool spd = true;
#pragma unroll
                for (int i = 0; i < NV; i++) {
                    if (i >= cv && i < q && spd) {
#pragma unroll
                        for (int jj = 0; jj <= i; jj++) {
                            if (spd) {
                                double s = W.M[i * NV + jj];
#pragma unroll
                                for (int k = 0; k < jj; k++) s -= Lr[i][k] * Lr[jj][k];
                                if (jj == i) {
                                    if (s <= 1e-14 * fabs(W.M[i * NV + i]) || s <= 0) spd = false;
                                    else { Lr[i][i] = sqrt(s); Li[i] = 1.0 / Lr[i][i]; }
                                } else {
                                    Lr[i][jj] = s * Li[jj];
                                }
                            }
                        }
                        if (spd) {
#pragma unroll
                            for (int k = 0; k <= i; k++) W.Lc[i][k] = Lr[i][k];
                            W.Lci[i] = Li[i];
                            cv = i + 1;
                        }
                    }
                }
                if (!spd) {   // dependent active set: skip this row
                    if (reg_rows) { if (lane == (p & 63)) exc_bits |= 1u << (p >> 6); }
                    else if (lane == 0) excluded[p] = 1;
                    break;
                }
                double tt[NV];
#pragma unroll
                for (int i = 0; i < NV; i++) {
                    tt[i] = 0.0;
                    if (i < q) {
                        double s = W.rhs[i];
#pragma unroll
                        for (int k = 0; k < i; k++) s -= Lr[i][k] * tt[k];
                        tt[i] = s * Li[i];
                    }
                }
#pragma unroll
                for (int i = NV - 1; i >= 0; i--) {
                    if (i < q) {
                        double s = tt[i];
#pragma unroll
                        for (int k = i + 1; k < NV; k++) if (k < q) s -= Lr[k][i] * rr[k];
                        rr[i] = s * Li[i];
                    }
                }
            }
            QPW_LAP(3)
            // z: lane j forms entry j (np_j - sum_i An[i][j] r_i in row order, times 1/Hd_j), then every lane takes all of them
            double zmine = 0.0;
            {
                const int j = lane < NV ? lane : 0;
                double s = 0.0;
#pragma unroll
                for (int jj = 0; jj < NV; jj++) if (jj == j) s = npr[jj];
#pragma unroll
                for (int i = 0; i < NV; i++) if (i < q) s -= W.An[i][j] * rr[i];
                double ihj = 0.0;
#pragma unroll
                for (int jj = 0; jj < NV; jj++) if (jj == j) ihj = ih[jj];
                zmine = s * ihj;
            }
            double z[NV];
#pragma unroll
            for (int j = 0; j < NV; j++) z[j] = j < n ? readlane_f64(zmine, j) : 0.0;
            double zz = 0, znp = 0;
#pragma unroll
            for (int j = 0; j < NV; j++) if (j < n) { zz += z[j] * z[j]; znp += z[j] * npr[j]; }
            // step lengths
            double t1 = kInf;
            int l = -1;
#pragma unroll
            for (int i = 0; i < NV; i++)
                if (i < q && rr[i] > 1e-14) { const double ur = u[i] / rr[i]; if (ur < t1) { t1 = ur; l = i; } }
            double sp = -bp;
#pragma unroll
            for (int j = 0; j < NV; j++) if (j < n) sp += npr[j] * qx[j];
            double t2 = kInf;
            if (zz > 1e-24 && znp > 1e-16) t2 = -sp / znp;
            if (t2 < 0) t2 = 0;
            const double t = t1 < t2 ? t1 : t2;
            if (t >= kInf) { feasible = false; break; }
            QPW_LAP(4)
            const bool dual_only = t2 >= kInf;
            if (!dual_only) {
#pragma unroll
                for (int j = 0; j < NV; j++) if (j < n) qx[j] += t * z[j];
            }
#pragma unroll
            for (int i = 0; i < NV; i++) if (i < q) u[i] -= t * rr[i];
            up += t;
            if (!dual_only && t == t2) {  // full step: the row becomes active
                if (q >= n) { feasible = false; break; }
#pragma unroll
                for (int i = 0; i < NV; i++) if (i == q) u[i] = up;
                if (lane < NV) W.An[q][lane] = W.np[lane];
                if (lane == 0) W.A[q] = p;
                if (reg_rows) { if (lane == (p & 63)) act_bits |= 1u << (p >> 6); }
                else if (lane == 0) is_active[p] = 1;
                q++;
                added = true;
            } else {        // dual step only, or a partial step: drop the blocking row (and try again)
                {
                    const int d = ld_lds_int(&W.A[l]);   // (written by lane 0 in an earlier step, fenced since)
                    if (reg_rows) { if (lane == (d & 63)) act_bits &= ~(1u << (d >> 6)); }
                    else if (lane == 0) is_active[d] = 0;
                }
                WAVE_LDS_SYNC();
                for (int i = l; i < q - 1; i++) {   // (rare; rows move up one by one, every lane a column)
                    if (lane < NV) W.An[i][lane] = W.An[i + 1][lane];
                    if (lane == 0) W.A[i] = W.A[i + 1];
                    WAVE_LDS_SYNC();
                }
#pragma unroll
                for (int i = 0; i < NV; i++) if (i >= l && i < q - 1) u[i] = u[i + 1];
                q--;
                if (cv > l) cv = l;
            }
        }
        QPW_LAP(5)
        if (!feasible) break;
        WAVE_LDS_SYNC();
        if (!added) {   // could not make progress on this row
            if (reg_rows) { if (lane == (p & 63)) exc_bits |= 1u << (p >> 6); }
            else if (!excluded[p]) { WAVE_LDS_SYNC(); if (lane == 0) excluded[p] = 1; }
        }
        WAVE_LDS_SYNC();
    }
    // the result verified against every row: excluded rows that remain violated mean the linearisation is inconsistent, active rows may have drifted
    if (feasible) {
        int viol = 0;
        for (int i = lane; i < mrows; i += 64) {
            double ar[NV], br;   // (every row, active ones included: solver.hip solve_qp, "The result is VERIFIED")
            load_row(i, ar, br);
            double s = -br;
#pragma unroll
            for (int j = 0; j < NV; j++) if (j < n) s += ar[j] * qx[j];
            if (s < -1e-7) viol = 1;
        }
        if (__ballot(viol != 0) != 0ull) feasible = false;
    }
    double mm = 0.0;
#pragma unroll
    for (int i = 0; i < NV; i++) if (i < q && u[i] > mm) mm = u[i];
    if (lane == 0) {
        W.q = q; W.qp_iter = qp_iter; W.feasible = feasible ? 1 : 0; W.max_mult = mm;
        if (feasible) atomicMin(&S.first_ok, attempt);
    }
#pragma unroll
    for (int j = 0; j < NV; j++) if (lane == j) W.qx[j] = qx[j];
}

// gather the candidate rows of all blocks of the group, in block (= row) order, into the problem's contiguous buffer
__device__ inline bool gather_candidates(const SolveArgs& a, Leader& L, int b, SolveRow* cand) {
    const int tid = threadIdx.x, nb = a.nb;
    const BlockWord* bw = a.blk_word + (size_t)(b - a.b0) * nb;
    // exclusive prefix sum of the block counts (nb <= 1024)
    int mine[4], tot = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) { const int jb = tid * 4 + e; mine[e] = jb < nb ? ld_coh(&bw[jb].count) : 0; tot += mine[e]; }
    int total;
    int run = block_exclusive_scan(tot, L.ri, total);
#pragma unroll
    for (int e = 0; e < 4; e++) { L.scan[tid * 4 + e] = run; run += mine[e]; }
    bool ok = total <= a.cap_rows;
#pragma unroll
    for (int e = 0; e < 4; e++) if (tid * 4 + e < nb && mine[e] > a.cap_blk) ok = false;
    const bool overflow = __syncthreads_or(ok ? 0 : 1) != 0;   // (also: L.scan is written)
    if (overflow) return false;
    // candidate c of the problem lives in the slot of the block whose prefix range holds c
    for (int c = tid; c < total; c += 256) {
        int lo = 0, hi = nb - 1;
        while (lo < hi) {  // last block with scan[jb] <= c
            const int mid = (lo + hi + 1) >> 1;
            if (L.scan[mid] <= c) lo = mid; else hi = mid - 1;
        }
        cand[c] = ld_coh_row(a.blk_rows + ((size_t)(b - a.b0) * nb + lo) * a.cap_blk + (c - L.scan[lo]));
    }
    if (tid == 0) L.ncand = total;
    __threadfence_block();
    __syncthreads();
    return true;
}

// One step of the SQP state machine after a phase.  Returns the next command (written to the control block by the caller).
//
// Every phase evaluates g AND jac at its point and scans violation, candidate rows and the finalize_solution count together
// (a phase costs ~12 us of latency whatever it computes).  So a line-search trial that is accepted already IS the new
// linearisation -- the host form evaluates the accepted point a second time, with identical results -- and the verdict of the
// current point is always at hand: a solve that takes the full step needs one phase per SQP iteration, and none at the end.
// `evals` counts what the host form would have evaluated, so both forms report the same numbers.
// (Forced inline since round 5: as a function of its own it saved and restored 169 callee-saved VGPRs through scratch memory at every call -- 338
// scratch instructions per phase, 540 B of scratch per lane -- which was a tenth of a lone solve: sample problem 0.122 -> 0.116 ms.)
template <int WPS>
__device__ __forceinline__ int leader_step(const SolveArgs& a, Leader& L, QpShared& QS, int b, long long viol_fx, int bad, SolveRow* cand,
                                  unsigned char* flags_g, double* x_pub, double* lds_stage) {
    const int n = a.tb.n, tid = threadIdx.x;
    const double viol = viol_from_fixed(viol_fx);
#define LSTAMP(k) do { if (a.stamps && tid == 0 && first) a.stamps[(size_t)b * 64 + 32 + (k)] = wall_clock64() - L.t_start; } while (0)
    // ---- the point just evaluated: x itself (first phase) or the trial point xt
    const bool first = L.stage == ST_AFTER_FIRST;
    const double* pt = first ? L.x : L.xt;
    if (tid < n) {   // the squared error of each joint (eval_f) and eval_grad_f at that point
        const bool cont = (a.continuous_mask >> tid) & 1;
        const double qp = plan_point(a, L.bz, tid, pt[tid]);
        const double eg = cont ? wrap_to_pi_dev(qp - L.qdes[tid]) : (qp - L.qdes[tid]);
        L.np[tid] = 2 * eg * plan_dk(a, L.bz, tid) * a.cost_scale;   // (np / z double as scratch between QPs)
        const double e = cont ? wrap_to_pi_dev(L.qdes[tid] - qp) : (L.qdes[tid] - qp);
        L.z[tid] = e * e;
    }
    __syncthreads();
    if (tid == 0) {
        const double fpt = cost_sum(a, L.z);
        L.stop = 0;   // 0: new SQP iteration at x (linearisation at hand); 1: finished; 3: another trial
        if (first) {
            L.f = fpt; L.evals = 1; L.viol = viol; L.bad = bad; L.status = 0;
            for (int j = 0; j < n; j++) L.gradf[j] = L.np[j];
        } else {
            // one trial of the L1-merit backtracking line search
            L.evals++;
            const double phi = fpt + L.mu * viol;
            if (phi <= L.phi0 + 1e-4 * L.alpha * L.dphi || L.ls == a.max_ls) {
                if (L.ls == a.max_ls && phi > L.phi0) { L.status = 4; L.stop = 1; }
                else {   // accepted: this phase is the linearisation at the new x
                    for (int j = 0; j < n; j++) { L.x[j] = L.xt[j]; L.gradf[j] = L.np[j]; }
                    L.f = fpt; L.viol = viol; L.bad = bad; L.iters++; L.it++;
                    L.evals++;   // (the host form's second evaluation of the accepted point)
                }
            } else {
                L.alpha *= 0.5;
                L.ls++;
                for (int j = 0; j < n; j++) L.xt[j] = L.x[j] + L.alpha * L.d[j];
                L.stop = 3;
            }
        }
        L.stage = ST_AFTER_TRIAL;
        if (L.stop == 0) {   // top of an SQP iteration (solver.hip: `for (it ...)`)
            if (L.it >= a.max_iter) { L.status = 2; L.stop = 1; }
            else if (a.budget_ticks >= 0 && wall_clock64() - L.t_start >= a.budget_ticks) { L.status = 5; L.stop = 1; }
        }
    }
    __syncthreads();
    LSTAMP(0);
    if (L.stop == 0) {
        bool finish = false;
        if (!gather_candidates(a, L, b, cand)) {
            if (tid == 0) L.status = -1;   // candidate buffers too small for this problem: the host form takes over
            finish = true;
        }
        if (!finish) {
            LSTAMP(1);
            // QP with the elastic retries of solver.hip (sigma = fraction of the violation a row may keep): the four attempts side by side,
            // one per wave, the lowest feasible one taken (solve_qp_wave)
#ifdef SOLVE_QP_SEQUENTIAL   // (development: the attempts one after the other on the whole block, rounds 2-4; WITHOUT round 6's box-clipped first try, so not the host form's iterates any more)
            for (int attempt = 0; attempt < 4; attempt++) {
                if (tid == 0) L.sigma = attempt_sigma(attempt);
                __syncthreads();
                solve_qp_device<WPS>(a, L, cand, flags_g, flags_g + (a.cap_rows + 2 * NV));
                if (L.feasible) break;
                __syncthreads();
            }
            __syncthreads();
#else
            if (tid == 0) {
                QS.first_ok = 4;
                for (int j = 0; j < NV; j++) L.invHd[j] = 1.0 / L.Hd[j];
            }
            // the candidates' normals and values once more in LDS (the tiles' scratch, idle during the leader's step), component-major: what the
            // four attempts' searches read at every step -- from global memory a step of a 2 000-row QP waited ten load latencies
            const double* staged = nullptr;
            if (L.ncand > 64 * (WPS == 1 ? 4 : 2) - 2 * n && L.ncand <= a.lds_rows) {
                for (int i = tid; i < L.ncand; i += 256) {
                    const SolveRow& r = cand[i];
#pragma unroll
                    for (int j = 0; j < NV; j++) lds_stage[(size_t)j * a.lds_rows + i] = r.a[j];
                    lds_stage[(size_t)NV * a.lds_rows + i] = r.v;
                }
                staged = lds_stage;
            }
            __syncthreads();
            solve_qp_wave<WPS>(a, L, QS, cand, staged, flags_g);
            __syncthreads();
            if (tid == 0) {
                int k = 0;
                while (k < 3 && !QS.w[k].feasible) k++;
                const QpWave& W = QS.w[k];
                for (int j = 0; j < NV; j++) L.qx[j] = W.qx[j];
                L.max_mult = W.max_mult; L.feasible = W.feasible; L.sigma = attempt_sigma(k); L.qp_iter = W.qp_iter;
                if (a.stamps) {
                    for (int e = 0; e < 4; e++) {
                        a.stamps[(size_t)b * 64 + 44 + e] += QS.w[e].qp_iter;   // steps per attempt, summed over the solves
                        if (QS.w[e].feasible && QS.w[e].qp_iter > a.stamps[(size_t)b * 64 + 48]) a.stamps[(size_t)b * 64 + 48] = QS.w[e].qp_iter;   // longest attempt that ended feasible
                        if (QS.w[e].qp_iter > 60 && !QS.w[e].feasible) a.stamps[(size_t)b * 64 + 49] += 1;   // attempts that ran long and ended infeasible
                    }
                }
            }
            __syncthreads();
#endif
            LSTAMP(2);
            if (tid == 0 && a.stamps) { if (L.it == 0) { a.stamps[(size_t)b * 64 + 40] = L.qp_iter; a.stamps[(size_t)b * 64 + 41] = L.ncand; } a.stamps[(size_t)b * 64 + 42] += L.qp_iter; a.stamps[(size_t)b * 64 + 43] += 1; }
            if (tid == 0) {
                if (!L.feasible) { L.status = 3; L.stop = 1; }
                else {
                    double dn = 0, gd = 0;
                    for (int j = 0; j < n; j++) { L.d[j] = L.qx[j]; dn = fmax(dn, fabs(L.qx[j])); gd += L.gradf[j] * L.qx[j]; }
                    if (dn <= a.tol * 1e-2 || (dn <= a.tol && L.viol <= a.tol)) { L.status = 1; L.stop = 1; }
                    else {
                        L.mu = fmax(L.mu, 1.5 * L.max_mult + 1e-3);
                        L.phi0 = L.f + L.mu * L.viol;
                        L.dphi = gd - L.mu * (1.0 - L.sigma) * L.viol;
                        if (L.dphi > -1e-14) L.dphi = -1e-14;
                        L.alpha = 1.0;
                        L.ls = 0;
                        for (int j = 0; j < n; j++) L.xt[j] = L.x[j] + L.alpha * L.d[j];
                        L.stop = 3;
                    }
                }
            }
        }
        __syncthreads();
        if (finish && tid == 0) L.stop = 1;
        __syncthreads();
    }
    int next;
    if (L.stop == 3) {   // evaluate the trial point
        // (the culled form's row lists hold for points inside the variables' box; every accepted QP step is verified against the bounds, so a trial
        //  point outside it by more than 1e-6 cannot happen -- if it ever does, the problem goes back to the host form instead of being evaluated
        //  on lists that do not cover the point)
        const int outside = a.culled && tid < n && !(fabs(L.xt[tid]) <= 1.0 + 1e-6) ? 1 : 0;
        if (__syncthreads_or(outside)) {
            if (tid == 0) a.out[b].status = -1;
            __syncthreads();
            return CMD_DONE;
        }
        if (tid < n) st_coh(x_pub + tid, L.xt[tid]);   // (one lane per component: seven stores in flight instead of one after the other)
        next = CMD_EVAL;
    } else {             // finished: finalize_solution (RT/NLPclass.cu:422-538) on the last evaluation of x
        if (tid == 0) {
            ArmourSolveResult& r = a.out[b];
            if (L.status == -1) r.status = -1;   // hand the problem back to the host form
            else {
                for (int j = 0; j < NV; j++) r.k_opt[j] = j < n ? L.x[j] : 0.0;
                r.cost = L.f;
                r.max_violation = L.viol;
                r.feasible = L.bad == 0 ? 1 : 0;
                r.iterations = L.iters;
                r.evaluations = L.evals + 1;   // (+ the host form's evaluation for the verdict)
                r.status = L.status;
                r.time_ms = (double)(wall_clock64() - L.t_start);   // ticks; the host converts
            }
        }
        next = CMD_DONE;
    }
    __syncthreads();
    return next;
}

// WPS = waves per SIMD the kernel is compiled for.  2: <= 256 registers, two blocks per CU -- twice the blocks per problem, which is what
// large batches need (B = 128, O = 20: 9.8 ms against 14.7).  1: 256 VGPRs + AGPRs as spill space, one block per CU -- the leader's QP
// runs with far fewer spills (sample problem 0.142 -> 0.137 ms, one infeasible world 0.43 -> 0.35 ms, B = 16: 1.0 -> 0.82 ms); chosen
// while one block per CU leaves a block at most 24 tiles to walk (armour_solve_device_capacity).
template <bool LL, int PPW, int WPS>
__global__ __launch_bounds__(P2_BLOCK) __attribute__((amdgpu_waves_per_eu(WPS, WPS))) void armour_solve_kernel(const SolveArgs* __restrict__ args) {
    // The arguments come through device memory, not by value: the struct holds arrays that are indexed at run time (k_range[i]), and a
    // by-value copy of it was materialised in EVERY lane's scratch -- 26 MB of writes per launch (rocprofv3 WRITE_SIZE) and scratch
    // loads in the tile loops.  Read through the pointer, its fields are uniform scalar loads.
    const SolveArgs& a = *args;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ ScanShared scan_sh;
    __shared__ Leader L;
    __shared__ QpShared QS;
    __shared__ int s_cmd;
    __shared__ double s_x[NV];
    const int tid = threadIdx.x;
    const int b_local = blockIdx.x / a.nb, jb = blockIdx.x - b_local * a.nb;
    const int b = a.b0 + b_local;
    const bool leader = jb == 0;
    const int n = a.tb.n, m = a.tb.m;
    SolveCtl* c = a.ctl + b;
    // (culled: the problem's own count of listed tiles; a block past the last tile evaluates and scans nothing and still keeps the group's barrier)
    TileSet ts = {0, 0, 0, 0};
    int my_tiles = a.n_tiles;
    if (a.culled) { ts = problem_tiles(a, b); my_tiles = ts.ntq + ts.nsp + 1; }
    const int t0 = (int)(((long long)jb * my_tiles) / a.nb), t1 = (int)(((long long)(jb + 1) * my_tiles) / a.nb);
    int role = 0, r0 = 0, r1 = 0, rdummy;
    if (!a.culled) {
        tile_info(a, t0, role, r0, rdummy);
        tile_info(a, t1 - 1, role, rdummy, r1);
    }
    SolveRow* my_rows = a.blk_rows + ((size_t)(b - a.b0) * a.nb + jb) * a.cap_blk;
    SolveRow* cand = a.qp_rows + (size_t)(b - a.b0) * a.cap_rows;
    unsigned char* flags_g = a.flags + (size_t)(b - a.b0) * 8 * (a.cap_rows + 2 * NV);   // active / excluded flags of the four QP attempts when they outgrow LDS
    double* g = a.g + (size_t)b * m;
    double* jac = a.jac + (size_t)b * m * n;
    if (leader && tid < 3 * NV) L.bz[tid] = tid < 3 * n ? a.tb.bez[(size_t)b * 3 * n + tid] : 0.0;
    if (leader && tid >= 32 && tid < 32 + NV) L.qdes[tid - 32] = tid - 32 < n ? a.q_des[(size_t)b * n + tid - 32] : 0.0;
    __syncthreads();
    if (leader && tid == 0) {
        L.t_start = wall_clock64();
        for (int j = 0; j < NV; j++) {
            L.x[j] = 0.0;   // get_starting_point: x = 0 (RT/NLPclass.cu:170-202)
            double Hd = 1e-12;
            if (j < n) {
                const double dk = plan_dk(a, L.bz, j);
                Hd = 2.0 * a.cost_scale * dk * dk;   // constant diagonal Hessian of the cost
                if (Hd < 1e-12) Hd = 1e-12;
            }
            L.Hd[j] = Hd;
        }
        L.mu = 1.0; L.it = 0; L.iters = 0; L.evals = 0; L.status = 0; L.stage = ST_AFTER_FIRST; L.ncand = 0; L.sigma = 0.0;
    }
    const double* my_x = s_x;   // (a generic pointer into LDS: the tile functions take the point through a plain `const double*`)
    unsigned phase = 0;
    // Every wait below is bounded (a.hard_ticks since this block started, set by the host from the solve's time budget): if a block of
    // the group never arrives -- a fault elsewhere, residency that does not match the occupancy query, competition for CUs -- the
    // others leave instead of spinning with the GPU, and the leader hands the problem back (status -1: the host form redoes it).
    const long long t_block = wall_clock64();
    for (;;) {
        // ---- wait until the leader has published this phase's point and command (one lane polls one word)
        if (tid == 0) {
            unsigned w;
            unsigned spins = 0;
            while (((w = ld_coh(&c->go)) >> 3) < phase) {   // go = 8 * phase + command: one word, one round trip
                __builtin_amdgcn_s_sleep(2);
                if ((++spins & 1023u) == 0 && wall_clock64() - t_block > a.hard_ticks) { w = (unsigned)CMD_DONE; break; }
            }
            asm volatile("" ::: "memory");
            s_cmd = (int)(w & 7u);
        }
        __syncthreads();
        const int cmd = s_cmd;
        if (cmd == CMD_DONE) {   // leave the flag word as the next launch expects it (nobody polls it any more: the leader has finished)
            if (tid == 0) st_coh(&a.blk_word[(size_t)(b - a.b0) * a.nb + jb].flag, 0u);
            break;
        }
        if (tid < NV) s_x[tid] = tid < n ? ld_coh(&c->x[tid]) : 0.0;   // the phase's point, into LDS: the tile code reads it from there
        __syncthreads();
        // ---- evaluate my tiles at x (the tile code of armour_eval_g_jac)
        const double kf = tid < n ? s_x[tid] : 0.0;
        for (int t = t0; t < t1; t++) {
            if (a.culled) {
                if (t < ts.ntq + ts.nsp) {   // P2_BLOCK listed torque / collision rows, one per thread (p2_sparse.h)
                    KPow& kp = *reinterpret_cast<KPow*>(smem_raw);
                    fill_kpow(kp, kf, n);
                    __syncthreads();
                    if (t < ts.ntq) {
                        const int i = t * P2_BLOCK + tid;
                        if (i < ts.cntq) sparse_torque_row(a.tb, b, a.tq_tiles[(size_t)b * a.tq_cap + i], a.lp.strideT, kp, g, jac);
                    } else {
                        const int i = (t - ts.ntq) * P2_BLOCK + tid;
                        if (i < ts.cnt2) sparse_collision_row<true>(a.tb, a.sl, b, i, kp, g, jac);
                    }
                    __syncthreads();
                    continue;
                }
                role = a.lp.nbc + a.lp.nbt;
            } else {
                int tr0, tr1;
                tile_info(a, t, role, tr0, tr1);
            }
            if (role < a.lp.nbc) {
                // d = A.c recomputed from the obstacle centres where the launch of armour_eval_g_jac does the same (batches of >= 8
                // problems, whose tables hold no d column: api.hip armour_make_tables) -- the same rows bit for bit either way
                if constexpr (LL) {
                    if (a.tb.obs_center) collision_block<true, true, false, true, LL, PPW, false>(a.tb, a.lp, b, role, my_x, kf, g, jac, smem_raw);
                    else collision_block<true, true, false, false, LL, PPW, false>(a.tb, a.lp, b, role, my_x, kf, g, jac, smem_raw);
                } else collision_block<true, true, false, false, LL, PPW, false>(a.tb, a.lp, b, role, my_x, kf, g, jac, smem_raw);
            }
            else if (role < a.lp.nbc + a.lp.nbt) torque_block<true, true, false, false, LL, PPW, false>(a.tb, a.lp, b, role, my_x, kf, g, jac, smem_raw);
            else limit_block<true, true, false, false, LL, PPW, false>(a.tb, a.lp, b, role, my_x, kf, g, jac, smem_raw);
            __syncthreads();
        }
        // ---- scan my rows.  They were written by this block: same CU, same L1 -- visible after the barrier, no fence needed
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __syncthreads();
        long long vfx = 0;
        int cnt = 0, bad = 0;
        if (a.culled) scan_tiles(a, b, ts, t0, t1, my_rows, a.cap_blk, scan_sh, vfx, cnt, bad);
        else scan_rows<3>(a, b, r0, r1, my_rows, a.cap_blk, scan_sh, vfx, cnt, bad);
        BlockWord* mine = a.blk_word + (size_t)(b - a.b0) * a.nb + jb;
        if (tid == 0) { st_coh(&mine->viol, vfx); st_coh(&mine->bad, bad); st_coh(&mine->count, cnt); }
        // ---- group barrier: a flag per block (199 atomics on one counter cost ~10 us per phase; 199 separate words cost nothing).
        //      Every lane's coherent stores (candidate rows, the words above) have completed before the flag is written.
        wait_my_memory_ops();
        __syncthreads();
        phase++;
        if (tid == 0) st_coh(&mine->flag, phase);
        if (!leader) continue;
        bool group_lost = false;
        for (unsigned spins = 0;; spins++) {   // the leader's 256 lanes poll the group's flags, four each
            int all = 1;
#pragma unroll
            for (int e = 0; e < 4; e++) { const int j2 = tid * 4 + e; if (j2 < a.nb && ld_coh(&a.blk_word[(size_t)(b - a.b0) * a.nb + j2].flag) < phase) all = 0; }
            if (__syncthreads_and(all)) break;
            if ((spins & 255u) == 255u && __syncthreads_or(wall_clock64() - t_block > a.hard_ticks ? 1 : 0)) { group_lost = true; break; }
        }
        asm volatile("" ::: "memory");
        if (group_lost) {   // a block of this group never arrived: release the ones that did and hand the problem back to the host form
            if (tid == 0) {
                a.out[b].status = -1;
                wait_my_memory_ops();
                st_coh(&c->go, phase * 8u + (unsigned)CMD_DONE);
                st_coh(&a.blk_word[(size_t)(b - a.b0) * a.nb + jb].flag, 0u);
            }
            break;
        }
        // the phase's L1 violation and verdict count: integer sums of the blocks' words (any order gives the same number)
        long long viol_fx = 0;
        int nbad = 0;
        {
            long long v = 0;
            int bd = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) { const int j2 = tid * 4 + e; if (j2 < a.nb) { v += ld_coh(&a.blk_word[(size_t)(b - a.b0) * a.nb + j2].viol); bd += ld_coh(&a.blk_word[(size_t)(b - a.b0) * a.nb + j2].bad); } }
            block_sum(v, bd, scan_sh.red, scan_sh.redi);
            viol_fx = v; nbad = bd;
            __syncthreads();
        }
        if (a.stamps && tid == 0 && phase <= 16) a.stamps[(size_t)b * 64 + 2 * (phase - 1)] = wall_clock64() - L.t_start;      // barrier passed
        const int next = leader_step<WPS>(a, L, QS, b, viol_fx, nbad, cand, flags_g, c->x, reinterpret_cast<double*>(smem_raw));
        if (a.stamps && tid == 0 && phase <= 16) a.stamps[(size_t)b * 64 + 2 * (phase - 1) + 1] = wall_clock64() - L.t_start;  // leader step done
        wait_my_memory_ops();                              // every lane's stores of x are at the memory side ...
        __syncthreads();
        if (tid == 0) st_coh(&c->go, phase * 8u + (unsigned)next);   // ... before the release (phase and command in one word) is
        __syncthreads();
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ host side
int armour_solve_device_capacity(const P2Tables& tb, int max_link, int max_torque, const unsigned long long* h_skip, int device, int b_launch, int waves_per_simd, SolvePlan* plan,
                                 int n_tiles_override) {
    P2Launch lp;
    size_t smem = 0;
    bool dfc, six, exact;
    int rc = armour_p2_plan(tb, max_link, max_torque, h_skip, 1, 0, 0, 0, &lp, &smem, &dfc, &six, &exact);
    if (rc != ARMOUR_OK) return rc;
    plan->lp = lp; plan->smem = smem; plan->six = six;
    plan->n_tiles = n_tiles_override > 0 ? n_tiles_override : lp.nbt + lp.nbc + 1;
    int coop = 0;
    HIPCHK(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, device));
    if (!coop) { plan->capacity = 0; return ARMOUR_OK; }
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    // one block per CU (the 512-register build) while that still leaves a block at most 24 tiles to walk; otherwise two per CU
    const void* fn1 = !tb.ll_shared ? (const void*)armour_solve_kernel<false, 9, 1> : six ? (const void*)armour_solve_kernel<true, 6, 1> : (const void*)armour_solve_kernel<true, 9, 1>;
    const void* fn2 = !tb.ll_shared ? (const void*)armour_solve_kernel<false, 9, 2> : six ? (const void*)armour_solve_kernel<true, 6, 2> : (const void*)armour_solve_kernel<true, 9, 2>;
    const int wps_env = waves_per_simd;   // ARMOUR_OPT_SOLVE_WAVES_PER_SIMD (0: automatic)
    // dynamic LDS: the tiles' scratch, and during the leader's step the staged candidate rows of the QP (8 (NV + 1) bytes a row): as much as the
    // occupancy leaves -- one block per CU: 96 KB (1 365 rows), two: 40 KB (568 rows); the kernel holds 29 KB of its own
    const size_t smem1 = std::max(smem, (size_t)96 * 1024), smem2 = std::max(smem, (size_t)40 * 1024);
    int per_cu = 0;
    {   // (per kernel function and device, once: the attribute call is a driver round trip and a lone solve is 0.14 ms)
        static std::atomic<unsigned long long> set1[3], set2[3];
        const int which = !tb.ll_shared ? 0 : six ? 1 : 2;
        const unsigned long long bit = 1ull << (device & 63);
        if (smem1 > 96 * 1024 || !(set1[which].load() & bit)) { HIPCHK(hipFuncSetAttribute(fn1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem1)); set1[which].fetch_or(bit); }
        if (smem2 > 40 * 1024 || !(set2[which].load() & bit)) { HIPCHK(hipFuncSetAttribute(fn2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2)); set2[which].fetch_or(bit); }
    }
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn1, P2_BLOCK, smem1));
    const int cap1 = per_cu * prop.multiProcessorCount;
    // (B = 16, O = 20: 16 blocks of 20 tiles per problem in the one-per-CU build: 0.83 ms; 32 blocks of 10 tiles in the other: 1.00 ms -- the
    //  leader's serial QP weighs more than the tiles; at B = 128 it is the other way round: 14.7 against 9.8 ms)
    const int blocks1 = cap1 / std::max(1, b_launch);
    // (the culled form's tiles are few and cheap -- what a phase costs there is the leader's QP: one block per CU whenever the batch fits;
    //  B = 128, O = 50: 5.6 against 6.7 ms, B = 64, O = 20: 4.4 against 5.9)
    const bool one = wps_env ? wps_env == 1 : (blocks1 >= 1 && (n_tiles_override > 0 || (plan->n_tiles + blocks1 - 1) / blocks1 <= 24));
    const void* fn = one ? fn1 : fn2;
    plan->fn = fn;
    plan->smem = one ? smem1 : smem2;
    if (!one) HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, P2_BLOCK, smem2));
    plan->capacity = per_cu * prop.multiProcessorCount;
    int rate_khz = 0;
    HIPCHK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, device));
    plan->ticks_per_ms = rate_khz > 0 ? (double)rate_khz : 100000.0;
    return ARMOUR_OK;
}

int armour_solve_device_launch(const SolveArgs* d_args, int nb, const SolvePlan& plan, int B, hipStream_t stream) {
    const SolveArgs* p = d_args;
    void* params[1] = {&p};
    HIPCHK(hipLaunchCooperativeKernel(plan.fn, dim3((unsigned)B * (unsigned)nb), dim3(P2_BLOCK), params, (unsigned)plan.smem, stream));
    return ARMOUR_OK;
}
