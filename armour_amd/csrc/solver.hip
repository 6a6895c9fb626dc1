// armour_solve: the NLP solve of one planning iteration on top of the device callbacks.
//
// Replaces, for this library, what the reference delegates to IPOPT (RT/armour_main.cu:237-304: OptimizeTNLP with
// tol 1e-4, a wall-time limit, L-BFGS Hessian, start x = 0) and armtd_NLP::finalize_solution
// (RT/NLPclass.cu:422-538).  IPOPT / HSL are not available here, so this is a from-scratch solver sized for the
// problem ARMOUR actually poses: n = 7 variables in [-1,1], an exactly quadratic cost (q(t_plan) is affine in k, so
// the Hessian is the constant diagonal 2*scale*(dq/dk)^2) and m ~ 1e4 inequality rows whose values and dense Jacobian
// come from ONE fused device launch per iterate for all B problems of the handle:
//
//   SQP:  d = argmin grad_f.d + 1/2 d'Hd   s.t.  g_l <= g + J d <= g_u,  -1 <= x + d <= 1      (dense QP, n = 7)
//         L1-merit backtracking line search (trial points need eval_g only), start x = 0
//   QP:   Goldfarb-Idnani dual active-set method; with n <= 7 the operators H = G^-1(I - N N*) and
//         N* = (N'G^-1N)^-1 N'G^-1 are re-formed from the <= n active normals at every step instead of being
//         updated by Givens rotations.
//
// g and jac stay on the device: after each evaluation armour_solve_scan_kernel (below) hands the host the L1 violation,
// the rows the QP can see and, at the end, finalize_solution's verdict.  The 7-variable QPs run on the host.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "bezier.h"
#include "common.h"
#include "solver_common.h"
#include "solver_device.h"

using slv::NV;
using slv::SolveRow;

namespace {

constexpr double kInf = 1e300;

// ---- tiny dense helpers (q <= NV) ----
// solve M y = rhs for symmetric positive definite M (q x q, row-major NV stride) by Cholesky; false if not SPD
// Divisions are taken once per pivot: every other use multiplies by the stored reciprocal (the device form runs this chain on one
// wave, where a dependent fp64 division costs ~40x a multiplication; both forms use the same expressions, so they agree bit for bit).
bool spd_solve(const double* M, int q, const double* rhs, double* y) {
    double L[NV][NV], inv[NV];
    for (int i = 0; i < q; i++)
        for (int j = 0; j <= i; j++) {
            double s = M[i * NV + j];
            for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k];
            if (i == j) {
                if (s <= 1e-14 * std::fabs(M[i * NV + i]) || s <= 0) return false;
                L[i][i] = std::sqrt(s);
                inv[i] = 1.0 / L[i][i];
            } else {
                L[i][j] = s * inv[j];
            }
        }
    double t[NV];
    for (int i = 0; i < q; i++) {
        double s = rhs[i];
        for (int k = 0; k < i; k++) s -= L[i][k] * t[k];
        t[i] = s * inv[i];
    }
    for (int i = q - 1; i >= 0; i--) {
        double s = t[i];
        for (int k = i + 1; k < q; k++) s -= L[k][i] * y[k];
        y[i] = s * inv[i];
    }
    return true;
}

// One inequality a'x >= b of the QP.  Rows come from three sources: upper sides (-J_i d >= g_i - hi_i), lower sides
// (J_i d >= lo_i - g_i) and the variable bounds.
struct QpRow {
    double a[NV];
    double b;
};

struct QpResult {
    double x[NV];
    double max_mult;  // largest multiplier (for the merit penalty)
    int iterations;
    bool feasible;
};

// Goldfarb-Idnani for  min 1/2 x'Gx + g0'x  s.t.  a_i'x >= b_i,  G = diag(Gd) > 0.
// box_rows = 2 n: the LAST 2 n rows are the variables' box, lower then upper bound of variable 0, 1, ... (what armour_solve appends); 0: no such promise.
QpResult solve_qp(int n, const double* Gd, const double* g0, const std::vector<QpRow>& rows, int max_iter = slv::kQpMaxSteps, int box_rows = 0) {
    QpResult res;
    res.iterations = 0; res.feasible = true; res.max_mult = 0;
    double x[NV], invG[NV];
    for (int j = 0; j < n; j++) { invG[j] = 1.0 / Gd[j]; x[j] = -g0[j] * invG[j]; }
    if (box_rows == 2 * n && (int)rows.size() >= box_rows) {   // the box-clipped minimiser first (solver_common.h)
        const int m0 = (int)rows.size() - box_rows;
        double lo[NV], hi[NV], d[NV];
        for (int j = 0; j < n; j++) { lo[j] = rows[m0 + 2 * j].b; hi[j] = -rows[m0 + 2 * j + 1].b; }
        const double mm = slv::box_clipped_step(n, Gd, invG, g0, lo, hi, d);
        bool ok = true;
        for (int i = 0; i < (int)rows.size() && ok; i++) {
            double s = -rows[i].b;
            for (int j = 0; j < n; j++) s += rows[i].a[j] * d[j];
            if (s < -1e-7) ok = false;
        }
        if (ok) {
            for (int j = 0; j < n; j++) res.x[j] = d[j];
            res.max_mult = mm;
            return res;
        }
    }
    int A[NV];       // active row ids
    double u[NV + 1];  // multipliers of the active rows (+ the entering one)
    int q = 0;
    const int m = (int)rows.size();
    std::vector<char> is_active(m, 0), excluded(m, 0);
    for (;;) {
        // most violated inactive row
        int p = -1;
        double worst = -1e-10;
        for (int i = 0; i < m; i++) {
            if (is_active[i] || excluded[i]) continue;
            double s = -rows[i].b;
            for (int j = 0; j < n; j++) s += rows[i].a[j] * x[j];
            // scale-free test: violation relative to the row norm
            if (s < worst) { worst = s; p = i; }
        }
        if (p < 0) break;
        if (++res.iterations > max_iter) { res.feasible = false; break; }
        const double* np = rows[p].a;
        double up = 0.0;
        bool added = false;
        for (int guard = 0; guard < 4 * NV + 8 && !added; guard++) {
            // r = N* np,  z = G^-1 (np - N r)
            double r[NV] = {0}, z[NV];
            if (q > 0) {
                double M[NV * NV], rhs[NV];
                for (int i = 0; i < q; i++) {
                    for (int k = 0; k <= i; k++) {
                        double s = 0;
                        for (int j = 0; j < n; j++) s += rows[A[i]].a[j] * rows[A[k]].a[j] * invG[j];
                        M[i * NV + k] = s; M[k * NV + i] = s;
                    }
                    double s = 0;
                    for (int j = 0; j < n; j++) s += rows[A[i]].a[j] * np[j] * invG[j];
                    rhs[i] = s;
                }
                if (!spd_solve(M, q, rhs, r)) { excluded[p] = 1; break; }  // dependent active set: skip this row
            }
            double zz = 0, znp = 0;
            for (int j = 0; j < n; j++) {
                double s = np[j];
                for (int i = 0; i < q; i++) s -= rows[A[i]].a[j] * r[i];
                z[j] = s * invG[j];
                zz += z[j] * z[j];
                znp += z[j] * np[j];
            }
            // step lengths
            double t1 = kInf;
            int l = -1;
            for (int i = 0; i < q; i++)
                if (r[i] > 1e-14 && u[i] / r[i] < t1) { t1 = u[i] / r[i]; l = i; }
            double sp = -rows[p].b;
            for (int j = 0; j < n; j++) sp += np[j] * x[j];
            double t2 = kInf;
            if (zz > 1e-24 && znp > 1e-16) t2 = -sp / znp;
            if (t2 < 0) t2 = 0;
            const double t = t1 < t2 ? t1 : t2;
            if (t >= kInf) { res.feasible = false; break; }
            if (t2 >= kInf) {  // dual step only, drop the blocking row
                for (int i = 0; i < q; i++) u[i] -= t * r[i];
                up += t;
                is_active[A[l]] = 0;
                for (int i = l; i < q - 1; i++) { A[i] = A[i + 1]; u[i] = u[i + 1]; }
                q--;
                continue;
            }
            for (int j = 0; j < n; j++) x[j] += t * z[j];
            for (int i = 0; i < q; i++) u[i] -= t * r[i];
            up += t;
            if (t == t2) {  // full step: the row becomes active
                if (q >= n) { res.feasible = false; break; }
                A[q] = p; u[q] = up; q++;
                is_active[p] = 1;
                added = true;
            } else {        // partial step: drop the blocking row and try again
                is_active[A[l]] = 0;
                for (int i = l; i < q - 1; i++) { A[i] = A[i + 1]; u[i] = u[i + 1]; }
                q--;
            }
        }
        if (!res.feasible) break;
        if (!added && !excluded[p]) { excluded[p] = 1; }  // could not make progress on this row
    }
    for (int j = 0; j < n; j++) res.x[j] = x[j];
    for (int i = 0; i < q; i++) if (u[i] > res.max_mult) res.max_mult = u[i];
    // The result is VERIFIED against every row, active ones included.  Excluded rows that remain violated mean the linearisation is inconsistent;
    // and an ACTIVE row can have drifted: with an ill-conditioned M = N'G^-1 N (cost Hessian entries four orders of magnitude apart, near-parallel
    // collision normals) z is not exactly in the null space of the active normals, the rows that are "satisfied with equality" are no longer, and
    // nothing above looks at them again.  Round 5 met a step that broke a variable's bound by 1.6 that way and was taken as feasible (a trajectory
    // parameter of -2.56 in a box of +-1: tools/dev/solve_stress.py, profiles/r05_solve_batches.txt).  A result that fails the check is infeasible:
    // the next elastic attempt, or the end of the solve -- so every accepted step keeps x + d inside the box to 1e-7.
    for (int i = 0; i < m && res.feasible; i++) {
        double s = -rows[i].b;
        for (int j = 0; j < n; j++) s += rows[i].a[j] * x[j];
        if (s < -1e-7) res.feasible = false;
    }
    return res;
}

struct ProblemState {
    double x[NV], f, gradf[NV], viol, mu;
    int iters = 0, evals = 0, status = 0;  // 0 running, 1 converged, 2 max iterations, 3 QP infeasible, 4 line search failed, 5 time limit
    bool done = false;
};

// ---- device-side scan of one evaluation: what the host QP needs instead of the whole g / jac ----
// One block per problem walks the rows in order and leaves (a) the L1 violation of g_l <= g <= g_u and, with ROWS,
// (b) the rows the QP has to see -- those that can become active for some |d|_inf <= 2, the filter of the host loop
// below -- compacted IN ROW ORDER (upper side before lower side of a row), each as {row, side, v = violation-signed
// distance to the bound, a = -+J_i}.  A problem with 20 obstacles moves a few hundred such rows instead of 0.94 MB.
constexpr int kScanRowsPerBlock = 2048;

// MODE 0: violation only (line-search trials); 1: violation + candidate rows (new linearisation); 2: violation + number of
// rows outside [g_l - slack, g_u + slack] with the slacks of armtd_NLP::finalize_solution (RT/NLPclass.cu:422-538:
// torque rows 1e-2, collision rows 1e-4, limit rows 0) -- the verdict armour_check_feasible gives on the host.
template <int MODE>
__global__ __launch_bounds__(256) void armour_solve_scan_kernel(int m, int n, int n_torque_rows, int n_collision_rows, int n_unchecked_rows, double torque_slack, double collision_slack, const double* __restrict__ g_all, const double* __restrict__ jac_all,
                                                                const double* __restrict__ lo_all, const double* __restrict__ hi_all, int cap,
                                                                long long* __restrict__ viol_out, int* __restrict__ count_out, SolveRow* __restrict__ rows_out) {
    // grid (segments, B): a block owns kScanRowsPerBlock consecutive rows and its own slice of the outputs; the host
    // concatenates the slices in segment order
    const int b = blockIdx.y, seg = blockIdx.x, nseg = gridDim.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double* g = g_all + (size_t)b * m;
    constexpr bool ROWS = MODE == 1;
    const double* jac = ROWS ? jac_all + (size_t)b * m * n : nullptr;
    const double* lo = lo_all + (size_t)b * m;
    const double* hi = hi_all + (size_t)b * m;
    SolveRow* rows = ROWS ? rows_out + ((size_t)b * nseg + seg) * cap : nullptr;
    const int row_end = min(m, (seg + 1) * kScanRowsPerBlock);
    __shared__ int wave_tot[4];
    __shared__ long long red[256];
    int base = 0, bad = 0;
    long long vsum = 0;   // L1 violation in 2^-32 fixed point: the sum does not depend on the reduction order (solver_common.h)
    for (int i0 = seg * kScanRowsPerBlock; i0 < row_end; i0 += 256) {
        const int i = i0 + tid;
        const bool in = i < row_end;
        const double gi = in ? g[i] : 0.0, li = in ? lo[i] : -1e300, ui = in ? hi[i] : 1e300;
        vsum += slv::row_violation(gi, li, ui);
        if (MODE == 2 && in) {
            // (ARMTD mode: the re-check of CMP/NLPclass.cu:391-402 skips the collision rows of the last links)
            const int ic = i - n_torque_rows - n_collision_rows;
            const double slack = i < n_torque_rows ? torque_slack : ic < 0 ? collision_slack : 0.0;
            if ((ic < 0 || ic >= n_unchecked_rows) && (gi < li - slack || gi > ui + slack)) bad++;
        }
        if (ROWS) {
            double J[NV], l1 = 0.0;
#pragma unroll
            for (int j = 0; j < NV; j++) { J[j] = (in && j < n) ? jac[(size_t)i * n + j] : 0.0; l1 += fabs(J[j]); }
            const bool fh = in && slv::row_upper_candidate(gi, ui, l1);
            const bool fl = in && slv::row_lower_candidate(gi, li, l1);
            const unsigned long long bh = __ballot(fh), bl = __ballot(fl);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (lane == 0) wave_tot[wv] = __popcll(bh) + __popcll(bl);
            __syncthreads();
            int pos = base + __popcll(bh & below) + __popcll(bl & below);
            for (int w2 = 0; w2 < wv; w2++) pos += wave_tot[w2];
            if (fh) {
                if (pos < cap) {
                    SolveRow r; r.idx = i; r.side = 0; r.v = gi - ui;
#pragma unroll
                    for (int j = 0; j < NV; j++) r.a[j] = -J[j];
                    rows[pos] = r;
                }
                pos++;
            }
            if (fl && pos < cap) {
                SolveRow r; r.idx = i; r.side = 1; r.v = li - gi;
#pragma unroll
                for (int j = 0; j < NV; j++) r.a[j] = J[j];
                rows[pos] = r;
            }
            base += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
            __syncthreads();
        }
    }
    red[tid] = vsum;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (tid < s2) red[tid] += red[tid + s2];
        __syncthreads();
    }
    if (MODE == 2) {
        __shared__ int bad_tot;
        if (tid == 0) bad_tot = 0;
        __syncthreads();
        if (bad) atomicAdd(&bad_tot, bad);
        __syncthreads();
        base = bad_tot;
    }
    if (tid == 0) { viol_out[(size_t)b * nseg + seg] = red[0]; if (MODE != 0) count_out[(size_t)b * nseg + seg] = base; }
}

}  // namespace

extern "C" void armour_solve_options_default(ArmourSolveOptions* o) {
    memset(o, 0, sizeof(*o));
    o->max_iterations = 60;
    o->max_line_search = 12;
    o->tolerance = 1e-4;       // IPOPT_OPTIMIZATION_TOLERANCE, RT/Parameters.h:50
    o->max_wall_time_s = 0.0;  // 0 = no limit (the reference passes 0.5 s - t(P1) - 0.05 s, RT/armour_main.cu:227-229)
}

// test hook: the QP solver alone.  rows: lo <= A x <= hi (|bound| >= 1e18 = absent).  Returns 0 / ARMOUR_EINVAL; feasible flag in *feasible.
extern "C" int armour_debug_qp(int32_t n, const double* Gd, const double* g0, int32_t m, const double* Amat, const double* lo,
                               const double* hi, double* x, int32_t* feasible) {
    if (n < 1 || n > NV || m < 0) { armour_set_error("armour_debug_qp: bad size"); return ARMOUR_EINVAL; }
    std::vector<QpRow> rows;
    for (int i = 0; i < m; i++) {
        if (hi[i] < 1e18) { QpRow r; for (int j = 0; j < n; j++) r.a[j] = -Amat[i * n + j]; r.b = -hi[i]; rows.push_back(r); }
        if (lo[i] > -1e18) { QpRow r; for (int j = 0; j < n; j++) r.a[j] = Amat[i * n + j]; r.b = lo[i]; rows.push_back(r); }
    }
    const QpResult q = solve_qp(n, Gd, g0, rows);
    for (int j = 0; j < n; j++) x[j] = q.x[j];
    *feasible = q.feasible ? 1 : 0;
    return ARMOUR_OK;
}

// test hook: the QP as armour_solve poses it -- the rows above, then the variables' box x_lo <= x <= x_hi as 2 n trailing rows -- with (first_try = 1)
// or without (0) the box-clipped first try of solver_common.h.  steps = active-set steps taken (0: the clipped point passed every row).
extern "C" int armour_debug_qp_box(int32_t n, const double* Gd, const double* g0, int32_t m, const double* Amat, const double* lo, const double* hi,
                                   const double* x_lo, const double* x_hi, int32_t first_try, double* x, int32_t* feasible, int32_t* steps, double* max_mult) {
    if (n < 1 || n > NV || m < 0 || !x_lo || !x_hi) { armour_set_error("armour_debug_qp_box: bad argument"); return ARMOUR_EINVAL; }
    std::vector<QpRow> rows;
    for (int i = 0; i < m; i++) {
        if (hi[i] < 1e18) { QpRow r; for (int j = 0; j < n; j++) r.a[j] = -Amat[i * n + j]; r.b = -hi[i]; rows.push_back(r); }
        if (lo[i] > -1e18) { QpRow r; for (int j = 0; j < n; j++) r.a[j] = Amat[i * n + j]; r.b = lo[i]; rows.push_back(r); }
    }
    for (int j = 0; j < n; j++) {
        QpRow r; memset(&r, 0, sizeof(r)); r.a[j] = 1.0; r.b = x_lo[j]; rows.push_back(r);
        QpRow r2; memset(&r2, 0, sizeof(r2)); r2.a[j] = -1.0; r2.b = -x_hi[j]; rows.push_back(r2);
    }
    const QpResult q = solve_qp(n, Gd, g0, rows, slv::kQpMaxSteps, first_try ? 2 * n : 0);
    for (int j = 0; j < n; j++) x[j] = q.x[j];
    *feasible = q.feasible ? 1 : 0;
    if (steps) *steps = q.iterations;
    if (max_mult) *max_mult = q.max_mult;
    return ARMOUR_OK;
}

// ---- the device-resident form (solver_device.hip): one persistent launch runs every problem's SQP to the end ----
// Returns 1 when it produced the results, 0 when the caller must use the host form (no cooperative launch, batch larger than
// the co-resident grid, or a problem's candidate rows outgrew the device buffers), < 0 on error.
template <class Tp>
static int grow_dev(Tp** p, size_t* cap, size_t need) {
    if (*p && need <= *cap) return ARMOUR_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    HIPCHK(hipMalloc((void**)p, (need ? need : 1) * sizeof(Tp)));
    *cap = need;
    return ARMOUR_OK;
}

static int solve_on_device(ArmourPlanner* h, const ArmourSolveOptions& opt, ArmourSolveResult* results, std::chrono::steady_clock::time_point t_begin) {
    const int B = h->B, n = h->n, m = h->m;
    const P2Tables tb = armour_make_tables(h);
    SolvePlan plan;
    int rc;
    // The culled form (ARMOUR_OPT_SOLVE_CULL; relevance.hip, solver_device.h): the kernel walks only the rows that can pass the candidate filter
    // for some k.  Same iterates; the lists cost one pass over the half-space table per problem set (about one evaluation) and 0.2 ms of host
    // round trips, so automatic = from kSolveCullMinRows collision rows in the batch on.
    const int cull_opt = h->tune(ARMOUR_OPT_SOLVE_CULL);
    constexpr long long kSolveCullMinRows = 150000;   // (first solve after armour_set_problems, lists included, O = 20: B = 8 1.43 against 1.44 ms, B = 16 1.61 against 1.98, B = 64 2.2 against 4.5)
    const bool culled = h->Q > 0 && (cull_opt > 0 || (cull_opt < 0 && (long long)B * h->Q >= kSolveCullMinRows));
    p2::SparseList sl = {nullptr, nullptr, nullptr, nullptr};
    const int* tq_tiles = nullptr; const int* tq_count = nullptr;
    int tq_cap = 0, n_tiles_culled = 0;
    if (culled) {
        if ((rc = armour_solver_lists(h, &sl, &tq_tiles, &tq_count, &tq_cap)) != ARMOUR_OK) return rc;
        for (int b = 0; b < B; b++) n_tiles_culled = std::max(n_tiles_culled, (h->h_rel2_tq_count[b] + P2_BLOCK - 1) / P2_BLOCK + (h->h_rel2_count[b] + P2_BLOCK - 1) / P2_BLOCK + 1);
    }
    if ((rc = armour_solve_device_capacity(tb, h->max_link, h->max_torque, h->h_plane_skip.data(), h->device, B, h->tune(ARMOUR_OPT_SOLVE_WAVES_PER_SIMD), &plan, n_tiles_culled)) != ARMOUR_OK) return rc;
    if (plan.capacity < 1) return 0;
    // A block walks its tiles one after the other (~4 us each), so what a phase costs is the tiles PER BLOCK, and the co-resident
    // grid has to be shared by the problems of a launch.  Round 2 ran the whole batch in one launch and fell back to the host form
    // where that left a block more than 100 tiles (B = 128 at O = 50: 4 blocks per problem, 159 tiles each, 50 ms against the host
    // form's 30).  Now such a batch is cut into SUB-BATCHES launched back to back on the handle's stream -- each a persistent
    // cooperative launch of its own over problems [b0, b0 + Bs) of the same tables, with enough blocks per problem for about
    // `sub_tiles` tiles per block; iterates and results do not depend on the cut (a problem's blocks see only that problem).
    const int sub_tiles = std::max(1, h->tune(ARMOUR_OPT_SOLVE_SUB_TILES));   // default 48: tiles per block aimed at (swept on B = 64 ... 256, O = 20 ... 50: profiles/r03_solve_subtiles.txt)
    // When to cut (profiles/r03_solve_subtiles.txt, swept with candidate buffers large enough never to overflow): one launch is best or within
    // 3 % up to ~154 tiles per block (B = 128 at O = 20 / 30 / 40: 10.2 / 15.0 / 13.3 ms; B = 256 at O = 20: 19.7 against 24.9 cut); at 159
    // (B = 128, O = 50) the cut wins, 14.3 against 18.7 ms.  The threshold sits between those two measurements.
    const int cut_tiles = std::max(1, h->tune(ARMOUR_OPT_SOLVE_CUT_TILES));   // default 156
    int Bs = B;
    if (plan.capacity < B || (plan.n_tiles + std::max(1, plan.capacity / B) - 1) / std::max(1, plan.capacity / B) > cut_tiles) {
        const int nb_want = std::min(plan.n_tiles, (plan.n_tiles + sub_tiles - 1) / sub_tiles);
        Bs = std::max(1, std::min(B, plan.capacity / std::max(1, nb_want)));
        if ((rc = armour_solve_device_capacity(tb, h->max_link, h->max_torque, h->h_plane_skip.data(), h->device, Bs, h->tune(ARMOUR_OPT_SOLVE_WAVES_PER_SIMD), &plan, n_tiles_culled)) != ARMOUR_OK) return rc;
        Bs = std::max(1, std::min(Bs, plan.capacity));
    }
    int nb = std::min(std::min(plan.n_tiles, plan.capacity / Bs), 1024);
    if (h->tune(ARMOUR_OPT_SOLVE_BLOCKS) > 0) nb = std::max(1, std::min(nb, h->tune(ARMOUR_OPT_SOLVE_BLOCKS)));          // tuning / tests
    if (h->tune(ARMOUR_OPT_SOLVE_SUB_BATCH) > 0) Bs = std::max(1, std::min(Bs, h->tune(ARMOUR_OPT_SOLVE_SUB_BATCH)));    // tuning / tests: force the cut
    const int n_launch = (B + Bs - 1) / Bs;
    if ((rc = armour_upload_bounds(h)) != ARMOUR_OK) return rc;
    // rows a block owns: at most ceil(n_tiles / nb) tiles of <= 64 rows, two candidates per row
    const int tiles_per_block = (plan.n_tiles + nb - 1) / nb;
    // (an overflow of either sends the whole batch to the host-driven form, i.e. costs a second solve: B = 128 at O = 20 with 77 tiles per
    //  block overflowed 1024 rows per block on two of four world seeds -- 25 ms instead of 10; the buffers are 72 B per row)
    int cap_blk = std::min(2 * tiles_per_block * (culled ? P2_BLOCK : 64), 4096);
    int cap_rows = std::min(2 * m, 16384);
    if (h->tune(ARMOUR_OPT_SOLVE_ROW_CAP) > 0) cap_blk = std::max(1, std::min(cap_blk, h->tune(ARMOUR_OPT_SOLVE_ROW_CAP)));  // tests: force the overflow fallback
    SolveDeviceWork& w = h->solve_dev;
    // everything the host hands the kernel sits in ONE device block, filled by ONE copy from its page-locked mirror: the per-problem
    // control words, the goals, the argument block (three copies and a fill ahead of the launch cost ~10 us of a 150 us solve)
    const size_t off_qdes = ((size_t)B * sizeof(SolveCtl) + 255) & ~(size_t)255;
    const size_t off_args = (off_qdes + (size_t)B * n * sizeof(double) + 255) & ~(size_t)255;
    const size_t args_stride = (sizeof(SolveArgs) + 255) & ~(size_t)255;
    const size_t block_bytes = off_args + (size_t)n_launch * args_stride;
    if ((rc = grow_dev(&w.ctl, &w.ctl_cap, block_bytes)) != ARMOUR_OK) return rc;
    // the blocks' flag words: zero before a launch.  Every block clears its own word when it leaves the kernel, so a fill is only
    // needed when the array is new (or grew)
    {
        // (sized for ONE sub-batch: the launches of a cut batch run one after the other on the stream and index these buffers by
        //  b - b0, so a batch of 1024 problems holds the candidate rows of the <= Bs it solves at a time -- ADVICE round 3)
        const size_t need = (size_t)Bs * nb * sizeof(BlockWord);
        const size_t had = w.word_cap;
        if ((rc = grow_dev(&w.blk_word, &w.word_cap, need)) != ARMOUR_OK) return rc;
        if (w.word_cap != had || !w.words_clean) HIPCHK(hipMemsetAsync(w.blk_word, 0, w.word_cap, h->stream));
        w.words_clean = 0;   // until this launch has been seen to end
    }
    if ((rc = grow_dev(&w.blk_rows, &w.blk_rows_cap, (size_t)Bs * nb * cap_blk * sizeof(SolveRow))) != ARMOUR_OK) return rc;
    if ((rc = grow_dev(&w.qp_rows, &w.qp_rows_cap, (size_t)Bs * cap_rows * sizeof(SolveRow))) != ARMOUR_OK) return rc;
    if ((rc = grow_dev(&w.flags, &w.flags_cap, (size_t)Bs * 8 * (cap_rows + 2 * NV))) != ARMOUR_OK) return rc;   // (active / excluded) x the four QP attempts
    ArmourSolveResult* hres = reinterpret_cast<ArmourSolveResult*>(armour_handle_pinned(h, 1, (size_t)B * sizeof(ArmourSolveResult)));
    unsigned char* hblock = reinterpret_cast<unsigned char*>(armour_handle_pinned(h, 2, block_bytes));
    if (!hres || !hblock) return ARMOUR_EDEVICE;
    SolveCtl* hctl = reinterpret_cast<SolveCtl*>(hblock);
    memset(hres, 0, (size_t)B * sizeof(ArmourSolveResult));
    memset(hctl, 0, (size_t)B * sizeof(SolveCtl));
    for (int b = 0; b < B; b++) { hctl[b].go = 1; /* phase 0: CMD_EVAL_GJ at x = 0 */ hres[b].status = -2; }
    memcpy(hblock + off_qdes, h->h_qdes.data(), (size_t)B * n * sizeof(double));
    SolveArgs a;
    memset(&a, 0, sizeof(a));
    a.tb = tb; a.lp = plan.lp; a.nb = nb; a.n_tiles = plan.n_tiles; a.cap_blk = cap_blk; a.cap_rows = cap_rows;
    a.lds_rows = (int)(plan.smem / ((NV + 1) * sizeof(double)));
    a.culled = culled ? 1 : 0; a.sl = sl; a.tq_tiles = tq_tiles; a.tq_count = tq_count; a.tq_cap = tq_cap;
    a.lo = h->d_bounds; a.hi = h->d_bounds + (size_t)B * m; a.g = h->d_g; a.jac = h->d_jac;
    a.ctl = reinterpret_cast<SolveCtl*>(w.ctl); a.blk_word = reinterpret_cast<BlockWord*>(w.blk_word);
    a.blk_rows = reinterpret_cast<SolveRow*>(w.blk_rows); a.qp_rows = reinterpret_cast<SolveRow*>(w.qp_rows); a.flags = w.flags;
    a.q_des = reinterpret_cast<const double*>(reinterpret_cast<unsigned char*>(w.ctl) + off_qdes); a.out = hres;   // the kernel writes the results straight into page-locked host memory
    for (int i = 0; i < n; i++) if (h->robot.continuous[i]) a.continuous_mask |= 1 << i;
    a.max_iter = opt.max_iterations; a.max_ls = opt.max_line_search; a.tol = opt.tolerance;
    a.n_checked_collision = armour_checked_collision_rows(h);
    a.t_plan = h->params.t_plan; a.cost_scale = h->params.cost_scale;
    a.torque_slack = h->params.torque_violation_threshold; a.collision_slack = h->params.collision_violation_threshold;
    a.budget_ticks = -1;
    if (opt.max_wall_time_s > 0) {   // (sub-batches run one after the other: each gets an equal share of what is left)
        const double left_ms = (opt.max_wall_time_s - std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count()) * 1e3 / n_launch;
        a.budget_ticks = left_ms > 0 ? (long long)(left_ms * plan.ticks_per_ms) : 0;
    }
    // no wait of the persistent kernel outlasts the budget by more than a second (without a budget: from the iteration limit and the tiles per block, below); the host's own poll gives up a little later.  ARMOUR_OPT_SOLVE_HARD_CAP_S: tuning / tests.
    const double hard_cap_s = h->tune_f(ARMOUR_OPT_SOLVE_HARD_CAP_S);
    // without a budget: 50 x what max_iterations iterations of (1 + max_line_search) evaluations cost at ~4 us per row tile of a block, at least 5 s
    const double no_budget_s = std::max(5.0, 1.0 + 50.0 * (double)std::max(1, opt.max_iterations) * (1.0 + std::max(0, opt.max_line_search)) * tiles_per_block * 4e-6);
    const double hard_s = hard_cap_s > 0 ? hard_cap_s : (a.budget_ticks >= 0 ? a.budget_ticks / plan.ticks_per_ms * 1e-3 + 1.0 : no_budget_s);
    a.hard_ticks = std::max<long long>(1, (long long)(hard_s * 1e3 * plan.ticks_per_ms));
    const bool timing = armour_trace_solve();
    long long* hstamps = timing ? reinterpret_cast<long long*>(armour_handle_pinned(h, 0, (size_t)B * 64 * sizeof(long long) + (size_t)B * n * sizeof(double))) : nullptr;
    if (hstamps) memset(hstamps, 0, (size_t)B * 64 * sizeof(long long));
    a.stamps = hstamps;
    const auto t_launch = std::chrono::steady_clock::now();
    for (int li = 0; li < n_launch; li++) { a.b0 = li * Bs; memcpy(hblock + off_args + (size_t)li * args_stride, &a, sizeof(SolveArgs)); }
    HIPCHK(hipMemcpyAsync(w.ctl, hblock, block_bytes, hipMemcpyHostToDevice, h->stream));   // (asynchronous from page-locked memory, ordered before the launches)
    for (int li = 0; li < n_launch; li++)
        if ((rc = armour_solve_device_launch(reinterpret_cast<const SolveArgs*>(reinterpret_cast<unsigned char*>(w.ctl) + off_args + (size_t)li * args_stride), nb, plan,
                                             std::min(Bs, B - li * Bs), h->stream)) != ARMOUR_OK) return rc;
    for (unsigned long long polls = 0;; polls++) {  // poll the stream: the whole solve is tens of microseconds, a sleeping wait would dominate it
        const hipError_t q = hipStreamQuery(h->stream);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) { armour_set_error("armour_solve (device form): %s", hipGetErrorString(q)); return ARMOUR_EDEVICE; }
        // the kernel bounds its own waits by hard_s; a stream that is still busy well after that is a device the caller must give up on
        // (recover in a fresh process; never by re-exec of this one)
        if ((polls & 0xfffull) == 0xfffull && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_launch).count() > n_launch * (2.0 * hard_s + 5.0)) {
            armour_set_error("armour_solve (device form): the persistent kernel did not finish within %.1f s", n_launch * (2.0 * hard_s + 5.0));
            return ARMOUR_EDEVICE;
        }
    }
    w.words_clean = 1;
    for (int b = 0; b < B; b++)
        if (hres[b].status < 0) { w.words_clean = 0; return 0; }   // candidate buffers too small for some problem, or a group lost a block: the host form redoes the solve
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    if (timing) {
        if (culled) fprintf(stderr, "[armour_solve, device form] culled: at most %d listed tiles per problem (problem 0: %d of %d collision rows, %d of %d torque rows), lists %.3f + %.3f ms\n",
                            n_tiles_culled, h->h_rel2_count[0], h->Q, h->h_rel2_tq_count[0], h->row0, h->rel_ms, h->rel2_ms);
        fprintf(stderr, "[armour_solve, device form] B=%d in %d launch(es) of <= %d problems: %d blocks per problem (%d tiles), %.3f ms wall (%.3f ms before the launch), kernel %.3f ms (problem 0); phases of problem 0 in us (barrier passed / leader done):", B, n_launch, Bs, nb, plan.n_tiles, ms,
                std::chrono::duration<double, std::milli>(t_launch - t_begin).count(), hres[0].time_ms / plan.ticks_per_ms);
        int bs = 0;   // the problem whose group left last: what the launch waited for
        for (int b = 1; b < B; b++) if (hres[b].time_ms > hres[bs].time_ms) bs = b;
        for (int which = 0; which < 2; which++) {
            const int b = which == 0 ? 0 : bs;
            const long long* st = hstamps + (size_t)b * 64;
            if (which == 1) fprintf(stderr, "\n   slowest problem %d: kernel %.3f ms, %d iterations, %d evaluations, status %d, %lld QP solves with %lld steps in all; phases in us:", b, hres[b].time_ms / plan.ticks_per_ms,
                                    hres[b].iterations, hres[b].evaluations, hres[b].status, st[43], st[42]);
            for (int i = 0; i < 32 && (i < 2 || st[i]); i++) fprintf(stderr, " %.1f", st[i] / plan.ticks_per_ms * 1e3);
            fprintf(stderr, " | first leader step: bookkeeping %.1f, candidates gathered %.1f, QP done %.1f us (%lld QP steps, %lld candidate rows)", st[32] / plan.ticks_per_ms * 1e3,
                    st[33] / plan.ticks_per_ms * 1e3, st[34] / plan.ticks_per_ms * 1e3, st[40], st[41]);
            fprintf(stderr, " | QP steps by attempt (sigma 0, 0.5, 0.9, 0.99), all solves: %lld %lld %lld %lld", st[44], st[45], st[46], st[47]);
            fprintf(stderr, " | attempt 0, all steps, us: row search %.1f, entering row %.1f, M + rhs %.1f, factor + solves %.1f, z + step lengths %.1f, update %.1f",
                    st[50] / plan.ticks_per_ms * 1e3, st[51] / plan.ticks_per_ms * 1e3, st[52] / plan.ticks_per_ms * 1e3, st[53] / plan.ticks_per_ms * 1e3, st[54] / plan.ticks_per_ms * 1e3, st[55] / plan.ticks_per_ms * 1e3);
        }
        long long longest_ok = 0, long_bad = 0, steps_all = 0;
        for (int b = 0; b < B; b++) { const long long* st = hstamps + (size_t)b * 64; longest_ok = std::max(longest_ok, st[48]); long_bad += st[49]; steps_all += st[44] + st[45] + st[46] + st[47]; }
        fprintf(stderr, "\n   all problems: %lld QP steps; longest attempt that ended feasible %lld steps; attempts of more than 60 steps that ended infeasible: %lld\n", steps_all, longest_ok, long_bad);
    }
    for (int b = 0; b < B; b++) { results[b] = hres[b]; results[b].time_ms = ms; }
    return 1;
}

extern "C" int armour_solve(ArmourPlanner* h, const ArmourSolveOptions* opt_in, ArmourSolveResult* results) {
    if (!h || !results) { armour_set_error("null argument"); return ARMOUR_EINVAL; }
    if (!h->ready) { armour_set_error("no problem set: call armour_set_problems first"); return ARMOUR_ESTATE; }
    ArmourSolveOptions opt;
    if (opt_in) opt = *opt_in; else armour_solve_options_default(&opt);
    const auto t_begin = std::chrono::steady_clock::now();
    const int B = h->B, n = h->n, m = h->m;
    {
        // Until round 5: the device-resident form from two problems on, the host-driven form below for a lone problem (force_host_qp: > 0 host form, < 0 device
        // form whatever the batch; ARMOUR_OPT_SOLVE_DEVICE = 0: never the device form, 2: always).  Both produce the same iterates.  Rounds 3-4 drew
        // the line at five: the leader ran the four elastic QP attempts of an infeasible linearisation one after the other at 6 us a step where a
        // host core takes 0.3.  With the attempts side by side on the leader's four waves (round 5) the persistent kernel wins from B = 2 on
        // (O = 20: 0.19 against 0.28 ms; B = 4, O = 10: 0.23 against 0.39; B = 6: 0.25 against 0.54) and ties for one problem (0.146 / 0.157
        // against 0.131 / 0.155 ms at O = 10 / 20), where the host form stayed.  Round 6: with the box-clipped first try (solver_common.h) a QP of the
        // reference's own worlds takes no active-set step, a lone problem's persistent kernel is 43-49 us instead of 118, and the device form wins for
        // ONE problem as well -- 0.070 against 0.120 ms (medians of the reference's 107 worlds, one at a time); on random worlds, most of them
        // infeasible with hundreds of QP steps, it loses 7 % (0.178 against 0.167 ms): the automatic choice is the device form for every batch size.
        // The host form remains the fallback (and ARMOUR_OPT_SOLVE_DEVICE = 0 / force_host_qp > 0 hold a handle / a call to it).
        const int dev_env = h->tune(ARMOUR_OPT_SOLVE_DEVICE);
        constexpr int kDeviceFormMinBatch = 1;
        const bool want_device = opt.force_host_qp > 0.0 ? false : opt.force_host_qp < 0.0 ? true : dev_env == 0 ? false : dev_env >= 2 ? true : B >= kDeviceFormMinBatch;
        if (want_device) {
            HIPCHK(hipSetDevice(h->device));
            const int r = solve_on_device(h, opt, results, t_begin);
            if (r != 0) return r < 0 ? r : ARMOUR_OK;
        }
    }
    // pinned host mirrors of k, g, jac for all problems
    // (owned by the handle and kept across solves; armour_eval_g_jac recognises them and lets the kernel write into them)
    HIPCHK(hipSetDevice(h->device));
    double* hk = armour_handle_pinned(h, 0, (size_t)B * n * sizeof(double));
    double *hg = nullptr, *hj = nullptr;  // whole-g / whole-jac mirrors: only the row-buffer overflow fallback allocates them
    if (!hk) return ARMOUR_EDEVICE;

    std::vector<double> xl(n), xu(n);
    int rc = ARMOUR_OK;
    // g and jac stay on the device; a scan kernel hands back the L1 violation and the compacted candidate rows
    const size_t bm = (size_t)B * m;
    if ((rc = armour_upload_bounds(h)) != ARMOUR_OK) return rc;   // (d_bounds: once per problem set, on the device)
    if (!h->bounds_on_host) {  // this form's own host copy, once per problem set
        h->h_gl.resize(bm); h->h_gu.resize(bm);
        if ((rc = armour_get_bounds(h, xl.data(), xu.data(), h->h_gl.data(), h->h_gu.data())) != ARMOUR_OK) return rc;
        h->bounds_on_host = true;
    } else if ((rc = armour_get_bounds(h, xl.data(), xu.data(), nullptr, nullptr)) != ARMOUR_OK) return rc;
    const std::vector<double>&gl = h->h_gl, &gu = h->h_gu;
    const int nseg = (m + kScanRowsPerBlock - 1) / kScanRowsPerBlock;
    // rows a segment may hand back (its slice of the page-locked buffer): everything if the batch is small
    int cap_rows = (int)std::max<size_t>(64, std::min<size_t>(2 * kScanRowsPerBlock, ((size_t)128 << 20) / ((size_t)B * nseg * sizeof(SolveRow))));
    if (h->tune(ARMOUR_OPT_SOLVE_ROW_CAP) > 0) cap_rows = std::max(1, h->tune(ARMOUR_OPT_SOLVE_ROW_CAP));  // tests: force the overflow fallback
    long long* hviol_seg = reinterpret_cast<long long*>(armour_handle_pinned(h, 3, (size_t)B * nseg * sizeof(long long)));
    int* hcount = reinterpret_cast<int*>(armour_handle_pinned(h, 4, (size_t)B * nseg * sizeof(int)));
    SolveRow* hrows = reinterpret_cast<SolveRow*>(armour_handle_pinned(h, 5, (size_t)B * nseg * cap_rows * sizeof(SolveRow)));
    if (!hviol_seg || !hcount || !hrows) return ARMOUR_EDEVICE;
    std::vector<double> hviol(B);
    bool full_on_host = false;  // g / jac of the current linearisation were copied to hg / hj (row-buffer overflow fallback)
    // constant diagonal Hessian of the cost: f = scale * sum (q_des - q0 - ... - c*kr*x)^2  (RT/NLPclass.cu:207-236)
    // (ARMTD mode, CMP/NLPclass.cu:183-243: the plan point is q0 + qd0/2 + k_range k/8 with the problem's own k_range)
    const double tp = h->params.t_plan;
    std::vector<double> Hd_all((size_t)B * NV, 1e-12);
    for (int b = 0; b < B; b++)
        for (int j = 0; j < n; j++) {
            const double dk = h->mode == ARMOUR_MODE_ARMTD ? 0.125 * h->h_krange[(size_t)b * n + j] : (tp * tp * tp) * (6 * tp * tp - 15 * tp + 10) * h->params.k_range[j];
            double& Hd = Hd_all[(size_t)b * NV + j];
            Hd = 2.0 * h->params.cost_scale * dk * dk;
            if (Hd < 1e-12) Hd = 1e-12;
        }
    std::vector<ProblemState> st(B);
    std::vector<double> fb(B), gfb((size_t)B * n);
    for (int b = 0; b < B; b++) { for (int j = 0; j < n; j++) { st[b].x[j] = 0.0; hk[b * n + j] = 0.0; } st[b].mu = 1.0; }  // get_starting_point: x = 0

    // one fused evaluation of all problems at hk (read by the kernel from page-locked memory) into the handle's device
    // buffers, then the scan; the host waits once
    auto wait = [&]() -> int {
        for (;;) {
            const hipError_t q = hipStreamQuery(h->stream);
            if (q == hipSuccess) return ARMOUR_OK;
            if (q != hipErrorNotReady) { armour_set_error("hipStreamQuery failed: %s", hipGetErrorString(q)); return ARMOUR_EDEVICE; }
        }
    };
    double t_eval = 0, t_qp = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto eval = [&](bool want_jac, bool verdict = false) -> int {
        const auto e0 = now();
        struct Acc { double& t; decltype(e0) s; ~Acc() { t += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - s).count(); } } acc{t_eval, e0};
        int r = armour_eval_g_jac_device(h, hk, h->d_g, want_jac ? h->d_jac : nullptr, h->stream);
        if (r != ARMOUR_OK) return r;
        const int nTq = h->row0, nCol = armour_checked_collision_rows(h), nSkip = h->Q - nCol;
        const double tsl = h->params.torque_violation_threshold, csl = h->params.collision_violation_threshold;
#define SCAN(MODE, JAC, CNT, ROWSP) hipLaunchKernelGGL(armour_solve_scan_kernel<MODE>, dim3(nseg, B), dim3(256), 0, h->stream, m, n, nTq, nCol, nSkip, tsl, csl, h->d_g, JAC, h->d_bounds, h->d_bounds + bm, cap_rows, hviol_seg, CNT, ROWSP)
        if (want_jac) SCAN(1, h->d_jac, hcount, hrows);
        else if (verdict) SCAN(2, (const double*)nullptr, hcount, (SolveRow*)nullptr);
        else SCAN(0, (const double*)nullptr, (int*)nullptr, (SolveRow*)nullptr);
#undef SCAN
        HIPCHK(hipGetLastError());
        if ((r = wait()) != ARMOUR_OK) return r;
        for (int b = 0; b < B; b++) {
            long long v = 0;
            for (int sg = 0; sg < nseg; sg++) v += hviol_seg[(size_t)b * nseg + sg];
            hviol[b] = slv::viol_from_fixed(v);
        }
        full_on_host = false;
        if (want_jac) {
            bool overflow = false;
            for (size_t i = 0; i < (size_t)B * nseg; i++) overflow = overflow || hcount[i] > cap_rows;
            if (overflow) {  // more candidate rows than the buffer holds: bring the whole linearisation over, as before
                hg = armour_handle_pinned(h, 1, bm * sizeof(double));
                hj = armour_handle_pinned(h, 2, bm * n * sizeof(double));
                if (!hg || !hj) return ARMOUR_EDEVICE;
                HIPCHK(hipMemcpyAsync(hg, h->d_g, bm * sizeof(double), hipMemcpyDeviceToHost, h->stream));
                HIPCHK(hipMemcpyAsync(hj, h->d_jac, bm * n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
                if ((r = wait()) != ARMOUR_OK) return r;
                full_on_host = true;
            }
        }
        return ARMOUR_OK;
    };
    auto time_left = [&]() {
        if (opt.max_wall_time_s <= 0) return true;
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() < opt.max_wall_time_s;
    };

    if ((rc = eval(true)) != ARMOUR_OK) return rc;
    if ((rc = armour_eval_f(h, hk, fb.data())) != ARMOUR_OK) return rc;
    if ((rc = armour_eval_grad_f(h, hk, gfb.data())) != ARMOUR_OK) return rc;
    for (int b = 0; b < B; b++) {
        st[b].f = fb[b];
        for (int j = 0; j < n; j++) st[b].gradf[j] = gfb[(size_t)b * n + j];
        st[b].viol = hviol[b];
        st[b].evals = 1;
    }
    std::vector<double> dstep((size_t)B * n), alpha(B), dphi(B), phi0(B);
    std::vector<char> searching(B);
    // the QPs are small now (tens to hundreds of rows each): a host thread per 32 problems
    const unsigned nthreads = std::max(1u, std::min(std::thread::hardware_concurrency(), (unsigned)B / 32u));

    for (int it = 0; it < opt.max_iterations; it++) {
        bool any = false;
        for (int b = 0; b < B; b++) any = any || !st[b].done;
        if (!any) break;
        if (!time_left()) { for (int b = 0; b < B; b++) if (!st[b].done) { st[b].done = true; st[b].status = 5; } break; }
        // ---- QP subproblems (host threads over problems) ----
        auto work = [&](int b0, int b1) {
            std::vector<QpRow> rows;
            for (int b = b0; b < b1; b++) {
                ProblemState& s = st[b];
                if (s.done) continue;
                const double* g = hg + (size_t)b * m;
                const double* J = hj + (size_t)b * m * n;
                const double* lo = gl.data() + (size_t)b * m;
                const double* hi = gu.data() + (size_t)b * m;
                QpResult qp;
                double sigma = 0.0;  // fraction of the current violation a row may keep (elastic retry)
                for (int attempt = 0; attempt < 4; attempt++) {
                    rows.clear();
                    if (!full_on_host) {
                        for (int sg = 0; sg < nseg; sg++) {
                            const SolveRow* cr = hrows + ((size_t)b * nseg + sg) * cap_rows;
                            for (int r0 = 0; r0 < hcount[(size_t)b * nseg + sg]; r0++) {
                                QpRow r; for (int j = 0; j < n; j++) r.a[j] = cr[r0].a[j];
                                const double v = cr[r0].v;
                                r.b = v - (v > 0 ? sigma * v : 0.0); rows.push_back(r);
                            }
                        }
                    }
                    for (int i = 0; full_on_host && i < m; i++) {
                        const double* Ji = J + (size_t)i * n;
                        double l1 = 0;
                        for (int j = 0; j < n; j++) l1 += std::fabs(Ji[j]);
                        // a row that cannot become active for any |d|_inf <= 2 is left out
                        if (hi[i] < 1e18 && g[i] + 2.0 * l1 > hi[i]) {
                            QpRow r; for (int j = 0; j < n; j++) r.a[j] = -Ji[j];
                            const double v = g[i] - hi[i];
                            r.b = v - (v > 0 ? sigma * v : 0.0); rows.push_back(r);
                        }
                        if (lo[i] > -1e18 && g[i] - 2.0 * l1 < lo[i]) {
                            QpRow r; for (int j = 0; j < n; j++) r.a[j] = Ji[j];
                            const double v = lo[i] - g[i];
                            r.b = v - (v > 0 ? sigma * v : 0.0); rows.push_back(r);
                        }
                    }
                    for (int j = 0; j < n; j++) {
                        QpRow r; memset(&r, 0, sizeof(r)); r.a[j] = 1.0; r.b = xl[j] - s.x[j]; rows.push_back(r);
                        QpRow r2; memset(&r2, 0, sizeof(r2)); r2.a[j] = -1.0; r2.b = -(xu[j] - s.x[j]); rows.push_back(r2);
                    }
                    qp = solve_qp(n, &Hd_all[(size_t)b * NV], s.gradf, rows, slv::kQpMaxSteps, 2 * n);
                    if (qp.feasible) break;
                    sigma = attempt == 0 ? 0.5 : attempt == 1 ? 0.9 : 0.99;
                }
                if (armour_trace_solve() && B == 1) {
                    fprintf(stderr, "[armour_solve, host form] iteration %d: QP %s at sigma %.2f after %d steps, %zu rows; x =", it, qp.feasible ? "feasible" : "INFEASIBLE", sigma, qp.iterations, rows.size());
                    for (int j = 0; j < n; j++) fprintf(stderr, " %.6g", s.x[j]);
                    fprintf(stderr, "; d =");
                    for (int j = 0; j < n; j++) fprintf(stderr, " %.6g", qp.x[j]);
                    fprintf(stderr, "; viol %.6g\n", s.viol);
                }
                if (!qp.feasible) { s.done = true; s.status = 3; continue; }
                double dn = 0, gd = 0;
                for (int j = 0; j < n; j++) { dstep[(size_t)b * n + j] = qp.x[j]; dn = std::fmax(dn, std::fabs(qp.x[j])); gd += s.gradf[j] * qp.x[j]; }
                if (dn <= opt.tolerance * 1e-2 || (dn <= opt.tolerance && s.viol <= opt.tolerance)) { s.done = true; s.status = 1; continue; }
                s.mu = std::fmax(s.mu, 1.5 * qp.max_mult + 1e-3);
                phi0[b] = s.f + s.mu * s.viol;
                dphi[b] = gd - s.mu * (1.0 - sigma) * s.viol;
                if (dphi[b] > -1e-14) dphi[b] = -1e-14;
                alpha[b] = 1.0;
                searching[b] = 1;
            }
        };
        for (int b = 0; b < B; b++) searching[b] = 0;
        bool linearised_at_trial = false;
        const auto q0 = now();
        if (nthreads <= 1) work(0, B);
        else {
            std::vector<std::thread> th;
            const int per = (B + (int)nthreads - 1) / (int)nthreads;
            for (unsigned tI = 0; tI < nthreads; tI++) { const int b0 = (int)tI * per, b1 = std::min(B, b0 + per); if (b0 < b1) th.emplace_back(work, b0, b1); }
            for (auto& t : th) t.join();
        }
        t_qp += std::chrono::duration<double, std::milli>(now() - q0).count();
        // ---- L1-merit backtracking line search, all problems in lock step (one eval_g launch per trial) ----
        for (int ls = 0; ls <= opt.max_line_search; ls++) {
            bool need = false;
            for (int b = 0; b < B; b++) {
                for (int j = 0; j < n; j++) hk[b * n + j] = st[b].x[j] + (searching[b] ? alpha[b] * dstep[(size_t)b * n + j] : 0.0);
                need = need || searching[b];
            }
            if (!need) break;
            // The full step is accepted nine times in ten, and the new linearisation is then wanted at exactly this point: the first trial is
            // evaluated WITH the Jacobian and the candidate rows (the same fused launch, 2 us more), and if every problem that was searching
            // takes it the round trip of the re-linearisation below is saved (an evaluation request less in flight, not in `evaluations`: that
            // counts what the algorithm asked for, in both forms of the solver).
            const bool speculative = ls == 0;
            if ((rc = eval(speculative)) != ARMOUR_OK) return rc;
            if (speculative) linearised_at_trial = true;
            if ((rc = armour_eval_f(h, hk, fb.data())) != ARMOUR_OK) return rc;
            for (int b = 0; b < B; b++) {
                if (!searching[b]) continue;
                st[b].evals++;
                const double v = hviol[b];
                const double phi = fb[b] + st[b].mu * v;
                if (phi <= phi0[b] + 1e-4 * alpha[b] * dphi[b] || ls == opt.max_line_search) {
                    if (ls == opt.max_line_search && phi > phi0[b]) { st[b].done = true; st[b].status = 4; searching[b] = 0; continue; }
                    for (int j = 0; j < n; j++) st[b].x[j] = hk[b * n + j];
                    st[b].f = fb[b]; st[b].viol = v; searching[b] = 0; st[b].iters++;
                } else {
                    alpha[b] *= 0.5;
                    linearised_at_trial = false;   // (this problem moves to another point: the linearisation taken at the full step is not its)
                }
            }
        }
        // ---- new linearisation at the accepted points (already there if every searching problem took its full step) ----
        for (int b = 0; b < B; b++) for (int j = 0; j < n; j++) hk[b * n + j] = st[b].x[j];
        if (!linearised_at_trial && (rc = eval(true)) != ARMOUR_OK) return rc;
        if ((rc = armour_eval_grad_f(h, hk, gfb.data())) != ARMOUR_OK) return rc;
        for (int b = 0; b < B; b++) {
            if (st[b].done) continue;
            st[b].evals++;
            for (int j = 0; j < n; j++) st[b].gradf[j] = gfb[(size_t)b * n + j];
            st[b].viol = hviol[b];
        }
    }
    // ---- finalize_solution: feasibility re-check with the reference's slack thresholds ----
    for (int b = 0; b < B; b++) for (int j = 0; j < n; j++) hk[b * n + j] = st[b].x[j];
    if ((rc = eval(false, true)) != ARMOUR_OK) return rc;  // with the per-row slack test of finalize_solution done on the device
    if ((rc = armour_eval_f(h, hk, fb.data())) != ARMOUR_OK) return rc;
    std::vector<int32_t> feas(B);
    for (int b = 0; b < B; b++) {
        int bad = 0;
        for (int sg = 0; sg < nseg; sg++) bad += hcount[(size_t)b * nseg + sg];
        feas[b] = bad == 0 ? 1 : 0;
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    if (armour_trace_solve()) fprintf(stderr, "[armour_solve] B=%d total %.3f ms: evaluations+scan %.3f ms, QP %.3f ms, rows of problem 0, segment 0: %d (cap %d, %d segments)\n", B, ms, t_eval, t_qp, hcount[0], cap_rows, nseg);
    for (int b = 0; b < B; b++) {
        ArmourSolveResult& r = results[b];
        memset(&r, 0, sizeof(r));
        for (int j = 0; j < n; j++) r.k_opt[j] = st[b].x[j];
        r.cost = fb[b];
        r.max_violation = hviol[b];
        r.feasible = feas[b];
        r.iterations = st[b].iters;
        r.evaluations = st[b].evals + 1;
        r.status = st[b].done ? st[b].status : 2;
        r.time_ms = ms;
    }
    return ARMOUR_OK;
}
