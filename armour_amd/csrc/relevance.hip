// Row relevance: which constraint rows can be violated for SOME trajectory parameter k in [-1, 1]^n -- the pruned constraint list of the
// reference's MATLAB path, on the device.
//
// KSI/uarmtd_planner.m keeps an obstacle constraint only if the forward occupancy can reach the buffered obstacle at all (:577-583: the centre of
// the link's zonotope inside the obstacle buffered by ALL of the link's generators) and an input / joint-limit constraint only if its interval
// enclosure is not entirely feasible (`~(interval(...).sup < 0)`, :628-690).  The C++ path has no such list: armtd_NLP evaluates every row of
// every table at every iterate (RT/NLPclass.cu:272-396).  On random worlds about 2 % of the collision rows can ever be violated (20 .. 50
// obstacles: tools/dev) -- the other 98 % are 98 % of what a reduced-output evaluation streams from HBM.
//
// The test, per collision row q = (link l, time step t, obstacle o): the row is g(k) = -max_p max(A_p x(k) - (d_p + delta_p), -A_p x(k) -
// (-d_p + delta_p)) over the half-space pairs p of the buffered obstacle (RT/CollisionChecking.cu:230-299), x(k) = the sliced link centre
// c + sum_i coef_i k^alpha_i (RT/PZsparse.cu:404-435), |k^alpha_i| <= 1.  If ONE live plane p separates the whole family,
//     |A_p c - d_p| - sum_i |A_p coef_i| >= delta_p + margin,
// then one of the two values of that plane is positive for every k, the maximum is positive, g(k) < 0: the row can never be violated and is
// IRRELEVANT.  Sound by construction (a sufficient condition with a 1e-9 margin against the rounding of the evaluation's own arithmetic); not
// tight (the MATLAB path tests against the exact zonotope).  Torque rows: the sliced torque centre lies in cen +- sum |coef|, relevant iff that
// interval reaches a bound; the 4n limit rows are always relevant.
//
// What uses it: armour_get_row_relevance (the mask: a MATLAB caller hands fmincon the kept rows only), and -- ARMOUR_OPT_CULL_ROWS = 1 --
// armour_eval_violations*, which then evaluates the torque / limit blocks and the LISTED collision rows only (one lane per listed row, the
// evaluation's own arithmetic in the evaluation's own order: the same g bit for bit) and reduces over the same rows: L1 violation, counts and
// verdict are those of the full evaluation bit for bit (an unlisted row adds exactly 0).  armour_eval_g_jac* stays full (IPOPT parity).
#include <vector>

#include "p2_sparse.h"

using namespace p2;

namespace {

constexpr double kRelMargin = 1e-9;

struct RelTables {
    P2Tables tb;
    const double* lo; const double* hi;   // [B][m] bounds (uploaded: armour_upload_bounds)
    unsigned char* rel;                   // [B][m]
    unsigned char* rel2;                  // [B][m] the solver's mask: rows that can pass armour_solve's candidate filter for SOME k (a superset of rel), or null
    int* tq_tiles; int* tq_count; int tq_cap;   // the torque ROWS of the mask, ascending: [B][tq_cap] | [B]
    int* rows; int* count;                // [B][Q] relevant collision rows in ascending order | [B]
    int* rows_res; int* count_res;        // the same rows by (row0 + q) mod 256, ascending within a class: [B][256][ceil(Q / 256)] | [B][256]
    // the half-space entries of the listed rows, PACKED: problem b's block starts at packed + pack_off[2 b] doubles and holds, for its j-th live
    // plane (ascending) and component c of {Ax, Ay, Az, d, delta}, the values of its listed rows i at [(j * 5 + c) * pack_off[2 b + 1] + i] -- a
    // gather from the full table fetches a 64-byte sector for every 8 .. 16 bytes it uses (measured: 143 us for 160 k listed rows, as long as a
    // third of the full evaluation); packed, a lane's loads are consecutive with its neighbours'
    double* packed; const long long* pack_off;
};

// plane p of row q as the fused evaluation sees it: normal from the table (link x link planes from the compact records when the table shares
// them), delta from the table, d stored or recomputed from the obstacle centre -- the same choices as armour_p2_plan makes (dfc)
__device__ inline PlaneVals load_plane(const P2Tables& tb, const double* pl0, const double* pll, int Q, int q, int p, bool dfc, double oc0, double oc1, double oc2) {
    PlaneVals v;
    if (tb.ll_shared && p >= ARMOUR_FIRST_LL_PLANE) {
        const double* al = pll + (size_t)(p - ARMOUR_FIRST_LL_PLANE) * 3;
        v.a0 = al[0]; v.a1 = al[1]; v.a2 = al[2];
    } else {
        v.a0 = pl0[armour_plane_index(Q, q, p, 0)]; v.a1 = pl0[armour_plane_index(Q, q, p, 1)]; v.a2 = pl0[armour_plane_index(Q, q, p, 2)];
    }
    v.dl = pl0[armour_plane_index(Q, q, p, 4)];
    v.dd = dfc ? v.a0 * oc0 + v.a1 * oc1 + v.a2 * oc2 : pl0[armour_plane_index(Q, q, p, 3)];
    return v;
}

// ---- collision rows: one lane per row
// (1 + 1e-6)^21 < 1 + kBoxSlack: a monomial of total degree <= 3 n = 21 at a point the persistent solver still evaluates on the lists (its guard
// admits |x_j| <= 1 + 1e-6, solver_device.hip) against its bound on the box -- the factor the solver's mask inflates its sums by (the
// per-factor degree 3 under-counted it)
constexpr double kBoxSlack = 3e-5;

__global__ __launch_bounds__(64) void armour_rel_collision_kernel(RelTables a, int dfc, int any_normals) {
    const P2Tables& tb = a.tb;
    const int b = blockIdx.y, q = blockIdx.x * 64 + threadIdx.x;
    const int Q = tb.Q, O = tb.O, JT = tb.J * tb.T;
    if (q >= Q) return;
    const int lt = q / O, o = q - lt * O;
    const size_t idx = (size_t)b * JT + lt;
    const int cnt = min(tb.link_count[idx], tb.capL);
    const double c0 = tb.link_center[idx * 3 + 0], c1 = tb.link_center[idx * 3 + 1], c2 = tb.link_center[idx * 3 + 2];
    const double* co = tb.link_coeff + idx * tb.capL * 3;
    const double* pl0 = tb.planes + (size_t)b * armour_planes_per_problem(Q);
    const double* pll = tb.planes_ll + (size_t)b * armour_planes_ll_per_problem(JT) + (size_t)lt * ARMOUR_LL_RECORD;
    double oc0 = 0.0, oc1 = 0.0, oc2 = 0.0;
    if (dfc) { const double* oc = tb.obs_center + (size_t)b * 3 * O + o; oc0 = oc[0]; oc1 = oc[O]; oc2 = oc[2 * (size_t)O]; }
    const unsigned long long live0 = ~tb.plane_skip[b] & ((1ull << ARMOUR_NPLANES) - 1ull);
    // first with the interval hull of the family (sum_i |A coef_i| <= |A| . r, r[e] = sum_i |coef_i[e]|: three products a plane) and out at the
    // first plane that separates -- most rows are far from their obstacle; the tight sums only for the rows the hull leaves undecided.
    // The SOLVER's mask asks for more: armour_solve takes row i into its QP when g_i + 2 |J_i|_1 > u_i (solver_common.h row_upper_candidate).
    // With a separating plane of margin L (one of its two values >= L for every k) g_i <= -L, and |J_i|_1 = |A_win . dx|_1 <= |rd|_2 for the
    // unit normal of whichever plane wins, rd[e] = sum_i deg_i |coef_i[e]| (|d k^alpha / d k_j| <= alpha_j on the box): the row can never be a
    // candidate when L >= 2 |rd|_2 - u_i + margin.
    double r0 = 0.0, r1 = 0.0, r2 = 0.0, rd0 = 0.0, rd1 = 0.0, rd2 = 0.0;
    {
        const uint32_t* kk = tb.link_keys + idx * tb.capL;
        for (int mo = 0; mo < cnt; mo++) {
            const double f0 = fabs(co[mo * 3]), f1 = fabs(co[mo * 3 + 1]), f2 = fabs(co[mo * 3 + 2]);
            uint32_t key = kk[mo];
            int deg = 0;
            for (int j = 0; j < ARMOUR_MAX_FACTORS; j++) { deg += (int)(key & 3u); key >>= 2; }
            r0 += f0; r1 += f1; r2 += f2;
            rd0 += deg * f0; rd1 += deg * f1; rd2 += deg * f2;
        }
    }
    const size_t row = (size_t)b * tb.m + tb.row0 + q;
    const double ui = a.hi[row];
    const bool lower_bounded = a.lo[row] > -1e18;   // (collision rows have no lower bound; if one had, it stays a solver row)
    // (the solver's points stay inside the box to 1e-7 -- every accepted QP step is verified against the variables' bounds, solver.hip -- and the
    //  persistent kernel hands a point beyond 1 + 1e-6 back to the host form: the sums of the solver's mask carry a factor 1 + kBoxSlack where the
    //  relevance mask's carry 1 + 1e-12.  |J_i|_1 <= |A_win|_2 |rd|_2: the build's normals are unit vectors; tables loaded from the host
    //  (armour_debug_load_tables, `any_normals`) carry whatever the caller gave, so the bound takes each plane's own norm there.)
    double amax = 1.0;   // the largest norm a winning normal can have
    if (any_normals && a.rel2)
        for (unsigned long long live = live0; live; live &= live - 1ull) {
            const PlaneVals v = load_plane(tb, pl0, pll, Q, q, __builtin_ctzll(live), dfc != 0, oc0, oc1, oc2);
            amax = fmax(amax, sqrt(v.a0 * v.a0 + v.a1 * v.a1 + v.a2 * v.a2) * (1.0 + 1e-12));
        }
    const double need2 = kRelMargin + 2.0 * sqrt(rd0 * rd0 + rd1 * rd1 + rd2 * rd2) * amax * (1.0 + kBoxSlack) - ui;
    bool separated = false, sep2 = a.rel2 == nullptr;
    for (unsigned long long live = live0; live && !(separated && sep2); live &= live - 1ull) {
        const PlaneVals v = load_plane(tb, pl0, pll, Q, q, __builtin_ctzll(live), dfc != 0, oc0, oc1, oc2);
        const bool nz = (v.a0 != 0.0) | (v.a1 != 0.0) | (v.a2 != 0.0);
        const double s = fabs(v.a0 * c0 + v.a1 * c1 + v.a2 * c2 - v.dd);
        const double hh = fabs(v.a0) * r0 + fabs(v.a1) * r1 + fabs(v.a2) * r2;
        const double L = s - hh * (1.0 + 1e-12) - v.dl;
        separated |= nz && L >= kRelMargin;
        sep2 |= nz && L - hh * kBoxSlack >= need2;
    }
    for (unsigned long long live = live0; live && !(separated && sep2); live &= live - 1ull) {
        const PlaneVals v = load_plane(tb, pl0, pll, Q, q, __builtin_ctzll(live), dfc != 0, oc0, oc1, oc2);
        const bool nz = (v.a0 != 0.0) | (v.a1 != 0.0) | (v.a2 != 0.0);
        const double s = fabs(v.a0 * c0 + v.a1 * c1 + v.a2 * c2 - v.dd);
        double h = 0.0;
        for (int mo = 0; mo < cnt; mo++) h += fabs(v.a0 * co[mo * 3] + v.a1 * co[mo * 3 + 1] + v.a2 * co[mo * 3 + 2]);
        const double L = s - h * (1.0 + 1e-12) - v.dl;
        separated |= nz && L >= kRelMargin;
        sep2 |= nz && L - h * kBoxSlack >= need2;
    }
    if (a.rel2) a.rel2[row] = (sep2 && separated && !lower_bounded) ? 0 : 1;
    a.rel[(size_t)b * tb.m + tb.row0 + q] = separated ? 0 : 1;
}

// ---- torque rows (one thread per row) and the limit rows (always relevant)
__global__ __launch_bounds__(256) void armour_rel_other_kernel(RelTables a) {
    const P2Tables& tb = a.tb;
    const int b = blockIdx.y, r = blockIdx.x * 256 + threadIdx.x;
    const int nT = tb.row0, lim0 = tb.row0 + tb.Q;
    if (r < nT) {   // row t * n + j
        const int t = r / tb.n, j = r - t * tb.n;
        const size_t idx = ((size_t)b * tb.n + j) * tb.T + t;
        const int cnt = min(tb.tq_count[idx], tb.capT);
        double rad = 0.0, radd = 0.0;   // sum |coef|, sum deg |coef| (>= |J_row|_1 on the box)
        for (int mo = 0; mo < cnt; mo++) {
            const double f = fabs(tb.tq_coeff[idx * tb.capT + mo]);
            uint32_t key = tb.tq_keys[idx * tb.capT + mo];
            int deg = 0;
            for (int j2 = 0; j2 < ARMOUR_MAX_FACTORS; j2++) { deg += (int)(key & 3u); key >>= 2; }
            rad += f; radd += deg * f;
        }
        const double cen = tb.tq_center[idx];
        const double lo = a.lo[(size_t)b * tb.m + r], hi = a.hi[(size_t)b * tb.m + r];
        const bool relevant = cen + rad >= hi - kRelMargin || cen - rad <= lo + kRelMargin;
        a.rel[(size_t)b * tb.m + r] = relevant ? 1 : 0;
        // the solver's filter: g + 2 |J|_1 > u or g - 2 |J|_1 < l for some k (solver_common.h)
        const double reach = (rad + 2.0 * radd) * (1.0 + kBoxSlack);
        if (a.rel2) a.rel2[(size_t)b * tb.m + r] = (relevant || cen + reach >= hi - kRelMargin || cen - reach <= lo + kRelMargin) ? 1 : 0;
    } else if (r - nT < tb.m - lim0) {
        a.rel[(size_t)b * tb.m + lim0 + (r - nT)] = 1;
        if (a.rel2) a.rel2[(size_t)b * tb.m + lim0 + (r - nT)] = 1;
    }
}

// ---- the relevant collision rows of every problem in ascending order (one block per problem)
__global__ __launch_bounds__(256) void armour_rel_list_kernel(RelTables a) {
    const P2Tables& tb = a.tb;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    __shared__ int wtot[4];
    const unsigned char* rel = a.rel + (size_t)b * tb.m + tb.row0;
    int* rows = a.rows + (size_t)b * tb.Q;
    // ... and by residue class of the ROW INDEX row0 + q modulo 256: the 256 rows of a pass are 256 consecutive integers, so every class gets at
    // most one of them per pass and thread tid appends to its class without a conflict, in ascending order (what the row test needs: it adds a
    // thread's rows in ascending order, as the full kernel does)
    const int per = (tb.Q + 255) / 256;
    int* rres = a.rows_res ? a.rows_res + (size_t)b * 256 * per : nullptr;
    __shared__ int cres[256];   // entries per class so far
    cres[tid] = 0;
    __syncthreads();
    int base = 0;
    for (int q0 = 0; q0 < tb.Q; q0 += 256) {
        const int q = q0 + tid;
        const bool f = q < tb.Q && rel[q] != 0;
        if (f && a.rows_res) { const int c = (tb.row0 + q) & 255; rres[(size_t)c * per + cres[c]] = q; cres[c]++; }
        const unsigned long long bl = __ballot(f);
        if (lane == 0) wtot[wv] = __popcll(bl);
        __syncthreads();
        int pos = base + __popcll(bl & ((1ull << lane) - 1ull));
        for (int w = 0; w < wv; w++) pos += wtot[w];
        if (f) rows[pos] = q;
        base += wtot[0] + wtot[1] + wtot[2] + wtot[3];
        __syncthreads();
    }
    if (tid == 0) a.count[b] = base;
    if (a.rows_res) a.count_res[(size_t)b * 256 + tid] = cres[tid];
    // the torque rows of the mask, ascending
    if (a.tq_tiles) {
        const unsigned char* relt = a.rel + (size_t)b * tb.m;
        int tbase = 0;
        __syncthreads();
        for (int r0 = 0; r0 < tb.row0; r0 += 256) {
            const int r = r0 + tid;
            const bool f = r < tb.row0 && relt[r] != 0;
            const unsigned long long bl = __ballot(f);
            if (lane == 0) wtot[wv] = __popcll(bl);
            __syncthreads();
            int pos = tbase + __popcll(bl & ((1ull << lane) - 1ull));
            for (int w = 0; w < wv; w++) pos += wtot[w];
            if (f) a.tq_tiles[(size_t)b * a.tq_cap + pos] = r;
            tbase += wtot[0] + wtot[1] + wtot[2] + wtot[3];
            __syncthreads();
        }
        if (tid == 0) a.tq_count[b] = tbase;
    }
}

// ---- pack the listed rows' plane entries (once per problem set)
__global__ __launch_bounds__(64) void armour_rel_pack_kernel(RelTables a, int dfc) {
    const P2Tables& tb = a.tb;
    const int b = blockIdx.y, i = blockIdx.x * 64 + threadIdx.x;
    if (i >= a.count[b]) return;
    const int Q = tb.Q, O = tb.O, JT = tb.J * tb.T;
    const int q = a.rows[(size_t)b * Q + i];
    const int lt = q / O, o = q - lt * O;
    const double* pl0 = tb.planes + (size_t)b * armour_planes_per_problem(Q);
    const double* pll = tb.planes_ll + (size_t)b * armour_planes_ll_per_problem(JT) + (size_t)lt * ARMOUR_LL_RECORD;
    double oc0 = 0.0, oc1 = 0.0, oc2 = 0.0;
    if (dfc) { const double* oc = tb.obs_center + (size_t)b * 3 * O + o; oc0 = oc[0]; oc1 = oc[O]; oc2 = oc[2 * (size_t)O]; }
    double* dst = a.packed + a.pack_off[2 * b] + i;
    const size_t stride = (size_t)a.pack_off[2 * b + 1];
    int j = 0;
    for (unsigned long long live = ~tb.plane_skip[b] & ((1ull << ARMOUR_NPLANES) - 1ull); live; live &= live - 1ull, j++) {
        const PlaneVals v = load_plane(tb, pl0, pll, Q, q, __builtin_ctzll(live), dfc != 0, oc0, oc1, oc2);
        dst[(size_t)(j * 5 + 0) * stride] = v.a0; dst[(size_t)(j * 5 + 1) * stride] = v.a1; dst[(size_t)(j * 5 + 2) * stride] = v.a2;
        dst[(size_t)(j * 5 + 3) * stride] = v.dd; dst[(size_t)(j * 5 + 4) * stride] = v.dl;
    }
}

// ---- g of the listed collision rows at k: one lane per listed row, the arithmetic of p2_tiles.h's collision_block in its order
// (slice: centre + monomials in table order, interval centre -- RT/PZsparse.cu:404-435; planes in ascending order -- the value of the maximum
// does not depend on who wins a tie)
__global__ __launch_bounds__(64) void armour_sparse_collision_g_kernel(RelTables a, const double* __restrict__ k_all, double* __restrict__ g_all) {
    const P2Tables& tb = a.tb;
    __shared__ KPow kp;
    const int b = blockIdx.y, i = blockIdx.x * 64 + threadIdx.x;
    const int cntb = a.count[b];
    if (blockIdx.x * 64 >= cntb) return;
    fill_kpow(kp, (int)threadIdx.x < tb.n ? k_all[(size_t)b * tb.n + threadIdx.x] : 0.0, tb.n);
    __syncthreads();
    if (i >= cntb) return;
    const SparseList sl{a.rows, a.count, a.packed, a.pack_off};
    sparse_collision_row<false>(tb, sl, b, i, kp, g_all + (size_t)b * tb.m, nullptr);
}

// ---- armour_violation_kernel (api.hip) over the torque rows, the LISTED collision rows and the limit rows.  Thread t takes the rows r with
// r mod 256 == t in ascending order, as the full kernel does, and the same tree combines the partial records: an unlisted row would have added
// exactly 0 to every field, so the record is the full kernel's bit for bit (worst / worst_row: whenever any row is violated).
struct SparseViolArgs {
    int m, row0, Q, n_checked;
    double torque_slack, collision_slack;
    const double* g; const double* lo; const double* hi;
    const int* rows_res; const int* count_res;   // (by residue class of the row index: armour_rel_list_kernel)
    const double* k; int n;                      // the point [B][n]: the lists hold for k inside the box only
    ArmourViolation* out;
};
__global__ __launch_bounds__(256) void armour_sparse_violation_kernel(SparseViolArgs a) {
    __shared__ double s_l1[256], s_w[256];
    __shared__ int s_row[256], s_nv[256], s_no[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const double* g = a.g + (size_t)b * a.m;
    const double* lo = a.lo + (size_t)b * a.m;
    const double* hi = a.hi + (size_t)b * a.m;
    double l1 = 0.0, worst = 0.0;
    int wrow = -1, nv = 0, no = 0;
    auto take = [&](int r) {
        const double v = g[r], l = lo[r], u = hi[r];
        const double viol = fmax(0.0, fmax(l - v, v - u));
        l1 += viol;
        if (viol > 0.0) nv++;
        if (viol > worst) { worst = viol; wrow = r; }
        bool outside;
        if (r < a.row0) outside = v < l - a.torque_slack || v > u + a.torque_slack;
        else if (r < a.row0 + a.Q) outside = (r - a.row0) < a.n_checked && v > a.collision_slack;
        else outside = v < l || v > u;
        if (outside) no++;
    };
    for (int r = tid; r < a.row0; r += 256) take(r);
    const int per = (a.Q + 255) / 256;
    const int* mine = a.rows_res + ((size_t)b * 256 + tid) * per;   // the listed rows with (row0 + q) mod 256 == tid, ascending
    const int cnt = a.count_res[(size_t)b * 256 + tid];
    for (int i = 0; i < cnt; i++) take(a.row0 + mine[i]);
    for (int r = a.row0 + a.Q + ((tid - (a.row0 + a.Q)) & 255); r < a.m; r += 256) take(r);
    s_l1[tid] = l1; s_w[tid] = worst; s_row[tid] = wrow; s_nv[tid] = nv; s_no[tid] = no;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            s_l1[tid] += s_l1[tid + s]; s_nv[tid] += s_nv[tid + s]; s_no[tid] += s_no[tid + s];
            const double ow = s_w[tid + s];
            const int orow = s_row[tid + s];
            if (ow > s_w[tid] || (ow == s_w[tid] && orow >= 0 && (s_row[tid] < 0 || orow < s_row[tid]))) { s_w[tid] = ow; s_row[tid] = orow; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        ArmourViolation o;
        o.l1_violation = s_l1[0]; o.worst = s_w[0]; o.worst_row = s_row[0]; o.n_violated = s_nv[0]; o.n_outside_slack = s_no[0];
        o.feasible = s_no[0] == 0 ? 1 : 0;
        // The mask says "never violated for a k of [-1, 1]^n".  A point outside the box may violate an unlisted row, so its record cannot claim to be
        // the full evaluation's: it is marked instead (feasible = -1, worst_row = -2; armour_hip.h) -- the host entry never gets here with such a point
        // (it takes every row then), a device caller re-evaluates the problem with ARMOUR_OPT_CULL_ROWS = 0.
        bool in_box = true;
        for (int j = 0; j < a.n; j++) in_box = in_box && fabs(a.k[(size_t)b * a.n + j]) <= 1.0;
        if (!in_box) { o.feasible = -1; o.worst_row = -2; }
        a.out[b] = o;
    }
}

template <class Tp>
int rel_alloc(Tp** p, size_t* cap, size_t need) {
    if (*cap >= need) return ARMOUR_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    HIPCHK(hipMalloc((void**)p, need * sizeof(Tp)));
    *cap = need;
    return ARMOUR_OK;
}

}  // namespace

void armour_relevance_free(ArmourPlanner* h) {
    if (h->d_rel) (void)hipFree(h->d_rel);
    if (h->d_rel_rows) (void)hipFree(h->d_rel_rows);
    if (h->d_rel_count) (void)hipFree(h->d_rel_count);
    if (h->d_rel_rows_res) (void)hipFree(h->d_rel_rows_res);
    if (h->d_rel_packed) (void)hipFree(h->d_rel_packed);
    if (h->d_rel_pack_off) (void)hipFree(h->d_rel_pack_off);
    if (h->d_rel2) (void)hipFree(h->d_rel2);
    if (h->d_rel2_rows) (void)hipFree(h->d_rel2_rows);
    if (h->d_rel2_count) (void)hipFree(h->d_rel2_count);
    if (h->d_rel2_packed) (void)hipFree(h->d_rel2_packed);
    if (h->d_rel2_pack_off) (void)hipFree(h->d_rel2_pack_off);
    if (h->d_rel2_tq_tiles) (void)hipFree(h->d_rel2_tq_tiles);
    h->d_rel2 = nullptr; h->d_rel2_rows = nullptr; h->d_rel2_count = nullptr; h->d_rel2_packed = nullptr; h->d_rel2_pack_off = nullptr; h->d_rel2_tq_tiles = nullptr;
    h->rel2_cap = h->rel2_rows_cap = h->rel2_count_cap = h->rel2_packed_cap = h->rel2_pack_off_cap = h->rel2_tq_tiles_cap = 0;
    h->rel2_fresh = false;
    h->d_rel_packed = nullptr; h->d_rel_pack_off = nullptr; h->rel_packed_cap = h->rel_pack_off_cap = 0;
    h->d_rel = nullptr; h->d_rel_rows = nullptr; h->d_rel_count = nullptr; h->d_rel_rows_res = nullptr;
    h->rel_cap = h->rel_rows_cap = h->rel_count_cap = h->rel_rows_res_cap = 0;
    h->rel_fresh = false;
}

// offsets of the packed plane entries of B row lists: [2 b] first double of problem b's block, [2 b + 1] its row stride; returns the total
static long long pack_offsets(const ArmourPlanner* h, const std::vector<int>& count, std::vector<long long>& off) {
    off.resize(2 * count.size());
    long long total = 0;
    for (size_t b = 0; b < count.size(); b++) {
        const long long stride = (count[b] + 15) & ~15;
        const int nlive = __builtin_popcountll(~h->h_plane_skip[b] & ((1ull << ARMOUR_NPLANES) - 1ull));
        off[2 * b] = total; off[2 * b + 1] = stride;
        total += (long long)5 * nlive * stride;
    }
    return total;
}

// mask + row lists of the current problem set (once per problem set: begin_problem_set clears rel_fresh / rel2_fresh).
// for_solver: also the lists of the solver's mask (rows that can pass armour_solve's candidate filter), their packed plane entries and the
// torque rows of the mask -- what the culled device form of armour_solve walks (solver_device.hip).
int armour_relevance_build(ArmourPlanner* h, bool for_solver) {
    if (h->rel_fresh && (!for_solver || h->rel2_fresh)) return ARMOUR_OK;
    int rc = armour_upload_bounds(h);
    if (rc != ARMOUR_OK) return rc;
    const size_t B = (size_t)h->B;
    RelTables a;
    a.tb = armour_make_tables(h);
    a.lo = h->d_bounds; a.hi = h->d_bounds + B * h->m;
    const int dfc = a.tb.obs_center != nullptr && a.tb.ll_shared;
    struct Events {   // (destroyed on every way out)
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Events() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
    } ev;
    HIPCHK(hipEventCreate(&ev.e0)); HIPCHK(hipEventCreate(&ev.e1));
    hipEvent_t e0 = ev.e0, e1 = ev.e1;
    if (!h->rel_fresh) {
        if ((rc = rel_alloc(&h->d_rel, &h->rel_cap, B * h->m)) != ARMOUR_OK) return rc;
        if ((rc = rel_alloc(&h->d_rel2, &h->rel2_cap, B * h->m)) != ARMOUR_OK) return rc;
        if ((rc = rel_alloc(&h->d_rel_rows, &h->rel_rows_cap, B * (size_t)std::max(h->Q, 1))) != ARMOUR_OK) return rc;
        if ((rc = rel_alloc(&h->d_rel_count, &h->rel_count_cap, B * 257)) != ARMOUR_OK) return rc;   // [B] counts | [B][256] per residue class
        if ((rc = rel_alloc(&h->d_rel_rows_res, &h->rel_rows_res_cap, B * 256 * (size_t)std::max((h->Q + 255) / 256, 1))) != ARMOUR_OK) return rc;
        a.rel = h->d_rel; a.rel2 = h->d_rel2; a.rows = h->d_rel_rows; a.count = h->d_rel_count; a.rows_res = h->d_rel_rows_res; a.count_res = h->d_rel_count + (size_t)h->B;
        a.tq_tiles = nullptr; a.tq_count = nullptr; a.tq_cap = 0;
        a.packed = nullptr; a.pack_off = nullptr;
        HIPCHK(hipEventRecord(e0, h->stream));
        if (h->Q > 0) hipLaunchKernelGGL(armour_rel_collision_kernel, dim3((h->Q + 63) / 64, h->B), dim3(64), 0, h->stream, a, dfc, h->tables_from_host ? 1 : 0);
        const int other = h->row0 + (h->m - h->row0 - h->Q);
        hipLaunchKernelGGL(armour_rel_other_kernel, dim3((other + 255) / 256, h->B), dim3(256), 0, h->stream, a);
        hipLaunchKernelGGL(armour_rel_list_kernel, dim3(h->B), dim3(256), 0, h->stream, a);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(e1, h->stream));
        h->h_rel_count.resize(B);
        HIPCHK(hipMemcpyAsync(h->h_rel_count.data(), h->d_rel_count, B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        h->rel_ms = ms;
        h->rel_max_count = 0;
        for (int c : h->h_rel_count) h->rel_max_count = std::max(h->rel_max_count, c);
        // the packed plane entries of the listed rows: offsets from the counts just read back
        std::vector<long long> off;
        const long long total = pack_offsets(h, h->h_rel_count, off);
        if ((rc = rel_alloc(&h->d_rel_packed, &h->rel_packed_cap, (size_t)std::max(total, 1ll))) != ARMOUR_OK) return rc;
        if ((rc = rel_alloc(&h->d_rel_pack_off, &h->rel_pack_off_cap, 2 * B)) != ARMOUR_OK) return rc;
        HIPCHK(hipMemcpyAsync(h->d_rel_pack_off, off.data(), 2 * B * sizeof(long long), hipMemcpyHostToDevice, h->stream));
        a.packed = h->d_rel_packed; a.pack_off = h->d_rel_pack_off;
        if (h->rel_max_count > 0) hipLaunchKernelGGL(armour_rel_pack_kernel, dim3((h->rel_max_count + 63) / 64, h->B), dim3(64), 0, h->stream, a, dfc);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(h->stream));   // (`off` is read by the copy)
        h->rel_fresh = true;
    }
    if (for_solver && !h->rel2_fresh) {
        const int tq_cap = std::max(1, h->row0);
        if ((rc = rel_alloc(&h->d_rel2_rows, &h->rel2_rows_cap, B * (size_t)std::max(h->Q, 1))) != ARMOUR_OK) return rc;
        if ((rc = rel_alloc(&h->d_rel2_count, &h->rel2_count_cap, 2 * B)) != ARMOUR_OK) return rc;   // [B] listed collision rows | [B] listed torque rows
        if ((rc = rel_alloc(&h->d_rel2_tq_tiles, &h->rel2_tq_tiles_cap, B * (size_t)tq_cap)) != ARMOUR_OK) return rc;
        a.rel = h->d_rel2; a.rel2 = nullptr; a.rows = h->d_rel2_rows; a.count = h->d_rel2_count; a.rows_res = nullptr; a.count_res = nullptr;
        a.tq_tiles = h->d_rel2_tq_tiles; a.tq_count = h->d_rel2_count + B; a.tq_cap = tq_cap;
        a.packed = nullptr; a.pack_off = nullptr;
        HIPCHK(hipEventRecord(e0, h->stream));
        hipLaunchKernelGGL(armour_rel_list_kernel, dim3(h->B), dim3(256), 0, h->stream, a);
        HIPCHK(hipGetLastError());
        std::vector<int> cnt(2 * B);
        HIPCHK(hipMemcpyAsync(cnt.data(), h->d_rel2_count, 2 * B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->h_rel2_count.assign(cnt.begin(), cnt.begin() + B);
        h->h_rel2_tq_count.assign(cnt.begin() + B, cnt.end());
        h->rel2_tq_cap = tq_cap;
        h->rel2_max_count = 0;
        for (int c : h->h_rel2_count) h->rel2_max_count = std::max(h->rel2_max_count, c);
        std::vector<long long> off;
        const long long total = pack_offsets(h, h->h_rel2_count, off);
        if ((rc = rel_alloc(&h->d_rel2_packed, &h->rel2_packed_cap, (size_t)std::max(total, 1ll))) != ARMOUR_OK) return rc;
        if ((rc = rel_alloc(&h->d_rel2_pack_off, &h->rel2_pack_off_cap, 2 * B)) != ARMOUR_OK) return rc;
        HIPCHK(hipMemcpyAsync(h->d_rel2_pack_off, off.data(), 2 * B * sizeof(long long), hipMemcpyHostToDevice, h->stream));
        a.packed = h->d_rel2_packed; a.pack_off = h->d_rel2_pack_off;
        if (h->rel2_max_count > 0) hipLaunchKernelGGL(armour_rel_pack_kernel, dim3((h->rel2_max_count + 63) / 64, h->B), dim3(64), 0, h->stream, a, dfc);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(e1, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        h->rel2_ms = ms;
        h->rel2_fresh = true;
    }
    return ARMOUR_OK;
}

// the solver's row lists of the current problem set as the persistent kernel takes them (solver.hip)
int armour_solver_lists(ArmourPlanner* h, p2::SparseList* sl, const int** tq_tiles, const int** tq_count, int* tq_cap) {
    const int rc = armour_relevance_build(h, true);
    if (rc != ARMOUR_OK) return rc;
    sl->rows = h->d_rel2_rows; sl->count = h->d_rel2_count; sl->packed = h->d_rel2_packed; sl->pack_off = h->d_rel2_pack_off;
    *tq_tiles = h->d_rel2_tq_tiles; *tq_count = h->d_rel2_count + (size_t)h->B; *tq_cap = h->rel2_tq_cap;
    return ARMOUR_OK;
}

extern "C" int armour_get_row_relevance(ArmourPlanner* h, uint8_t* relevant, int32_t* n_relevant_collision_rows, double* ms) {
    if (!h || !h->ready) { armour_set_error("no problem set: call armour_set_problems first"); return ARMOUR_ESTATE; }
    HIPCHK(hipSetDevice(h->device));
    const int rc = armour_relevance_build(h, false);
    if (rc != ARMOUR_OK) return rc;
    if (relevant) HIPCHK(hipMemcpy(relevant, h->d_rel, (size_t)h->B * h->m, hipMemcpyDeviceToHost));
    if (n_relevant_collision_rows) for (int b = 0; b < h->B; b++) n_relevant_collision_rows[b] = h->h_rel_count[b];
    if (ms) *ms = h->rel_ms;
    return ARMOUR_OK;
}

extern "C" int armour_get_solver_rows(ArmourPlanner* h, uint8_t* solver_rows, int32_t* n_collision_rows, int32_t* n_torque_rows, double* ms) {
    if (!h || !h->ready) { armour_set_error("no problem set: call armour_set_problems first"); return ARMOUR_ESTATE; }
    HIPCHK(hipSetDevice(h->device));
    const int rc = armour_relevance_build(h, true);
    if (rc != ARMOUR_OK) return rc;
    if (solver_rows) HIPCHK(hipMemcpy(solver_rows, h->d_rel2, (size_t)h->B * h->m, hipMemcpyDeviceToHost));
    for (int b = 0; b < h->B; b++) {
        if (n_collision_rows) n_collision_rows[b] = h->h_rel2_count[b];
        if (n_torque_rows) n_torque_rows[b] = h->h_rel2_tq_count[b];
    }
    if (ms) *ms = h->rel_ms + h->rel2_ms;
    return ARMOUR_OK;
}

// the culled form of armour_eval_violations_device (api.hip): torque + limit blocks of the fused evaluation, the listed collision rows, the
// row test over the same rows
int armour_eval_violations_culled(ArmourPlanner* h, const double* d_k, ArmourViolation* d_out, hipStream_t st) {
    int rc = armour_relevance_build(h, false);
    if (rc != ARMOUR_OK) return rc;
    const P2Tables tb = armour_make_tables(h);
    rc = armour_p2_launch(tb, h->max_link, h->max_torque, h->h_plane_skip.data(), d_k, h->d_g, nullptr, st, 1, 0, 0, 0, /*skip_collision_blocks=*/true);
    if (rc != ARMOUR_OK) return rc;
    RelTables a;
    a.tb = tb;
    a.lo = h->d_bounds; a.hi = h->d_bounds + (size_t)h->B * h->m;
    a.rel = h->d_rel; a.rows = h->d_rel_rows; a.count = h->d_rel_count; a.rows_res = h->d_rel_rows_res; a.count_res = h->d_rel_count + (size_t)h->B;
    a.packed = h->d_rel_packed; a.pack_off = h->d_rel_pack_off;
    if (h->rel_max_count > 0)
        hipLaunchKernelGGL(armour_sparse_collision_g_kernel, dim3((h->rel_max_count + 63) / 64, h->B), dim3(64), 0, st, a, d_k, h->d_g);
    SparseViolArgs v;
    v.m = h->m; v.row0 = h->row0; v.Q = h->Q; v.n_checked = armour_checked_collision_rows(h);
    v.torque_slack = h->params.torque_violation_threshold; v.collision_slack = h->params.collision_violation_threshold;
    v.g = h->d_g; v.lo = a.lo; v.hi = a.hi; v.rows_res = a.rows_res; v.count_res = a.count_res; v.k = d_k; v.n = h->n; v.out = d_out;
    hipLaunchKernelGGL(armour_sparse_violation_kernel, dim3(h->B), dim3(256), 0, st, v);
    HIPCHK(hipGetLastError());
    return ARMOUR_OK;
}
