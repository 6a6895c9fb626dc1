// Pieces shared by the two forms of armour_solve (solver.hip: QPs on the host; solver_device.hip: the whole SQP iterate
// in one persistent kernel): the candidate-row record, the row filter, and the fixed-point violation sum.
#pragma once
#include <cmath>

#include "common.h"

namespace slv {

constexpr int NV = ARMOUR_MAX_FACTORS;

// One inequality a'd >= v of the QP as the scan hands it over: row `idx` of g, side 0 = upper bound (a = -J_i,
// v = g_i - hi_i), side 1 = lower bound (a = J_i, v = lo_i - g_i).  v > 0 means violated at the linearisation point.
struct SolveRow {
    int idx, side;
    double v;
    double a[NV];
};

// The L1 violation of g_l <= g <= g_u is accumulated in 2^-32 fixed point (rows clamped at 2^16): integer addition is
// associative, so the sum -- which steers the merit function and the convergence test -- is the same number whatever
// the reduction order: per-segment sums on the host path, per-block atomics in the persistent kernel, any batch size,
// any block count.  Resolution 2.3e-10, far below the solver's tolerances (1e-4 .. 1e-7).
constexpr double kViolScale = 4294967296.0;
__host__ __device__ inline long long viol_to_fixed(double v) {
    if (!(v > 0.0)) return 0;
    if (!(v < 65536.0)) v = 65536.0;  // (also NaN)
    return llrint(v * kViolScale);
}
__host__ __device__ inline double viol_from_fixed(long long s) { return (double)s / kViolScale; }

// violation of one row: what the fixed-point sum adds for it
__host__ __device__ inline long long row_violation(double gi, double li, double ui) {
    if (gi > ui) return viol_to_fixed(gi - ui);
    if (gi < li) return viol_to_fixed(li - gi);
    return 0;
}

// Steps one QP attempt may take before it is given up as infeasible (and the next elastic attempt is tried).  An attempt that ends feasible takes
// <= 18 steps on every problem tried (<= 8 variables: each step adds or drops one row); the attempts that run longer are cycling on degenerate
// rows and never end feasible -- round 5 counted them over batches of 64 ... 256 random worlds, O = 10 ... 50: 0 ... 3 per batch, every one ran
// into the limit, which was 400 until then.  Since the row culling a batch waits for its slowest leader, i.e. for exactly those attempts.
constexpr int kQpMaxSteps = 100;

// The box-clipped minimiser (round 6).  The QP of an SQP step is  min 1/2 d'Gd + g0'd  over the variables' box and the linearised rows, with G
// DIAGONAL: without the rows it is separable, and its minimiser is the unconstrained one clipped to the box, variable by variable.  If that point
// satisfies every row it is the QP's solution (the minimiser over a superset that lies in the set) -- and on the reference's own worlds it nearly
// always does: the waypoint is 1 rad away, every trajectory parameter goes to its bound, no collision row is near (7 dual active-set steps per QP, and 7
// more to find the same corner again from the next iterate: 74 of the persistent kernel's 118 us for one problem, profiles/r06_solve.txt).  Both forms of
// the solver try it first, with this arithmetic, and verify it against every row exactly as a Goldfarb-Idnani result is verified; a point that fails goes
// through the active-set method as before.  Returns the largest multiplier of the clipped variables' bounds (G_j |d_j - d0_j|: stationarity of the
// bound row), which the merit function's penalty takes as it takes the active-set method's.
__host__ __device__ inline double box_clipped_step(int n, const double* Gd, const double* invG, const double* g0, const double* lo, const double* hi, double* d) {
    double mm = 0.0;
#pragma unroll
    for (int j = 0; j < NV; j++) {   // (a compile-time trip count: on the device the arrays are registers, and a run-time index would put them in scratch)
        if (j < n) {
            const double d0 = -g0[j] * invG[j];
            const double dj = d0 < lo[j] ? lo[j] : (d0 > hi[j] ? hi[j] : d0);
            const double uj = fabs(dj - d0) * Gd[j];
            if (uj > mm) mm = uj;
            d[j] = dj;
        }
    }
    return mm;
}

// a row that cannot become active for any |d|_inf <= 2 is left out of the QP (l1 = sum_j |J_ij|)
__host__ __device__ inline bool row_upper_candidate(double gi, double ui, double l1) { return ui < 1e18 && gi + 2.0 * l1 > ui; }
__host__ __device__ inline bool row_lower_candidate(double gi, double li, double l1) { return li > -1e18 && gi - 2.0 * l1 < li; }

}  // namespace slv
