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

// a row that cannot become active for any |d|_inf <= 2 is left out of the QP (l1 = sum_j |J_ij|)
__host__ __device__ inline bool row_upper_candidate(double gi, double ui, double l1) { return ui < 1e18 && gi + 2.0 * l1 > ui; }
__host__ __device__ inline bool row_lower_candidate(double gi, double li, double l1) { return li > -1e18 && gi - 2.0 * l1 < li; }

}  // namespace slv
