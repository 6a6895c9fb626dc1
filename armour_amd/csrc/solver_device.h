// Interface between solver.hip (armour_solve, host side) and solver_device.hip (the persistent SQP kernel).
#pragma once
#include "common.h"
#include "p2_tiles.h"
#include "p2_sparse.h"
#include "solver_common.h"

// per-problem control block in device memory: the group's barrier, the leader's command and the phase accumulators
struct alignas(128) SolveCtl {
    unsigned go;          // 8 * (highest phase the leader has released) + what that phase does (CMD_* of solver_device.hip)
    int pad0;
    double x[slv::NV];    // the point the released phase evaluates
};
// what one block hands the leader at the end of a phase (its own cache line: no contention)
struct alignas(64) BlockWord {
    unsigned flag;        // phases this block has finished (the group's barrier)
    int count;            // candidate rows in the block's slot
    int bad;              // rows outside the finalize_solution slacks
    int pad0;
    long long viol;       // fixed-point L1 violation of the block's rows
};

struct SolveArgs {
    P2Tables tb;
    p2::P2Launch lp;
    int nb, n_tiles;               // blocks per problem; row tiles per problem
    // culled != 0 (ARMOUR_OPT_SOLVE_CULL; relevance.hip): a problem's tiles are its LISTED torque rows, then its listed collision rows, in tiles of
    // P2_BLOCK (one thread per row: p2_sparse.h), then the limit rows -- the rows that can pass the candidate filter for some k; every other row
    // adds nothing to any quantity the solver reads (no violation, no candidate, inside the slacks), so the iterates are those of the full form
    int culled;
    p2::SparseList sl;
    const int* tq_tiles; const int* tq_count; int tq_cap;   // [B][tq_cap] listed torque rows, ascending | [B]
    int b0;                        // first problem of this launch: a batch too large to give every problem enough co-resident blocks runs as
                                   // several launches back to back, each over problems [b0, b0 + gridDim.x / nb) of the same tables
    int cap_blk, cap_rows;         // candidate rows a block / a problem may hand over
    int lds_rows;                  // candidate rows the dynamic LDS holds during the leader's step (NV + 1 doubles each)
    const double* lo; const double* hi;   // bounds [B][m]
    double* g; double* jac;        // [B][m], [B][m][n]
    SolveCtl* ctl;                 // [B]
    BlockWord* blk_word;           // [B][nb]
    slv::SolveRow* blk_rows;       // [B][nb][cap_blk]
    slv::SolveRow* qp_rows;        // [B][cap_rows]
    unsigned char* flags;          // [B][2][cap_rows + 2 NV]
    const double* q_des;           // [B][n]
    ArmourSolveResult* out;        // [B]
    int continuous_mask;
    int max_iter, max_ls;
    int n_checked_collision;       // collision rows finalize_solution re-checks (all of them; ARMTD: the first (n-1) links')
    double t_plan, cost_scale, tol, torque_slack, collision_slack;
    long long budget_ticks;        // wall-clock budget in wall_clock64() ticks; < 0: none
    long long hard_ticks;          // > 0.  No wait inside the kernel outlasts this many ticks since the block started: a block that
                                   // never hears from its leader leaves, a leader that never hears from a block hands the problem
                                   // back with status -1 (the host form redoes the solve) -- the kernel always drains
    long long* stamps;             // development (ARMOUR_SOLVE_TIMING): [B][64] ticks since kernel start at (barrier passed, leader step done) of each phase; or null
};

struct SolvePlan {
    p2::P2Launch lp;
    size_t smem;
    bool six;
    int n_tiles;
    int capacity;                  // co-resident blocks of the kernel on this device (0: no cooperative launch)
    double ticks_per_ms;
    const void* fn;
};

int armour_p2_plan(const P2Tables& tb, int max_link, int max_torque, const unsigned long long* h_skip, int steps, long long k_stride,
                   long long g_stride, long long j_stride, p2::P2Launch* lp_out, size_t* smem_out, bool* dfc_out, bool* six_out, bool* exact_out);
// b_launch: problems per launch the occupancy choice is made for (the whole batch, or a sub-batch of it)
// n_tiles_override > 0: the tiles a problem of the launch has at most (the culled form's count, from the row lists)
int armour_solve_device_capacity(const P2Tables& tb, int max_link, int max_torque, const unsigned long long* h_skip, int device, int b_launch, int waves_per_simd, SolvePlan* plan,
                                 int n_tiles_override = 0);
int armour_solver_lists(ArmourPlanner* h, p2::SparseList* sl, const int** tq_tiles, const int** tq_count, int* tq_cap);   // relevance.hip
// d_args: the SolveArgs in device memory (the kernel reads them through the pointer)
int armour_solve_device_launch(const SolveArgs* d_args, int nb, const SolvePlan& plan, int B, hipStream_t stream);
