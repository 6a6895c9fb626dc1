// Shared definitions of libarmour_hip.so: the planner handle, the device tables the kernels read,
// and error plumbing.  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/armour_hip.h"

#define ARMOUR_NPLANES 36
#define ARMOUR_PLANE_COMPONENTS 5  // Ax, Ay, Az, d, delta

void armour_set_error(const char* fmt, ...);

// The half-space table of one problem, layout v2 (round 4).  The row index q = (l*T + t)*O + o is the fastest axis and every array
// of rows is Qs = armour_row_stride(Q) long -- Q rounded up to 16 doubles, so that each array starts on a 128-B line (with the
// unpadded Q of rounds 1-3 every second array of configs[2] started 64 B into a line and a wave's 512-B read spanned 5 lines
// instead of 4).  Components are PAIRED so that the fused evaluation reads 16 B per lane (1 KB per wave instruction):
//   AB  [36][Qs] double2 {Ax, Ay}
//   CD  [36][Qs] double2 {Az, delta}           delta of planes 0..20 only (an obstacle generator in the pair)
//   DLL [ 8][Qs] double2 {delta of link x link plane 21 + 2i, delta of plane 22 + 2i}
//   D   [36][Qs] double  d                     (read only by launches that do not recompute d = A.c, i.e. < 8 problems)
// consecutive planes of a row are Qs*16 B apart: a wave's requests still land on different HBM pages / channels.  (A tiled variant
// that made each block's rows contiguous was measured 18 % slower at B=128, O=50: it concentrates a block on few channels.)
// armour_plane_index() is the single definition of this layout: the index, in doubles, of component c of plane p of row q.
#define ARMOUR_FIRST_LL_PLANE 21
#define ARMOUR_N_LL_PLANES 15
__host__ __device__ inline int armour_row_stride(int Q) { return (Q + 15) & ~15; }
__host__ __device__ inline size_t armour_planes_ab_offset(int Q) { (void)Q; return 0; }
__host__ __device__ inline size_t armour_planes_cd_offset(int Q) { return (size_t)2 * ARMOUR_NPLANES * armour_row_stride(Q); }
__host__ __device__ inline size_t armour_planes_dll_offset(int Q) { return (size_t)4 * ARMOUR_NPLANES * armour_row_stride(Q); }
__host__ __device__ inline size_t armour_planes_d_offset(int Q) { return (size_t)(4 * ARMOUR_NPLANES + 16) * armour_row_stride(Q); }
__host__ __device__ inline size_t armour_planes_per_problem(int Q) { return (size_t)(5 * ARMOUR_NPLANES + 16) * armour_row_stride(Q); }
__host__ __device__ inline size_t armour_plane_index(int Q, int q, int p, int c) {
    const size_t Qs = (size_t)armour_row_stride(Q);
    if (c < 2) return ((size_t)p * Qs + q) * 2 + c;
    if (c == 2) return armour_planes_cd_offset(Q) + ((size_t)p * Qs + q) * 2;
    if (c == 3) return armour_planes_d_offset(Q) + (size_t)p * Qs + q;
    if (p < ARMOUR_FIRST_LL_PLANE) return armour_planes_cd_offset(Q) + ((size_t)p * Qs + q) * 2 + 1;
    return armour_planes_dll_offset(Q) + ((size_t)((p - ARMOUR_FIRST_LL_PLANE) >> 1) * Qs + q) * 2 + ((p - ARMOUR_FIRST_LL_PLANE) & 1);
}
// The pair order of RT/CollisionChecking.cu:26-39 puts the 3 obstacle generators first, so planes 21..35 pair two of the
// 6 link generators: their NORMALS do not depend on the obstacle.  They are also kept once per (link, time step) in
// planes_ll[b][lt][48] (lt = l*T + t; entry 3*(p - 21) + c, c in {Ax,Ay,Az}; 45 used, 384 B = three lines per (link, time step)),
// which P2 reads instead of O identical copies: the 64 rows of a collision block touch 64/O + 1 such records.
// tables of at least this many collision rows (or of >= 8 problems) carry no d column: the fused evaluation recomputes d = A.c
#define ARMOUR_RECOMPUTE_D_ROWS 32768
#define ARMOUR_LL_RECORD 48
__host__ __device__ inline size_t armour_planes_ll_per_problem(int JT) { return (size_t)ARMOUR_LL_RECORD * (size_t)JT; }
__host__ __device__ inline size_t armour_plane_ll_index(int JT, int lt, int pll, int c) { (void)JT; return (size_t)lt * ARMOUR_LL_RECORD + (size_t)pll * 3 + c; }

#define HIPCHK(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t e__ = (expr);                                                                         \
        if (e__ != hipSuccess) {                                                                         \
            armour_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return ARMOUR_EDEVICE;                                                                       \
        }                                                                                                \
    } while (0)

// Reach-set tables of B problems as the P2 kernel reads them (all device pointers).
//
// Final link / torque PZs hold only k-dependent monomials (key < 2^(2n)); keys are stored as u32.
//   link  index: (b*J + l)*T + t     torque index: (b*n + j)*T + t
// Half-space table: component pairs x plane p in [0,36) x row q = (l*T + t)*O + o -- the row index is the fastest axis and
// is also the output row order of the collision block (RT/NLPclass.cu:117-164: g[nT + (l*T+t)*O + o]).
// armour_plane_index() above is the single definition of this layout.
struct P2Tables {
    int B, T, J, n, O, Q, m;
    int capL, capT;
    int row0;  // first collision row: n*T (after the torque rows), or 0 in ARMTD comparison mode (no torque rows)
    const int* link_count;
    const double* link_center;  // [..][3]
    const double* link_indep;   // [..][3]
    const uint32_t* link_keys;  // [..][capL]
    const double* link_coeff;   // [..][capL][3]
    const int* tq_count;
    const double* tq_center;    // [..]
    const double* tq_indep;     // [..]
    const uint32_t* tq_keys;    // [..][capT]
    const double* tq_coeff;     // [..][capT]
    const double* planes;       // [B][armour_planes_per_problem(Q)], see armour_plane_index
    const double* obs_center;   // [B][3][O] obstacle centres; when non-null P2 computes d = A.c itself instead of reading it (tables built by P1)
    const double* planes_ll;    // [B][J*T][48], see armour_plane_ll_index; used when ll_shared != 0
    int ll_shared;
    int ex_allowed;  // ARMOUR_OPT_P2_EX of the handle the tables belong to
    int mode;  // ARMOUR_MODE_*: which trajectory the joint-limit rows belong to
    const unsigned long long* plane_skip;  // [B] bit p set: plane p is degenerate or an exact +-duplicate of an earlier plane in EVERY row of the problem
    const double* bez;          // [B][3][n] : q0, Tqd0, TTqdd0  (ARMTD mode: q0, qd0, k_range)
    double k_range[ARMOUR_MAX_FACTORS];
    double duration;
};

// Which planner the current problem set belongs to.  ARMOUR: degree-5 Bezier trajectory, online JRS, torque rows (RT/).
// ARMTD: the comparison planner (CMP/ = kinova_planner_realtime_armtd_comparison): constant-acceleration trajectory,
// cos/sin JRS from the caller's offline tables, no torque rows.
#define ARMOUR_MODE_ARMOUR 0
#define ARMOUR_MODE_ARMTD 1

// device buffers of the device-resident armour_solve (solver_device.hip), grown on demand and kept across solves
struct SolveDeviceWork {
    unsigned char* ctl = nullptr; size_t ctl_cap = 0;            // one block: control words | goals | SolveArgs (one copy per solve)
    unsigned char* blk_word = nullptr; size_t word_cap = 0;
    int words_clean = 0;                                         // the last launch ended normally: every block has cleared its flag word
    unsigned char* blk_rows = nullptr; size_t blk_rows_cap = 0;
    unsigned char* qp_rows = nullptr; size_t qp_rows_cap = 0;
    unsigned char* flags = nullptr; size_t flags_cap = 0;
};

struct ArmourPlanner {
    ArmourRobot robot;
    ArmourParams params;
    ArmourLimits lim;
    ArmourUltimateBound ub;
    int device = 0;
    hipStream_t stream = nullptr;
    int T = 0, J = 0, n = 0;
    // current problem set
    int B = 0, O = 0, Q = 0, m = 0;
    bool ready = false;
    int mode = ARMOUR_MODE_ARMOUR;
    // no torque rows, forward kinematics only: the comparison planner (CMP/), or the ARMOUR trajectory with TURN_OFF_INPUT_CONSTRAINTS
    // (ArmourParams.input_constraints_off: RT/Parameters.h:46-47, RT/armour_main.cu:115,149-165, RT/NLPclass.cu:46-54)
    bool no_torque() const { return mode == ARMOUR_MODE_ARMTD || params.input_constraints_off != 0; }
    int row0 = 0;                   // rows before the collision block (n*T torque rows, or 0 in ARMTD mode)
    std::vector<double> h_krange;   // ARMTD mode: [B][n] acceleration range of each problem's JRS tables
    double* d_jrs = nullptr;        // ARMTD mode: [B][n][6][T] c/g/r of cos, then of sin (the order of armtd.in)
    size_t jrs_cap = 0;
    // page-locked host scratch of armour_solve (k, g, jac mirrors), grown on demand and kept across solves
    void* solve_pin[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // k, g, jac mirrors; violation sums, row counts, compact rows; [6] k and [7] records of armour_eval_violations; [8] the reach-set build's read-back, [9] its status words
    size_t solve_pin_bytes[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    bool stats_fresh = false;       // armour_p1_build has read the table statistics back itself (armour_refresh_table_stats has nothing to do)
    int opt_p1_build = 0;           // ARMOUR_OPT_P1_BUILD: 0 automatic, 1 per time step, 2 time-vectorised
    double opt_p1_work_mb = 0;      // ARMOUR_OPT_P1_WORK_MEMORY_MB: cap on the time-vectorised build's arena, MiB (0: none)
    int opt_p1_keep_work = 1;       // ARMOUR_OPT_P1_KEEP_WORK_MEMORY: 1 the device's shared work arena stays allocated between builds, 0 this handle's builds release it
    // launch-shape options ARMOUR_OPT_FIRST_TUNING .. ARMOUR_OPT_LAST_TUNING (include/armour_hip.h), indexed by option - FIRST; defaults: armour_tuning_defaults
    double tuning[ARMOUR_OPT_LAST_TUNING - ARMOUR_OPT_FIRST_TUNING + 1];
    int tune(int option) const { return (int)tuning[option - ARMOUR_OPT_FIRST_TUNING]; }
    double tune_f(int option) const { return tuning[option - ARMOUR_OPT_FIRST_TUNING]; }
    ArmourViolation* d_viol = nullptr; size_t viol_cap = 0;   // [B] records of armour_eval_violations
    // row relevance of the current problem set (relevance.hip): the mask [B][m], the relevant collision rows of every problem in ascending
    // order [B][Q] and their counts; built on first use (armour_get_row_relevance, the culled armour_eval_violations)
    unsigned char* d_rel = nullptr; size_t rel_cap = 0;
    int* d_rel_rows = nullptr; size_t rel_rows_cap = 0;
    int* d_rel_count = nullptr; size_t rel_count_cap = 0;
    double* d_rel_packed = nullptr; size_t rel_packed_cap = 0;       // the listed rows' plane entries, packed (relevance.hip)
    long long* d_rel_pack_off = nullptr; size_t rel_pack_off_cap = 0;   // [B][2]: offset and row stride of every problem's block, in doubles
    int* d_rel_rows_res = nullptr; size_t rel_rows_res_cap = 0;   // the same rows by (row index mod 256): what the culled row test iterates
    std::vector<int> h_rel_count;
    // the solver's mask (rows that can pass armour_solve's candidate filter for some k), its row list, packed entries and torque tile list
    unsigned char* d_rel2 = nullptr; size_t rel2_cap = 0;
    int* d_rel2_rows = nullptr; size_t rel2_rows_cap = 0;
    int* d_rel2_count = nullptr; size_t rel2_count_cap = 0;        // [B] listed collision rows | [B] listed torque tiles
    double* d_rel2_packed = nullptr; size_t rel2_packed_cap = 0;
    long long* d_rel2_pack_off = nullptr; size_t rel2_pack_off_cap = 0;
    int* d_rel2_tq_tiles = nullptr; size_t rel2_tq_tiles_cap = 0;  // [B][rel2_tq_cap]
    std::vector<int> h_rel2_count, h_rel2_tq_count;
    bool rel2_fresh = false;
    double rel2_ms = 0;
    int rel2_max_count = 0, rel2_tq_cap = 0;
    bool rel_fresh = false;
    double rel_ms = 0;
    int rel_max_count = 0;
    SolveDeviceWork solve_dev;
    double* d_bounds = nullptr;      // [2][B][m] g_l, g_u for the solver's device-side scan (uploaded on the first solve of a problem set)
    bool bounds_on_device = false, bounds_on_host = false;
    bool tables_from_host = false;   // armour_debug_load_tables: plane normals are the caller's, not necessarily unit vectors (relevance.hip)
    std::vector<double> h_gl, h_gu;  // host copy of the same bounds (valid while bounds_on_host: the host-driven solver form fills it)
    std::vector<double> h_q0, h_qd0, h_qdd0, h_qdes;  // [B][n]
    std::vector<double> h_torque_radius;              // [B][n][T]
    double* d_tr_stage = nullptr; size_t tr_stage_cap = 0;   // the same on the device, for the kernel that fills the bounds (armour_upload_bounds)
    std::vector<double> h_link_gens;                  // [B][T][J][18]
    std::vector<unsigned long long> h_plane_skip;     // [B] host copy of d_plane_skip
    std::vector<double> h_prune_margin;               // [B] how close the last build's simplify() verdicts came to flipping (armour_get_prune_margin)
    // device tables (sized for allocB x allocO)
    int allocB = 0, allocO = 0;
    int* d_link_count = nullptr;
    double* d_link_center = nullptr;
    double* d_link_indep = nullptr;
    uint32_t* d_link_keys = nullptr;
    double* d_link_coeff = nullptr;
    int* d_tq_count = nullptr;
    double* d_tq_center = nullptr;
    double* d_tq_indep = nullptr;
    uint32_t* d_tq_keys = nullptr;
    double* d_tq_coeff = nullptr;
    double* d_planes = nullptr;
    double* d_obs_center = nullptr;
    int d_from_center = 0;  // the d column of the table is exactly A.c of the stored normals and obs_center (tables built by P1)
    int planes_lean = 0;    // the table holds only what the fused evaluation reads (p1_reach.hip planes_of_group); armour_get_hyperplanes rebuilds the full one
    int planes_have_d = 1;  // its d column is stored (loaded tables, full tables, lean tables of < 8 problems)
    double planes_ms = 0;   // device time of the half-space kernels of the last build
    double* d_planes_ll = nullptr;
    int ll_shared = 0;  // the link x link normals of the loaded table are identical over the obstacles (always so for tables built by P1)
    unsigned long long* d_plane_skip = nullptr;
    double* d_bez = nullptr;
    // staging for the host-pointer API
    double* d_k = nullptr;
    double* d_g = nullptr;
    double* d_jac = nullptr;
    double build_ms = 0;
    // where the LAST build's search for a launch shape ended (a sort-buffer overflow sends a build to the next larger shape): the next build of this
    // handle -- the same robot, a similar problem -- starts there instead of repeating the failed launches (8-factor arms on the halved key buffers of
    // the 128-bit build: 4 launches, 67 ms, for a 25 ms build).  0 = from the beginning.
    int p1_step_cap_hint = 0, p1_tv_shape_hint = 0;
    bool p1_two_cu_off = false;   // a two-CU build of this handle lost its helper blocks once (ERR_HELPER): one CU per time step from then on
    // the hints only ever grow inside a run of like builds: they are dropped when the problem set changes class (B, T or O), with every
    // armour_set_option, and every 64th build (one hard problem must not pin a long-lived handle to larger buffers for good: ADVICE r5)
    int p1_hint_B = 0, p1_hint_T = 0, p1_hint_O = 0, p1_hint_builds = 0;
    int build_info[4] = {0, 0, 0, 0};          // armour_get_build_info: kernel of the last reach-set build, waves per block, sort-buffer entries, launches
    int max_link = 0, max_torque = 0;          // largest monomial counts in the current tables (LDS sizing of P2)
    long long sum_link = 0, sum_torque = 0;
    // P1 workspace (p1_reach.hip)
    void* p1 = nullptr;
    // instantiated graphs of armour_eval_g_jac_device_steps, valid for one problem set (api.hip)
    struct StepsGraph { const double* d_k; int steps; double* d_g; double* d_jac; hipGraphExec_t exec; unsigned long long last_use; };
    std::vector<StepsGraph> step_graphs;
    unsigned long long graph_clock = 0;
};

// api.hip: the options' defaults / accepted ranges; the one place that reads the tracing variables of the environment
void armour_tuning_defaults(double* tuning);
bool armour_trace_p1();      // ARMOUR_P1_TRACE
bool armour_trace_solve();   // ARMOUR_SOLVE_TIMING

// p2_eval.hip
int armour_p2_launch(const P2Tables& tb, int max_link, int max_torque, const unsigned long long* h_skip, const double* d_k, double* d_g, double* d_jac, hipStream_t stream,
                     int steps = 1, long long k_stride = 0, long long g_stride = 0, long long j_stride = 0, bool skip_collision_blocks = false);
// relevance.hip
int armour_relevance_build(ArmourPlanner* h, bool for_solver);   // (relevance.hip; for_solver: also the lists armour_solve's culled device form walks)
void armour_relevance_free(ArmourPlanner* h);
int armour_eval_violations_culled(ArmourPlanner* h, const double* d_k, ArmourViolation* d_out, hipStream_t st);
int armour_refresh_table_stats(ArmourPlanner* h);
// collision rows the feasibility re-check looks at (all Q in ARMOUR mode; the first (n-1)*T*O in ARMTD mode, CMP/NLPclass.cu:391-402)
int armour_checked_collision_rows(const ArmourPlanner* h);
// page-locked scratch slot of the handle with at least `bytes` bytes (registered for the zero-copy eval path); nullptr on failure
double* armour_handle_pinned(ArmourPlanner* h, int slot, size_t bytes);
int armour_p2_slice_links_launch(const P2Tables& tb, const double* d_k, double* d_centers, hipStream_t stream);
P2Tables armour_make_tables(const ArmourPlanner* h);
// g_l / g_u of the current problem set in h->d_bounds ([2][B][m]): filled by a kernel once per problem set (api.hip)
int armour_upload_bounds(ArmourPlanner* h);
// ARMOUR_P1_TRACE: host-side stamps of one armour_set_problems* call (where its wall time goes beside the kernels), printed by the call when it returns
struct BuildStamps { double t[12]; const char* name[12]; int n; };
BuildStamps& armour_build_stamps();
void armour_build_stamp(const char* name);
int armour_bounds_launch(ArmourPlanner* h, const double* d_torque_radius);   // the same kernel queued on the handle's stream (the reach-set build calls it)

// p1_reach.hip
int armour_p1_build(ArmourPlanner* h, const double* obstacles);  // h->mode selects the ARMOUR or the ARMTD chain
void armour_p1_free(ArmourPlanner* h);
int armour_p1_full_planes(ArmourPlanner* h, double* d_full);
int armour_p1_debug_pz_op(ArmourPlanner* h, int op, int nops, const int* sz, const int* cnt, const uint64_t* const* keys,
                          const double* const* coef, const double* cen, const double* ind, const double* ind2, const double* consts,
                          int r, int out_cap, uint64_t* out_keys, double* out_coef, double* out_misc);
