/*
 * armour_hip.h -- C ABI of libarmour_hip.so, the MI355X (gfx950) implementation of ARMOUR's
 * per-planning-iteration reachable-set + constraint pipeline.
 *
 * This is the drop-in boundary for the hot path of roahmlab/armour (RT/ =
 * kinova_src/kinova_simulator_interfaces/kinova_planner_realtime/).  The reference has no FFI for
 * this path: it spawns `armour_main` and exchanges text files (KSI/uarmtd_planner.m:158-219); its
 * only in-process native precedent is the MEX gateway of the robust controller
 * (MEXC/kinova_controller.cpp:19-84).  The entry points below are what a MEX / ctypes / cgo binding
 * of the planner would bind; each cites the reference interface it replaces.  INTEGRATION.md shows
 * the MEX stub and the file-protocol CLI built on them.
 *
 * Conventions: plain pointers and sizes; fp64; all host buffers caller-owned; the handle owns all
 * device memory; every function returns 0 on success or a negative ARMOUR_E* code and never
 * throws; armour_last_error() returns a thread-local message.  A handle is bound to one device and
 * one HIP stream; calls on different handles may run concurrently from different host threads.
 *
 * Batch semantics: a handle holds B independent planning problems (B >= 1) that share the robot,
 * the parameters and the obstacle count O.  B = 1 is the reference's use.  Arrays are
 * problem-major: q0[B][n], obstacles[B][O][12], k[B][n], g[B][m], jac[B][m][n].
 *
 * Constraint rows (RT/NLPclass.cu:46-49,117-164,304-320), m = n*T + J*T*O + 4n:
 *   [0, nT)                torque centre      g[t*n + j]
 *   [nT, nT + J*T*O)       collision          g[nT + (l*T + t)*O + o] = -max_p(...)  (feasible <= 0)
 *   then n rows min position, n rows max position, n rows min velocity, n rows max velocity.
 * Jacobian: dense, row-major values[row*n + col] exactly as IPOPT's `values` (RT/NLPclass.cu:348-357).
 */
#ifndef ARMOUR_HIP_H
#define ARMOUR_HIP_H

#include <stdint.h>
#include "armour_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ARMOUR_OK 0
#define ARMOUR_EINVAL (-1)    /* bad argument */
#define ARMOUR_EDEVICE (-2)   /* HIP runtime error (message has the hipError string) */
#define ARMOUR_ECAPACITY (-3) /* a polynomial zonotope outgrew its device table; raise the capacity in ArmourLimits */
#define ARMOUR_ESTATE (-4)    /* call order violated (e.g. eval before set_problems) */

typedef struct ArmourPlanner ArmourPlanner;

/* Device-table capacities (monomials). Zero fields take the defaults in parentheses. */
typedef struct ArmourLimits {
    int32_t max_batch;          /* problems per handle (1) */
    int32_t max_obstacles;      /* O upper bound (40, RT/Parameters.h:26 MAX_OBSTACLE_NUM; any value allowed) */
    int32_t link_monomials;     /* k-only monomials kept per link PZ (32) */
    int32_t torque_monomials;   /* k-only monomials kept per torque PZ (128) */
    int32_t work_monomials;     /* monomials of an intermediate PZ during FK/RNEA (1024) */
    int32_t raw_terms;          /* unsimplified terms of one PZ product (4096) */
} ArmourLimits;

/* ---- robot / parameter presets (RT/KinovaWithoutGripperInfo.h, RT/Parameters.h) ---- */
void armour_robot_kinova_gen3_no_gripper(ArmourRobot* robot);
void armour_robot_kinova_gen3_gripper(ArmourRobot* robot);   /* RT/KinovaInfo.h: 8 links, last joint fixed */
/* CMP/FetchInfo.h: 9 links (two fixed), 7 factors, joint axes {z,y,x,y,x,y,x}; link boxes and M_max are stand-ins
 * (include/armour_robot_fetch.h).  For payload-mass uncertainty set robot->mass_uncertainty_link[i] afterwards. */
void armour_robot_fetch(ArmourRobot* robot);
/* "Fetch 8-DOF" (BASELINE configs[4]): the arm above behind a torso yaw joint -- 9 links, 8 factors.  Filled by the 128-bit-key library only
 * (libarmour_hip_k128.so, armour_abi_max_factors() == 8); the 64-bit-key library returns ARMOUR_EINVAL (its key holds seven factors, as the
 * reference's: RT/PZsparse.h:8-21).  A derived preset, not reference data: include/armour_robot_fetch.h says what is assumed. */
int armour_robot_fetch8(ArmourRobot* robot);
void armour_params_default(ArmourParams* params, int32_t num_time_steps);

/* ---- lifetime ---- */
/* Replaces the process start-up of RT/armour_main.cu:11-86 (Obstacles ctor cudaMalloc x10, constant uploads).
 * `device` is a HIP device ordinal.  limits may be NULL. */
int armour_create(const ArmourRobot* robot, const ArmourParams* params, const ArmourLimits* limits,
                  int32_t device, ArmourPlanner** out);
void armour_destroy(ArmourPlanner* h);
const char* armour_last_error(void);
/* 1 if a HIP device is visible to this process, else 0 (never fails). */
int armour_device_available(void);

/* Optional page-locked host buffers for k / g / jac: armour_eval_g_jac DMAs straight into them instead of
 * staging through the runtime's bounce buffers (the reference pays 8-16 blocking cudaMemcpy per callback,
 * RT/CollisionChecking.cu:98-132). */
int armour_alloc_pinned(uint64_t bytes, void** out);
void armour_free_pinned(void* p);

/* ---- per-handle options ---- */
/* ARMOUR_OPT_P1_BUILD selects the reach-set build kernel of armour_set_problems* (value 0, 1 or 2):
 *   0  automatic (default): one wavefront per (problem, time step) for small batches, the time-vectorised kernel (one wavefront per
 *      50-64 time steps of a problem) from B*T >= 1550 on;
 *   1  always per time step;   2  always time-vectorised.
 * TOLERANCE CONTRACT.  Both kernels run the reference's operator sequence on the same operands and produce identical monomial keys,
 * coefficients and centres bit for bit; they add the pruned amounts of simplify() (RT/PZsparse.cu:327-341) into the independent
 * radii in a different order (per-lane partial sums + a wave reduction against a running sum in key order), so radii -- and through
 * them torque_radius, the link generators, g and jac -- may differ by rounding between the two: <= 1e-12 absolute on every table
 * and output, far inside the parity tolerances (tests/test_baseline_configs.py).  Within ONE kernel results are bit-reproducible
 * and independent of the batch mates and of the batch size.  A caller that needs bit-identical tables for the same problem across
 * batch sizes (or across the ranks of a sharded batch, whose shard sizes differ) sets option 1 (or 2) on every handle. */
#define ARMOUR_OPT_P1_BUILD 1
/* ARMOUR_OPT_P1_WORK_MEMORY_MB caps the device memory the time-vectorised build takes for its work slots, in MiB (0 = no cap, the
 * default: one block of ~112 MiB per compute unit while the batch has that many groups of time steps, 28.6 GiB for a batch of 128
 * problems on a 256-CU device).  With a cap the build runs on fewer blocks, which loop over the groups -- same tables bit for bit,
 * the build time grows with the rounds (a batch of 128 problems on half the blocks: about twice).  A cap below one block's slots
 * sends batches to the step-by-step kernel (a few GiB at most).
 * The work slots are ONE arena per device, shared by all handles of the process and held by a build only while it runs: two handles
 * building batches at the same time take turns on it (the kernel fills the device anyway), so they need it once, not twice. */
#define ARMOUR_OPT_P1_WORK_MEMORY_MB 2
/* ARMOUR_OPT_P1_KEEP_WORK_MEMORY (0 | 1, default 1): 1 leaves the device's shared work slots allocated between builds (they go with the
 * device's last handle); 0 releases them at the end of every armour_set_problems* of this handle, so that between builds the process
 * holds tables only.  Releasing costs the next large build its allocation: 0.6 ms of a 128-problem build's 10.4 ms as a rule, but
 * seconds now and then, when the runtime has to get 28.6 GiB back from the driver (measured) -- hence the default. */
#define ARMOUR_OPT_P1_KEEP_WORK_MEMORY 3

/* ---- launch-shape options (round 4: these were environment variables read once per process; two handles of one process can now
 * differ, and no stray variable changes a result).  Every one of them selects HOW the same operator sequence is laid out on the
 * device, never which operators run on which operands.  TOLERANCE CONTRACT of the whole group: monomial keys, coefficients and
 * centres of every table are bit-identical for every value; options marked (r) may change the ORDER in which pruned amounts are
 * added into the independent radii (<= 1e-12 on every table and output, as ARMOUR_OPT_P1_BUILD above); unmarked options leave
 * every bit alone.  Defaults are what bench.py and the tests measure; the values are tuning / test knobs. */
/* per-step reach-set kernel (p1_reach.hip: armour_p1_chain_kernel) */
#define ARMOUR_OPT_P1_STEP_WAVES 101          /* 0 automatic (default) | 1 | 3 | 4 wavefronts per (problem, time step) block */
#define ARMOUR_OPT_P1_STEP_FREE 102           /* 1 (default) free-running role waves | 0 a block barrier per joint */
#define ARMOUR_OPT_P1_STEP_SPLIT_FK 103       /* -1 automatic (default) | 0 | 1: forward kinematics as work items of their own */
#define ARMOUR_OPT_P1_STEP_AUX3 104           /* 1 (default) | 0: the w_aux recursion on the fourth wave of four-wave blocks */
#define ARMOUR_OPT_P1_MAX_WAVES_PER_CU 105    /* 1..8 (default 4): one-wave blocks resident per compute unit; the shipped kernels hold one wave per SIMD, so values above 4 act as 4 (they matter only in two-waves-per-SIMD development builds) */
#define ARMOUR_OPT_P1_TWO_PASS 106            /* 1 (default) | 0: large batches first with 2048-entry sort buffers, overflowing items rebuilt alone */
#define ARMOUR_OPT_P1_STEP_TAIL_CROSS 108     /* 0 off | n | 10 + n: four-wave blocks of a lone problem, w x (w_aux x com) of the last n <= 4 links built by the fourth (n) / the angular (10 + n) wave once its recursion is through */
#define ARMOUR_OPT_P1_STEP_QUEUE 109          /* 1 (default): the blocks draw their items from a counter, every problem's late time steps first | 2, 3: in index order, early steps first | 0: block k builds items k, k + blocks, ... */
#define ARMOUR_OPT_P1_STEP_PAIRS 107          /* 1 (default) | 0 | 2: four-wave blocks, backward pass -- the two idle waves join the recursion waves' operators (1: when every item has a block of its own; 2: always) */
#define ARMOUR_OPT_P1_STEP_TWO_CU 123         /* 3 (default) | 0 | 1 | 2: a lone problem's time step on TWO compute units -- a helper block per (problem, time step) reruns the two
                                                 angular-velocity recursions, builds the cross products that read nothing else (w x (w_aux x p), w x (w_aux x com), w_aux x (I w); level 3,
                                                 chains of at most 7 joints: R_t w_aux x (qd e) as well) and the step's forward kinematics, and hands the products over through the L2 of the
                                                 XCD the two blocks share (checked per item; otherwise, and whenever 2 T + 7 blocks do not fit the device, one CU per step as before).
                                                 Level 2: the main block drops its own w recursion; level 3: its w_aux recursion too.  Same tables bit for bit at every level.
                                                 10 + level, 20 + level: test hooks -- the helper agrees and hands over nothing (the build must notice and start again on one CU per step) /
                                                 the helper never says it has started (the main blocks give up after ~1.5 ms and build alone: good tables, late).  30 + level: the helper of an item is placed on
                                                 another XCD than its main block (every item must then decide against two CUs: one launch, good tables, no fall-back of the handle).  After either event the handle
                                                 stays on one CU per step and armour_get_option reads 0; setting the option again re-enables it */
#define ARMOUR_OPT_P1_STEP_LEAN_BACK 124      /* 1 (default) | 0 | 2 | 3: with a time step on two CUs at level 3 -- bit 0: the helper block also runs the backward pass's f-recursion (F_i handed over as the forward pass builds them, p x (R f) handed back), the main block's pair keeps R n and the four-term sum alone; bit 1 CLEAR: the JRS of the joints past the fourth is built by one wave of each block while the recursions run their first steps */
/* time-vectorised reach-set kernel (p1_tv.inc.h: armour_p1_tv_kernel) */
#define ARMOUR_OPT_P1_TV_MIN_GROUPS 110       /* default 31: automatic choice of ARMOUR_OPT_P1_BUILD takes this kernel from B*T >= 50 * value on */
#define ARMOUR_OPT_P1_TV_WAVES 111            /* 0 automatic (default) | 1 | 3 | 4 | 8 wavefronts per block (r: 4 and 8 share walks between waves) */
#define ARMOUR_OPT_P1_TV_FREE 112             /* 1 (default) | 0: free-running role waves | a block barrier per joint */
#define ARMOUR_OPT_P1_TV_SPLIT_FK 113         /* -1 automatic (default) | 0 | 1 */
#define ARMOUR_OPT_P1_TV_DEDICATED 114        /* 1 (default) | 0: eight-wave blocks may be chosen (development builds with two waves per SIMD only) */
#define ARMOUR_OPT_P1_TV_HELP_SHIFT 115       /* 0..3 (default 0): which helper wave serves which role wave in eight-wave blocks */
#define ARMOUR_OPT_P1_TV_HELPERS 116          /* (r) 0 every walk on its own wave | 1 (default) shared walks, the owner keeps 16/32 | 2..31: it keeps value/32 */
#define ARMOUR_OPT_P1_TV_HELP_MIN 117         /* (r) default 192: walks with fewer sorted terms stay on one wave */
#define ARMOUR_OPT_P1_TV_HELP_N 118           /* (r) 1 (default) | 0: the n-recursion shares its walks as well */
#define ARMOUR_OPT_P1_TV_AUX3 119             /* 1 (default) | 0: as ARMOUR_OPT_P1_STEP_AUX3 */
#define ARMOUR_OPT_P1_TV_TAIL_CROSS 121       /* as ARMOUR_OPT_P1_STEP_TAIL_CROSS, for the four-wave blocks of the time-vectorised kernel; default 111 since the end of round 6 (the cross product of the last link on the angular wave, every moment N on the F / N wave, which waited a fifth of the forward pass: B = 128 -1 %); + 100 (n + 1): the angular wave builds the moments of the last n links */
#define ARMOUR_OPT_P1_TV_ROW_WIDTH 122        /* 0 automatic (default) | 50 | 64: doubles per row of the kernel's work slots.  Automatic = 50 when a group of time steps fits (T = 100: two groups of 50; the rows are packed, -19 % of the build's L2-miss traffic), 64 otherwise; 50 with longer groups is refused.  Every bit of every table is the same for both. */
#define ARMOUR_OPT_P1_FULL_PLANES 120         /* 0 (default) the lean half-space table | 1 every plane and component resident (armour_get_hyperplanes builds it on demand otherwise) */
/* fused evaluation (p2_eval.hip) and its host entries (api.hip) */
#define ARMOUR_OPT_P2_EX 130                  /* 1 (default) | 0: the fixed-load-count kernels for problems with exactly 24 live planes */
#define ARMOUR_OPT_STEPS_GRAPH_MIN 131        /* default 2: armour_eval_g_jac_device_steps submits >= this many steps as one graph; 0 never */
#define ARMOUR_OPT_CULL_ROWS 133              /* 0 (default) | 1: armour_eval_violations* evaluate the rows that can be violated for some k only (armour_get_row_relevance): same L1 violation, counts and verdict bit for bit; `worst` is the full evaluation's whenever a row is violated.
                                                * PRECONDITION: k inside [-1, 1]^n (the mask is a statement about the box).  armour_eval_violations (host k) takes every row when a component
                                                * lies outside; armour_eval_violations_device cannot look at k on the host: the record of a problem whose k is outside the box comes back
                                                * with feasible = -1 and worst_row = -2, and is to be re-evaluated with the option at 0 */
#define ARMOUR_OPT_PINNED_MODE 132            /* 0 (default) page-locked buffers through device staging + one DMA transfer | 1 the kernel reads / writes host memory itself */
/* armour_solve (solver.hip, solver_device.hip): iterates are bit-identical for every value (tests/test_solve.py) */
#define ARMOUR_OPT_SOLVE_SUB_TILES 140        /* default 48: row tiles per block aimed at when a batch is cut into sub-batches */
#define ARMOUR_OPT_SOLVE_DEVICE 141           /* 1 (default): the device-resident form (round 6: for every batch size; until round 5 from 2 problems on) | 0: never | 2: always */
#define ARMOUR_OPT_SOLVE_CUT_TILES 142        /* default 156: a batch whose blocks would walk more tiles each is cut */
#define ARMOUR_OPT_SOLVE_BLOCKS 143           /* 0 (default: as many as fit) | n: at most n blocks per problem */
#define ARMOUR_OPT_SOLVE_SUB_BATCH 144        /* 0 (default: by tiles) | n: sub-batches of at most n problems */
#define ARMOUR_OPT_SOLVE_ROW_CAP 145          /* 0 (default) | n: candidate-row buffers of n rows (tests: forces the overflow path) */
#define ARMOUR_OPT_SOLVE_HARD_CAP_S 146       /* 0 (default: from the time budget / iteration limit) | seconds: wall-clock cap of one persistent launch */
#define ARMOUR_OPT_SOLVE_WAVES_PER_SIMD 147   /* 0 automatic (default) | 1 | 2: register build of the persistent kernel */
#define ARMOUR_OPT_SOLVE_CULL 148              /* -1 automatic (default: from 150 000 collision rows in the batch on) | 0 | 1: the device-resident form walks only the rows that can pass
                                                * its candidate filter for some k (a second, wider mask than armour_get_row_relevance's: g_i + 2 |J_i|_1 can reach the bound); every
                                                * other row adds nothing the solver reads, so iterates and results are those of the full form bit for bit */
#define ARMOUR_OPT_FIRST_TUNING 101
#define ARMOUR_OPT_LAST_TUNING 148
/* bytes of device memory free / in all on `device` as the runtime reports them (hipMemGetInfo): what a caller sizes
 * ARMOUR_OPT_P1_WORK_MEMORY_MB against; either pointer may be NULL */
int armour_device_memory(int32_t device, uint64_t* free_bytes, uint64_t* total_bytes);
/* ARMOUR_MAX_FACTORS this library was built with: 7 (the shipped library: the reference's 64-bit monomial key, RT/PZsparse.h:8-21) or 8
 * (libarmour_hip_k128.so: 128-bit keys; every struct of armour_types.h that holds per-factor arrays is laid out for 8 there) */
int armour_abi_max_factors(void);
int armour_set_option(ArmourPlanner* h, int32_t option, double value);
/* armour_get_option: the current value of an option of this handle (status ARMOUR_EINVAL for an unknown option) */
int armour_get_option(ArmourPlanner* h, int32_t option, double* value);
/* Environment variables the library still reads (none of them changes a result): ARMOUR_P1_TRACE, ARMOUR_SOLVE_TIMING (one line of
 * timings per call on stderr), ARMOUR_WORKER_PARENT (armour_worker: exit with the parent process). */

/* ---- P1: reach-set build, once per planning iteration ---- */
/* Replaces RT/armour_main.cu:36-216 (parse armour.in, JRS, FK, RNEA x2, torque radius, half-space tables)
 * for B problems at once.  obstacles: [B][O][12], each = column-major Z=[c g1 g2 g3]
 * (KSI/uarmtd_planner.m:178; RT/CollisionChecking.cu:156-160).  Synchronous. */
int armour_set_problems(ArmourPlanner* h, int32_t B, int32_t O, const double* q0, const double* qd0,
                        const double* qdd0, const double* q_des, const double* obstacles);

/* ARMTD comparison mode: the same handle and the same callback surface for the reference's second
 * planner, kinova_planner_realtime_armtd_comparison ("CMP").  Replaces CMP/armtd_main.cu:36-216: constant-acceleration
 * trajectory q(t) = q0 + qd0 t + k t^2/2 (CMP/Trajectory.h:6-15), cos/sin JRS taken from the caller's offline tables and
 * rotated by q0 (CMP/Trajectory.cu:29-81), forward kinematics, half-space tables; no torque rows:
 * m = J*T*O + 4n (CMP/NLPclass.cu:42-43), collision rows first, then the joint-limit rows of the constant-acceleration
 * curve (CMP/Trajectory.cu:83-383).  After this call armour_get_sizes / get_bounds / eval_f / eval_grad_f / eval_g_jac* /
 * check_feasible / armour_solve and the table getters follow CMP/NLPclass.cu instead of RT/NLPclass.cu.
 *   jrs: [B][n][6][T], per joint the six rows armtd.in holds for it (KSI/uarmtd_planner.m:277-312): centre, k-generator
 *        and radius of cos(q - q0), then of sin(q - q0);  k_range: [B][n] (the number after each joint's rows, :314-315).
 * The ArmourParams of the handle supply T, the simplify threshold, the violation thresholds and the cost scale; its
 * k_range, duration and t_plan are not used in this mode. */
int armour_set_problems_armtd(ArmourPlanner* h, int32_t B, int32_t O, const double* q0, const double* qd0, const double* q_des,
                              const double* jrs, const double* k_range, const double* obstacles);

/* sizes after set_problems: n = NUM_FACTORS, m = constraint_number (RT/NLPclass.cu:46-49, get_nlp_info :62-82) */
int armour_get_sizes(const ArmourPlanner* h, int32_t* B, int32_t* n, int32_t* m);

/* ---- NLP callback surface (RT/NLPclass.h armtd_NLP : Ipopt::TNLP) ---- */
/* get_bounds_info, RT/NLPclass.cu:87-165.  x_l,x_u: [n] (shared by all problems); g_l,g_u: [B][m]. */
int armour_get_bounds(ArmourPlanner* h, double* x_l, double* x_u, double* g_l, double* g_u);
/* eval_f / eval_grad_f, RT/NLPclass.cu:207-267 (host arithmetic, 7 numbers).  k: [B][n]; f: [B]; grad_f: [B][n]. */
int armour_eval_f(ArmourPlanner* h, const double* k, double* f);
int armour_eval_grad_f(ArmourPlanner* h, const double* k, double* grad_f);
/* eval_g + eval_jac_g fused, RT/NLPclass.cu:272-396.  Host pointers; g and/or jac may be NULL.
 * Synchronous: H2D of k, one kernel launch, D2H of the requested outputs. */
int armour_eval_g_jac(ArmourPlanner* h, const double* k, double* g, double* jac);
/* Same, device-resident: d_k [B][n], d_g [B][m], d_jac [B][m][n] are device pointers on the handle's device
 * (d_g / d_jac may be NULL).  Enqueued on `stream` (a hipStream_t; NULL = the handle's own stream) and NOT
 * synchronised -- this is the entry the throughput benchmark and a device-side solver use. */
int armour_eval_g_jac_device(ArmourPlanner* h, const double* d_k, double* d_g, double* d_jac, void* stream);
/* `steps` consecutive fused evaluations enqueued back to back on `stream` without host synchronisation:
 * step s reads d_k + s*B*n (d_k holds [steps][B][n]) and overwrites d_g / d_jac.  One kernel launch per step
 * (the IPOPT iterate sequence of RT/armour_main.cu:273 with the solver's own arithmetic removed), executed in
 * order.
 * armour_prepare_steps(h, d_k, steps, d_g, d_jac) builds an instantiated hipGraph of those launches -- a chain of
 * `steps` kernel nodes -- without running it; later armour_eval_g_jac_device_steps calls with exactly these arguments
 * submit that graph (no host work per step) until the next armour_set_problems*.  The CONTENT of d_k may change
 * between calls, the pointers are part of the graph.  The four most recently used graphs are kept. */
int armour_eval_g_jac_device_steps(ArmourPlanner* h, const double* d_k, int32_t steps, double* d_g, double* d_jac,
                                   void* stream);
int armour_prepare_steps(ArmourPlanner* h, const double* d_k, int32_t steps, double* d_g, double* d_jac);
/* `points` evaluations of the SAME problems in ONE kernel launch: point s reads d_k + s*B*n and writes
 * d_g + s*B*m, d_jac + s*B*m*n (d_k [points][B][n], d_g [points][B][m], d_jac [points][B][m][n]; d_g / d_jac may be
 * NULL).  Every block keeps its share of the plane and PZ tables in registers across the points, so the tables are
 * read once per launch instead of once per point.  Results are bit-identical to `points` separate
 * armour_eval_g_jac_device calls.  For callers that know several trial points at once (line search, multi-start,
 * finite-difference checks -- the derivative checker of RT/armour_main.cu:248-251); IPOPT's own iterate sequence is
 * serial and uses the one-point entries above. */
int armour_eval_g_jac_device_multi(ArmourPlanner* h, const double* d_k, int32_t points, double* d_g, double* d_jac,
                                   void* stream);
/* finalize_solution feasibility re-check, RT/NLPclass.cu:422-538: feasible[b] = 1/0 from g[B][m] (host). */
int armour_check_feasible(ArmourPlanner* h, const double* g, int32_t* feasible);

/* ---- reduced outputs for throughput callers ---- */
/* A batch that evaluates 1e5 points/s cannot pull g and jac (2.29 MB per problem-evaluation at O = 50) over PCIe.  These entries run
 * the same fused kernel for g only, keep it on the device and hand back one record per problem: the row test of
 * armtd_NLP::finalize_solution (RT/NLPclass.cu:422-538; ARMTD mode: CMP/NLPclass.cu:391-402) applied on the device.
 *   l1_violation      sum over all rows of max(0, g_l - g, g - g_u)   (bounds of armour_get_bounds)
 *   worst, worst_row  the largest single-row violation and its row (lowest row among equals; -1 / 0.0 if no row is violated)
 *   n_violated        rows with a violation > 0
 *   n_outside_slack   rows finalize_solution rejects: torque rows beyond the bound by more than torque_violation_threshold, checked
 *                     collision rows above collision_violation_threshold, position / velocity rows outside their bounds
 *   feasible          n_outside_slack == 0: the verdict of armour_check_feasible on the same g
 * Sums run in a fixed order: the record depends on (problem, k) only, not on the batch or the device. */
typedef struct ArmourViolation {
    double l1_violation, worst;
    int32_t worst_row, n_violated, n_outside_slack, feasible;
} ArmourViolation;
/* d_k [B][n] and d_out [B] are device pointers; enqueued on `stream` (NULL = the handle's), not synchronised. */
int armour_eval_violations_device(ArmourPlanner* h, const double* d_k, ArmourViolation* d_out, void* stream);
/* host pointers, synchronous: 56 B in and 32 B out per problem instead of 8 m (1 + n) bytes */
int armour_eval_violations(ArmourPlanner* h, const double* k, ArmourViolation* out);
/* Row relevance of the current problem set -- the pruned constraint list of the reference's MATLAB path (KSI/uarmtd_planner.m:577-583: an
 * obstacle constraint is kept only if the forward occupancy can reach the buffered obstacle; :628-690: an input / limit constraint only if
 * `~(interval(...).sup < 0)`), which its C++ path does not have (RT/NLPclass.cu:272-396 evaluates every row at every iterate).
 * relevant[B][m] (row order of armour_eval_g_jac): 0 = the row cannot be violated for ANY k in [-1,1]^n -- a plane of its buffered obstacle
 * separates the whole family of sliced link centres (collision rows), the sliced torque's interval stays inside the bounds (torque rows) --
 * 1 otherwise; the 4n joint-limit rows are always 1.  Sound (a 0 is never violated: 1e-9 margin on the test), not tight.  On random worlds
 * about 2 % of the collision rows are 1.  n_relevant_collision_rows [B] and ms (device time of the test) may be NULL; so may `relevant`.
 * Computed once per problem set, on the device (relevance.hip).  ARMOUR_OPT_CULL_ROWS = 1 makes armour_eval_violations* use it. */
int armour_get_row_relevance(ArmourPlanner* h, uint8_t* relevant, int32_t* n_relevant_collision_rows, double* ms);
/* The SOLVER's rows of the current problem set: solver_rows[B][m], 1 = the row can pass armour_solve's candidate filter for some k in [-1,1]^n
 * (g_i + 2 |J_i|_1 > u_i or g_i - 2 |J_i|_1 < l_i: solver_common.h; the filter leaves out rows that cannot become active within the variables'
 * box, and this mask the rows it leaves out at EVERY k) -- a superset of armour_get_row_relevance's mask; what the culled device form of armour_solve
 * walks (ARMOUR_OPT_SOLVE_CULL).  n_collision_rows [B] and n_torque_rows [B] may be NULL; so may `solver_rows`. */
int armour_get_solver_rows(ArmourPlanner* h, uint8_t* solver_rows, int32_t* n_collision_rows, int32_t* n_torque_rows, double* ms);

/* ---- in-process multi-device batch ---- */
/* One caller thread (MATLAB / MEX, a Python host) drives several GPUs: an ArmourBatch owns one ArmourPlanner per entry of `devices`
 * (an ordinal may repeat: two handles, two streams on one device).  B independent planning problems are dealt to the handles in
 * contiguous blocks -- device slot d owns problems [first[d], first[d+1]) with first[d] = d*(B/G) + min(d, B%G), the partition of
 * armour_amd/sharding.py -- and every call runs one host thread per slot, each driving its own handle and stream; the caller's
 * arrays are problem-major exactly as for a single handle and results are gathered in place.  There is no data-path collective and
 * no device-to-device traffic (problems share nothing).  The reference has no counterpart: it runs one problem per process
 * (RT/armour_main.cu); this is the in-process form of bench.py's one-rank-per-GPU sharding. */
typedef struct ArmourBatch ArmourBatch;
int armour_batch_partition(int32_t B, int32_t n_slots, int32_t* first /* [n_slots + 1] */);   /* pure host arithmetic (no GPU needed) */
int armour_batch_create(const ArmourRobot* robot, const ArmourParams* params, const ArmourLimits* limits,
                        const int32_t* devices, int32_t n_devices, ArmourBatch** out);
void armour_batch_destroy(ArmourBatch* bt);
int armour_batch_set_option(ArmourBatch* bt, int32_t option, double value);
int armour_batch_set_problems(ArmourBatch* bt, int32_t B, int32_t O, const double* q0, const double* qd0, const double* qdd0,
                              const double* q_des, const double* obstacles);
int armour_batch_get_sizes(const ArmourBatch* bt, int32_t* B, int32_t* n, int32_t* m, int32_t* n_slots);
int armour_batch_get_bounds(ArmourBatch* bt, double* x_l, double* x_u, double* g_l, double* g_u);
int armour_batch_eval_g_jac(ArmourBatch* bt, const double* k, double* g, double* jac);             /* full outputs (parity path) */
int armour_batch_eval_violations(ArmourBatch* bt, const double* k, ArmourViolation* out /* [B] */);  /* reduced outputs */
/* declared after ArmourSolveOptions below: armour_batch_solve */
/* armour_get_prune_margin of every problem of the batch: margin[B], problem-major like every other array */
int armour_batch_get_prune_margin(ArmourBatch* bt, double* margin /* [B] */);
/* ms of the slowest slot's last reach-set build (device time); per_slot may be NULL or [n_slots] */
int armour_batch_get_build_ms(ArmourBatch* bt, double* max_ms, double* per_slot);
/* armour_get_build_info of every slot: info[4 * slot + 0..3] (zeros for a slot without problems).  All slots of a problem set are built
 * by ONE kernel: with ARMOUR_OPT_P1_BUILD left automatic the one the smallest shard would take. */
int armour_batch_get_build_info(ArmourBatch* bt, int32_t* info /* [n_slots][4] */);

/* ---- caller side of the path: the trajectory the planner hands to the controller ---- */
/* uarmtd_planner.desired_trajectory, traj_type 'bernstein' (KSI/uarmtd_planner.m:846-925, with
 * PZM/utility/match_deg5_bernstein_coefficients.m and bernstein_to_poly.m): position / velocity / acceleration at time
 * t in [0, duration] of the degree-5 Bezier curve that starts at (q0, qd0, qdd0) and ends at rest at
 * q0 + k_range .* k -- the same curve the NLP constrains (RT/Trajectory.cu:542-602).  Stateless host arithmetic;
 * any of q / qd / qdd may be NULL.  The braking fallback (k = NaN: re-use the previous plan shifted by t_plan, or
 * hold position) is caller logic, see armour_amd/planner.py desired_trajectory. */
int armour_desired_trajectory(int32_t n, const double* q0, const double* qd0, const double* qdd0, const double* k_range,
                              double duration, const double* k, double t, double* q, double* qd, double* qdd);

/* ---- the tracking controller the planner's trajectories are executed with ---- */
/* kinova_controller(Kr, alpha, V_max, r_norm_threshold, q, qd, q_des, qd_des, qdd_des[, eps]) ->  [u, tau, v]
 * (kinova_robust_controllers_mex/kinova_controller.cpp:19-84; RobustController::update, ARMOUR method,
 * robust_controller.cpp:63-168) for B states at once: tau = nominal passivity-RNEA torque, the interval RNEA over
 * masses and inertias within +-model_uncertainty bounds the model error, v = robust input from the control barrier
 * condition on V = 1/2 r'M r <= V_max, u = tau - v.  One device thread per state.  All pointers are host pointers,
 * [B][n] row-major with n = robot->num_factors; Kr: [n] (diagonal gain).  ARMOUR_ESTATE if a nominal torque leaves its
 * interval (the reference throws). */
int armour_robust_controller(const ArmourRobot* robot, double model_uncertainty, const double* Kr, double alpha, double V_max,
                             double r_norm_threshold, int32_t B, const double* q, const double* qd, const double* q_des,
                             const double* qd_des, const double* qdd_des, double* u, double* tau, double* v);
/* Which of the controller's two kernels a call uses (the entry has no handle, so this is per process; bit-identical results
 * either way, tests/test_controller.py): -1 (default) the four-wave latency kernel up to 4096 states and one lane per state
 * beyond, 0 always one lane per state, 1 always the latency kernel. */
int armour_controller_set_kernel(int32_t which);

/* ---- NLP solve of the planning iteration ---- */
/* Replaces IpoptApplication::OptimizeTNLP + armtd_NLP::finalize_solution (RT/armour_main.cu:237-304,
 * RT/NLPclass.cu:422-538) for all B problems of the handle at once: SQP on the device callbacks, start x = 0,
 * exact (constant, diagonal) cost Hessian, dense QP by a dual active-set method, L1-merit line search.
 * IPOPT itself is not part of this library; this solver returns a local optimum of the same NLP. */
typedef struct ArmourSolveOptions {
    int32_t max_iterations;   /* SQP iterations (60) */
    int32_t max_line_search;  /* halvings per iteration (12) */
    double tolerance;         /* step / violation tolerance (1e-4 = IPOPT_OPTIMIZATION_TOLERANCE, RT/Parameters.h:50) */
    double max_wall_time_s;   /* 0 = unlimited (reference: 0.5 s - t(P1) - 0.05 s, RT/armour_main.cu:227-229) */
    double force_host_qp;     /* which of the two forms of the solver runs -- both give the same iterates bit for bit:
                               *   0 (default) automatic: the whole SQP iterate in one persistent kernel (since round 6 for a lone problem too: the QP's
                               *     box-clipped first try -- armour_debug_qp_box -- settles the QPs of feasible problems without an active-set step, 0.07 ms
                               *     per solve on the reference's worlds; the host-driven form -- evaluations on the device, one launch each, the 7-variable
                               *     QPs on the host -- is 7 % faster on lone INFEASIBLE problems, which take hundreds of QP steps, and remains the fallback);
                               *   > 0: the host-QP form whatever the batch;   < 0: the persistent-kernel form whatever the batch. */
    double reserved[3];
} ArmourSolveOptions;
typedef struct ArmourSolveResult {
    double k_opt[ARMOUR_MAX_FACTORS];
    double cost;              /* eval_f at k_opt (includes COST_FUNCTION_OPTIMALITY_SCALE) */
    double max_violation;     /* L1 violation of g_l <= g <= g_u at k_opt */
    int32_t feasible;         /* finalize_solution verdict: what armour.out's "k_opt or -1" is decided on */
    int32_t iterations, evaluations;
    int32_t status;           /* 1 converged, 2 iteration limit, 3 inconsistent linearisation, 4 line search failed, 5 time limit */
    double time_ms;
} ArmourSolveResult;
void armour_solve_options_default(ArmourSolveOptions* opt);
int armour_solve(ArmourPlanner* h, const ArmourSolveOptions* opt, ArmourSolveResult* results /* [B] */);
int armour_batch_solve(ArmourBatch* bt, const ArmourSolveOptions* opt, ArmourSolveResult* results /* [B] */);
/* test hook for the dense QP: min 1/2 x'diag(Gd)x + g0'x  s.t. lo <= A x <= hi (A row-major [m][n], n <= 7,
 * |bound| >= 1e18 = absent).  Host-only: runs without a GPU. */
int armour_debug_qp(int32_t n, const double* Gd, const double* g0, int32_t m, const double* A, const double* lo,
                    const double* hi, double* x, int32_t* feasible);

/* the same with the variables' box x_lo <= x <= x_hi appended as armour_solve appends it, and armour_solve's FIRST TRY (round 6): with a diagonal G the
 * minimiser over the box alone is the unconstrained one clipped variable by variable; when that point satisfies every row it is the QP's solution and
 * no active-set step is taken (steps = 0) -- on the reference's own worlds nearly every QP ends there.  first_try = 0: the active-set method alone.
 * Host-only.  steps / max_mult may be NULL. */
int armour_debug_qp_box(int32_t n, const double* Gd, const double* g0, int32_t m, const double* A, const double* lo, const double* hi,
                        const double* x_lo, const double* x_hi, int32_t first_try, double* x, int32_t* feasible, int32_t* steps, double* max_mult);

/* ---- diagnostics the reference writes to its 4 extra files (RT/armour_main.cu:329-372) ---- */
/* torque_radius [B][n][T]   (armour_control_input_radius.out holds its transpose) */
int armour_get_torque_radius(ArmourPlanner* h, double* torque_radius);
/* link_independent_generators [B][T][J][3][6] row-major (armour_joint_position_radius.out) */
int armour_get_link_generators(ArmourPlanner* h, double* gens);
/* sliced link centres at k: [B][T][J][3] (armour_joint_position_center.out; RT/NLPclass.cu:311-314) */
int armour_get_link_centers(ArmourPlanner* h, const double* k, double* centers);

/* ---- table introspection (parity tests, roofline accounting: SURVEY 8d Sigma M_link / Sigma M_torque) ---- */
/* which: 0 = links(l,t) 3x1, 1 = u_nom(j,t) 1x1.  Returns the monomial count (>= 0) or a negative error.
 * center (may be NULL) receives [2][sz]: the centre, then the independent (interval) radius.
 * keys/coeffs may be NULL; otherwise keys[count], coeffs[count][sz]. */
int armour_get_pz(ArmourPlanner* h, int32_t b, int32_t which, int32_t i, int32_t t, double* center,
                  uint64_t* keys, double* coeffs, int32_t capacity);
/* sum over all problems of monomial counts: out4 = {sum_link, sum_torque, max_link, max_torque} */
int armour_get_table_sizes(ArmourPlanner* h, int64_t* out4);
/* half-space tables in the reference's layout (RT/CollisionChecking.cu:215-227):
 * A [B][T][J][O][36][3], d and delta [B][T][J][O][36]; any pointer may be NULL. */
int armour_get_hyperplanes(ArmourPlanner* h, double* A, double* d, double* delta);
/* plane_skip [B]: bit p set = half-space p is redundant in EVERY collision row of the problem (zero normal, or bit for bit +- the
 * normal of an earlier plane of its row: RT/CollisionChecking.cu:252-259,268-279 can then neither take its value nor its normal), so
 * the table holds nothing for it and the fused evaluation does not read it.  With axis-aligned box obstacles 12 of the 36 planes.
 * For the byte accounting of bench.py (what one evaluation reads) and for tests. */
int armour_get_plane_skip(ArmourPlanner* h, uint64_t* plane_skip);
/* Prune margin of the last armour_set_problems*, per problem: margin[B] = the minimum over EVERY simplify() verdict of the build (RT/PZsparse.cu:305-316:
 * a summed monomial whose Frobenius norm is <= SIMPLIFY_THRESHOLD moves into the independent part, else it stays) of |norm - threshold| / threshold,
 * verdicts on an exactly zero norm left out -- how close the build came to a prune FLIP, where rounding of the last bits decides whether a monomial
 * of norm ~5e-4 is kept as a monomial or folded into the radius: a legitimate change of g by up to that norm, and of the key sets.  Every stated
 * tolerance of this library against the reference's arithmetic (tests/: identical key sets, 1e-11 on tables, 1e-9 / 1e-8 on g / jac) holds while no
 * verdict came closer than ~1e-9; the reference cannot tell (it does not record it), the CPU oracle can (oracle_min_margin), and with
 * this entry so can a production call: both reach-set kernels keep, per lane, the smallest distance of a verdict's squared norm to the squared
 * threshold (two instructions per verdict) and reduce it per problem.  Two conventions differ from the oracle's figure, neither where it matters:
 * which side of the threshold the nearest verdict fell on is not recorded and the kept side's figure is reported -- below the other side's by a
 * relative O(margin) -- and a verdict on an exactly zero norm (a lane of the time-vectorised build without that monomial) counts with the margin
 * sqrt(2) - 1, so 0.414 is the largest value reported while any exists.  1e300 for a problem without a verdict.  Cost: profiles/r06_prune_margin.txt.
 * ARMOUR_ESTATE after armour_debug_load_tables. */
int armour_get_prune_margin(ArmourPlanner* h, double* margin /* [B] */);
/* ms spent in the last armour_set_problems (device time, hipEvent) */
int armour_get_build_ms(ArmourPlanner* h, double* ms);
/* how the last armour_set_problems built its tables: out4 = {kernel that produced them (ARMOUR_P1_KERNEL_*), waves per block of
 * its last launch, sort-buffer entries per wave of that launch, launches of reach-set kernels in all}.  A launch whose sort buffers or
 * slot pools overflow is repeated with the next block shape (and a time-vectorised build that cannot be helped that way is repeated
 * step by step), so `launches` > 1 tells a caller that the limits in ArmourLimits are too small for the fast path; tools and tests
 * read the kernel.  {0,0,0,0} after armour_debug_load_tables. */
#define ARMOUR_P1_KERNEL_PER_STEP 1
#define ARMOUR_P1_KERNEL_TIME_VECTORISED 2
int armour_get_build_info(ArmourPlanner* h, int32_t* out4);
/* name of the P2 kernel as it appears in rocprofv3 kernel traces */
const char* armour_p2_kernel_name(void);

/* ---- test hook: one wave-level polynomial-zonotope operator on caller-supplied operands ---- */
/* Runs the device implementation of one operator of RT/PZsparse.cu (the same functions the reach-set kernel uses):
 *   op 0 mul 3x3*3x1, 1 mul 3x3*3x3, 2 mul 1x1*1x1, 3 mul 1x1*3x1 (:864-994); 4 a+b, 5 a-b on 3x1 (:743-834);
 *   6 addOneDimPZ(a 3x1, b 1x1, r) (:1068-1085); 7 stack(a,b,c) (:1087-1116); 8 cross(a, const), 9 cross(const, a),
 *   10 cross(a, b) (:1118-1167); 11 consts[0]*a + consts[1]*b on 1x1 (:996-1030 + :743-764).
 * Operand o: sz[o] in {1,3,9} entries per coefficient (row-major), cnt[o] monomials with keys[o][cnt] sorted unique and
 * coef[o][cnt][sz]; cen / ind / ind2 are [nops][9].  Result: out_keys[<=out_cap], out_coef[<=out_cap][sz],
 * out_misc[64] = {count, sz, error flags, cen[9], ind[9], ind2[9], [30] the operator's shader-clock cycles, [31] raw terms,
 * [32..] phase counters in -DP1_PROFILE builds, [60] the smallest |s - [61]| over the SQUARED norms s of the operator's simplify() verdicts,
 * [61] the squared threshold they were compared with (+inf in [60]: no verdict; armour_get_prune_margin)}. */
int armour_debug_pz_op(ArmourPlanner* h, int32_t op, int32_t nops, const int32_t* sz, const int32_t* cnt, const uint64_t* const* keys,
                       const double* const* coef, const double* cen, const double* ind, const double* ind2, const double* consts,
                       int32_t r, int32_t out_cap, uint64_t* out_keys, double* out_coef, double* out_misc);

/* ---- test hook: load externally built reach-set tables instead of running P1 ---- */
/* Used only by tests to isolate P2 (tables built by the CPU oracle); never called by the product path.
 * link_* : [B][J][T] counts, centers [..][2][3] (centre, independent radius), keys [..][cap_l], coeffs [..][cap_l][3]
 * torque_*: [B][n][T] counts, centers [..][2], keys [..][cap_t], coeffs [..][cap_t]
 * A,d,delta in the reference layout (see armour_get_hyperplanes); torque_radius [B][n][T]. */
int armour_debug_load_tables(ArmourPlanner* h, int32_t B, int32_t O, const double* q0, const double* qd0,
                             const double* qdd0, const double* q_des, const int32_t* link_count,
                             const double* link_center, const uint64_t* link_keys, const double* link_coeffs,
                             int32_t cap_l, const int32_t* torque_count, const double* torque_center,
                             const uint64_t* torque_keys, const double* torque_coeffs, int32_t cap_t,
                             const double* A, const double* d, const double* delta, const double* torque_radius);

#ifdef __cplusplus
}
#endif
#endif
