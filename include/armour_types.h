/*
 * armour_types.h -- plain-C data types shared by the MI355X library (libarmour_hip.so),
 * the MEX gateway, the file-protocol CLI and (read-only) the CPU oracle.
 *
 * Everything the reference bakes in as compile-time macros / header constants is a
 * runtime struct here:
 *   ArmourRobot  <- RT/KinovaWithoutGripperInfo.h:10-112 (NUM_JOINTS, NUM_FACTORS, axes, trans,
 *                   rots, mass, com, inertia, friction, damping, armature, limits, link
 *                   zonotopes, ultimate-bound constants)
 *   ArmourParams <- RT/Parameters.h:4-58 (SIMPLIFY_THRESHOLD, DURATION, NUM_TIME_STEPS, k_range,
 *                   COST_FUNCTION_OPTIMALITY_SCALE, violation thresholds) and
 *                   RT/armour_main.cu:78 (t_plan)
 * (RT/ = kinova_src/kinova_simulator_interfaces/kinova_planner_realtime/ of roahmlab/armour.)
 */
#ifndef ARMOUR_TYPES_H
#define ARMOUR_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARMOUR_MAX_JOINTS 9   /* links incl. fixed joints at the end of the chain (RT/KinovaInfo.h: 8; CMP/FetchInfo.h: 9) */
/* 9 bits of monomial key per factor: 7 factors fill the reference's u64 key (RT/PZsparse.h:8-21), and that is what the shipped
 * library is built for.  A build with -DARMOUR_KEY128 (make -C armour_amd/csrc k128: libarmour_hip_k128.so; make -C oracle k128) carries
 * 128-bit monomial keys -- the same field order, LSB first, 9 bits per factor -- and holds 8 factors (BASELINE configs[4]'s "8-DOF");
 * every array below grows with ARMOUR_MAX_FACTORS, so the two builds are two ABIs: armour_abi_max_factors() tells a loader which. */
#if defined(ARMOUR_KEY128)
#define ARMOUR_MAX_FACTORS 8
#else
#define ARMOUR_MAX_FACTORS 7
#endif
#define ARMOUR_OBS_DOUBLES 12 /* one obstacle = column-major Z=[c g1 g2 g3] (KSI/uarmtd_planner.m:178) */
#define ARMOUR_NUM_PLANES 36  /* C(9,2) generator pairs of a buffered obstacle (RT/CollisionChecking.h:6-7) */

typedef struct ArmourRobot {
    int32_t num_joints;   /* NUM_JOINTS  */
    int32_t num_factors;  /* NUM_FACTORS (actuated joints; fixed joints, if any, are last) */
    int32_t axes[ARMOUR_MAX_JOINTS];              /* 1,2,3 = x,y,z ; 0 = fixed */
    int32_t continuous[ARMOUR_MAX_FACTORS];       /* 1 = wrap-to-pi in the cost (RT/NLPclass.cu:225-231) */
    double trans[(ARMOUR_MAX_JOINTS + 1) * 3];
    double rots[ARMOUR_MAX_JOINTS * 3];           /* rpy of each joint frame */
    double mass[ARMOUR_MAX_JOINTS];
    double mass_uncertainty;
    double com[ARMOUR_MAX_JOINTS * 3];
    double inertia[ARMOUR_MAX_JOINTS * 9];
    double inertia_uncertainty;
    double friction[ARMOUR_MAX_JOINTS];
    double damping[ARMOUR_MAX_JOINTS];
    double armature[ARMOUR_MAX_JOINTS];
    double state_limits_lb[ARMOUR_MAX_FACTORS];
    double state_limits_ub[ARMOUR_MAX_FACTORS];
    double speed_limits[ARMOUR_MAX_FACTORS];
    double torque_limits[ARMOUR_MAX_FACTORS];
    double gravity;
    double link_zonotope_center[ARMOUR_MAX_JOINTS * 3];
    double link_zonotope_generators[ARMOUR_MAX_JOINTS * 3];
    /* ultimate bound (RT/KinovaWithoutGripperInfo.h:103-112); eps, qe, qde, qdae, qddae are derived */
    double alpha, V_m, M_max, M_min, K;
    /* per-link overrides of mass_uncertainty / inertia_uncertainty: entry i != 0 replaces the scalar for link i (the reference has one
     * scalar for every link, RT/Dynamics.cu:33,40; a payload of uncertain mass on the last link is e.g. mass_uncertainty_link[J-1] = 0.5,
     * BASELINE configs[4]).  Zero-initialised structs behave as before. */
    double mass_uncertainty_link[ARMOUR_MAX_JOINTS];
    double inertia_uncertainty_link[ARMOUR_MAX_JOINTS];
} ArmourRobot;

/* relative uncertainty of link i's mass / inertia (the per-link entry if set, else the robot-wide scalar); usable from
 * plain C / C++ hosts and, when this header is compiled by hipcc, from device code */
#if defined(__HIPCC__)
#define ARMOUR_HD __host__ __device__
#else
#define ARMOUR_HD
#endif
ARMOUR_HD static inline double armour_mass_uncertainty(const ArmourRobot* r, int i) { return r->mass_uncertainty_link[i] != 0.0 ? r->mass_uncertainty_link[i] : r->mass_uncertainty; }
ARMOUR_HD static inline double armour_inertia_uncertainty(const ArmourRobot* r, int i) { return r->inertia_uncertainty_link[i] != 0.0 ? r->inertia_uncertainty_link[i] : r->inertia_uncertainty; }

typedef struct ArmourParams {
    int32_t num_time_steps;                 /* NUM_TIME_STEPS, must be even (RT/Parameters.h:16) */
    int32_t input_constraints_off;          /* TURN_OFF_INPUT_CONSTRAINTS (RT/Parameters.h:46-47), 0 = false (the default): non-zero plans the same Bezier
                                             * trajectory without the torque rows -- the reach-set build stops after the forward kinematics
                                             * (RT/armour_main.cu:115,149-165,175), m = J T O + 4n, collision rows first (RT/NLPclass.cu:46-54,117,289-301,
                                             * 361-373,453), the torque-radius file is not written (RT/armour_main.cu:355).  (This field was `reserved`.) */
    double duration;                        /* DURATION */
    double k_range[ARMOUR_MAX_FACTORS];     /* radians */
    double simplify_threshold;              /* SIMPLIFY_THRESHOLD */
    double t_plan;                          /* RT/armour_main.cu:78 */
    double cost_scale;                      /* COST_FUNCTION_OPTIMALITY_SCALE */
    double collision_violation_threshold;   /* 1e-4 m   (RT/Parameters.h:38) */
    double torque_violation_threshold;      /* 1e-2 N*m (RT/Parameters.h:41) */
} ArmourParams;

/* derived ultimate-bound quantities, same formulas as RT/KinovaWithoutGripperInfo.h:108-112 */
typedef struct ArmourUltimateBound {
    double eps, qe, qde, qdae, qddae;
} ArmourUltimateBound;

#ifdef __cplusplus
}
#endif
#endif
