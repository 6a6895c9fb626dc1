/*
 * Kinova Gen3 7-DOF (no gripper) robot constants and the default planner parameters.
 * Data source: RT/KinovaWithoutGripperInfo.h:10-112 and RT/Parameters.h:10-48 of roahmlab/armour
 * (physical constants of the arm and its bounding boxes; they are data, not code).
 */
#ifndef ARMOUR_ROBOT_KINOVA_H
#define ARMOUR_ROBOT_KINOVA_H

#include <math.h>
#include <string.h>
#include "armour_types.h"

#ifndef ARMOUR_PI
#define ARMOUR_PI 3.14159265358979323846
#endif

static inline void armour_fill_kinova_gen3_no_gripper(ArmourRobot* r) {
    static const double trans[8 * 3] = {
        0, 0, 0.15643,          0, 0.005375, -0.12838,   0, -0.21038, -0.006375, 0, 0.006375, -0.21038,
        0, -0.20843, -0.006375, 0, 0.00017505, -0.10593, 0, -0.10593, -0.00017505, 0, 0, 0};
    static const double mass[7] = {1.3773, 1.1636, 1.1636, 0.9302, 0.6781, 0.6781, 0.5};
    static const double com[7 * 3] = {
        -0.000023, -0.010364, -0.07336,  -0.000044, -0.09958, -0.013278, -0.000044, -0.006641, -0.117892,
        -0.000018, -0.075478, -0.015006, 0.000001, -0.009432, -0.063883, 0.000001, -0.045483, -0.00965,
        0.000281, 0.011402, -0.029798};
    static const double inertia[7 * 9] = {
        0.00457, 0.000001, 0.000002, 0.000001, 0.004831, 0.000448, 0.000002, 0.000448, 0.001409,
        0.011088, 0.000005, 0, 0.000005, 0.001072, -0.000691, 0, -0.000691, 0.011255,
        0.010932, 0, -0.000007, 0, 0.011127, 0.000606, -0.000007, 0.000606, 0.001043,
        0.008147, -0.000001, 0, -0.000001, 0.000631, -0.0005, 0, -0.0005, 0.008316,
        0.001596, 0, 0, 0, 0.001607, 0.000256, 0, 0.000256, 0.000399,
        0.001641, 0, 0, 0, 0.00041, -0.000278, 0, -0.000278, 0.001641,
        0.000587, 0.000003, 0.000003, 0.000003, 0.000369, -0.000118, 0.000003, -0.000118, 0.000609};
    static const double armature[7] = {8.03, 11.9962024615303644, 9.0025427861751517, 11.5806439316706360,
                                       8.4665040917914123, 8.8537069373742430, 8.8587303664685315};
    static const double lb[7] = {-1000.0, -2.41, -1000.0, -2.66, -1000.0, -2.23, -1000.0};
    static const double ub[7] = {1000.0, 2.41, 1000.0, 2.66, 1000.0, 2.23, 1000.0};
    static const double speed[7] = {1.3963, 1.3963, 1.3963, 1.3963, 1.2218, 1.2218, 1.2218};
    static const double torque[7] = {56.7, 56.7, 56.7, 56.7, 29.4, 29.4, 29.4};
    static const double zc[7 * 3] = {
        0.000000, -0.001297, -0.088375, 0.000000, -0.089400, -0.007877, 0.000000, -0.001502, -0.129375,
        0.000000, -0.087450, -0.013648, 0.000001, -0.009023, -0.071752, 0.000000, -0.041661, -0.009251,
        0.000000, -0.018585, -0.033462};
    static const double zg[7 * 3] = {
        0.046358, 0.047354, 0.086000, 0.046000, 0.135400, 0.047501, 0.046000, 0.047501, 0.127000,
        0.046000, 0.133450, 0.042293, 0.034999, 0.044023, 0.069252, 0.035000, 0.076739, 0.044076,
        0.045500, 0.056085, 0.030963};
    memset(r, 0, sizeof(*r));
    r->num_joints = 7;
    r->num_factors = 7;
    for (int i = 0; i < 7; i++) {
        r->axes[i] = 3;
        r->continuous[i] = (i % 2 == 0);
        r->rots[3 * i] = (i == 0) ? ARMOUR_PI : ((i % 2) ? ARMOUR_PI * 0.5 : -ARMOUR_PI * 0.5);
        r->mass[i] = mass[i];
        r->armature[i] = armature[i];
        r->friction[i] = 0.0; /* disabled in the reference (RT/KinovaWithoutGripperInfo.h:64-69) */
        r->damping[i] = 0.0;
        r->state_limits_lb[i] = lb[i];
        r->state_limits_ub[i] = ub[i];
        r->speed_limits[i] = speed[i];
        r->torque_limits[i] = torque[i];
    }
    memcpy(r->trans, trans, sizeof(trans));
    memcpy(r->com, com, sizeof(com));
    memcpy(r->inertia, inertia, sizeof(inertia));
    memcpy(r->link_zonotope_center, zc, sizeof(zc));
    memcpy(r->link_zonotope_generators, zg, sizeof(zg));
    r->mass_uncertainty = 0.03;
    r->inertia_uncertainty = 0.03;
    r->gravity = 9.81;
    r->alpha = 10.0;
    r->V_m = 1e-2;
    r->M_max = 15.79635774;
    r->M_min = 5.095620491878957;
    r->K = 5.0;
}

/* Kinova Gen3 with the 1.72 kg gripper as a fixed 8th joint: RT/KinovaInfo.h:10-119 (NUM_JOINTS 8, NUM_FACTORS 7).
 * Differs from the no-gripper arm by the extra link (mass, com, inertia, bounding box 0.07 x 0.09 x 0.07) and by the
 * controller constants alpha = 1, M_min = 8.29938, K = 10. */
static inline void armour_fill_kinova_gen3_gripper(ArmourRobot* r) {
    armour_fill_kinova_gen3_no_gripper(r);
    r->num_joints = 8;
    r->axes[7] = 0;
    r->trans[7 * 3 + 0] = 0; r->trans[7 * 3 + 1] = 0; r->trans[7 * 3 + 2] = -0.061525 - 0.10155;
    r->trans[8 * 3 + 0] = 0; r->trans[8 * 3 + 1] = 0; r->trans[8 * 3 + 2] = 0;
    r->rots[7 * 3] = ARMOUR_PI * 0.5;
    r->mass[7] = 1.72;
    r->com[7 * 3 + 0] = 0.00000691; r->com[7 * 3 + 1] = 0.0000044117; r->com[7 * 3 + 2] = 0.031656;
    { const double in7[9] = {0.0004596, 0, 0, 0, 0.0005181, 0, 0, 0, 0.00036051}; memcpy(&r->inertia[7 * 9], in7, sizeof(in7)); }
    r->link_zonotope_center[7 * 3 + 0] = 0.0; r->link_zonotope_center[7 * 3 + 1] = -0.0; r->link_zonotope_center[7 * 3 + 2] = -0.0;
    r->link_zonotope_generators[7 * 3 + 0] = 0.07; r->link_zonotope_generators[7 * 3 + 1] = 0.09; r->link_zonotope_generators[7 * 3 + 2] = 0.07;
    r->alpha = 1.0;
    r->M_min = 8.29938;
    r->K = 10.0;
}

/* RT/Parameters.h defaults except num_time_steps, which the caller chooses
 * (128 in RT/Parameters.h:17; 100 in the BASELINE configs and CMP/Parameters.h:17). */
static inline void armour_fill_default_params(ArmourParams* p, int num_time_steps) {
    memset(p, 0, sizeof(*p));
    p->num_time_steps = num_time_steps;
    p->duration = 1.0;
    for (int i = 0; i < ARMOUR_MAX_FACTORS; i++) p->k_range[i] = ARMOUR_PI / 48;
    p->simplify_threshold = 5e-4;
    p->t_plan = 0.5;
    p->cost_scale = 10.0;
    p->collision_violation_threshold = 1e-4;
    p->torque_violation_threshold = 1e-2;
}

static inline ArmourUltimateBound armour_ultimate_bound(const ArmourRobot* r) {
    ArmourUltimateBound u;
    u.eps = sqrt(2 * r->V_m / r->M_min);
    u.qe = u.eps / r->K;
    u.qde = 2 * u.eps;
    u.qdae = u.eps;
    u.qddae = 2 * r->K * u.eps;
    return u;
}

#endif
