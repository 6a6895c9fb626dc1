/*
 * Fetch mobile manipulator arm: robot constants for BASELINE configs[4] ("Fetch ... payload-mass uncertainty,
 * interval-RNEA torque-bound stress").  Data source: CMP/FetchInfo.h:9-105 of roahmlab/armour
 * (CMP/ = kinova_src/kinova_simulator_interfaces/kinova_planner_realtime_armtd_comparison/): NUM_JOINTS 9 (two fixed
 * links at the end), NUM_FACTORS 7, MIXED joint axes {3,2,1,2,1,2,1,0,0}, no joint-frame rotations, friction / damping /
 * armature all zero, V_m = 1e-7, alpha = 1.
 *
 * What that header does NOT hold, and what stands in for it here (the oracle's generated table makes the same choices,
 * oracle/gen_robot_tables.py):
 *   - link zonotopes: FetchInfo.h:95-101 only has link_radius[7][3] = 0.04 for the actuated links (and the comparison
 *     planner would not compile against it: CMP/Dynamics.cu:34-35 reads link_zonotope_*).  Stand-in: a box of +-radius
 *     centred on the joint frame for links 0..6, a zero box for the two fixed links.
 *   - M_max (needed by the ARMOUR robust-input radius, RT/armour_main.cu:176): taken equal to M_min, so the
 *     alpha*(M_max - M_min)*eps term vanishes.
 * The cost wraps joints 0,2,4,6 as both planners hard-code it (RT/NLPclass.cu:225-231, CMP/NLPclass.cu:199-205).
 */
#ifndef ARMOUR_ROBOT_FETCH_H
#define ARMOUR_ROBOT_FETCH_H

#include <string.h>
#include "armour_types.h"

static inline void armour_fill_fetch(ArmourRobot* r) {
    static const int32_t axes[9] = {3, 2, 1, 2, 1, 2, 1, 0, 0};
    static const double trans[10 * 3] = {
        -0.0326, 0, 0.726,  0.117, 0, 0.06,  0.219, 0, 0,  0.133, 0, 0,  0.197, 0, 0,
        0.1245, 0, 0,  0.1385, 0, 0,  0.16645, 0, 0,  0, 0, 0,  0, 0, 0};
    static const double mass[9] = {2.5587, 2.6615, 2.3311, 2.1299, 1.6563, 1.725, 0.1354, 1.5175, 2.26796};
    static const double com[9 * 3] = {
        0.0927, -0.0056, 0.0564,  0.1432, 0.0072, -0.0001,  0.1165, 0.0014, 0,  0.1279, 0.0073, 0,  0.1097, -0.0266, 0,
        0.0882, 0.0009, -0.0001,  0.0095, 0.0004, -0.0002,  -0.09, -0.0001, -0.0017,  0, 0, 0};
    static const double inertia[9 * 9] = {
        0.0043, -0.0001, 0.001, -0.0001, 0.0087, -0.0001, 0.001, -0.0001, 0.0087,
        0.0028, -0.0021, 0, -0.0021, 0.0111, 0, 0, 0, 0.0112,
        0.0019, -0.0001, 0, -0.0001, 0.0045, 0, 0, 0, 0.0047,
        0.0024, -0.0016, 0, -0.0016, 0.0082, 0, 0, 0, 0.0084,
        0.0016, -0.0003, 0, -0.0003, 0.003, 0, 0, 0, 0.0035,
        0.0018, -0.0001, 0, -0.0001, 0.0042, 0, 0, 0, 0.0042,
        0.0001, 0, 0, 0, 0.0001, 0, 0, 0, 0.0001,
        0.0013, 0, 0, 0, 0.0019, 0, 0, 0, 0.0024,
        0, 0, 0, 0, 0, 0, 0, 0, 0};
    static const double lb[7] = {-1.6056, -1.221, -1000.0, -2.251, -1000.0, -2.16, -1000.0};
    static const double ub[7] = {1.6056, 1.518, 1000.0, 2.251, 1000.0, 2.16, 1000.0};
    static const double speed[7] = {1.256, 1.454, 1.571, 1.521, 1.571, 2.268, 2.268};
    static const double torque[7] = {33.82, 131.76, 76.94, 66.18, 29.35, 25.7, 7.36};
    memset(r, 0, sizeof(*r));
    r->num_joints = 9;
    r->num_factors = 7;
    memcpy(r->axes, axes, sizeof(axes));
    memcpy(r->trans, trans, sizeof(trans));
    memcpy(r->mass, mass, sizeof(mass));
    memcpy(r->com, com, sizeof(com));
    memcpy(r->inertia, inertia, sizeof(inertia));
    for (int i = 0; i < 7; i++) {
        r->continuous[i] = (i % 2 == 0);
        r->state_limits_lb[i] = lb[i];
        r->state_limits_ub[i] = ub[i];
        r->speed_limits[i] = speed[i];
        r->torque_limits[i] = torque[i];
        for (int e = 0; e < 3; e++) r->link_zonotope_generators[3 * i + e] = 0.04; /* link_radius, FetchInfo.h:95-101 */
    }
    r->mass_uncertainty = 0.03;
    r->inertia_uncertainty = 0.03;
    r->gravity = 9.81;
    r->alpha = 1.0;
    r->V_m = 1e-7;
    r->M_min = 5.09562049;
    r->M_max = r->M_min; /* stand-in, see above */
    r->K = 5.0;
}

#if ARMOUR_MAX_FACTORS >= 8
/*
 * "Fetch 8-DOF" of BASELINE configs[4]: the arm above behind ONE MORE actuated joint -- eight trajectory parameters, 72 key bits: only in the
 * 128-bit-key ABI (ARMOUR_MAX_FACTORS = 8; the reference's own key holds seven factors, RT/PZsparse.h:8-21, and cannot represent this robot).
 * The Fetch's eighth degree of freedom is its torso LIFT, a prismatic joint; the reference's formulation has rotations only (RT/Dynamics.cu:69-81:
 * every joint is a rotation about one of its frame's axes), so the eighth joint here is a torso YAW -- a revolute joint about the base z axis
 * under the shoulder -- which loads the pipeline exactly as an eighth revolute joint does: one more factor in every monomial key, one more
 * rotation in the forward kinematics of every link, one more step in both RNEA recursions.  NOT reference data (there is none): the torso link's
 * mass / centre of mass / inertia are the Fetch URDF's torso_lift_link as recalled (10.78 kg), its limits and link box are stated stand-ins.
 * Links: 0 torso (axis z) | 1..7 the arm's seven actuated links of armour_fill_fetch, frames unchanged relative to the base at torso angle 0 |
 * 8 the fixed gripper link (FetchInfo.h's eighth link; its ninth -- a fixed 2.27 kg point mass in the same frame -- is dropped: 9 links is
 * ARMOUR_MAX_JOINTS), the link whose mass / inertia uncertainty models the payload: mass_uncertainty_link[8].
 */
static inline void armour_fill_fetch8(ArmourRobot* r) {
    ArmourRobot f;
    armour_fill_fetch(&f);
    memset(r, 0, sizeof(*r));
    r->num_joints = 9;
    r->num_factors = 8;
    /* the torso */
    r->axes[0] = 3;
    r->mass[0] = 10.7796;
    r->com[0] = -0.0013; r->com[1] = -0.0009; r->com[2] = 0.2935;
    r->inertia[0] = 0.3354; r->inertia[2] = -0.0162; r->inertia[4] = 0.3354; r->inertia[5] = -0.0006; r->inertia[6] = -0.0162; r->inertia[7] = -0.0006; r->inertia[8] = 0.0954;
    r->continuous[0] = 0;
    r->state_limits_lb[0] = -1.0; r->state_limits_ub[0] = 1.0;
    r->speed_limits[0] = 0.5;
    r->torque_limits[0] = 150.0;
    r->link_zonotope_center[2] = 0.363;                                                              /* a column between the base and the shoulder */
    r->link_zonotope_generators[0] = 0.12; r->link_zonotope_generators[1] = 0.12; r->link_zonotope_generators[2] = 0.363;
    /* links 1..8 = the arm's links 0..7; frame i + 1 of this chain = frame i of the arm (the torso frame coincides with the base) */
    for (int i = 0; i < 8; i++) {
        const int j = i + 1;
        r->axes[j] = f.axes[i];
        r->mass[j] = f.mass[i];
        for (int e = 0; e < 3; e++) { r->trans[3 * j + e] = f.trans[3 * i + e]; r->com[3 * j + e] = f.com[3 * i + e]; r->rots[3 * j + e] = f.rots[3 * i + e]; }
        for (int e = 0; e < 9; e++) r->inertia[9 * j + e] = f.inertia[9 * i + e];
        for (int e = 0; e < 3; e++) { r->link_zonotope_center[3 * j + e] = f.link_zonotope_center[3 * i + e]; r->link_zonotope_generators[3 * j + e] = f.link_zonotope_generators[3 * i + e]; }
        r->friction[j] = f.friction[i]; r->damping[j] = f.damping[i]; r->armature[j] = f.armature[i];
    }
    for (int e = 0; e < 3; e++) r->trans[3 * 9 + e] = f.trans[3 * 8 + e];
    for (int i = 0; i < 7; i++) {
        r->continuous[i + 1] = f.continuous[i];
        r->state_limits_lb[i + 1] = f.state_limits_lb[i]; r->state_limits_ub[i + 1] = f.state_limits_ub[i];
        r->speed_limits[i + 1] = f.speed_limits[i]; r->torque_limits[i + 1] = f.torque_limits[i];
    }
    r->mass_uncertainty = f.mass_uncertainty; r->inertia_uncertainty = f.inertia_uncertainty;
    r->gravity = f.gravity; r->alpha = f.alpha; r->V_m = f.V_m; r->M_min = f.M_min; r->M_max = f.M_max; r->K = f.K;
}
#endif

#endif
