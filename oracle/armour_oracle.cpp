/*
 * oracle/armour_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement ("port") of ARMOUR's per-planning-iteration pipeline, following the reference
 * roahmlab/armour under RT/ = kinova_src/kinova_simulator_interfaces/kinova_planner_realtime/ :
 *   JRS                 RT/Trajectory.cu:15-254
 *   joint-limit rows    RT/Trajectory.cu:256-540 (+ helpers :542-822, restated by the chain rule)
 *   FK / RNEA           RT/Dynamics.cu:6-181
 *   P1 ordering, disturbance, torque radius   RT/armour_main.cu:96-205
 *   half-space tables   RT/CollisionChecking.cu:136-228
 *   collision rows      RT/CollisionChecking.cu:230-299
 *   NLP callback        RT/NLPclass.cu:87-165 (bounds), :207-267 (cost), :272-396 (eval_g / eval_jac_g)
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * PARITY UNPINNED (see pz.hpp): no reference golden vectors exist and the reference cannot be
 * built in this image; pinned by invariants only.
 */
#include <omp.h>

#include <chrono>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../include/armour_types.h"
#include "interval.hpp"
#include "pz.hpp"
#include "robot_tables.hpp" /* the oracle's own constants, generated from the reference headers (gen_robot_tables.py) */

namespace oracle {

/* ------------------------------------------------------------------ Bezier scalar helpers */
/* RT/Trajectory.cu:542-558 */
static double q_des_func(double q0, double Tqd0, double TTqdd0, double k, double t) {
    double B0 = -pow(t - 1, 5);
    double B1 = 5 * t * pow(t - 1, 4);
    double B2 = -10 * pow(t, 2) * pow(t - 1, 3);
    double B3 = 10 * pow(t, 3) * pow(t - 1, 2);
    double B4 = -5 * pow(t, 4) * (t - 1);
    double B5 = pow(t, 5);
    double beta0 = q0, beta1 = q0 + Tqd0 / 5, beta2 = q0 + (2 * Tqd0) / 5 + TTqdd0 / 20;
    double beta3 = q0 + k, beta4 = q0 + k, beta5 = q0 + k;
    return B0 * beta0 + B1 * beta1 + B2 * beta2 + B3 * beta3 + B4 * beta4 + B5 * beta5;
}
/* RT/Trajectory.cu:560-574 */
static double qd_des_func(double q0, double Tqd0, double TTqdd0, double k, double t) {
    double dB0 = pow(t - 1.0, 4.0) * -5.0;
    double dB1 = t * pow(t - 1.0, 3.0) * 20.0 + pow(t - 1.0, 4.0) * 5.0;
    double dB2 = t * pow(t - 1.0, 3.0) * -20.0 - (t * t) * pow(t - 1.0, 2.0) * 30.0;
    double dB3 = pow(t, 3.0) * (t * 2.0 - 2.0) * 10.0 + (t * t) * pow(t - 1.0, 2.0) * 30.0;
    double dB4 = pow(t, 3.0) * (t - 1.0) * -20.0 - pow(t, 4.0) * 5.0;
    double dB5 = pow(t, 4.0) * 5.0;
    double beta0 = q0, beta1 = q0 + Tqd0 / 5, beta2 = q0 + (2 * Tqd0) / 5 + TTqdd0 / 20;
    double beta3 = q0 + k, beta4 = q0 + k, beta5 = q0 + k;
    return dB0 * beta0 + dB1 * beta1 + dB2 * beta2 + dB3 * beta3 + dB4 * beta4 + dB5 * beta5;
}
/* RT/Trajectory.cu:576-602 (second derivative of the same Bernstein form) */
static double qdd_des_func(double q0, double Tqd0, double TTqdd0, double k, double t) {
    const double u = t - 1.0;
    double ddB0 = -20.0 * u * u * u;
    double ddB1 = 40.0 * u * u * u + 60.0 * t * u * u;
    double ddB2 = -20.0 * u * u * u - 120.0 * t * u * u - 30.0 * t * t * (2.0 * t - 2.0);
    double ddB3 = 20.0 * t * t * t + 60.0 * t * u * u + 60.0 * t * t * (2.0 * t - 2.0);
    double ddB4 = -40.0 * t * t * t - 60.0 * t * t * u;
    double ddB5 = 20.0 * t * t * t;
    double beta0 = q0, beta1 = q0 + Tqd0 / 5, beta2 = q0 + (2 * Tqd0) / 5 + TTqdd0 / 20;
    double beta3 = q0 + k, beta4 = q0 + k, beta5 = q0 + k;
    return ddB0 * beta0 + ddB1 * beta1 + ddB2 * beta2 + ddB3 * beta3 + ddB4 * beta4 + ddB5 * beta5;
}
/* RT/Trajectory.cu:812-822 */
static double q_des_k_indep(double q0, double Tqd0, double TTqdd0, double s) {
    return q0 + Tqd0 * s - 6 * Tqd0 * pow(s, 3) + 8 * Tqd0 * pow(s, 4) - 3 * Tqd0 * pow(s, 5) + (TTqdd0 * pow(s, 2)) * 0.5 -
           (3 * TTqdd0 * pow(s, 3)) * 0.5 + (3 * TTqdd0 * pow(s, 4)) * 0.5 - (TTqdd0 * pow(s, 5)) * 0.5;
}
static double qd_des_k_indep(double, double Tqd0, double TTqdd0, double s, double DUR) {
    return (pow(s - 1, 2) * (2 * Tqd0 + 4 * Tqd0 * s + 2 * TTqdd0 * s - 30 * Tqd0 * pow(s, 2) - 5 * TTqdd0 * pow(s, 2))) * 0.5 / DUR;
}
static double qdd_des_k_indep(double, double Tqd0, double TTqdd0, double s, double DUR) {
    return -(s - 1.0) * (TTqdd0 - (36 * Tqd0 + 8 * TTqdd0) * s + (60 * Tqd0 + 10 * TTqdd0) * pow(s, 2)) / (DUR * DUR);
}

/* d/dk of q_des(t*(k); k) at the interior stationary points t* = extrema2/3.
 * The reference (RT/Trajectory.cu:604-700) holds the MATLAB-symbolic expansion of exactly this
 * total derivative; it is restated here by the chain rule
 *     d/dk q(t*(k),k) = dq/dk|_t* + qdot(t*) * dt_star/dk,
 * which is the same function of (q0,Tqd0,TTqdd0,k) up to rounding. sign=+1: extrema2, -1: extrema3. */
static double q_des_extrema_k_derivative(double q0, double a, double b, double k, int sign) {
    const double D = 64 * a * a + 14 * a * b - 120 * k * a + b * b;
    const double sq = sqrt(D);
    const double den = 6 * a - 12 * k + b;
    const double num = 2 * a + b + sign * sq;
    const double ts = num / (5 * den);
    const double dnum = sign * (-120 * a) / (2 * sq);
    const double dts = (dnum * 5 * den - num * 5 * (-12)) / (25 * den * den);
    const double dqdk = ts * ts * ts * (6 * ts * ts - 15 * ts + 10);
    return dqdk + qd_des_func(q0, a, b, k, ts) * dts;
}
/* same for qd_des at its stationary points (RT/Trajectory.cu:702-810) */
static double qd_des_extrema_k_derivative(double q0, double a, double b, double k, int sign) {
    const double E = 150 * k * k - 180 * k * a - 20 * k * b + 54 * a * a + 14 * a * b + b * b;
    const double sq = sqrt(6 * E);
    const double den = 6 * a - 12 * k + b;
    const double num = 18 * a - 30 * k + 4 * b + sign * sq;
    const double ts = num / (10 * den);
    const double dE = 300 * k - 180 * a - 20 * b;
    const double dnum = -30 + sign * (6 * dE) / (2 * sq);
    const double dts = (dnum * 10 * den - num * 10 * (-12)) / (100 * den * den);
    const double dqddk = 30 * ts * ts * (ts - 1) * (ts - 1);
    return dqddk + qdd_des_func(q0, a, b, k, ts) * dts;
}

/* ultimate bound of the tracking error, RT/KinovaWithoutGripperInfo.h:103-112 */
static ArmourUltimateBound ultimate_bound(const ArmourRobot& r) {
    ArmourUltimateBound u;
    u.eps = sqrt(2 * r.V_m / r.M_min);
    u.qe = u.eps / r.K;
    u.qde = 2 * u.eps;
    u.qdae = u.eps;
    u.qddae = 2 * r.K * u.eps;
    return u;
}

/* ------------------------------------------------------------------ problem state */
struct Problem {
    ArmourRobot rb;
    ArmourParams pr;
    ArmourUltimateBound ub;
    int T, J, n, O = 0;
    double q0[ARMOUR_MAX_FACTORS], qd0[ARMOUR_MAX_FACTORS], qdd0[ARMOUR_MAX_FACTORS], q_des[ARMOUR_MAX_FACTORS], Tqd0[ARMOUR_MAX_FACTORS], TTqdd0[ARMOUR_MAX_FACTORS];
    std::vector<double> obstacles; /* O*12 */
    /* Bezier k-independent extrema (RT/Trajectory.cu:36-58) */
    double qx1[ARMOUR_MAX_FACTORS], qx2[ARMOUR_MAX_FACTORS], qm1[ARMOUR_MAX_FACTORS], qm2[ARMOUR_MAX_FACTORS], vx1[ARMOUR_MAX_FACTORS], vx2[ARMOUR_MAX_FACTORS], vm1[ARMOUR_MAX_FACTORS], vm2[ARMOUR_MAX_FACTORS], ax1[ARMOUR_MAX_FACTORS], ax2[ARMOUR_MAX_FACTORS], am1[ARMOUR_MAX_FACTORS], am2[ARMOUR_MAX_FACTORS];
    /* JRS PZs, index [i*T + t] */
    std::vector<PZ> R, R_t, qd_des, qda_des, qdda_des;
    std::vector<PZ> links, u_nom, u_nom_int;
    std::vector<double> link_gens;     /* [t*J + l][3][6] row-major */
    std::vector<double> torque_radius; /* [j*T + t]  (reference: Eigen(j,t)) */
    std::vector<double> A, d, delta;   /* [((t*J+l)*O+o)*36+p] (x3 for A) */
    Stats st;
    double build_ms = 0;
    /* ARMTD comparison mode (CMP/ = kinova_planner_realtime_armtd_comparison): constant-acceleration curve, offline JRS tables */
    bool armtd = false;
    /* TURN_OFF_INPUT_CONSTRAINTS (RT/Parameters.h:46-47): the ARMOUR trajectory without the torque rows -- P1 runs JRS + forward kinematics only
     * (RT/armour_main.cu:115,149-165,175), m = J T O + 4n with the collision rows first (RT/NLPclass.cu:46-54,117,289-301,361-373,453) */
    bool no_torque() const { return armtd || pr.input_constraints_off != 0; }
    std::vector<double> jrs; /* [n][6][T]: c,g,r of cos then of sin (CMP/armtd_main.cu:76-96) */
    double k_range_a[ARMOUR_MAX_FACTORS];     /* per-joint k_range read with the tables (:97) */
};

/* RT/Trajectory.cu:15-61 */
static void bezier_init(Problem& P) {
    const double DUR = P.pr.duration;
    for (int i = 0; i < P.n; i++) {
        P.Tqd0[i] = P.qd0[i] * DUR;
        P.TTqdd0[i] = P.qdd0[i] * DUR * DUR;
        const double a = P.Tqd0[i], b = P.TTqdd0[i];
        P.qx1[i] = (2 * a + b + sqrt(64 * pow(a, 2) + 14 * a * b + pow(b, 2))) / (5 * (6 * a + b));
        P.qx2[i] = (2 * a + b - sqrt(64 * pow(a, 2) + 14 * a * b + pow(b, 2))) / (5 * (6 * a + b));
        P.qm1[i] = q_des_k_indep(P.q0[i], a, b, P.qx1[i]);
        P.qm2[i] = q_des_k_indep(P.q0[i], a, b, P.qx2[i]);
        P.vx1[i] = (18 * a + 4 * b + sqrt(6 * (54 * pow(a, 2) + 14 * a * b + pow(b, 2)))) / (10 * (6 * a + b));
        P.vx2[i] = (18 * a + 4 * b - sqrt(6 * (54 * pow(a, 2) + 14 * a * b + pow(b, 2)))) / (10 * (6 * a + b));
        P.vm1[i] = qd_des_k_indep(P.q0[i], a, b, P.vx1[i], DUR);
        P.vm2[i] = qd_des_k_indep(P.q0[i], a, b, P.vx2[i], DUR);
        P.ax1[i] = (32 * a + 6 * b + sqrt(2 * (152 * pow(a, 2) + 42 * a * b + 3 * pow(b, 2)))) / (10 * (6 * a + b));
        P.ax2[i] = (32 * a + 6 * b - sqrt(2 * (152 * pow(a, 2) + 42 * a * b + 3 * pow(b, 2)))) / (10 * (6 * a + b));
        P.am1[i] = qdd_des_k_indep(P.q0[i], a, b, P.ax1[i], DUR);
        P.am2[i] = qdd_des_k_indep(P.q0[i], a, b, P.ax2[i], DUR);
    }
}

static inline void bound_with_extrema(double& lb, double& ub_, double s_lb, double s_ub, double x1, double m1, double x2, double m2) {
    if (lb > ub_) std::swap(lb, ub_);
    if (s_lb < x1 && x1 < s_ub) { lb = std::min(lb, m1); ub_ = std::max(ub_, m1); }
    if (s_lb < x2 && x2 < s_ub) { lb = std::min(lb, m2); ub_ = std::max(ub_, m2); }
}

/* RT/Trajectory.cu:63-254 */
static void make_poly_zono(Problem& P, Ctx& cx, int s_ind) {
    const int T = P.T, n = P.n, J = P.J;
    const double ds = 1.0 / T, DUR = P.pr.duration;
    const double s_lb = s_ind * ds, s_ub = (s_ind + 1) * ds;
    const KeyLayout& kl = cx.kl;
    const double MAXIMA = 0.5 - sqrt(3) / 6, MINIMA = 0.5 + sqrt(3) / 6; /* RT/Trajectory.h:7-8 */
    for (int i = 0; i < n; i++) {
        const double kr = P.pr.k_range[i];
        const double q0 = P.q0[i], a = P.Tqd0[i], b = P.TTqdd0[i];
        /* Part 1: q_des */
        double kd_lb = pow(s_lb, 3) * (6 * pow(s_lb, 2) - 15 * s_lb + 10);
        double kd_ub = pow(s_ub, 3) * (6 * pow(s_ub, 2) - 15 * s_ub + 10);
        double kd_c = (kd_ub + kd_lb) * 0.5;
        double kd_r = (kd_ub - kd_lb) * 0.5 * kr;
        double ki_lb = q_des_k_indep(q0, a, b, s_lb), ki_ub = q_des_k_indep(q0, a, b, s_ub);
        bound_with_extrema(ki_lb, ki_ub, s_lb, s_ub, P.qx1[i], P.qm1[i], P.qx2[i], P.qm2[i]);
        double ki_r = (ki_ub - ki_lb) * 0.5;
        const double q_c = (ki_lb + ki_ub) * 0.5;
        const Interval q_rad(-kd_r - ki_r - P.ub.qe, kd_r + ki_r + P.ub.qe);
        const Interval kint = kd_c * Interval(-kr, kr);

        /* 1.a cos(q_des): first-order Taylor + Lagrange remainder (:103-110) */
        double cos_c = std::cos(q_c);
        Interval cos_rad = (-q_rad) * std::sin(q_c) - (0.5 * icos(q_c + kint + q_rad)) * sqr(q_rad + kint);
        cos_c += getCenter(cos_rad);
        cos_rad = cos_rad - getCenter(cos_rad);
        const double cos_coeff[2] = {-kd_c * kr * std::sin(q_c), getRadius(cos_rad)};
        const okey_t cos_keys[2] = {(okey_t)1 << kl.shift_k(i), (okey_t)1 << kl.shift_cosqe(i)};
        /* 1.b sin(q_des) (:120-127) */
        double sin_c = std::sin(q_c);
        Interval sin_rad = q_rad * std::cos(q_c) - (0.5 * isin(q_c + kint + q_rad)) * sqr(q_rad + kint);
        sin_c += getCenter(sin_rad);
        sin_rad = sin_rad - getCenter(sin_rad);
        const double sin_coeff[2] = {kd_c * kr * std::cos(q_c), getRadius(sin_rad)};
        const okey_t sin_keys[2] = {(okey_t)1 << kl.shift_k(i), (okey_t)1 << kl.shift_sinqe(i)};

        PZ Ri = pz_rpy(P.rb.rots[3 * i], P.rb.rots[3 * i + 1], P.rb.rots[3 * i + 2]);
        if (P.rb.axes[i] != 0)
            Ri = mul(cx, Ri, pz_rotation(cx, cos_c, cos_coeff, cos_keys, 2, sin_c, sin_coeff, sin_keys, 2, P.rb.axes[i]));
        P.R_t[i * T + s_ind] = transpose(Ri);
        P.R[i * T + s_ind] = std::move(Ri);

        /* Part 2: qd_des (:140-187) */
        kd_lb = (30 * pow(s_lb, 2) * pow(s_lb - 1, 2)) / DUR;
        kd_ub = (30 * pow(s_ub, 2) * pow(s_ub - 1, 2)) / DUR;
        if (kd_ub < kd_lb) std::swap(kd_lb, kd_ub);
        kd_c = (kd_ub + kd_lb) * 0.5 * kr;
        kd_r = (kd_ub - kd_lb) * 0.5 * kr;
        ki_lb = qd_des_k_indep(q0, a, b, s_lb, DUR);
        ki_ub = qd_des_k_indep(q0, a, b, s_ub, DUR);
        bound_with_extrema(ki_lb, ki_ub, s_lb, s_ub, P.vx1[i], P.vm1[i], P.vx2[i], P.vm2[i]);
        ki_r = (ki_ub - ki_lb) * 0.5;
        const double qd_c = (ki_lb + ki_ub) * 0.5;
        {
            const double co[2] = {kd_c, kd_r + ki_r + P.ub.qde};
            const okey_t ke[2] = {(okey_t)1 << kl.shift_k(i), (okey_t)1 << kl.shift_qde(i)};
            P.qd_des[i * T + s_ind] = pz_scalar_poly(cx, qd_c, co, ke, 2);
            const double co2[2] = {kd_c, kd_r + ki_r + P.ub.qdae};
            const okey_t ke2[2] = {(okey_t)1 << kl.shift_k(i), (okey_t)1 << kl.shift_qdae(i)};
            P.qda_des[i * T + s_ind] = pz_scalar_poly(cx, qd_c, co2, ke2, 2);
        }
        /* Part 3: qdd_des (:189-243) */
        auto acc = [&](double s) { return (60 * s * (2 * pow(s, 2) - 3 * s + 1)) / DUR / DUR; };
        const double t_lb = acc(s_lb), t_ub = acc(s_ub);
        if (s_ub <= MAXIMA) { kd_lb = t_lb; kd_ub = t_ub; }
        else if (s_lb <= MAXIMA) { kd_lb = std::min(t_lb, t_ub); kd_ub = acc(MAXIMA); }
        else if (s_ub <= MINIMA) { kd_lb = t_ub; kd_ub = t_lb; }
        else if (s_lb <= MINIMA) { kd_lb = acc(MINIMA); kd_ub = std::max(t_lb, t_ub); }
        else { kd_lb = t_lb; kd_ub = t_ub; }
        kd_c = (kd_ub + kd_lb) * 0.5 * kr;
        kd_r = (kd_ub - kd_lb) * 0.5 * kr;
        ki_lb = qdd_des_k_indep(q0, a, b, s_lb, DUR);
        ki_ub = qdd_des_k_indep(q0, a, b, s_ub, DUR);
        bound_with_extrema(ki_lb, ki_ub, s_lb, s_ub, P.ax1[i], P.am1[i], P.ax2[i], P.am2[i]);
        ki_r = (ki_ub - ki_lb) * 0.5;
        const double qdd_c = (ki_lb + ki_ub) * 0.5;
        {
            const double co[2] = {kd_c, kd_r + ki_r + P.ub.qddae};
            const okey_t ke[2] = {(okey_t)1 << kl.shift_k(i), (okey_t)1 << kl.shift_qddae(i)};
            P.qdda_des[i * T + s_ind] = pz_scalar_poly(cx, qdd_c, co, ke, 2);
        }
    }
    for (int i = n; i < J; i++) { /* fixed joints at the end of the chain (:246-250) */
        P.R[i * T + s_ind] = pz_rpy(P.rb.rots[3 * i], P.rb.rots[3 * i + 1], P.rb.rots[3 * i + 2]);
        P.R_t[i * T + s_ind] = transpose(P.R[i * T + s_ind]);
    }
    P.R[J * T + s_ind] = pz_rpy(0, 0, 0);
}

/* CMP/Trajectory.cu:29-81 (ConstantAccelerationCurve::makePolyZono) */
static void make_poly_zono_armtd(Problem& P, Ctx& cx, int t_ind) {
    const int T = P.T, n = P.n, J = P.J;
    const KeyLayout& kl = cx.kl;
    for (int i = 0; i < n; i++) {
        const double* tab = &P.jrs[(size_t)i * 6 * T];
        const double *c_cos = tab, *g_cos = tab + T, *r_cos = tab + 2 * T, *c_sin = tab + 3 * T, *g_sin = tab + 4 * T, *r_sin = tab + 5 * T;
        const double cos_q0 = std::cos(P.q0[i]), sin_q0 = std::sin(P.q0[i]);
        const double cos_c = cos_q0 * c_cos[t_ind] - sin_q0 * c_sin[t_ind];
        double cos_coeff[2];
        cos_coeff[0] = cos_q0 * g_cos[t_ind] - sin_q0 * g_sin[t_ind];
        cos_coeff[1] = std::fabs(cos_q0) * r_cos[t_ind] + std::fabs(sin_q0) * r_sin[t_ind];
        cos_coeff[1] *= 4.0;
        const okey_t cos_keys[2] = {(okey_t)1 << kl.shift_k(i), (okey_t)1 << kl.shift_cosqe(i)};
        const double sin_c = cos_q0 * c_sin[t_ind] + sin_q0 * c_cos[t_ind];
        double sin_coeff[2];
        sin_coeff[0] = cos_q0 * g_sin[t_ind] + sin_q0 * g_cos[t_ind];
        sin_coeff[1] = std::fabs(cos_q0) * r_sin[t_ind] + std::fabs(sin_q0) * r_cos[t_ind];
        sin_coeff[1] *= 4.0;
        const okey_t sin_keys[2] = {(okey_t)1 << kl.shift_k(i), (okey_t)1 << kl.shift_sinqe(i)};
        PZ Ri = pz_rpy(P.rb.rots[3 * i], P.rb.rots[3 * i + 1], P.rb.rots[3 * i + 2]);
        if (P.rb.axes[i] != 0)
            Ri = mul(cx, Ri, pz_rotation(cx, cos_c, cos_coeff, cos_keys, 2, sin_c, sin_coeff, sin_keys, 2, P.rb.axes[i]));
        P.R_t[i * T + t_ind] = transpose(Ri);
        P.R[i * T + t_ind] = std::move(Ri);
    }
    for (int i = n; i < J; i++) { /* fixed joints at the end of the chain (:73-78) */
        P.R[i * T + t_ind] = pz_rpy(P.rb.rots[3 * i], P.rb.rots[3 * i + 1], P.rb.rots[3 * i + 2]);
        P.R_t[i * T + t_ind] = transpose(P.R[i * T + t_ind]);
    }
    P.R[J * T + t_ind] = pz_rpy(0, 0, 0);
}

/* CMP/Trajectory.cu:83-205 (values) and :207-383 (gradient): extremum[0..n) q_min, [n..2n) q_max, [2n..3n) qd_min,
 * [3n..4n) qd_max; the gradient is what the reference writes on the Jacobian diagonal (d/d(k_range k), no k_range factor). */
static void armtd_state_extremum(const Problem& P, const double* k, double* extremum, double* grad_diag) {
    const int n = P.n;
    const double t_move = 0.5, t_total = 1.0, t_to_stop = t_total - t_move;
    for (int i = 0; i < n; i++) {
        const double q0 = P.q0[i], qd0 = P.qd0[i];
        const double k_actual = P.k_range_a[i] * k[i];
        const double q_peak = q0 + qd0 * t_move + k_actual * t_move * t_move * 0.5;
        const double q_dot_peak = qd0 + k_actual * t_move;
        const double q_ddot_to_stop = -q_dot_peak / t_to_stop;
        const double q_stop = q_peak + q_dot_peak * t_to_stop + 0.5 * q_ddot_to_stop * t_to_stop * t_to_stop;
        const double t_max_min_to_peak = -qd0 / k_actual;
        double q_lo, q_hi, g_lo, g_hi; /* q_endpoints_ordered / grad_q_endpoints_ordered */
        if (q_peak >= q0) { q_lo = q0; q_hi = q_peak; g_lo = 0; g_hi = 0.5 * t_move * t_move; }
        else { q_lo = q_peak; q_hi = q0; g_lo = 0.5 * t_move * t_move; g_hi = 0; }
        double q_min_to_peak = q_lo, q_max_to_peak = q_hi, gq_min_to_peak = g_lo, gq_max_to_peak = g_hi;
        if (t_max_min_to_peak > 0 && t_max_min_to_peak < t_move) {
            const double turn = q0 + qd0 * t_max_min_to_peak + 0.5 * k_actual * t_max_min_to_peak * t_max_min_to_peak;
            const double gturn = (0.5 * qd0 * qd0) / (k_actual * k_actual);
            if (k_actual >= 0) { q_min_to_peak = turn; gq_min_to_peak = gturn; }
            else { q_max_to_peak = turn; gq_max_to_peak = gturn; }
        }
        double qd_min_to_peak, qd_max_to_peak, gqd_min_to_peak, gqd_max_to_peak;
        if (q_dot_peak >= qd0) { qd_min_to_peak = qd0; qd_max_to_peak = q_dot_peak; gqd_min_to_peak = 0; gqd_max_to_peak = t_move; }
        else { qd_min_to_peak = q_dot_peak; qd_max_to_peak = qd0; gqd_min_to_peak = t_move; gqd_max_to_peak = 0; }
        double q_min_to_stop, q_max_to_stop, gq_min_to_stop, gq_max_to_stop;
        if (q_stop >= q_peak) {
            q_min_to_stop = q_peak; q_max_to_stop = q_stop;
            gq_min_to_stop = 0.5 * t_move * t_move; gq_max_to_stop = 0.5 * t_move * t_move + 0.5 * t_move * t_to_stop;
        } else {
            q_min_to_stop = q_stop; q_max_to_stop = q_peak;
            gq_min_to_stop = 0.5 * t_move * t_move + 0.5 * t_move * t_to_stop; gq_max_to_stop = 0.5 * t_move * t_move;
        }
        double qd_min_to_stop, qd_max_to_stop, gqd_min_to_stop, gqd_max_to_stop;
        if (q_dot_peak >= 0) { qd_min_to_stop = 0; qd_max_to_stop = q_dot_peak; gqd_min_to_stop = 0; gqd_max_to_stop = t_move; }
        else { qd_min_to_stop = q_dot_peak; qd_max_to_stop = 0; gqd_min_to_stop = t_move; gqd_max_to_stop = 0; }
        const bool a = q_min_to_peak <= q_min_to_stop, b = q_max_to_peak >= q_max_to_stop;
        const bool c = qd_min_to_peak <= qd_min_to_stop, d = qd_max_to_peak >= qd_max_to_stop;
        if (extremum) {
            extremum[i] = a ? q_min_to_peak : q_min_to_stop;
            extremum[i + n] = b ? q_max_to_peak : q_max_to_stop;
            extremum[i + 2 * n] = c ? qd_min_to_peak : qd_min_to_stop;
            extremum[i + 3 * n] = d ? qd_max_to_peak : qd_max_to_stop;
        }
        if (grad_diag) {
            grad_diag[i] = a ? gq_min_to_peak : gq_min_to_stop;
            grad_diag[i + n] = b ? gq_max_to_peak : gq_max_to_stop;
            grad_diag[i + 2 * n] = c ? gqd_min_to_peak : gqd_min_to_stop;
            grad_diag[i + 3 * n] = d ? gqd_max_to_peak : gqd_max_to_stop;
        }
    }
}

/* RT/Dynamics.cu:49-66 : link bounding-box PZ with pseudo-variables at key fields n, 2n, 3n */
static PZ make_link_box(Problem& P, Ctx& cx, int i) {
    PZ comp[3];
    for (int j = 0; j < 3; j++) {
        const okey_t key = (okey_t)1 << (j == 0 ? cx.kl.shift_qde(0) : j == 1 ? cx.kl.shift_qdae(0) : cx.kl.shift_qddae(0));
        const double g = P.rb.link_zonotope_generators[3 * i + j];
        comp[j] = pz_scalar_poly(cx, P.rb.link_zonotope_center[3 * i + j], &g, &key, 1);
    }
    return stack3(cx, comp[0], comp[1], comp[2]);
}

/* RT/Dynamics.cu:69-81 */
static void fk(Problem& P, Ctx& cx, int t) {
    const int T = P.T;
    PZ FK_R = pz_rpy(0, 0, 0);
    PZ FK_T(3, 1);
    for (int i = 0; i < P.J; i++) {
        PZ Pm = pz_matrix(3, 1, &P.rb.trans[3 * i]);
        FK_T = add(cx, FK_T, mul(cx, FK_R, Pm));
        FK_R = mul(cx, FK_R, P.R[i * T + t]);
        P.links[i * T + t] = add(cx, mul(cx, FK_R, P.links[i * T + t]), FK_T);
    }
}

/* RT/Dynamics.cu:83-181 */
static void rnea(Problem& P, Ctx& cx, int t, const std::vector<PZ>& mass_arr, const std::vector<PZ>& I_arr, std::vector<PZ>& u) {
    const int T = P.T, J = P.J;
    PZ w(3, 1), wdot(3, 1), w_aux(3, 1), linear_acc(3, 1);
    std::vector<PZ> F(J), N(J);
    linear_acc.center[2] = P.rb.gravity;
    for (int i = 0; i < J; i++) {
        const double* tr = &P.rb.trans[3 * i];
        const double* cm = &P.rb.com[3 * i];
        const PZ& Rt = P.R_t[i * T + t];
        const int ax = std::abs(P.rb.axes[i]) - 1;
        /* line 16 */
        linear_acc = mul(cx, Rt, add(cx, add(cx, linear_acc, cross_pz_mat(cx, wdot, tr)), cross_pz_pz(cx, w, cross_pz_mat(cx, w_aux, tr))));
        if (P.rb.axes[i] != 0) {
            w = mul(cx, Rt, w);                                           /* line 13 */
            add_one_dim(cx, w, P.qd_des[i * T + t], ax, 0);
            w_aux = mul(cx, Rt, w_aux);                                   /* line 14 */
            wdot = mul(cx, Rt, wdot);                                     /* line 15 */
            PZ temp(3, 1);
            add_one_dim(cx, temp, P.qd_des[i * T + t], ax, 0);
            wdot = add(cx, wdot, cross_pz_pz(cx, w_aux, temp));
            add_one_dim(cx, wdot, P.qdda_des[i * T + t], ax, 0);
            add_one_dim(cx, w_aux, P.qda_des[i * T + t], ax, 0);          /* line 14 */
        } else {
            w = mul(cx, Rt, w);
            w_aux = mul(cx, Rt, w_aux);
            wdot = mul(cx, Rt, wdot);
        }
        /* line 23 & 27 */
        F[i] = mul(cx, mass_arr[i], add(cx, add(cx, linear_acc, cross_pz_mat(cx, wdot, cm)), cross_pz_pz(cx, w, cross_pz_mat(cx, w_aux, cm))));
        /* line 29 */
        N[i] = add(cx, mul(cx, I_arr[i], wdot), cross_pz_pz(cx, w_aux, mul(cx, I_arr[i], w)));
    }
    PZ f(3, 1), nn(3, 1);
    for (int i = J - 1; i >= 0; i--) {
        const PZ& Rn = P.R[(i + 1) * T + t];
        nn = add(cx, add(cx, add(cx, N[i], mul(cx, Rn, nn)), cross_mat_pz(cx, &P.rb.com[3 * i], F[i])),
                 cross_mat_pz(cx, &P.rb.trans[3 * (i + 1)], mul(cx, Rn, f)));
        f = add(cx, mul(cx, Rn, f), F[i]);
        if (P.rb.axes[i] != 0) {
            const int ax = std::abs(P.rb.axes[i]) - 1;
            PZ ui = elem(nn, ax, 0);
            ui = add(cx, ui, scale(P.qdda_des[i * T + t], P.rb.armature[i]));
            ui = add(cx, ui, scale(P.qd_des[i * T + t], P.rb.damping[i]));
            u[i * T + t] = std::move(ui);
        }
    }
}

/* RT/CollisionChecking.cu:136-228: buffered obstacle = [3 obstacle gens | 3 link gens | 3 error gens] */
static void build_hyperplanes(Problem& P) {
    const int T = P.T, J = P.J, O = P.O;
    P.A.assign((size_t)T * J * O * 36 * 3, 0.0);
    P.d.assign((size_t)T * J * O * 36, 0.0);
    P.delta.assign((size_t)T * J * O * 36, 0.0);
    int combA[36], combB[36];
    {
        int a_id = 0, b_id = 1; /* RT/CollisionChecking.cu:26-39 */
        for (int i = 0; i < 36; i++) {
            combA[i] = a_id; combB[i] = b_id;
            if (b_id < 8) b_id++; else { a_id++; b_id = a_id + 1; }
        }
    }
#pragma omp parallel for schedule(static)
    for (int t = 0; t < T; t++)
        for (int l = 0; l < J; l++)
            for (int o = 0; o < O; o++) {
                double G[9][3], c[3];
                const double* ob = &P.obstacles[(size_t)o * 12];
                const double* lg = &P.link_gens[(size_t)(t * J + l) * 18];
                for (int ax = 0; ax < 3; ax++) {
                    c[ax] = ob[ax];
                    for (int g = 0; g < 3; g++) G[g][ax] = ob[(g + 1) * 3 + ax];
                    for (int g = 0; g < 6; g++) G[3 + g][ax] = lg[ax * 6 + g];
                }
                for (int p = 0; p < 36; p++) {
                    const double* ga = G[combA[p]];
                    const double* gb = G[combB[p]];
                    double cr[3] = {ga[1] * gb[2] - ga[2] * gb[1], ga[2] * gb[0] - ga[0] * gb[2], ga[0] * gb[1] - ga[1] * gb[0]};
                    const double nrm = std::sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
                    double C[3] = {0, 0, 0};
                    if (nrm > 0) { C[0] = cr[0] / nrm; C[1] = cr[1] / nrm; C[2] = cr[2] / nrm; }
                    const size_t idx = ((size_t)(t * J + l) * O + o) * 36 + p;
                    P.A[idx * 3 + 0] = C[0]; P.A[idx * 3 + 1] = C[1]; P.A[idx * 3 + 2] = C[2];
                    P.d[idx] = C[0] * c[0] + C[1] * c[1] + C[2] * c[2];
                    double dl = 0.0;
                    for (int j = 0; j < 9; j++) dl += std::fabs(C[0] * G[j][0] + C[1] * G[j][1] + C[2] * G[j][2]);
                    P.delta[idx] = dl;
                }
            }
}

/* RT/armour_main.cu:86-216 */
static void build(Problem& P, int num_threads) {
    auto t0 = std::chrono::steady_clock::now();
    const int T = P.T, J = P.J, n = P.n;
    P.ub = ultimate_bound(P.rb);
    if (!P.armtd) bezier_init(P);
    P.R.assign((size_t)(J + 1) * T, PZ()); P.R_t.assign((size_t)J * T, PZ());
    P.qd_des.assign((size_t)n * T, PZ()); P.qda_des.assign((size_t)n * T, PZ()); P.qdda_des.assign((size_t)n * T, PZ());
    P.links.assign((size_t)J * T, PZ()); P.u_nom.assign((size_t)n * T, PZ()); P.u_nom_int.assign((size_t)n * T, PZ());
    P.link_gens.assign((size_t)T * J * 18, 0.0);
    P.torque_radius.assign((size_t)n * T, 0.0);
    P.st = Stats();
    if (num_threads > 0) omp_set_num_threads(num_threads);

    Ctx cx0; cx0.kl.n = n; cx0.threshold = P.pr.simplify_threshold;
    /* KinematicsDynamics ctor, RT/Dynamics.cu:6-67 */
    std::vector<PZ> mass_nom(J), mass_unc(J), I_nom(J), I_unc(J), link_box(J);
    for (int i = 0; i < J; i++) {
        mass_nom[i] = pz_matrix(1, 1, &P.rb.mass[i]);
        mass_unc[i] = pz_matrix_uncertain(1, 1, &P.rb.mass[i], armour_mass_uncertainty(&P.rb, i));
        I_nom[i] = pz_matrix(3, 3, &P.rb.inertia[9 * i]);   /* symmetric: Eigen's column-major fill == row-major */
        I_unc[i] = pz_matrix_uncertain(3, 3, &P.rb.inertia[9 * i], armour_inertia_uncertainty(&P.rb, i));
        link_box[i] = make_link_box(P, cx0, i);
    }
    /* (a team wider than the T items only adds idle threads to every barrier: bench.py's sweep on a 256-thread host) */
    const int team = std::max(1, std::min(omp_get_max_threads(), T));
#pragma omp parallel num_threads(team)
    {
        Ctx cx; cx.kl.n = n; cx.threshold = P.pr.simplify_threshold;
#pragma omp for schedule(dynamic, 1)
        for (int t = 0; t < T; t++) { if (P.armtd) make_poly_zono_armtd(P, cx, t); else make_poly_zono(P, cx, t); }
#pragma omp for schedule(dynamic)
        for (int t = 0; t < T; t++) {
            for (int i = 0; i < J; i++) P.links[i * T + t] = link_box[i];
            fk(P, cx, t);
            for (int i = 0; i < J; i++) reduce_link_PZ(cx, P.links[i * T + t], &P.link_gens[(size_t)(t * J + i) * 18]);
            if (P.no_torque()) continue; /* CMP/armtd_main.cu:141-156, RT/armour_main.cu:149-165: forward kinematics only */
            rnea(P, cx, t, mass_nom, I_nom, P.u_nom);
            rnea(P, cx, t, mass_unc, I_unc, P.u_nom_int);
            for (int i = 0; i < n; i++) P.u_nom_int[i * T + t] = sub(cx, P.u_nom_int[i * T + t], P.u_nom[i * T + t]);
            for (int i = 0; i < n; i++) reduce(cx, P.u_nom[i * T + t]);
        }
#pragma omp critical
        P.st.merge(cx.st);
    }
    /* robust input bound, RT/armour_main.cu:172-205 */
    for (int t = 0; t < T && !P.no_torque(); t++) {   /* (RT/armour_main.cu:175: the radius stays zero without input constraints) */
        Interval rho(0.0);
        for (int i = 0; i < n; i++) {
            double lo, hi;
            to_interval(P.u_nom_int[i * T + t], &lo, &hi);
            const Interval tmp(lo, hi);
            rho = rho + tmp * tmp;
            P.torque_radius[i * T + t] = P.rb.alpha * (P.rb.M_max - P.rb.M_min) * P.ub.eps + 0.5 * std::max(std::fabs(lo), std::fabs(hi));
        }
        rho = isqrt(rho);
        for (int i = 0; i < n; i++) P.torque_radius[i * T + t] += 0.5 * rho.hi;
        for (int i = 0; i < n; i++) P.torque_radius[i * T + t] += P.u_nom[i * T + t].indep[0];
        for (int i = 0; i < n; i++) P.torque_radius[i * T + t] += P.rb.friction[i];
    }
    if (P.O > 0) build_hyperplanes(P);
    P.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

/* RT/Trajectory.cu:256-288 and :399-431 (value rows); :290-397 and :433-540 (gradient rows) */
static void joint_extremum(const Problem& P, const double* k, bool velocity, double* ext, double* grad) {
    const int n = P.n;
    const double DUR = P.pr.duration;
    for (int i = 0; i < n; i++) {
        const double ka = P.pr.k_range[i] * k[i];
        const double q0 = P.q0[i], a = P.Tqd0[i], b = P.TTqdd0[i];
        double e2, e3;
        if (!velocity) {
            const double sq = sqrt(64 * pow(a, 2) + 14 * a * b - 120 * ka * a + pow(b, 2));
            e2 = (2 * a + b + sq) / (5 * (6 * a - 12 * ka + b));
            e3 = (2 * a + b - sq) / (5 * (6 * a - 12 * ka + b));
        } else {
            const double sq = sqrt(6 * (150 * pow(ka, 2) - 180 * ka * a - 20 * ka * b + 54 * pow(a, 2) + 14 * a * b + pow(b, 2)));
            e2 = (18 * a - 30 * ka + 4 * b + sq) / (10 * (6 * a - 12 * ka + b));
            e3 = (18 * a - 30 * ka + 4 * b - sq) / (10 * (6 * a - 12 * ka + b));
        }
        auto f = [&](double t) { return velocity ? qd_des_func(q0, a, b, ka, t) : q_des_func(q0, a, b, ka, t); };
        const double v1 = f(0), v2 = f(e2), v3 = f(e3), v4 = f(1);
        double mn, mx; int mnId, mxId;
        if (v1 < v4) { mn = v1; mnId = 1; mx = v4; mxId = 4; } else { mn = v4; mnId = 4; mx = v1; mxId = 1; }
        if (0 <= e2 && e2 <= 1) { if (v2 < mn) { mn = v2; mnId = 2; } if (mx < v2) { mx = v2; mxId = 2; } }
        if (0 <= e3 && e3 <= 1) { if (v3 < mn) { mn = v3; mnId = 3; } if (mx < v3) { mx = v3; mxId = 3; } }
        if (ext) {
            /* value path uses min()/max(), which picks the same numbers as the id path above */
            ext[i] = velocity ? mn / DUR : mn;
            ext[i + n] = velocity ? mx / DUR : mx;
        }
        if (grad) {
            auto dk = [&](int id) -> double {
                switch (id) {
                    case 1: return 0.0;
                    case 2: return velocity ? qd_des_extrema_k_derivative(q0, a, b, ka, +1) : q_des_extrema_k_derivative(q0, a, b, ka, +1);
                    case 3: return velocity ? qd_des_extrema_k_derivative(q0, a, b, ka, -1) : q_des_extrema_k_derivative(q0, a, b, ka, -1);
                    default: return 1.0; /* t = 1: the reference returns 1.0 for velocity rows too (RT/Trajectory.cu:503,521) */
                }
            };
            const double sc = velocity ? P.pr.k_range[i] / DUR : P.pr.k_range[i];
            for (int j = 0; j < n; j++) {
                grad[(i)*n + j] = (i == j) ? dk(mnId) * sc : 0.0;
                grad[(i + n) * n + j] = (i == j) ? dk(mxId) * sc : 0.0;
            }
        }
    }
}

/* RT/CollisionChecking.cu:230-299 for one (t, link, obstacle) */
static inline void check_collision(const Problem& P, int t, int l, int o, const double* x, const double* dx /*[n][3] or null*/, double* c_out, double* grad_out) {
    const size_t base = ((size_t)(t * P.J + l) * P.O + o) * 36;
    double max_elt = -100000000;
    int max_id = 0;
    bool neg = false;
    for (int p = 0; p < 36; p++) {
        const double* Ap = &P.A[(base + p) * 3];
        double pos_res, neg_res;
        if (std::sqrt(Ap[0] * Ap[0] + Ap[1] * Ap[1] + Ap[2] * Ap[2]) > 0) {
            const double dot = Ap[0] * x[0] + Ap[1] * x[1] + Ap[2] * x[2];
            pos_res = dot - (P.d[base + p] + P.delta[base + p]);
            neg_res = -dot - (-P.d[base + p] + P.delta[base + p]);
        } else {
            pos_res = -100000000; neg_res = -100000000;
        }
        if (pos_res > max_elt) { max_elt = pos_res; max_id = p; neg = false; }
        if (neg_res > max_elt) { max_elt = neg_res; max_id = p; neg = true; }
    }
    if (c_out) *c_out = -max_elt;
    if (grad_out && dx) {
        const double* Am = &P.A[(base + max_id) * 3];
        for (int k = 0; k < P.n; k++) {
            const double dot = Am[0] * dx[k * 3 + 0] + Am[1] * dx[k * 3 + 1] + Am[2] * dx[k * 3 + 2];
            grad_out[k] = neg ? dot : -dot;
        }
    }
}

/* RT/NLPclass.cu:272-324 (g) and :330-396 (values); either output may be null */
static void eval_g_jac(Problem& P, const double* x, double* g, double* jac, int num_threads) {
    const int T = P.T, J = P.J, n = P.n, O = P.O;
    if (num_threads > 0) omp_set_num_threads(num_threads);
    Ctx cx; cx.kl.n = n;
    /* ARMTD mode, CMP/NLPclass.cu:245-330: collision rows first, then the state-limit rows; no torque rows */
    const size_t off_col = P.no_torque() ? 0 : (size_t)T * n, off_lim = off_col + (size_t)T * J * O;   /* (RT/NLPclass.cu:289-301: the same order with TURN_OFF_INPUT_CONSTRAINTS) */
    const int team = std::max(1, std::min(omp_get_max_threads(), T));   /* never more threads than time steps (idle members of a wide team) */
#pragma omp parallel for schedule(dynamic) num_threads(team)
    for (int t = 0; t < T; t++) {
        double cen[3], dcen[ARMOUR_MAX_FACTORS * 3];
        for (int j = 0; j < n && !P.no_torque(); j++) {
            if (g) { double c; slice_value(cx, P.u_nom[j * T + t], x, &c, nullptr); g[t * n + j] = c; }
            if (jac) slice_gradient(cx, P.u_nom[j * T + t], x, &jac[(size_t)(t * n + j) * n]);
        }
        for (int l = 0; l < J; l++) {
            slice_value(cx, P.links[l * T + t], x, cen, nullptr);
            if (jac) slice_gradient(cx, P.links[l * T + t], x, dcen);
            for (int o = 0; o < O; o++) {
                const size_t row = off_col + ((size_t)l * T + t) * O + o;
                check_collision(P, t, l, o, cen, jac ? dcen : nullptr, g ? &g[row] : nullptr, jac ? &jac[row * n] : nullptr);
            }
        }
    }
    if (P.armtd) {
        if (g) armtd_state_extremum(P, x, g + off_lim, nullptr);
        if (jac) {
            /* the reference clears only 4*n*n BYTES of these rows (CMP/Trajectory.cu:208) and leaves the rest as IPOPT
             * handed them over; the intended value, zero off the diagonal, is what is written here */
            double diag[4 * ARMOUR_MAX_FACTORS];
            armtd_state_extremum(P, x, nullptr, diag);
            for (int r = 0; r < 4 * n; r++)
                for (int c = 0; c < n; c++) jac[(off_lim + r) * n + c] = (c == r % n) ? diag[r] : 0.0;
        }
        return;
    }
    if (g) { joint_extremum(P, x, false, g + off_lim, nullptr); joint_extremum(P, x, true, g + off_lim + 2 * n, nullptr); }
    if (jac) { joint_extremum(P, x, false, nullptr, jac + off_lim * n); joint_extremum(P, x, true, nullptr, jac + (off_lim + 2 * n) * n); }
}

static double wrap_to_pi(double a) { /* RT/NLPclass.cu:6-15 */
    while (a < -M_PI) a += 2 * M_PI;
    while (a > M_PI) a -= 2 * M_PI;
    return a;
}

}  // namespace oracle

/* ====================================================================== C API for ctypes */
using namespace oracle;
extern "C" {

void oracle_fill_kinova(ArmourRobot* rb) { oracle_tables::fill_kinova_gen3_no_gripper(rb); }
void oracle_fill_kinova_gripper(ArmourRobot* rb) { oracle_tables::fill_kinova_gen3_gripper(rb); }
void oracle_fill_fetch(ArmourRobot* rb) { oracle_tables::fill_fetch(rb); }
#if ARMOUR_MAX_FACTORS >= 8
/* "Fetch 8-DOF": the Fetch arm (CMP/FetchInfo.h via the generated table) behind a torso yaw joint.  The derivation is restated here, not shared with
 * the product's include/armour_robot_fetch.h (armour_fill_fetch8); tests/test_robot_constants.py holds the two byte for byte.  Stand-in values for
 * the torso link: see that header -- the reference has no such robot (its key holds seven factors). */
void oracle_fill_fetch8(ArmourRobot* r) {
    ArmourRobot f;
    oracle_tables::fill_fetch(&f);
    std::memset(r, 0, sizeof(*r));
    r->num_joints = 9; r->num_factors = 8;
    const double torso_com[3] = {-0.0013, -0.0009, 0.2935};
    const double torso_I[9] = {0.3354, 0, -0.0162, 0, 0.3354, -0.0006, -0.0162, -0.0006, 0.0954};
    const double torso_box_c[3] = {0, 0, 0.363}, torso_box_g[3] = {0.12, 0.12, 0.363};
    r->axes[0] = 3; r->mass[0] = 10.7796;
    for (int e = 0; e < 3; e++) { r->com[e] = torso_com[e]; r->link_zonotope_center[e] = torso_box_c[e]; r->link_zonotope_generators[e] = torso_box_g[e]; }
    for (int e = 0; e < 9; e++) r->inertia[e] = torso_I[e];
    r->state_limits_lb[0] = -1.0; r->state_limits_ub[0] = 1.0; r->speed_limits[0] = 0.5; r->torque_limits[0] = 150.0;
    for (int i = 0; i < 8; i++) {
        const int j = i + 1;
        r->axes[j] = f.axes[i]; r->mass[j] = f.mass[i];
        r->friction[j] = f.friction[i]; r->damping[j] = f.damping[i]; r->armature[j] = f.armature[i];
        for (int e = 0; e < 3; e++) {
            r->trans[3 * j + e] = f.trans[3 * i + e]; r->rots[3 * j + e] = f.rots[3 * i + e]; r->com[3 * j + e] = f.com[3 * i + e];
            r->link_zonotope_center[3 * j + e] = f.link_zonotope_center[3 * i + e]; r->link_zonotope_generators[3 * j + e] = f.link_zonotope_generators[3 * i + e];
        }
        for (int e = 0; e < 9; e++) r->inertia[9 * j + e] = f.inertia[9 * i + e];
    }
    for (int e = 0; e < 3; e++) r->trans[27 + e] = f.trans[24 + e];
    for (int i = 0; i < 7; i++) {
        r->continuous[i + 1] = f.continuous[i];
        r->state_limits_lb[i + 1] = f.state_limits_lb[i]; r->state_limits_ub[i + 1] = f.state_limits_ub[i];
        r->speed_limits[i + 1] = f.speed_limits[i]; r->torque_limits[i + 1] = f.torque_limits[i];
    }
    r->mass_uncertainty = f.mass_uncertainty; r->inertia_uncertainty = f.inertia_uncertainty;
    r->gravity = f.gravity; r->alpha = f.alpha; r->V_m = f.V_m; r->M_min = f.M_min; r->M_max = f.M_max; r->K = f.K;
}
#endif
void oracle_fill_default_params(ArmourParams* pr, int T) { oracle_tables::fill_default_params(pr, T); }

/* The scalar Bezier helpers on their own (tests/test_ref_bezier.py checks them against the REFERENCE's functions compiled
 * from RT/Trajectory.cu:542-822, oracle/_ref/libref_bezier.so, and against vectors recorded from that library).
 * which: 0 q_des_func  1 qd_des_func  2 qdd_des_func (x = t)   3/4 q_des_extrema{2,3}_k_derivative   5/6 qd_des_extrema{2,3}_k_derivative
 *        7 q_des_k_indep  8 qd_des_k_indep  9 qdd_des_k_indep (x = s; k unused) */
double oracle_bezier_scalar(int which, double q0, double a, double b, double k, double x, double DUR) {
    switch (which) {
        case 0: return q_des_func(q0, a, b, k, x);
        case 1: return qd_des_func(q0, a, b, k, x);
        case 2: return qdd_des_func(q0, a, b, k, x);
        case 3: return q_des_extrema_k_derivative(q0, a, b, k, +1);
        case 4: return q_des_extrema_k_derivative(q0, a, b, k, -1);
        case 5: return qd_des_extrema_k_derivative(q0, a, b, k, +1);
        case 6: return qd_des_extrema_k_derivative(q0, a, b, k, -1);
        case 7: return q_des_k_indep(q0, a, b, x);
        case 8: return qd_des_k_indep(q0, a, b, x, DUR);
        default: return qdd_des_k_indep(q0, a, b, x, DUR);
    }
}

void* oracle_create(const ArmourRobot* rb, const ArmourParams* pr) {
    Problem* P = new Problem();
    P->rb = *rb; P->pr = *pr;
    P->T = pr->num_time_steps; P->J = rb->num_joints; P->n = rb->num_factors;
    return P;
}
void oracle_destroy(void* h) { delete (Problem*)h; }

/* P1: RT/armour_main.cu:36-216 */
int oracle_set_problem(void* h, const double* q0, const double* qd0, const double* qdd0, const double* q_des, int O, const double* obstacles, int num_threads) {
    Problem& P = *(Problem*)h;
    if (P.T % 2 != 0 || P.n > ARMOUR_MAX_FACTORS) return -1;
    for (int i = 0; i < P.n; i++) { P.q0[i] = q0[i]; P.qd0[i] = qd0[i]; P.qdd0[i] = qdd0[i]; P.q_des[i] = q_des[i]; }
    P.O = O;
    P.obstacles.assign(obstacles, obstacles + (size_t)O * 12);
    build(P, num_threads);
    return 0;
}
/* ARMTD comparison mode: CMP/armtd_main.cu:36-156.  jrs [n][6][T], k_range [n] */
int oracle_set_problem_armtd(void* h, const double* q0, const double* qd0, const double* q_des, const double* jrs, const double* k_range,
                             int O, const double* obstacles, int num_threads) {
    Problem& P = *(Problem*)h;
    if (P.n > ARMOUR_MAX_FACTORS) return -1;
    for (int i = 0; i < P.n; i++) { P.q0[i] = q0[i]; P.qd0[i] = qd0[i]; P.qdd0[i] = 0; P.q_des[i] = q_des[i]; P.k_range_a[i] = k_range[i]; }
    P.armtd = true;
    P.jrs.assign(jrs, jrs + (size_t)P.n * 6 * P.T);
    P.O = O;
    P.obstacles.assign(obstacles, obstacles + (size_t)O * 12);
    build(P, num_threads);
    return 0;
}
int oracle_num_constraints(void* h) { Problem& P = *(Problem*)h; return (P.no_torque() ? 0 : P.n * P.T) + P.J * P.T * P.O + 4 * P.n; }   /* RT/NLPclass.cu:46-54 */
double oracle_build_ms(void* h) { return ((Problem*)h)->build_ms; }

void oracle_eval_g_jac(void* h, const double* k, double* g, double* jac, int num_threads) { eval_g_jac(*(Problem*)h, k, g, jac, num_threads); }

/* timed loop for bench.py's cpu_baseline: `reps` fused evaluations over `nk` k-points, returns seconds */
double oracle_time_eval(void* h, const double* ks, int nk, int reps, double* g, double* jac, int num_threads) {
    Problem& P = *(Problem*)h;
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++) eval_g_jac(P, ks + (size_t)(r % nk) * P.n, g, jac, num_threads);
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

/* RT/NLPclass.cu:87-165 */
void oracle_get_bounds(void* h, double* x_l, double* x_u, double* g_l, double* g_u) {
    Problem& P = *(Problem*)h;
    const int T = P.T, J = P.J, n = P.n, O = P.O;
    for (int i = 0; i < n; i++) { x_l[i] = -1.0; x_u[i] = 1.0; }
    for (int i = 0; i < T && !P.no_torque(); i++)   /* RT/NLPclass.cu:117 */
        for (int j = 0; j < n; j++) {
            g_l[i * n + j] = -P.rb.torque_limits[j] + P.torque_radius[j * T + i];
            g_u[i * n + j] = P.rb.torque_limits[j] - P.torque_radius[j * T + i];
        }
    size_t off = P.no_torque() ? 0 : (size_t)n * T; /* CMP/NLPclass.cu:73-140: the same rows without the torque block */
    for (size_t i = off; i < off + (size_t)T * J * O; i++) { g_l[i] = -1e19; g_u[i] = 0; }
    off += (size_t)T * J * O;
    for (int rep = 0; rep < 2; rep++, off += n)
        for (int i = 0; i < n; i++) { g_l[off + i] = P.rb.state_limits_lb[i] + P.ub.qe; g_u[off + i] = P.rb.state_limits_ub[i] - P.ub.qe; }
    for (int rep = 0; rep < 2; rep++, off += n)
        for (int i = 0; i < n; i++) { g_l[off + i] = -P.rb.speed_limits[i] + P.ub.qde; g_u[off + i] = P.rb.speed_limits[i] - P.ub.qde; }
}

/* RT/NLPclass.cu:207-236 */
double oracle_eval_f(void* h, const double* x) {
    Problem& P = *(Problem*)h;
    double obj = 0;
    /* the reference sums the four wrapped (continuous) joints first, then the others (:225-231) */
    for (int pass = 0; pass < 2; pass++)
        for (int i = 0; i < P.n; i++) {
            if ((P.rb.continuous[i] != 0) != (pass == 0)) continue;
            const double qp = P.armtd ? P.q0[i] + P.qd0[i] * 0.5 + P.k_range_a[i] * x[i] * 0.125 /* CMP/NLPclass.cu:197 */
                                      : q_des_func(P.q0[i], P.Tqd0[i], P.TTqdd0[i], P.pr.k_range[i] * x[i], P.pr.t_plan);
            const double e = P.rb.continuous[i] ? wrap_to_pi(P.q_des[i] - qp) : (P.q_des[i] - qp);
            obj += pow(e, 2);
        }
    return obj * P.pr.cost_scale;
}
/* RT/NLPclass.cu:241-267 */
void oracle_eval_grad_f(void* h, const double* x, double* grad) {
    Problem& P = *(Problem*)h;
    const double tp = P.pr.t_plan;
    for (int i = 0; i < P.n; i++) {
        const double qp = P.armtd ? P.q0[i] + P.qd0[i] * 0.5 + P.k_range_a[i] * x[i] * 0.125 /* CMP/NLPclass.cu:229-230 */
                                  : q_des_func(P.q0[i], P.Tqd0[i], P.TTqdd0[i], P.pr.k_range[i] * x[i], tp);
        const double dk = P.armtd ? P.k_range_a[i] * 0.125 : pow(tp, 3) * (6 * pow(tp, 2) - 15 * tp + 10) * P.pr.k_range[i];
        grad[i] = P.rb.continuous[i] ? (2 * wrap_to_pi(qp - P.q_des[i]) * dk) : (2 * (qp - P.q_des[i]) * dk);
        grad[i] *= P.pr.cost_scale;
    }
}

/* ---- table getters (what the device path must reproduce) ---- */
void oracle_get_torque_radius(void* h, double* out /*[n][T]*/) { Problem& P = *(Problem*)h; memcpy(out, P.torque_radius.data(), sizeof(double) * P.n * P.T); }
void oracle_get_link_generators(void* h, double* out /*[T*J][3][6]*/) { Problem& P = *(Problem*)h; memcpy(out, P.link_gens.data(), sizeof(double) * P.link_gens.size()); }
void oracle_get_hyperplanes(void* h, double* A, double* d, double* delta) {
    Problem& P = *(Problem*)h;
    if (A) memcpy(A, P.A.data(), sizeof(double) * P.A.size());
    if (d) memcpy(d, P.d.data(), sizeof(double) * P.d.size());
    if (delta) memcpy(delta, P.delta.data(), sizeof(double) * P.delta.size());
}
/* which: 0 = links(l,t) [3x1], 1 = u_nom(j,t) [1x1], 2 = u_nom_int(j,t) (disturbance) [1x1] */
static const PZ& pick(Problem& P, int which, int i, int t) {
    return which == 0 ? P.links[i * P.T + t] : which == 1 ? P.u_nom[i * P.T + t] : P.u_nom_int[i * P.T + t];
}
int oracle_pz_size(void* h, int which, int i, int t) { return (int)pick(*(Problem*)h, which, i, t).poly.size(); }
void oracle_pz_get(void* h, int which, int i, int t, double* center, double* indep, uint64_t* keys, double* coeffs) {
    const PZ& p = pick(*(Problem*)h, which, i, t);
    const int n = p.sz();
    for (int e = 0; e < n; e++) { center[e] = p.center[e]; indep[e] = p.indep[e]; }
    for (size_t m = 0; m < p.poly.size(); m++) {
        keys[m] = (uint64_t)p.poly[m].key;   /* (tables returned here hold k-only monomials: below 2^(2n)) */
        for (int e = 0; e < n; e++) coeffs[m * n + e] = p.poly[m].c[e];
    }
}
/* total monomials: sum over (l,t) of links and over (j,t) of u_nom (Sigma M_link, Sigma M_torque) */
void oracle_table_sizes(void* h, int64_t* sum_link, int64_t* sum_torque, int64_t* max_link, int64_t* max_torque) {
    Problem& P = *(Problem*)h;
    int64_t sl = 0, stq = 0, ml = 0, mt = 0;
    for (auto& p : P.links) { sl += p.poly.size(); ml = std::max<int64_t>(ml, p.poly.size()); }
    for (auto& p : P.u_nom) { stq += p.poly.size(); mt = std::max<int64_t>(mt, p.poly.size()); }
    *sum_link = sl; *sum_torque = stq; *max_link = ml; *max_torque = mt;
}
void oracle_stats(void* h, uint64_t* out6) {
    Problem& P = *(Problem*)h;
    out6[0] = P.st.mul_calls; out6[1] = P.st.mul_pairs; out6[2] = P.st.simplify_calls;
    out6[3] = P.st.simplify_terms; out6[4] = P.st.max_raw_terms; out6[5] = P.st.max_out_terms;
}

/* scalar slices used by the containment tests (RT/PZ_tests.cu:198-220) */
void oracle_slice_torque(void* h, const double* k, double* center /*[T][n]*/) {
    Problem& P = *(Problem*)h;
    Ctx cx; cx.kl.n = P.n;
    for (int t = 0; t < P.T; t++) for (int j = 0; j < P.n; j++) slice_value(cx, P.u_nom[j * P.T + t], k, &center[t * P.n + j], nullptr);
}
void oracle_slice_links(void* h, const double* k, double* center /*[T][J][3]*/) {
    Problem& P = *(Problem*)h;
    Ctx cx; cx.kl.n = P.n;
    for (int t = 0; t < P.T; t++) for (int l = 0; l < P.J; l++) slice_value(cx, P.links[l * P.T + t], k, &center[(t * P.J + l) * 3], nullptr);
}
/* one operator of pz.hpp on caller-supplied operands (mirror of armour_debug_pz_op, same op codes and layout) */
int oracle_pz_op(int op, int nops, const int* sz, const int* cnt, const uint64_t* const* keys, const double* const* coef,
                 const double* cen, const double* ind, const double* consts, int r, double threshold, int out_cap,
                 uint64_t* out_keys, double* out_coef, double* out_misc) {
    Ctx cx; cx.kl.n = 7; cx.threshold = threshold;
    PZ in[3];
    for (int o = 0; o < nops; o++) {
        in[o] = sz[o] == 9 ? PZ(3, 3) : sz[o] == 3 ? PZ(3, 1) : PZ(1, 1);
        for (int e = 0; e < sz[o]; e++) { in[o].center[e] = cen[o * 9 + e]; in[o].indep[e] = ind[o * 9 + e]; }
        for (int m = 0; m < cnt[o]; m++) {
            Mono mo{};
            mo.key = keys[o][m];
            for (int e = 0; e < sz[o]; e++) mo.c[e] = coef[o][(size_t)m * sz[o] + e];
            in[o].poly.push_back(mo);
        }
    }
    PZ out;
    switch (op) {
        case 0: case 1: case 2: case 3: out = mul(cx, in[0], in[1]); break;
        case 4: out = add(cx, in[0], in[1]); break;
        case 5: out = sub(cx, in[0], in[1]); break;
        case 6: out = in[0]; add_one_dim(cx, out, in[1], r, 0); break;
        case 7: out = stack3(cx, in[0], in[1], in[2]); break;
        case 8: out = cross_pz_mat(cx, in[0], consts); break;
        case 9: out = cross_mat_pz(cx, consts, in[0]); break;
        case 10: out = cross_pz_pz(cx, in[0], in[1]); break;
        default: out = add(cx, scale(in[0], consts[0]), scale(in[1], consts[1])); break;
    }
    const int n = (int)out.poly.size(), osz = out.sz();
    if (n > out_cap) return -1;
    for (int m = 0; m < n; m++) {
        out_keys[m] = (uint64_t)out.poly[m].key;
        for (int e = 0; e < osz; e++) out_coef[(size_t)m * osz + e] = out.poly[m].c[e];
    }
    out_misc[0] = n; out_misc[1] = osz; out_misc[2] = cx.st.min_margin;
    for (int e = 0; e < osz; e++) { out_misc[3 + e] = out.center[e]; out_misc[12 + e] = out.indep[e]; }
    return 0;
}
double oracle_min_margin(void* h) { return ((Problem*)h)->st.min_margin; }
int oracle_max_threads(void) { return omp_get_max_threads(); }
int oracle_abi_max_factors(void) { return ARMOUR_MAX_FACTORS; }   /* 7: u64 keys; 8: the -DARMOUR_KEY128 build */

}  // extern "C"
