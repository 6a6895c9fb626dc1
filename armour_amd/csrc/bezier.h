// Degree-5 Bezier joint trajectory of ARMOUR (RT/Trajectory.h:10-31): closed-form position / velocity /
// acceleration, their extrema over t in [0,1] and the derivative of those extrema w.r.t. the trajectory
// parameter.  Used on the device by the joint-limit rows of the P2 kernel and on the host by eval_f.
//
// Reference: RT/Trajectory.cu:256-540 (returnJoint{Position,Velocity}Extremum{,Gradient}) and the helper
// functions at :542-822.  The reference's *_extrema{2,3}_k_derivative helpers are MATLAB-symbolic
// expansions of d/dk f(t*(k), k); here they are written as the chain rule
// df/dk|_{t*} + df/dt|_{t*} * dt*/dk, the same function up to rounding.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace bez {

__host__ __device__ inline double p2(double x) { return x * x; }
__host__ __device__ inline double p3(double x) { return x * x * x; }
__host__ __device__ inline double p4(double x) { const double y = x * x; return y * y; }
__host__ __device__ inline double p5(double x) { const double y = x * x; return y * y * x; }

// RT/Trajectory.cu:542-558
__host__ __device__ inline double q_des(double q0, double a, double b, double k, double t) {
    const double u = t - 1;
    const double B0 = -p5(u), B1 = 5 * t * p4(u), B2 = -10 * p2(t) * p3(u), B3 = 10 * p3(t) * p2(u),
                 B4 = -5 * p4(t) * u, B5 = p5(t);
    const double b0 = q0, b1 = q0 + a / 5, b2 = q0 + (2 * a) / 5 + b / 20, b3 = q0 + k;
    return B0 * b0 + B1 * b1 + B2 * b2 + B3 * b3 + B4 * b3 + B5 * b3;
}
// RT/Trajectory.cu:560-574
__host__ __device__ inline double qd_des(double q0, double a, double b, double k, double t) {
    const double u = t - 1.0;
    const double dB0 = p4(u) * -5.0, dB1 = t * p3(u) * 20.0 + p4(u) * 5.0,
                 dB2 = t * p3(u) * -20.0 - (t * t) * p2(u) * 30.0,
                 dB3 = p3(t) * (t * 2.0 - 2.0) * 10.0 + (t * t) * p2(u) * 30.0,
                 dB4 = p3(t) * u * -20.0 - p4(t) * 5.0, dB5 = p4(t) * 5.0;
    const double b0 = q0, b1 = q0 + a / 5, b2 = q0 + (2 * a) / 5 + b / 20, b3 = q0 + k;
    return dB0 * b0 + dB1 * b1 + dB2 * b2 + dB3 * b3 + dB4 * b3 + dB5 * b3;
}
// RT/Trajectory.cu:576-602
__host__ __device__ inline double qdd_des(double q0, double a, double b, double k, double t) {
    const double u = t - 1.0;
    const double ddB0 = -20.0 * p3(u), ddB1 = 40.0 * p3(u) + 60.0 * t * p2(u),
                 ddB2 = -20.0 * p3(u) - 120.0 * t * p2(u) - 30.0 * t * t * (2.0 * t - 2.0),
                 ddB3 = 20.0 * p3(t) + 60.0 * t * p2(u) + 60.0 * t * t * (2.0 * t - 2.0),
                 ddB4 = -40.0 * p3(t) - 60.0 * t * t * u, ddB5 = 20.0 * p3(t);
    const double b0 = q0, b1 = q0 + a / 5, b2 = q0 + (2 * a) / 5 + b / 20, b3 = q0 + k;
    return ddB0 * b0 + ddB1 * b1 + ddB2 * b2 + ddB3 * b3 + ddB4 * b3 + ddB5 * b3;
}

// stationary points of q_des in t (RT/Trajectory.cu:263-264) and d/dk of the extremum value
__host__ __device__ inline void q_stationary(double a, double b, double k, double* e2, double* e3) {
    const double sq = sqrt(64 * p2(a) + 14 * a * b - 120 * k * a + p2(b));
    const double den = 5 * (6 * a - 12 * k + b);
    *e2 = (2 * a + b + sq) / den;
    *e3 = (2 * a + b - sq) / den;
}
__host__ __device__ inline double q_extremum_dk(double q0, double a, double b, double k, int sign) {
    const double sq = sqrt(64 * a * a + 14 * a * b - 120 * k * a + b * b);
    const double den = 6 * a - 12 * k + b;
    const double num = 2 * a + b + sign * sq;
    const double ts = num / (5 * den);
    const double dnum = sign * (-120 * a) / (2 * sq);
    const double dts = (dnum * 5 * den - num * 5 * (-12)) / (25 * den * den);
    const double dqdk = ts * ts * ts * (6 * ts * ts - 15 * ts + 10);
    return dqdk + qd_des(q0, a, b, k, ts) * dts;
}
// stationary points of qd_des in t (RT/Trajectory.cu:406-407) and d/dk of the extremum value
__host__ __device__ inline void qd_stationary(double a, double b, double k, double* e2, double* e3) {
    const double sq = sqrt(6 * (150 * p2(k) - 180 * k * a - 20 * k * b + 54 * p2(a) + 14 * a * b + p2(b)));
    const double den = 10 * (6 * a - 12 * k + b);
    *e2 = (18 * a - 30 * k + 4 * b + sq) / den;
    *e3 = (18 * a - 30 * k + 4 * b - sq) / den;
}
__host__ __device__ inline double qd_extremum_dk(double q0, double a, double b, double k, int sign) {
    const double E = 150 * k * k - 180 * k * a - 20 * k * b + 54 * a * a + 14 * a * b + b * b;
    const double sq = sqrt(6 * E);
    const double den = 6 * a - 12 * k + b;
    const double num = 18 * a - 30 * k + 4 * b + sign * sq;
    const double ts = num / (10 * den);
    const double dE = 300 * k - 180 * a - 20 * b;
    const double dnum = -30 + sign * (6 * dE) / (2 * sq);
    const double dts = (dnum * 10 * den - num * 10 * (-12)) / (100 * den * den);
    const double dqddk = 30 * ts * ts * (ts - 1) * (ts - 1);
    return dqddk + qdd_des(q0, a, b, k, ts) * dts;
}

// One joint's min / max of position (velocity == false) or velocity (true) over t in [0,1] and their
// derivative w.r.t. the *normalised* parameter k in [-1,1] (RT/Trajectory.cu:290-397, 433-540).
// The t = 1 branch returns slope 1.0 for velocity rows as well, as the reference does (:503,:521).
__host__ __device__ inline void joint_extremum(double q0, double a, double b, double k_norm, double k_range,
                                               double duration, bool velocity, double* mn_out, double* mx_out,
                                               double* dmn_out, double* dmx_out) {
    const double ka = k_range * k_norm;
    double e2, e3;
    if (!velocity) q_stationary(a, b, ka, &e2, &e3); else qd_stationary(a, b, ka, &e2, &e3);
    const double v1 = velocity ? qd_des(q0, a, b, ka, 0.0) : q_des(q0, a, b, ka, 0.0);
    const double v2 = velocity ? qd_des(q0, a, b, ka, e2) : q_des(q0, a, b, ka, e2);
    const double v3 = velocity ? qd_des(q0, a, b, ka, e3) : q_des(q0, a, b, ka, e3);
    const double v4 = velocity ? qd_des(q0, a, b, ka, 1.0) : q_des(q0, a, b, ka, 1.0);
    double mn, mx;
    int mnId, mxId;
    if (v1 < v4) { mn = v1; mnId = 1; mx = v4; mxId = 4; } else { mn = v4; mnId = 4; mx = v1; mxId = 1; }
    if (0 <= e2 && e2 <= 1) { if (v2 < mn) { mn = v2; mnId = 2; } if (mx < v2) { mx = v2; mxId = 2; } }
    if (0 <= e3 && e3 <= 1) { if (v3 < mn) { mn = v3; mnId = 3; } if (mx < v3) { mx = v3; mxId = 3; } }
    *mn_out = velocity ? mn / duration : mn;
    *mx_out = velocity ? mx / duration : mx;
    const double sc = velocity ? k_range / duration : k_range;
    double d[2];
    const int ids[2] = {mnId, mxId};
    for (int s = 0; s < 2; s++) {
        switch (ids[s]) {
            case 1: d[s] = 0.0; break;
            case 2: d[s] = velocity ? qd_extremum_dk(q0, a, b, ka, +1) : q_extremum_dk(q0, a, b, ka, +1); break;
            case 3: d[s] = velocity ? qd_extremum_dk(q0, a, b, ka, -1) : q_extremum_dk(q0, a, b, ka, -1); break;
            default: d[s] = 1.0; break;
        }
    }
    *dmn_out = d[0] * sc;
    *dmx_out = d[1] * sc;
}

}  // namespace bez
