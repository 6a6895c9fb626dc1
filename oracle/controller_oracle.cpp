/*
 * oracle/controller_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  PARITY UNPINNED: the reference holds no recorded
 * outputs of its controller; this restatement is checked by invariants (tests/test_controller.py).
 *
 * CPU restatement of the reference's robust tracking controller for one state:
 *   kinova_src/kinova_simulator_interfaces/kinova_robust_controllers_mex/{spatial.cpp, spatial_interval.cpp, rnea.cpp,
 *   robot_models.cpp:124-255, robust_controller.cpp:63-168}
 * written independently of armour_amd/csrc/controller_core.h: plain arrays, one generic scalar (double or
 * oracle::Interval from interval.hpp), functions instead of operator-heavy classes.
 */
#include <cmath>
#include <cstring>

#include "../include/armour_types.h"
#include "interval.hpp"

namespace {
using oracle::Interval;

template <class S> S lift(double x);
template <> double lift<double>(double x) { return x; }
template <> Interval lift<Interval>(double x) { return Interval(x); }

template <class S> struct Vec { S v[3]; };
template <class S> struct Mat { S m[3][3]; };

template <class S> Vec<S> vadd(const Vec<S>& a, const Vec<S>& b) { Vec<S> r; for (int i = 0; i < 3; i++) r.v[i] = a.v[i] + b.v[i]; return r; }
template <class S> Vec<S> vsub(const Vec<S>& a, const Vec<S>& b) { Vec<S> r; for (int i = 0; i < 3; i++) r.v[i] = a.v[i] - b.v[i]; return r; }
template <class S> Vec<S> vneg(const Vec<S>& a) { Vec<S> r; for (int i = 0; i < 3; i++) r.v[i] = -a.v[i]; return r; }
template <class S, class K> Vec<S> vmul(const Vec<S>& a, K s) { Vec<S> r; for (int i = 0; i < 3; i++) r.v[i] = a.v[i] * s; return r; }
template <class S, class K> Vec<S> smulv(K s, const Vec<S>& a) { Vec<S> r; for (int i = 0; i < 3; i++) r.v[i] = s * a.v[i]; return r; }
template <class S> Vec<S> vcross(const Vec<S>& a, const Vec<S>& b) {
    Vec<S> r;
    r.v[0] = a.v[1] * b.v[2] - a.v[2] * b.v[1]; r.v[1] = a.v[2] * b.v[0] - a.v[0] * b.v[2]; r.v[2] = a.v[0] * b.v[1] - a.v[1] * b.v[0];
    return r;
}
template <class S> S vdot(const Vec<S>& a, const Vec<S>& b) { S s = a.v[0] * b.v[0]; s = s + a.v[1] * b.v[1]; s = s + a.v[2] * b.v[2]; return s; }
template <class S> Vec<S> mv(const Mat<S>& A, const Vec<S>& x) {
    Vec<S> r;
    for (int i = 0; i < 3; i++) { S s = A.m[i][0] * x.v[0]; s = s + A.m[i][1] * x.v[1]; s = s + A.m[i][2] * x.v[2]; r.v[i] = s; }
    return r;
}
template <class S> Mat<S> mm(const Mat<S>& A, const Mat<S>& B) {
    Mat<S> r;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { S s = A.m[i][0] * B.m[0][j]; s = s + A.m[i][1] * B.m[1][j]; s = s + A.m[i][2] * B.m[2][j]; r.m[i][j] = s; }
    return r;
}
template <class S> Mat<S> madd(const Mat<S>& A, const Mat<S>& B) { Mat<S> r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = A.m[i][j] + B.m[i][j]; return r; }
template <class S> Mat<S> msub(const Mat<S>& A, const Mat<S>& B) { Mat<S> r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = A.m[i][j] - B.m[i][j]; return r; }
template <class S, class K> Mat<S> mmul(const Mat<S>& A, K s) { Mat<S> r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = A.m[i][j] * s; return r; }
template <class S, class K> Mat<S> smulm(K s, const Mat<S>& A) { Mat<S> r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = s * A.m[i][j]; return r; }
template <class S> Mat<S> mt(const Mat<S>& A) { Mat<S> r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = A.m[j][i]; return r; }
template <class S> Mat<S> eye() { Mat<S> r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = lift<S>(i == j ? 1.0 : 0.0); return r; }
template <class S> Vec<S> zeros() { Vec<S> r; for (int i = 0; i < 3; i++) r.v[i] = lift<S>(0.0); return r; }
template <class S> Mat<S> skew(const Vec<S>& w) {
    Mat<S> r = eye<S>();
    const S z = lift<S>(0.0);
    r.m[0][0] = z; r.m[0][1] = -w.v[2]; r.m[0][2] = w.v[1];
    r.m[1][0] = w.v[2]; r.m[1][1] = z; r.m[1][2] = -w.v[0];
    r.m[2][0] = -w.v[1]; r.m[2][1] = w.v[0]; r.m[2][2] = z;
    return r;
}

template <class S> struct Twist { Vec<S> w, v; };
template <class S> struct Wrench { Vec<S> tau, f; };
template <class S> struct Xform { Mat<S> R; Vec<S> p; };
template <class S> struct Inertia { S m; Mat<S> Ibar, mch; };

template <class S> Twist<S> tw_add(const Twist<S>& a, const Twist<S>& b) { return {vadd(a.w, b.w), vadd(a.v, b.v)}; }
template <class S, class K> Twist<S> tw_mul(const Twist<S>& a, K s) { return {vmul(a.w, s), vmul(a.v, s)}; }
template <class S> Twist<S> tw_cross(const Twist<S>& a, const Twist<S>& b) {
    const Mat<S> wh = skew(a.w);
    return {mv(wh, b.w), vadd(mv(wh, b.v), vcross(a.v, b.w))};
}
template <class S> Xform<S> x_from_twist(const Twist<S>& z, double th) {
    const Mat<S> I = eye<S>(), wh = skew(z.w);
    Xform<S> x;
    x.R = madd(madd(I, mmul(wh, std::sin(th))), mm(smulm(1 - std::cos(th), wh), wh));
    const Vec<S> p = mv(mm(msub(I, x.R), wh), z.v);
    x.p = vneg(mv(mt(x.R), p));
    return x;
}
template <class S> Twist<S> x_apply(const Xform<S>& x, const Twist<S>& z) { return {mv(x.R, z.w), mv(x.R, vsub(z.v, vcross(x.p, z.w)))}; }
template <class S> Twist<S> x_invapply(const Xform<S>& x, const Twist<S>& z) { const Vec<S> nw = mv(mt(x.R), z.w); return {nw, vadd(mv(mt(x.R), z.v), vcross(x.p, nw))}; }
template <class S> Wrench<S> x_invapply(const Xform<S>& x, const Wrench<S>& w) { const Mat<S> Rt = mt(x.R); return {vadd(mv(Rt, w.tau), vcross(x.p, mv(Rt, w.f))), mv(Rt, w.f)}; }
template <class S> Xform<S> x_compose(const Xform<S>& x, const Xform<S>& y) { return {mm(x.R, y.R), vadd(y.p, mv(mt(y.R), x.p))}; }
template <class S> Xform<S> x_inverse(const Xform<S>& x) { return {mt(x.R), vneg(mv(x.R, x.p))}; }
template <class S> Wrench<S> i_apply(const Inertia<S>& I, const Twist<S>& z) { return {vadd(mv(I.Ibar, z.w), mv(I.mch, z.v)), vsub(smulv(I.m, z.v), mv(I.mch, z.w))}; }

constexpr int NJ = ARMOUR_MAX_FACTORS;
template <class S> struct Model { int n; Twist<S> S_[NJ]; Inertia<S> I[NJ]; Xform<S> X[NJ]; S transI[NJ]; double damping[NJ], friction[NJ]; Twist<S> grav; };

void rpy(double roll, double pitch, double yaw, Mat<double>& R) {
    R.m[0][0] = cos(pitch) * cos(yaw); R.m[0][1] = -cos(pitch) * sin(yaw); R.m[0][2] = sin(pitch);
    R.m[1][0] = cos(roll) * sin(yaw) + cos(yaw) * sin(pitch) * sin(roll); R.m[1][1] = cos(roll) * cos(yaw) - sin(pitch) * sin(roll) * sin(yaw); R.m[1][2] = -cos(pitch) * sin(roll);
    R.m[2][0] = sin(roll) * sin(yaw) - cos(roll) * cos(yaw) * sin(pitch); R.m[2][1] = cos(yaw) * sin(roll) + cos(roll) * sin(pitch) * sin(yaw); R.m[2][2] = cos(pitch) * cos(roll);
}

void build(const ArmourRobot& rb, double eps, Model<double>& md, Model<Interval>& im) {
    const int n = rb.num_factors;
    Twist<double> S[NJ]; Inertia<double> I[NJ]; Xform<double> X[NJ], C[NJ];
    for (int i = 0; i < n; i++) {
        S[i].w = zeros<double>(); S[i].v = zeros<double>();
        S[i].w.v[std::abs(rb.axes[i]) - 1] = rb.axes[i] > 0 ? 1.0 : -1.0;
        Vec<double> c{{rb.com[3 * i], rb.com[3 * i + 1], rb.com[3 * i + 2]}};
        Mat<double> Ic;
        for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) Ic.m[r][k] = rb.inertia[9 * i + 3 * r + k];
        const Mat<double> ch = skew(c);
        I[i].m = rb.mass[i]; I[i].mch = smulm(rb.mass[i], ch); I[i].Ibar = msub(Ic, mm(I[i].mch, ch));
        Mat<double> R; rpy(rb.rots[3 * i], rb.rots[3 * i + 1], rb.rots[3 * i + 2], R);
        X[i].R = mt(R); X[i].p = Vec<double>{{rb.trans[3 * i], rb.trans[3 * i + 1], rb.trans[3 * i + 2]}};
        C[i].R = eye<double>(); C[i].p = c;
    }
    md.n = n;
    for (int i = 0; i < n; i++) {
        Xform<double> Xwj = X[i];
        for (int p = i - 1; p >= 0; p--) Xwj = x_compose(Xwj, X[p]);
        md.S_[i] = x_invapply(Xwj, S[i]);
        {   // Transform::apply(RigidInertia) with the CoM offset (spatial.cpp:221-236)
            const Xform<double>& x = C[i];
            const Mat<double> ph = skew(x.p), Rt = mt(x.R), mRp = mm(smulm(I[i].m, x.R), ph);
            md.I[i].m = I[i].m;
            md.I[i].mch = msub(mm(mm(x.R, I[i].mch), Rt), mm(mm(smulm(I[i].m, x.R), ph), Rt));
            md.I[i].Ibar = mm(msub(mm(x.R, madd(I[i].Ibar, mm(smulm(2.0, I[i].mch), ph))), mm(mRp, ph)), Rt);
        }
        Xform<double> prev{eye<double>(), zeros<double>()};
        if (i > 0) prev = C[i - 1];
        md.X[i] = x_compose(prev, x_compose(x_inverse(X[i]), x_inverse(C[i])));
        md.transI[i] = rb.armature[i]; md.damping[i] = rb.damping[i]; md.friction[i] = rb.friction[i];
    }
    md.grav.w = zeros<double>(); md.grav.v = Vec<double>{{0.0, 0.0, -rb.gravity}};
    im.n = n;
    const double lo = 1 - eps, hi = 1 + eps;
    for (int i = 0; i < n; i++) {
        for (int e = 0; e < 3; e++) { im.S_[i].w.v[e] = Interval(md.S_[i].w.v[e]); im.S_[i].v.v[e] = Interval(md.S_[i].v.v[e]); im.X[i].p.v[e] = Interval(md.X[i].p.v[e]); }
        for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) {
            im.X[i].R.m[r][k] = Interval(md.X[i].R.m[r][k]);
            im.I[i].mch.m[r][k] = Interval(md.I[i].mch.m[r][k]);
            const double val = md.I[i].Ibar.m[r][k];
            im.I[i].Ibar.m[r][k] = val >= 0 ? Interval(val * lo, val * hi) : Interval(val * hi, val * lo);
        }
        im.I[i].m = Interval(md.I[i].m * lo, md.I[i].m * hi);
        im.transI[i] = Interval(md.transI[i]); im.damping[i] = md.damping[i]; im.friction[i] = md.friction[i];
    }
    for (int e = 0; e < 3; e++) { im.grav.w.v[e] = Interval(0.0); im.grav.v.v[e] = Interval(md.grav.v.v[e]); }
}

template <class S>
void rnea(const Model<S>& md, const double* q, const double* qd, const double* qda, const double* qdd, bool friction, bool gravity, S* tau) {
    const int n = md.n;
    Twist<S> ng{zeros<S>(), zeros<S>()};
    if (gravity) ng = {vneg(md.grav.w), vneg(md.grav.v)};
    Twist<S> v[NJ], va[NJ], a[NJ], Sb[NJ]; Wrench<S> f[NJ]; Xform<S> Xbw[NJ], Xl[NJ];
    for (int i = 0; i < n; i++) {
        Xbw[i] = i ? x_compose(Xbw[i - 1], md.X[i]) : md.X[i];
        Sb[i] = x_invapply(Xbw[i], md.S_[i]);
        Xl[i] = x_compose(x_from_twist(Sb[i], -q[i]), x_inverse(md.X[i]));
        if (i == 0) {
            v[i] = tw_mul(Sb[i], qd[i]); va[i] = tw_mul(Sb[i], qda[i]);
            a[i] = tw_add(tw_add(x_apply(Xl[i], ng), tw_mul(Sb[i], qdd[i])), tw_cross(v[i], va[i]));
        } else {
            v[i] = tw_add(x_apply(Xl[i], v[i - 1]), tw_mul(Sb[i], qd[i]));
            const Twist<S> tmp = tw_mul(Sb[i], qda[i]);
            va[i] = tw_add(x_apply(Xl[i], va[i - 1]), tmp);
            a[i] = tw_add(tw_add(x_apply(Xl[i], a[i - 1]), tw_mul(Sb[i], qdd[i])), tw_cross(v[i], tmp));
        }
        Wrench<S> vIv;
        vIv.tau = vcross(va[i].w, mv(md.I[i].Ibar, v[i].w));
        vIv.tau = vadd(vIv.tau, mv(md.I[i].Ibar, vcross(va[i].w, v[i].w)));
        vIv.f = smulv(md.I[i].m, vcross(va[i].w, v[i].v));
        const Wrench<S> Ia = i_apply(md.I[i], a[i]);
        f[i] = {vadd(Ia.tau, vIv.tau), vadd(Ia.f, vIv.f)};
    }
    for (int i = n - 1; i >= 0; i--) {
        tau[i] = vdot(Sb[i].w, f[i].tau) + vdot(Sb[i].v, f[i].f);
        tau[i] = tau[i] + md.transI[i] * qdd[i];
        tau[i] = tau[i] + lift<S>(md.damping[i] * qd[i]);
        if (friction) tau[i] = tau[i] + lift<S>(md.friction[i] * ((qd[i] > 0) - (qd[i] < 0)));
        if (i > 0) { const Wrench<S> b = x_invapply(Xl[i], f[i]); f[i - 1] = {vadd(f[i - 1].tau, b.tau), vadd(f[i - 1].f, b.f)}; }
    }
}

double wrap(double x) { const double pi = 3.14159265358979323846; while (x >= pi) x -= 2 * pi; while (x < -pi) x += 2 * pi; return x; }
}  // namespace

extern "C" {
/* [u, tau, v] of RobustController::update (ARMOUR method) for one state; also the interval torque [n][2].  Returns 1 if
 * the nominal torque lies inside the interval torque (the reference throws otherwise). */
int oracle_robust_controller(const ArmourRobot* rb, double eps, const double* Kr, double alpha, double V_max, double r_thr, const double* q, const double* q_d,
                             const double* qd, const double* qd_d, const double* qd_dd, double* u, double* tau, double* v, double* tau_interval) {
    Model<double> md; Model<Interval> im;
    build(*rb, eps, md, im);
    const int n = md.n;
    double qa_d[NJ], qa_dd[NJ], r[NJ], z[NJ];
    for (int i = 0; i < n; i++) {
        const double e = wrap(qd[i] - q[i]);
        qa_d[i] = qd_d[i] + Kr[i] * e; qa_dd[i] = qd_dd[i] + Kr[i] * (qd_d[i] - q_d[i]); r[i] = (qd_d[i] - q_d[i]) + Kr[i] * e; z[i] = 0;
    }
    rnea<double>(md, q, q_d, qa_d, qa_dd, false, true, tau);
    Interval ui[NJ];
    rnea<Interval>(im, q, q_d, qa_d, qa_dd, false, true, ui);
    int ok = 1;
    double b2 = 0;
    for (int i = 0; i < n; i++) {
        if (tau[i] > ui[i].hi || tau[i] < ui[i].lo) ok = 0;
        const Interval phi = ui[i] - Interval(tau[i]);
        const double bd = std::fmax(std::fabs(phi.lo), std::fabs(phi.hi));
        b2 += bd * bd; v[i] = 0;
        if (tau_interval) { tau_interval[2 * i] = ui[i].lo; tau_interval[2 * i + 1] = ui[i].hi; }
    }
    double rn = 0;
    for (int i = 0; i < n; i++) rn += r[i] * r[i];
    rn = std::sqrt(rn);
    if (rn > r_thr) {
        Interval Mr[NJ];
        rnea<Interval>(im, q, z, z, r, false, false, Mr);
        Interval V(0.0);
        for (int i = 0; i < n; i++) V = V + (0.5 * r[i]) * Mr[i];
        const double h = -V.hi + V_max;
        const double lam = std::fmax(0.0, -alpha * h / rn + std::sqrt(b2));
        for (int i = 0; i < n; i++) v[i] = -lam * r[i] / rn;
    }
    for (int i = 0; i < n; i++) u[i] = tau[i] - v[i];
    return ok;
}
/* nominal passivity RNEA alone with caller-chosen masses / inertias scaled by (1 + s_m[i]), (1 + s_I[i]): enclosure tests */
void oracle_pass_rnea_scaled(const ArmourRobot* rb, const double* s_m, const double* s_I, const double* q, const double* qd, const double* qda, const double* qdd, int gravity, double* tau) {
    Model<double> md; Model<Interval> im;
    build(*rb, 0.0, md, im);
    for (int i = 0; i < md.n; i++) {
        md.I[i].m = md.I[i].m * (1 + s_m[i]);
        for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) md.I[i].Ibar.m[r][k] = md.I[i].Ibar.m[r][k] * (1 + s_I[i]);
    }
    rnea<double>(md, q, qd, qda, qdd, false, gravity != 0, tau);
}
}
