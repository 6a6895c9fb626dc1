// Spatial-vector passivity RNEA and the ARMOUR robust input of the reference's tracking controller, for one state.
//
// Reference: kinova_src/kinova_simulator_interfaces/kinova_robust_controllers_mex/ (MEXC/):
//   spatial.cpp / spatial_interval.cpp   Twist, Wrench, RigidInertia, Transform and their interval twins
//   rnea.cpp:6-96 / :98-185              passRNEA / passRNEA_Int
//   robot_models.cpp:124-160, :176-255   conversion of the model file to CoM frames; the interval model (mass and
//                                        inertia widened by +-eps, everything else a point interval)
//   robust_controller.cpp:63-168         RobustController::update, ARMOUR method
// The reference writes every class twice (double and Boost interval); here the scalar type is a template parameter
// and both instantiations follow the same operation order (sums of products accumulate k = 0, 1, 2 as Eigen's
// fixed-size products do).  Intervals round outward by one ulp per operation (the value nextafter gives), as the reach-set code does.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/armour_types.h"

namespace ctl {

#define CTL_HD __host__ __device__ inline
// The halves of the RNEA go into their caller whole, with their getter / putter.  (As functions of their own they reached the kernel's frame --
// the kinematics store, the lambdas' captures -- through generic pointers, and both kernels built that way ended in a memory-aperture violation
// on the MI355X; inlined, no generic pointer into a frame is left.)
// ROOT CAUSE (round 5, profiles/r05_codegen_hazards.txt): as functions of their own the interval rnea_dynamics is > 128 KB of code, the compiler
// relaxes four of its branches through s[30:31] WITHOUT saving the return address held there, and the function returns into its own middle
// (pz_wave.h PZ_KEEP_RETURN_ADDRESS; tools/check_long_branches.py reports exactly these two functions in a -DCTL_NO_FLATTEN object).  With the
// return address declared clobbered at the entry (CTL_KEEP_RETURN_ADDRESS below) the non-inlined build runs and gives the same outputs.
#ifdef CTL_NO_FLATTEN   // (development: the halves as functions of their own; add -DCTL_NO_RETURN_ADDRESS_GUARD for the faulting form of round 3)
#define CTL_FLATTEN
#else
#define CTL_FLATTEN __attribute__((always_inline))
#endif
#if defined(__HIP_DEVICE_COMPILE__) && defined(CTL_NO_FLATTEN) && !defined(CTL_NO_RETURN_ADDRESS_GUARD)
#define CTL_KEEP_RETURN_ADDRESS() asm volatile("; return address kept out of s[30:31]" ::: "s30", "s31")
#else
#define CTL_KEEP_RETURN_ADDRESS()
#endif

struct Itv { double lo, hi; };
// nextafter(x, -+inf): one step of the bit pattern for finite x -- the library call is ~10x the instructions and the
// interval RNEA makes ~10^4 of them per state (a single-state call went from 1.5 ms to the time below with this)
CTL_HD double step_ulp(double x, bool toward_plus) {
    if (!(fabs(x) < INFINITY)) return nextafter(x, toward_plus ? INFINITY : -INFINITY);  // inf / nan: the library's answer
    if (x == 0.0) return toward_plus ? 4.9406564584124654e-324 : -4.9406564584124654e-324;
    const long long b = __builtin_bit_cast(long long, x);
    return __builtin_bit_cast(double, b + (((b >= 0) == toward_plus) ? 1LL : -1LL));  // magnitude grows iff the step points away from zero
}
CTL_HD double dn(double x) { return step_ulp(x, false); }
CTL_HD double up(double x) { return step_ulp(x, true); }
CTL_HD Itv outw(double l, double h) { return Itv{dn(l), up(h)}; }
CTL_HD Itv operator-(Itv a) { return Itv{-a.hi, -a.lo}; }
CTL_HD Itv operator+(Itv a, Itv b) { return outw(a.lo + b.lo, a.hi + b.hi); }
CTL_HD Itv operator-(Itv a, Itv b) { return outw(a.lo - b.hi, a.hi - b.lo); }
CTL_HD Itv operator*(Itv a, Itv b) {
    const double p0 = a.lo * b.lo, p1 = a.lo * b.hi, p2 = a.hi * b.lo, p3 = a.hi * b.hi;
    return outw(fmin(fmin(p0, p1), fmin(p2, p3)), fmax(fmax(p0, p1), fmax(p2, p3)));
}
CTL_HD Itv operator*(Itv a, double b) { return a * Itv{b, b}; }
CTL_HD Itv operator*(double a, Itv b) { return Itv{a, a} * b; }

// scalar traits: lift a double, zero, one
template <class S> struct Sc;
template <> struct Sc<double> { CTL_HD static double of(double x) { return x; } };
template <> struct Sc<Itv> { CTL_HD static Itv of(double x) { return Itv{x, x}; } };

template <class S> struct V3 { S x[3]; };
template <class S> struct M3 { S a[9]; };  // row-major

template <class S> CTL_HD V3<S> vzero() { V3<S> r; for (int i = 0; i < 3; i++) r.x[i] = Sc<S>::of(0.0); return r; }
template <class S> CTL_HD M3<S> mident() { M3<S> r; for (int i = 0; i < 9; i++) r.a[i] = Sc<S>::of((i % 4 == 0) ? 1.0 : 0.0); return r; }
template <class S> CTL_HD V3<S> operator+(const V3<S>& a, const V3<S>& b) { V3<S> r; for (int i = 0; i < 3; i++) r.x[i] = a.x[i] + b.x[i]; return r; }
template <class S> CTL_HD V3<S> operator-(const V3<S>& a, const V3<S>& b) { V3<S> r; for (int i = 0; i < 3; i++) r.x[i] = a.x[i] - b.x[i]; return r; }
template <class S> CTL_HD V3<S> operator-(const V3<S>& a) { V3<S> r; for (int i = 0; i < 3; i++) r.x[i] = -a.x[i]; return r; }
template <class S, class K> CTL_HD V3<S> scale(const V3<S>& a, K s) { V3<S> r; for (int i = 0; i < 3; i++) r.x[i] = a.x[i] * s; return r; }
template <class S, class K> CTL_HD V3<S> lscale(K s, const V3<S>& a) { V3<S> r; for (int i = 0; i < 3; i++) r.x[i] = s * a.x[i]; return r; }
template <class S> CTL_HD V3<S> cross(const V3<S>& a, const V3<S>& b) {
    V3<S> r;
    r.x[0] = a.x[1] * b.x[2] - a.x[2] * b.x[1];
    r.x[1] = a.x[2] * b.x[0] - a.x[0] * b.x[2];
    r.x[2] = a.x[0] * b.x[1] - a.x[1] * b.x[0];
    return r;
}
template <class S> CTL_HD S dot(const V3<S>& a, const V3<S>& b) { return (a.x[0] * b.x[0] + a.x[1] * b.x[1]) + a.x[2] * b.x[2]; }
template <class S> CTL_HD V3<S> operator*(const M3<S>& m, const V3<S>& v) {
    V3<S> r;
    for (int i = 0; i < 3; i++) r.x[i] = (m.a[3 * i] * v.x[0] + m.a[3 * i + 1] * v.x[1]) + m.a[3 * i + 2] * v.x[2];
    return r;
}
template <class S> CTL_HD M3<S> operator*(const M3<S>& a, const M3<S>& b) {
    M3<S> r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.a[3 * i + j] = (a.a[3 * i] * b.a[j] + a.a[3 * i + 1] * b.a[3 + j]) + a.a[3 * i + 2] * b.a[6 + j];
    return r;
}
template <class S> CTL_HD M3<S> operator+(const M3<S>& a, const M3<S>& b) { M3<S> r; for (int i = 0; i < 9; i++) r.a[i] = a.a[i] + b.a[i]; return r; }
template <class S> CTL_HD M3<S> operator-(const M3<S>& a, const M3<S>& b) { M3<S> r; for (int i = 0; i < 9; i++) r.a[i] = a.a[i] - b.a[i]; return r; }
template <class S, class K> CTL_HD M3<S> scale(const M3<S>& a, K s) { M3<S> r; for (int i = 0; i < 9; i++) r.a[i] = a.a[i] * s; return r; }
template <class S, class K> CTL_HD M3<S> lscale(K s, const M3<S>& a) { M3<S> r; for (int i = 0; i < 9; i++) r.a[i] = s * a.a[i]; return r; }
template <class S> CTL_HD M3<S> tr(const M3<S>& a) { M3<S> r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.a[3 * i + j] = a.a[3 * j + i]; return r; }
template <class S> CTL_HD M3<S> hat(const V3<S>& w) {  // spatial.cpp:46-48
    M3<S> r;
    const S z = Sc<S>::of(0.0);
    r.a[0] = z; r.a[1] = -w.x[2]; r.a[2] = w.x[1];
    r.a[3] = w.x[2]; r.a[4] = z; r.a[5] = -w.x[0];
    r.a[6] = -w.x[1]; r.a[7] = w.x[0]; r.a[8] = z;
    return r;
}

template <class S> struct Tw { V3<S> w, v; };   // Twist (w_hat is hat(w), recomputed where the reference stores it)
template <class S> struct Wr { V3<S> tau, f; }; // Wrench
template <class S> struct Xf { M3<S> R; V3<S> p; };           // Transform
template <class S> struct Ri { S m; M3<S> I_bar, m_c_hat; };  // RigidInertia

template <class S> CTL_HD Tw<S> operator+(const Tw<S>& a, const Tw<S>& b) { return Tw<S>{a.w + b.w, a.v + b.v}; }
template <class S> CTL_HD Tw<S> operator-(const Tw<S>& a) { return Tw<S>{-a.w, -a.v}; }
template <class S, class K> CTL_HD Tw<S> scale(const Tw<S>& a, K s) { return Tw<S>{scale(a.w, s), scale(a.v, s)}; }
template <class S> CTL_HD Wr<S> operator+(const Wr<S>& a, const Wr<S>& b) { return Wr<S>{a.tau + b.tau, a.f + b.f}; }
template <class S> CTL_HD S dot(const Tw<S>& z, const Wr<S>& f) { return dot(z.w, f.tau) + dot(z.v, f.f); }  // spatial.cpp:76-79
template <class S> CTL_HD Tw<S> cross(const Tw<S>& z, const Tw<S>& z2) {                                      // :81-85
    const M3<S> wh = hat(z.w);
    return Tw<S>{wh * z2.w, wh * z2.v + cross(z.v, z2.w)};
}
template <class S> CTL_HD Wr<S> apply(const Ri<S>& I, const Tw<S>& z) {  // :144-148
    return Wr<S>{I.I_bar * z.w + I.m_c_hat * z.v, lscale(I.m, z.v) - I.m_c_hat * z.w};
}
template <class S> CTL_HD Xf<S> xf_identity() { return Xf<S>{mident<S>(), vzero<S>()}; }
template <class S> CTL_HD Xf<S> xf_joint(const Tw<S>& z, double theta) {  // Transform(Twist, theta), :157-172
    const M3<S> I = mident<S>(), wh = hat(z.w);
    Xf<S> x;
    x.R = (I + scale(wh, sin(theta))) + lscale(1 - cos(theta), wh) * wh;
    const V3<S> p = ((I - x.R) * wh) * z.v;
    x.p = -(tr(x.R) * p);
    return x;
}
template <class S> CTL_HD Tw<S> apply(const Xf<S>& x, const Tw<S>& z) { return Tw<S>{x.R * z.w, x.R * (z.v - cross(x.p, z.w))}; }           // :189-193
template <class S> CTL_HD Tw<S> invapply(const Xf<S>& x, const Tw<S>& z) { const V3<S> nw = tr(x.R) * z.w; return Tw<S>{nw, tr(x.R) * z.v + cross(x.p, nw)}; }  // :197-201
template <class S> CTL_HD Wr<S> invapply(const Xf<S>& x, const Wr<S>& w) {                                                                // :211-215
    const M3<S> Rt = tr(x.R);
    return Wr<S>{Rt * w.tau + cross(x.p, Rt * w.f), Rt * w.f};
}
template <class S> CTL_HD Xf<S> apply(const Xf<S>& x, const Xf<S>& x2) { return Xf<S>{x.R * x2.R, x2.p + tr(x2.R) * x.p}; }  // :238-245
template <class S> CTL_HD Xf<S> inverse(const Xf<S>& x) { return Xf<S>{tr(x.R), -(x.R * x.p)}; }                              // :247-252
// Transform::apply(RigidInertia), spatial.cpp:221-236 (model conversion only)
CTL_HD Ri<double> apply(const Xf<double>& x, const Ri<double>& I) {
    const M3<double> ph = hat(x.p), Rt = tr(x.R);
    const M3<double> mRp = lscale(I.m, x.R) * ph;
    Ri<double> n;
    n.m = I.m;
    n.m_c_hat = (x.R * I.m_c_hat) * Rt - (lscale(I.m, x.R) * ph) * Rt;
    n.I_bar = (x.R * (I.I_bar + lscale(2.0, I.m_c_hat) * ph) - mRp * ph) * Rt;
    return n;
}

// the model in CoM frames (robot_models.cpp:124-160) with scalar type S
template <class S>
struct Model {
    int n;
    Tw<S> S_[ARMOUR_MAX_FACTORS];
    Ri<S> I[ARMOUR_MAX_FACTORS];
    Xf<S> XTree[ARMOUR_MAX_FACTORS];
    S transI[ARMOUR_MAX_FACTORS];
    double damping[ARMOUR_MAX_FACTORS], friction[ARMOUR_MAX_FACTORS];
    Tw<S> gravity;
};

// passRNEA / passRNEA_Int (rnea.cpp:6-96, :98-185), serial chain (lam[i] = i - 1), in two halves:
//   kinematics -- per joint the twist axis in body coordinates Sb_i and the transform from the parent's frame Xli_i: functions of q and the
//                 model alone (rnea.cpp:22-31 / :114-123), so the two interval passes of a controller update share one evaluation;
//   dynamics   -- the velocity / acceleration / wrench recursions and the backward projection (:33-95 / :125-184), which read (Xli_i, Sb_i)
//                 joint by joint through `get`: from the caller's arrays, or from LDS as another wave produces them (controller.hip).
// Operation for operation the code of the one-function form: results are unchanged to the bit.
template <class S, class Put>
CTL_HD CTL_FLATTEN void rnea_kinematics(const Model<S>& md, const double* q, Put put) {
    CTL_KEEP_RETURN_ADDRESS();
    Xf<S> Xbw = md.XTree[0];
    for (int i = 0; i < md.n; i++) {
        if (i > 0) Xbw = apply(Xbw, md.XTree[i]);
        const Tw<S> Sb = invapply(Xbw, md.S_[i]);
        const Xf<S> Xli = apply(xf_joint(Sb, -q[i]), inverse(md.XTree[i]));
        put(i, Xli, Sb);
    }
}
template <class S, class Get>
CTL_HD CTL_FLATTEN void rnea_dynamics(const Model<S>& md, Get get, const double* qd, const double* qda, const double* qdd, bool apply_friction, bool apply_gravity,
                          S* tau) {
    CTL_KEEP_RETURN_ADDRESS();
    const int n = md.n;
    Tw<S> neg_g{vzero<S>(), vzero<S>()};
    if (apply_gravity) neg_g = -md.gravity;
    Tw<S> v, va, a;
    Wr<S> f[ARMOUR_MAX_FACTORS];
    for (int i = 0; i < n; i++) {
        Xf<S> Xli; Tw<S> Sb;
        get(i, Xli, Sb);
        if (i == 0) {
            v = scale(Sb, qd[i]);
            va = scale(Sb, qda[i]);
            a = (apply(Xli, neg_g) + scale(Sb, qdd[i])) + cross(v, va);
        } else {
            v = apply(Xli, v) + scale(Sb, qd[i]);
            const Tw<S> temp = scale(Sb, qda[i]);
            va = apply(Xli, va) + temp;
            a = (apply(Xli, a) + scale(Sb, qdd[i])) + cross(v, temp);
        }
        Wr<S> vIv;  // "v x Iv Jon's way"
        vIv.tau = cross(va.w, md.I[i].I_bar * v.w);
        vIv.tau = vIv.tau + md.I[i].I_bar * cross(va.w, v.w);
        vIv.f = lscale(md.I[i].m, cross(va.w, v.v));
        f[i] = apply(md.I[i], a) + vIv;
    }
    for (int i = n - 1; i >= 0; i--) {
        Xf<S> Xli; Tw<S> Sb;
        get(i, Xli, Sb);
        tau[i] = dot(Sb, f[i]);
        tau[i] = tau[i] + md.transI[i] * qdd[i];
        tau[i] = tau[i] + Sc<S>::of(md.damping[i] * qd[i]);
        if (apply_friction) tau[i] = tau[i] + Sc<S>::of(md.friction[i] * ((qd[i] > 0) - (qd[i] < 0)));
        if (i > 0) f[i - 1] = f[i - 1] + invapply(Xli, f[i]);
    }
}
template <class S> struct KinStore {   // the kinematics of one state in the caller's own memory
    Xf<S> Xli[ARMOUR_MAX_FACTORS];
    Tw<S> Sb[ARMOUR_MAX_FACTORS];
};
template <class S>
CTL_HD CTL_FLATTEN void rnea_kinematics(const Model<S>& md, const double* q, KinStore<S>& k) {
    CTL_KEEP_RETURN_ADDRESS();
    rnea_kinematics(md, q, [&](int i, const Xf<S>& Xli, const Tw<S>& Sb) { k.Xli[i] = Xli; k.Sb[i] = Sb; });
}
template <class S>
CTL_HD CTL_FLATTEN void rnea_dynamics(const Model<S>& md, const KinStore<S>& k, const double* qd, const double* qda, const double* qdd, bool apply_friction,
                          bool apply_gravity, S* tau) {
    CTL_KEEP_RETURN_ADDRESS();
    rnea_dynamics(md, [&](int i, Xf<S>& Xli, Tw<S>& Sb) { Xli = k.Xli[i]; Sb = k.Sb[i]; }, qd, qda, qdd, apply_friction, apply_gravity, tau);
}
template <class S>
CTL_HD CTL_FLATTEN void pass_rnea(const Model<S>& md, const double* q, const double* qd, const double* qda, const double* qdd, bool apply_friction,
                      bool apply_gravity, S* tau) {
    CTL_KEEP_RETURN_ADDRESS();
    KinStore<S> k;
    rnea_kinematics(md, q, k);
    rnea_dynamics(md, k, qd, qda, qdd, apply_friction, apply_gravity, tau);
}

CTL_HD double clamp_angle(double x) {  // robust_controller.hpp:11-16
    const double pi = 3.14159265358979323846;
    double r = x;
    while (r >= pi) r -= 2 * pi;
    while (r < -pi) r += 2 * pi;
    return r;
}

// RobustController::update, ARMOUR method (robust_controller.cpp:63-168), in three pieces so that the three RNEA passes between
// `prepare` and `combine` can run on different waves (controller.hip, the latency kernel) -- one thread running all of it is robust_update.
// prepare: modified reference velocity / acceleration and the tracking error r (:70-83); returns |r|
CTL_HD CTL_FLATTEN double robust_prepare(int n, const double* Kr, const double* q, const double* q_d, const double* qd, const double* qd_d, const double* qd_dd,
                             double* qa_d, double* qa_dd, double* r) {
    CTL_KEEP_RETURN_ADDRESS();
    for (int i = 0; i < n; i++) {
        const double q_diff = clamp_angle(qd[i] - q[i]);
        qa_d[i] = qd_d[i] + Kr[i] * q_diff;
        qa_dd[i] = qd_dd[i] + Kr[i] * (qd_d[i] - q_d[i]);
        r[i] = (qd_d[i] - q_d[i]) + Kr[i] * q_diff;
    }
    double r_norm = 0.0;
    for (int i = 0; i < n; i++) r_norm += r[i] * r[i];
    return sqrt(r_norm);
}
// combine: the robust input from the nominal torque, the interval torque and (when |r| is above the threshold) the interval M r (:88-166).
// Returns false if the nominal torque leaves the interval torque (the reference throws).  u = tau = u_nominal - v.
CTL_HD CTL_FLATTEN bool robust_combine(int n, double alpha, double V_max, double r_norm_threshold, const double* r, double r_norm, const double* u_nominal,
                           const Itv* u_int, const Itv* Mr, double* u, double* v_out) {
    CTL_KEEP_RETURN_ADDRESS();
    bool ok = true;
    double bound_sq = 0.0;
    for (int i = 0; i < n; i++) {
        if (u_nominal[i] > u_int[i].hi || u_nominal[i] < u_int[i].lo) ok = false;
        const Itv phi = u_int[i] - Itv{u_nominal[i], u_nominal[i]};
        const double bnd = fmax(fabs(phi.lo), fabs(phi.hi));
        bound_sq += bnd * bnd;
        v_out[i] = 0.0;
    }
    if (r_norm > r_norm_threshold) {
        Itv V{0.0, 0.0};
        for (int i = 0; i < n; i++) V = V + (0.5 * r[i]) * Mr[i];
        const double h = -V.hi + V_max;
        const double lambda = fmax(0.0, -alpha * h / r_norm + sqrt(bound_sq));
        for (int i = 0; i < n; i++) v_out[i] = -lambda * r[i] / r_norm;
    }
    for (int i = 0; i < n; i++) u[i] = u_nominal[i] - v_out[i];
    return ok;
}
CTL_HD CTL_FLATTEN bool robust_update(const Model<double>& md, const Model<Itv>& imd, const double* Kr, double alpha, double V_max, double r_norm_threshold,
                          const double* q, const double* q_d, const double* qd, const double* qd_d, const double* qd_dd, double* u, double* u_nominal,
                          double* v_out) {
    CTL_KEEP_RETURN_ADDRESS();
    const int n = md.n;
    double qa_d[ARMOUR_MAX_FACTORS], qa_dd[ARMOUR_MAX_FACTORS], r[ARMOUR_MAX_FACTORS], zero[ARMOUR_MAX_FACTORS];
    const double r_norm = robust_prepare(n, Kr, q, q_d, qd, qd_d, qd_dd, qa_d, qa_dd, r);
    for (int i = 0; i < n; i++) zero[i] = 0.0;
    pass_rnea<double>(md, q, q_d, qa_d, qa_dd, false, true, u_nominal);
    Itv u_int[ARMOUR_MAX_FACTORS], Mr[ARMOUR_MAX_FACTORS];
    KinStore<Itv> k;   // (both interval passes are at the same q: one kinematics)
    rnea_kinematics(imd, q, k);
    rnea_dynamics(imd, k, q_d, qa_d, qa_dd, false, true, u_int);
    if (r_norm > r_norm_threshold) rnea_dynamics(imd, k, zero, zero, r, false, false, Mr);
    return robust_combine(n, alpha, V_max, r_norm_threshold, r, r_norm, u_nominal, u_int, Mr, u, v_out);
}

}  // namespace ctl
