/*
 * oracle/interval.hpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Minimal outward-rounded interval type standing in for the reference's
 *   bn::interval<double, policies<save_state<rounded_transc_std<double>>, checking_base<double>>>
 * (RT/Headers.h:26-36; Boost.Interval 1.71, not vendored in the reference and absent here).
 * Boost switches the FPU to directed rounding; here every endpoint is computed in
 * round-to-nearest and then moved one ulp outward (nextafter), which encloses Boost's result
 * and differs from it by at most a few ulp.  Only the operations the hot path uses exist:
 * call sites RT/Trajectory.cu:97-127 (cos/sin Lagrange remainder) and
 * RT/armour_main.cu:179-190 (norm of the disturbance interval).
 */
#ifndef ORACLE_INTERVAL_HPP
#define ORACLE_INTERVAL_HPP

#include <algorithm>
#include <cmath>
#include <limits>

namespace oracle {

struct Interval {
    double lo = 0, hi = 0;
    Interval() {}
    Interval(double v) : lo(v), hi(v) {}
    Interval(double l, double h) : lo(l), hi(h) {}
};

inline double dn(double x) { return std::nextafter(x, -std::numeric_limits<double>::infinity()); }
inline double up(double x) { return std::nextafter(x, std::numeric_limits<double>::infinity()); }
inline Interval outward(double l, double h) { return Interval(dn(l), up(h)); }

inline double getCenter(const Interval& a) { return (a.lo + a.hi) * 0.5; } /* RT/PZsparse.cu:10-12 */
inline double getRadius(const Interval& a) { return (a.hi - a.lo) * 0.5; } /* RT/PZsparse.cu:14-16 */

inline Interval operator-(const Interval& a) { return Interval(-a.hi, -a.lo); }
inline Interval operator+(const Interval& a, const Interval& b) { return outward(a.lo + b.lo, a.hi + b.hi); }
inline Interval operator+(double a, const Interval& b) { return outward(a + b.lo, a + b.hi); }
inline Interval operator+(const Interval& a, double b) { return b + a; }
inline Interval operator-(const Interval& a, const Interval& b) { return outward(a.lo - b.hi, a.hi - b.lo); }
inline Interval operator-(const Interval& a, double b) { return outward(a.lo - b, a.hi - b); }
inline Interval operator*(const Interval& a, const Interval& b) {
    const double p[4] = {a.lo * b.lo, a.lo * b.hi, a.hi * b.lo, a.hi * b.hi};
    return outward(*std::min_element(p, p + 4), *std::max_element(p, p + 4));
}
inline Interval operator*(double a, const Interval& b) { return Interval(a) * b; }
inline Interval operator*(const Interval& a, double b) { return a * Interval(b); }

/* pow(I, 2): even power, [0, max^2] when the interval straddles zero (Boost pow for even n) */
inline Interval sqr(const Interval& a) {
    if (a.lo >= 0) return outward(a.lo * a.lo, a.hi * a.hi);
    if (a.hi <= 0) return outward(a.hi * a.hi, a.lo * a.lo);
    const double m = std::max(-a.lo, a.hi);
    return Interval(0.0, up(m * m));
}

/* sqrt: Boost clamps a negative lower bound to 0 (only .upper() is consumed, armour_main.cu:185-188) */
inline Interval isqrt(const Interval& a) {
    const double l = a.lo <= 0 ? 0.0 : dn(std::sqrt(a.lo));
    return Interval(l, up(std::sqrt(a.hi)));
}

/* cos over an interval: extrema at multiples of pi */
inline Interval icos(const Interval& a) {
    const double pi = 3.14159265358979323846;
    if (a.hi - a.lo >= 2 * pi) return Interval(-1, 1);
    double lo = std::min(std::cos(a.lo), std::cos(a.hi));
    double hi = std::max(std::cos(a.lo), std::cos(a.hi));
    /* does [a.lo, a.hi] contain 2*pi*m (max) or pi + 2*pi*m (min)?  conservative by 1e-15 */
    const double kmax = std::ceil((a.lo - 1e-15) / (2 * pi));
    if (kmax * 2 * pi <= a.hi + 1e-15) hi = 1.0;
    const double kmin = std::ceil((a.lo - pi - 1e-15) / (2 * pi));
    if (pi + kmin * 2 * pi <= a.hi + 1e-15) lo = -1.0;
    return Interval(std::max(-1.0, dn(lo)), std::min(1.0, up(hi)));
}

/* sin(x) = cos(x - pi/2) (as Boost does) */
inline Interval isin(const Interval& a) {
    const double pi = 3.14159265358979323846;
    if (a.hi - a.lo >= 2 * pi) return Interval(-1, 1);
    double lo = std::min(std::sin(a.lo), std::sin(a.hi));
    double hi = std::max(std::sin(a.lo), std::sin(a.hi));
    const double kmax = std::ceil((a.lo - pi / 2 - 1e-15) / (2 * pi));
    if (pi / 2 + kmax * 2 * pi <= a.hi + 1e-15) hi = 1.0;
    const double kmin = std::ceil((a.lo + pi / 2 - 1e-15) / (2 * pi));
    if (-pi / 2 + kmin * 2 * pi <= a.hi + 1e-15) lo = -1.0;
    return Interval(std::max(-1.0, dn(lo)), std::min(1.0, up(hi)));
}

}  // namespace oracle
#endif
