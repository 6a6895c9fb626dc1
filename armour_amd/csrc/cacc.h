// Constant-acceleration joint trajectory of the ARMTD comparison planner (CMP/Trajectory.h:6-15):
//     q(t) = q0 + qd0 t + ka t^2 / 2  on [0, 0.5], then a constant-deceleration stop at t = 1;   ka = k_range * k.
// Closed-form minimum / maximum of position and velocity over the whole curve and the derivative of each with respect to
// ka.  Used on the device by the joint-limit rows of the P2 kernel (ARMTD mode) and on the host by eval_f / eval_grad_f.
//
// Reference: CMP/Trajectory.cu:83-205 (returnJointStateExtremum) and :207-383 (returnJointStateExtremumGradient); the
// arithmetic below keeps the reference's operation order, the candidate/selection structure is ours.
//
// Two reference behaviours kept on purpose (results must be identical on the same inputs):
//   * the gradient rows hold d/d(ka), NOT d/dk -- the reference stores them into the Jacobian without the k_range factor
//     (CMP/Trajectory.cu:347-381);
//   * a stationary point inside (0, 0.5) replaces only the bound on the side of the acceleration's sign (:135-148).
#pragma once
#include <hip/hip_runtime.h>

namespace cacc {

struct Cand { double v, d; };  // a candidate extremum and its derivative with respect to ka

__host__ __device__ inline Cand pick_lo(Cand a, Cand b) { return a.v <= b.v ? a : b; }  // reference: `if (a <= b) a else b`
__host__ __device__ inline Cand pick_hi(Cand a, Cand b) { return a.v >= b.v ? a : b; }

struct Extrema { Cand q_min, q_max, qd_min, qd_max; };

__host__ __device__ inline Extrema joint_extrema(double q0, double qd0, double ka) {
    const double t_move = 0.5, t_total = 1.0, t_stop = t_total - t_move;
    const double q_peak = q0 + qd0 * t_move + ka * t_move * t_move * 0.5;
    const double qd_peak = qd0 + ka * t_move;
    const double qdd_stop = -qd_peak / t_stop;
    const double q_stop = q_peak + qd_peak * t_stop + 0.5 * qdd_stop * t_stop * t_stop;
    const double t_turn = -qd0 / ka;  // where the first piece is stationary (inf / nan when ka = 0: the test below is then false)
    const double d_peak = 0.5 * t_move * t_move, d_stop = 0.5 * t_move * t_move + 0.5 * t_move * t_stop;

    // first piece, position: the end points, and the turning point if it lies inside
    const Cand at0{q0, 0.0}, atp{q_peak, d_peak};
    Cand lo1 = q_peak >= q0 ? at0 : atp, hi1 = q_peak >= q0 ? atp : at0;
    if (t_turn > 0 && t_turn < t_move) {
        const Cand turn{q0 + qd0 * t_turn + 0.5 * ka * t_turn * t_turn, (0.5 * qd0 * qd0) / (ka * ka)};
        if (ka >= 0) lo1 = turn; else hi1 = turn;
    }
    // first piece, velocity (linear in t)
    const Cand v0{qd0, 0.0}, vp{qd_peak, t_move};
    const Cand vlo1 = qd_peak >= qd0 ? v0 : vp, vhi1 = qd_peak >= qd0 ? vp : v0;
    // braking piece: position is monotone (the velocity does not change sign), velocity runs from qd_peak to 0
    const Cand ats{q_stop, d_stop};
    const Cand lo2 = q_stop >= q_peak ? atp : ats, hi2 = q_stop >= q_peak ? ats : atp;
    const Cand rest{0.0, 0.0};
    const Cand vlo2 = qd_peak >= 0 ? rest : vp, vhi2 = qd_peak >= 0 ? vp : rest;

    Extrema e;
    e.q_min = pick_lo(lo1, lo2);
    e.q_max = pick_hi(hi1, hi2);
    e.qd_min = pick_lo(vlo1, vlo2);
    e.qd_max = pick_hi(vhi1, vhi2);
    return e;
}

// position at t_plan = 0.5, the point the cost pulls towards q_des (CMP/NLPclass.cu:197) and its derivative w.r.t. k
__host__ __device__ inline double q_plan(double q0, double qd0, double k_range, double k) { return q0 + qd0 * 0.5 + k_range * k * 0.125; }
__host__ __device__ inline double q_plan_dk(double k_range) { return k_range * 0.125; }

}  // namespace cacc
