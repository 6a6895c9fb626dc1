// Collision rows taken from a LIST (relevance.hip: the rows that can ever matter), one lane per listed row: the arithmetic of p2_tiles.h's
// collision_block in its order -- slice: centre + monomials in table order, interval centre (RT/PZsparse.cu:404-435), the partials summed in
// the same order from 0 (:454-472); planes in ascending order, pos before neg, strict > (RT/CollisionChecking.cu:230-299: the reference's
// serial winner, which is also what the four-wave merge of collision_block produces) -- so a listed row's g and Jacobian row are the fused
// evaluation's bit for bit (tests/test_row_relevance.py, tests/test_solve.py).  Plane entries come from the packed copy of the listed rows.
#pragma once
#include "p2_tiles.h"

namespace p2 {

struct SparseList {
    const int* rows;            // [B][Q] listed collision rows q of every problem, ascending
    const int* count;           // [B]
    const double* packed;       // problem b's block at packed + pack_off[2 b]: [live plane j][component of {Ax, Ay, Az, d, delta}][listed row i], row stride pack_off[2 b + 1]
    const long long* pack_off;
};

struct PlaneVals { double a0, a1, a2, dd, dl; };

// listed row i of problem b at the point whose power table is kp: g[row0 + q] and, WANT_J, jac[(row0 + q) * n + ..]
template <bool WANT_J>
__device__ inline void sparse_collision_row(const P2Tables& tb, const SparseList& sl, int b, int i, const KPow& kp, double* __restrict__ g, double* __restrict__ jac) {
    const int Q = tb.Q, O = tb.O, JT = tb.J * tb.T, n = tb.n;
    const int q = sl.rows[(size_t)b * Q + i];
    const int lt = q / O;
    const size_t idx = (size_t)b * JT + lt;
    const int cnt = min(tb.link_count[idx], tb.capL);
    double x[3], dx[WANT_J ? ARMOUR_MAX_FACTORS : 1][3];
    {
        double acc0 = tb.link_center[idx * 3 + 0], acc1 = tb.link_center[idx * 3 + 1], acc2 = tb.link_center[idx * 3 + 2];
#pragma unroll
        for (int kk = 0; kk < (WANT_J ? ARMOUR_MAX_FACTORS : 1); kk++) { dx[kk][0] = 0.0; dx[kk][1] = 0.0; dx[kk][2] = 0.0; }
        const uint32_t* kk_ = tb.link_keys + idx * tb.capL;
        const double* cc = tb.link_coeff + idx * tb.capL * 3;
        const int cmax = cnt > 0 ? cnt - 1 : 0;
        // two monomials' table entries requested before they are used (clamped index: a lane past its count adds nothing)
        for (int m0 = 0; m0 < cnt; m0 += 2) {
            uint32_t key[2]; double c3[2][3];
#pragma unroll
            for (int u = 0; u < 2; u++) { const int mo = min(m0 + u, cmax); key[u] = kk_[mo]; c3[u][0] = cc[mo * 3]; c3[u][1] = cc[mo * 3 + 1]; c3[u][2] = cc[mo * 3 + 2]; }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                if (m0 + u < cnt) {   // (per lane; in monomial order, column by column the sums of the fused evaluation)
#pragma unroll
                    for (int e = 0; e < 3; e++) {
                        double o8[P2_TQW];
                        mono_all<WANT_J>(kp, key[u], c3[u][e], n, o8);
                        if (e == 0) acc0 += o8[0]; else if (e == 1) acc1 += o8[0]; else acc2 += o8[0];
                        if (WANT_J) {
#pragma unroll
                            for (int kk = 0; kk < ARMOUR_MAX_FACTORS; kk++) dx[kk][e] += o8[1 + kk];
                        }
                    }
                }
            }
        }
        x[0] = interval_center(acc0, tb.link_indep[idx * 3 + 0]); x[1] = interval_center(acc1, tb.link_indep[idx * 3 + 1]); x[2] = interval_center(acc2, tb.link_indep[idx * 3 + 2]);
    }
    // the packed entries of this row: plane j of the problem's live planes (ascending), six planes' values requested, then used
    const double* src = sl.packed + sl.pack_off[2 * b] + i;
    const size_t stride = (size_t)sl.pack_off[2 * b + 1];
    const unsigned long long live = ~tb.plane_skip[b] & ((1ull << ARMOUR_NPLANES) - 1ull);
    const int nlive = __popcll(live);
    double max_elt = -100000000.0;
    double mA0 = 0.0, mA1 = 0.0, mA2 = 0.0;   // max_id defaults to plane 0 (RT/CollisionChecking.cu:262): its normal if it is live, zero if it was skipped
    bool neg = false;
    const bool plane0_live = (live & 1ull) != 0;
    for (int j0 = 0; j0 < nlive; j0 += 6) {
        PlaneVals v[6];
#pragma unroll
        for (int u = 0; u < 6; u++) {
            const int j = min(j0 + u, nlive - 1);
            v[u].a0 = src[(size_t)(j * 5 + 0) * stride]; v[u].a1 = src[(size_t)(j * 5 + 1) * stride]; v[u].a2 = src[(size_t)(j * 5 + 2) * stride];
            v[u].dd = src[(size_t)(j * 5 + 3) * stride]; v[u].dl = src[(size_t)(j * 5 + 4) * stride];
        }
        if (WANT_J && j0 == 0 && plane0_live) { mA0 = v[0].a0; mA1 = v[0].a1; mA2 = v[0].a2; }
#pragma unroll
        for (int u = 0; u < 6; u++) {
            const bool nz = j0 + u < nlive && ((v[u].a0 != 0.0) | (v[u].a1 != 0.0) | (v[u].a2 != 0.0));
            const double dot = v[u].a0 * x[0] + v[u].a1 * x[1] + v[u].a2 * x[2];
            const double pos_res = nz ? dot - (v[u].dd + v[u].dl) : -100000000.0;
            const double neg_res = nz ? -dot - (-v[u].dd + v[u].dl) : -100000000.0;
            const bool c1 = pos_res > max_elt;
            max_elt = c1 ? pos_res : max_elt;
            const bool c2 = neg_res > max_elt;
            max_elt = c2 ? neg_res : max_elt;
            if (WANT_J) {
                const bool hit = c1 | c2;
                mA0 = hit ? v[u].a0 : mA0; mA1 = hit ? v[u].a1 : mA1; mA2 = hit ? v[u].a2 : mA2;
                neg = c2 ? true : (c1 ? false : neg);
            }
        }
    }
    g[(size_t)tb.row0 + q] = -max_elt;
    if (WANT_J) {
        double* jr = jac + ((size_t)tb.row0 + q) * n;
#pragma unroll
        for (int kk = 0; kk < ARMOUR_MAX_FACTORS; kk++) {
            if (kk < n) {
                const double dot = mA0 * dx[kk][0] + mA1 * dx[kk][1] + mA2 * dx[kk][2];
                jr[kk] = neg ? dot : -dot;
            }
        }
    }
}

// Torque row `row` (= t * n + j) of problem b on ONE thread: the sums of p2_tiles.h's torque_block in its order -- every monomial's value and
// partials by mono_all, the value summed from the centre in table order and centred on the interval (RT/PZsparse.cu:404-435), each partial summed
// from 0 in table order (:454-472) -- so g and the Jacobian row are the fused evaluation's bit for bit.  For the culled solver: 256 listed rows
// per pass instead of 8 rows per tile of 32 monomial lanes.
__device__ inline void sparse_torque_row(const P2Tables& tb, int b, int row, int strideT, const KPow& kp, double* __restrict__ g, double* __restrict__ jac) {
    const int n = tb.n, T = tb.T;
    const int t = row / n, j = row - t * n;
    const size_t idx = ((size_t)b * n + j) * T + t;
    const int cnt = min(tb.tq_count[idx], strideT);
    double cen = tb.tq_center[idx];
    double gr[ARMOUR_MAX_FACTORS];
#pragma unroll
    for (int kk = 0; kk < ARMOUR_MAX_FACTORS; kk++) gr[kk] = 0.0;
    const uint32_t* keys = tb.tq_keys + idx * tb.capT;
    const double* co = tb.tq_coeff + idx * tb.capT;
    const int cmax = cnt > 0 ? cnt - 1 : 0;
    for (int m0 = 0; m0 < cnt; m0 += 4) {   // four monomials' table entries requested before they are used
        uint32_t key[4]; double c[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int mo = min(m0 + u, cmax); key[u] = keys[mo]; c[u] = co[mo]; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (m0 + u < cnt) {
                double o8[P2_TQW];
                mono_all<true>(kp, key[u], c[u], n, o8);
                cen += o8[0];
#pragma unroll
                for (int kk = 0; kk < ARMOUR_MAX_FACTORS; kk++) gr[kk] += o8[1 + kk];
            }
        }
    }
    g[row] = interval_center(cen, tb.tq_indep[idx]);
#pragma unroll
    for (int kk = 0; kk < ARMOUR_MAX_FACTORS; kk++) if (kk < n) jac[(size_t)row * n + kk] = gr[kk];
}

}  // namespace p2
