// P2: the fused eval_g + eval_jac_g kernel (gfx950).
//
// Replaces, per IPOPT iterate, the reference's 2 x { OpenMP slice loop over 128 steps + 7 kernel launches
// + 8-16 blocking copies } (RT/NLPclass.cu:272-396, RT/CollisionChecking.cu:90-134,230-299,
// RT/PZsparse.cu:404-555, RT/Trajectory.cu:256-540) by ONE launch over all B problems:
//
//   blockIdx.y = problem b;  blockIdx.x selects the role:
//     [0, nbc)          collision blocks: 256 consecutive rows q = (l*T+t)*O + o each
//     [nbc, nbc+nbt)    torque blocks: 32 (t,j) rows x 8 outputs (value + 7 partials) each
//     nbc+nbt           joint position / velocity limit rows (4n rows, closed form)
//
// Collision block: the (l,t) link PZs the block's rows touch are sliced first (value + gradient,
// RT/PZsparse.cu:404-435,477-516) into LDS; then every lane streams its row's 36 half-spaces from the
// plane table planes[b][c][p][q] (q fastest: each load is one 512-B coalesced request per wave), keeps the
// running arg-max with the reference's scan order (pos_p before neg_p, strict >) and finally dots the
// winning normal with the 7 centre derivatives.  Bound: HBM -- 1440 B of table per row against ~450 flop.
// g rows are written coalesced; Jacobian rows are staged through LDS so the block writes 256*7 contiguous
// doubles.
//
// Arithmetic order follows the CPU statement of the reference exactly (products of k-powers in factor
// order, sums in monomial order), so with identical tables the outputs agree with the oracle to the ulp of
// pow() vs. repeated multiplication; FMA contraction is disabled for this file (see Makefile).
#include "bezier.h"
#include "common.h"

#define P2_BLOCK 256
#define P2_TQ_ROWS 32

namespace {

// k-power tables in LDS: pw[j][d] = k_j^d (d = 0..3); df[j][d] = d * k_j^(d-1)
struct KPow {
    double pw[ARMOUR_MAX_FACTORS][4];
    double df[ARMOUR_MAX_FACTORS][4];
};

__device__ inline void fill_kpow(KPow& kp, const double* k, int n) {
    const int j = threadIdx.x;
    if (j < n) {
        const double x = k[j];
        kp.pw[j][0] = 1.0; kp.pw[j][1] = x; kp.pw[j][2] = x * x; kp.pw[j][3] = x * x * x;
        kp.df[j][0] = 0.0; kp.df[j][1] = 1.0; kp.df[j][2] = 2.0 * x; kp.df[j][3] = 3.0 * (x * x);
    }
}

// value of coeff * prod_j k_j^{d_j}, multiplying in factor order (RT/PZsparse.cu:416-418)
__device__ inline double mono_value(const KPow& kp, uint32_t key, double c, int n) {
    double v = c;
    for (int j = 0; j < n; j++) {
        const uint32_t d = (key >> (2 * j)) & 3u;
        if (d) v *= kp.pw[j][d];
    }
    return v;
}
// d/dk_kk of the same monomial (RT/PZsparse.cu:454-468); caller guarantees degree(kk) > 0
__device__ inline double mono_grad(const KPow& kp, uint32_t key, double c, int n, int kk) {
    double v = c;
    for (int j = 0; j < n; j++) {
        const uint32_t d = (key >> (2 * j)) & 3u;
        if (j == kk) v *= kp.df[j][d];
        else if (d) v *= kp.pw[j][d];
    }
    return v;
}
// centre of Interval(c - r, c + r) as getCenter computes it (RT/PZsparse.cu:10-12,427-432)
__device__ inline double interval_center(double c, double r) {
    const double lo = c - r, hi = c + r;
    return (lo + hi) * 0.5;
}

template <bool WANT_G, bool WANT_J>
__global__ __launch_bounds__(P2_BLOCK) void armour_p2_eval_kernel(P2Tables tb, const double* __restrict__ k_all,
                                                                  double* __restrict__ g_all, double* __restrict__ jac_all,
                                                                  int nbc, int nbt, int max_pairs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    KPow& kp = *reinterpret_cast<KPow*>(smem_raw);
    double* sx = reinterpret_cast<double*>(smem_raw + sizeof(KPow));  // [max_pairs][24]: x[3], dx[7][3]
    double* sj = sx + (size_t)max_pairs * 24;                          // [256][7] Jacobian staging

    const int b = blockIdx.y;
    const int n = tb.n, T = tb.T, O = tb.O, Q = tb.Q, m = tb.m;
    const int tid = threadIdx.x;
    const double* k = k_all + (size_t)b * n;
    double* g = WANT_G ? g_all + (size_t)b * m : nullptr;
    double* jac = WANT_J ? jac_all + (size_t)b * m * n : nullptr;

    fill_kpow(kp, k, n);
    __syncthreads();

    const int role = blockIdx.x;
    if (role < nbc) {
        // ------------------------------------------------------------------ collision rows
        const int q_begin = role * P2_BLOCK;
        const int q_end = min(Q, q_begin + P2_BLOCK);
        const int lt_first = q_begin / O;
        const int npairs = (q_end - 1) / O - lt_first + 1;
        // slice link PZs: task = (pair, axis)
        for (int task = tid; task < npairs * 3; task += P2_BLOCK) {
            const int pair = task / 3, e = task - pair * 3;
            const size_t idx = (size_t)b * tb.J * T + (lt_first + pair);
            const int cnt = tb.link_count[idx];
            const uint32_t* keys = tb.link_keys + idx * tb.capL;
            const double* co = tb.link_coeff + idx * tb.capL * 3;
            double cen = tb.link_center[idx * 3 + e];
            double gr[ARMOUR_MAX_FACTORS];
            for (int kk = 0; kk < ARMOUR_MAX_FACTORS; kk++) gr[kk] = 0.0;
            for (int mo = 0; mo < cnt; mo++) {
                const uint32_t key = keys[mo];
                const double c = co[mo * 3 + e];
                cen += mono_value(kp, key, c, n);
                if (WANT_J) {
                    for (int kk = 0; kk < n; kk++)
                        if ((key >> (2 * kk)) & 3u) gr[kk] += mono_grad(kp, key, c, n, kk);
                }
            }
            sx[pair * 24 + e] = interval_center(cen, tb.link_indep[idx * 3 + e]);
            if (WANT_J)
                for (int kk = 0; kk < n; kk++) sx[pair * 24 + 3 + kk * 3 + e] = gr[kk];
        }
        __syncthreads();

        const int q = q_begin + tid;
        const bool valid = q < q_end;
        const int qc = valid ? q : q_end - 1;
        const double* xs = sx + (qc / O - lt_first) * 24;
        const double x0 = xs[0], x1 = xs[1], x2 = xs[2];
        const double* pl = tb.planes + (size_t)b * ARMOUR_PLANE_COMPONENTS * ARMOUR_NPLANES * Q + qc;
        const size_t cs = (size_t)ARMOUR_NPLANES * Q;  // component stride

        double max_elt = -100000000.0;
        double mA0 = 0, mA1 = 0, mA2 = 0;
        bool neg = false;
        bool first = true;  // max_id == 0 initially: the winning normal defaults to plane 0
#pragma unroll 6
        for (int p = 0; p < ARMOUR_NPLANES; p++) {
            const size_t o = (size_t)p * Q;
            const double a0 = pl[o], a1 = pl[cs + o], a2 = pl[2 * cs + o];
            const double dd = pl[3 * cs + o], dl = pl[4 * cs + o];
            if (first) { mA0 = a0; mA1 = a1; mA2 = a2; first = false; }
            double pos_res = -100000000.0, neg_res = -100000000.0;
            if (a0 != 0.0 || a1 != 0.0 || a2 != 0.0) {  // A_elt.norm() > 0 (RT/CollisionChecking.cu:252)
                const double dot = a0 * x0 + a1 * x1 + a2 * x2;
                pos_res = dot - (dd + dl);
                neg_res = -dot - (-dd + dl);
            }
            if (pos_res > max_elt) { max_elt = pos_res; mA0 = a0; mA1 = a1; mA2 = a2; neg = false; }
            if (neg_res > max_elt) { max_elt = neg_res; mA0 = a0; mA1 = a1; mA2 = a2; neg = true; }
        }
        const size_t row0 = (size_t)n * T + q_begin;
        if (WANT_G && valid) g[row0 + tid] = -max_elt;
        if (WANT_J) {
            for (int kk = 0; kk < n; kk++) {
                const double* dx = xs + 3 + kk * 3;
                const double dot = mA0 * dx[0] + mA1 * dx[1] + mA2 * dx[2];
                sj[tid * n + kk] = neg ? dot : -dot;
            }
            __syncthreads();
            const int total = (q_end - q_begin) * n;
            double* jrow = jac + row0 * n;
            for (int i = tid; i < total; i += P2_BLOCK) jrow[i] = sj[i];
        }
    } else if (role < nbc + nbt) {
        // ------------------------------------------------------------------ torque rows (t*n + j)
        const int row = (role - nbc) * P2_TQ_ROWS + tid / 8;
        const int out = tid & 7;
        if (row < n * T && out <= n) {
            const int t = row / n, j = row - t * n;
            const size_t idx = ((size_t)b * n + j) * T + t;
            const int cnt = tb.tq_count[idx];
            const uint32_t* keys = tb.tq_keys + idx * tb.capT;
            const double* co = tb.tq_coeff + idx * tb.capT;
            if (out == 0) {
                if (WANT_G) {
                    double cen = tb.tq_center[idx];
                    for (int mo = 0; mo < cnt; mo++) cen += mono_value(kp, keys[mo], co[mo], n);
                    g[row] = interval_center(cen, tb.tq_indep[idx]);
                }
            } else if (WANT_J) {
                const int kk = out - 1;
                double gr = 0.0;
                for (int mo = 0; mo < cnt; mo++) {
                    const uint32_t key = keys[mo];
                    if ((key >> (2 * kk)) & 3u) gr += mono_grad(kp, key, co[mo], n, kk);
                }
                jac[(size_t)row * n + kk] = gr;
            }
        }
    } else {
        // ------------------------------------------------------------------ joint limit rows
        if (tid < n) {
            const int i = tid;
            const double* bz = tb.bez + (size_t)b * 3 * n;
            const double q0 = bz[i], a = bz[n + i], bb = bz[2 * n + i];
            const size_t off = (size_t)n * T + Q;
            for (int vel = 0; vel < 2; vel++) {
                double mn, mx, dmn, dmx;
                bez::joint_extremum(q0, a, bb, k[i], tb.k_range[i], tb.duration, vel != 0, &mn, &mx, &dmn, &dmx);
                const size_t r_mn = off + vel * 2 * n + i, r_mx = r_mn + n;
                if (WANT_G) { g[r_mn] = mn; g[r_mx] = mx; }
                if (WANT_J) {
                    for (int c = 0; c < n; c++) {
                        jac[r_mn * n + c] = (c == i) ? dmn : 0.0;
                        jac[r_mx * n + c] = (c == i) ? dmx : 0.0;
                    }
                }
            }
        }
    }
}

// link centres only (diagnostic file armour_joint_position_center.out): one thread per (b, l*T+t, axis)
__global__ void armour_p2_slice_links_kernel(P2Tables tb, const double* __restrict__ k_all, double* __restrict__ centers) {
    __shared__ KPow kp;
    const int b = blockIdx.y;
    fill_kpow(kp, k_all + (size_t)b * tb.n, tb.n);
    __syncthreads();
    const int task = blockIdx.x * blockDim.x + threadIdx.x;
    const int JT = tb.J * tb.T;
    if (task >= JT * 3) return;
    const int lt = task / 3, e = task - lt * 3;
    const size_t idx = (size_t)b * JT + lt;
    const int cnt = tb.link_count[idx];
    double cen = tb.link_center[idx * 3 + e];
    for (int mo = 0; mo < cnt; mo++)
        cen += mono_value(kp, tb.link_keys[idx * tb.capL + mo], tb.link_coeff[(idx * tb.capL + mo) * 3 + e], tb.n);
    const int l = lt / tb.T, t = lt - l * tb.T;
    centers[(((size_t)b * tb.T + t) * tb.J + l) * 3 + e] = interval_center(cen, tb.link_indep[idx * 3 + e]);
}

}  // namespace

const char* armour_p2_kernel_name(void) { return "armour_p2_eval_kernel"; }

int armour_p2_launch(const P2Tables& tb, const double* d_k, double* d_g, double* d_jac, hipStream_t stream) {
    if (!d_g && !d_jac) return ARMOUR_OK;
    const int nbc = (tb.Q + P2_BLOCK - 1) / P2_BLOCK;
    const int nbt = (tb.n * tb.T + P2_TQ_ROWS - 1) / P2_TQ_ROWS;
    const int max_pairs = tb.O > 0 ? (P2_BLOCK - 1) / tb.O + 2 : 1;
    const size_t smem = sizeof(KPow) + (size_t)max_pairs * 24 * sizeof(double) + (size_t)P2_BLOCK * ARMOUR_MAX_FACTORS * sizeof(double);
    dim3 grid(nbc + nbt + 1, tb.B), block(P2_BLOCK);
    if (d_g && d_jac)
        hipLaunchKernelGGL((armour_p2_eval_kernel<true, true>), grid, block, smem, stream, tb, d_k, d_g, d_jac, nbc, nbt, max_pairs);
    else if (d_g)
        hipLaunchKernelGGL((armour_p2_eval_kernel<true, false>), grid, block, smem, stream, tb, d_k, d_g, d_jac, nbc, nbt, max_pairs);
    else
        hipLaunchKernelGGL((armour_p2_eval_kernel<false, true>), grid, block, smem, stream, tb, d_k, d_g, d_jac, nbc, nbt, max_pairs);
    HIPCHK(hipGetLastError());
    return ARMOUR_OK;
}

int armour_p2_slice_links_launch(const P2Tables& tb, const double* d_k, double* d_centers, hipStream_t stream) {
    const int tasks = tb.J * tb.T * 3;
    dim3 grid((tasks + 255) / 256, tb.B), block(256);
    hipLaunchKernelGGL(armour_p2_slice_links_kernel, grid, block, 0, stream, tb, d_k, d_centers);
    HIPCHK(hipGetLastError());
    return ARMOUR_OK;
}
