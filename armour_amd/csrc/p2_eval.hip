// P2: the fused eval_g + eval_jac_g kernel (gfx950).
//
// Replaces, per IPOPT iterate, the reference's 2 x { OpenMP slice loop over 128 steps + 7 kernel launches
// + 8-16 blocking copies } (RT/NLPclass.cu:272-396, RT/CollisionChecking.cu:90-134,230-299,
// RT/PZsparse.cu:404-555, RT/Trajectory.cu:256-540) by ONE launch over all B problems:
//
//   blockIdx.y = problem b;  blockIdx.x selects the role:
//     [0, nbc)          collision blocks: 64 consecutive rows q = (l*T+t)*O + o each (lane = row), the block's
//                       four waves splitting each row's 36 half-spaces 9 apiece
//     [nbc, nbc+nbt)    torque blocks: 8 (t,j) rows x 32 monomial lanes each
//     nbc+nbt           joint position / velocity limit rows (4n rows, closed form)
//
// Collision block: every lane first issues its 9 planes x 5 components from planes[b][c][p][q] (q fastest:
// each load is one 512-B coalesced request per wave; 45 loads in flight per lane) and only then, while those
// are in flight, the block slices the (l,t) link PZs its rows touch (value + gradient,
// RT/PZsparse.cu:404-435,477-516): one thread per (pair, monomial, output entry) writes the monomial term to
// LDS, one thread per (pair, output entry) sums the terms in monomial order.  Each wave then scans its 9
// planes in the reference's order (pos_p before neg_p, strict >); wave 0 merges the four partial arg-maxes in
// plane order -- the same winner as the reference's serial scan -- and dots the winning normal with the 7
// centre derivatives.  Bound: HBM -- 1440 B of table per row against ~450 flop.  g rows are written coalesced;
// Jacobian rows are staged through LDS so the block writes 64*7 contiguous doubles.
//
// Arithmetic order follows the CPU statement of the reference exactly (products of k-powers in factor
// order, sums in monomial order), so with identical tables the outputs agree with the oracle to the ulp of
// pow() vs. repeated multiplication; FMA contraction is disabled for this file (see Makefile).
#include "p2_tiles.h"

using namespace p2;

namespace {

// EX: every wave holds exactly PPW live planes and the block slices in one pass: the PZ-table loads are issued first and
// the plane loads, a fixed number, after them, so the slicing (waiting for the tables with vmcnt(#plane loads)) runs while
// the planes are still in flight; the barriers before the scan wait for LDS traffic only.
template <bool WANT_G, bool WANT_J, bool MULTI, bool DFC, bool LL, int PPW, bool EX>
__global__ __launch_bounds__(P2_BLOCK) __attribute__((amdgpu_waves_per_eu(P2_WPE(DFC, MULTI)))) void armour_p2_eval_kernel(P2Tables tb, const double* __restrict__ k_all,
                                                                  double* __restrict__ g_all, double* __restrict__ jac_all,
                                                                  P2Launch lp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // (Round 4, measured and dropped: an XCD-aware mapping -- linear workgroup l runs on XCD l % 8; give each XCD a CONTIGUOUS range of the
    //  (problem, row block) pairs so that neighbouring blocks, which share their (link, time step) records, share an L2 -- configs[2]
    //  539 us against 500: what a block streams is 30 KB of its own rows and 3 KB of shared records, and eight XCDs that each walk one
    //  contiguous stretch of the table concentrate on fewer HBM channels at a time, like the tiled table layout of round 2.)
    const int b = blockIdx.y, role = blockIdx.x + lp.role0;
    const int n = tb.n, m = tb.m;   // (read together: one scalar load of the kernel argument block instead of two dependent ones)
    const double* k0 = k_all + (size_t)b * n;
    double* g0 = WANT_G ? g_all + (size_t)b * m : nullptr;
    double* jac0 = WANT_J ? jac_all + (size_t)b * m * n : nullptr;
    const double k_first = ((int)threadIdx.x < n) ? k0[threadIdx.x] : 0.0;   // issued before anything else: see p2_tiles.h
#ifdef P2_ABLATE  // development only: skip roles to attribute kernel time
    if ((P2_ABLATE & 1) && role < lp.nbc) return;
    if ((P2_ABLATE & 2) && role >= lp.nbc && role < lp.nbc + lp.nbt) return;
    if ((P2_ABLATE & 4) && role >= lp.nbc + lp.nbt) return;
#endif
    if (role < lp.nbc) collision_block<WANT_G, WANT_J, MULTI, DFC, LL, PPW, EX>(tb, lp, b, role, k0, k_first, g0, jac0, smem_raw);
    else if (role < lp.nbc + lp.nbt) torque_block<WANT_G, WANT_J, MULTI, DFC, LL, PPW, EX>(tb, lp, b, role, k0, k_first, g0, jac0, smem_raw);
    else limit_block<WANT_G, WANT_J, MULTI, DFC, LL, PPW, EX>(tb, lp, b, role, k0, k_first, g0, jac0, smem_raw);
}

// link centres only (diagnostic file armour_joint_position_center.out): one thread per (b, l*T+t, axis)
__global__ void armour_p2_slice_links_kernel(P2Tables tb, const double* __restrict__ k_all, double* __restrict__ centers) {
    __shared__ KPow kp;
    const int b = blockIdx.y;
    fill_kpow(kp, threadIdx.x < (unsigned)tb.n ? k_all[(size_t)b * tb.n + threadIdx.x] : 0.0, tb.n);
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

// What the EX kernels assume of a problem's live planes (p2_tiles.h collision_block): 24 of them; the first twelve pair an obstacle
// generator (waves 0 and 1, six each); the last twelve are link x link planes, six CONSECUTIVE ones per wave starting at an even
// p - 21 (their normals are then 144 contiguous, 16-B aligned bytes of the compact record and their deltas three aligned pairs).
// Worlds of axis-aligned boxes -- the reference's -- give planes {0..11 of the obstacle pairs} + 21..32.
bool armour_p2_ex_layout_ok(unsigned long long skip) {
    unsigned long long live = ~skip & ((1ull << ARMOUR_NPLANES) - 1ull);
    if (__builtin_popcountll(live) != 24) return false;
    int pl[24];
    for (int i = 0; i < 24; i++) { pl[i] = __builtin_ctzll(live); live &= live - 1ull; }
    if (pl[11] >= ARMOUR_FIRST_LL_PLANE || pl[12] < ARMOUR_FIRST_LL_PLANE) return false;
    for (int w = 2; w < 4; w++) {
        const int p0 = pl[6 * w];
        if ((p0 - ARMOUR_FIRST_LL_PLANE) & 1) return false;
        for (int i = 1; i < 6; i++) if (pl[6 * w + i] != p0 + i) return false;
    }
    return true;
}

int armour_p2_plan(const P2Tables& tb, int max_link, int max_torque, const unsigned long long* h_skip, int steps, long long k_stride,
                   long long g_stride, long long j_stride, P2Launch* lp_out, size_t* smem_out, bool* dfc_out, bool* six_out, bool* exact_out) {
    P2Launch lp;
    lp.role0 = 0;
    lp.nbc = (tb.Q + P2_ROWS - 1) / P2_ROWS;
    lp.nbt = tb.row0 == 0 ? 0 : (tb.n * tb.T + P2_TQ_ROWS - 1) / P2_TQ_ROWS;  // no torque rows in ARMTD mode and with TURN_OFF_INPUT_CONSTRAINTS (row0 = 0: the collision rows come first)
    lp.max_pairs = tb.O > 0 ? (P2_ROWS - 1) / tb.O + 2 : 1;
    lp.strideL = max_link > 0 ? max_link : 1;
    lp.strideT = max_torque > 0 ? max_torque : 1;
    lp.skip_by_value = (tb.B == 1 && h_skip) ? 1 : 0;
    lp.skip0 = lp.skip_by_value ? h_skip[0] : 0ull;
    lp.steps = steps; lp.k_stride = k_stride; lp.g_stride = g_stride; lp.j_stride = j_stride;
    lp.magic_O = div_magic(tb.O > 0 ? tb.O : 1); lp.magic_pp3 = div_magic(3 * lp.strideL);
    if (tb.Q >= (1 << 20)) { armour_set_error("more than 2^20 collision rows per problem"); return ARMOUR_ECAPACITY; }
    if (lp.strideL * 3 > P2_BLOCK * P2_TASK_ROUNDS) { armour_set_error("link PZ with %d monomials exceeds the P2 kernel's %d", lp.strideL, P2_BLOCK * P2_TASK_ROUNDS / 3); return ARMOUR_ECAPACITY; }
    lp.pair_chunk = std::max(1, std::min(std::min(lp.max_pairs, P2_BLOCK / P2_SL), (P2_BLOCK * P2_TASK_ROUNDS) / (lp.strideL * 3)));
    if (lp.strideT > 32 * P2_TQ_ROUNDS) { armour_set_error("torque PZ with %d monomials exceeds the P2 kernel's %d", lp.strideT, 32 * P2_TQ_ROUNDS); return ARMOUR_ECAPACITY; }
    const size_t col = ((size_t)lp.max_pairs * P2_SL + (size_t)lp.pair_chunk * lp.strideL * P2_SL + 4 * 64 * 4 + 64 * ARMOUR_MAX_FACTORS) * sizeof(double) + 4 * 64 * sizeof(int);
    const size_t tq = (size_t)P2_TQ_ROWS * lp.strideT * P2_TQW * sizeof(double);
    const size_t lim = (size_t)2 * ARMOUR_MAX_FACTORS * 8 * sizeof(double);
    const size_t smem = 2 * sizeof(KPow) + std::max(std::max(col, tq), lim);
    if (smem > 64 * 1024) { armour_set_error("P2 kernel needs %zu B of LDS (link/torque monomial counts too large)", smem); return ARMOUR_ECAPACITY; }
    const bool dfc = tb.obs_center != nullptr && tb.ll_shared;
    // every problem of the launch has at most 24 live planes (6 per wave): use the 6-slot kernels
    bool six = h_skip != nullptr;
    for (int b = 0; six && b < tb.B; b++) six = __builtin_popcountll(~h_skip[b] & ((1ull << ARMOUR_NPLANES) - 1ull)) <= 24;
    // ... exactly 24 (6 per wave) and the blocks slice in one pass: the EX kernels (ARMOUR_OPT_P2_EX = 0 turns them off)
    bool exact = six && tb.ex_allowed && lp.max_pairs <= lp.pair_chunk;
    for (int b = 0; exact && b < tb.B; b++) exact = armour_p2_ex_layout_ok(h_skip[b]);
    *lp_out = lp; *smem_out = smem; *dfc_out = dfc; *six_out = six; *exact_out = exact;
    return ARMOUR_OK;
}

int armour_p2_launch(const P2Tables& tb, int max_link, int max_torque, const unsigned long long* h_skip, const double* d_k, double* d_g, double* d_jac,
                     hipStream_t stream, int steps, long long k_stride, long long g_stride, long long j_stride, bool skip_collision_blocks) {
    if (!d_g && !d_jac) return ARMOUR_OK;
    P2Launch lp;
    size_t smem = 0;
    bool dfc, six, exact;
    const int rc = armour_p2_plan(tb, max_link, max_torque, h_skip, steps, k_stride, g_stride, j_stride, &lp, &smem, &dfc, &six, &exact);
    if (rc != ARMOUR_OK) return rc;
    if (skip_collision_blocks) lp.role0 = lp.nbc;   // the torque blocks and the limit block only
    dim3 grid(lp.nbc + lp.nbt + 1 - lp.role0, tb.B), block(P2_BLOCK);
#define P2_LAUNCH_M(G, J, M)                                                                                                                      \
    do {                                                                                                                                          \
        if (dfc && exact && !(M)) hipLaunchKernelGGL((armour_p2_eval_kernel<G, J, false, true, true, 6, true>), grid, block, smem, stream, tb, d_k, d_g, d_jac, lp);      \
        else if (dfc && six) hipLaunchKernelGGL((armour_p2_eval_kernel<G, J, M, true, true, 6, false>), grid, block, smem, stream, tb, d_k, d_g, d_jac, lp);                \
        else if (dfc) hipLaunchKernelGGL((armour_p2_eval_kernel<G, J, M, true, true, 9, false>), grid, block, smem, stream, tb, d_k, d_g, d_jac, lp);              \
        else if (tb.ll_shared && exact && !(M)) hipLaunchKernelGGL((armour_p2_eval_kernel<G, J, false, false, true, 6, true>), grid, block, smem, stream, tb, d_k, d_g, d_jac, lp); \
        else if (tb.ll_shared && six) hipLaunchKernelGGL((armour_p2_eval_kernel<G, J, M, false, true, 6, false>), grid, block, smem, stream, tb, d_k, d_g, d_jac, lp); \
        else if (tb.ll_shared) hipLaunchKernelGGL((armour_p2_eval_kernel<G, J, M, false, true, 9, false>), grid, block, smem, stream, tb, d_k, d_g, d_jac, lp); \
        else hipLaunchKernelGGL((armour_p2_eval_kernel<G, J, M, false, false, 9, false>), grid, block, smem, stream, tb, d_k, d_g, d_jac, lp);             \
    } while (0)
#define P2_LAUNCH(G, J)                                                                                                     \
    do {                                                                                                                    \
        if (steps > 1) P2_LAUNCH_M(G, J, true); else P2_LAUNCH_M(G, J, false);                                                                         \
    } while (0)
    if (d_g && d_jac) P2_LAUNCH(true, true);
    else if (d_g) P2_LAUNCH(true, false);
    else P2_LAUNCH(false, true);
#undef P2_LAUNCH
#undef P2_LAUNCH_M
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
