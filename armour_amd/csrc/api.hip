// C ABI of libarmour_hip.so (see include/armour_hip.h): handle lifetime, device-table ownership, the
// NLP-callback entry points and the host-side closed forms (bounds, cost, feasibility re-check).
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <chrono>
#include <cstring>
#include <mutex>
#include <new>

#include "../../include/armour_robot_fetch.h"
#include "../../include/armour_robot_kinova.h"
#include "bezier.h"
#include "cacc.h"
#include "common.h"

static thread_local char g_err[512] = "";

void armour_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* armour_last_error(void) { return g_err; }

extern "C" void armour_robot_kinova_gen3_no_gripper(ArmourRobot* robot) { armour_fill_kinova_gen3_no_gripper(robot); }
extern "C" void armour_robot_kinova_gen3_gripper(ArmourRobot* robot) { armour_fill_kinova_gen3_gripper(robot); }
extern "C" void armour_robot_fetch(ArmourRobot* robot) { armour_fill_fetch(robot); }
// (an 8-factor robot: the 128-bit-key ABI fills it, the 64-bit one says why it cannot -- include/armour_robot_fetch.h)
extern "C" int armour_robot_fetch8(ArmourRobot* robot) {
#if ARMOUR_MAX_FACTORS >= 8
    if (!robot) { armour_set_error("null argument"); return ARMOUR_EINVAL; }
    armour_fill_fetch8(robot);
    return ARMOUR_OK;
#else
    (void)robot;
    armour_set_error("armour_robot_fetch8: an 8-factor robot needs the 128-bit-key library (libarmour_hip_k128.so: armour_abi_max_factors() == 8)");
    return ARMOUR_EINVAL;
#endif
}
extern "C" void armour_params_default(ArmourParams* params, int32_t T) { armour_fill_default_params(params, T); }

extern "C" int armour_device_available(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n > 0 ? 1 : 0;
}

// page-locked host ranges handed out by armour_alloc_pinned: armour_eval_g_jac lets the kernel read k from and write
// g / jac to them directly over PCIe (one launch, no staging copies) when every buffer of the call lies in one
namespace {
struct PinnedRange { const char* p; size_t bytes; };
std::mutex g_pinned_mu;
std::vector<PinnedRange> g_pinned;
bool in_pinned(const void* q, size_t bytes) {
    if (!q) return true;
    const char* c = static_cast<const char*>(q);
    std::lock_guard<std::mutex> lk(g_pinned_mu);
    for (const PinnedRange& r : g_pinned)
        if (c >= r.p && c + bytes <= r.p + r.bytes) return true;
    return false;
}
}  // namespace

extern "C" int armour_alloc_pinned(uint64_t bytes, void** out) {
    if (!out) { armour_set_error("null argument"); return ARMOUR_EINVAL; }
    HIPCHK(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    std::lock_guard<std::mutex> lk(g_pinned_mu);
    g_pinned.push_back({static_cast<const char*>(*out), (size_t)(bytes ? bytes : 1)});
    return ARMOUR_OK;
}
extern "C" void armour_free_pinned(void* p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_pinned_mu);
        for (size_t i = 0; i < g_pinned.size(); i++)
            if (g_pinned[i].p == p) { g_pinned.erase(g_pinned.begin() + i); break; }
    }
    (void)hipHostFree(p);
}

template <class Tp>
static int dev_alloc(Tp** p, size_t count) {
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    if (count == 0) count = 1;
    HIPCHK(hipMalloc((void**)p, count * sizeof(Tp)));
    return ARMOUR_OK;
}
template <class Tp>
static void dev_free(Tp** p) {
    if (*p) { (void)hipFree(*p); *p = nullptr; }
}

static int ensure_capacity(ArmourPlanner* h, int B, int O) {
    if (B <= h->allocB && O <= h->allocO) return ARMOUR_OK;
    const int nb = B > h->allocB ? B : h->allocB, no = O > h->allocO ? O : h->allocO;
    // the buffers are replaced one by one below: until all of them exist at the new size the handle has NO capacity, so
    // that a hipMalloc failing half-way cannot leave small (or null) buffers behind capacity fields a later call trusts
    h->allocB = 0; h->allocO = 0;
    const size_t nl = (size_t)nb * h->J * h->T, nt = (size_t)nb * h->n * h->T;
    int rc;
#define TRY(x) if ((rc = (x)) != ARMOUR_OK) return rc
    TRY(dev_alloc(&h->d_link_count, nl));
    TRY(dev_alloc(&h->d_link_center, nl * 3));
    TRY(dev_alloc(&h->d_link_indep, nl * 3));
    TRY(dev_alloc(&h->d_link_keys, nl * h->lim.link_monomials));
    TRY(dev_alloc(&h->d_link_coeff, nl * h->lim.link_monomials * 3));
    TRY(dev_alloc(&h->d_tq_count, nt));
    TRY(dev_alloc(&h->d_tq_center, nt));
    TRY(dev_alloc(&h->d_tq_indep, nt));
    TRY(dev_alloc(&h->d_tq_keys, nt * h->lim.torque_monomials));
    TRY(dev_alloc(&h->d_tq_coeff, nt * h->lim.torque_monomials));
    TRY(dev_alloc(&h->d_planes, (size_t)nb * armour_planes_per_problem(h->J * h->T * no)));
    TRY(dev_alloc(&h->d_planes_ll, (size_t)nb * armour_planes_ll_per_problem(h->J * h->T)));
    TRY(dev_alloc(&h->d_obs_center, (size_t)nb * 3 * (no > 0 ? no : 1)));
    TRY(dev_alloc(&h->d_plane_skip, (size_t)nb));
    TRY(dev_alloc(&h->d_bez, (size_t)nb * 3 * h->n));
    const size_t mmax = (size_t)h->n * h->T + (size_t)h->J * h->T * no + 4 * h->n;
    TRY(dev_alloc(&h->d_k, (size_t)nb * h->n));
    // g and jac staging: ONE allocation, jac placed directly behind the g of the current problem set (begin_problem_set), so that the
    // synchronous host call can bring both back with a single device-to-host transfer when the caller's buffers are adjacent too.
    // The jac part doubles as the scratch of armour_get_link_centers (B*T*J*3 doubles): m*n >= T*J*3 does not hold for every
    // robot armour_create accepts (n*n < 3J with O = 0), so it is sized for both uses
    TRY(dev_alloc(&h->d_g, (size_t)nb * mmax + std::max((size_t)nb * mmax * h->n, (size_t)nb * h->T * h->J * 3)));
    h->d_jac = h->d_g + (size_t)nb * mmax;
    TRY(dev_alloc(&h->d_bounds, (size_t)2 * nb * mmax));
#undef TRY
    h->allocB = nb;
    h->allocO = no;
    return ARMOUR_OK;
}

P2Tables armour_make_tables(const ArmourPlanner* h) {
    P2Tables tb;
    memset(&tb, 0, sizeof(tb));
    tb.B = h->B; tb.T = h->T; tb.J = h->J; tb.n = h->n; tb.O = h->O; tb.Q = h->Q; tb.m = h->m;
    tb.capL = h->lim.link_monomials; tb.capT = h->lim.torque_monomials;
    tb.row0 = h->row0; tb.mode = h->mode;
    tb.link_count = h->d_link_count; tb.link_center = h->d_link_center; tb.link_indep = h->d_link_indep;
    tb.link_keys = h->d_link_keys; tb.link_coeff = h->d_link_coeff;
    tb.tq_count = h->d_tq_count; tb.tq_center = h->d_tq_center; tb.tq_indep = h->d_tq_indep;
    tb.tq_keys = h->d_tq_keys; tb.tq_coeff = h->d_tq_coeff;
    tb.planes = h->d_planes; tb.planes_ll = h->d_planes_ll; tb.ll_shared = h->ll_shared;
    tb.ex_allowed = h->tune(ARMOUR_OPT_P2_EX);
    // recomputing d = A.c in the kernel trades 8 B per plane and row for 5 flops: pays once the launch is bandwidth-bound
    // (measured: -25 % at B = 128, O = 50; +1 % at B = 1)
    // (a lean table of >= 8 problems has no d column at all: p1_reach.hip)
    tb.obs_center = (h->d_from_center && (h->B >= 8 || !h->planes_have_d)) ? h->d_obs_center : nullptr;
    tb.plane_skip = h->d_plane_skip; tb.bez = h->d_bez;
    for (int i = 0; i < ARMOUR_MAX_FACTORS; i++) tb.k_range[i] = h->params.k_range[i];
    tb.duration = h->params.duration;
    return tb;
}

// ---- per-handle launch-shape options (include/armour_hip.h: ARMOUR_OPT_FIRST_TUNING .. ARMOUR_OPT_LAST_TUNING) ----
namespace {
struct TuningSpec { int option; double def, lo, hi; };
// default and accepted range of every option (an option id without an entry is unknown)
const TuningSpec kTuning[] = {
    {ARMOUR_OPT_P1_STEP_WAVES, 0, 0, 4}, {ARMOUR_OPT_P1_STEP_FREE, 1, 0, 1}, {ARMOUR_OPT_P1_STEP_SPLIT_FK, -1, -1, 1}, {ARMOUR_OPT_P1_STEP_AUX3, 1, 0, 1},
    {ARMOUR_OPT_P1_MAX_WAVES_PER_CU, 4, 1, 8}, {ARMOUR_OPT_P1_TWO_PASS, 1, 0, 1}, {ARMOUR_OPT_P1_STEP_PAIRS, 1, 0, 2}, {ARMOUR_OPT_P1_STEP_QUEUE, 1, 0, 3}, {ARMOUR_OPT_P1_STEP_TAIL_CROSS, 0, 0, 514}, {ARMOUR_OPT_P1_STEP_TWO_CU, 3, 0, 33}, {ARMOUR_OPT_P1_STEP_LEAN_BACK, 1, 0, 3}, {ARMOUR_OPT_P1_TV_TAIL_CROSS, 111, 0, 514} /* (set_option refuses the values without a meaning) */,
    {ARMOUR_OPT_P1_TV_MIN_GROUPS, 31, 1, 1e6}, {ARMOUR_OPT_P1_TV_WAVES, 0, 0, 8}, {ARMOUR_OPT_P1_TV_FREE, 1, 0, 1}, {ARMOUR_OPT_P1_TV_SPLIT_FK, -1, -1, 1},
    {ARMOUR_OPT_P1_TV_DEDICATED, 1, 0, 1}, {ARMOUR_OPT_P1_TV_HELP_SHIFT, 0, 0, 3}, {ARMOUR_OPT_P1_TV_HELPERS, 1, 0, 31}, {ARMOUR_OPT_P1_TV_HELP_MIN, 192, 1, 1e6},
    {ARMOUR_OPT_P1_TV_HELP_N, 1, 0, 1}, {ARMOUR_OPT_P1_TV_AUX3, 1, 0, 1}, {ARMOUR_OPT_P1_TV_ROW_WIDTH, 0, 0, 64}, {ARMOUR_OPT_P1_FULL_PLANES, 0, 0, 1},
    {ARMOUR_OPT_P2_EX, 1, 0, 1}, {ARMOUR_OPT_STEPS_GRAPH_MIN, 2, 0, 1e6}, {ARMOUR_OPT_PINNED_MODE, 0, 0, 2}, {ARMOUR_OPT_CULL_ROWS, 0, 0, 1},
    {ARMOUR_OPT_SOLVE_SUB_TILES, 48, 1, 1e6}, {ARMOUR_OPT_SOLVE_DEVICE, 1, 0, 2}, {ARMOUR_OPT_SOLVE_CUT_TILES, 156, 1, 1e6}, {ARMOUR_OPT_SOLVE_BLOCKS, 0, 0, 1e6},
    {ARMOUR_OPT_SOLVE_SUB_BATCH, 0, 0, 1e6}, {ARMOUR_OPT_SOLVE_ROW_CAP, 0, 0, 1e7}, {ARMOUR_OPT_SOLVE_HARD_CAP_S, 0, 0, 1e6}, {ARMOUR_OPT_SOLVE_WAVES_PER_SIMD, 0, 0, 2}, {ARMOUR_OPT_SOLVE_CULL, -1, -1, 1},
};
const TuningSpec* tuning_spec(int option) {
    for (const TuningSpec& t : kTuning) if (t.option == option) return &t;
    return nullptr;
}
}  // namespace
void armour_tuning_defaults(double* tuning) {
    for (int o = ARMOUR_OPT_FIRST_TUNING; o <= ARMOUR_OPT_LAST_TUNING; o++) tuning[o - ARMOUR_OPT_FIRST_TUNING] = 0.0;
    for (const TuningSpec& t : kTuning) tuning[t.option - ARMOUR_OPT_FIRST_TUNING] = t.def;
}
// tracing: the only environment variables the library reads besides the worker's ARMOUR_WORKER_PARENT; neither changes a result
bool armour_trace_p1() { static const bool on = getenv("ARMOUR_P1_TRACE") != nullptr; return on; }
BuildStamps& armour_build_stamps() { static thread_local BuildStamps s; return s; }
void armour_build_stamp(const char* name) {
    if (!armour_trace_p1()) return;
    BuildStamps& s = armour_build_stamps();
    if (s.n < 12) { s.t[s.n] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); s.name[s.n] = name; s.n++; }
}
static void armour_build_stamps_print(const char* what) {
    if (!armour_trace_p1()) return;
    BuildStamps& s = armour_build_stamps();
    fprintf(stderr, "[%s, host side, us]", what);
    for (int i = 1; i < s.n; i++) fprintf(stderr, " %s %.1f", s.name[i], s.t[i] - s.t[i - 1]);
    if (s.n > 1) fprintf(stderr, " | total %.1f\n", s.t[s.n - 1] - s.t[0]);
}
bool armour_trace_solve() { static const bool on = getenv("ARMOUR_SOLVE_TIMING") != nullptr; return on; }

extern "C" int armour_create(const ArmourRobot* robot, const ArmourParams* params, const ArmourLimits* limits,
                             int32_t device, ArmourPlanner** out) {
    if (!robot || !params || !out) { armour_set_error("armour_create: null argument"); return ARMOUR_EINVAL; }
    if (robot->num_joints < 1 || robot->num_joints > ARMOUR_MAX_JOINTS || robot->num_factors < 1 ||
        robot->num_factors > robot->num_joints || robot->num_factors > ARMOUR_MAX_FACTORS) {
        armour_set_error("armour_create: unsupported robot (joints=%d, factors=%d; max %d joints, %d factors -- the monomial key "
                         "packs 9 bits per factor into 64, RT/PZsparse.h:8-21)", robot->num_joints, robot->num_factors,
                         ARMOUR_MAX_JOINTS, ARMOUR_MAX_FACTORS);
        return ARMOUR_EINVAL;
    }
    for (int i = 0; i < robot->num_joints; i++)
        if ((i < robot->num_factors) != (robot->axes[i] != 0)) {
            armour_set_error("armour_create: the actuated joints must come first and the fixed ones last (RT/Dynamics.cu:103-105)");
            return ARMOUR_EINVAL;
        }
    if (params->num_time_steps < 2 || (params->num_time_steps & 1)) {
        armour_set_error("armour_create: num_time_steps must be even and >= 2 (RT/Parameters.h:16)");
        return ARMOUR_EINVAL;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        armour_set_error("armour_create: no HIP device visible -- libarmour_hip has no CPU path");
        return ARMOUR_EDEVICE;
    }
    if (device < 0 || device >= ndev) { armour_set_error("armour_create: device %d out of range (%d visible)", device, ndev); return ARMOUR_EINVAL; }
    HIPCHK(hipSetDevice(device));
    ArmourPlanner* h = new (std::nothrow) ArmourPlanner();
    if (!h) { armour_set_error("out of host memory"); return ARMOUR_EINVAL; }
    h->robot = *robot;
    h->params = *params;
    armour_tuning_defaults(h->tuning);
    ArmourLimits lim;
    memset(&lim, 0, sizeof(lim));
    if (limits) lim = *limits;
    if (lim.max_batch <= 0) lim.max_batch = 1;
    if (lim.max_obstacles <= 0) lim.max_obstacles = 40;
    if (lim.link_monomials <= 0) lim.link_monomials = 32;
    if (lim.torque_monomials <= 0) lim.torque_monomials = 128;
    if (lim.work_monomials <= 0) lim.work_monomials = 1024;
    if (lim.raw_terms <= 0) lim.raw_terms = 4096;
    h->lim = lim;
    h->ub = armour_ultimate_bound(robot);
    h->device = device;
    h->T = params->num_time_steps; h->J = robot->num_joints; h->n = robot->num_factors;
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { armour_set_error("hipStreamCreate failed: %s", hipGetErrorString(e)); delete h; return ARMOUR_EDEVICE; }
    *out = h;
    return ARMOUR_OK;
}

double* armour_handle_pinned(ArmourPlanner* h, int slot, size_t bytes) {
    if (h->solve_pin[slot] && h->solve_pin_bytes[slot] >= bytes) return static_cast<double*>(h->solve_pin[slot]);
    if (h->solve_pin[slot]) { armour_free_pinned(h->solve_pin[slot]); h->solve_pin[slot] = nullptr; h->solve_pin_bytes[slot] = 0; }
    void* p = nullptr;
    if (armour_alloc_pinned(bytes, &p) != ARMOUR_OK) return nullptr;
    h->solve_pin[slot] = p; h->solve_pin_bytes[slot] = bytes;
    return static_cast<double*>(p);
}

static void drop_step_graphs(ArmourPlanner* h) {
    for (auto& g : h->step_graphs) (void)hipGraphExecDestroy(g.exec);
    h->step_graphs.clear();
}

extern "C" void armour_destroy(ArmourPlanner* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    drop_step_graphs(h);
    for (void* pin : h->solve_pin) armour_free_pinned(pin);
    dev_free(&h->d_bounds); dev_free(&h->d_viol);
    armour_relevance_free(h);
    if (h->d_tr_stage) (void)hipFree(h->d_tr_stage);
    dev_free(&h->solve_dev.ctl); dev_free(&h->solve_dev.blk_word); dev_free(&h->solve_dev.blk_rows); dev_free(&h->solve_dev.qp_rows);
    dev_free(&h->solve_dev.flags);
    dev_free(&h->d_jrs);
    armour_p1_free(h);
    dev_free(&h->d_link_count); dev_free(&h->d_link_center); dev_free(&h->d_link_indep);
    dev_free(&h->d_link_keys); dev_free(&h->d_link_coeff);
    dev_free(&h->d_tq_count); dev_free(&h->d_tq_center); dev_free(&h->d_tq_indep);
    dev_free(&h->d_tq_keys); dev_free(&h->d_tq_coeff);
    dev_free(&h->d_planes); dev_free(&h->d_planes_ll); dev_free(&h->d_obs_center); dev_free(&h->d_plane_skip); dev_free(&h->d_bez);
    dev_free(&h->d_k); dev_free(&h->d_g); h->d_jac = nullptr;   // (d_jac lives inside d_g's allocation)
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

// mode ARMTD: `qdd0` carries k_range [B][n] (the curve has no initial acceleration) and the third row of d_bez holds it
static int begin_problem_set(ArmourPlanner* h, int B, int O, const double* q0, const double* qd0, const double* qdd0,
                             const double* q_des, int mode = ARMOUR_MODE_ARMOUR) {
    if (!h || !q0 || !qd0 || !qdd0 || !q_des) { armour_set_error("null argument"); return ARMOUR_EINVAL; }
    if (B < 1 || O < 0) { armour_set_error("bad batch/obstacle count (B=%d, O=%d)", B, O); return ARMOUR_EINVAL; }
    // the reference parses whatever the file holds ("There is no check and warning, so be careful!", RT/armour_main.cu:14);
    // a NaN / inf state would only surface as NaN constraint rows, so it is refused here
    {
        const double* arrs[4] = {q0, qd0, qdd0, q_des};
        for (int a = 0; a < 4; a++)
            for (size_t i = 0; i < (size_t)B * h->n; i++)
                if (!std::isfinite(arrs[a][i])) { armour_set_error("non-finite entry in the initial state / goal arrays (array %d, entry %zu)", a, i); return ARMOUR_EINVAL; }
    }
    HIPCHK(hipSetDevice(h->device));
    h->ready = false;
    h->rel_fresh = false; h->rel2_fresh = false;
    h->bounds_on_device = false; h->bounds_on_host = false;
    h->tables_from_host = false;
    h->stats_fresh = false;
    drop_step_graphs(h);  // they bake in the tables of the previous problem set
    int rc = ensure_capacity(h, B, O);
    if (rc != ARMOUR_OK) return rc;
    h->B = B; h->O = O; h->Q = h->J * h->T * O;
    h->mode = mode;
    h->row0 = h->no_torque() ? 0 : h->n * h->T;  // CMP/NLPclass.cu:42-43, RT/NLPclass.cu:51-54 (TURN_OFF_INPUT_CONSTRAINTS): no torque rows, the collision rows first
    h->m = h->row0 + h->Q + 4 * h->n;
    h->d_jac = h->d_g + (size_t)B * h->m;   // directly behind this problem set's g (the allocation holds max_B * m_max * (1 + n) doubles)
    const size_t bn = (size_t)B * h->n;
    h->h_q0.assign(q0, q0 + bn); h->h_qd0.assign(qd0, qd0 + bn);
    h->h_qdes.assign(q_des, q_des + bn);
    if (mode == ARMOUR_MODE_ARMTD) { h->h_qdd0.assign(bn, 0.0); h->h_krange.assign(qdd0, qdd0 + bn); }
    else { h->h_qdd0.assign(qdd0, qdd0 + bn); h->h_krange.clear(); }
    // Bezier scalars: q0, Tqd0 = qd0*DURATION, TTqdd0 = qdd0*DURATION^2 (RT/Trajectory.cu:22-26); ARMTD: q0, qd0, k_range
    std::vector<double> bz((size_t)B * 3 * h->n);
    const double D = mode == ARMOUR_MODE_ARMTD ? 1.0 : h->params.duration;
    for (int b = 0; b < B; b++)
        for (int i = 0; i < h->n; i++) {
            bz[((size_t)b * 3 + 0) * h->n + i] = q0[b * h->n + i];
            bz[((size_t)b * 3 + 1) * h->n + i] = qd0[b * h->n + i] * D;
            bz[((size_t)b * 3 + 2) * h->n + i] = qdd0[b * h->n + i] * D * D;
        }
    HIPCHK(hipMemcpy(h->d_bez, bz.data(), bz.size() * sizeof(double), hipMemcpyHostToDevice));
    return ARMOUR_OK;
}

int armour_refresh_table_stats(ArmourPlanner* h) {
    if (h->stats_fresh) { h->stats_fresh = false; return ARMOUR_OK; }   // (the build read them back with its own results: p1_reach.hip)
    const size_t nl = (size_t)h->B * h->J * h->T, nt = (size_t)h->B * h->n * h->T;
    std::vector<int> lc(nl), tc(nt);
    HIPCHK(hipMemcpy(lc.data(), h->d_link_count, nl * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(tc.data(), h->d_tq_count, nt * sizeof(int), hipMemcpyDeviceToHost));
    long long sl = 0, st = 0;
    int ml = 0, mt = 0;
    for (int v : lc) { sl += v; if (v > ml) ml = v; }
    for (int v : tc) { st += v; if (v > mt) mt = v; }
    h->sum_link = sl; h->sum_torque = st; h->max_link = ml; h->max_torque = mt;
    h->h_plane_skip.assign((size_t)h->B, 0ull);
    if (h->O > 0) HIPCHK(hipMemcpy(h->h_plane_skip.data(), h->d_plane_skip, (size_t)h->B * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (h->O > 0 && armour_trace_p1()) {   // development: which planes the collision rows never need (p1_reach.hip, armour_p1_planes_kernel)
        unsigned long long all_and = ~0ull, all_or = 0ull;
        for (unsigned long long v : h->h_plane_skip) { all_and &= v; all_or |= v; }
        fprintf(stderr, "[P1 planes] plane_skip of %d problem(s): AND 0x%09llx OR 0x%09llx\n", h->B, all_and & ((1ull << ARMOUR_NPLANES) - 1ull), all_or & ((1ull << ARMOUR_NPLANES) - 1ull));
    }
    return ARMOUR_OK;
}

extern "C" int armour_set_problems(ArmourPlanner* h, int32_t B, int32_t O, const double* q0, const double* qd0,
                                   const double* qdd0, const double* q_des, const double* obstacles) {
    armour_build_stamps().n = 0;
    armour_build_stamp("start");
    int rc = begin_problem_set(h, B, O, q0, qd0, qdd0, q_des);
    if (rc != ARMOUR_OK) return rc;
    if (O > 0 && !obstacles) { armour_set_error("obstacles is null but O=%d", O); return ARMOUR_EINVAL; }
    for (size_t i = 0; i < (size_t)B * O * 12; i++)
        if (!std::isfinite(obstacles[i])) { armour_set_error("non-finite obstacle entry %zu", i); return ARMOUR_EINVAL; }
    armour_build_stamp("checks+bezier-upload");
    rc = armour_p1_build(h, obstacles);
    if (rc != ARMOUR_OK) return rc;
    armour_build_stamp("copy-out");
    rc = armour_refresh_table_stats(h);
    if (rc != ARMOUR_OK) return rc;
    h->ready = true;
    armour_build_stamp("table-stats");
    armour_build_stamps_print("armour_set_problems");
    return ARMOUR_OK;
}

extern "C" int armour_set_problems_armtd(ArmourPlanner* h, int32_t B, int32_t O, const double* q0, const double* qd0, const double* q_des,
                                         const double* jrs, const double* k_range, const double* obstacles) {
    if (!jrs || !k_range) { armour_set_error("null argument"); return ARMOUR_EINVAL; }
    int rc = begin_problem_set(h, B, O, q0, qd0, k_range, q_des, ARMOUR_MODE_ARMTD);
    if (rc != ARMOUR_OK) return rc;
    if (O > 0 && !obstacles) { armour_set_error("obstacles is null but O=%d", O); return ARMOUR_EINVAL; }
    for (size_t i = 0; i < (size_t)B * O * 12; i++)
        if (!std::isfinite(obstacles[i])) { armour_set_error("non-finite obstacle entry %zu", i); return ARMOUR_EINVAL; }
    for (size_t i = 0; i < (size_t)B * h->n * 6 * h->T; i++)
        if (!std::isfinite(jrs[i])) { armour_set_error("non-finite entry %zu in the offline JRS tables", i); return ARMOUR_EINVAL; }
    const size_t cnt = (size_t)B * h->n * 6 * h->T;
    if (cnt > h->jrs_cap) {
        if ((rc = dev_alloc(&h->d_jrs, cnt)) != ARMOUR_OK) return rc;
        h->jrs_cap = cnt;
    }
    HIPCHK(hipMemcpy(h->d_jrs, jrs, cnt * sizeof(double), hipMemcpyHostToDevice));
    rc = armour_p1_build(h, obstacles);
    if (rc != ARMOUR_OK) return rc;
    rc = armour_refresh_table_stats(h);
    if (rc != ARMOUR_OK) return rc;
    h->ready = true;
    return ARMOUR_OK;
}

extern "C" int armour_debug_load_tables(ArmourPlanner* h, int32_t B, int32_t O, const double* q0, const double* qd0,
                                        const double* qdd0, const double* q_des, const int32_t* link_count,
                                        const double* link_center, const uint64_t* link_keys, const double* link_coeffs,
                                        int32_t cap_l, const int32_t* torque_count, const double* torque_center,
                                        const uint64_t* torque_keys, const double* torque_coeffs, int32_t cap_t,
                                        const double* A, const double* d, const double* delta, const double* torque_radius) {
    int rc = begin_problem_set(h, B, O, q0, qd0, qdd0, q_des);
    if (rc != ARMOUR_OK) return rc;
    h->build_ms = 0; h->build_info[0] = h->build_info[1] = h->build_info[2] = h->build_info[3] = 0;   // no reach-set kernel ran
    h->h_prune_margin.clear();   // (no verdicts were taken here)
    const int J = h->J, T = h->T, n = h->n, capL = h->lim.link_monomials, capT = h->lim.torque_monomials;
    const size_t nl = (size_t)B * J * T, nt = (size_t)B * n * T;
    // centres arrive as [..][2][sz]: slot 0 = centre, slot 1 = independent radius
    std::vector<double> lc(nl * 3), li(nl * 3), lco(nl * capL * 3, 0.0), tc(nt), ti(nt), tco(nt * capT, 0.0);
    std::vector<uint32_t> lk(nl * capL, 0), tk(nt * capT, 0);
    for (size_t i = 0; i < nl; i++) {
        if (link_count[i] > capL) { armour_set_error("link PZ %zu has %d monomials > capacity %d", i, link_count[i], capL); return ARMOUR_ECAPACITY; }
        for (int e = 0; e < 3; e++) { lc[i * 3 + e] = link_center[(i * 2) * 3 + e]; li[i * 3 + e] = link_center[(i * 2 + 1) * 3 + e]; }
        for (int mo = 0; mo < link_count[i]; mo++) {
            lk[i * capL + mo] = (uint32_t)link_keys[i * cap_l + mo];
            for (int e = 0; e < 3; e++) lco[(i * capL + mo) * 3 + e] = link_coeffs[(i * cap_l + mo) * 3 + e];
        }
    }
    for (size_t i = 0; i < nt; i++) {
        if (torque_count[i] > capT) { armour_set_error("torque PZ %zu has %d monomials > capacity %d", i, torque_count[i], capT); return ARMOUR_ECAPACITY; }
        tc[i] = torque_center[i * 2]; ti[i] = torque_center[i * 2 + 1];
        for (int mo = 0; mo < torque_count[i]; mo++) { tk[i * capT + mo] = (uint32_t)torque_keys[i * cap_t + mo]; tco[i * capT + mo] = torque_coeffs[i * cap_t + mo]; }
    }
    // reference layout [b][t][l][o][p] -> tiled device layout (common.h: armour_plane_index)
    const size_t ppp = armour_planes_per_problem(h->Q);
    std::vector<double> pl((size_t)B * ppp, 0.0);
    for (int b = 0; b < B; b++)
        for (int t = 0; t < T; t++)
            for (int l = 0; l < J; l++)
                for (int o = 0; o < O; o++)
                    for (int p = 0; p < 36; p++) {
                        const size_t src = ((((size_t)b * T + t) * J + l) * O + o) * 36 + p;
                        const int q = (l * T + t) * O + o;
                        double* base = &pl[(size_t)b * ppp];
                        base[armour_plane_index(h->Q, q, p, 0)] = A[src * 3 + 0]; base[armour_plane_index(h->Q, q, p, 1)] = A[src * 3 + 1];
                        base[armour_plane_index(h->Q, q, p, 2)] = A[src * 3 + 2];
                        base[armour_plane_index(h->Q, q, p, 3)] = d[src]; base[armour_plane_index(h->Q, q, p, 4)] = delta[src];
                    }
#define UP(dst, vec) HIPCHK(hipMemcpy(dst, vec.data(), vec.size() * sizeof(vec[0]), hipMemcpyHostToDevice))
    HIPCHK(hipMemcpy(h->d_link_count, link_count, nl * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_tq_count, torque_count, nt * sizeof(int), hipMemcpyHostToDevice));
    UP(h->d_link_center, lc); UP(h->d_link_indep, li); UP(h->d_link_keys, lk); UP(h->d_link_coeff, lco);
    UP(h->d_tq_center, tc); UP(h->d_tq_indep, ti); UP(h->d_tq_keys, tk); UP(h->d_tq_coeff, tco);
    if (!pl.empty()) UP(h->d_planes, pl);
    {
        // compact link x link normals (taken at obstacle 0); used only if the loaded normals really are obstacle-independent
        const int JT = J * T;
        std::vector<double> pll((size_t)B * armour_planes_ll_per_problem(JT), 0.0);
        bool shared = O > 0;
        for (int b = 0; b < B && O > 0; b++)
            for (int t = 0; t < T; t++)
                for (int l = 0; l < J; l++)
                    for (int p = ARMOUR_FIRST_LL_PLANE; p < 36; p++)
                        for (int o = 0; o < O; o++) {
                            const size_t src = ((((size_t)b * T + t) * J + l) * O + o) * 36 + p, src0 = ((((size_t)b * T + t) * J + l) * O) * 36 + p;
                            for (int cc = 0; cc < 3; cc++) {
                                if (o == 0) pll[(size_t)b * armour_planes_ll_per_problem(JT) + armour_plane_ll_index(JT, l * T + t, p - ARMOUR_FIRST_LL_PLANE, cc)] = A[src * 3 + cc];
                                else if (memcmp(&A[src * 3 + cc], &A[src0 * 3 + cc], sizeof(double)) != 0) shared = false;
                            }
                        }
        UP(h->d_planes_ll, pll);
        h->ll_shared = shared ? 1 : 0;
        h->d_from_center = 0;  // loaded tables: use the d column as given
        h->planes_lean = 0; h->planes_have_d = 1;
    }
    HIPCHK(hipMemset(h->d_plane_skip, 0, (size_t)B * sizeof(unsigned long long)));  // loaded tables: evaluate every plane
#undef UP
    h->h_torque_radius.assign(torque_radius, torque_radius + (size_t)B * n * T);
    h->h_link_gens.assign((size_t)B * T * J * 18, 0.0);
    rc = armour_refresh_table_stats(h);
    if (rc != ARMOUR_OK) return rc;
    h->tables_from_host = true;
    h->ready = true;
    return ARMOUR_OK;
}

#define NEED_READY(h)                                                                        \
    if (!(h)) { armour_set_error("null handle"); return ARMOUR_EINVAL; }                     \
    if (!(h)->ready) { armour_set_error("no problem set: call armour_set_problems first"); return ARMOUR_ESTATE; }

extern "C" int armour_get_sizes(const ArmourPlanner* h, int32_t* B, int32_t* n, int32_t* m) {
    NEED_READY(h);
    if (B) *B = h->B;
    if (n) *n = h->n;
    if (m) *m = h->m;
    return ARMOUR_OK;
}

extern "C" int armour_get_bounds(ArmourPlanner* h, double* x_l, double* x_u, double* g_l, double* g_u) {
    NEED_READY(h);
    const int T = h->T, n = h->n;
    if (x_l) for (int i = 0; i < n; i++) x_l[i] = -1.0;
    if (x_u) for (int i = 0; i < n; i++) x_u[i] = 1.0;
    if (!g_l || !g_u) return ARMOUR_OK;
    for (int b = 0; b < h->B; b++) {
        double* gl = g_l + (size_t)b * h->m;
        double* gu = g_u + (size_t)b * h->m;
        if (!h->no_torque()) {   // RT/NLPclass.cu:117
            const double* tr = &h->h_torque_radius[(size_t)b * n * T];
            for (int t = 0; t < T; t++)
                for (int j = 0; j < n; j++) {
                    gl[t * n + j] = -h->robot.torque_limits[j] + tr[j * T + t];
                    gu[t * n + j] = h->robot.torque_limits[j] - tr[j * T + t];
                }
        }
        size_t off = (size_t)h->row0;  // ARMTD mode (CMP/NLPclass.cu:73-140): the same collision and limit bounds, no torque rows
        for (size_t i = off; i < off + (size_t)h->Q; i++) { gl[i] = -1e19; gu[i] = 0; }
        off += h->Q;
        for (int rep = 0; rep < 2; rep++, off += n)
            for (int i = 0; i < n; i++) { gl[off + i] = h->robot.state_limits_lb[i] + h->ub.qe; gu[off + i] = h->robot.state_limits_ub[i] - h->ub.qe; }
        for (int rep = 0; rep < 2; rep++, off += n)
            for (int i = 0; i < n; i++) { gl[off + i] = -h->robot.speed_limits[i] + h->ub.qde; gu[off + i] = h->robot.speed_limits[i] - h->ub.qde; }
    }
    return ARMOUR_OK;
}

int armour_checked_collision_rows(const ArmourPlanner* h) {
    return h->mode == ARMOUR_MODE_ARMTD ? (h->n - 1 < h->J ? h->n - 1 : h->J) * h->T * h->O : h->Q;
}

static double wrap_to_pi(double a) {
    const double pi = 3.14159265358979323846;
    while (a < -pi) a += 2 * pi;
    while (a > pi) a -= 2 * pi;
    return a;
}

extern "C" int armour_eval_f(ArmourPlanner* h, const double* k, double* f) {
    NEED_READY(h);
    const int n = h->n;
    const double D = h->params.duration;
    for (int b = 0; b < h->B; b++) {
        double obj = 0;
        for (int pass = 0; pass < 2; pass++)  // continuous joints first, as RT/NLPclass.cu:225-231 sums them
            for (int i = 0; i < n; i++) {
                if ((h->robot.continuous[i] != 0) != (pass == 0)) continue;
                const size_t ix = (size_t)b * n + i;
                const double qp = h->mode == ARMOUR_MODE_ARMTD  // CMP/NLPclass.cu:183-212
                                      ? cacc::q_plan(h->h_q0[ix], h->h_qd0[ix], h->h_krange[ix], k[ix])
                                      : bez::q_des(h->h_q0[ix], h->h_qd0[ix] * D, h->h_qdd0[ix] * D * D, h->params.k_range[i] * k[ix], h->params.t_plan);
                const double e = h->robot.continuous[i] ? wrap_to_pi(h->h_qdes[ix] - qp) : (h->h_qdes[ix] - qp);
                obj += e * e;
            }
        f[b] = obj * h->params.cost_scale;
    }
    return ARMOUR_OK;
}

extern "C" int armour_eval_grad_f(ArmourPlanner* h, const double* k, double* grad_f) {
    NEED_READY(h);
    const int n = h->n;
    const double D = h->params.duration, tp = h->params.t_plan;
    for (int b = 0; b < h->B; b++)
        for (int i = 0; i < n; i++) {
            const size_t ix = (size_t)b * n + i;
            const bool armtd = h->mode == ARMOUR_MODE_ARMTD;  // CMP/NLPclass.cu:217-243
            const double qp = armtd ? cacc::q_plan(h->h_q0[ix], h->h_qd0[ix], h->h_krange[ix], k[ix])
                                    : bez::q_des(h->h_q0[ix], h->h_qd0[ix] * D, h->h_qdd0[ix] * D * D, h->params.k_range[i] * k[ix], tp);
            const double dk = armtd ? cacc::q_plan_dk(h->h_krange[ix]) : (tp * tp * tp) * (6 * tp * tp - 15 * tp + 10) * h->params.k_range[i];
            const double e = h->robot.continuous[i] ? wrap_to_pi(qp - h->h_qdes[ix]) : (qp - h->h_qdes[ix]);
            grad_f[ix] = 2 * e * dk * h->params.cost_scale;
        }
    return ARMOUR_OK;
}

extern "C" int armour_desired_trajectory(int32_t n, const double* q0, const double* qd0, const double* qdd0, const double* k_range,
                                         double duration, const double* k, double t, double* q, double* qd, double* qdd) {
    if (n < 1 || !q0 || !qd0 || !qdd0 || !k_range || !k || !(duration > 0)) { armour_set_error("bad argument"); return ARMOUR_EINVAL; }
    const double D = duration, s = t / D;  // the curve is parameterised on s = t / DURATION in [0,1] with qd0, qdd0 scaled accordingly (RT/Trajectory.cu:22-23)
    for (int i = 0; i < n; i++) {
        const double a = qd0[i] * D, b = qdd0[i] * D * D, ka = k_range[i] * k[i];
        if (q) q[i] = bez::q_des(q0[i], a, b, ka, s);
        if (qd) qd[i] = bez::qd_des(q0[i], a, b, ka, s) / D;
        if (qdd) qdd[i] = bez::qdd_des(q0[i], a, b, ka, s) / (D * D);
    }
    return ARMOUR_OK;
}

extern "C" int armour_eval_g_jac_device(ArmourPlanner* h, const double* d_k, double* d_g, double* d_jac, void* stream) {
    NEED_READY(h);
    if (!d_k) { armour_set_error("d_k is null"); return ARMOUR_EINVAL; }
    const P2Tables tb = armour_make_tables(h);
    return armour_p2_launch(tb, h->max_link, h->max_torque, h->h_plane_skip.data(), d_k, d_g, d_jac, stream ? (hipStream_t)stream : h->stream);
}

static int enqueue_steps(ArmourPlanner* h, const double* d_k, int steps, double* d_g, double* d_jac, hipStream_t st) {
    const P2Tables tb = armour_make_tables(h);
    const size_t stride = (size_t)h->B * h->n;
    for (int s = 0; s < steps; s++) {
        int rc = armour_p2_launch(tb, h->max_link, h->max_torque, h->h_plane_skip.data(), d_k + (size_t)s * stride, d_g, d_jac, st);
        if (rc != ARMOUR_OK) return rc;
    }
    return ARMOUR_OK;
}

// The `steps` launches as ONE instantiated hipGraph (a chain: each launch still waits for the one before).  Submitting a
// kernel costs the host 2.4-3.3 us per hipLaunchKernel on this platform -- as long as a small launch runs -- while the
// nodes of a graph are pre-built AQL packets the command processor takes back to back (1.55 us per empty node measured,
// tools/micro/launch_floor.hip).  The graph bakes in the pointers and the tables: it is keyed on (d_k, steps, d_g, d_jac)
// and dropped with the problem set.
static int steps_graph(ArmourPlanner* h, const double* d_k, int steps, double* d_g, double* d_jac, hipGraphExec_t* out) {
    h->graph_clock++;
    for (auto& g : h->step_graphs)
        if (g.d_k == d_k && g.steps == steps && g.d_g == d_g && g.d_jac == d_jac) { g.last_use = h->graph_clock; *out = g.exec; return ARMOUR_OK; }
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    hipGraph_t graph = nullptr;
    HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    const int rc = enqueue_steps(h, d_k, steps, d_g, d_jac, h->stream);
    const hipError_t e = hipStreamEndCapture(h->stream, &graph);
    if (rc != ARMOUR_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) { armour_set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e)); return ARMOUR_EDEVICE; }
    hipGraphExec_t exec = nullptr;
    const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess) { armour_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(ei)); return ARMOUR_EDEVICE; }
    if (h->step_graphs.size() >= 4) {  // keep the four most recently used
        size_t lru = 0;
        for (size_t i = 1; i < h->step_graphs.size(); i++) if (h->step_graphs[i].last_use < h->step_graphs[lru].last_use) lru = i;
        (void)hipGraphExecDestroy(h->step_graphs[lru].exec);
        h->step_graphs.erase(h->step_graphs.begin() + lru);
    }
    h->step_graphs.push_back({d_k, steps, d_g, d_jac, exec, h->graph_clock});
    *out = exec;
    return ARMOUR_OK;
}

static bool steps_use_graph(const ArmourPlanner* h, int steps) {
    const int min_steps = h->tune(ARMOUR_OPT_STEPS_GRAPH_MIN);   // (0 = never)
    return min_steps > 0 && steps >= min_steps;
}

extern "C" int armour_prepare_steps(ArmourPlanner* h, const double* d_k, int32_t steps, double* d_g, double* d_jac) {
    NEED_READY(h);
    if (!d_k || steps < 0) { armour_set_error("bad argument"); return ARMOUR_EINVAL; }
    if (!steps_use_graph(h, steps)) return ARMOUR_OK;
    hipGraphExec_t exec;
    return steps_graph(h, d_k, steps, d_g, d_jac, &exec);
}

extern "C" int armour_eval_g_jac_device_steps(ArmourPlanner* h, const double* d_k, int32_t steps, double* d_g, double* d_jac,
                                              void* stream) {
    NEED_READY(h);
    if (!d_k || steps < 0) { armour_set_error("bad argument"); return ARMOUR_EINVAL; }
    if (steps == 0 || (!d_g && !d_jac)) return ARMOUR_OK;
    const hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    // a graph only if armour_prepare_steps built one for exactly these arguments: a caller that slides its pointers
    // from call to call would otherwise pay a capture + instantiate every time
    if (steps_use_graph(h, steps))
        for (auto& g : h->step_graphs)
            if (g.d_k == d_k && g.steps == steps && g.d_g == d_g && g.d_jac == d_jac) {
                g.last_use = ++h->graph_clock;
                HIPCHK(hipGraphLaunch(g.exec, st));
                return ARMOUR_OK;
            }
    return enqueue_steps(h, d_k, steps, d_g, d_jac, st);
}

extern "C" int armour_eval_g_jac_device_multi(ArmourPlanner* h, const double* d_k, int32_t points, double* d_g, double* d_jac,
                                              void* stream) {
    NEED_READY(h);
    if (!d_k || points < 0) { armour_set_error("bad argument"); return ARMOUR_EINVAL; }
    if (points == 0) return ARMOUR_OK;
    const P2Tables tb = armour_make_tables(h);
    const long long bn = (long long)h->B * h->n, bm = (long long)h->B * h->m;
    return armour_p2_launch(tb, h->max_link, h->max_torque, h->h_plane_skip.data(), d_k, d_g, d_jac, stream ? (hipStream_t)stream : h->stream,
                            points, bn, bm, bm * h->n);
}

static int spin_on_stream(hipStream_t st) {
    // spin on the stream instead of sleeping in hipStreamSynchronize: the whole call is tens of microseconds and an
    // interrupt-driven wake-up would dominate it
    for (;;) {
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) return ARMOUR_OK;
        if (q != hipErrorNotReady) { armour_set_error("hipStreamQuery failed: %s", hipGetErrorString(q)); return ARMOUR_EDEVICE; }
    }
}

extern "C" int armour_eval_g_jac(ArmourPlanner* h, const double* k, double* g, double* jac) {
    NEED_READY(h);
    if (!k) { armour_set_error("k is null"); return ARMOUR_EINVAL; }
    HIPCHK(hipSetDevice(h->device));
    const size_t bn = (size_t)h->B * h->n, bm = (size_t)h->B * h->m;
    const P2Tables tb = armour_make_tables(h);
    // Buffers from armour_alloc_pinned.  Default (mode 0): the copies below run as true asynchronous DMA transfers (k in, g / jac out)
    // ordered with the launch on the handle's stream -- 2 x 8 m (1 + n) bytes cross PCIe once, in two large transfers.
    // ARMOUR_OPT_PINNED_MODE = 1 is the round-2 form: the kernel reads k from and writes g / jac to host memory itself (zero copy).  It
    // measured 39-43 us per call on some boxes and ~0.9 ms on freshly started ones (BENCH_r02.json, profiles/r03_pinned_probe.txt:
    // every block's k read and every row tile's stores become individual PCIe transactions whose cost depends on how the host
    // mapping is set up), and once did not return at all, so it is no longer chosen by default.  Mode 2: only k is read in place.
    const int pinned_mode = h->tune(ARMOUR_OPT_PINNED_MODE);
    const bool all_pinned = in_pinned(k, bn * sizeof(double)) && in_pinned(g, bm * sizeof(double)) && in_pinned(jac, bm * h->n * sizeof(double));
    if (all_pinned && pinned_mode == 1) {
        int rc = armour_p2_launch(tb, h->max_link, h->max_torque, h->h_plane_skip.data(), k, g, jac, h->stream);
        if (rc != ARMOUR_OK) return rc;
        return spin_on_stream(h->stream);
    }
    const double* k_dev = h->d_k;
    if (all_pinned && pinned_mode == 2) k_dev = k;
    else HIPCHK(hipMemcpyAsync(h->d_k, k, bn * sizeof(double), hipMemcpyHostToDevice, h->stream));
    int rc = armour_p2_launch(tb, h->max_link, h->max_torque, h->h_plane_skip.data(), k_dev, g ? h->d_g : nullptr, jac ? h->d_jac : nullptr, h->stream);
    if (rc != ARMOUR_OK) return rc;
    if (g && jac && jac == g + bm) {   // the caller's g and jac are one block (as the device copies are): one transfer instead of two
        HIPCHK(hipMemcpyAsync(g, h->d_g, bm * (1 + (size_t)h->n) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        return spin_on_stream(h->stream);
    }
    if (g) HIPCHK(hipMemcpyAsync(g, h->d_g, bm * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (jac) HIPCHK(hipMemcpyAsync(jac, h->d_jac, bm * h->n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    return spin_on_stream(h->stream);
}

extern "C" int armour_check_feasible(ArmourPlanner* h, const double* g, int32_t* feasible) {
    NEED_READY(h);
    const int T = h->T, n = h->n;
    const double tt = h->params.torque_violation_threshold, ct = h->params.collision_violation_threshold;
    for (int b = 0; b < h->B; b++) {
        const double* gb = g + (size_t)b * h->m;
        bool ok = true;
        if (!h->no_torque()) {   // RT/NLPclass.cu:453
            const double* tr = &h->h_torque_radius[(size_t)b * n * T];
            for (int t = 0; t < T && ok; t++)
                for (int j = 0; j < n; j++) {
                    const double v = gb[t * n + j];
                    if (v < -h->robot.torque_limits[j] + tr[j * T + t] - tt || v > h->robot.torque_limits[j] - tr[j * T + t] + tt) { ok = false; break; }
                }
        }
        size_t off = (size_t)h->row0;
        // CMP/NLPclass.cu:391-402 re-checks the collision rows of links 0 .. NUM_FACTORS-2 only (RT checks every link)
        const size_t n_checked = (size_t)armour_checked_collision_rows(h);
        for (size_t i = 0; i < n_checked && ok; i++)
            if (gb[off + i] > ct) ok = false;
        off += h->Q;
        for (int rep = 0; rep < 2 && ok; rep++, off += n)
            for (int i = 0; i < n; i++)
                if (gb[off + i] < h->robot.state_limits_lb[i] + h->ub.qe || gb[off + i] > h->robot.state_limits_ub[i] - h->ub.qe) { ok = false; break; }
        for (int rep = 0; rep < 2 && ok; rep++, off += n)
            for (int i = 0; i < n; i++)
                if (gb[off + i] < -h->robot.speed_limits[i] + h->ub.qde || gb[off + i] > h->robot.speed_limits[i] - h->ub.qde) { ok = false; break; }
        feasible[b] = ok ? 1 : 0;
    }
    return ARMOUR_OK;
}

// g_l / g_u of every problem on the device, the values of armour_get_bounds (RT/NLPclass.cu:87-165) expression for expression, filled by a kernel
// from the torque radii (B n T doubles to upload instead of 2 B m: 73 MB at B = 128, O = 50 -- 3 ms of the first armour_solve after a build)
namespace {
struct BoundsArgs {
    int n, T, m, row0, Q, no_torque;
    double torque_limits[ARMOUR_MAX_FACTORS], lb[ARMOUR_MAX_FACTORS], ub[ARMOUR_MAX_FACTORS], speed[ARMOUR_MAX_FACTORS];
    double qe, qde;
    const double* tr;   // [B][n][T]
    double* lo; double* hi;   // [B][m]
};
__global__ __launch_bounds__(256) void armour_bounds_kernel(BoundsArgs a) {
    const int b = blockIdx.y, r = blockIdx.x * 256 + threadIdx.x;
    if (r >= a.m) return;
    double l, u;
    if (r < a.row0) {   // row t * n + j
        const int t = r / a.n, j = r - t * a.n;
        const double tr = a.tr[((size_t)b * a.n + j) * a.T + t];
        l = -a.torque_limits[j] + tr; u = a.torque_limits[j] - tr;
    } else if (r < a.row0 + a.Q) { l = -1e19; u = 0; }
    else {
        const int e = r - a.row0 - a.Q, rep = e / a.n, i = e - rep * a.n;
        if (rep < 2) { l = a.lb[i] + a.qe; u = a.ub[i] - a.qe; }
        else { l = -a.speed[i] + a.qde; u = a.speed[i] - a.qde; }
    }
    a.lo[(size_t)b * a.m + r] = l;
    a.hi[(size_t)b * a.m + r] = u;
}
}  // namespace

// the kernel, queued on the handle's stream: d_tr = the torque radii [B][n][T] on the device (unused without torque rows)
int armour_bounds_launch(ArmourPlanner* h, const double* d_tr) {
    const size_t bm = (size_t)h->B * h->m;
    BoundsArgs a;
    a.n = h->n; a.T = h->T; a.m = h->m; a.row0 = h->row0; a.Q = h->Q; a.no_torque = h->no_torque() ? 1 : 0;
    for (int j = 0; j < ARMOUR_MAX_FACTORS; j++) {
        a.torque_limits[j] = h->robot.torque_limits[j]; a.lb[j] = h->robot.state_limits_lb[j]; a.ub[j] = h->robot.state_limits_ub[j]; a.speed[j] = h->robot.speed_limits[j];
    }
    a.qe = h->ub.qe; a.qde = h->ub.qde;
    a.lo = h->d_bounds; a.hi = h->d_bounds + bm;
    a.tr = d_tr;
    hipLaunchKernelGGL(armour_bounds_kernel, dim3((h->m + 255) / 256, h->B), dim3(256), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    h->bounds_on_device = true;   // (for everything queued on the handle's stream after this)
    return ARMOUR_OK;
}

// Lazy form, for problem sets whose tables came from the host (armour_set_tables): the reach-set build queues the kernel itself, behind its own
// kernels, from the torque radii it has on the device (p1_reach.hip) -- no upload, no wait.
int armour_upload_bounds(ArmourPlanner* h) {
    if (h->bounds_on_device) return ARMOUR_OK;
    const double* d_tr = nullptr;
    if (!h->no_torque()) {
        const size_t ntr = (size_t)h->B * h->n * h->T;
        if (h->tr_stage_cap < ntr) {
            if (h->d_tr_stage) (void)hipFree(h->d_tr_stage);
            h->d_tr_stage = nullptr; h->tr_stage_cap = 0;
            HIPCHK(hipMalloc((void**)&h->d_tr_stage, ntr * sizeof(double)));
            h->tr_stage_cap = ntr;
        }
        HIPCHK(hipMemcpyAsync(h->d_tr_stage, h->h_torque_radius.data(), ntr * sizeof(double), hipMemcpyHostToDevice, h->stream));
        d_tr = h->d_tr_stage;
    }
    const int rc = armour_bounds_launch(h, d_tr);
    if (rc != ARMOUR_OK) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));   // (h_torque_radius may change with the next problem set)
    return ARMOUR_OK;
}

extern "C" int armour_set_option(ArmourPlanner* h, int32_t option, double value) {
    if (!h) { armour_set_error("null handle"); return ARMOUR_EINVAL; }
    if (option == ARMOUR_OPT_P1_BUILD && (value == 0.0 || value == 1.0 || value == 2.0)) { h->opt_p1_build = (int)value; return ARMOUR_OK; }
    if (option == ARMOUR_OPT_P1_WORK_MEMORY_MB && value >= 0.0 && value <= 1e9) { h->opt_p1_work_mb = value; return ARMOUR_OK; }
    if (option == ARMOUR_OPT_P1_KEEP_WORK_MEMORY && (value == 0.0 || value == 1.0)) { h->opt_p1_keep_work = (int)value; return ARMOUR_OK; }
    if (const TuningSpec* t = tuning_spec(option)) {
        bool ok = value >= t->lo && value <= t->hi && (option == ARMOUR_OPT_SOLVE_HARD_CAP_S || value == (double)(long long)value);
        if (ok && (option == ARMOUR_OPT_P1_STEP_TAIL_CROSS || option == ARMOUR_OPT_P1_TV_TAIL_CROSS)) {
            // 0 off | 1..4 links on the fourth wave | 11..14 links on the angular wave; + 100 * (n + 1), n = 0..4: the development form
            // that holds the tail moments to n links.  Anything else has no meaning and is refused.
            const int v = (int)value, lo2 = v % 100, hi2 = v / 100;
            ok = (lo2 <= 4 || (lo2 >= 10 && lo2 <= 14)) && hi2 <= 5;
        }
        if (ok && option == ARMOUR_OPT_P1_TV_ROW_WIDTH) ok = value == 0.0 || value == 50.0 || value == 64.0;
        if (ok && option == ARMOUR_OPT_P1_STEP_TWO_CU) { const int lv = (int)value % 10; ok = lv <= 3 && (value < 10.0 || lv >= 1); if (ok) h->p1_two_cu_off = false; }   // (setting it again undoes a fall-back the handle had taken)
        if (ok) {
            h->tuning[option - ARMOUR_OPT_FIRST_TUNING] = value;
            h->p1_step_cap_hint = 0; h->p1_tv_shape_hint = 0;   // (an option may re-enable a block shape the hints had moved past)
            return ARMOUR_OK;
        }
    }
    armour_set_error("armour_set_option: unknown option %d or bad value %g", option, value);
    return ARMOUR_EINVAL;
}

extern "C" int armour_abi_max_factors(void) { return ARMOUR_MAX_FACTORS; }

extern "C" int armour_device_memory(int32_t device, uint64_t* free_bytes, uint64_t* total_bytes) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { (void)hipGetLastError(); armour_set_error("armour_device_memory: no such device %d", device); return ARMOUR_EDEVICE; }
    HIPCHK(hipSetDevice(device));
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return ARMOUR_OK;
}

extern "C" int armour_get_option(ArmourPlanner* h, int32_t option, double* value) {
    if (!h || !value) { armour_set_error("null argument"); return ARMOUR_EINVAL; }
    if (option == ARMOUR_OPT_P1_BUILD) { *value = h->opt_p1_build; return ARMOUR_OK; }
    if (option == ARMOUR_OPT_P1_WORK_MEMORY_MB) { *value = h->opt_p1_work_mb; return ARMOUR_OK; }
    if (option == ARMOUR_OPT_P1_KEEP_WORK_MEMORY) { *value = h->opt_p1_keep_work; return ARMOUR_OK; }
    if (tuning_spec(option)) { *value = h->tuning[option - ARMOUR_OPT_FIRST_TUNING]; return ARMOUR_OK; }
    armour_set_error("armour_get_option: unknown option %d", option);
    return ARMOUR_EINVAL;
}

// ---- reduced outputs: the row test of finalize_solution on the device (RT/NLPclass.cu:422-538; CMP/NLPclass.cu:391-402) ----
namespace {
struct ViolArgs {
    int m, row0, Q, n_checked;   // rows per problem; first collision row; collision rows; how many of them the verdict re-checks
    double torque_slack, collision_slack;
    const double* g; const double* lo; const double* hi;   // [B][m] each
    ArmourViolation* out;                                   // [B]
};
// One 256-thread block per problem.  Thread t takes rows t, t + 256, ... in ascending order and the partial records are
// combined by a fixed tree, so a record depends on (problem, k) alone.  The outside-the-slack test repeats
// armour_check_feasible's expressions on the uploaded bounds: (g_l - slack), (g_u + slack) are the host's
// (-limit + radius) - slack and (limit - radius) + slack.
__global__ __launch_bounds__(256) void armour_violation_kernel(ViolArgs a) {
    __shared__ double s_l1[256], s_w[256];
    __shared__ int s_row[256], s_nv[256], s_no[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const double* g = a.g + (size_t)b * a.m;
    const double* lo = a.lo + (size_t)b * a.m;
    const double* hi = a.hi + (size_t)b * a.m;
    double l1 = 0.0, worst = 0.0;
    int wrow = -1, nv = 0, no = 0;
    for (int r = tid; r < a.m; r += 256) {
        const double v = g[r], l = lo[r], u = hi[r];
        const double viol = fmax(0.0, fmax(l - v, v - u));
        l1 += viol;
        if (viol > 0.0) nv++;
        if (viol > worst) { worst = viol; wrow = r; }
        bool outside;
        if (r < a.row0) outside = v < l - a.torque_slack || v > u + a.torque_slack;
        else if (r < a.row0 + a.Q) outside = (r - a.row0) < a.n_checked && v > a.collision_slack;
        else outside = v < l || v > u;
        if (outside) no++;
    }
    s_l1[tid] = l1; s_w[tid] = worst; s_row[tid] = wrow; s_nv[tid] = nv; s_no[tid] = no;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            s_l1[tid] += s_l1[tid + s]; s_nv[tid] += s_nv[tid + s]; s_no[tid] += s_no[tid + s];
            const double ow = s_w[tid + s];
            const int orow = s_row[tid + s];
            // the larger violation wins; among equals the lower row
            if (ow > s_w[tid] || (ow == s_w[tid] && orow >= 0 && (s_row[tid] < 0 || orow < s_row[tid]))) { s_w[tid] = ow; s_row[tid] = orow; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        ArmourViolation o;
        o.l1_violation = s_l1[0]; o.worst = s_w[0]; o.worst_row = s_row[0]; o.n_violated = s_nv[0]; o.n_outside_slack = s_no[0];
        o.feasible = s_no[0] == 0 ? 1 : 0;
        a.out[b] = o;
    }
}
}  // namespace

// allow_cull = false: every row whatever ARMOUR_OPT_CULL_ROWS says (the host entry's second pass for a k outside the box)
static int eval_violations_device_impl(ArmourPlanner* h, const double* d_k, ArmourViolation* d_out, void* stream, bool allow_cull) {
    NEED_READY(h);
    if (!d_k || !d_out) { armour_set_error("null argument"); return ARMOUR_EINVAL; }
    HIPCHK(hipSetDevice(h->device));
    int rc = armour_upload_bounds(h);
    if (rc != ARMOUR_OK) return rc;
    const hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    if (allow_cull && h->tune(ARMOUR_OPT_CULL_ROWS)) return armour_eval_violations_culled(h, d_k, d_out, st);   // the relevant rows only: relevance.hip
    const P2Tables tb = armour_make_tables(h);
    // g only (the Jacobian tile, 7/8 of the output bytes, is neither computed nor written), into the handle's own g buffer
    rc = armour_p2_launch(tb, h->max_link, h->max_torque, h->h_plane_skip.data(), d_k, h->d_g, nullptr, st);
    if (rc != ARMOUR_OK) return rc;
    ViolArgs a;
    a.m = h->m; a.row0 = h->row0; a.Q = h->Q; a.n_checked = armour_checked_collision_rows(h);
    a.torque_slack = h->params.torque_violation_threshold; a.collision_slack = h->params.collision_violation_threshold;
    a.g = h->d_g; a.lo = h->d_bounds; a.hi = h->d_bounds + (size_t)h->B * h->m; a.out = d_out;
    hipLaunchKernelGGL(armour_violation_kernel, dim3(h->B), dim3(256), 0, st, a);
    HIPCHK(hipGetLastError());
    return ARMOUR_OK;
}

extern "C" int armour_eval_violations_device(ArmourPlanner* h, const double* d_k, ArmourViolation* d_out, void* stream) {
    return eval_violations_device_impl(h, d_k, d_out, stream, true);
}

extern "C" int armour_eval_violations(ArmourPlanner* h, const double* k, ArmourViolation* out) {
    NEED_READY(h);
    if (!k || !out) { armour_set_error("null argument"); return ARMOUR_EINVAL; }
    HIPCHK(hipSetDevice(h->device));
    const size_t bn = (size_t)h->B * h->n;
    if ((size_t)h->B > h->viol_cap) {
        dev_free(&h->d_viol); h->viol_cap = 0;
        int rc = dev_alloc(&h->d_viol, (size_t)h->B);
        if (rc != ARMOUR_OK) return rc;
        h->viol_cap = (size_t)h->B;
    }
    // page-locked staging on both sides: the two copies are asynchronous and ordered with the launches on the handle's stream
    double* hk = armour_handle_pinned(h, 6, bn * sizeof(double));
    ArmourViolation* hv = reinterpret_cast<ArmourViolation*>(armour_handle_pinned(h, 7, (size_t)h->B * sizeof(ArmourViolation)));
    if (!hk || !hv) return ARMOUR_EDEVICE;
    memcpy(hk, k, bn * sizeof(double));
    HIPCHK(hipMemcpyAsync(h->d_k, hk, bn * sizeof(double), hipMemcpyHostToDevice, h->stream));
    // The relevant-rows form (ARMOUR_OPT_CULL_ROWS) holds for k inside [-1, 1]^n -- its mask says "never violated for a k of the box" (relevance.hip).
    // A call with any component outside the box takes every row instead: the records then are the full evaluation's, as the header promises.
    bool in_box = true;
    for (size_t i = 0; i < bn && in_box; i++) in_box = fabs(k[i]) <= 1.0;
    int rc = eval_violations_device_impl(h, h->d_k, h->d_viol, h->stream, in_box);
    if (rc != ARMOUR_OK) return rc;
    HIPCHK(hipMemcpyAsync(hv, h->d_viol, (size_t)h->B * sizeof(ArmourViolation), hipMemcpyDeviceToHost, h->stream));
    rc = spin_on_stream(h->stream);
    if (rc != ARMOUR_OK) return rc;
    memcpy(out, hv, (size_t)h->B * sizeof(ArmourViolation));
    return ARMOUR_OK;
}

extern "C" int armour_get_torque_radius(ArmourPlanner* h, double* out) {
    NEED_READY(h);
    memcpy(out, h->h_torque_radius.data(), h->h_torque_radius.size() * sizeof(double));
    return ARMOUR_OK;
}

extern "C" int armour_get_link_generators(ArmourPlanner* h, double* out) {
    NEED_READY(h);
    memcpy(out, h->h_link_gens.data(), h->h_link_gens.size() * sizeof(double));
    return ARMOUR_OK;
}

extern "C" int armour_get_link_centers(ArmourPlanner* h, const double* k, double* centers) {
    NEED_READY(h);
    HIPCHK(hipSetDevice(h->device));
    const size_t bn = (size_t)h->B * h->n, cnt = (size_t)h->B * h->T * h->J * 3;
    HIPCHK(hipMemcpyAsync(h->d_k, k, bn * sizeof(double), hipMemcpyHostToDevice, h->stream));
    const P2Tables tb = armour_make_tables(h);
    // d_jac holds max(B*m*n, B*T*J*3) doubles (ensure_capacity): reuse it as scratch
    int rc = armour_p2_slice_links_launch(tb, h->d_k, h->d_jac, h->stream);
    if (rc != ARMOUR_OK) return rc;
    HIPCHK(hipMemcpyAsync(centers, h->d_jac, cnt * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return ARMOUR_OK;
}

extern "C" int armour_get_pz(ArmourPlanner* h, int32_t b, int32_t which, int32_t i, int32_t t, double* center,
                             uint64_t* keys, double* coeffs, int32_t capacity) {
    NEED_READY(h);
    if (b < 0 || b >= h->B || t < 0 || t >= h->T || which < 0 || which > 1 || i < 0 || i >= (which == 0 ? h->J : h->n)) {
        armour_set_error("armour_get_pz: index out of range");
        return ARMOUR_EINVAL;
    }
    if (which == 1 && h->no_torque()) {  // the comparison planner -- and ARMOUR without input constraints -- has no torque PZs: empty, centred at 0
        if (center) center[0] = center[1] = 0.0;
        return 0;
    }
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const int sz = which == 0 ? 3 : 1;
    const int cap = which == 0 ? h->lim.link_monomials : h->lim.torque_monomials;
    const size_t idx = which == 0 ? ((size_t)b * h->J + i) * h->T + t : ((size_t)b * h->n + i) * h->T + t;
    int cnt = 0;
    HIPCHK(hipMemcpy(&cnt, (which == 0 ? h->d_link_count : h->d_tq_count) + idx, sizeof(int), hipMemcpyDeviceToHost));
    if (center) {
        // [0..sz) centre, [sz..2sz) independent radius
        HIPCHK(hipMemcpy(center, (which == 0 ? h->d_link_center : h->d_tq_center) + idx * sz, sz * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(center + sz, (which == 0 ? h->d_link_indep : h->d_tq_indep) + idx * sz, sz * sizeof(double), hipMemcpyDeviceToHost));
    }
    if (keys && coeffs) {
        if (capacity < cnt) { armour_set_error("armour_get_pz: capacity %d < count %d", capacity, cnt); return ARMOUR_EINVAL; }
        std::vector<uint32_t> k32(cnt > 0 ? cnt : 1);
        if (cnt > 0) {
            HIPCHK(hipMemcpy(k32.data(), (which == 0 ? h->d_link_keys : h->d_tq_keys) + idx * cap, cnt * sizeof(uint32_t), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(coeffs, (which == 0 ? h->d_link_coeff : h->d_tq_coeff) + idx * cap * sz, (size_t)cnt * sz * sizeof(double), hipMemcpyDeviceToHost));
        }
        for (int mo = 0; mo < cnt; mo++) keys[mo] = k32[mo];
    }
    return cnt;
}

extern "C" int armour_get_table_sizes(ArmourPlanner* h, int64_t* out4) {
    NEED_READY(h);
    out4[0] = h->sum_link; out4[1] = h->sum_torque; out4[2] = h->max_link; out4[3] = h->max_torque;
    return ARMOUR_OK;
}

extern "C" int armour_get_hyperplanes(ArmourPlanner* h, double* A, double* d, double* delta) {
    NEED_READY(h);
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const int B = h->B, T = h->T, J = h->J, O = h->O;
    const size_t ppp = armour_planes_per_problem(h->Q);
    std::vector<double> pl(h->Q > 0 ? (size_t)B * ppp : 0);
    if (pl.empty()) return ARMOUR_OK;
    if (h->planes_lean) {
        // the resident table holds only what the fused evaluation reads: build the full one, with the same kernel code, into a scratch buffer
        double* d_full = nullptr;
        HIPCHK(hipMalloc((void**)&d_full, pl.size() * sizeof(double)));
        int rc = armour_p1_full_planes(h, d_full);
        if (rc == ARMOUR_OK && hipMemcpy(pl.data(), d_full, pl.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) { armour_set_error("copy of the half-space table failed"); rc = ARMOUR_EDEVICE; }
        (void)hipFree(d_full);
        if (rc != ARMOUR_OK) return rc;
    } else
    HIPCHK(hipMemcpy(pl.data(), h->d_planes, pl.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int b = 0; b < B; b++)
        for (int t = 0; t < T; t++)
            for (int l = 0; l < J; l++)
                for (int o = 0; o < O; o++)
                    for (int p = 0; p < 36; p++) {
                        const size_t dst = ((((size_t)b * T + t) * J + l) * O + o) * 36 + p;
                        const int q = (l * T + t) * O + o;
                        const double* base = &pl[(size_t)b * ppp];
                        if (A) { A[dst * 3 + 0] = base[armour_plane_index(h->Q, q, p, 0)]; A[dst * 3 + 1] = base[armour_plane_index(h->Q, q, p, 1)]; A[dst * 3 + 2] = base[armour_plane_index(h->Q, q, p, 2)]; }
                        if (d) d[dst] = base[armour_plane_index(h->Q, q, p, 3)];
                        if (delta) delta[dst] = base[armour_plane_index(h->Q, q, p, 4)];
                    }
    return ARMOUR_OK;
}

extern "C" int armour_debug_pz_op(ArmourPlanner* h, int32_t op, int32_t nops, const int32_t* sz, const int32_t* cnt, const uint64_t* const* keys,
                                  const double* const* coef, const double* cen, const double* ind, const double* ind2, const double* consts,
                                  int32_t r, int32_t out_cap, uint64_t* out_keys, double* out_coef, double* out_misc) {
    if (!h || nops < 1 || nops > 3 || op < 0 || op > 11) { armour_set_error("armour_debug_pz_op: bad argument"); return ARMOUR_EINVAL; }
    for (int o = 0; o < nops; o++)
        if ((sz[o] != 1 && sz[o] != 3 && sz[o] != 9) || cnt[o] < 0 || cnt[o] > h->lim.work_monomials || (sz[o] == 9 && o == 1 && cnt[o] > 8)) {
            armour_set_error("armour_debug_pz_op: bad operand"); return ARMOUR_EINVAL;
        }
    return armour_p1_debug_pz_op(h, op, nops, sz, cnt, keys, coef, cen, ind, ind2, consts, r, out_cap, out_keys, out_coef, out_misc);
}

extern "C" int armour_get_plane_skip(ArmourPlanner* h, uint64_t* plane_skip) {
    NEED_READY(h);
    if (!plane_skip) { armour_set_error("null argument"); return ARMOUR_EINVAL; }
    for (int b = 0; b < h->B; b++) plane_skip[b] = (uint64_t)h->h_plane_skip[b] & ((1ull << ARMOUR_NPLANES) - 1ull);
    return ARMOUR_OK;
}

extern "C" int armour_get_prune_margin(ArmourPlanner* h, double* margin) {
    NEED_READY(h);
    if (!margin) { armour_set_error("null argument"); return ARMOUR_EINVAL; }
    if ((int)h->h_prune_margin.size() != h->B) { armour_set_error("no reach-set build on this handle (tables loaded by armour_debug_load_tables carry no prune margin)"); return ARMOUR_ESTATE; }
    for (int b = 0; b < h->B; b++) margin[b] = h->h_prune_margin[(size_t)b];
    return ARMOUR_OK;
}

extern "C" int armour_get_build_ms(ArmourPlanner* h, double* ms) {
    NEED_READY(h);
    *ms = h->build_ms;
    return ARMOUR_OK;
}

extern "C" int armour_get_build_info(ArmourPlanner* h, int32_t* out4) {
    NEED_READY(h);
    if (!out4) { armour_set_error("null argument"); return ARMOUR_EINVAL; }
    for (int i = 0; i < 4; i++) out4[i] = h->build_info[i];
    return ARMOUR_OK;
}
