// armour_robust_controller: the reference's tracking controller for a BATCH of states -- nominal and interval passivity
// RNEA and the ARMOUR robust input.
//
// Replaces kinova_controller.cpp:19-84 (the MEX gateway: builds the model and a RobustController, calls update once)
// for B states at a time: one thread per state, everything in registers / thread-local memory.  The model preparation
// (model file -> CoM frames -> interval model, robot_models.cpp:124-255) is host arithmetic done once per call from
// the ArmourRobot constants; the per-state arithmetic is controller_core.h.
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "controller_core.h"

namespace {

using namespace ctl;

void rpy(double roll, double pitch, double yaw, double* c) {  // RT/PZsparse.cu:160-176 (the same convention as the planner)
    c[0] = cos(pitch) * cos(yaw);
    c[1] = -cos(pitch) * sin(yaw);
    c[2] = sin(pitch);
    c[3] = cos(roll) * sin(yaw) + cos(yaw) * sin(pitch) * sin(roll);
    c[4] = cos(roll) * cos(yaw) - sin(pitch) * sin(roll) * sin(yaw);
    c[5] = -cos(pitch) * sin(roll);
    c[6] = sin(roll) * sin(yaw) - cos(roll) * cos(yaw) * sin(pitch);
    c[7] = cos(yaw) * sin(roll) + cos(roll) * sin(pitch) * sin(yaw);
    c[8] = cos(pitch) * cos(roll);
}

// The model file of the reference (kinova_without_gripper.txt) holds, per joint: the joint twist (rotation about the
// joint's own axis), the spatial inertia at the joint frame, the parent-to-joint transform and the CoM offset.  The
// same quantities follow from ArmourRobot; Model::Model then re-expresses everything in CoM frames.
int build_models(const ArmourRobot& rb, double eps, Model<double>& md, Model<Itv>& imd) {
    const int n = rb.num_factors;
    if (n < 1 || n > ARMOUR_MAX_FACTORS) return -1;
    Tw<double> S[ARMOUR_MAX_FACTORS];
    Ri<double> I[ARMOUR_MAX_FACTORS];
    Xf<double> XT[ARMOUR_MAX_FACTORS], CoM[ARMOUR_MAX_FACTORS];
    for (int i = 0; i < n; i++) {
        const int ax = std::abs(rb.axes[i]);
        if (ax < 1 || ax > 3) return -1;
        S[i].w = vzero<double>(); S[i].v = vzero<double>();
        S[i].w.x[ax - 1] = rb.axes[i] > 0 ? 1.0 : -1.0;
        // RigidInertia(m, c, Ic), spatial.cpp:123-131
        V3<double> c{{rb.com[3 * i], rb.com[3 * i + 1], rb.com[3 * i + 2]}};
        M3<double> Ic;
        for (int e = 0; e < 9; e++) Ic.a[e] = rb.inertia[9 * i + e];
        const M3<double> ch = hat(c);
        I[i].m = rb.mass[i];
        I[i].m_c_hat = lscale(rb.mass[i], ch);
        I[i].I_bar = Ic - I[i].m_c_hat * ch;
        // parent-to-joint transform: E = R_rpy^T, r = trans (Featherstone's X = (E, r))
        double R[9];
        rpy(rb.rots[3 * i], rb.rots[3 * i + 1], rb.rots[3 * i + 2], R);
        M3<double> Rm;
        for (int e = 0; e < 9; e++) Rm.a[e] = R[e];
        XT[i].R = tr(Rm);
        XT[i].p = V3<double>{{rb.trans[3 * i], rb.trans[3 * i + 1], rb.trans[3 * i + 2]}};
        CoM[i].R = mident<double>();
        CoM[i].p = c;
    }
    md.n = n;
    for (int i = 0; i < n; i++) {  // robot_models.cpp:133-155
        Xf<double> Xwj = XT[i];
        for (int pind = i - 1; pind > -1; pind--) Xwj = apply(Xwj, XT[pind]);
        md.S_[i] = invapply(Xwj, S[i]);
        md.I[i] = apply(CoM[i], I[i]);
        const Xf<double> prev = i > 0 ? CoM[i - 1] : xf_identity<double>();
        md.XTree[i] = apply(prev, apply(inverse(XT[i]), inverse(CoM[i])));
        md.transI[i] = rb.armature[i];
        md.damping[i] = rb.damping[i];
        md.friction[i] = rb.friction[i];
    }
    md.gravity.w = vzero<double>();
    md.gravity.v = V3<double>{{0.0, 0.0, -rb.gravity}};
    // IntModel(model, eps), robot_models.cpp:176-255: point intervals, then mass and I_bar widened by 1 -+ eps
    imd.n = n;
    const double lowP = 1 - eps, highP = 1 + eps;
    for (int i = 0; i < n; i++) {
        for (int e = 0; e < 3; e++) { imd.S_[i].w.x[e] = Itv{md.S_[i].w.x[e], md.S_[i].w.x[e]}; imd.S_[i].v.x[e] = Itv{md.S_[i].v.x[e], md.S_[i].v.x[e]}; imd.XTree[i].p.x[e] = Itv{md.XTree[i].p.x[e], md.XTree[i].p.x[e]}; }
        for (int e = 0; e < 9; e++) {
            imd.XTree[i].R.a[e] = Itv{md.XTree[i].R.a[e], md.XTree[i].R.a[e]};
            imd.I[i].m_c_hat.a[e] = Itv{md.I[i].m_c_hat.a[e], md.I[i].m_c_hat.a[e]};
            const double val = md.I[i].I_bar.a[e];
            imd.I[i].I_bar.a[e] = val >= 0 ? Itv{val * lowP, val * highP} : Itv{val * highP, val * lowP};
        }
        imd.I[i].m = Itv{md.I[i].m * lowP, md.I[i].m * highP};
        imd.transI[i] = Itv{md.transI[i], md.transI[i]};
        imd.damping[i] = md.damping[i];
        imd.friction[i] = md.friction[i];
    }
    for (int e = 0; e < 3; e++) { imd.gravity.w.x[e] = Itv{0.0, 0.0}; imd.gravity.v.x[e] = Itv{md.gravity.v.x[e], md.gravity.v.x[e]}; }
    return 0;
}

struct CtlArgs {
    Model<double> md;
    Model<Itv> imd;
    double Kr[ARMOUR_MAX_FACTORS];
    double alpha, V_max, r_norm_threshold;
};

__global__ __launch_bounds__(64) void armour_controller_kernel(const CtlArgs* __restrict__ ap, int B, const double* __restrict__ q, const double* __restrict__ qd,
                                                               const double* __restrict__ q_des, const double* __restrict__ qd_des,
                                                               const double* __restrict__ qdd_des, double* __restrict__ u, double* __restrict__ tau,
                                                               double* __restrict__ v, int* __restrict__ status) {
    // the models go to LDS first: a lone state is one lane working through ~10^4 dependent operations, and every model
    // entry read from global memory would be a full-latency miss on that chain
    __shared__ CtlArgs sa;
    static_assert(sizeof(CtlArgs) % sizeof(double) == 0, "CtlArgs is copied as doubles");
    for (unsigned i2 = threadIdx.x; i2 < sizeof(CtlArgs) / sizeof(double); i2 += blockDim.x)
        reinterpret_cast<double*>(&sa)[i2] = reinterpret_cast<const double*>(ap)[i2];
    __syncthreads();
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const CtlArgs& a = sa;
    const int n = a.md.n;
    double lq[ARMOUR_MAX_FACTORS], lqd[ARMOUR_MAX_FACTORS], lqdes[ARMOUR_MAX_FACTORS], lqddes[ARMOUR_MAX_FACTORS], lqdddes[ARMOUR_MAX_FACTORS];
    double lu[ARMOUR_MAX_FACTORS], lt[ARMOUR_MAX_FACTORS], lv[ARMOUR_MAX_FACTORS];
    for (int i = 0; i < n; i++) {
        lq[i] = q[(size_t)b * n + i]; lqd[i] = qd[(size_t)b * n + i];
        lqdes[i] = q_des[(size_t)b * n + i]; lqddes[i] = qd_des[(size_t)b * n + i]; lqdddes[i] = qdd_des[(size_t)b * n + i];
    }
    const bool ok = robust_update(a.md, a.imd, a.Kr, a.alpha, a.V_max, a.r_norm_threshold, lq, lqd, lqdes, lqddes, lqdddes, lu, lt, lv);
    for (int i = 0; i < n; i++) { u[(size_t)b * n + i] = lu[i]; tau[(size_t)b * n + i] = lt[i]; v[(size_t)b * n + i] = lv[i]; }
    if (!ok) atomicOr(status, 1);
}

// Latency form for few states -- the simulator's own call pattern is ONE state per ODE step (kinova_controller.cpp:19-84), and one lane
// working through robust_update is 1.35 ms of dependent interval arithmetic.  A block of four waves, lane = state in every wave:
//   wave 0  the nominal RNEA, and at the end the robust input from everybody's torques;
//   wave 3  the interval kinematics (Xli_i, Sb_i), published joint by joint through LDS;
//   wave 1  the interval dynamics of the torque enclosure, wave 2 those of M r, both reading the kinematics as they appear.
// Every quantity is computed by the same code in the same order as in the one-thread form: results are bit-identical
// (tests/test_controller.py).  Lanes past the last state shadow it, so that every wave runs its loops whole.  16 states per block: the
// kinematics of 64 would need 130 KB of LDS (and the first version, with a 154 KB dynamic allocation, ended in a memory-aperture violation
// that the core dump did not explain); with 16 everything is static and below 64 KB.
#define CTL_LDS __attribute__((address_space(3)))
constexpr int kSplitStates = 16;
constexpr int kKinWords = 18;   // Xli: R 9 + p 3, Sb: w 3 + v 3 intervals per joint
struct SplitLds {
    double u_nom[ARMOUR_MAX_FACTORS][kSplitStates];
    double u_int[ARMOUR_MAX_FACTORS][2][kSplitStates], Mr[ARMOUR_MAX_FACTORS][2][kSplitStates];   // intervals as {lo, hi} planes
    double kin[ARMOUR_MAX_FACTORS][kKinWords][2][kSplitStates];
    int joints_published;
};
__device__ inline void lds_put(CTL_LDS double (*p)[kSplitStates], int lane, Itv v) { p[0][lane] = v.lo; p[1][lane] = v.hi; }
__device__ inline Itv lds_get(const CTL_LDS double (*p)[kSplitStates], int lane) { return Itv{p[0][lane], p[1][lane]}; }
__global__ __launch_bounds__(256) void armour_controller_split_kernel(const CtlArgs* __restrict__ ap, int B, const double* __restrict__ q,
                                                                      const double* __restrict__ qd, const double* __restrict__ q_des,
                                                                      const double* __restrict__ qd_des, const double* __restrict__ qdd_des,
                                                                      double* __restrict__ u, double* __restrict__ tau, double* __restrict__ v,
                                                                      int* __restrict__ status) {
    __shared__ CtlArgs sa;
    __shared__ SplitLds xs;
    CTL_LDS SplitLds& x = *(CTL_LDS SplitLds*)&xs;   // (LDS-typed: ds instructions, no generic pointers into the exchange area)
    for (unsigned i2 = threadIdx.x; i2 < sizeof(CtlArgs) / sizeof(double); i2 += blockDim.x)
        reinterpret_cast<double*>(&sa)[i2] = reinterpret_cast<const double*>(ap)[i2];
    if (threadIdx.x == 0) x.joints_published = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b0 = blockIdx.x * kSplitStates + lane;
    const bool active = lane < kSplitStates && b0 < B;
    const int slot = lane < kSplitStates ? lane : kSplitStates - 1;   // (idle lanes shadow the last state of the block; only lanes < kSplitStates write)
    const bool own = lane < kSplitStates;
    const int b = min(blockIdx.x * kSplitStates + slot, B - 1);
    const CtlArgs& a = sa;
    const int n = a.md.n;
    double lq[ARMOUR_MAX_FACTORS], lqd[ARMOUR_MAX_FACTORS], lqdes[ARMOUR_MAX_FACTORS], lqddes[ARMOUR_MAX_FACTORS], lqdddes[ARMOUR_MAX_FACTORS];
    double qa_d[ARMOUR_MAX_FACTORS], qa_dd[ARMOUR_MAX_FACTORS], r[ARMOUR_MAX_FACTORS];
    for (int i = 0; i < n; i++) {
        lq[i] = q[(size_t)b * n + i]; lqd[i] = qd[(size_t)b * n + i];
        lqdes[i] = q_des[(size_t)b * n + i]; lqddes[i] = qd_des[(size_t)b * n + i]; lqdddes[i] = qdd_des[(size_t)b * n + i];
    }
    const double r_norm = robust_prepare(n, a.Kr, lq, lqd, lqdes, lqddes, lqdddes, qa_d, qa_dd, r);   // (every wave for itself: a few dozen operations)
    // the kinematics of joint i are in LDS once joints_published > i
    CTL_LDS SplitLds* const xp = &x;   // (the lambdas below capture by value: nothing of this frame is reached through a pointer)
    auto get = [xp, slot](int i, Xf<Itv>& Xli, Tw<Itv>& Sb) {
        CTL_LDS SplitLds& x = *xp;
        while (__hip_atomic_load(&x.joints_published, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= i) __builtin_amdgcn_s_sleep(1);
        for (int e = 0; e < 9; e++) Xli.R.a[e] = lds_get(x.kin[i][e], slot);
        for (int e = 0; e < 3; e++) { Xli.p.x[e] = lds_get(x.kin[i][9 + e], slot); Sb.w.x[e] = lds_get(x.kin[i][12 + e], slot); Sb.v.x[e] = lds_get(x.kin[i][15 + e], slot); }
    };
    if (wave == 0) {
        double t[ARMOUR_MAX_FACTORS];
        pass_rnea<double>(a.md, lq, lqd, qa_d, qa_dd, false, true, t);
        if (own) for (int i = 0; i < n; i++) x.u_nom[i][lane] = t[i];
    } else if (wave == 3) {
        rnea_kinematics(a.imd, lq, [xp, own, lane](int i, const Xf<Itv>& Xli, const Tw<Itv>& Sb) {
            CTL_LDS SplitLds& x = *xp;
            if (own) {
                for (int e = 0; e < 9; e++) lds_put(x.kin[i][e], lane, Xli.R.a[e]);
                for (int e = 0; e < 3; e++) { lds_put(x.kin[i][9 + e], lane, Xli.p.x[e]); lds_put(x.kin[i][12 + e], lane, Sb.w.x[e]); lds_put(x.kin[i][15 + e], lane, Sb.v.x[e]); }
            }
            // (the wave runs in lockstep: lane 0's release store follows every lane's writes of this joint)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_store(&x.joints_published, i + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        });
    } else if (wave == 1) {
        Itv t[ARMOUR_MAX_FACTORS];
        rnea_dynamics(a.imd, get, lqd, qa_d, qa_dd, false, true, t);
        if (own) for (int i = 0; i < n; i++) lds_put(x.u_int[i], lane, t[i]);
    } else if (__ballot(r_norm > a.r_norm_threshold) != 0) {   // (no lane needs M r: nothing to do)
        Itv t[ARMOUR_MAX_FACTORS];
        double zero[ARMOUR_MAX_FACTORS];
        for (int i = 0; i < n; i++) zero[i] = 0.0;
        rnea_dynamics(a.imd, get, zero, zero, r, false, false, t);
        if (own) for (int i = 0; i < n; i++) lds_put(x.Mr[i], lane, t[i]);
    }
    __syncthreads();
    if (wave == 0 && active) {
        double lt[ARMOUR_MAX_FACTORS], lu[ARMOUR_MAX_FACTORS], lv[ARMOUR_MAX_FACTORS];
        Itv ui[ARMOUR_MAX_FACTORS], mr[ARMOUR_MAX_FACTORS];
        for (int i = 0; i < n; i++) { lt[i] = x.u_nom[i][lane]; ui[i] = lds_get(x.u_int[i], lane); mr[i] = lds_get(x.Mr[i], lane); }
        const bool ok = robust_combine(n, a.alpha, a.V_max, a.r_norm_threshold, r, r_norm, lt, ui, mr, lu, lv);
        for (int i = 0; i < n; i++) { u[(size_t)b * n + i] = lu[i]; tau[(size_t)b * n + i] = lt[i]; v[(size_t)b * n + i] = lv[i]; }
        if (!ok) atomicOr(status, 1);
    }
}

// Per-thread cache across calls: the MEX gateway this replaces is called once per ODE step with ONE state, so the fixed
// cost of a call matters more than the kernel.  Kept: the prepared models on the device (rebuilt only when the robot or the
// controller parameters change), the device buffer, and a page-locked staging buffer so that a small call is one H2D
// copy, one launch and one D2H copy on a private stream (first version: three hipMalloc / hipFree and ten blocking
// copies per call, 1.7 ms; now tens of microseconds).
struct CtlCache {
    bool have_model = false;
    ArmourRobot rb;
    double eps = 0, Kr[ARMOUR_MAX_FACTORS] = {0}, alpha = 0, V_max = 0, r_thr = 0;
    int n = 0, device = -1;
    CtlArgs* d_args = nullptr;
    double* d_buf = nullptr; size_t d_cap = 0;   // doubles
    double* h_pin = nullptr; size_t h_cap = 0;   // doubles
    hipStream_t stream = nullptr;
    ~CtlCache() {
        // (process exit: the runtime may already be gone; errors are ignored)
        if (d_args) (void)hipFree(d_args);
        if (d_buf) (void)hipFree(d_buf);
        if (h_pin) (void)hipHostFree(h_pin);
        if (stream) (void)hipStreamDestroy(stream);
    }
};
thread_local CtlCache g_ctl;

// spin on the stream instead of sleeping in hipStreamSynchronize: a small call is tens of microseconds and an
// interrupt-driven wake-up would dominate it (same reasoning as armour_eval_g_jac)
int ctl_wait(hipStream_t st) {
    for (;;) {
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) return ARMOUR_OK;
        if (q != hipErrorNotReady) { armour_set_error("hipStreamQuery failed: %s", hipGetErrorString(q)); return ARMOUR_EDEVICE; }
    }
}

constexpr int kSplitMaxStates = kSplitStates * 256;   // up to one four-wave block per CU
constexpr size_t kStagedDoubles = (size_t)1 << 20;  // calls up to this many input doubles go through the page-locked buffer

}  // namespace

// C ABI: see include/armour_hip.h.  Host pointers; B states of n = robot->num_factors joints each, row-major [B][n].
static std::atomic<int> g_ctl_kernel{-1};
extern "C" int armour_controller_set_kernel(int32_t which) {
    if (which < -1 || which > 1) { armour_set_error("armour_controller_set_kernel: -1 (automatic), 0 or 1"); return ARMOUR_EINVAL; }
    g_ctl_kernel.store(which, std::memory_order_relaxed);
    return ARMOUR_OK;
}

extern "C" int armour_robust_controller(const ArmourRobot* robot, double model_uncertainty, const double* Kr, double alpha, double V_max,
                                        double r_norm_threshold, int32_t B, const double* q, const double* qd, const double* q_des,
                                        const double* qd_des, const double* qdd_des, double* u, double* tau, double* v) {
    if (!robot || !Kr || !q || !qd || !q_des || !qd_des || !qdd_des || !u || !tau || !v || B < 1) { armour_set_error("null or empty argument"); return ARMOUR_EINVAL; }
    CtlCache& c = g_ctl;
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    if (c.device != dev) {  // first call of this thread, or the caller switched devices: start over
        if (c.d_args) { (void)hipFree(c.d_args); c.d_args = nullptr; }
        if (c.d_buf) { (void)hipFree(c.d_buf); c.d_buf = nullptr; c.d_cap = 0; }
        if (c.stream) { (void)hipStreamDestroy(c.stream); c.stream = nullptr; }
        c.have_model = false;
        c.device = dev;
    }
    if (!c.stream) HIPCHK(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    if (!c.d_args) HIPCHK(hipMalloc((void**)&c.d_args, sizeof(CtlArgs)));
    const int nf = robot->num_factors;
    bool same = c.have_model && memcmp(&c.rb, robot, sizeof(ArmourRobot)) == 0 && c.eps == model_uncertainty && c.alpha == alpha && c.V_max == V_max &&
                c.r_thr == r_norm_threshold;
    for (int i = 0; same && i < nf; i++) same = c.Kr[i] == Kr[i];
    if (!same) {
        CtlArgs args;
        memset(&args, 0, sizeof(args));
        if (build_models(*robot, model_uncertainty, args.md, args.imd) != 0) { armour_set_error("unsupported robot model (joint axes / count)"); return ARMOUR_EINVAL; }
        for (int i = 0; i < args.md.n; i++) args.Kr[i] = Kr[i];
        args.alpha = alpha; args.V_max = V_max; args.r_norm_threshold = r_norm_threshold;
        HIPCHK(hipMemcpy(c.d_args, &args, sizeof(args), hipMemcpyHostToDevice));
        c.rb = *robot; c.eps = model_uncertainty; c.alpha = alpha; c.V_max = V_max; c.r_thr = r_norm_threshold; c.n = args.md.n;
        for (int i = 0; i < c.n; i++) c.Kr[i] = Kr[i];
        c.have_model = true;
    }
    const int n = c.n;
    const size_t bn = (size_t)B * n, total = 8 * bn + 1;  // 5 inputs | 3 outputs | status word
    if (total > c.d_cap) {
        if (c.d_buf) { (void)hipFree(c.d_buf); c.d_buf = nullptr; c.d_cap = 0; }
        HIPCHK(hipMalloc((void**)&c.d_buf, total * sizeof(double)));
        c.d_cap = total;
    }
    double* d_in = c.d_buf;
    double* d_out = c.d_buf + 5 * bn;
    int* d_status = reinterpret_cast<int*>(c.d_buf + 8 * bn);
    const double* in[5] = {q, qd, q_des, qd_des, qdd_des};
    double* out[3] = {u, tau, v};
    const bool staged = 5 * bn <= kStagedDoubles;
    if (staged && total > c.h_cap) {
        if (c.h_pin) { (void)hipHostFree(c.h_pin); c.h_pin = nullptr; c.h_cap = 0; }
        HIPCHK(hipHostMalloc((void**)&c.h_pin, total * sizeof(double), hipHostMallocDefault));
        c.h_cap = total;
    }
    if (staged) {
        for (int k = 0; k < 5; k++) memcpy(c.h_pin + (size_t)k * bn, in[k], bn * sizeof(double));
        HIPCHK(hipMemcpyAsync(d_in, c.h_pin, 5 * bn * sizeof(double), hipMemcpyHostToDevice, c.stream));
    } else {
        for (int k = 0; k < 5; k++) HIPCHK(hipMemcpyAsync(d_in + (size_t)k * bn, in[k], bn * sizeof(double), hipMemcpyHostToDevice, c.stream));
    }
    HIPCHK(hipMemsetAsync(d_status, 0, sizeof(double), c.stream));
    // few states: four waves per 16 states (latency); many: one lane per state (throughput) -- bit-identical results
    const int split_env = g_ctl_kernel.load(std::memory_order_relaxed);   // armour_controller_set_kernel: -1 automatic, 0 never, 1 always
    const bool split = split_env >= 0 ? split_env != 0 : B <= kSplitMaxStates;
    if (split)
        hipLaunchKernelGGL(armour_controller_split_kernel, dim3((B + kSplitStates - 1) / kSplitStates), dim3(256), 0, c.stream, c.d_args, B, d_in,
                           d_in + bn, d_in + 2 * bn, d_in + 3 * bn, d_in + 4 * bn, d_out, d_out + bn, d_out + 2 * bn, d_status);
    else
        hipLaunchKernelGGL(armour_controller_kernel, dim3((B + 63) / 64), dim3(64), 0, c.stream, c.d_args, B, d_in, d_in + bn, d_in + 2 * bn, d_in + 3 * bn, d_in + 4 * bn,
                           d_out, d_out + bn, d_out + 2 * bn, d_status);
    HIPCHK(hipGetLastError());
    int st = 0;
    if (staged) {
        HIPCHK(hipMemcpyAsync(c.h_pin + 5 * bn, d_out, (3 * bn + 1) * sizeof(double), hipMemcpyDeviceToHost, c.stream));
        if (ctl_wait(c.stream) != ARMOUR_OK) return ARMOUR_EDEVICE;
        for (int k = 0; k < 3; k++) memcpy(out[k], c.h_pin + (5 + (size_t)k) * bn, bn * sizeof(double));
        memcpy(&st, c.h_pin + 8 * bn, sizeof(int));
    } else {
        for (int k = 0; k < 3; k++) HIPCHK(hipMemcpyAsync(out[k], d_out + (size_t)k * bn, bn * sizeof(double), hipMemcpyDeviceToHost, c.stream));
        HIPCHK(hipMemcpyAsync(&st, d_status, sizeof(int), hipMemcpyDeviceToHost, c.stream));
        HIPCHK(hipStreamSynchronize(c.stream));
    }
    if (st) { armour_set_error("nominal model output falls outside interval output (robust_controller.cpp:88-94)"); return ARMOUR_ESTATE; }
    return ARMOUR_OK;
}
