// P1: reach-set build on the device (gfx950), once per planning iteration and problem.
//
// Replaces RT/armour_main.cu:96-216 -- the OpenMP loop over time steps running
// BezierCurve::makePolyZono (RT/Trajectory.cu:63-254), KinematicsDynamics::fk / rnea x2
// (RT/Dynamics.cu:69-181), the disturbance / robust-input radius (RT/armour_main.cu:133-141,172-205) -- and
// Obstacles::initializeHyperPlane (RT/CollisionChecking.cu:69-88,136-228).
//
// Fused nominal / interval RNEA.  The reference runs KinematicsDynamics::rnea twice per time step, with the nominal and
// with the 3 %-uncertain mass / inertia PZs (RT/Dynamics.h:41-47, RT/armour_main.cu:126-129).  Those parameter PZs have
// no monomials -- only a centre and an independent radius (RT/Dynamics.cu:27-40, RT/PZsparse.cu:93-98) -- and no
// operator lets the independent radius flow back into a centre or a monomial coefficient (RT/PZsparse.cu:864-994:
// it only feeds res.independent; simplify() prunes on coefficients alone).  Hence both passes produce bit-identical
// centres, keys and coefficients and differ in the independent radii only.  Here every PZ carries both radii
// (ind = nominal, ind2 = uncertain) and the recursion runs ONCE; u_nom_int - u_nom then has centre 0, all monomials
// cancel exactly and its radius is ind2 + ind, which is what RT/armour_main.cu:133-136,179-191 computes the long way.
//
// Mapping: one 64-lane wavefront per (problem, time interval) runs the whole dependent chain
// JRS -> FK -> link reduce -> RNEA(nominal) -> RNEA(interval) -> disturbance -> torque radius on the
// wave-level PZ arithmetic of pz_wave.h; the grid is persistent (a fixed number of waves, each with a
// private global-memory arena of PZ slots, stride over the (b,t) work list).  A second, trivially parallel
// kernel builds the half-space table one thread per (problem, link, time, obstacle) row.
#include <algorithm>
#include <cstdlib>
#include <chrono>
#include <cmath>
#include <cstring>
#include <functional>
#include <mutex>

#include "common.h"
#include "pz_wave.h"
#include "pz_tv.h"

using namespace pzw;

namespace {

constexpr int kNS = 24, kNM = 2;   // 1x1 and 3x3 work slots per block; 3x1 work slots: Layout::nV (32 for one wave, 48 for three)
// The 3x1 pool is split between the roles of a block (see run_rnea): each role allocates and frees only in its own part,
// so three waves can run different operators at the same time without sharing allocator state.
constexpr int kRoles = 3;
constexpr int kPartFirst[kRoles] = {0, 12, 34}, kPartCount[kRoles] = {12, 22, 28};  // of the 62 slots of a 3-wave block (run_rnea_free keeps a few joints' states alive)
// ... and of a 4-wave block (round 3: the forward kinematics, the omega recursion and the constant cross products on a wave of their own,
// as the time-vectorised kernel's four-wave blocks have had since round 2: p1_tv.inc.h kTvPart4First)
constexpr int kPart4First[4] = {0, 8, 23, 43}, kPart4Count[4] = {8, 15, 20, 20};   // (63 slots: the free mask is one 64-bit word, built as (1 << nV) - 1)
// ... and of the HELPER block of a time step on two CUs (p1_free.inc.h): J products + one temporary on each product wave; 2 (J + 1) states + the forward kinematics on the fourth
// (level 3, J <= 7: the second product wave also builds the angular step's cross product, 2 J products + three temporaries -- Chain::helper_parts)
constexpr int kPartHFirst[4] = {0, 10, 20, 30}, kPartHCount[4] = {10, 10, 10, 33};
static_assert(ARMOUR_MAX_JOINTS + 1 <= 10 && 2 * (ARMOUR_MAX_JOINTS + 1) + 12 <= 33, "helper pool");
constexpr int kTwoCuC4MaxJ = 7;   // 4 J + 4 + 2 (J + 1) + 10 <= 63
// Sort-buffer KEY entries for `cap` raw terms.  A wave's LDS sort buffers are skey[cap_key] (monomial keys) + sidx[cap_raw] (permutation, 2 B each).
// The 64-bit build gives both `cap` entries; with 128-bit keys (pz_key.h, -DARMOUR_KEY128) the same bytes hold half as many keys -- and the
// sorters ask for cap_key and cap_raw separately (pz_wave.h sort_terms: the tree merge needs 2 N keys, the ranked merge the two operands' key
// lists, the bitonic network next_pow2(N)), so a block keeps its shape -- four waves, one block per CU -- with HALF the key entries instead of
// falling back to one-wave blocks (round 4: 7.2 ms for one 8-factor problem).  A product too long for them raises ERR_RAW_OVERFLOW as before.
constexpr int kKeyDiv = (int)(sizeof(pzkey_t) / sizeof(uint64_t));
__host__ __device__ constexpr int key_cap(int cap) { return cap / kKeyDiv; }
constexpr int kFkCapKey = 1024, kFkCapRaw = 2048;   // sort buffers of that fourth wave: its products stay below 0.9 k raw terms; the ranked rotation x vector products of omega need room for the permutation only
constexpr int kRoleN = 2;  // the role that computes (and later frees) the moments N_i (measured with it in role 0 / 1 / 2 and the forward kinematics split off: 2.06 / 2.01 / 1.95 ms)
constexpr int kNVOneWave = 32;  // a 1-wave block plays the roles in turn: one pool, one set of scratch slots
constexpr int kCapSmall = 8;                 // capacity of the JRS / constant slots
constexpr int kMaxSlots = 208;

struct P1Cfg {
    int B, T, J, n, O;
    int capW, capRaw, capKey, capL, capT;
    size_t arena_bytes;
    unsigned char* arena;
    ArmourRobot rb;
    ArmourParams pr;
    ArmourUltimateBound ub;
    const double* bez;  // [B][3][n]: q0, Tqd0, TTqdd0
    int mode;           // ARMOUR_MODE_*; ARMTD: cos/sin JRS from `jrs`, forward kinematics only (no RNEA, no torque tables)
    int fk_only;        // forward kinematics only: ARMTD mode, or the ARMOUR trajectory without input constraints (RT/armour_main.cu:149-165)
    const double* jrs;  // ARMTD mode: [B][n][6][T] centre / k-generator / radius of cos(q - q0), then of sin(q - q0)
    // outputs
    int* link_count; double* link_center; double* link_indep; uint32_t* link_keys; double* link_coeff;
    int* tq_count; double* tq_center; double* tq_indep; uint32_t* tq_keys; double* tq_coeff;
    double* link_gens;      // [B][T][J][18]
    double* torque_radius;  // [B][n][T]
    unsigned* status;
    unsigned long long* margin;   // [B] or nullptr: per problem, (bits of +inf) - bits of the smallest |squared norm - thr_sq| of its simplify() verdicts (grows: atomicMax; pz_wave.h Wave::mg)
    // work list: items[0..n_items) are the (problem, time step) indices b*T + t to build (nullptr: all 0..n_items-1).
    // retry_list != nullptr: an item whose sort buffers overflowed is appended there (its error bits are dropped) for a
    // second launch with larger buffers, instead of failing the whole launch.
    const int* items;
    int n_items;
    // fk_items > 0 (small launches with idle CUs): work items [n_items, n_items + fk_items) redo the JRS rotations of item
    // (it - n_items) and run ONLY its forward kinematics, on a CU of their own; the first n_items then skip it.  The
    // forward kinematics shares nothing with the RNEA but the read-only JRS, and role 2 is the busiest of the three.
    int fk_items;
    int* retry_list;
    unsigned* retry_count;
    // time-vectorised build (p1_tv.inc.h): an item is a (problem, group of time steps) pair
    int tv_free_running;  // three-wave blocks: run_rnea_free (progress counters) instead of run_rnea (a barrier per joint)
    int free_running;     // the same choice for the per-step kernel's three-wave blocks
    int tv_groups, tv_lanes, tv_cap, tv_stage_rows, tv_stage_rows_other;  // staging rows of wave 1 (or of the only wave) / of the other waves
    int tv_help_min, tv_help_n;  // shared walks: smallest walk that is shared; whether the n-recursion shares its walks as well (development switches)
    int tv_aux_on_fk_wave;  // four-wave blocks: the w_aux recursion runs on the forward-kinematics wave (p1_free.inc.h)
    int tv_help_shift;    // eight-wave blocks: role wave p's helper is wave 4 + (p + shift) % 4
    int tv_walk_helpers;  // four-wave blocks: the idle waves of the backward pass walk part of the recursions' operators (pz_tv.h); 0 off, 1 on, n > 1: the primary keeps n / 32 of the terms
    int tail_cross;       // four-wave blocks: w x (w_aux x com) of the last links on a wave that is through with its recursion (p1_free.inc.h): 0 off | links | 10 + links (on wave 1)
    unsigned* queue;      // per-step kernel: the blocks draw their items from this counter (nullptr: block k takes items k, k + blocks, ...)
    int queue_order;      // 1: the late time steps of every problem first (they are the expensive ones) | 2: in index order | 3: the early steps first
    int step_pairs;       // per-step kernel, four-wave blocks: in the backward pass the two idle waves join the recursion waves' operators (pz_wave.h psync)
    // a time step on two CUs (p1_free.inc.h): block b < n_items builds item b, block helper0 + b is its helper; xch: kXchBytes per item
    int two_cu, helper0, xch_epoch;
    long long* phase_log; // ARMOUR_P1_TRACE: [blocks][8] clock of wave 0 at the phase boundaries of its (last) item -- start, JRS, decision, forward pass, backward pass, end (release build: one test of a pointer per boundary)
    int late_jrs;         // two CUs, level 3: the JRS of joints 4 .. J - 1 behind the start of the roles (development switch)
    int lean_back;        // two CUs, level 3: the backward pass with the n-recursion stripped to its chain (p1_free.inc.h run_backward_remote); development switch
    unsigned char* xch;
};

__host__ __device__ inline size_t align64(size_t v) { return (v + 63) & ~(size_t)63; }
__host__ __device__ inline size_t slot_bytes(int cap, int sz) { return align64((size_t)cap * sizeof(pzkey_t)) + align64((size_t)cap * sz * 8) + align64((size_t)sz * 8) * 2; }

// arena = work V | work S | work M | small M | small V | small S
struct Layout {
    int nJM, nJV, nJS;
    int nV, nroles;  // 3x1 work slots; sets of per-role scratch slots (1 or kRoles)
    size_t offV, offS, offM, offJM, offJV, offJS, total;
    int idV, idS, idM, idJM, idJV, idJS;
};
__host__ __device__ inline Layout make_layout(int J, int n, int capW, int nroles) {
    Layout L;
    L.nroles = nroles;
    L.nV = nroles == 1 ? kNVOneWave : nroles == kRoles ? kPartFirst[kRoles - 1] + kPartCount[kRoles - 1] : kPart4First[3] + kPart4Count[3];
    L.nJM = (J + 1) + J + 3 * nroles + J;  // R[0..J], R_t[0..J-1], per role: three scratch slots (raw rot, simplified rot, rpy: unused since the JRS is built in closed form, kept so that slot numbers stay what the profiles and notes refer to); inertia
    L.nJV = (J + 1) + J;              // trans P_i, link boxes
    L.nJS = 3 * n + J + 4 * nroles;   // qd, qda, qdda; mass; per role: 4 raw temps
    L.offV = 0;
    L.offS = L.offV + (size_t)L.nV * slot_bytes(capW, 3);
    L.offM = L.offS + (size_t)kNS * slot_bytes(capW, 1);
    L.offJM = L.offM + (size_t)kNM * slot_bytes(capW, 9);
    L.offJV = L.offJM + (size_t)L.nJM * slot_bytes(kCapSmall, 9);
    L.offJS = L.offJV + (size_t)L.nJV * slot_bytes(kCapSmall, 3);
    L.total = align64(L.offJS + (size_t)L.nJS * slot_bytes(kCapSmall, 1));
    L.idV = 0; L.idS = L.idV + L.nV; L.idM = L.idS + kNS; L.idJM = L.idM + kNM; L.idJV = L.idJM + L.nJM; L.idJS = L.idJV + L.nJV;
    return L;
}

// keys / coefficients of a slot live in the wave's global arena; its centre and independent radius (sz doubles each)
// live in LDS (`ci`), so the per-operator scalar traffic never leaves the CU.
__device__ inline PZ mk_slot(GLB_AS unsigned char* base, size_t off, int index, int cap, int sz, int id0, LDS_AS double* ci) {
    GLB_AS unsigned char* p = base + off + (size_t)index * slot_bytes(cap, sz);
    PZ z;
    z.keys = (GLB_AS pzkey_t*)p;
    z.coef = (GLB_AS double*)(p + align64((size_t)cap * sizeof(pzkey_t)));
    z.cen = ci + (size_t)index * 3 * sz;
    z.ind = z.cen + sz;
    z.ind2 = z.ind + sz;
    z.sz = sz; z.cap = cap; z.id = id0 + index;
    return z;
}

// ------------------------------------------------------------------ device interval helpers (RT/Headers.h:26-36)
struct Itv { double lo, hi; };
__device__ inline double dn(double x) { return nextafter(x, -INFINITY); }
__device__ inline double up(double x) { return nextafter(x, INFINITY); }
__device__ inline Itv iv(double l, double h) { return Itv{l, h}; }
__device__ inline Itv iout(double l, double h) { return Itv{dn(l), up(h)}; }
__device__ inline Itv ineg(Itv a) { return Itv{-a.hi, -a.lo}; }
__device__ inline Itv iadd(Itv a, Itv b) { return iout(a.lo + b.lo, a.hi + b.hi); }
__device__ inline Itv iadds(double a, Itv b) { return iout(a + b.lo, a + b.hi); }
__device__ inline Itv isub(Itv a, Itv b) { return iout(a.lo - b.hi, a.hi - b.lo); }
__device__ inline Itv isubs(Itv a, double b) { return iout(a.lo - b, a.hi - b); }
__device__ inline Itv imul(Itv a, Itv b) {
    const double p0 = a.lo * b.lo, p1 = a.lo * b.hi, p2 = a.hi * b.lo, p3 = a.hi * b.hi;
    return iout(fmin(fmin(p0, p1), fmin(p2, p3)), fmax(fmax(p0, p1), fmax(p2, p3)));
}
__device__ inline Itv imuls(double a, Itv b) { return imul(Itv{a, a}, b); }
__device__ inline Itv isqr(Itv a) {
    if (a.lo >= 0) return iout(a.lo * a.lo, a.hi * a.hi);
    if (a.hi <= 0) return iout(a.hi * a.hi, a.lo * a.lo);
    const double m = fmax(-a.lo, a.hi);
    return Itv{0.0, up(m * m)};
}
__device__ inline Itv icos(Itv a) {
    const double pi = 3.14159265358979323846;
    if (a.hi - a.lo >= 2 * pi) return Itv{-1, 1};
    double lo = fmin(cos(a.lo), cos(a.hi)), hi = fmax(cos(a.lo), cos(a.hi));
    const double kmax = ceil((a.lo - 1e-15) / (2 * pi));
    if (kmax * 2 * pi <= a.hi + 1e-15) hi = 1.0;
    const double kmin = ceil((a.lo - pi - 1e-15) / (2 * pi));
    if (pi + kmin * 2 * pi <= a.hi + 1e-15) lo = -1.0;
    return Itv{fmax(-1.0, dn(lo)), fmin(1.0, up(hi))};
}
__device__ inline Itv isin(Itv a) {
    const double pi = 3.14159265358979323846;
    if (a.hi - a.lo >= 2 * pi) return Itv{-1, 1};
    double lo = fmin(sin(a.lo), sin(a.hi)), hi = fmax(sin(a.lo), sin(a.hi));
    const double kmax = ceil((a.lo - pi / 2 - 1e-15) / (2 * pi));
    if (pi / 2 + kmax * 2 * pi <= a.hi + 1e-15) hi = 1.0;
    const double kmin = ceil((a.lo + pi / 2 - 1e-15) / (2 * pi));
    if (-pi / 2 + kmin * 2 * pi <= a.hi + 1e-15) lo = -1.0;
    return Itv{fmax(-1.0, dn(lo)), fmin(1.0, up(hi))};
}
__device__ inline double icen(Itv a) { return (a.lo + a.hi) * 0.5; }
__device__ inline double irad(Itv a) { return (a.hi - a.lo) * 0.5; }

__device__ inline double p2(double x) { return x * x; }
__device__ inline double p3(double x) { return x * x * x; }
__device__ inline double p4(double x) { const double y = x * x; return y * y; }
__device__ inline double p5(double x) { const double y = x * x; return y * y * x; }

// k-independent parts of the Bezier curve (RT/Trajectory.cu:812-822)
__device__ inline double q_indep(double q0, double a, double b, double s) {
    return q0 + a * s - 6 * a * p3(s) + 8 * a * p4(s) - 3 * a * p5(s) + (b * p2(s)) * 0.5 - (3 * b * p3(s)) * 0.5 + (3 * b * p4(s)) * 0.5 - (b * p5(s)) * 0.5;
}
__device__ inline double qd_indep(double a, double b, double s, double D) {
    return (p2(s - 1) * (2 * a + 4 * a * s + 2 * b * s - 30 * a * p2(s) - 5 * b * p2(s))) * 0.5 / D;
}
__device__ inline double qdd_indep(double a, double b, double s, double D) {
    return -(s - 1.0) * (b - (36 * a + 8 * b) * s + (60 * a + 10 * b) * p2(s)) / (D * D);
}
__device__ inline void bound_ext(double& lb, double& ub_, double s_lb, double s_ub, double x1, double m1, double x2, double m2) {
    if (lb > ub_) { const double t = lb; lb = ub_; ub_ = t; }
    if (s_lb < x1 && x1 < s_ub) { lb = fmin(lb, m1); ub_ = fmax(ub_, m1); }
    if (s_lb < x2 && x2 < s_ub) { lb = fmin(lb, m2); ub_ = fmax(ub_, m2); }
}

// scalars of one joint's JRS on [s/T, (s+1)/T] (RT/Trajectory.cu:63-243)
struct JrsScalars {
    double cos_c, cos_k, cos_e, sin_c, sin_k, sin_e;          // centre, k coefficient, error coefficient
    double qd_c, qd_k, qd_e, qda_e, qdd_c, qdd_k, qdd_e;
};
struct JrsLane { double rp[9]; JrsScalars js; };   // lane i: the fixed rotation and the scalars of joint i (jrs_lane)

__device__ PZW_NOINLINE JrsScalars jrs_scalars(const P1Cfg& cf, double q0, double a, double b, int i, int s_ind) {
    PZ_KEEP_RETURN_ADDRESS();
    JrsScalars o;
    const double ds = 1.0 / cf.T, D = cf.pr.duration, kr = cf.pr.k_range[i];
    const double s_lb = s_ind * ds, s_ub = (s_ind + 1) * ds;
    // k-independent extrema (RT/Trajectory.cu:36-58)
    const double dq = sqrt(64 * p2(a) + 14 * a * b + p2(b)), dv = sqrt(6 * (54 * p2(a) + 14 * a * b + p2(b))), da = sqrt(2 * (152 * p2(a) + 42 * a * b + 3 * p2(b)));
    const double qx1 = (2 * a + b + dq) / (5 * (6 * a + b)), qx2 = (2 * a + b - dq) / (5 * (6 * a + b));
    const double vx1 = (18 * a + 4 * b + dv) / (10 * (6 * a + b)), vx2 = (18 * a + 4 * b - dv) / (10 * (6 * a + b));
    const double ax1 = (32 * a + 6 * b + da) / (10 * (6 * a + b)), ax2 = (32 * a + 6 * b - da) / (10 * (6 * a + b));
    // Part 1: q_des
    double kd_lb = p3(s_lb) * (6 * p2(s_lb) - 15 * s_lb + 10), kd_ub = p3(s_ub) * (6 * p2(s_ub) - 15 * s_ub + 10);
    double kd_c = (kd_ub + kd_lb) * 0.5, kd_r = (kd_ub - kd_lb) * 0.5 * kr;
    double ki_lb = q_indep(q0, a, b, s_lb), ki_ub = q_indep(q0, a, b, s_ub);
    bound_ext(ki_lb, ki_ub, s_lb, s_ub, qx1, q_indep(q0, a, b, qx1), qx2, q_indep(q0, a, b, qx2));
    double ki_r = (ki_ub - ki_lb) * 0.5;
    const double q_c = (ki_lb + ki_ub) * 0.5;
    const Itv q_rad = iv(-kd_r - ki_r - cf.ub.qe, kd_r + ki_r + cf.ub.qe);
    const Itv kint = imuls(kd_c, iv(-kr, kr));
    const double sin_qc = sin(q_c), cos_qc = cos(q_c);   // (once each; the shared interval terms once as well: same values, same bits)
    const Itv q_all = iadd(iadds(q_c, kint), q_rad), sq_all = isqr(iadd(q_rad, kint));
    {
        Itv rad = isub(imul(ineg(q_rad), iv(sin_qc, sin_qc)), imul(imuls(0.5, icos(q_all)), sq_all));
        o.cos_c = cos_qc + icen(rad);
        rad = isubs(rad, icen(rad));
        o.cos_k = -kd_c * kr * sin_qc;
        o.cos_e = irad(rad);
    }
    {
        Itv rad = isub(imul(q_rad, iv(cos_qc, cos_qc)), imul(imuls(0.5, isin(q_all)), sq_all));
        o.sin_c = sin_qc + icen(rad);
        rad = isubs(rad, icen(rad));
        o.sin_k = kd_c * kr * cos_qc;
        o.sin_e = irad(rad);
    }
    // Part 2: qd_des
    kd_lb = (30 * p2(s_lb) * p2(s_lb - 1)) / D;
    kd_ub = (30 * p2(s_ub) * p2(s_ub - 1)) / D;
    if (kd_ub < kd_lb) { const double t = kd_lb; kd_lb = kd_ub; kd_ub = t; }
    kd_c = (kd_ub + kd_lb) * 0.5 * kr;
    kd_r = (kd_ub - kd_lb) * 0.5 * kr;
    ki_lb = qd_indep(a, b, s_lb, D); ki_ub = qd_indep(a, b, s_ub, D);
    bound_ext(ki_lb, ki_ub, s_lb, s_ub, vx1, qd_indep(a, b, vx1, D), vx2, qd_indep(a, b, vx2, D));
    ki_r = (ki_ub - ki_lb) * 0.5;
    o.qd_c = (ki_lb + ki_ub) * 0.5;
    o.qd_k = kd_c;
    o.qd_e = kd_r + ki_r + cf.ub.qde;
    o.qda_e = kd_r + ki_r + cf.ub.qdae;
    // Part 3: qdd_des
    const double MAXIMA = 0.5 - sqrt(3.0) / 6, MINIMA = 0.5 + sqrt(3.0) / 6;
#define ACC(s) ((60 * (s) * (2 * p2(s) - 3 * (s) + 1)) / D / D)
    const double t_lb = ACC(s_lb), t_ub = ACC(s_ub);
    if (s_ub <= MAXIMA) { kd_lb = t_lb; kd_ub = t_ub; }
    else if (s_lb <= MAXIMA) { kd_lb = fmin(t_lb, t_ub); kd_ub = ACC(MAXIMA); }
    else if (s_ub <= MINIMA) { kd_lb = t_ub; kd_ub = t_lb; }
    else if (s_lb <= MINIMA) { kd_lb = ACC(MINIMA); kd_ub = fmax(t_lb, t_ub); }
    else { kd_lb = t_lb; kd_ub = t_ub; }
#undef ACC
    kd_c = (kd_ub + kd_lb) * 0.5 * kr;
    kd_r = (kd_ub - kd_lb) * 0.5 * kr;
    ki_lb = qdd_indep(a, b, s_lb, D); ki_ub = qdd_indep(a, b, s_ub, D);
    bound_ext(ki_lb, ki_ub, s_lb, s_ub, ax1, qdd_indep(a, b, ax1, D), ax2, qdd_indep(a, b, ax2, D));
    ki_r = (ki_ub - ki_lb) * 0.5;
    o.qdd_c = (ki_lb + ki_ub) * 0.5;
    o.qdd_k = kd_c;
    o.qdd_e = kd_r + ki_r + cf.ub.qddae;
    return o;
}

__device__ inline void make_rotation(double* R, double c, double s, int axis, bool from_zero) {  // RT/PZsparse.cu:211-250
    for (int i = 0; i < 9; i++) R[i] = 0.0;
    if (!from_zero) R[0] = R[4] = R[8] = 1.0;
    const double ns = -1.0 * s;
    if (axis == 1) { R[4] = c; R[5] = ns; R[7] = s; R[8] = c; }
    else if (axis == 2) { R[0] = c; R[2] = s; R[6] = ns; R[8] = c; }
    else if (axis == 3) { R[0] = c; R[1] = ns; R[3] = s; R[4] = c; }
}
__device__ inline void rpy_matrix(double roll, double pitch, double yaw, double* c) {  // RT/PZsparse.cu:160-176
    // (each of the six values once: the same function of the same argument gives the same bits, and 22 calls were 14 k cycles of every item's start)
    const double cp = cos(pitch), sp = sin(pitch), cy = cos(yaw), sy = sin(yaw), cr = cos(roll), sr = sin(roll);
    c[0] = cp * cy;
    c[1] = -cp * sy;
    c[2] = sp;
    c[3] = cr * sy + cy * sp * sr;
    c[4] = cr * cy - sp * sr * sy;
    c[5] = -cp * sr;
    c[6] = sr * sy - cr * cy * sp;
    c[7] = cy * sr + cr * sp * sy;
    c[8] = cp * cr;
}

// ------------------------------------------------------------------ per-wave state and slot pools
struct Chain {
    typedef PZ PZT;
    static constexpr bool kWalkHelpers = false;   // (the time-vectorised chain has them: p1_tv.inc.h)
    static constexpr bool kFusedCross = false;    // (likewise)
    static constexpr bool kPairs = true;          // run_rnea_free: two waves per operator in the backward pass of a four-wave block
    static constexpr bool kTwoCu = true;          // a time step on two CUs (p1_free.inc.h)
    bool helper = false, two_cu = false;          // this block is the helper of its item | this item runs on two CUs (decided per item: xch_decide)
    long long* phase_log = nullptr;               // (ARMOUR_P1_TRACE) this block's row of P1Cfg::phase_log
    __device__ void phase(int k) const { if (phase_log != nullptr && wid == 0 && w.lane == 0) phase_log[k] = clock64(); }
    JrsLane jl;                                   // (late_jrs) this wave's copy of the per-lane JRS scalars, for the joints built after the roles have begun
    bool late_jrs = false;                        // two CUs, level 3: joints 4 .. J - 1 are built by one wave of the block while the recursions run their first steps (jrs_late_joints)
    int two_level = 0;                            // 1: the three families of velocity-only products | 2: + the main block drops its own w recursion and the fourth wave's cross products | 3: + the angular step's cross product from the helper, no velocity recursion left in the main block
    int hpf[4] = {0, 10, 20, 30}, hpc[4] = {10, 10, 10, 33};   // the helper block's pool parts
    __device__ void helper_parts(int lvl) {
        if (lvl >= 3) { hpf[0] = 0; hpc[0] = J + 1; hpf[1] = J + 1; hpc[1] = 2 * J + 3; hpf[2] = 3 * J + 4; hpc[2] = J + 1; hpf[3] = 4 * J + 5; hpc[3] = L.nV - (4 * J + 5); }
        else for (int r = 0; r < 4; r++) { hpf[r] = kPartHFirst[r]; hpc[r] = kPartHCount[r]; }
    }
    GLB_AS unsigned char* xch = nullptr;          // the item's exchange area
    GLB_AS unsigned char* peer_arena = nullptr;   // main block: the helper's arena
    int xch_epoch = 0;
    Wave w;
    const P1Cfg* cf;
    GLB_AS unsigned char* arena;
    Layout L;
    unsigned long long freeV;  // bit i: 3x1 work slot i is free (a wave only touches the bits of its own roles' parts)
    unsigned freeS;
    int n, J;
    int wid, nw;               // this wave's index in the block and the number of waves (1 or kRoles)
    LDS_AS int* mb;            // LDS mailbox: slot indices handed from one role to another across a block barrier
    // role r runs on wave r of a 3-wave block, every role on the one wave of a 1-wave block
    __device__ bool is(int role) const { return nw == 1 || wid == role; }
    __device__ Wave& wave() { return w; }
    __device__ const Wave& wave() const { return w; }
    // (profile hooks of run_rnea_free; -DP1_STAMPS: these alone, without the per-operator counters of -DP1_PROFILE)
#if defined(P1_PROFILE) || defined(P1_STAMPS)
    long long fwd_done = 0, wait_fwd = 0;
    __device__ long long prof_clock() const { return clock64(); }
    __device__ void prof_waited(long long t0) { bar_wait += clock64() - t0; }
    __device__ void prof_forward_done() { fwd_done = clock64(); wait_fwd = bar_wait; }
    // -DP1_STAMPS: when this wave raised which progress counter (word, value, cycles since the item began)
    long long ph_item = 0; int n_log = 0; int log_word[48]; int log_val[48]; long long log_clk[48];
    __device__ void prof_signal(int word, int value) { if (n_log < 48) { log_word[n_log] = word; log_val[n_log] = value; log_clk[n_log] = clock64() - ph_item; n_log++; } }
#else
    __device__ long long prof_clock() const { return 0; }
    __device__ void prof_waited(long long) {}
    __device__ void prof_forward_done() {}
    __device__ void prof_signal(int, int) {}
#endif
#if defined(P1_PROFILE) || defined(P1_STAMPS)
    mutable long long bar_wait = 0;
    __device__ void bar() const { const long long t0 = clock64(); __syncthreads(); bar_wait += clock64() - t0; }
#else
    __device__ void bar() const { __syncthreads(); }  // all waves of the block: the only cross-wave synchronisation
#endif
    __device__ void post(int slot, const PZ& p) const { if (w.lane == 0) mb[slot] = p.id - L.idV; }
    __device__ PZ take(int slot) const { return V(mb[slot]); }

    LDS_AS double* ci;  // LDS: centre / indep of every slot, classes laid out V | S | M | JM | JV | JS
    __device__ PZ V(int i) const { return mk_slot(arena, L.offV, i, cf->capW, 3, L.idV, ci); }
    __device__ PZ S(int i) const { return mk_slot(arena, L.offS, i, cf->capW, 1, L.idS, ci + L.nV * 9); }
    __device__ PZ M(int i) const { return mk_slot(arena, L.offM, i, cf->capW, 9, L.idM, ci + L.nV * 9 + kNS * 3); }
    __device__ PZ JM(int i) const { return mk_slot(arena, L.offJM, i, kCapSmall, 9, L.idJM, ci + L.nV * 9 + kNS * 3 + kNM * 27); }
    __device__ PZ JV(int i) const { return mk_slot(arena, L.offJV, i, kCapSmall, 3, L.idJV, ci + L.nV * 9 + kNS * 3 + kNM * 27 + L.nJM * 27); }
    __device__ PZ JS(int i) const { return mk_slot(arena, L.offJS, i, kCapSmall, 1, L.idJS, ci + L.nV * 9 + kNS * 3 + kNM * 27 + L.nJM * 27 + L.nJV * 9); }
    // named small slots
    __device__ PZ R(int i) const { return JM(i); }                       // 0..J
    __device__ PZ Rt(int i) const { return JM(J + 1 + i); }              // 0..J-1
    __device__ int scratch(int role) const { return L.nroles == 1 ? 0 : role; }
    __device__ PZ rotRaw(int role) const { return JM(2 * J + 1 + 3 * scratch(role)); }
    __device__ PZ rotS(int role) const { return JM(2 * J + 2 + 3 * scratch(role)); }
    __device__ PZ rpy(int role) const { return JM(2 * J + 3 + 3 * scratch(role)); }
    __device__ PZ inertia(int i) const { return JM(2 * J + 1 + 3 * L.nroles + i); }
    __device__ PZ Ptr(int i) const { return JV(i); }                     // trans, 0..J
    __device__ PZ linkbox(int i) const { return JV(J + 1 + i); }
    __device__ PZ qd(int i) const { return JS(i); }
    __device__ PZ qda(int i) const { return JS(n + i); }
    __device__ PZ qdda(int i) const { return JS(2 * n + i); }
    __device__ PZ mass(int i) const { return JS(3 * n + i); }
    __device__ PZ rawS(int role, int i) const { return JS(3 * n + J + 4 * scratch(role) + i); }  // i = 0..3

    // ---- two waves on one operator (pz_wave.h psync()): this wave becomes one half of the pair led by wave `first`, whose sort buffers both use
    LDS_AS pzkey_t* peer_skey; LDS_AS uint16_t* peer_sidx;   // the sort buffers of the wave this one would join (waves 2 / 3 of four: waves 1 / 0), full size
    LDS_AS pzkey_t* own_skey; LDS_AS uint16_t* own_sidx; int own_cap_raw, own_cap_key;
    __device__ void pair_begin(int first, LDS_AS int* words) {
        own_skey = w.skey; own_sidx = w.sidx; own_cap_raw = w.cap_raw; own_cap_key = w.cap_key;
        w.half = wid == first ? 0 : 1; w.nl = 2 * WAVE; w.lane2 = w.lane + WAVE * w.half; w.pair = words;
        if (w.half) { w.skey = peer_skey; w.sidx = peer_sidx; w.cap_raw = cf->capRaw; w.cap_key = cf->capKey; }
    }
    __device__ void pair_end() { w.skey = own_skey; w.sidx = own_sidx; w.cap_raw = own_cap_raw; w.cap_key = own_cap_key; solo(w); }
    // an operator of this wave alone in the middle of a pair's work (its sort buffers are the pair's: the other half must be kept off them, psync() after)
    struct PairState { int lane2, nl, half; LDS_AS int* pair; };
    __device__ PairState solo_begin() { PairState st{w.lane2, w.nl, w.half, w.pair}; solo(w); return st; }
    __device__ void solo_end(const PairState& st) { w.lane2 = st.lane2; w.nl = st.nl; w.half = st.half; w.pair = st.pair; }
    __device__ unsigned long long part_mask(int r) const {
        if (helper) return ((1ull << hpc[r]) - 1ull) << hpf[r];
        return L.nroles == 1 ? ~0ull : L.nroles == kRoles ? ((1ull << kPartCount[r]) - 1ull) << kPartFirst[r] : ((1ull << kPart4Count[r]) - 1ull) << kPart4First[r];
    }

    int role = 0;  // the role the code being executed belongs to: selects the part of the 3x1 pool allocV() draws from
    __device__ PZ allocV() {
        const unsigned long long part = part_mask(role);
        const int i = __ffsll((long long)(freeV & part)) - 1;
        if (i < 0) { flag(w, ERR_SLOT_OVERFLOW); if (w.lane == 0) w.lstat[3] = 100 + role; return V(helper ? hpf[role] : L.nroles == 1 ? 0 : L.nroles == kRoles ? kPartFirst[role] : kPart4First[role]); }
        freeV &= ~(1ull << i);
        return V(i);
    }
    __device__ void freeVs(const PZ& p) { freeV |= 1ull << (p.id - L.idV); }
    __device__ PZ allocS() {
        const int i = __ffs(freeS) - 1;
        if (i < 0) { flag(w, ERR_SLOT_OVERFLOW); if (w.lane == 0) w.lstat[3] = 200 + role; return S(0); }
        freeS &= ~(1u << i);
        return S(i);
    }
    __device__ void freeSs(const PZ& p) { freeS |= 1u << (p.id - L.idS); }

    // ---- composite operators (names as in RT/PZsparse.cu) ----
    __device__ PZ add(const PZ& a, const PZ& b, double sb = 1.0) {  // operator+ / operator- on 3x1
        PZ o = allocV();
        Seg s[2] = {{view(w, a), 1.0, -1}, {view(w, b), sb, -1}};
        lincomb<3, 2>(w, o, s);
        return o;
    }
    __device__ PZ addOneDim(const PZ& p, const PZ& a, int r) {  // RT/PZsparse.cu:1068-1085 (fresh slot)
        PZ o = allocV();
        Seg s[2] = {{view(w, p), 1.0, -1}, {view(w, a), 1.0, r}};
        lincomb<3, 2>(w, o, s);
        return o;
    }
    __device__ PZ embedOneDim(const PZ& a, int r) {   // addOneDimPZ(PZsparse(0,0,0), a, r): the same two operators as before
        PZ zero = allocV();
        set_const(w, zero, nullptr, nullptr);
        PZ o = addOneDim(zero, a, r);
        freeVs(zero);
        return o;
    }
    __device__ PZ stack(const PZ& r0, const PZ& r1, const PZ& r2) {  // RT/PZsparse.cu:1087-1116
        PZ o = allocV();
        Seg s[3] = {{view(w, r0), 1.0, 0}, {view(w, r1), 1.0, 1}, {view(w, r2), 1.0, 2}};
        lincomb<3, 3>(w, o, s);
        return o;
    }
    // (a + b) + c with simplify() after each sum, in one pass (pz_wave.h lincomb_chain); c may be a 1x1 PZ added to entry comp_c
    __device__ PZ sum3(const PZ& a, const PZ& b, const PZ& c3, int comp_c = -1) {
        PZ o = allocV();
        Seg s[3] = {{view(w, a), 1.0, -1}, {view(w, b), 1.0, -1}, {view(w, c3), 1.0, comp_c}};
        lincomb_chain<3, 3>(w, o, s);
        return o;
    }
    __device__ PZ sum4(const PZ& a, const PZ& b, const PZ& c3, const PZ& d) {  // ((a + b) + c) + d
        PZ o = allocV();
        Seg s[4] = {{view(w, a), 1.0, -1}, {view(w, b), 1.0, -1}, {view(w, c3), 1.0, -1}, {view(w, d), 1.0, -1}};
        lincomb_chain<3, 4>(w, o, s);
        return o;
    }
    __device__ PZ comb3(const View& a, double sa, const View& b, double sb, const View& c3, double sc) {  // 1x1: (sa*a + sb*b) + sc*c
        PZ o = allocS();
        Seg s[3] = {{a, sa, -1}, {b, sb, -1}, {c3, sc, -1}};
        lincomb_chain<1, 3>(w, o, s);
        return o;
    }
    __device__ PZ comb2(const View& a, double sa, const View& b, double sb) {  // 1x1: sa*a + sb*b
        PZ o = allocS();
        Seg s[2] = {{a, sa, -1}, {b, sb, -1}};
        lincomb<1, 2>(w, o, s);
        return o;
    }
    __device__ PZ crossPzMat(const PZ& a, const double* b) {  // RT/PZsparse.cu:1153-1167: a x b
        PZ o = allocV();
        const double sA[3] = {b[2], b[0], b[1]}, sB[3] = {-b[1], -b[2], -b[0]};
        const int cA[3] = {1, 2, 0}, cB[3] = {2, 0, 1};
        cross_const(w, o, view(w, a), sA, cA, sB, cB);
        return o;
    }
    __device__ PZ crossMatPz(const double* a, const PZ& b) {  // RT/PZsparse.cu:1118-1132: a x b
        PZ o = allocV();
        const double sA[3] = {a[1], a[2], a[0]}, sB[3] = {-a[2], -a[0], -a[1]};
        const int cA[3] = {2, 0, 1}, cB[3] = {1, 2, 0};
        cross_const(w, o, view(w, b), sA, cA, sB, cB);
        return o;
    }
    __device__ PZ mulSS(const View& a, const View& b) {
        PZ o = allocS();
        mul<1, 1, 1, 1>(w, o, a, b);
        return o;
    }
    __device__ PZ crossComp(const PZ& a, int a1, int a2, const PZ& b, int b1, int b2) {  // a[a1]*b[b1] - a[a2]*b[b2]
        PZ t1 = mulSS(elem(w, a, a1), elem(w, b, b1));
        PZ t2 = mulSS(elem(w, a, a2), elem(w, b, b2));
        PZ r = comb2(view(w, t1), 1.0, view(w, t2), -1.0);
        freeSs(t1); freeSs(t2);
        return r;
    }
    __device__ PZ crossPzPz(const PZ& a, const PZ& b) {  // RT/PZsparse.cu:1134-1151
#ifdef P1_COMPOSED_CROSS  // the reference's composition out of 1x1 operators (10 passes); kept for A/B runs
        PZ r0 = crossComp(a, 1, 2, b, 2, 1);
        PZ r1 = crossComp(a, 2, 0, b, 0, 2);
        PZ r2 = crossComp(a, 0, 1, b, 1, 0);
        PZ o = stack(r0, r1, r2);
        freeSs(r0); freeSs(r1); freeSs(r2);
        return o;
#else
        PZ o = allocV();
        cross_pzpz(w, o, view(w, a), view(w, b));  // the same three simplify() stages in one pass (pz_wave.h)
        return o;
#endif
    }
    __device__ PZ mulMV(const PZ& A, const PZ& v) {
        PZ o = allocV();
        mul<3, 3, 3, 1>(w, o, view(w, A), view(w, v));
        return o;
    }
    __device__ PZ mulSV(const PZ& s, const PZ& v) {
        PZ o = allocV();
        mul<1, 1, 3, 1>(w, o, view(w, s), view(w, v));
        return o;
    }
};

// ARMTD comparison mode, CMP/Trajectory.cu:29-61: cos / sin of the offline JRS (tabulated for q0 = 0) rotated by the
// joint's initial angle; the radius term is scaled by 4 as the reference does.
__device__ PZW_NOINLINE JrsScalars armtd_jrs_scalars(const P1Cfg& cf, double q0, int b, int i, int t) {
    PZ_KEEP_RETURN_ADDRESS();
    const int T = cf.T;
    const double* tb = cf.jrs + ((size_t)b * cf.n + i) * 6 * T + t;
    const double cc = tb[0], gc = tb[T], rc = tb[2 * T], cs = tb[3 * T], gs = tb[4 * T], rs = tb[5 * T];
    const double c0 = cos(q0), s0 = sin(q0);
    JrsScalars js;
    js.cos_c = c0 * cc - s0 * cs;
    js.cos_k = c0 * gc - s0 * gs;
    js.cos_e = (fabs(c0) * rc + fabs(s0) * rs) * 4.0;
    js.sin_c = c0 * cs + s0 * cc;
    js.sin_k = c0 * gs + s0 * gc;
    js.sin_e = (fabs(c0) * rs + fabs(s0) * rc) * 4.0;
    js.qd_c = js.qd_k = js.qd_e = js.qda_e = js.qdd_c = js.qdd_k = js.qdd_e = 0.0;
    return js;
}

// ---- the small PZs of the JRS in closed form -------------------------------------------------------------------------------------------
// A joint's rotation, its velocity / acceleration polynomials and its link box are PZs of one to four raw terms whose keys and order are known
// in advance.  Built through the operators (a raw slot, lincomb's simplify(), the constant-left product, transpose33) each of them is a chain of
// round trips through the arena: ~135 k cycles per joint, 0.2 M of a time step's 2.4 M.  Here every lane evaluates the same few expressions in
// registers -- exactly the operators' arithmetic, in their order, including the order in which the wave reduction adds the pruned amounts of
// lanes 0..3 -- and lane 0 writes the finished slots.  Same tables bit for bit (the launch digests of tools/ab.py --reps are those
// of the operator form).
__device__ inline double quad_sum(double r0, double r1, double r2, double r3) { return (r0 + r1) + (r2 + r3); }   // wave_sum() of four lanes
// (No register array below is indexed by a run-time value -- the first version kept the surviving terms in arrays indexed by their count, which
//  the compiler put into scratch memory: 57 k cycles for a rotation that is 9 k of arithmetic.  A term's place in the output is a run-time
//  ADDRESS instead.)
// PZsparse(cos / sin polynomials about the joint axis) (RT/PZsparse.cu:211-250: four raw terms {k: cos_k, e_c: cos_e, k: sin_k, e_s: sin_e},
// simplify()) -> R_i = R_rpy * it (:129-134) and its transpose
__device__ inline void jrs_rotation_direct(Chain& c, int i, const JrsScalars& js, const double* rp) {
    const P1Cfg& cf = *c.cf;
    const Wave& w = c.w;
    const int n = c.n, ax = cf.rb.axes[i];
    const pzkey_t kk = pzkey_bit(2 * i), kc = pzkey_bit(5 * n + 2 * i), ks = pzkey_bit(7 * n + 2 * i);   // kk < kc < ks: the sorted order
    double cen[9], m0[9], m1[9], m2[9], t2[9];
    make_rotation(cen, js.cos_c, js.sin_c, ax, false);
    make_rotation(m0, js.cos_k, 0.0, ax, true);
    make_rotation(m1, js.cos_e, 0.0, ax, true);
    make_rotation(t2, 0.0, js.sin_k, ax, true);
    make_rotation(m2, 0.0, js.sin_e, ax, true);
#pragma unroll
    for (int e = 0; e < 9; e++) m0[e] = 1.0 * m0[e] + 1.0 * t2[e];   // the two k terms, summed in generation order
    // simplify(): sorted positions 0 (k, head), 1 (k, member), 2 (e_c), 3 (e_s); a pruned term's |.| goes to the radius from its head's lane
    const bool k0 = !norm_le_lds<9>(m0, w), k1 = !norm_le_lds<9>(m1, w), k2 = !norm_le_lds<9>(m2, w);   // (tracked: Wave::mg, the prune margin)
    double rind[9];
#pragma unroll
    for (int e = 0; e < 9; e++) rind[e] = 0.0 + quad_sum(k0 ? 0.0 : fabs(m0[e]), 0.0, k1 ? 0.0 : fabs(m1[e]), k2 ? 0.0 : fabs(m2[e]));
    // R = R_rpy * rot (mul<3,3,3,3> with a constant left operand: emit_presorted over rot's surviving monomials, one per lane)
    typedef MulShape<3, 3, 3, 3> SH;
    double arp[9], base[9], Rcen[9], a0[9], a1[9], a2[9];
#pragma unroll
    for (int e = 0; e < 9; e++) arp[e] = fabs(rp[e]) + 0.0;
    SH::mul(arp, rind, base);           // (|c_a| + sum |coef_a|) * indep_b; the two terms with indep_a = 0 are exact zeros
    SH::mul(rp, cen, Rcen);
    SH::mul(rp, m0, a0);
    SH::mul(rp, m1, a1);
    SH::mul(rp, m2, a2);
    const bool K0 = k0 && !norm_le_lds<9>(a0, w), K1 = k1 && !norm_le_lds<9>(a1, w), K2 = k2 && !norm_le_lds<9>(a2, w);
    // lanes of the product's reduction = places among rot's survivors
    const bool second_is_1 = k0 && k1, second_is_2 = (k0 != k1) && k2, third_is_2 = k0 && k1 && k2;
    double Rind[9];
#pragma unroll
    for (int e = 0; e < 9; e++) {
        const double q0 = (k0 && !K0) ? fabs(a0[e]) : 0.0, q1 = (k1 && !K1) ? fabs(a1[e]) : 0.0, q2 = (k2 && !K2) ? fabs(a2[e]) : 0.0;
        const double l0 = k0 ? q0 : k1 ? q1 : k2 ? q2 : 0.0;
        const double l1 = second_is_1 ? q1 : second_is_2 ? q2 : 0.0;
        const double l2 = third_is_2 ? q2 : 0.0;
        Rind[e] = (0.0 + (base[e] + 0.0)) + quad_sum(l0, l1, l2, 0.0);
    }
    if (w.lane == 0) {
        const PZ R = c.R(i), Rt = c.Rt(i);
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int cc = 0; cc < 3; cc++) {
                const int e = r * 3 + cc, et = cc * 3 + r;
                R.cen[e] = Rcen[e]; R.ind[e] = Rind[e]; R.ind2[e] = Rind[e];
                Rt.cen[et] = Rcen[e]; Rt.ind[et] = Rind[e]; Rt.ind2[et] = Rind[e];
            }
        int pos = 0;
        if (K0) {
            R.keys[pos] = kk; Rt.keys[pos] = kk;
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int cc = 0; cc < 3; cc++) { R.coef[(size_t)pos * 9 + r * 3 + cc] = a0[r * 3 + cc]; Rt.coef[(size_t)pos * 9 + cc * 3 + r] = a0[r * 3 + cc]; }
            pos++;
        }
        if (K1) {
            R.keys[pos] = kc; Rt.keys[pos] = kc;
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int cc = 0; cc < 3; cc++) { R.coef[(size_t)pos * 9 + r * 3 + cc] = a1[r * 3 + cc]; Rt.coef[(size_t)pos * 9 + cc * 3 + r] = a1[r * 3 + cc]; }
            pos++;
        }
        if (K2) {
            R.keys[pos] = ks; Rt.keys[pos] = ks;
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int cc = 0; cc < 3; cc++) { R.coef[(size_t)pos * 9 + r * 3 + cc] = a2[r * 3 + cc]; Rt.coef[(size_t)pos * 9 + cc * 3 + r] = a2[r * 3 + cc]; }
            pos++;
        }
        c.w.cnt[R.id] = pos; c.w.cnt[Rt.id] = pos;
    }
}
// PZsparse(centre, {k: a, e: b}) of a velocity / acceleration polynomial (:176-243), simplify()
__device__ inline void jrs_scalar_direct(Chain& c, const PZ& out, double cen, pzkey_t k0, double a, pzkey_t k1, double b) {
    const Wave& w = c.w;
    const double va = 1.0 * a, vb = 1.0 * b;
    const bool ka = !norm_le_lds<1>(&va, w), kb = !norm_le_lds<1>(&vb, w);
    const double ind = 0.0 + quad_sum(ka ? 0.0 : fabs(va), kb ? 0.0 : fabs(vb), 0.0, 0.0);
    if (w.lane == 0) {
        out.cen[0] = cen; out.ind[0] = ind; out.ind2[0] = ind;
        int pos = 0;
        if (ka) { out.keys[pos] = k0; out.coef[pos] = va; pos++; }
        if (kb) { out.keys[pos] = k1; out.coef[pos] = vb; pos++; }
        c.w.cnt[out.id] = pos;
    }
}
// link box (RT/Dynamics.cu:49-61): three 1x1 PZs {centre_j, one generator on key field (j + 2) n}, each simplify()d, then stack()ed and
// simplify()d.  A generator that survives its own simplify() (|g| > threshold) survives the stack's (norm of (0, g, 0) > threshold) as well.
__device__ inline void jrs_linkbox_direct(Chain& c, int i) {
    const P1Cfg& cf = *c.cf;
    const Wave& w = c.w;
    const int n = c.n;
    const double g0 = 1.0 * cf.rb.link_zonotope_generators[3 * i], g1 = 1.0 * cf.rb.link_zonotope_generators[3 * i + 1], g2 = 1.0 * cf.rb.link_zonotope_generators[3 * i + 2];
    const bool k0 = !norm_le_lds<1>(&g0, w), k1 = !norm_le_lds<1>(&g1, w), k2 = !norm_le_lds<1>(&g2, w);
    if (w.lane == 0) {
        const PZ out = c.linkbox(i);
        const double i0 = 0.0 + (k0 ? 0.0 : 0.0 + quad_sum(fabs(g0), 0.0, 0.0, 0.0)) * 1.0, i1 = 0.0 + (k1 ? 0.0 : 0.0 + quad_sum(fabs(g1), 0.0, 0.0, 0.0)) * 1.0,
                     i2 = 0.0 + (k2 ? 0.0 : 0.0 + quad_sum(fabs(g2), 0.0, 0.0, 0.0)) * 1.0;
        out.cen[0] = 0.0 + 1.0 * (1.0 * cf.rb.link_zonotope_center[3 * i]); out.cen[1] = 0.0 + 1.0 * (1.0 * cf.rb.link_zonotope_center[3 * i + 1]); out.cen[2] = 0.0 + 1.0 * (1.0 * cf.rb.link_zonotope_center[3 * i + 2]);
        out.ind[0] = i0 + 0.0; out.ind[1] = i1 + 0.0; out.ind[2] = i2 + 0.0;
        out.ind2[0] = i0 + 0.0; out.ind2[1] = i1 + 0.0; out.ind2[2] = i2 + 0.0;
        int pos = 0;
        if (k0) { out.keys[pos] = pzkey_bit(2 * n); out.coef[(size_t)pos * 3] = 1.0 * g0; out.coef[(size_t)pos * 3 + 1] = 0.0; out.coef[(size_t)pos * 3 + 2] = 0.0; pos++; }
        if (k1) { out.keys[pos] = pzkey_bit(3 * n); out.coef[(size_t)pos * 3] = 0.0; out.coef[(size_t)pos * 3 + 1] = 1.0 * g1; out.coef[(size_t)pos * 3 + 2] = 0.0; pos++; }
        if (k2) { out.keys[pos] = pzkey_bit(4 * n); out.coef[(size_t)pos * 3] = 0.0; out.coef[(size_t)pos * 3 + 1] = 0.0; out.coef[(size_t)pos * 3 + 2] = 1.0 * g2; pos++; }
        c.w.cnt[out.id] = pos;
    }
}
// constant PZs without a local array on the way (set_const takes pointers: a caller's array behind them lives in scratch memory)
__device__ inline void jrs_mass_inertia_direct(Chain& c, int i) {
    const P1Cfg& cf = *c.cf;
    const Wave& w = c.w;
    const PZ pm = c.mass(i), pi = c.inertia(i);
    if (w.lane == 0) {
        pm.cen[0] = cf.rb.mass[i]; pm.ind[0] = 0.0; pm.ind2[0] = armour_mass_uncertainty(&cf.rb, i) * fabs(cf.rb.mass[i]);
        c.w.cnt[pm.id] = 0; c.w.cnt[pi.id] = 0;
        const double unc = armour_inertia_uncertainty(&cf.rb, i);
#pragma unroll
        for (int e = 0; e < 9; e++) { const double v = cf.rb.inertia[9 * i + e]; pi.cen[e] = v; pi.ind[e] = 0.0; pi.ind2[e] = unc * fabs(v); }
    }
}

// JRS of one time interval (RT/Trajectory.cu:63-254) + the constant PZs of KinematicsDynamics (RT/Dynamics.cu:6-67).
// The joints are independent of each other: joint i is built by wave i % (waves of the block), with that wave's scratch slots.  (One joint
// is ~135 k cycles of tiny operators, each a round trip through the arena: on three waves of four this phase was 0.40 M of the 2.5 M cycles
// of a lone problem's time step.  Dealing the three parts of a joint -- rotation, velocity polynomials, link box -- to different waves
// balances better but evaluates jrs_scalars twice per joint, and came out slower: profiles/r03_p1_four_waves.txt.)  What the item does not
// read is not built: a chain that stops at the forward kinematics has no velocity polynomials, an RNEA item whose forward kinematics runs
// as an item of its own no link boxes.
// The caller follows with a block barrier.
// the per-lane half of it: lane i holds the scalars of joint i (the other lanes shadow the last joint)
__device__ PZW_NOINLINE JrsLane jrs_lane(Chain& c, int b, int t) {
    PZ_KEEP_RETURN_ADDRESS();
    const P1Cfg& cf = *c.cf;
    const int n = c.n, J = c.J;
    const double* bz = cf.bez + (size_t)b * 3 * n;
    // The scalar work of a joint -- the trigonometry of its fixed rotation (14 k cycles) and jrs_scalars (19 k) -- is the same in all 64 lanes
    // of the per-step wave: lane i does it for joint i instead, all of this wave's joints at once, and the joint loop below reads lane i's
    // results.  Same functions on the same arguments: same bits.
    const int lane = c.w.lane;
    const int il = lane < J ? lane : J - 1;   // (the other lanes shadow the last joint)
    const bool il_actuated = il < n && cf.rb.axes[il] != 0;
    JrsLane o;
    rpy_matrix(cf.rb.rots[3 * il], cf.rb.rots[3 * il + 1], cf.rb.rots[3 * il + 2], o.rp);
    const int ia = il_actuated ? il : 0;   // (a lane without an actuated joint of its own computes joint 0's: never read)
    if (cf.mode == ARMOUR_MODE_ARMTD) o.js = armtd_jrs_scalars(cf, bz[ia], b, ia, t);
    else o.js = jrs_scalars(cf, bz[ia], bz[n + ia], bz[2 * n + ia], ia, t);
    return o;
}
// joint i from lane i's scalars, by the calling wave (with the scratch slots of the role it is in)
__device__ PZW_NOINLINE void jrs_joint(Chain& c, const JrsLane& jl, int i, bool kin_only, bool boxes) {
    PZ_KEEP_RETURN_ADDRESS();
    const P1Cfg& cf = *c.cf;
    const int n = c.n;
    auto bcast = [&](double v, int l) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l)); };
    const bool actuated = i < n && cf.rb.axes[i] != 0;
    JrsScalars js;
    const pzkey_t kk = pzkey_bit(2 * i);
    js.cos_c = bcast(jl.js.cos_c, i); js.cos_k = bcast(jl.js.cos_k, i); js.cos_e = bcast(jl.js.cos_e, i);
    js.sin_c = bcast(jl.js.sin_c, i); js.sin_k = bcast(jl.js.sin_k, i); js.sin_e = bcast(jl.js.sin_e, i);
    js.qd_c = bcast(jl.js.qd_c, i); js.qd_k = bcast(jl.js.qd_k, i); js.qd_e = bcast(jl.js.qd_e, i); js.qda_e = bcast(jl.js.qda_e, i);
    js.qdd_c = bcast(jl.js.qdd_c, i); js.qdd_k = bcast(jl.js.qdd_k, i); js.qdd_e = bcast(jl.js.qdd_e, i);
    {
        double rp[9];
#pragma unroll
        for (int e = 0; e < 9; e++) rp[e] = bcast(jl.rp[e], i);
        if (actuated) {
            jrs_rotation_direct(c, i, js, rp);   // R_i and its transpose (:129-134)
        } else {
            set_const(c.w, c.R(i), rp, nullptr);
            transpose33(c.w, c.Rt(i), c.R(i));
        }
        set_const(c.w, c.Ptr(i), &cf.rb.trans[3 * i], nullptr);
    }
    if (!kin_only) {
        // qd_des, qda_des, qdda_des (:176-243)
        if (actuated) {
            jrs_scalar_direct(c, c.qd(i), js.qd_c, kk, js.qd_k, pzkey_bit(2 * n + i), js.qd_e);
            jrs_scalar_direct(c, c.qda(i), js.qd_c, kk, js.qd_k, pzkey_bit(3 * n + i), js.qda_e);
            jrs_scalar_direct(c, c.qdda(i), js.qdd_c, kk, js.qdd_k, pzkey_bit(4 * n + i), js.qdd_e);
        }
        jrs_mass_inertia_direct(c, i);   // radius 0 for the nominal pass, uncertainty * |centre| for the interval pass (RT/Dynamics.cu:27-40)
    }
    if (boxes) {
        jrs_linkbox_direct(c, i);   // three 1x1 PZs with pseudo-variables at key fields n, 2n, 3n, stacked (RT/Dynamics.cu:49-61)
    }
}
// (i_end < J, `keep`: a time step on two CUs at level 3 -- the first joints here, one per wave, the others by ONE wave that has time to spare
//  once the roles have begun: jrs_late_joints; the recursions start a round of joints earlier)
__device__ PZW_NOINLINE void build_jrs(Chain& c, int b, int t, bool kin_only, bool with_boxes = false, int i_end = ARMOUR_MAX_JOINTS, JrsLane* keep = nullptr) {
    PZ_KEEP_RETURN_ADDRESS();
    const P1Cfg& cf = *c.cf;
    const int J = c.J;
    const bool boxes = kin_only || cf.fk_items == 0 || with_boxes;   // (with_boxes: the helper block of a time step on two CUs -- velocity polynomials for its recursions AND link boxes for the forward kinematics)
    const JrsLane jl = jrs_lane(c, b, t);
    c.prof_signal(10200, 0);   // (-DP1_STAMPS: the scalars are done)
    for (int i = 0; i < J && i < i_end; i++) {
        const int role = c.nw == 1 ? 0 : i % c.nw;   // the wave that builds joint i, with its own scratch slots
        if (!c.is(role)) continue;
        jrs_joint(c, jl, i, kin_only, boxes);
    }
    WSYNC();   // lane 0's slot writes are visible to the wave (the caller's block barrier does the same for the block)
    c.prof_signal(10202, 0);   // (the joints are done)
    if (c.is(0)) {
        double id[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};  // R(NUM_JOINTS) = PZsparse(0,0,0) (:253)
        set_const(c.w, c.R(J), id, nullptr);
        set_const(c.w, c.Ptr(J), &cf.rb.trans[3 * J], nullptr);
    }
    if (keep) *keep = jl;
}
// the joints build_jrs left out, on the calling wave alone
__device__ PZW_NOINLINE void jrs_late_joints(Chain& c, const JrsLane& jl, int i_begin, bool with_boxes) {
    PZ_KEEP_RETURN_ADDRESS();
    const bool boxes = c.cf->fk_items == 0 || with_boxes;
    for (int i = i_begin; i < c.J; i++) jrs_joint(c, jl, i, false, boxes);
    WSYNC();
}

// RT/PZsparse.cu:370-402 reduce_link_PZ + write of the final link table entry
__device__ PZW_NOINLINE void emit_link(Chain& c, const PZ& p, int b, int l, int t) {
    PZ_KEEP_RETURN_ADDRESS();
    const P1Cfg& cf = *c.cf;
    Wave& w = c.w;
    const int n = c.n, cnt = w.cnt[p.id];
    const pzkey_t kmax = pzkey_bit(2 * n), lmax = pzkey_bit(5 * n), kmask = kmax - 1;
    const size_t idx = ((size_t)b * c.J + l) * cf.T + t;
    double* gens = cf.link_gens + (((size_t)b * cf.T + t) * c.J + l) * 18;
    if (w.lane < 18) gens[w.lane] = 0.0;
    WSYNC();
    double ra[3] = {0, 0, 0};
    int nk = 0, ng = 0;
    for (int base = 0; base < cnt; base += WAVE) {
        const int m = base + w.lane;
        bool isk = false, isg = false;
        if (m < cnt) {
            const pzkey_t key = p.keys[m];
            isk = key < kmax;
            isg = !isk && key < lmax && (key & kmask) == 0;
            if (isk) {
                if (m < cf.capL) {
                    cf.link_keys[idx * cf.capL + m] = (uint32_t)key;
                    for (int e = 0; e < 3; e++) cf.link_coeff[(idx * cf.capL + m) * 3 + e] = p.coef[(size_t)m * 3 + e];
                }
            } else if (!isg) {
                for (int e = 0; e < 3; e++) ra[e] += fabs(p.coef[(size_t)m * 3 + e]);
            }
        }
        const unsigned long long mg = __ballot(isg);
        if (isg) {
            const int j = ng + __popcll(mg & ((1ull << w.lane) - 1ull));
            if (j < 3) for (int r = 0; r < 3; r++) gens[r * 6 + j] = p.coef[(size_t)m * 3 + r];
        }
        ng += __popcll(mg);
        nk += __popcll(__ballot(isk));
    }
    if (ng > 3) flag(w, ERR_LINK_GENS);
    if (nk > cf.capL) { flag(w, ERR_TABLE_OVERFLOW); nk = cf.capL; }
    for (int e = 0; e < 3; e++) ra[e] = p.ind[e] + wave_sum(ra[e]);
    if (w.lane == 0) {
        cf.link_count[idx] = nk;
        for (int e = 0; e < 3; e++) {
            cf.link_center[idx * 3 + e] = p.cen[e];
            cf.link_indep[idx * 3 + e] = ra[e];
            gens[e * 6 + 3 + e] = ra[e];
        }
    }
    WSYNC();
}

// RT/Dynamics.cu:69-81 + RT/armour_main.cu:121-124
// Forward kinematics / forward occupancy (RT/Dynamics.cu:69-81), one joint per call so that role 2 can run it alongside
// the RNEA phases of roles 0 and 1 (it shares nothing with them but the read-only JRS).
// (templates on the chain type: the per-step chain of this file and the time-vectorised chain of p1_tv.inc.h run the same
//  operator sequence and the same role choreography on two arithmetics; the operators are found by their argument types)
template <class PZT> struct FkStateT { PZT R, Rn, T; };
typedef FkStateT<PZ> FkState;
template <class CH>
__device__ PZW_NOINLINE void fk_begin(CH& c, FkStateT<typename CH::PZT>& f) {
    PZ_KEEP_RETURN_ADDRESS();
    f.R = c.M(0); f.Rn = c.M(1);
    double id[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    set_const(c.w, f.R, id, nullptr);
    f.T = c.allocV();
    set_const(c.w, f.T, nullptr, nullptr);
}
template <class CH>
__device__ PZW_NOINLINE void fk_step(CH& c, FkStateT<typename CH::PZT>& f, int i, int b, int t) {
    PZ_KEEP_RETURN_ADDRESS();
    typedef typename CH::PZT PZ;
    auto& w = c.w;
    PZ tp = c.mulMV(f.R, c.Ptr(i));
    PZ nt = c.add(f.T, tp);
    c.freeVs(tp); c.freeVs(f.T);
    f.T = nt;
    mul<3, 3, 3, 3>(w, f.Rn, view(w, f.R), view(w, c.R(i)));
    { PZ s = f.R; f.R = f.Rn; f.Rn = s; }
    PZ l1 = c.mulMV(f.R, c.linkbox(i));
    PZ lk = c.add(l1, f.T);
    emit_link(c, lk, b, i, t);
    c.freeVs(l1); c.freeVs(lk);
}

// RT/Dynamics.cu:83-181; u[i] receives freshly allocated scalar slots
// Passivity RNEA with nominal and interval parameters in one pass (RT/Dynamics.cu:83-181), and the FK steps.
//
// Work of one time step is dealt to three ROLES.  In a 3-wave block role r is wave r and the roles run concurrently; in a
// 1-wave block the one wave plays all three in turn.  Every operator is executed by exactly one wave with the same
// operands either way, so the results do not depend on the block shape.
//   forward, step s = 0..J:   role 0  linear_acc_s     role 1  w_s, w_aux_s, wdot_s     role 2  F_{s-1}, N_{s-1}, FK of joint s (unless split off)
//   (F and N of a joint only feed the backward pass, so they trail the state recursion by one step: one barrier per joint)
//   backward, joint i  phase 1:  role 0  R n, com x F_i          role 1  R f, p x (R f)
//                      phase 2:  role 0  n_i, u_i                role 1  f_i
// A role allocates and frees only in its own part of the slot pool; what another role still reads is freed by its owner
// after the next block barrier.  Results cross roles as slot indices in the LDS mailbox.
enum { MB_WV = 0, MB_WDOT, MB_WAUX, MB_LACC, MB_F, MB_N = MB_F + ARMOUR_MAX_JOINTS, MB_A2 = MB_N + ARMOUR_MAX_JOINTS, MB_C2, MB_FF, MB_NN, MB_WORDS };
template <class CH>
__device__ PZW_NOINLINE void run_rnea(CH& c, typename CH::PZT* u, int b, int t) {
    PZ_KEEP_RETURN_ADDRESS();
    typedef typename CH::PZT PZ;
    const P1Cfg& cf = *c.cf;
    auto& w = c.w;
    const int J = c.J;
    FkStateT<PZ> fk;
    const bool with_fk = cf.fk_items == 0;  // otherwise another block runs this item's forward kinematics
    if (c.is(2) && with_fk) { c.role = 2; fk_begin(c, fk); }
    if (c.is(1)) {
        c.role = 1;
        PZ wv = c.allocV(), wdot = c.allocV(), waux = c.allocV();
        set_const(w, wv, nullptr, nullptr);
        set_const(w, wdot, nullptr, nullptr);
        set_const(w, waux, nullptr, nullptr);
        c.post(MB_WV, wv); c.post(MB_WDOT, wdot); c.post(MB_WAUX, waux);
    }
    if (c.is(0)) {
        c.role = 0;
        PZ lacc = c.allocV();
        double g[3] = {0.0, 0.0, cf.rb.gravity};
        set_const(w, lacc, g, nullptr);
        c.post(MB_LACC, lacc);
    }
    c.bar();
    for (int s = 0; s <= J; s++) {
        // state and linear acceleration of joint s-1 (read-only during this step)
        const PZ wv = c.take(MB_WV), wdot = c.take(MB_WDOT), waux = c.take(MB_WAUX), lacc = c.take(MB_LACC);
        c.bar();  // everyone has taken the handles before this step posts the new ones
        if (c.is(0)) {
            c.role = 0;
            if (s < J) {  // line 16: linear_acc = R_t * (linear_acc + cross(wdot, tr) + cross(w, cross(w_aux, tr)))
                const double* tr = &cf.rb.trans[3 * s];
                PZ c1 = c.crossPzMat(wdot, tr);
                PZ c2 = c.crossPzMat(waux, tr);
                PZ c3 = c.crossPzPz(wv, c2); c.freeVs(c2);
                PZ s2 = c.sum3(lacc, c1, c3); c.freeVs(c1); c.freeVs(c3);  // (linear_acc + c1) + c3
                PZ nl = c.mulMV(c.Rt(s), s2); c.freeVs(s2);
                c.post(MB_LACC, nl);
            }
        }
        if (c.is(1) && s < J) {  // lines 13-15: rotate w, w_aux, wdot into the joint frame, add the joint's own motion
            c.role = 1;
            const PZ Rt = c.Rt(s);
            const int ax = abs(cf.rb.axes[s]) - 1;
            PZ nw = c.mulMV(Rt, wv);
            if (cf.rb.axes[s] != 0) { PZ t2 = c.addOneDim(nw, c.qd(s), ax); c.freeVs(nw); nw = t2; }
            PZ na = c.mulMV(Rt, waux);
            PZ nd = c.mulMV(Rt, wdot);
            if (cf.rb.axes[s] != 0) {
                PZ zero = c.allocV();
                set_const(w, zero, nullptr, nullptr);
                PZ temp = c.addOneDim(zero, c.qd(s), ax); c.freeVs(zero);
                PZ c4 = c.crossPzPz(na, temp); c.freeVs(temp);
                PZ nd2 = c.sum3(nd, c4, c.qdda(s), ax); c.freeVs(c4); c.freeVs(nd); nd = nd2;  // (wdot + c4), then + qdda on the axis
                PZ na2 = c.addOneDim(na, c.qda(s), ax); c.freeVs(na); na = na2;
            }
            c.post(MB_WV, nw); c.post(MB_WDOT, nd); c.post(MB_WAUX, na);
        }
        if (c.is(kRoleN) && s >= 1) {  // line 29 for joint s-1: N = I * wdot + cross(w_aux, I * w)
            c.role = kRoleN;
            const PZ I = c.inertia(s - 1);
            PZ t1 = c.mulMV(I, wdot);
            PZ t2 = c.mulMV(I, wv);
            PZ cr = c.crossPzPz(waux, t2); c.freeVs(t2);
            PZ N = c.add(t1, cr); c.freeVs(t1); c.freeVs(cr);
            c.post(MB_N + s - 1, N);
        }
        if (c.is(2)) {
            c.role = 2;
            if (s >= 1) {  // lines 23 & 27 for joint s-1: F = m * (linear_acc + cross(wdot, com) + cross(w, cross(w_aux, com)))
                const double* cm = &cf.rb.com[3 * (s - 1)];
                PZ c1 = c.crossPzMat(wdot, cm);
                PZ c2 = c.crossPzMat(waux, cm);
                PZ c3 = c.crossPzPz(wv, c2); c.freeVs(c2);
                PZ s2 = c.sum3(lacc, c1, c3); c.freeVs(c1); c.freeVs(c3);
                PZ F = c.mulSV(c.mass(s - 1), s2); c.freeVs(s2);
                c.post(MB_F + s - 1, F);
            }
            if (s < J && with_fk) fk_step(c, fk, s, b, t);
        }
        c.bar();
        if (c.is(1)) { c.role = 1; c.freeVs(wv); c.freeVs(wdot); c.freeVs(waux); }  // joint s-1's state: every reader is done
        if (c.is(0)) { c.role = 0; c.freeVs(lacc); }
    }
    if (c.is(2) && with_fk) { c.role = 2; c.freeVs(fk.T); }
    if (c.is(0)) { c.role = 0; PZ nn = c.allocV(); set_const(w, nn, nullptr, nullptr); c.post(MB_NN, nn); }
    if (c.is(1)) { c.role = 1; PZ f = c.allocV(); set_const(w, f, nullptr, nullptr); c.post(MB_FF, f); }
    c.bar();
    for (int i = J - 1; i >= 0; i--) {
        const PZ Rn = c.R(i + 1);
        // n = N + R*n + cross(com, F) + cross(trans_next, R*f);  f = R*f + F
        const PZ nn = c.take(MB_NN), f = c.take(MB_FF), Fi = c.take(MB_F + i), Ni = c.take(MB_N + i);
        PZ a1 = nn, c1 = nn;  // role 0's temporaries of phase 1 (kept in its registers across the barrier)
        if (c.is(0)) {
            c.role = 0;
            a1 = c.mulMV(Rn, nn);
            c1 = c.crossMatPz(&cf.rb.com[3 * i], Fi);
        }
        if (c.is(1)) {
            c.role = 1;
            PZ a2 = c.mulMV(Rn, f);
            PZ c2 = c.crossMatPz(&cf.rb.trans[3 * (i + 1)], a2);
            c.post(MB_A2, a2); c.post(MB_C2, c2);
        }
        c.bar();
        const PZ a2 = c.take(MB_A2), c2 = c.take(MB_C2);
        if (c.is(0)) {
            c.role = 0;
            PZ n2 = c.sum4(Ni, a1, c1, c2); c.freeVs(a1); c.freeVs(c1); c.freeVs(nn);  // ((N + a1) + c1) + c2
            c.post(MB_NN, n2);
            if (cf.rb.axes[i] != 0) {
                const int ax = abs(cf.rb.axes[i]) - 1;
                u[i] = c.comb3(elem(w, n2, ax), 1.0, view(w, c.qdda(i)), cf.rb.armature[i], view(w, c.qd(i)), cf.rb.damping[i]);
            }
        }
        if (c.is(1)) {
            c.role = 1;
            PZ f2 = c.add(a2, Fi); c.freeVs(f);
            c.post(MB_FF, f2);
        }
        c.bar();
        // what the other role was still reading in phase 2
        if (c.is(1)) { c.freeVs(a2); c.freeVs(c2); }
        if (c.is(kRoleN)) c.freeVs(Ni);
        if (c.is(2)) c.freeVs(Fi);
    }
    {
        const PZ nn = c.take(MB_NN), f = c.take(MB_FF);
        if (c.is(0)) c.freeVs(nn);
        if (c.is(1)) c.freeVs(f);
    }
    c.bar();
}

#include "p1_free.inc.h"

// disturbance w = u_int - u_nom, reduce(u_nom), robust-input radius (RT/armour_main.cu:133-141,172-205)
// (part / parts: the waves of a free-running block share the joints' tables -- joint j on wave j % parts -- after the last barrier of the RNEA;
//  the radii every table entry needs are a few LDS reads, which every wave does for itself)
__device__ PZW_NOINLINE void finish_torque(Chain& c, PZ* u_nom, int b, int t, int part = 0, int parts = 1) {
    PZ_KEEP_RETURN_ADDRESS();
    const P1Cfg& cf = *c.cf;
    Wave& w = c.w;
    const int n = c.n, T = cf.T;
    const pzkey_t kmax = pzkey_bit(2 * n);
    Itv rho = {0.0, 0.0};
    double tr[ARMOUR_MAX_FACTORS], un_ind[ARMOUR_MAX_FACTORS];
    for (int j = 0; j < n; j++) {
        // disturbance u_nom_int - u_nom (RT/armour_main.cu:133-136) and its toInterval (RT/PZsparse.cu:557-576): the two
        // PZs share centre, keys and coefficients (see the header note), so the centre is 0, every monomial cancels to an
        // exact 0 that simplify() drops, and the radius is independent(u_nom_int) + independent(u_nom) (RT/PZsparse.cu:829)
        const double dcen = u_nom[j].cen[0] - u_nom[j].cen[0];
        const double rad = u_nom[j].ind2[0] + u_nom[j].ind[0];
        const double lo = dcen - rad, hi = dcen + rad;
        rho = iadd(rho, imul(iv(lo, hi), iv(lo, hi)));
        tr[j] = cf.rb.alpha * (cf.rb.M_max - cf.rb.M_min) * cf.ub.eps + 0.5 * fmax(fabs(lo), fabs(hi));
        un_ind[j] = 0.0;
        if (j % parts != part) continue;
        // reduce(u_nom) (RT/PZsparse.cu:352-368) straight into the final torque table
        const PZ& p = u_nom[j];
        const int cnt = w.cnt[p.id];
        const size_t idx = ((size_t)b * n + j) * T + t;
        double ra = 0.0;
        int nk = 0;
        for (int base = 0; base < cnt; base += WAVE) {
            const int m = base + w.lane;
            bool isk = false;
            if (m < cnt) {
                const pzkey_t key = p.keys[m];
                isk = key < kmax;
                if (isk) {
                    if (m < cf.capT) { cf.tq_keys[idx * cf.capT + m] = (uint32_t)key; cf.tq_coeff[idx * cf.capT + m] = p.coef[m]; }
                } else {
                    ra += fabs(p.coef[m]);
                }
            }
            nk += __popcll(__ballot(isk));
        }
        if (nk > cf.capT) { flag(w, ERR_TABLE_OVERFLOW); nk = cf.capT; }
        const double ind = p.ind[0] + wave_sum(ra);
        if (w.lane == 0) { cf.tq_count[idx] = nk; cf.tq_center[idx] = p.cen[0]; cf.tq_indep[idx] = ind; }
        un_ind[j] = ind;
    }
    WSYNC();
    // sqrt of the interval sum: Boost clamps a negative lower bound to 0; only .upper() is used (:185-188)
    const double rho_hi = up(sqrt(rho.hi));
    for (int j = 0; j < n; j++) {
        if (j % parts != part) continue;
        double v = tr[j];
        v += 0.5 * rho_hi;
        v += un_ind[j];
        v += cf.rb.friction[j];
        if (w.lane == 0) cf.torque_radius[((size_t)b * n + j) * T + t] = v;
    }
    WSYNC();
}

// P1_WAVES_PER_SIMD: 1 (shipped) = no occupancy request, the kernels hold 256 VGPRs + spill AGPRs, one wave per SIMD.
// 2 (development, `make EXTRA=-DP1_WAVES_PER_SIMD=2`) = every kernel of this file that calls the operators asks for two
// waves per SIMD; with all of them agreeing the <= 256-register budget propagates to the non-inlined operator functions
// (AMDGPU attributor, closed world of this object: 248 VGPRs, no AGPRs, a little more scratch) and a fifth one-wave block
// fits a CU: B = 128 builds in 51.7 ms instead of 58.4.  NOT shipped: in that build, batches with >= 3 blocks per CU come out
// wrong about every second launch, and what makes them wrong has not been found ("Two waves per SIMD",
// profiles/r02_p1_two_waves_per_simd.txt).
#ifndef P1_WAVES_PER_SIMD
#define P1_WAVES_PER_SIMD 1
#endif
#if P1_WAVES_PER_SIMD > 1
#define P1_OCC __attribute__((amdgpu_waves_per_eu(P1_WAVES_PER_SIMD, P1_WAVES_PER_SIMD)))
#define P1_PIN_ONE_WAVE_PER_SIMD()
#else
// The shipped build PINS one wave per SIMD instead of relying on the operators happening to need more than 256 registers
// (ADVICE r2): every kernel that calls the pz_wave / pz_tv operators asks for waves_per_eu(1, 1) and claims the whole
// accumulation-register half of the unified file (a255), so its descriptor holds VGPRs + 256 > 256 registers whatever an edit or a
// compiler update does to the vector-register count -- a second wave cannot be placed on the SIMD.  `make` then reads the
// compiler's resource remarks for these kernels and fails unless each reports occupancy 1 (tools/check_p1_occupancy.py).
#define P1_OCC __attribute__((amdgpu_waves_per_eu(1, 1)))
#define P1_PIN_ONE_WAVE_PER_SIMD() __asm__ volatile("" ::: "a255")
#endif

// The time-vectorised kernels (p1_tv.inc.h) take TWO waves per SIMD when built with -DP1_TV_WAVES_PER_SIMD=2: their eight-wave
// blocks put a dedicated helper wave behind each of the four role waves (pz_tv.h "Dedicated helpers").  The attribute sits on
// every instantiation of the kernel so that the operator functions they share are compiled for 256 registers.
#ifndef P1_TV_WAVES_PER_SIMD
#define P1_TV_WAVES_PER_SIMD P1_WAVES_PER_SIMD
#endif
#if P1_TV_WAVES_PER_SIMD > 1
#define P1_TV_OCC __attribute__((amdgpu_waves_per_eu(P1_TV_WAVES_PER_SIMD, P1_TV_WAVES_PER_SIMD)))
#define P1_TV_PIN()
#else
#define P1_TV_OCC P1_OCC
#define P1_TV_PIN() P1_PIN_ONE_WAVE_PER_SIMD()
#endif
constexpr bool kTvDedicatedHelpers = P1_TV_WAVES_PER_SIMD > 1;

#include "p1_tv.inc.h"

#ifndef P1_TV_VARIANT   // (a variant unit -- p1_reach_tv50.hip -- holds the time-vectorised kernel alone)
// One block per (problem, time step) item.  NW = 1: one wave plays every role in turn (throughput: up to 4 items per CU).
// NW = 3: the roles run concurrently on three waves, each with its own sort buffers (latency: small batches).
// LDS: NW x { skey[capKey] u64, sidx[capRaw] u16, lstat[ST_WORDS] } | cnt[kMaxSlots] | mailbox | ci (centres / radii).
// (round 6: + the wave's prune-margin row, Wave::mg, behind the status words)
__host__ __device__ inline size_t p1_wave_mg_off(int cap_key, int cap_raw) { return (((size_t)cap_key * sizeof(pzkey_t) + (size_t)cap_raw * 2 + ST_WORDS * sizeof(int)) + 7) & ~(size_t)7; }
__host__ __device__ inline size_t p1_wave_lds(int cap_key, int cap_raw) { return ((p1_wave_mg_off(cap_key, cap_raw) + WAVE * sizeof(double)) + 15) & ~(size_t)15; }
__host__ __device__ inline size_t p1_shared_lds(size_t ci_doubles) { return (((size_t)(kMaxSlots + kMbWords) * sizeof(int)) + 15 & ~(size_t)15) + ci_doubles * sizeof(double); }

template <int NW>
__global__ __launch_bounds__(64 * NW) P1_OCC void armour_p1_chain_kernel(P1Cfg cf) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Chain c;
    c.cf = &cf;
    c.n = cf.n; c.J = cf.J;
    c.L = make_layout(cf.J, cf.n, cf.capW, NW);
    c.arena = (GLB_AS unsigned char*)cf.arena + (size_t)blockIdx.x * cf.arena_bytes;
    c.nw = NW;
    c.wid = NW == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    LDS_AS unsigned char* lds = (LDS_AS unsigned char*)smem;
    // waves 0..2: sort buffers of capKey / capRaw entries; wave 3 of a four-wave block: the small ones (kFkCapKey / kFkCapRaw)
    const bool fk_bufs = NW == 4 && c.wid == 3;
    const int my_cap_key = fk_bufs ? kFkCapKey : cf.capKey, my_cap_raw = fk_bufs ? kFkCapRaw : cf.capRaw;
    LDS_AS unsigned char* mine = lds + (size_t)c.wid * p1_wave_lds(cf.capKey, cf.capRaw);
    c.w.skey = (LDS_AS pzkey_t*)mine;
    c.w.sidx = (LDS_AS uint16_t*)(mine + (size_t)my_cap_key * sizeof(pzkey_t));
    c.w.lstat = (LDS_AS int*)(mine + (size_t)my_cap_key * sizeof(pzkey_t) + (size_t)my_cap_raw * 2);
    c.w.mg = cf.margin ? (LDS_AS double*)(mine + p1_wave_mg_off(my_cap_key, my_cap_raw)) : nullptr;
    LDS_AS unsigned char* shared = lds + (size_t)(NW == 4 ? 3 : NW) * p1_wave_lds(cf.capKey, cf.capRaw) + (NW == 4 ? p1_wave_lds(kFkCapKey, kFkCapRaw) : 0);
    c.w.cnt = (LDS_AS int*)shared;
    c.mb = c.w.cnt + kMaxSlots;
    c.ci = (LDS_AS double*)(shared + ((((size_t)(kMaxSlots + kMbWords) * sizeof(int)) + 15) & ~(size_t)15));
#ifdef P1_PROFILE
    unsigned long long* prof_lds = c.w.prof;  // (name kept from the LDS version: these are this wave's own counters now)
    for (int i = 0; i < PR_WORDS; i++) prof_lds[i] = 0;
    const long long prof_start = clock64();
#endif
    P1_PIN_ONE_WAVE_PER_SIMD();
    c.w.cap_raw = my_cap_raw;
    c.w.cap_key = my_cap_key;
    c.w.thr = cf.pr.simplify_threshold;
    c.w.thr_sq = sq_threshold(c.w.thr);
    c.w.lane = threadIdx.x & 63;
    solo(c.w);
    {   // (four-wave blocks: wave 2 may join wave 1's operators, wave 3 wave 0's -- run_rnea_free)
        const int peer = c.wid == 2 ? 1 : 0;
        LDS_AS unsigned char* pb = lds + (size_t)peer * p1_wave_lds(cf.capKey, cf.capRaw);
        c.peer_skey = (LDS_AS pzkey_t*)pb; c.peer_sidx = (LDS_AS uint16_t*)(pb + (size_t)cf.capKey * sizeof(pzkey_t));
    }
    if (c.w.lane < ST_WORDS) c.w.lstat[c.w.lane] = 0;
    // Which block builds which item (round 4).  Round 3 dealt them out by index -- block k took items k, k + blocks, ... -- and an item's cost
    // grows with its time step: three problems, 300 items on 256 CUs, took two full rounds, the second for 44 items.  When there are more items than
    // blocks the blocks now draw from a counter, and the order of the draw puts the late (expensive) steps of every problem first -- longest-
    // processing-time-first on identical machines.  Measured (four-wave blocks, B problems of 100 steps): B = 3 2.12 -> 1.94 ms, 4 2.34 -> 2.09,
    // 8 4.15 -> 3.98, 12 5.57 -> 5.18, 16 7.50 -> 7.10 -- less than the spread of the instrumented build's item costs (0.4 .. 1.0) promised: in the
    // release build the cheapest items are no cheaper than 0.9 of the slowest.  An item's result does not depend on who builds it or when.
    LDS_AS int* qword = c.mb + kMbWords - 1;   // (the last word of the walk-helper channels this chain does not use; pairs: the first 2 * PW_WORDS + 4)
    int it0 = blockIdx.x;
    bool helper_block = false;   // a time step on two CUs (p1_free.inc.h): block helper0 + k is the helper of item k, the blocks between the two ranges have nothing to do
    if constexpr (NW == 4) {
        if (cf.two_cu) {
            // (cf.two_cu >= 30, a test hook: block helper0 + k helps item k + 1 -- the two blocks of an item then sit on DIFFERENT XCDs, what HIP is free to do)
            const int shift = cf.two_cu >= 30 ? 1 : 0;
            if ((int)blockIdx.x >= cf.helper0) { helper_block = true; it0 = ((int)blockIdx.x - cf.helper0 + shift) % cf.n_items; }
            else if ((int)blockIdx.x >= cf.n_items) it0 = cf.n_items + cf.fk_items;
        }
    }
    for (;;) {
        if (cf.queue) {
            __syncthreads();   // everybody is through with the previous item, and has read its index
            if (threadIdx.x == 0) *qword = (int)atomicAdd(cf.queue, 1u);
            __syncthreads();
            it0 = __builtin_amdgcn_readfirstlane(*qword);
        }
        if (it0 >= cf.n_items + cf.fk_items) break;
        const bool fk_only = !helper_block && (it0 >= cf.n_items || cf.fk_only != 0);
        int it = it0 >= cf.n_items ? it0 - cf.n_items : it0;
        if (cf.queue && !cf.items && cf.queue_order != 2) {   // position in the draw -> item: time steps from the last (order 1) or the first (3), every problem's in turn
            const int nb = cf.n_items / cf.T;   // (all problems, all steps: n_items = B * T)
            const int tq = it / nb, bq = it - tq * nb;
            it = bq * cf.T + (cf.queue_order == 1 ? cf.T - 1 - tq : tq);
        }
        const int item = cf.items ? cf.items[it] : it;
        const int b = item / cf.T, t = item - b * cf.T;
        const int err_before = c.w.lstat[ST_ERR];  // (the retry list is only used with NW = 1)
        c.freeV = (1ull << c.L.nV) - 1ull;
        c.freeS = (1u << kNS) - 1u;
        c.role = 0;
        for (int i = threadIdx.x; i < kMaxSlots; i += 64 * NW) c.w.cnt[i] = 0;
        margin_reset(c.w);
        c.helper = helper_block; c.two_cu = false;
        c.phase_log = cf.phase_log ? cf.phase_log + (size_t)blockIdx.x * 8 : nullptr;
        c.phase(0);
        c.two_level = (cf.two_cu % 10) >= 3 && c.J > kTwoCuC4MaxJ ? 2 : cf.two_cu % 10;   // (cf.two_cu >= 10: the tests of the fall-backs -- 10 + level: the helper says yes and hands over nothing; 20 + level: it never says it has started)
        if (helper_block) c.helper_parts(c.two_level);
        if constexpr (NW == 4) {
            if (cf.two_cu) {
                c.xch = (GLB_AS unsigned char*)cf.xch + (size_t)it0 * kXchBytes;
                c.xch_epoch = cf.xch_epoch;
                const int shift = cf.two_cu >= 30 ? 1 : 0;
                c.peer_arena = (GLB_AS unsigned char*)cf.arena + (size_t)(helper_block ? it0 : cf.helper0 + (it0 - shift + cf.n_items) % cf.n_items) * cf.arena_bytes;   // (the other block of the item)
                if (!(helper_block && cf.two_cu >= 20 && cf.two_cu < 30)) xch_hello(c, helper_block);   // (cf.two_cu >= 20: the test of a helper that starts late -- it never says so)
                if (threadIdx.x == 0) c.mb[kMbWords - 3] = 0;   // (xch_take: no take of this item has been lost)
            }
        }
        __syncthreads();
#if defined(P1_PROFILE) || defined(P1_STAMPS)
        const long long ph0 = clock64();
        c.ph_item = ph0; c.n_log = 0;
#endif
        bool late = false;   // a round of joints later: see Chain::late_jrs
        if constexpr (NW == 4) late = cf.two_cu != 0 && c.two_level >= 3 && cf.late_jrs != 0 && c.J > NW && !fk_only;
        build_jrs(c, b, t, fk_only, helper_block, late ? NW : ARMOUR_MAX_JOINTS, late ? &c.jl : nullptr);
        __syncthreads();
        c.phase(1);
#if defined(P1_STAMPS)
        const long long ph_jrs = clock64();
#endif
        if constexpr (NW == 4) {
            if (cf.two_cu) c.two_cu = xch_decide(c, helper_block);   // same XCD as the other block of the item, and both alive: two CUs; otherwise this block does everything (the helper: the forward kinematics alone)
        }
#if defined(P1_STAMPS)
        if (c.w.lane == 0 && t == cf.T - 1 && !fk_only) printf("[t=%d %swave %d] JRS done at %lld, two-CU decision at %lld\n", t, helper_block ? "helper " : "", c.wid, ph_jrs - ph0, (long long)clock64() - ph0);
#endif
        c.phase(2);
        if (c.phase_log != nullptr && threadIdx.x == 0) { c.phase_log[6] = c.two_cu ? 1 : 2; c.phase_log[7] = (long long)(xch_xcc() - 1u); }
        c.late_jrs = late && c.two_cu;
        if (late && !c.two_cu) {   // (no helper after all: the other joints the way build_jrs deals them)
            for (int i = NW; i < c.J; i++) if (c.wid == i % NW) jrs_joint(c, c.jl, i, false, cf.fk_items == 0 || helper_block);
            WSYNC();
            __syncthreads();
        }
#ifdef P1_PROFILE
        const long long ph1 = clock64();
#endif
        PZ u_nom[ARMOUR_MAX_FACTORS];
        if (helper_block) {
            if constexpr (NW == 4) run_helper_free(c, b, t, c.two_cu && (cf.two_cu < 10 || cf.two_cu >= 30));   // (20 + level: the main block has decided against two CUs by itself)
        } else if (fk_only) {
            // forward kinematics and the link tables, nothing else: the whole ARMTD chain (CMP/armtd_main.cu:141-156), or
            // the forward-kinematics half of a split ARMOUR item.  A single role: one wave works.
            if (c.is(2)) {
                FkState fk;
                c.role = 2;
                fk_begin(c, fk);
                for (int i = 0; i < c.J; i++) fk_step(c, fk, i, b, t);
                c.freeVs(fk.T);
            }
        } else if (NW >= kRoles && cf.free_running) {
            run_rnea_free(c, u_nom, b, t);
        } else {
            run_rnea(c, u_nom, b, t);
        }
#ifdef P1_PROFILE
        const long long ph3 = clock64();
#endif
        if (!fk_only && !helper_block) {
            if (NW >= kRoles && cf.free_running) {   // every wave takes its share of the joints' tables (the 1x1 slots of u_nom: the mailbox, past the last barrier of the RNEA)
                for (int j = 0; j < c.n; j++) u_nom[j] = c.S(t3_ld(&c.mb[T3_U + j]));
                finish_torque(c, u_nom, b, t, c.wid, NW);
            } else if (c.is(0)) finish_torque(c, u_nom, b, t);
        }
        c.phase(5);
        if (((c.w.lstat[ST_ERR] & ~err_before)) == 0) margin_item_end(c.w, cf.margin ? cf.margin + (size_t)b : nullptr);   // (not an item this wave flagged: it is built again -- here with larger buffers, or the whole launch, which then starts from cleared words)
        __syncthreads();
        if (cf.retry_list && ((c.w.lstat[ST_ERR] & ~err_before) & ERR_RAW_OVERFLOW)) {
            // this item needs larger sort buffers: hand it to the second launch and forget what it flagged
            if (threadIdx.x == 0) { cf.retry_list[atomicAdd(cf.retry_count, 1u)] = item; c.w.lstat[ST_ERR] = err_before; }
            __syncthreads();
        }
#ifdef P1_PROFILE
        if (threadIdx.x == 0 && (t % 10 == 0 || t == cf.T - 1))
            printf("[t=%d] total %lld: N<=64 %llu (%llu calls), mid %llu (%llu calls, %llu terms), big %llu (%llu calls, %llu terms) | fill %llu sort %llu (rank %llu bitonic %llu linmerge %llu mulmerge %llu) emit %llu abs %llu\n", t, (long long)clock64() - ph0,
                   prof_lds[PR_CYC64], prof_lds[PR_SMALL], prof_lds[PR_CYC512], prof_lds[PR_N512], prof_lds[PR_TERMS512], prof_lds[PR_CYCBIG], prof_lds[PR_NBIG], prof_lds[PR_TERMSBIG],
                   prof_lds[PR_FILL], prof_lds[PR_SORT], prof_lds[PR_S_RANK], prof_lds[PR_S_BITONIC], prof_lds[PR_S_LINMERGE], prof_lds[PR_S_MULMERGE], prof_lds[PR_EMIT], prof_lds[PR_ABS]);
        if (c.w.lane == 0 && (t == 60 || t == 0)) printf("[t=%d wave %d] waited %lld cycles at role barriers / mailbox waits of %lld; forward pass done at %lld\n", t, c.wid, c.bar_wait, (long long)clock64() - ph0, c.fwd_done - ph0);
        c.bar_wait = 0;
        if (threadIdx.x == 0 && blockIdx.x == 0)
            printf("[P1 phases, wave 0] jrs %lld fk+rnea %lld torque %lld cycles\n", ph1 - ph0, ph3 - ph1, (long long)clock64() - ph3);
#endif
#if defined(P1_STAMPS) && !defined(P1_PROFILE)
        if (c.w.lane == 0 && (t == 60 || t == cf.T - 1 || t == 0) && !fk_only) {
            printf("[t=%d %swave %d%s] %lld cycles, %lld of them at barriers / mailbox waits (%lld in the forward pass); forward pass done at %lld\n", t, helper_block ? "helper " : "", c.wid, c.two_cu ? ", two CUs" : "", (long long)clock64() - ph0, c.bar_wait, c.wait_fwd, c.fwd_done - ph0);
            if (t == cf.T - 1) for (int q = 0; q < c.n_log; q++) printf("[t=%d %swave %d signal] word %d value %d at %lld\n", t, helper_block ? "helper " : "", c.wid, c.log_word[q] - T3_CNT, c.log_val[q], c.log_clk[q]);
        }
        if (c.w.lane == 0 && t == cf.T - 1 && fk_only) printf("[t=%d fk item wave %d] %lld cycles\n", t, c.wid, (long long)clock64() - ph0);
        c.bar_wait = 0;
#endif
        if (!cf.queue) it0 += gridDim.x;
    }
#ifdef P1_PROFILE
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        prof_lds[PR_TOTAL] = (unsigned long long)(clock64() - prof_start);
        unsigned long long* dst = (unsigned long long*)(cf.status + ST_WORDS);
        for (int i = 0; i < PR_WORDS; i++) dst[i] = prof_lds[i];
    }
#endif
    if (c.w.lane == 0) {  // every wave reports its own flags and maxima
        if (c.w.lstat[ST_ERR]) atomicOr(&cf.status[ST_ERR], (unsigned)c.w.lstat[ST_ERR]);
        atomicMax(&cf.status[ST_MAX_RAW], (unsigned)c.w.lstat[ST_MAX_RAW]);
        atomicMax(&cf.status[ST_MAX_OUT], (unsigned)c.w.lstat[ST_MAX_OUT]);
        atomicMax(&cf.status[3], (unsigned)c.w.lstat[3]);
    }
}

// pair order of RT/CollisionChecking.cu:26-39: plane p pairs generators (a < b), enumerated row by row over the 9 generators
__host__ __device__ constexpr int plane_pair_a(int p) { int a = 0, b = 1; for (int s = 0; s < p; s++) { if (++b == 9) { a++; b = a + 1; } } return a; }
__host__ __device__ constexpr int plane_pair_b(int p) { int a = 0, b = 1; for (int s = 0; s < p; s++) { if (++b == 9) { a++; b = a + 1; } } return b; }

// Planes 9 GRP .. 9 GRP + 8 of one row.  The generator indices are compile-time constants: indexed with run-time values the
// 9 x 3 generator array lived in scratch memory, and its ~300 scratch loads per thread were what the kernel spent its time on
// (2.0 ms for the 2.6 GB table of 128 worlds at O = 20, where a plain fill of the same addresses takes 0.45 ms).
// A normal up to its sign, as three bit patterns: +-0 -> +0, then the whole vector negated if its first non-zero component is
// negative.  Two normals satisfy (E == C) || (E == -C) component by component exactly when these triples are equal bit for bit.
__device__ inline void canonical_normal(double C0, double C1, double C2, unsigned long long (&b)[3]) {
    const double z0 = C0 == 0.0 ? 0.0 : C0, z1 = C1 == 0.0 ? 0.0 : C1, z2 = C2 == 0.0 ? 0.0 : C2;
    const bool neg = z0 != 0.0 ? z0 < 0.0 : z1 != 0.0 ? z1 < 0.0 : z2 < 0.0;
    b[0] = (unsigned long long)__double_as_longlong(z0 == 0.0 ? 0.0 : neg ? -z0 : z0);
    b[1] = (unsigned long long)__double_as_longlong(z1 == 0.0 ? 0.0 : neg ? -z1 : z1);
    b[2] = (unsigned long long)__double_as_longlong(z2 == 0.0 ? 0.0 : neg ? -z2 : z2);
}
// Signature of a canonical normal.  The zero normal and the three unit axes -- every redundant plane of a world of axis-aligned boxes is
// one of those -- get the EXACT codes 0..3 (equal codes <=> equal normals, nothing to verify); any other normal a 63-bit mix of the
// triple with the top bit set (a filter: equal triples give equal signatures, a match is verified on the normals themselves).
constexpr unsigned long long kOneBits = 0x3FF0000000000000ull;
__device__ inline unsigned long long normal_signature(const unsigned long long (&b)[3]) {
    if ((b[0] | b[1] | b[2]) == 0ull) return 0ull;
    if (b[0] == kOneBits && (b[1] | b[2]) == 0ull) return 1ull;
    if (b[1] == kOneBits && (b[0] | b[2]) == 0ull) return 2ull;
    if (b[2] == kOneBits && (b[0] | b[1]) == 0ull) return 3ull;
    unsigned long long h = b[0] * 0x9E3779B97F4A7C15ull;
    h = (h ^ (h >> 29)) + b[1] * 0xBF58476D1CE4E5B9ull;
    h = (h ^ (h >> 31)) + b[2] * 0x94D049BB133111EBull;
    h ^= h >> 30; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32;
    return h | (1ull << 63);
}
__device__ inline unsigned long long skipped_signature(int p) { return 0x7FFFFFFFFFFFFF00ull + (unsigned long long)p; }   // (top bit clear, > 3: equals no normal's signature)

// a 64-bit value every lane holds alike, moved to scalar registers (readfirstlane returns a SIGNED int: widen through unsigned)
__device__ inline unsigned long long uniform_u64(unsigned long long v) {
    return (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xffffffffull)) |
           ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32);
}

// the normal of one generator pair exactly as RT/CollisionChecking.cu:176-190 computes it (cross product, norm, three divisions)
__device__ inline void pair_normal(const double* ga, const double* gb, double& C0, double& C1, double& C2) {
    const double cr0 = ga[1] * gb[2] - ga[2] * gb[1], cr1 = ga[2] * gb[0] - ga[0] * gb[2], cr2 = ga[0] * gb[1] - ga[1] * gb[0];
    const double nrm = sqrt(cr0 * cr0 + cr1 * cr1 + cr2 * cr2);
    C0 = 0; C1 = 0; C2 = 0;
    if (nrm > 0) { C0 = cr0 / nrm; C1 = cr1 / nrm; C2 = cr2 / nrm; }
}

// What a table built by P1 holds (`lean`): only what the fused evaluation reads (p2_tiles.h: collision_block with LL) --
//   planes 0..20 (an obstacle generator in the pair): the normal and delta;   planes 21..35 (link x link): delta only, their
//   normals once per (link, time step) in the compact planes_ll;   d = A.c only when `store_d` (small batches read it, batches
//   of >= 8 problems recompute it from the obstacle centres);   nothing at all for the planes of `pre`, which are redundant in every
//   row of the problem by the CLASS of their generators (armour_p1_plane_class_kernel) and therefore in plane_skip.
// With axis-aligned box obstacles that is 480 B per row instead of 1440 B.  The full table (`LEAN` false: every plane, every
// component, the reference's layout content) is built on demand for armour_get_hyperplanes.
template <int GRP, bool LEAN>
__device__ inline void planes_of_group(const double (&G)[9][3], const double (&c)[3], bool in, int Q, int q, int o, int lt, int JT,
                                       double* __restrict__ out, double* __restrict__ ll, unsigned long long (&Sg)[ARMOUR_NPLANES][64], unsigned long long (&sig)[9], int lane,
                                       unsigned long long pre, bool store_d) {
#pragma unroll
    for (int k = 0; k < 9; k++) {
        constexpr int p0 = GRP * 9;
        const int p = p0 + k;
        if (LEAN && ((pre >> p) & 1ull)) {   // (wave-uniform) redundant everywhere by class: neither computed nor stored
            sig[k] = skipped_signature(p);
            Sg[p][lane] = sig[k];
            continue;
        }
        const int a_id = plane_pair_a(p), b_id = plane_pair_b(p);  // (constants once the loop is unrolled)
        double C0, C1, C2;
        pair_normal(G[a_id], G[b_id], C0, C1, C2);
        double dl = 0.0;
#pragma unroll
        for (int j = 0; j < 9; j++) dl += fabs(C0 * G[j][0] + C1 * G[j][1] + C2 * G[j][2]);
        if (in) {
            // (layout: common.h armour_plane_index -- {Ax,Ay} and {Az,delta} are 16-B pairs, written as such)
            if (!LEAN || p < ARMOUR_FIRST_LL_PLANE) *reinterpret_cast<double2*>(out + armour_plane_index(Q, q, p, 0)) = make_double2(C0, C1);
            if (p < ARMOUR_FIRST_LL_PLANE) *reinterpret_cast<double2*>(out + armour_plane_index(Q, q, p, 2)) = make_double2(C2, dl);
            else {
                if (!LEAN) out[armour_plane_index(Q, q, p, 2)] = C2;
                out[armour_plane_index(Q, q, p, 4)] = dl;
            }
            if (!LEAN || store_d) out[armour_plane_index(Q, q, p, 3)] = C0 * c[0] + C1 * c[1] + C2 * c[2];
            if (ll && o == 0 && p >= ARMOUR_FIRST_LL_PLANE) {  // link x link normal: the same for every obstacle of this (l, t), kept once
                ll[armour_plane_ll_index(JT, lt, p - ARMOUR_FIRST_LL_PLANE, 0)] = C0;
                ll[armour_plane_ll_index(JT, lt, p - ARMOUR_FIRST_LL_PLANE, 1)] = C1;
                ll[armour_plane_ll_index(JT, lt, p - ARMOUR_FIRST_LL_PLANE, 2)] = C2;
            }
        }
        unsigned long long cb[3];
        canonical_normal(C0, C1, C2, cb);
        sig[k] = normal_signature(cb);
        Sg[p][lane] = sig[k];
    }
}

// canonical normal of plane p of this thread's row, recomputed from the generators in memory (run-time pair indices: only the
// verification path below uses it -- indexing the register copy G with them would move it to scratch memory)
__device__ inline void canonical_normal_of_plane(const double* ob, const double* lg, int p, unsigned long long (&cb)[3]) {
    int a = 0, b = 1;
    for (int s = 0; s < p; s++) { if (++b == 9) { a++; b = a + 1; } }
    double ga[3], gb[3];
    for (int ax = 0; ax < 3; ax++) {
        ga[ax] = a < 3 ? ob[(a + 1) * 3 + ax] : lg[ax * 6 + (a - 3)];
        gb[ax] = b < 3 ? ob[(b + 1) * 3 + ax] : lg[ax * 6 + (b - 3)];
    }
    double C0, C1, C2;
    pair_normal(ga, gb, C0, C1, C2);
    canonical_normal(C0, C1, C2, cb);
}

// Which of this thread's nine planes (9 GRP .. 9 GRP + 8 of row q) repeat an earlier plane of the row or have no normal: bit k of the
// result.  Every earlier signature is read once and compared with the nine own ones (no branches).  Equal EXACT codes (zero normal,
// unit axes) decide at once; equal hashed signatures -- two general normals, which no workload of the suite produces -- are verified on
// the normals themselves, recomputed from the generators.  Nothing is read back from the table.
template <int GRP>
__device__ inline unsigned planes_redundant(const unsigned long long (&Sg)[ARMOUR_NPLANES][64], const unsigned long long (&sig)[9],
                                            const double* ob, const double* lg, int lane, unsigned long long pre) {
    constexpr int p0 = GRP * 9;
    unsigned long long cand[9];
#pragma unroll
    for (int k = 0; k < 9; k++) cand[k] = 0ull;
#pragma unroll
    for (int e2 = 0; e2 < p0 + 8; e2++) {
        const unsigned long long s = Sg[e2][lane];
#pragma unroll
        for (int k = 0; k < 9; k++)
            if (e2 < p0 + k) cand[k] |= s == sig[k] ? 1ull << e2 : 0ull;
    }
    unsigned red_mask = (unsigned)((pre >> p0) & 0x1ffull);   // redundant by class in every row of the problem (not computed at all)
#pragma unroll
    for (int k = 0; k < 9; k++) {
        if (sig[k] == 0ull) { red_mask |= 1u << k; continue; }                      // no normal
        if (cand[k] == 0ull) continue;
        if (sig[k] <= 3ull) { red_mask |= 1u << k; continue; }                      // a unit axis an earlier plane has as well: exact
        if (!(sig[k] >> 63)) continue;                                              // (a skipped plane's own code)
        unsigned long long cb[3];
        canonical_normal_of_plane(ob, lg, p0 + k, cb);
        bool red = false;
        unsigned long long mm = cand[k];
        while (mm != 0ull && !red) {
            const int e2 = __ffsll((long long)mm) - 1;
            mm &= mm - 1ull;
            unsigned long long eb[3];   // same signature: compare the normals themselves
            canonical_normal_of_plane(ob, lg, e2, eb);
            red = eb[0] == cb[0] && eb[1] == cb[1] && eb[2] == cb[2];
        }
        if (red) red_mask |= 1u << k;
    }
    return red_mask;
}

// RT/CollisionChecking.cu:136-228: one thread per (b, q = (l*T+t)*O + o) builds the row's 36 half-spaces.
// It also derives which planes can never win the row's arg-max: a plane whose normal is zero (the -1e8 sentinel
// of :252-259) or bit-for-bit equal to +-the normal of an earlier plane of the same row reproduces values the
// strict-> scan of :268-279 has already seen (d and delta follow the normal exactly), so skipping it changes
// neither the maximum nor the winning normal.  The AND over all rows of a problem goes to plane_skip[b]; with
// axis-aligned box obstacles that is 12 of the 36 planes (all obstacle x error-generator and error x error pairs).
// Four threads per row (wave w of the 256-thread block builds planes 9w .. 9w+8 of the block's 64 rows); the signatures of the
// normals go through LDS so that each thread can test its planes against all earlier planes of its row.
template <bool LEAN>
__global__ __launch_bounds__(256) void armour_p1_planes_kernel(int B, int T, int J, int O, const double* __restrict__ link_gens,
                                                               const double* __restrict__ obstacles, double* __restrict__ planes,
                                                               double* __restrict__ planes_ll, double* __restrict__ obs_center,
                                                               unsigned long long* __restrict__ skip_part, const unsigned long long* __restrict__ pre_mask,
                                                               const unsigned long long* __restrict__ live_mask, int store_d) {
    __shared__ unsigned long long Sg[ARMOUR_NPLANES][64];  // [plane][row of the block]: signatures of the normals (18 KB; the normals themselves, 55 KB, held the kernel at two blocks per CU)
    const int Q = J * T * O;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + lane, b = blockIdx.y;
    const bool in = q < Q;
    const int qc = in ? q : Q - 1;
    const int o = qc % O, lt = qc / O, l = lt / T, t = lt - l * T;
    const double* ob = obstacles + ((size_t)b * O + o) * 12;
    const double* lg = link_gens + (((size_t)b * T + t) * J + l) * 18;
    // planes that are redundant in every row of this problem by the class of their generators (wave-uniform)
    const unsigned long long pre = LEAN ? uniform_u64(pre_mask[b]) : 0ull;
    double G[9][3], c[3];
#pragma unroll
    for (int ax = 0; ax < 3; ax++) {
        c[ax] = ob[ax];
#pragma unroll
        for (int g = 0; g < 3; g++) G[g][ax] = ob[(g + 1) * 3 + ax];
#pragma unroll
        for (int g = 0; g < 6; g++) G[3 + g][ax] = lg[ax * 6 + g];
    }
    double* out = planes + (size_t)b * armour_planes_per_problem(Q);
    if (obs_center && in && grp == 0 && lt == 0) { obs_center[((size_t)b * 3 + 0) * O + o] = c[0]; obs_center[((size_t)b * 3 + 1) * O + o] = c[1]; obs_center[((size_t)b * 3 + 2) * O + o] = c[2]; }
    // this wave's nine planes, with compile-time generator indices (see planes_of_group)
    double* ll = planes_ll ? planes_ll + (size_t)b * armour_planes_ll_per_problem(J * T) : nullptr;
    unsigned long long sig[9];
    switch (grp) {
        case 0: planes_of_group<0, LEAN>(G, c, in, Q, q, o, lt, J * T, out, ll, Sg, sig, lane, pre, store_d != 0); break;
        case 1: planes_of_group<1, LEAN>(G, c, in, Q, q, o, lt, J * T, out, ll, Sg, sig, lane, pre, store_d != 0); break;
        case 2: planes_of_group<2, LEAN>(G, c, in, Q, q, o, lt, J * T, out, ll, Sg, sig, lane, pre, store_d != 0); break;
        default: planes_of_group<3, LEAN>(G, c, in, Q, q, o, lt, J * T, out, ll, Sg, sig, lane, pre, store_d != 0); break;
    }
    if (!skip_part) return;   // (the full table for armour_get_hyperplanes: the masks exist already)
    // plane_skip[b] is an AND over the problem's rows, so a plane that ONE row needs is settled: armour_p1_plane_sample_kernel ran the
    // exact test on a few rows of the problem beforehand.  If every plane is either redundant everywhere by class (`pre`) or known
    // to be needed (`live`) -- always so for box obstacles -- the result is `pre` and this block skips the signature exchange and the
    // ~800 compares per row of the test; otherwise (wave-uniform) it runs the full test as before.  Identical masks either way.
    if (LEAN) {
        const unsigned long long lv = uniform_u64(live_mask[b]);
        if (((pre | lv) & ((1ull << ARMOUR_NPLANES) - 1ull)) == ((1ull << ARMOUR_NPLANES) - 1ull)) {
            if (lane == 0) skip_part[((size_t)b * gridDim.x + blockIdx.x) * 4 + grp] = ~lv;
            return;
        }
    }
    __syncthreads();
    // redundancy of this thread's planes: zero normal, or bit-for-bit +- the normal of an earlier plane of the row
    unsigned long long skip = ~0ull;
    if (in) {
        unsigned red = 0u;
        switch (grp) {
            case 0: red = planes_redundant<0>(Sg, sig, ob, lg, lane, pre); break;
            case 1: red = planes_redundant<1>(Sg, sig, ob, lg, lane, pre); break;
            case 2: red = planes_redundant<2>(Sg, sig, ob, lg, lane, pre); break;
            default: red = planes_redundant<3>(Sg, sig, ob, lg, lane, pre); break;
        }
        // planes of the other three groups: not this thread's to clear
        skip = ~(((unsigned long long)(~red & 0x1ffu)) << (grp * 9));
    }
    // AND over the wave, then one atomic per wave (each wave owns 9 of the 36 bits; the other bits are all ones)
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)skip, o2, 64), hi = __shfl_xor((unsigned)(skip >> 32), o2, 64);
        skip &= ((unsigned long long)hi << 32) | lo;
    }
    // (one atomicAnd per wave on the problem's mask cost a seventh of the kernel: 112 k of them on 128 words at B = 128)
    if (lane == 0) skip_part[((size_t)b * gridDim.x + blockIdx.x) * 4 + grp] = skip;
}

// The exact redundancy test on sampled rows of problem b -- one block per problem, wave s takes sample s, lane p = plane p -- gives the
// planes NEEDED by at least one of them (a non-zero normal that no earlier plane of that row repeats up to the sign): same normals bit
// for bit as the planes kernel computes (pair_normal), compared as canonical triples, no hashing.  The same block ANDs the class
// pre-pass's per-block words of the problem into pre_mask[b] (one launch less in front of the planes kernel).
constexpr int kPlaneSamples = 16;
__global__ __launch_bounds__(64 * kPlaneSamples) void armour_p1_plane_sample_kernel(int T, int J, int O, const double* __restrict__ link_gens,
                                                                                    const double* __restrict__ obstacles, const unsigned long long* __restrict__ pre_part, int n_part,
                                                                                    unsigned long long* __restrict__ pre_mask, unsigned long long* __restrict__ live_mask) {
    __shared__ unsigned long long tri[kPlaneSamples][ARMOUR_NPLANES][3];
    __shared__ unsigned long long s_live[kPlaneSamples], s_pre[kPlaneSamples];
    const int Q = J * T * O, b = blockIdx.x, p = threadIdx.x & 63, sidx = threadIdx.x >> 6;
    {
        const int q = (int)(((long long)sidx * Q) / kPlaneSamples + (sidx * 7) % O) % Q;
        const int o = q % O, lt = q / O, l = lt / T, t = lt - l * T;
        const double* ob = obstacles + ((size_t)b * O + o) * 12;
        const double* lg = link_gens + (((size_t)b * T + t) * J + l) * 18;
        unsigned long long cb[3] = {0ull, 0ull, 0ull};
        if (p < ARMOUR_NPLANES) {
            canonical_normal_of_plane(ob, lg, p, cb);
            tri[sidx][p][0] = cb[0]; tri[sidx][p][1] = cb[1]; tri[sidx][p][2] = cb[2];
        }
        __syncthreads();
        bool needed = p < ARMOUR_NPLANES && (cb[0] | cb[1] | cb[2]) != 0ull;
        for (int e2 = 0; e2 < p && needed; e2++) needed = !(tri[sidx][e2][0] == cb[0] && tri[sidx][e2][1] == cb[1] && tri[sidx][e2][2] == cb[2]);
        const unsigned long long live = __ballot(needed);
        // the class pre-pass's words of this problem, ANDed by all threads of the block
        unsigned long long m = ~0ull;
        for (int i = threadIdx.x; i < n_part; i += blockDim.x) m &= pre_part[(size_t)b * n_part + i];
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) {
            const unsigned lo = __shfl_xor((unsigned)m, o2, 64), hi = __shfl_xor((unsigned)(m >> 32), o2, 64);
            m &= ((unsigned long long)hi << 32) | lo;
        }
        if (p == 0) { s_live[sidx] = live; s_pre[sidx] = m; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long live = 0ull, pre = ~0ull;
        for (int i = 0; i < kPlaneSamples; i++) { live |= s_live[i]; pre &= s_pre[i]; }
        live_mask[b] = live;
        pre_mask[b] = pre;
    }
}

// Class pre-pass of the lean table: which planes are redundant in EVERY row of a problem by the class of their two generators alone
// -- a generator that is the zero vector or has exactly one non-zero component (every generator of an axis-aligned box obstacle and
// the three error generators diag(independent) of a link).  A pair with a zero generator, or two generators on the same axis, has the
// zero normal; two generators on different axes i, j have the normal +-e_k exactly (the one non-zero product of the cross product is
// x = fl(alpha * beta), the norm sqrt(fl(x * x)) = |x| exactly while x * x neither overflows nor underflows, x / |x| = +-1), which an
// earlier pair of the same class repeats bit for bit.  One thread per row, integer logic on nine class codes; the result is a
// subset of what the signature test of armour_p1_planes_kernel finds (that test still produces plane_skip), known BEFORE the planes are
// computed, so that the lean kernel neither computes nor stores those planes.
__global__ __launch_bounds__(256) void armour_p1_plane_class_kernel(int T, int J, int O, const double* __restrict__ link_gens,
                                                                    const double* __restrict__ obstacles, unsigned long long* __restrict__ part) {
    __shared__ unsigned long long s[4];
    const int Q = J * T * O;
    const int q = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    unsigned long long m = ~0ull;
    if (q < Q) {
        const int o = q % O, lt = q / O, l = lt / T, t = lt - l * T;
        const double* ob = obstacles + ((size_t)b * O + o) * 12;
        const double* lg = link_gens + (((size_t)b * T + t) * J + l) * 18;
        int cls[9];     // 0: zero vector; 1..3: only component x / y / z is non-zero; 4: anything else
        double mag[9];  // |the one non-zero component|
#pragma unroll
        for (int g = 0; g < 9; g++) {
            double v[3];
#pragma unroll
            for (int ax = 0; ax < 3; ax++) v[ax] = g < 3 ? ob[(g + 1) * 3 + ax] : lg[ax * 6 + (g - 3)];
            const int nz = (v[0] != 0.0) + (v[1] != 0.0) + (v[2] != 0.0);
            cls[g] = nz == 0 ? 0 : nz > 1 ? 4 : v[0] != 0.0 ? 1 : v[1] != 0.0 ? 2 : 3;
            mag[g] = fabs(v[0]) + fabs(v[1]) + fabs(v[2]);
            if (!(mag[g] == mag[g]) || mag[g] > 1e100) cls[g] = 4;   // NaN / inf / huge: no claim
        }
        m = 0ull;
        unsigned seen = 0u;   // bit k: an earlier pair of this row has the normal +-e_k by class
        int p = 0;
#pragma unroll
        for (int a = 0; a < 9; a++)
#pragma unroll
            for (int bb = a + 1; bb < 9; bb++, p++) {
                const int ca = cls[a], cb = cls[bb];
                bool red = false;
                if (ca == 0 || cb == 0) red = ca != 4 && cb != 4 ? true : false;   // (0 x anything finite = 0; a general partner is finite unless flagged above, but make no claim)
                else if (ca < 4 && cb < 4) {
                    if (ca == cb) red = true;
                    else {
                        const double x = mag[a] * mag[bb];
                        if (x > 1e-100 && x < 1e100) {   // x * x is a normal number: the norm is |x| exactly
                            const int k = 6 - ca - cb;
                            if (seen & (1u << k)) red = true; else seen |= 1u << k;
                        }
                    }
                }
                if (red) m |= 1ull << p;
            }
    }
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)m, o2, 64), hi = __shfl_xor((unsigned)(m >> 32), o2, 64);
        m &= ((unsigned long long)hi << 32) | lo;
    }
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[(size_t)b * gridDim.x + blockIdx.x] = s[0] & s[1] & s[2] & s[3];
}

// plane_skip[b] = AND of the planes kernel's per-wave masks of problem b
__global__ __launch_bounds__(256) void armour_p1_skip_reduce_kernel(const unsigned long long* __restrict__ part, int per_problem, unsigned long long* __restrict__ plane_skip) {
    __shared__ unsigned long long s[4];
    const int b = blockIdx.x;
    unsigned long long v = ~0ull;
    for (int i = threadIdx.x; i < per_problem; i += 256) v &= part[(size_t)b * per_problem + i];
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)v, o2, 64), hi = __shfl_xor((unsigned)(v >> 32), o2, 64);
        v &= ((unsigned long long)hi << 32) | lo;
    }
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) plane_skip[b] = s[0] & s[1] & s[2] & s[3];
}


// ------------------------------------------------------------------ test hook: one PZ operator on caller-supplied operands
// Operand block layout (doubles / u64 in two flat buffers): see armour_debug_pz_op in api.  ops:
//   0 mul 3x3*3x1   1 mul 3x3*3x3   2 mul 1x1*1x1   3 mul 1x1*3x1   4 a+b (3x1)   5 a-b (3x1)   6 addOneDimPZ(a 3x1, b 1x1, r)
//   7 stack(a,b,c 1x1)   8 cross(a 3x1, const)   9 cross(const, a 3x1)   10 cross(a 3x1, b 3x1)   11 sa*a + sb*b (1x1)
struct PzOpArgs {
    int op, nops, r;
    int sz[3], cnt[3];
    const uint64_t* keys[3];
    const double* coef[3];
    double cen[3][9], ind[3][9], ind2[3][9];
    double consts[4];
    uint64_t* out_keys; double* out_coef; double* out_misc;  // misc: cnt, cen[9], ind[9], ind2[9]
    int out_cap;
};

__global__ __launch_bounds__(64) P1_OCC void armour_p1_pzop_kernel(P1Cfg cf, const PzOpArgs* ap) {
    P1_PIN_ONE_WAVE_PER_SIMD();
    const PzOpArgs a = *ap;  // passed through memory: P1Cfg alone nearly fills the 4 KB kernel-argument segment
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Chain c;
    c.cf = &cf;
    c.n = cf.n; c.J = cf.J;
    c.L = make_layout(cf.J, cf.n, cf.capW, 1);
    c.arena = (GLB_AS unsigned char*)cf.arena;
    LDS_AS unsigned char* lds = (LDS_AS unsigned char*)smem;
    c.nw = 1; c.wid = 0; c.role = 0;
    c.w.skey = (LDS_AS pzkey_t*)lds;
    c.w.sidx = (LDS_AS uint16_t*)(lds + (size_t)cf.capKey * sizeof(pzkey_t));
    c.w.lstat = (LDS_AS int*)(lds + (size_t)cf.capKey * sizeof(pzkey_t) + (size_t)cf.capRaw * 2);
    c.w.mg = (LDS_AS double*)(lds + p1_wave_mg_off(cf.capKey, cf.capRaw));
    LDS_AS unsigned char* shared = lds + p1_wave_lds(cf.capKey, cf.capRaw);
    c.w.cnt = (LDS_AS int*)shared;
    c.mb = c.w.cnt + kMaxSlots;
    c.ci = (LDS_AS double*)(shared + ((((size_t)(kMaxSlots + kMbWords) * sizeof(int)) + 15) & ~(size_t)15));
    c.w.cap_raw = cf.capRaw; c.w.cap_key = cf.capKey;
    c.w.thr = cf.pr.simplify_threshold;
    c.w.thr_sq = sq_threshold(c.w.thr);
    c.w.lane = threadIdx.x;
    solo(c.w);
    if (threadIdx.x < ST_WORDS) c.w.lstat[threadIdx.x] = 0;
    margin_reset(c.w);
    c.freeV = (1ull << c.L.nV) - 1ull; c.freeS = (1u << kNS) - 1u;
    for (int i = threadIdx.x; i < kMaxSlots; i += WAVE) c.w.cnt[i] = 0;
    __syncthreads();
    PZ in[3];
    for (int o = 0; o < a.nops; o++) {
        // 3x3 operands: the first in a work slot, the second in a small (<= 8 monomials) slot like a joint rotation
        in[o] = a.sz[o] == 9 ? (o == 0 ? c.M(0) : c.JM(0)) : a.sz[o] == 3 ? c.allocV() : c.allocS();
        const int n = a.cnt[o], sz = a.sz[o];
        for (int m = threadIdx.x; m < n; m += WAVE) in[o].keys[m] = (pzkey_t)a.keys[o][m];
        for (int m = threadIdx.x; m < n * sz; m += WAVE) in[o].coef[m] = a.coef[o][m];
        if (threadIdx.x < sz) { in[o].cen[threadIdx.x] = a.cen[o][threadIdx.x]; in[o].ind[threadIdx.x] = a.ind[o][threadIdx.x]; in[o].ind2[threadIdx.x] = a.ind2[o][threadIdx.x]; }
        if (threadIdx.x == 0) c.w.cnt[in[o].id] = n;
        __syncthreads();
    }
    Wave& w = c.w;
    PZ out;
#ifdef P1_PROFILE
    for (int i = 0; i < PR_WORDS; i++) c.w.prof[i] = 0;
#endif
    const long long cyc0 = clock64();  // shader-clock cycles of the operator alone, reported in out_misc[30] (tools/dev/gpu_pzop_cost.py)
    switch (a.op) {
        case 0: out = c.mulMV(in[0], in[1]); break;
        case 1: out = c.M(1); mul<3, 3, 3, 3>(w, out, view(w, in[0]), view(w, in[1])); break;
        case 2: out = c.mulSS(view(w, in[0]), view(w, in[1])); break;
        case 3: out = c.mulSV(in[0], in[1]); break;
        case 4: out = c.add(in[0], in[1]); break;
        case 5: out = c.add(in[0], in[1], -1.0); break;
        case 6: out = c.addOneDim(in[0], in[1], a.r); break;
        case 7: out = c.stack(in[0], in[1], in[2]); break;
        case 8: out = c.crossPzMat(in[0], a.consts); break;
        case 9: out = c.crossMatPz(a.consts, in[0]); break;
        case 10: out = c.crossPzPz(in[0], in[1]); break;
        default: out = c.comb2(view(w, in[0]), a.consts[0], view(w, in[1]), a.consts[1]); break;
    }
    const long long cyc1 = clock64();
    const int n = w.cnt[out.id], sz = out.sz;
    if (threadIdx.x == 0) {
        a.out_misc[30] = (double)(cyc1 - cyc0); a.out_misc[31] = (double)w.lstat[ST_MAX_RAW];
#ifdef P1_PROFILE  // phase attribution of the one operator (slots PR_* of pz_wave.h)
        for (int i = 0; i < PR_WORDS && 32 + i < 64; i++) a.out_misc[32 + i] = (double)w.prof[i];
#endif
    }
    for (int m = threadIdx.x; m < n && m < a.out_cap; m += WAVE) a.out_keys[m] = (uint64_t)out.keys[m];   // (a 128-bit key build hands back the low word: the operator tests run on the 64-bit build)
    for (int m = threadIdx.x; m < n * sz && m < a.out_cap * sz; m += WAVE) a.out_coef[m] = out.coef[m];
    {   // the operator's prune margin (pz_wave.h Wave::mg): the smallest |squared norm - thr_sq| over the lanes -> out_misc[60]; thr_sq -> [61]
        double m = w.mg[threadIdx.x];
        for (int k = 32; k >= 1; k >>= 1) m = fmin(m, __shfl_xor(m, k, WAVE));
        if (threadIdx.x == 0) { a.out_misc[60] = m; a.out_misc[61] = w.thr_sq; }
    }
    if (threadIdx.x == 0) {
        a.out_misc[0] = n; a.out_misc[1] = sz; a.out_misc[2] = w.lstat[ST_ERR];
        for (int e = 0; e < sz; e++) { a.out_misc[3 + e] = out.cen[e]; a.out_misc[12 + e] = out.ind[e]; a.out_misc[21 + e] = out.ind2[e]; }
    }
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per kernel FUNCTION, not per launch: two host threads building on two handles
// (armour_batch_*: one thread per device slot, and a device may host two slots) must not interleave "set the attribute" and "launch".
// Held for the two calls only -- the kernels of different handles still run concurrently on their own streams.
static std::mutex g_p1_launch_mu;

// Work slots of the time-vectorised build: ONE arena per device, shared by every handle of the process (round 4).  A build holds the
// device's arena -- and its mutex -- from before its launch until the kernel has finished (the build is synchronous, and a launch of
// one block per compute unit leaves no room for another handle's beside it anyway), so two handles that build batches at the same
// time need the 111.6 MiB per block once, not twice.  The arena stays allocated between builds and goes with the device's last handle;
// ARMOUR_OPT_P1_KEEP_WORK_MEMORY = 0 releases it at the end of every build of that handle instead.  (Why keeping is the default --
// measured, tools/dev/gpu_p1_wall.py: releasing and re-allocating 28.6 GiB costs 0.6 ms of a 128-problem build's 10.4 ms as a rule, and 3.5 - 4.4 s
// twice in twelve builds, when the runtime has to get the memory back from the driver.)
struct TvArenaPool {
    std::mutex mu;
    unsigned char* ptr = nullptr;
    size_t bytes = 0;
    int users = 0;   // live handles of the device that have built something: the last one to go frees the arena
};
constexpr int kMaxPoolDevices = 64;
TvArenaPool g_tv_pool[kMaxPoolDevices];

struct P1Work {
    unsigned char* arena = nullptr;
    size_t arena_total = 0;
    int waves = 0;
    unsigned* d_status = nullptr;
    double* d_link_gens = nullptr; size_t gens_cap = 0;
    double* d_torque_radius = nullptr; size_t tr_cap = 0;
    double* d_obstacles = nullptr; size_t obs_cap = 0;
    int* d_retry = nullptr; size_t retry_cap = 0;  // [1 + B*T]: count, then item indices
    unsigned long long* d_skip_part = nullptr; size_t skip_part_cap = 0;  // [B][blocks per problem][4 waves]: the planes kernel's masks before the AND
    unsigned long long* d_margin = nullptr; size_t margin_cap = 0;   // [B]: the prune margin word of every problem (P1Cfg::margin)
    long long* d_phase = nullptr;   // ARMOUR_P1_TRACE: [1024][8] phase clocks of the last launch's blocks (P1Cfg::phase_log), allocated on first use
    unsigned char* d_xch = nullptr; size_t xch_cap = 0; int xch_epoch = 0;   // a time step on two CUs: kXchBytes per item, flags tagged with the launch's epoch (cleared when allocated and when the epoch wraps)
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;   // around the reach-set kernel | around the half-space kernels
};

#endif  // P1_TV_VARIANT
}  // namespace

// ---- launch of THIS translation unit's time-vectorised kernel (rows of tv::GR doubles; pz_tv.h).  p1_reach.hip holds the 64-double rows,
// p1_reach_tv50.hip the same source with rows of 50 (BASELINE's T = 100 makes groups of 50 time steps): armour_p1_build picks by group size.
#ifndef P1_TV_LAUNCH
#define P1_TV_LAUNCH armour_p1_tv_launch_g64
#endif
__attribute__((visibility("hidden"))) hipError_t P1_TV_LAUNCH(int nw_launch, int nw, int blocks, size_t smem, hipStream_t stream, const void* cfg) {
    const P1Cfg& cf = *static_cast<const P1Cfg*>(cfg);
    hipError_t e = hipSuccess;
    if (nw_launch == 8) {
        if constexpr (kTvDedicatedHelpers) {
            if ((e = hipFuncSetAttribute((const void*)tvchain::armour_p1_tv_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)) != hipSuccess) return e;
            hipLaunchKernelGGL(tvchain::armour_p1_tv_kernel<8>, dim3(blocks), dim3(WAVE * 8), smem, stream, cf);
        }
    } else if (nw == 4) {
        if ((e = hipFuncSetAttribute((const void*)tvchain::armour_p1_tv_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)) != hipSuccess) return e;
        hipLaunchKernelGGL(tvchain::armour_p1_tv_kernel<4>, dim3(blocks), dim3(WAVE * 4), smem, stream, cf);
    } else if (nw == kRoles) {
        if ((e = hipFuncSetAttribute((const void*)tvchain::armour_p1_tv_kernel<kRoles>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)) != hipSuccess) return e;
        hipLaunchKernelGGL(tvchain::armour_p1_tv_kernel<kRoles>, dim3(blocks), dim3(WAVE * kRoles), smem, stream, cf);
    } else {
        if ((e = hipFuncSetAttribute((const void*)tvchain::armour_p1_tv_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)) != hipSuccess) return e;
        hipLaunchKernelGGL(tvchain::armour_p1_tv_kernel<1>, dim3(blocks), dim3(WAVE), smem, stream, cf);
    }
    return hipGetLastError();
}

#ifndef P1_TV_VARIANT
__attribute__((visibility("hidden"))) hipError_t armour_p1_tv_launch_g50(int nw_launch, int nw, int blocks, size_t smem, hipStream_t stream, const void* cfg);   // p1_reach_tv50.hip
#ifndef P1_TV_NARROW
#define P1_TV_NARROW 50
#endif
constexpr int kTvNarrowRows = P1_TV_NARROW;   // the row width of that unit (its TV_GROW; development builds pass both)

void armour_p1_free(ArmourPlanner* h) {
    P1Work* wk = (P1Work*)h->p1;
    if (!wk) return;
    {   // the device's last handle takes the shared work slots with it
        TvArenaPool& pool = g_tv_pool[h->device % kMaxPoolDevices];
        std::lock_guard<std::mutex> lk(pool.mu);
        if (--pool.users <= 0) {
            pool.users = 0;
            if (pool.ptr) { (void)hipSetDevice(h->device); (void)hipFree(pool.ptr); pool.ptr = nullptr; pool.bytes = 0; }
        }
    }
    if (wk->arena) (void)hipFree(wk->arena);
    if (wk->d_status) (void)hipFree(wk->d_status);
    if (wk->d_link_gens) (void)hipFree(wk->d_link_gens);
    if (wk->d_torque_radius) (void)hipFree(wk->d_torque_radius);
    if (wk->d_margin) (void)hipFree(wk->d_margin);
    if (wk->d_xch) (void)hipFree(wk->d_xch);
    if (wk->d_phase) (void)hipFree(wk->d_phase);
    if (wk->d_obstacles) (void)hipFree(wk->d_obstacles);
    if (wk->d_skip_part) (void)hipFree(wk->d_skip_part);
    if (wk->d_retry) (void)hipFree(wk->d_retry);
    if (wk->ev0) (void)hipEventDestroy(wk->ev0);
    if (wk->ev1) (void)hipEventDestroy(wk->ev1);
    if (wk->ev2) (void)hipEventDestroy(wk->ev2);
    if (wk->ev3) (void)hipEventDestroy(wk->ev3);
    delete wk;
    h->p1 = nullptr;
}

template <class Tp>
static int grow(Tp** p, size_t* cap, size_t need) {
    if (need <= *cap && *p) return ARMOUR_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    HIPCHK(hipMalloc((void**)p, (need ? need : 1) * sizeof(Tp)));
    *cap = need;
    return ARMOUR_OK;
}


// test hook (include/armour_hip.h: armour_debug_pz_op)
int armour_p1_debug_pz_op(ArmourPlanner* h, int op, int nops, const int* sz, const int* cnt, const uint64_t* const* keys,
                          const double* const* coef, const double* cen, const double* ind, const double* ind2, const double* consts,
                          int r, int out_cap, uint64_t* out_keys, double* out_coef, double* out_misc) {
    HIPCHK(hipSetDevice(h->device));
    const int J = h->J, n = h->n;
    int cap_raw = 64;
    while (cap_raw < h->lim.raw_terms) cap_raw <<= 1;
    const Layout L = make_layout(J, n, h->lim.work_monomials, 1);
    const size_t ci_doubles = (size_t)L.nV * 9 + kNS * 3 + kNM * 27 + (size_t)L.nJM * 27 + (size_t)L.nJV * 9 + (size_t)L.nJS * 3;
    const size_t smem = p1_wave_lds(cap_raw, cap_raw) + p1_shared_lds(ci_doubles);
    HIPCHK(hipFuncSetAttribute((const void*)armour_p1_pzop_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    unsigned char* arena = nullptr;
    HIPCHK(hipMalloc((void**)&arena, L.total));
    PzOpArgs a;
    memset(&a, 0, sizeof(a));
    a.op = op; a.nops = nops; a.r = r; a.out_cap = out_cap;
    std::vector<void*> dev;
    auto up = [&](const void* src, size_t bytes) -> void* {
        void* d = nullptr;
        if (hipMalloc(&d, bytes ? bytes : 8) != hipSuccess) return nullptr;
        dev.push_back(d);
        if (bytes && src) (void)hipMemcpy(d, src, bytes, hipMemcpyHostToDevice);
        return d;
    };
    for (int o = 0; o < nops; o++) {
        a.sz[o] = sz[o]; a.cnt[o] = cnt[o];
        a.keys[o] = (const uint64_t*)up(keys[o], (size_t)cnt[o] * 8);
        a.coef[o] = (const double*)up(coef[o], (size_t)cnt[o] * sz[o] * 8);
        for (int e = 0; e < sz[o]; e++) { a.cen[o][e] = cen[o * 9 + e]; a.ind[o][e] = ind[o * 9 + e]; a.ind2[o][e] = ind2[o * 9 + e]; }
    }
    for (int i = 0; i < 4; i++) a.consts[i] = consts ? consts[i] : 0.0;
    a.out_keys = (uint64_t*)up(nullptr, (size_t)out_cap * 8);
    a.out_coef = (double*)up(nullptr, (size_t)out_cap * 9 * 8);
    a.out_misc = (double*)up(nullptr, 64 * 8);
    if (a.out_misc) (void)hipMemset(a.out_misc, 0, 64 * 8);
    P1Cfg cf;
    memset(&cf, 0, sizeof(cf));
    cf.B = 1; cf.T = h->T; cf.J = J; cf.n = n;
    cf.capW = h->lim.work_monomials; cf.capRaw = cap_raw; cf.capKey = cap_raw; /* (the one-operator test hook: full-width key buffers) */ cf.capL = h->lim.link_monomials; cf.capT = h->lim.torque_monomials;
    cf.arena_bytes = L.total; cf.arena = arena;
    cf.rb = h->robot; cf.pr = h->params; cf.ub = h->ub;
    const PzOpArgs* d_args = (const PzOpArgs*)up(&a, sizeof(a));
    hipLaunchKernelGGL(armour_p1_pzop_kernel, dim3(1), dim3(WAVE), smem, h->stream, cf, d_args);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) {
        (void)hipMemcpy(out_keys, a.out_keys, (size_t)out_cap * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(out_coef, a.out_coef, (size_t)out_cap * 9 * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(out_misc, a.out_misc, 64 * 8, hipMemcpyDeviceToHost);
    }
    for (void* d : dev) (void)hipFree(d);
    (void)hipFree(arena);
    if (e != hipSuccess) { armour_set_error("armour_debug_pz_op: %s", hipGetErrorString(e)); return ARMOUR_EDEVICE; }
    return ARMOUR_OK;
}

// Wait for the build's stream: a lone problem's build is ~1 ms, and an interrupt-driven wake-up after hipStreamSynchronize adds tens of
// microseconds to each of the build's waits; so poll for a while (2 ms) and only then sleep (batches take 8 ms and more: they sleep).
static hipError_t p1_wait_stream(hipStream_t st) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int spins = 0;; spins++) {
        const hipError_t q = hipStreamQuery(st);
        if (q != hipErrorNotReady) return q;
        if ((spins & 63) == 63 && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > 2.0) return hipStreamSynchronize(st);
    }
}

// Prune margin of every problem from the build's words (P1Cfg::margin): the smallest |s - thr_sq| over the squared norms s of every simplify()
// verdict (RT/PZsparse.cu:305-316), as |norm - threshold| / threshold.  Which side of the threshold the nearest verdict fell on is not recorded (the
// tracking is a subtraction and a minimum); the kept side's figure, sqrt(thr_sq + m) - thr, is the smaller of the two and differs from the other by a
// relative O(margin): exact where it matters.  The oracle's min_margin() (oracle/pz.hpp:131) is the same minimum, zero norms left out there and
// counted as a margin of 1 here.  1e300 for a problem without a verdict.
static void armour_margin_from_words(ArmourPlanner* h, const unsigned long long* w) {
    const double thr = h->params.simplify_threshold;
    double S = thr * thr;   // sq_threshold() of pz_wave.h: the largest double whose root is <= thr
    while (std::sqrt(std::nextafter(S, INFINITY)) <= thr) S = std::nextafter(S, INFINITY);
    while (S > 0.0 && std::sqrt(S) > thr) S = std::nextafter(S, -INFINITY);
    h->h_prune_margin.assign((size_t)h->B, 1e300);
    for (int b = 0; b < h->B; b++) {
        if (w[b] == 0) continue;
        const unsigned long long bits = 0x7ff0000000000000ull - w[b];
        double m;
        memcpy(&m, &bits, 8);
        h->h_prune_margin[(size_t)b] = std::fabs(std::sqrt(S + m) - thr) / thr;
    }
}

int armour_p1_build(ArmourPlanner* h, const double* obstacles) {
    if (!h->p1) {
        P1Work* nw = new P1Work();
        h->p1 = nw;
        { TvArenaPool& pool = g_tv_pool[h->device % kMaxPoolDevices]; std::lock_guard<std::mutex> lk(pool.mu); pool.users++; }
        HIPCHK(hipMalloc((void**)&nw->d_status, (ST_WORDS + 64) * sizeof(unsigned)));
        HIPCHK(hipEventCreate(&nw->ev0));
        HIPCHK(hipEventCreate(&nw->ev1));
        HIPCHK(hipEventCreate(&nw->ev2));
        HIPCHK(hipEventCreate(&nw->ev3));
    }
    P1Work* wk = (P1Work*)h->p1;
    const int B = h->B, T = h->T, J = h->J, n = h->n, O = h->O;
    int rc;
    if ((rc = grow(&wk->d_link_gens, &wk->gens_cap, (size_t)B * T * J * 18)) != ARMOUR_OK) return rc;
    if ((rc = grow(&wk->d_torque_radius, &wk->tr_cap, (size_t)B * n * T)) != ARMOUR_OK) return rc;
    if ((rc = grow(&wk->d_margin, &wk->margin_cap, (size_t)B)) != ARMOUR_OK) return rc;
    if ((rc = grow(&wk->d_obstacles, &wk->obs_cap, (size_t)B * O * 12)) != ARMOUR_OK) return rc;
    if (O > 0) HIPCHK(hipMemcpyAsync(wk->d_obstacles, obstacles, (size_t)B * O * 12 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    armour_build_stamp("buffers+obstacles");

    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, h->device));
    if (h->p1_hint_B != B || h->p1_hint_T != T || h->p1_hint_O != O || ++h->p1_hint_builds >= 64) {
        h->p1_step_cap_hint = 0; h->p1_tv_shape_hint = 0; h->p1_hint_builds = 0;
        h->p1_hint_B = B; h->p1_hint_T = T; h->p1_hint_O = O;
    }
    int cap_raw = 64;
    while (cap_raw < h->lim.raw_terms || cap_raw < h->p1_step_cap_hint) cap_raw <<= 1;   // (the hint: what the last build of this handle ended up with)
    if ((rc = grow(&wk->d_retry, &wk->retry_cap, (size_t)1 + (size_t)B * T)) != ARMOUR_OK) return rc;
    const Layout L1 = make_layout(J, n, h->lim.work_monomials, 1), L3 = make_layout(J, n, h->lim.work_monomials, kRoles), L4 = make_layout(J, n, h->lim.work_monomials, 4);
    if (L4.idJS + L4.nJS > kMaxSlots) { armour_set_error("slot table too small"); return ARMOUR_EINVAL; }
    auto ci_doubles = [&](const Layout& L) { return (size_t)L.nV * 9 + kNS * 3 + kNM * 27 + (size_t)L.nJM * 27 + (size_t)L.nJV * 9 + (size_t)L.nJS * 3; };
    auto lds_bytes = [&](int cap, int nw = 1) {
        return (size_t)(nw == 4 ? 3 : nw) * p1_wave_lds(key_cap(cap), cap) + (nw == 4 ? p1_wave_lds(kFkCapKey, kFkCapRaw) : 0) + p1_shared_lds(ci_doubles(nw == 1 ? L1 : nw == kRoles ? L3 : L4));
    };
    const int max_waves_env = std::min(h->tune(ARMOUR_OPT_P1_MAX_WAVES_PER_CU), 4 * P1_WAVES_PER_SIMD);
    auto waves_per_cu = [&](int cap) { return std::max(1, std::min(max_waves_env, (int)((size_t)160 * 1024 / lds_bytes(cap)))); };
    float total_ms = 0;
    h->build_info[0] = h->build_info[1] = h->build_info[2] = h->build_info[3] = 0;
    unsigned st[ST_WORDS];
    // one launch of the chain kernel over `n_items` work items (d_items == nullptr: all of them) with sort buffers of `cap`
    // entries; with `collect` the items that overflow them are listed in wk->d_retry instead of failing the launch
    // (`defer`: the launch is queued and NOT waited for -- the caller queues the half-space kernels and the read-back behind it, waits once and
    //  calls launch_done(); the status words then come back through a page-locked block, a copy into pageable memory would wait by itself)
    unsigned* st_pin = reinterpret_cast<unsigned*>(armour_handle_pinned(h, 9, 64));
    if (!st_pin) { armour_set_error("page-locked status block"); return ARMOUR_EDEVICE; }
    std::function<int()> launch_done = [] { return ARMOUR_OK; };
    auto launch = [&](int cap, const int* d_items, int n_items, bool collect, bool defer = false) -> int {
        // few items (at most one per CU): three waves per item, the roles of a time step run concurrently (latency);
        // otherwise one wave per item and as many items per CU as the LDS holds (throughput)
        const int nw_env = h->tune(ARMOUR_OPT_P1_STEP_WAVES);   // (0: automatic)
        const int cf_two_cu_env = h->tune(ARMOUR_OPT_P1_STEP_TWO_CU);
        // (round 3) ... FOUR waves when the block's LDS holds the fourth wave's small sort buffers as well: the forward kinematics,
        // the omega recursion and the constant cross products of the linear acceleration leave the three recursion waves for a wave
        // of their own (run_rnea_free, the choreography of the time-vectorised kernel's four-wave blocks) and no item is issued twice
        const int free_env0 = h->tune(ARMOUR_OPT_P1_STEP_FREE);
        const bool fits4 = free_env0 && lds_bytes(cap, 4) <= (size_t)160 * 1024;
        // Four-wave blocks take an item in 1.17 ms, one per CU at a time; one-wave blocks in 2.7 ms, waves_per_cu of them per CU.  With more
        // items than CUs the blocks loop, and what decides is the number of rounds either shape needs (measured, B problems of 100 steps,
        // four waves against one: B = 3..5 2.3-2.6 against 2.8-3.1 ms, 6..7 3.5-3.7 against 3.2-3.5, 8..10 4.4-4.8 against 5.2-5.7,
        // 12 6.0 either way, 14..15 7.0 against 6.3: profiles/r03_p1_four_waves.txt; re-measured at the end of round 3 with both shapes faster:
        // a round of four-wave items 1.0 ms, of one-wave items 2.6 ms -- B = 11 5.07 against 5.27, 16 7.16 against 7.70, 14 6.2 against 5.7)
        const int cus = prop.multiProcessorCount;
        const long long r4 = (n_items + cus - 1) / cus, r1 = (n_items + (long long)cus * waves_per_cu(cap) - 1) / ((long long)cus * waves_per_cu(cap));
        const bool multi = h->no_torque() ? false  // forward kinematics only: a single role
                           : nw_env ? nw_env >= 3
                           : collect ? false
                           : fits4 ? r4 <= 5 && 100 * r4 < 258 * r1   // (round 4, after the merge-path sorts and the item queue, B problems of 100 steps, four waves | one: 5: 2.22 | 2.66, 8: 3.73 | 3.76, 10: 4.03 | 4.46, 12: 4.83 | 4.94, 13: 5.63 | 5.04, 15: 5.90 | 5.28 -- from the sixth round on the one-wave blocks' throughput wins)
                                   : (n_items <= cus && lds_bytes(cap, kRoles) <= (size_t)160 * 1024);
        const bool four = multi && free_env0 && (nw_env ? nw_env == 4 : fits4);
        const bool three = multi && !four;
        const int nw = four ? 4 : three ? kRoles : 1;
        const Layout& L = four ? L4 : three ? L3 : L1;
        const size_t smem = lds_bytes(cap, nw);
        if (smem > (size_t)160 * 1024) { armour_set_error("raw_terms=%d does not fit the 160 KiB LDS", cap); return ARMOUR_EINVAL; }
        const int per_cu = multi ? 1 : waves_per_cu(cap);
        // with at least as many idle CUs as items, the forward kinematics of every item runs as an item of its own
        const int split_env = h->tune(ARMOUR_OPT_P1_STEP_SPLIT_FK);   // (-1: automatic)
        const bool split = split_env >= 0 ? (split_env != 0 && multi) : (multi && n_items <= prop.multiProcessorCount);   // (round 4: also when the two kinds of items together are more than the blocks -- the blocks draw the RNEA items first and fill up with the forward kinematics: B = 2 1.095 -> 1.081 ms)
        const int fk_items = split ? n_items : 0;
        // A time step on two CUs (p1_free.inc.h): every item gets a helper block on a CU of its own, which also runs the item's forward kinematics.
        // Block k builds item k, block helper0 + k helps it; helper0 a multiple of 8 puts the two on one XCD (checked on the device, item by item).
        const int helper0 = (n_items + 7) & ~7;
        const bool two_cu = four && split && !collect && !d_items && cf_two_cu_env && !h->p1_two_cu_off && h->mode == ARMOUR_MODE_ARMOUR && helper0 + n_items <= prop.multiProcessorCount * per_cu
                            && h->tune(ARMOUR_OPT_P1_STEP_FREE) && h->tune(ARMOUR_OPT_P1_STEP_AUX3);   // (the choreography it changes is the free-running one with the w_aux recursion on the fourth wave)
        const int waves = two_cu ? helper0 + n_items : std::min(n_items + fk_items, prop.multiProcessorCount * per_cu);
        if ((size_t)waves * L.total > wk->arena_total) {
            if (wk->arena) (void)hipFree(wk->arena);
            wk->arena = nullptr;
            HIPCHK(hipMalloc((void**)&wk->arena, (size_t)waves * L.total));
            wk->arena_total = (size_t)waves * L.total;
        }
        P1Cfg cf;
        memset(&cf, 0, sizeof(cf));
        cf.B = B; cf.T = T; cf.J = J; cf.n = n; cf.O = O;
        cf.capW = h->lim.work_monomials; cf.capRaw = cap; cf.capKey = key_cap(cap); cf.capL = h->lim.link_monomials; cf.capT = h->lim.torque_monomials;
        cf.arena_bytes = L.total; cf.arena = wk->arena;
        const int free_env = h->tune(ARMOUR_OPT_P1_STEP_FREE);   // (0 = a barrier per joint)
        cf.free_running = free_env;
        cf.rb = h->robot; cf.pr = h->params; cf.ub = h->ub;
        cf.bez = h->d_bez;
        cf.mode = h->mode; cf.fk_only = h->no_torque() ? 1 : 0; cf.jrs = h->d_jrs;
        cf.link_count = h->d_link_count; cf.link_center = h->d_link_center; cf.link_indep = h->d_link_indep;
        cf.link_keys = h->d_link_keys; cf.link_coeff = h->d_link_coeff;
        cf.tq_count = h->d_tq_count; cf.tq_center = h->d_tq_center; cf.tq_indep = h->d_tq_indep;
        cf.tq_keys = h->d_tq_keys; cf.tq_coeff = h->d_tq_coeff;
        cf.link_gens = wk->d_link_gens; cf.torque_radius = wk->d_torque_radius; cf.status = wk->d_status;
        cf.margin = wk->d_margin;
        if (!d_items) HIPCHK(hipMemsetAsync(wk->d_margin, 0, (size_t)B * sizeof(unsigned long long), h->stream));   // (a launch over ALL items starts the margins afresh; the second pass over a few overflowed items adds to them)
        cf.items = d_items; cf.n_items = n_items; cf.fk_items = fk_items;
        const int aux3_env = h->tune(ARMOUR_OPT_P1_STEP_AUX3);
        cf.tv_aux_on_fk_wave = aux3_env;   // (four-wave blocks: the w_aux recursion next to omega on the fourth wave)
        cf.step_pairs = h->tune(ARMOUR_OPT_P1_STEP_PAIRS) == 2 || (h->tune(ARMOUR_OPT_P1_STEP_PAIRS) == 1 && n_items + fk_items <= waves);   // (1: with a block per item only -- after the merge-path sorts the pairs are worth 1 % to a lone problem and cost 2 % when the blocks loop: B = 4 1.95 against 1.99 ms, 8: 3.73 against 3.82; 2: always)
        cf.queue_order = h->tune(ARMOUR_OPT_P1_STEP_QUEUE);
        cf.queue = cf.queue_order != 0 && n_items + fk_items > waves ? wk->d_status + ST_WORDS + 60 : nullptr;   // (with a block per item there is nothing to draw: B = 1 1.044 against 1.069 ms.  The word: one of the status block that the profile counters of development builds do not reach)
        if (cf.queue) HIPCHK(hipMemsetAsync(cf.queue, 0, sizeof(unsigned), h->stream));
        cf.tail_cross = split ? h->tune(ARMOUR_OPT_P1_STEP_TAIL_CROSS) : 0;   // (with its forward kinematics to do the fourth wave has no time to spare)
        cf.retry_list = collect ? wk->d_retry + 1 : nullptr; cf.retry_count = reinterpret_cast<unsigned*>(wk->d_retry);
        int phase_blocks = 0;
        if (armour_trace_p1() && nw == 4 && waves <= 1024) {
            if (!wk->d_phase) HIPCHK(hipMalloc((void**)&wk->d_phase, 1024 * 8 * sizeof(long long)));
            HIPCHK(hipMemsetAsync(wk->d_phase, 0, 1024 * 8 * sizeof(long long), h->stream));
            cf.phase_log = wk->d_phase; phase_blocks = waves;
        }
        cf.two_cu = two_cu ? cf_two_cu_env : 0; cf.helper0 = helper0;
        cf.lean_back = h->tune(ARMOUR_OPT_P1_STEP_LEAN_BACK) & 1; cf.late_jrs = (h->tune(ARMOUR_OPT_P1_STEP_LEAN_BACK) & 2) == 0;   // (development: + 2 builds every joint before the roles begin)
        if (two_cu) {
            const size_t need = (size_t)n_items * kXchBytes;
            bool clear = false;
            if (need > wk->xch_cap) {
                if (wk->d_xch) (void)hipFree(wk->d_xch);
                wk->d_xch = nullptr; wk->xch_cap = 0;
                HIPCHK(hipMalloc((void**)&wk->d_xch, need));
                wk->xch_cap = need; clear = true;
            }
            if (++wk->xch_epoch >= (1 << 23)) { wk->xch_epoch = 1; clear = true; }
            if (clear) HIPCHK(hipMemsetAsync(wk->d_xch, 0, wk->xch_cap, h->stream));
            cf.xch = wk->d_xch; cf.xch_epoch = wk->xch_epoch;
        }
        HIPCHK(hipMemsetAsync(wk->d_status, 0, ST_WORDS * sizeof(unsigned), h->stream));
        if (collect) HIPCHK(hipMemsetAsync(wk->d_retry, 0, sizeof(int), h->stream));
        HIPCHK(hipEventRecord(wk->ev0, h->stream));
        {
            std::lock_guard<std::mutex> lk(g_p1_launch_mu);
            if (four) HIPCHK(hipFuncSetAttribute((const void*)armour_p1_chain_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            else if (three) HIPCHK(hipFuncSetAttribute((const void*)armour_p1_chain_kernel<kRoles>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            else HIPCHK(hipFuncSetAttribute((const void*)armour_p1_chain_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            if (four) hipLaunchKernelGGL(armour_p1_chain_kernel<4>, dim3(waves), dim3(WAVE * 4), smem, h->stream, cf);
            else if (three) hipLaunchKernelGGL(armour_p1_chain_kernel<kRoles>, dim3(waves), dim3(WAVE * kRoles), smem, h->stream, cf);
            else hipLaunchKernelGGL(armour_p1_chain_kernel<1>, dim3(waves), dim3(WAVE), smem, h->stream, cf);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipEventRecord(wk->ev1, h->stream));
        HIPCHK(hipMemcpyAsync(st_pin, wk->d_status, sizeof(st), hipMemcpyDeviceToHost, h->stream));
        armour_build_stamp("launch-queued");
        launch_done = [&, cap, n_items, waves, nw, two_cu, per_cu, smem, helper0, phase_blocks]() -> int {
        memcpy(st, st_pin, sizeof(st));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, wk->ev0, wk->ev1));
        total_ms += ms;
        h->build_info[0] = ARMOUR_P1_KERNEL_PER_STEP; h->build_info[1] = nw; h->build_info[2] = cap; h->build_info[3]++;
#ifdef P1_PROFILE
        {
            unsigned long long pr[PR_WORDS];
            HIPCHK(hipMemcpy(pr, wk->d_status + ST_WORDS, sizeof(pr), hipMemcpyDeviceToHost));
            fprintf(stderr, "[P1 profile, wave 0] cycles: fill %llu sort %llu emit %llu abs_sum %llu total %llu | simplify calls %llu (N<=64: %llu) raw terms %llu\n",
                    pr[PR_FILL], pr[PR_SORT], pr[PR_EMIT], pr[PR_ABS], pr[PR_TOTAL], pr[PR_CALLS], pr[PR_SMALL], pr[PR_TERMS]);
            fprintf(stderr, "[P1 profile, wave 0] whole-call cycles: N<=64 %llu | 64<N<=512: %llu in %llu calls (%llu terms) | N>512: %llu in %llu calls (%llu terms)\n",
                    pr[PR_CYC64], pr[PR_CYC512], pr[PR_N512], pr[PR_TERMS512], pr[PR_CYCBIG], pr[PR_NBIG], pr[PR_TERMSBIG]);
        }
#endif
        if (armour_trace_p1() && nw == 4 && wk->d_phase && phase_blocks > 0) {   // where the blocks' time went (wave 0 of each): median / max over the main blocks, then over the helpers
            std::vector<long long> ph((size_t)phase_blocks * 8);
            HIPCHK(hipMemcpy(ph.data(), wk->d_phase, ph.size() * sizeof(long long), hipMemcpyDeviceToHost));
            const char* names[5] = {"JRS", "decision", "forward pass", "backward pass", "tables"};
            for (int grp = 0; grp < (two_cu ? 2 : 1); grp++) {
                const int b0 = grp == 0 ? 0 : helper0, b1 = grp == 0 ? n_items : helper0 + n_items;
                fprintf(stderr, "[P1 phases, %s blocks, cycles median / max]", grp == 0 ? "main" : "helper");
                for (int k = 0; k < 5; k++) {
                    std::vector<long long> d;
                    for (int bl = b0; bl < b1 && bl < phase_blocks; bl++) { const long long* r = &ph[(size_t)bl * 8]; if (r[k + 1] > 0 && r[k] > 0) d.push_back(r[k + 1] - r[k]); }
                    if (d.empty()) continue;
                    std::sort(d.begin(), d.end());
                    fprintf(stderr, " %s %lld / %lld", names[k], d[d.size() / 2], d.back());
                }
                std::vector<long long> tot;
                long long first = 0, last = 0;
                for (int bl = b0; bl < b1 && bl < phase_blocks; bl++) { const long long* r = &ph[(size_t)bl * 8]; if (r[5] > 0) { tot.push_back(r[5] - r[0]); if (!first || r[0] < first) first = r[0]; if (r[5] > last) last = r[5]; } }
                if (!tot.empty()) { std::sort(tot.begin(), tot.end()); fprintf(stderr, " | item %lld / %lld", tot[tot.size() / 2], tot.back()); }
                int local = 0, xcd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int bl = b0; bl < b1 && bl < phase_blocks; bl++) { const long long* r = &ph[(size_t)bl * 8]; if (r[6] == 2) { local++; fprintf(stderr, " [block %d on XCD %lld: one CU]", bl, r[7]); } xcd[r[7] & 7]++; }
                fprintf(stderr, " | %d of %d on their own; blocks per XCD %d %d %d %d %d %d %d %d\n", local, b1 - b0, xcd[0], xcd[1], xcd[2], xcd[3], xcd[4], xcd[5], xcd[6], xcd[7]);
            }
        }
        if (armour_trace_p1()) fprintf(stderr, "[P1] %d items, cap_raw %d: %d blocks of %d wave(s)%s (%d per CU, %zu B LDS), %.2f ms, flags 0x%x, max raw terms %u, max monomials %u\n", n_items, cap, waves, nw, two_cu ? ", two CUs per item" : "", per_cu, smem, ms, st[ST_ERR], st[ST_MAX_RAW], st[ST_MAX_OUT]);
        return ARMOUR_OK;
        };
        if (defer) return ARMOUR_OK;
        HIPCHK(p1_wait_stream(h->stream));
        armour_build_stamp("reach-set-kernel-waited");
        return launch_done();
    };
    auto other_errors = [&]() -> int {
        if (st[ST_ERR] & (unsigned)ERR_PAIR) {   // not a capacity limit: two waves that share an operator lost each other (psync), their barriers went off
            armour_set_error("reach-set build: a pair of waves lost its barrier (flags 0x%x) -- an internal synchronisation error, not a capacity limit; "
                             "ARMOUR_OPT_P1_STEP_PAIRS = 0 builds without paired waves", st[ST_ERR]);
            return ARMOUR_EDEVICE;
        }
        if (st[ST_ERR] & ~(unsigned)ERR_RAW_OVERFLOW) {
            armour_set_error("[dbg word3=%u] reach-set build overflow (flags 0x%x: 2=work_monomials, 4=link/torque_monomials, 8=link generators); max raw terms %u, max monomials %u",
                             st[3], st[ST_ERR], st[ST_MAX_RAW], st[ST_MAX_OUT]);
            return ARMOUR_ECAPACITY;
        }
        return ARMOUR_OK;
    };
    // Batches with more items than three waves per CU hold run a first pass with 2048-entry sort buffers -- 32 KB of LDS
    // per wave, so FOUR waves per CU -- and list the few items that overflow them; those alone are rebuilt with the full
    // buffers.  Small batches gain nothing from the fourth wave and go straight to the full buffers.
    if (h->no_torque()) {  // no torque tables in these modes: empty PZs, zero radius (RT/armour_main.cu:172-175: the radius stays zero)
        HIPCHK(hipMemsetAsync(h->d_tq_count, 0, (size_t)B * n * T * sizeof(int), h->stream));
        HIPCHK(hipMemsetAsync(wk->d_torque_radius, 0, (size_t)B * n * T * sizeof(double), h->stream));
    }
    // ---- Time-vectorised build (p1_tv.inc.h): one wave per (problem, group of <= 64 time steps), the symbolic work of an
    // operator done once per group.  Batches only: a single problem has two groups, i.e. two waves, and is faster step by step
    // (DESIGN.md 4.2).  Any capacity flag sends the whole batch down the per-step path below.
    bool built = false;
    const int tv_min_groups = h->tune(ARMOUR_OPT_P1_TV_MIN_GROUPS);  // default 31: below this the per-step kernel is faster (re-measured at the end of round 4, B problems of 100 steps, per-step against time-vectorised: 14: 5.84 / 6.35 ms, 16: 7.49 / 6.64 -- the per-step kernel steps up with every 768 items -- 18: 8.04 / 6.66; round 3: 36, profiles/r03_p1_breakeven.txt)
    const bool armtd = h->no_torque();  // comparison mode, or ARMOUR without input constraints: forward kinematics only -- every item is a forward-kinematics item
    // (its chain is a fifth of the RNEA chain: the per-step kernel stays ahead up to B = 30 there)
    // (the break-even is one of WORK: the per-step kernel's time grows with B * T, a chain's latency hardly depends on the lanes in
    //  use -- 64 problems of 20 time steps are faster step by step -- so the threshold counts groups of 50 time steps' worth of items)
    // (ARMOUR_OPT_P1_BUILD, include/armour_hip.h: a handle can be held to one of the two kernels)
    const bool want_tv = h->opt_p1_build == 1 ? false : h->opt_p1_build == 2 ? true
                         : (long long)B * T >= 50ll * (armtd ? tv_min_groups * 5 / 3 : tv_min_groups);   // (comparison mode: 60 groups -- 2.97 against 2.59 ms at B = 32, 2.40 against 2.57 at B = 28)
    if (want_tv) {
        const int G = (T + 63) / 64, LG = (T + G - 1) / G, groups = B * G;
        // rows of 50 doubles when a group's time steps fit them (T = 100: two groups of 50), of 64 otherwise (the reference's T = 128): pz_tv.h, TV_GROW
        static_assert(tv::GR == 64, "this unit holds the kernel with full-width rows");
        const int tv_rows_env = h->tune(ARMOUR_OPT_P1_TV_ROW_WIDTH);   // (0 automatic | 50 | 64)
        const int gr_multi = (tv_rows_env == 64 || LG > kTvNarrowRows) ? 64 : kTvNarrowRows;
        const int capTv = h->lim.work_monomials;
        // Block shapes, in the order tried: (a) while there is at most one group per CU, three waves per group -- the roles of
        // run_rnea run concurrently, as in the per-step kernel's small launches: the latency of ONE chain is what such a batch pays;
        // (b) one wave per group with sort buffers of 4096 entries (the union raw-term counts seen are <= 2.4 k), several waves per
        // CU, the forward kinematics of every group as a work item of its own while slots are free; (c) the same with 8192 entries.
        // A product that overflows the sort buffers sends the launch to the next shape; any other flag to the per-step path.
        const int tv_free_env = h->tune(ARMOUR_OPT_P1_TV_FREE);          // (0 = a barrier per joint)
        const int tv_nw_env = h->tune(ARMOUR_OPT_P1_TV_WAVES);            // (0 automatic | 1 | 3 | 4 | 8)
        const int tv_split_env = h->tune(ARMOUR_OPT_P1_TV_SPLIT_FK);      // (-1 automatic)
        const int tv_ded_env = h->tune(ARMOUR_OPT_P1_TV_DEDICATED);       // (0 = no eight-wave blocks)
        const int tv_shift_env = h->tune(ARMOUR_OPT_P1_TV_HELP_SHIFT);
        struct Shape { int nw, cap; };
        // (8: four role waves + four dedicated helper waves, two waves per SIMD -- only in a build whose operators fit 256 registers)
        const Shape shapes[5] = {{8, 4096}, {4, 4096}, {kRoles, 4096}, {1, 4096}, {1, 8192}};
        TvArenaPool& pool = g_tv_pool[h->device % kMaxPoolDevices];
        std::unique_lock<std::mutex> pool_lk(pool.mu);   // held until the last launch of this build has finished
        // ARMOUR_OPT_P1_KEEP_WORK_MEMORY = 0: released at the end of the build -- on every way out: success, an error return, and when the build
        // falls through to the per-step kernel (a cap below one block, a capacity flag): nothing of the time-vectorised build outlives it then.
        struct PoolRelease {
            TvArenaPool& p; bool keep;
            ~PoolRelease() { if (p.ptr && !keep) { (void)hipFree(p.ptr); p.ptr = nullptr; p.bytes = 0; } }
        } pool_release{pool, h->opt_p1_keep_work != 0};
        if (pool.ptr && h->opt_p1_work_mb > 0 && pool.bytes > (size_t)(h->opt_p1_work_mb * 1048576.0)) {   // a kept arena larger than this handle's cap: give it back first
            (void)hipFree(pool.ptr);
            pool.ptr = nullptr; pool.bytes = 0;
        }
        for (int si = tv_nw_env > 0 ? 0 : std::min(h->p1_tv_shape_hint, 4); si < 5 && !built; si++) {   // (a forced block shape searches from the beginning)
            const int nw_launch = shapes[si].nw, cap = shapes[si].cap;
            if (nw_launch == 8 && (!kTvDedicatedHelpers || !tv_ded_env || !tv_free_env)) continue;
            const int nw = nw_launch == 8 ? 4 : nw_launch;   // the waves that play roles
            const int nhelp = nw_launch == 8 ? 4 : 0;
            const bool multi = nw > 1;   // (a): the roles of the RNEA on three waves, with four the forward kinematics on a wave of its own
            // (multi-wave blocks also when there are more groups than CUs: the blocks loop over the groups.  Measured against one wave per group,
            //  two or three groups per CU at a time -- B = 160: 18.1 against 23.1 ms, 256: 20.0 against 24.1, 384: 29.7 against 43.9, 512: 38.8 against
            //  47.2 (profiles/r03_p1_breakeven.txt) -- the shared walks are what the one-wave shape lacks)
            if (multi && (armtd || tv_nw_env == 1)) continue;
            if (multi && tv_nw_env > 1 && tv_nw_env != nw_launch) continue;
            if (!multi && tv_nw_env > 1 && si == 3) continue;
            // (Until the return-address guard of pz_wave.h -- PZ_KEEP_RETURN_ADDRESS -- the ONE-WAVE blocks had to stay on the 64-double rows: their
            //  kernel ended in a memory-aperture violation with narrower rows, in every launch.  tv::mul<3,3,3,3> -- whose staged-operand walks only the
            //  one-wave blocks' large staging area reaches -- is 150 KB long in the narrow-row unit, the compiler relaxed two of its branches through
            //  s[30:31] without saving the return address, and the function returned into its own middle: DESIGN.md 7, profiles/r05_tv_one_wave_narrow_rows.txt.)
            const int gr = gr_multi;
            const tvchain::TLayout TL = tvchain::make_tlayout(J, n, capTv, nw, nhelp, gr);
            // blocks per CU by LDS; the staging area takes what is left
            const size_t fixed = (nw == 4 ? 3 * tvchain::tv_lds_fixed(cap) + tvchain::tv_lds_fixed_fk() : (size_t)nw * tvchain::tv_lds_fixed(cap)) + tvchain::tv_lds_shared();
            const int per_cu = multi ? 1 : std::max(1, std::min(4, (int)((size_t)160 * 1024 / (fixed + 24 * 1024))));
            if (fixed + 512 * (multi ? 37 + 2 * (nw - 1) : 16) + 256 > (size_t)160 * 1024 / per_cu) continue;
            const int stage_total = (int)(((size_t)160 * 1024 / per_cu - fixed - 256) / 512);
            // several waves: wave 1 first gets the 37 rows a joint rotation needs, the rest is shared out evenly
            const int stage_rows = !multi ? stage_total : std::min(stage_total - 2 * (nw - 1), std::max(stage_total / nw, 37));
            const int stage_other = !multi ? 0 : (stage_total - stage_rows) / (nw - 1);
            const size_t smem = tvchain::tv_lds_bytes(cap, stage_rows, stage_other, nw);
            const int slots = std::min(512, prop.multiProcessorCount * per_cu);
            const bool split = armtd ? true : multi ? false : (tv_split_env >= 0 ? tv_split_env != 0 : 2 * groups <= slots);
            const int fk_items = split ? groups : 0, rnea_items = armtd ? 0 : groups;
            int blocks = std::min(rnea_items + fk_items, slots);
            if (h->opt_p1_work_mb > 0) {   // ARMOUR_OPT_P1_WORK_MEMORY_MB: fewer blocks, looping over the groups
                const double max_blocks = h->opt_p1_work_mb * 1048576.0 / (double)TL.total;
                if (max_blocks < 1.0) continue;   // (not even one block of this shape: the next shape, in the end the per-step kernel)
                blocks = (int)std::min((double)blocks, max_blocks);
            }
            if ((size_t)blocks * TL.total > pool.bytes) {
                if (pool.ptr) (void)hipFree(pool.ptr);
                pool.ptr = nullptr; pool.bytes = 0;
                if (hipMalloc((void**)&pool.ptr, (size_t)blocks * TL.total) != hipSuccess) { (void)hipGetLastError(); pool.ptr = nullptr; break; }
                pool.bytes = (size_t)blocks * TL.total;
            }
            P1Cfg cf;
            memset(&cf, 0, sizeof(cf));
            cf.B = B; cf.T = T; cf.J = J; cf.n = n; cf.O = O;
            cf.capW = capTv; cf.capRaw = cap; cf.capKey = key_cap(cap); cf.capL = h->lim.link_monomials; cf.capT = h->lim.torque_monomials;
            cf.arena_bytes = TL.total; cf.arena = pool.ptr;
            cf.rb = h->robot; cf.pr = h->params; cf.ub = h->ub;
            cf.bez = h->d_bez;
            cf.mode = h->mode; cf.fk_only = h->no_torque() ? 1 : 0; cf.jrs = h->d_jrs;
            cf.link_count = h->d_link_count; cf.link_center = h->d_link_center; cf.link_indep = h->d_link_indep;
            cf.link_keys = h->d_link_keys; cf.link_coeff = h->d_link_coeff;
            cf.tq_count = h->d_tq_count; cf.tq_center = h->d_tq_center; cf.tq_indep = h->d_tq_indep;
            cf.tq_keys = h->d_tq_keys; cf.tq_coeff = h->d_tq_coeff;
            cf.link_gens = wk->d_link_gens; cf.torque_radius = wk->d_torque_radius; cf.status = wk->d_status;
            cf.margin = wk->d_margin;
            HIPCHK(hipMemsetAsync(wk->d_margin, 0, (size_t)B * sizeof(unsigned long long), h->stream));
            cf.n_items = rnea_items; cf.fk_items = fk_items; cf.tv_groups = G; cf.tv_lanes = LG; cf.tv_cap = capTv; cf.tv_stage_rows = stage_rows; cf.tv_stage_rows_other = stage_other; cf.tv_free_running = tv_free_env;
            const int tv_help_env = h->tune(ARMOUR_OPT_P1_TV_HELPERS);   // (0 = every walk on its own wave)
            cf.tv_walk_helpers = tv_help_env;
            const int tv_help_min_env = h->tune(ARMOUR_OPT_P1_TV_HELP_MIN);
            const int tv_help_n_env = h->tune(ARMOUR_OPT_P1_TV_HELP_N);
            cf.tv_help_min = tv_help_min_env; cf.tv_help_n = tv_help_n_env; cf.tv_help_shift = tv_shift_env;
            const int tv_aux3_env = h->tune(ARMOUR_OPT_P1_TV_AUX3);
            cf.tv_aux_on_fk_wave = tv_aux3_env;
            cf.tail_cross = h->tune(ARMOUR_OPT_P1_TV_TAIL_CROSS);
            if (armour_trace_p1() && blocks <= 1024) {   // when each block's (last) item began and ended: the spread between the groups
                if (!wk->d_phase) HIPCHK(hipMalloc((void**)&wk->d_phase, 1024 * 8 * sizeof(long long)));
                HIPCHK(hipMemsetAsync(wk->d_phase, 0, 1024 * 8 * sizeof(long long), h->stream));
                cf.phase_log = wk->d_phase;
            }
            HIPCHK(hipMemsetAsync(wk->d_status, 0, ST_WORDS * sizeof(unsigned), h->stream));
            HIPCHK(hipEventRecord(wk->ev0, h->stream));
            {
                std::lock_guard<std::mutex> lk(g_p1_launch_mu);
                HIPCHK(gr == 64 ? armour_p1_tv_launch_g64(nw_launch, nw, blocks, smem, h->stream, &cf) : armour_p1_tv_launch_g50(nw_launch, nw, blocks, smem, h->stream, &cf));
            }
            HIPCHK(hipEventRecord(wk->ev1, h->stream));
            HIPCHK(hipMemcpyAsync(st, wk->d_status, sizeof(st), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(p1_wait_stream(h->stream));
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, wk->ev0, wk->ev1));
            total_ms += ms;
            h->build_info[0] = ARMOUR_P1_KERNEL_TIME_VECTORISED; h->build_info[1] = nw_launch; h->build_info[2] = cap; h->build_info[3]++;
            if (cf.phase_log) {
                std::vector<long long> ph((size_t)blocks * 8);
                HIPCHK(hipMemcpy(ph.data(), cf.phase_log, ph.size() * sizeof(long long), hipMemcpyDeviceToHost));
                std::vector<std::vector<long long>> byg((size_t)G);
                std::vector<std::vector<long long>> fw((size_t)G), bw((size_t)G);
                for (int bl = 0; bl < blocks; bl++) {
                    const long long* r = &ph[(size_t)bl * 8];
                    if (!(r[5] > r[0] && r[0] > 0 && r[6] < (long long)rnea_items)) continue;
                    const size_t g2 = (size_t)(r[6] % G);
                    byg[g2].push_back(r[5] - r[0]);
                    if (r[3] > r[0] && r[4] > r[3]) { fw[g2].push_back(r[3] - r[0]); bw[g2].push_back(r[4] - r[3]); }
                }
                fprintf(stderr, "[P1 tv phases, cycles of a block's item by group of time steps, median / max]");
                for (int g2 = 0; g2 < G; g2++) {
                    auto& d = byg[(size_t)g2];
                    if (d.empty()) continue;
                    std::sort(d.begin(), d.end());
                    fprintf(stderr, " group %d: %lld / %lld (%zu)", g2, d[d.size() / 2], d.back(), d.size());
                    auto& f = fw[(size_t)g2]; auto& bk = bw[(size_t)g2];
                    if (!f.empty()) { std::sort(f.begin(), f.end()); std::sort(bk.begin(), bk.end()); fprintf(stderr, " [JRS + forward %lld / %lld, backward %lld / %lld]", f[f.size() / 2], f.back(), bk[bk.size() / 2], bk.back()); }
                }
                fprintf(stderr, "\n");
            }
            if (armour_trace_p1()) fprintf(stderr, "[P1 tv] %d groups of <= %d steps%s, sort cap %d: %d blocks of %d wave(s) (%d per CU, %zu B LDS, %d / %d staging rows, %.1f MB arena each, rows of %d), %.2f ms, flags 0x%x, max raw terms %u, max monomials %u\n", groups, LG, fk_items ? " (+ as many forward-kinematics items)" : "", cap, blocks, nw_launch, per_cu, smem, stage_rows, stage_other, TL.total / 1048576.0, gr, ms, st[ST_ERR], st[ST_MAX_RAW], st[ST_MAX_OUT]);
            if (st[ST_ERR] == 0) { built = true; h->p1_tv_shape_hint = si; }
            else if (!(st[ST_ERR] == (unsigned)ERR_RAW_OVERFLOW)) break;  // only the sort buffers can be helped by the next shape
        }
        // (pool_release, at the end of this scope: every launch above has been waited for)
    }
    const int kFirstPassCap = 2048;
    const int* d_items = nullptr;
    int n_items = built ? 0 : B * T;
    // (the second pass costs at least one item's latency, ~4 ms: worth it from about 16 items per CU)
    const int two_pass_env = h->tune(ARMOUR_OPT_P1_TWO_PASS);
    if (!built && two_pass_env && cap_raw > kFirstPassCap && waves_per_cu(kFirstPassCap) > waves_per_cu(cap_raw) && B * T >= 16 * prop.multiProcessorCount) {
        if ((rc = launch(kFirstPassCap, nullptr, B * T, true)) != ARMOUR_OK) return rc;
        if ((rc = other_errors()) != ARMOUR_OK) return rc;
        int nretry = 0;
        HIPCHK(hipMemcpy(&nretry, wk->d_retry, sizeof(int), hipMemcpyDeviceToHost));
        d_items = wk->d_retry + 1;
        n_items = nretry;
    }
    // The per-step kernel is queued and NOT waited for: the half-space kernels, the bounds kernel and the read-back of everything the host keeps go
    // into the stream behind it, and ONE wait ends the build (round 6: the wait for the kernel's status words in between cost the launch latency
    // of five kernels on a device that had gone idle, ~40 us of a lone problem's 0.8 ms).  A launch that reports a capacity flag or a lost helper
    // is rare: everything is queued again behind the next one.
    const size_t n_tr = (size_t)B * n * T, n_lg = (size_t)B * T * J * 18, n_lc = (size_t)B * J * T, n_tc = (size_t)B * n * T;
    const size_t off_lg = n_tr * sizeof(double), off_lc = off_lg + n_lg * sizeof(double), off_tc = off_lc + n_lc * sizeof(int),
                 off_ps = (off_tc + n_tc * sizeof(int) + 7) & ~(size_t)7, off_mg = off_ps + (size_t)B * sizeof(unsigned long long), rb_bytes = off_mg + (size_t)B * sizeof(unsigned long long);
    unsigned char* rb = rb_bytes <= ((size_t)4 << 20) ? reinterpret_cast<unsigned char*>(armour_handle_pinned(h, 8, rb_bytes)) : nullptr;
    for (;;) {
    const bool chain = n_items > 0;
    if (chain && (rc = launch(cap_raw, d_items, n_items, false, true)) != ARMOUR_OK) return rc;
    if (O > 0) {
        const int Q = J * T * O;
        HIPCHK(hipEventRecord(wk->ev2, h->stream));
        const int nbx = (Q + 63) / 64, nbc = (Q + 255) / 256;
        // [B][nbx][4] per-wave masks of the planes kernel | [B][nbc] per-block masks of the class pre-pass | [B] its result | [B] planes a sampled row needs
        const size_t part_words = (size_t)B * nbx * 4, pre_words = (size_t)B * nbc;
        if (part_words + pre_words + 2 * (size_t)B > wk->skip_part_cap) {
            if (wk->d_skip_part) (void)hipFree(wk->d_skip_part);
            wk->d_skip_part = nullptr; wk->skip_part_cap = 0;
            HIPCHK(hipMalloc((void**)&wk->d_skip_part, (part_words + pre_words + 2 * (size_t)B) * sizeof(unsigned long long)));
            wk->skip_part_cap = part_words + pre_words + 2 * (size_t)B;
        }
        unsigned long long* d_pre_part = wk->d_skip_part + part_words;
        unsigned long long* d_pre = d_pre_part + pre_words;
        unsigned long long* d_live = d_pre + B;
        // The table holds what the fused evaluation reads and nothing else (planes_of_group): ARMOUR_OPT_P1_FULL_PLANES = 1 builds
        // the full one.  d = A.c is stored for the small launches that read it; batches of >= 8 problems or >= 32 768 rows recompute it (armour_make_tables).
        const int full_env = h->tune(ARMOUR_OPT_P1_FULL_PLANES);
        const bool lean = !full_env;
        // (round 4, one box, interleaved: configs[4] -- one problem, 90 000 rows, a 60 MB table -- 13.1 us with d recomputed against 15.8 stored;
        //  configs[1] -- 14 000 rows, L2-resident -- 5.05 against 5.02: the launch that streams its table from beyond L2 wants the fewer bytes)
        const int store_d = (B >= 8 || (long long)B * Q >= ARMOUR_RECOMPUTE_D_ROWS) ? 0 : 1;
        if (lean) {
            hipLaunchKernelGGL(armour_p1_plane_class_kernel, dim3(nbc, B), dim3(256), 0, h->stream, T, J, O, wk->d_link_gens, wk->d_obstacles, d_pre_part);
            hipLaunchKernelGGL(armour_p1_plane_sample_kernel, dim3(B), dim3(64 * kPlaneSamples), 0, h->stream, T, J, O, wk->d_link_gens, wk->d_obstacles, d_pre_part, nbc, d_pre, d_live);
            hipLaunchKernelGGL(armour_p1_planes_kernel<true>, dim3(nbx, B), dim3(256), 0, h->stream, B, T, J, O,
                               wk->d_link_gens, wk->d_obstacles, h->d_planes, h->d_planes_ll, h->d_obs_center, wk->d_skip_part, d_pre, d_live, store_d);
        } else {
            hipLaunchKernelGGL(armour_p1_planes_kernel<false>, dim3(nbx, B), dim3(256), 0, h->stream, B, T, J, O,
                               wk->d_link_gens, wk->d_obstacles, h->d_planes, h->d_planes_ll, h->d_obs_center, wk->d_skip_part, nullptr, nullptr, 1);
        }
        hipLaunchKernelGGL(armour_p1_skip_reduce_kernel, dim3(B), dim3(256), 0, h->stream, wk->d_skip_part, nbx * 4, h->d_plane_skip);
        h->ll_shared = 1; h->d_from_center = 1;
        h->planes_lean = lean ? 1 : 0; h->planes_have_d = (!lean || store_d) ? 1 : 0;
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(wk->ev3, h->stream));
    }
    // the constraint bounds of this problem set (torque limits -+ the radii just built, collision and joint-limit constants: RT/NLPclass.cu:87-165),
    // filled on the device behind the build's kernels: the first armour_solve / armour_eval_violations of the set finds them there
    if (h->d_bounds) { const int rcb = armour_bounds_launch(h, wk->d_torque_radius); if (rcb != ARMOUR_OK) return rcb; }
    // What the host keeps of a build -- torque radii, link generators, the monomial counts behind the table statistics, the plane masks -- comes
    // back through ONE page-locked block, queued behind the half-space kernels and waited for once (five blocking copies from pageable memory
    // and a wait of their own before: 0.1 ms of a lone problem's 1.2 ms call).  Large batches keep the plain copies.
    if (rb) {
        HIPCHK(hipMemcpyAsync(rb, wk->d_torque_radius, n_tr * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(rb + off_lg, wk->d_link_gens, n_lg * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(rb + off_lc, h->d_link_count, n_lc * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(rb + off_tc, h->d_tq_count, n_tc * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        if (O > 0) HIPCHK(hipMemcpyAsync(rb + off_ps, h->d_plane_skip, (size_t)B * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(rb + off_mg, wk->d_margin, (size_t)B * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    }
    // (unconditional: the bounds kernel above and the blocking copies below are ordered against h->stream -- a non-blocking stream -- by this
    //  wait alone; with O == 0 and a batch beyond the page-locked block it used to be skipped: ADVICE r5)
    armour_build_stamp("planes+bounds+readback-queued");
    HIPCHK(p1_wait_stream(h->stream));
    armour_build_stamp("planes-waited");
    if (chain) {
        if ((rc = launch_done()) != ARMOUR_OK) return rc;
        if ((st[ST_ERR] & (unsigned)ERR_HELPER) && !h->p1_two_cu_off) {   // a time step on two CUs lost its helper: again on one CU per step, and this handle stays there
            if (armour_trace_p1()) fprintf(stderr, "[P1] a helper block's results did not arrive (flags 0x%x): building again on one CU per time step\n", st[ST_ERR]);
            h->p1_two_cu_off = true;
            h->tuning[ARMOUR_OPT_P1_STEP_TWO_CU - ARMOUR_OPT_FIRST_TUNING] = 0;   // (armour_get_option shows it; armour_set_option switches it on again)
            continue;
        }
        if (st[ST_ERR] & (unsigned)ERR_HELPER_LATE) {   // good tables, built late: some main block waited 1.5 ms for a helper that had not started (a shared device, fewer CUs than reported)
            if (armour_trace_p1()) fprintf(stderr, "[P1] a helper block started too late for its item (flags 0x%x): one CU per time step from the next build on\n", st[ST_ERR]);
            h->p1_two_cu_off = true;
            h->tuning[ARMOUR_OPT_P1_STEP_TWO_CU - ARMOUR_OPT_FIRST_TUNING] = 0;
            st[ST_ERR] &= ~(unsigned)ERR_HELPER_LATE;
        }
        if (st[ST_ERR] & ERR_RAW_OVERFLOW) {
            if (cap_raw < 16384) { cap_raw <<= 1; h->p1_step_cap_hint = cap_raw; continue; }  // retry with larger LDS sort buffers (and start there next time)
            armour_set_error("a PZ product produced more than %d raw terms (raise ArmourLimits.raw_terms)", cap_raw);
            return ARMOUR_ECAPACITY;
        }
        if ((rc = other_errors()) != ARMOUR_OK) return rc;
    }
    if (O > 0) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, wk->ev2, wk->ev3));
        total_ms += ms;
        h->planes_ms = ms;
    }
    break;
    }
    h->build_ms = total_ms;  // device time of every launch of this build, retries included

    h->h_torque_radius.resize(n_tr);
    h->h_link_gens.resize(n_lg);
    if (rb) {
        memcpy(h->h_torque_radius.data(), rb, n_tr * sizeof(double));
        memcpy(h->h_link_gens.data(), rb + off_lg, n_lg * sizeof(double));
        const int* lc = reinterpret_cast<const int*>(rb + off_lc);
        const int* tc = reinterpret_cast<const int*>(rb + off_tc);
        long long sl = 0, st2 = 0;
        int ml = 0, mt = 0;
        for (size_t i = 0; i < n_lc; i++) { sl += lc[i]; if (lc[i] > ml) ml = lc[i]; }
        for (size_t i = 0; i < n_tc; i++) { st2 += tc[i]; if (tc[i] > mt) mt = tc[i]; }
        h->sum_link = sl; h->sum_torque = st2; h->max_link = ml; h->max_torque = mt;
        h->h_plane_skip.assign((size_t)B, 0ull);
        if (O > 0) memcpy(h->h_plane_skip.data(), rb + off_ps, (size_t)B * sizeof(unsigned long long));
        h->stats_fresh = !armour_trace_p1();   // (the trace line of armour_refresh_table_stats wants its own pass)
        armour_margin_from_words(h, reinterpret_cast<const unsigned long long*>(rb + off_mg));
    } else {
        std::vector<unsigned long long> mw((size_t)B);
        HIPCHK(hipMemcpy(mw.data(), wk->d_margin, mw.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        armour_margin_from_words(h, mw.data());
        HIPCHK(hipMemcpy(h->h_torque_radius.data(), wk->d_torque_radius, n_tr * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(h->h_link_gens.data(), wk->d_link_gens, n_lg * sizeof(double), hipMemcpyDeviceToHost));
    }
    return ARMOUR_OK;
}

// The FULL half-space table of the current problem set -- every plane, all five components, what RT/CollisionChecking.cu:215-227 holds --
// into `d_full` ([B][5][36][Q] doubles, device): the same kernel code as the build's, with nothing left out.  For armour_get_hyperplanes
// (parity tests, diagnostics); the product path reads the lean table.
int armour_p1_full_planes(ArmourPlanner* h, double* d_full) {
    P1Work* wk = (P1Work*)h->p1;
    if (!wk || !wk->d_link_gens || !wk->d_obstacles || h->O <= 0) { armour_set_error("no reach sets on the device to rebuild the half-space table from"); return ARMOUR_ESTATE; }
    const int Q = h->J * h->T * h->O, nbx = (Q + 63) / 64;
    hipLaunchKernelGGL(armour_p1_planes_kernel<false>, dim3(nbx, h->B), dim3(256), 0, h->stream, h->B, h->T, h->J, h->O,
                       wk->d_link_gens, wk->d_obstacles, d_full, nullptr, nullptr, nullptr, nullptr, nullptr, 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return ARMOUR_OK;
}
#endif  // P1_TV_VARIANT
