// Time-vectorised reach-set build: one wavefront per (problem, group of <= 64 time steps), lane = time step, on the
// arithmetic of pz_tv.h.  Included by p1_reach.hip (it shares P1Cfg, the JRS scalars and the host-side launch code).
// The operator sequence is that of run_rnea / fk_step above -- the reference's RT/Dynamics.cu:69-181, RT/armour_main.cu:96-216
// -- played by one wave in the order a 1-wave block of the per-step kernel plays it.

namespace tvchain {

using tv::TPZ;
using tv::TView;
using tv::TSeg;
using tv::TW;

constexpr int kTvNS = 12, kTvNM = 2;   // 1x1 and 3x3 work slots; 3x1 work slots: 32 for one wave, 60 (split by role) for three
// 3x1 pool of a three-wave block (run_rnea_free: the waves run ahead of each other, so a few joints' states are alive at a time)
constexpr int kTvPartFirst[kRoles] = {0, 12, 34}, kTvPartCount[kRoles] = {12, 22, 28};
// ... and of a four-wave block (the forward kinematics on a wave of its own)
constexpr int kTvPart4First[4] = {0, 8, 23, 43}, kTvPart4Count[4] = {8, 15, 20, 20};   // (the forward-kinematics wave also owns w_aux and R_t w_aux: p1_free.inc.h, aux3)

// (one spare key and one spare row block beyond `cap`)
// (gr: doubles per row -- tv::GR in the kernel; the host sizes the arena for the unit it is about to launch)
__host__ __device__ inline size_t tv_slot_bytes(int cap, int sz, int gr = tv::GR) { return align64((size_t)(cap + 1) * sizeof(pzkey_t)) + ((size_t)(cap + 1) * sz + 4 * (size_t)sz) * gr * sizeof(double); }

struct TLayout {
    int nJM, nJV, nJS, nV, nroles, npools;   // nroles: sets of JRS scratch slots (one per wave); npools: parts of the 3x1 pool (1, 3 or 4)
    size_t offV, offS, offM, offJM, offJV, offJS, offH, total;
    int idV, idS, idM, idJM, idJV, idJS;
};
// scratch slot of a dedicated helper wave (pz_tv.h serve_loop): a 3x1 work slot with 16 partial-sum rows where the header rows are
__host__ __device__ inline size_t tv_helper_slot_bytes(int cap, int gr = tv::GR) { return tv_slot_bytes(cap, 3, gr) + (size_t)4 * gr * sizeof(double); }
// nwaves: the waves that play roles (1, 3 or 4); nhelp: dedicated helper waves behind them (0, or 4 in an eight-wave block)
__host__ __device__ inline TLayout make_tlayout(int J, int n, int capW, int nwaves, int nhelp = 0, int gr = tv::GR) {
    TLayout L;
    const int nroles = nwaves;   // (a four-wave block builds the JRS on all four waves)
    L.nroles = nroles; L.npools = nwaves;
    L.nV = nwaves == 1 ? kNVOneWave : nwaves == kRoles ? kTvPartFirst[kRoles - 1] + kTvPartCount[kRoles - 1] : kTvPart4First[3] + kTvPart4Count[3];
    L.nJM = (J + 1) + J + 3 * nroles + J;  // R[0..J], R_t[0..J-1], per role: three scratch slots (raw rot, simplified rot, rpy: unused since the JRS is built in closed form, kept so that slot numbers stay what the profiles and notes refer to); inertia
    L.nJV = (J + 1) + J;                   // trans P_i, link boxes
    L.nJS = 3 * n + J + 4 * nroles;        // qd, qda, qdda; mass; per role: 4 raw temps
    L.offV = 0;
    L.offS = L.offV + (size_t)L.nV * tv_slot_bytes(capW, 3, gr);
    L.offM = L.offS + (size_t)kTvNS * tv_slot_bytes(capW, 1, gr);
    L.offJM = L.offM + (size_t)kTvNM * tv_slot_bytes(capW, 9, gr);
    L.offJV = L.offJM + (size_t)L.nJM * tv_slot_bytes(kCapSmall, 9, gr);
    L.offJS = L.offJV + (size_t)L.nJV * tv_slot_bytes(kCapSmall, 3, gr);
    L.offH = align64(L.offJS + (size_t)L.nJS * tv_slot_bytes(kCapSmall, 1, gr));
    L.total = align64(L.offH + (size_t)nhelp * tv_helper_slot_bytes(capW, gr));
    L.idV = 0; L.idS = L.idV + L.nV; L.idM = L.idS + kTvNS; L.idJM = L.idM + kTvNM; L.idJV = L.idJM + L.nJM; L.idJS = L.idJV + L.nJV;
    return L;
}

__device__ inline TPZ mk_tslot(GLB_AS unsigned char* base, size_t off, int index, int cap, int sz, int id0) {
    GLB_AS unsigned char* p = base + off + (size_t)index * tv_slot_bytes(cap, sz);
    TPZ z;
    z.keys = (GLB_AS pzkey_t*)p;
    z.hdr = (GLB_AS double*)(p + align64((size_t)(cap + 1) * sizeof(pzkey_t)));
    z.coef = z.hdr + (size_t)4 * sz * tv::GR;
    z.sz = sz; z.cap = cap; z.id = id0 + index;
    return z;
}

// The same interface as Chain above (run_rnea / fk_step are templates on it): roles, mailbox, slot pools, composite operators.
struct TChain {
    typedef TPZ PZT;
    static constexpr bool kWalkHelpers = true;   // pz_tv.h "One walk on two waves"; used by run_rnea_free in four-wave blocks
    static constexpr bool kPairs = false;        // (the per-step chain's two-waves-per-operator backward pass; this chain shares its walks instead)
    static constexpr bool kTwoCu = false;        // (a time step on two CUs: the per-step chain's, p1_free.inc.h)
    static constexpr bool kFusedCross = true;    // run_rnea_free: (a + cross(w, b)) + c with the constant cross product taken inside the sum's walk (sum3x)
    long long* phase_log = nullptr;              // (ARMOUR_P1_TRACE) this block's row of P1Cfg::phase_log
    __device__ void phase(int k) const { if (phase_log != nullptr && wid == 0 && (threadIdx.x & 63) == 0) phase_log[k] = clock64(); }
    TW w;
    const P1Cfg* cf;
    GLB_AS unsigned char* arena;
    TLayout L;
    unsigned long long freeV;
    unsigned freeS;
    int n, J, capW;
    int wid, nw;
    LDS_AS int* mb;
    int role = 0;
    __device__ bool is(int r) const { return nw == 1 || wid == r; }
    __device__ pzw::Wave& wave() { return w.w; }
    __device__ const pzw::Wave& wave() const { return w.w; }
#ifdef TV_PROFILE
    __device__ long long prof_clock() const { return clock64(); }
    __device__ void prof_waited(long long t0) { w.c_wait += clock64() - t0; }
    __device__ void prof_forward_done() { w.c_fwd = clock64(); w.c_wait_fwd = w.c_wait; }
    __device__ void prof_signal(int, int) {}
#else
    __device__ long long prof_clock() const { return 0; }
    __device__ void prof_waited(long long) {}
    __device__ void prof_forward_done() {}
    __device__ void prof_signal(int, int) {}
#endif
    __device__ void bar() {   // (a dedicated helper joins every block barrier of its primary)
        if (w.hded) tv::hj_post_ctl(w, tv::HK_BAR);
        __syncthreads();
    }
    __device__ void post(int slot, const TPZ& p) const { if (w.w.lane == 0) mb[slot] = p.id - L.idV; }
    __device__ TPZ take(int slot) const { return V(mb[slot]); }
    __device__ TPZ V(int i) const { return mk_tslot(arena, L.offV, i, capW, 3, L.idV); }
    __device__ TPZ S(int i) const { return mk_tslot(arena, L.offS, i, capW, 1, L.idS); }
    __device__ TPZ M(int i) const { return mk_tslot(arena, L.offM, i, capW, 9, L.idM); }
    __device__ TPZ JM(int i) const { return mk_tslot(arena, L.offJM, i, kCapSmall, 9, L.idJM); }
    __device__ TPZ JV(int i) const { return mk_tslot(arena, L.offJV, i, kCapSmall, 3, L.idJV); }
    __device__ TPZ JS(int i) const { return mk_tslot(arena, L.offJS, i, kCapSmall, 1, L.idJS); }
    __device__ TPZ H(int i) const {   // helper wave i's scratch slot (no entry in the count table: only its keys, rows and partial-sum rows are used)
        GLB_AS unsigned char* p = arena + L.offH + (size_t)i * tv_helper_slot_bytes(capW);
        TPZ z;
        z.keys = (GLB_AS pzkey_t*)p;
        z.hdr = (GLB_AS double*)(p + align64((size_t)(capW + 1) * sizeof(pzkey_t)));
        z.coef = z.hdr + (size_t)16 * tv::GR;
        z.sz = 3; z.cap = capW; z.id = 0;
        return z;
    }
    __device__ TPZ R(int i) const { return JM(i); }
    __device__ TPZ Rt(int i) const { return JM(J + 1 + i); }
    __device__ int scratch(int r) const { return L.nroles == 1 ? 0 : r; }
    __device__ TPZ rotRaw(int r) const { return JM(2 * J + 1 + 3 * scratch(r)); }
    __device__ TPZ rotS(int r) const { return JM(2 * J + 2 + 3 * scratch(r)); }
    __device__ TPZ rpy(int r) const { return JM(2 * J + 3 + 3 * scratch(r)); }
    __device__ TPZ inertia(int i) const { return JM(2 * J + 1 + 3 * L.nroles + i); }
    __device__ TPZ Ptr(int i) const { return JV(i); }
    __device__ TPZ linkbox(int i) const { return JV(J + 1 + i); }
    __device__ TPZ qd(int i) const { return JS(i); }
    __device__ TPZ qda(int i) const { return JS(n + i); }
    __device__ TPZ qdda(int i) const { return JS(2 * n + i); }
    __device__ TPZ mass(int i) const { return JS(3 * n + i); }
    __device__ TPZ rawS(int r, int i) const { return JS(3 * n + J + 4 * scratch(r) + i); }

    __device__ TPZ allocV() {
        const unsigned long long part = L.npools == 1 ? ~0ull : L.npools == kRoles ? ((1ull << kTvPartCount[role]) - 1ull) << kTvPartFirst[role]
                                                                                   : ((1ull << kTvPart4Count[role]) - 1ull) << kTvPart4First[role];
        const int i = __ffsll((long long)(freeV & part)) - 1;
        if (i < 0) { pzw::flag(w.w, pzw::ERR_SLOT_OVERFLOW); return V(L.npools == 1 ? 0 : L.npools == kRoles ? kTvPartFirst[role] : kTvPart4First[role]); }
        freeV &= ~(1ull << i);
        return V(i);
    }
    __device__ void freeVs(const TPZ& p) { freeV |= 1ull << (p.id - L.idV); }
    __device__ TPZ allocS() {
        const int i = __ffs(freeS) - 1;
        if (i < 0) { pzw::flag(w.w, pzw::ERR_SLOT_OVERFLOW); return S(0); }
        freeS &= ~(1u << i);
        return S(i);
    }

    __device__ TPZ add(const TPZ& a, const TPZ& b, double sb = 1.0) {
        TPZ o = allocV();
        TSeg s[2] = {{tv::view(w, a), 1.0, -1}, {tv::view(w, b), sb, -1}};
        tv::lincomb<3, 2, false>(w, o, s);
        return o;
    }
    __device__ TPZ addOneDim(const TPZ& p, const TPZ& a, int r) {
        TPZ o = allocV();
        TSeg s[2] = {{tv::view(w, p), 1.0, -1}, {tv::view(w, a), 1.0, r}};
        tv::lincomb<3, 2, false>(w, o, s);
        return o;
    }
    // addOneDimPZ(PZsparse(0,0,0), a, r) in closed form: a 1x1 PZ (two monomials for a joint velocity) embedded in entry r of a 3-vector.  Through
    // the operators -- a zero constant, then lincomb<3,2> over the two rows -- this was two operators' worth of round trips on the angular wave's
    // chain, once per joint.  The arithmetic of that lincomb per lane, in its order: centre 0 + 1*c, radius (0 + 1*ind) + pruned, a row kept while
    // any lane keeps it (the norm of (0, x, 0) is |x|), asum over the rows kept.
    __device__ TPZ embedOneDim(const TPZ& a, int r) {
        TPZ o = allocV();
        const int lane = w.w.lane, rl = w.rl, n = tv::uni(w.w.cnt[a.id]);
        const double thr_sq = w.w.thr_sq;
        const TView av = tv::view(w, a);
        const double cen = 0.0 + 1.0 * tv::ld_hdr(av, tv::H_CEN, 0, rl);
        const double ind = 0.0 + tv::ld_hdr(av, tv::H_IND, 0, rl) * 1.0, ind2 = 0.0 + tv::ld_hdr(av, tv::H_IND2, 0, rl) * 1.0;
        double ra = 0.0, as = 0.0;
        int pos = 0;
        for (int m = 0; m < n; m++) {
            const double x = 1.0 * a.coef[(size_t)m * tv::GR + rl];
            const bool small = x * x <= thr_sq;
            tv::mtrk(w.w.mabs, thr_sq, x * x);   // (prune margin: pz_tv.h)
            ra += small ? fabs(x) : 0.0;
            const bool keep = !small && w.active;
            if (__ballot(keep) != 0ull) {
                const double v = keep ? x : 0.0;
                if (lane == 0) o.keys[pos] = a.keys[m];
#pragma unroll
                for (int e = 0; e < 3; e++) o.coef[((size_t)pos * 3 + e) * tv::GR + rl] = (e == r) ? v : 0.0;
                as += fabs(v);
                pos++;
            }
        }
#pragma unroll
        for (int e = 0; e < 3; e++) {
            const bool me = e == r;
            tv::st_hdr(o, tv::H_CEN, e, rl, me ? cen : 0.0);
            tv::st_hdr(o, tv::H_IND, e, rl, me ? ind + ra : 0.0);
            tv::st_hdr(o, tv::H_IND2, e, rl, me ? ind2 + ra : 0.0);
            tv::st_hdr(o, tv::H_ASUM, e, rl, me ? as : 0.0);
        }
        if (lane == 0) w.w.cnt[o.id] = pos;
        WSYNC();
        return o;
    }
    __device__ TPZ sum3(const TPZ& a, const TPZ& b, const TPZ& c3, int comp_c = -1) {
        TPZ o = allocV();
        TSeg s[3] = {{tv::view(w, a), 1.0, -1}, {tv::view(w, b), 1.0, -1}, {tv::view(w, c3), 1.0, comp_c}};
        tv::lincomb<3, 3, true>(w, o, s);
        return o;
    }
    // (a + cross(w, bvec)) + c3: the sum3 of a, crossPzMat(w, bvec), c3 without the cross product as a PZ of its own (pz_tv.h LinCtx, XK)
    __device__ TPZ sum3x(const TPZ& a, const TPZ& wv, const double* bvec, const TPZ& c3) {
        TPZ o = allocV();
        TSeg s[3] = {{tv::view(w, a), 1.0, -1}, {tv::view(w, wv), 1.0, -1}, {tv::view(w, c3), 1.0, -1}};
        tv::lincomb<3, 3, true, 1>(w, o, s, bvec);
        return o;
    }
    __device__ TPZ sum4(const TPZ& a, const TPZ& b, const TPZ& c3, const TPZ& d) {
        TPZ o = allocV();
        TSeg s[4] = {{tv::view(w, a), 1.0, -1}, {tv::view(w, b), 1.0, -1}, {tv::view(w, c3), 1.0, -1}, {tv::view(w, d), 1.0, -1}};
        tv::lincomb<3, 4, true>(w, o, s);
        return o;
    }
    __device__ TPZ comb3(const TView& a, double sa, const TView& b, double sb, const TView& c3, double sc) {
        TPZ o = allocS();
        TSeg s[3] = {{a, sa, -1}, {b, sb, -1}, {c3, sc, -1}};
        tv::lincomb<1, 3, true>(w, o, s);
        return o;
    }
    __device__ TPZ crossPzMat(const TPZ& a, const double* b) {  // a x b
        TPZ o = allocV();
        const double sA[3] = {b[2], b[0], b[1]}, sB[3] = {-b[1], -b[2], -b[0]};
        const int cA[3] = {1, 2, 0}, cB[3] = {2, 0, 1};
        tv::cross_const(w, o, tv::view(w, a), sA, cA, sB, cB);
        return o;
    }
    __device__ TPZ crossMatPz(const double* a, const TPZ& b) {  // a x b
        TPZ o = allocV();
        const double sA[3] = {a[1], a[2], a[0]}, sB[3] = {-a[2], -a[0], -a[1]};
        const int cA[3] = {2, 0, 1}, cB[3] = {1, 2, 0};
        tv::cross_const(w, o, tv::view(w, b), sA, cA, sB, cB);
        return o;
    }
    __device__ TPZ crossPzPz(const TPZ& a, const TPZ& b) {
        TPZ o = allocV();
        tv::cross_pzpz(w, o, tv::view(w, a), tv::view(w, b));
        return o;
    }
    __device__ TPZ mulMV(const TPZ& A, const TPZ& v) {
        TPZ o = allocV();
        tv::mul<3, 3, 3, 1>(w, o, tv::view(w, A), tv::view(w, v));
        return o;
    }
    __device__ TPZ mulSV(const TPZ& s, const TPZ& v) {
        TPZ o = allocV();
        tv::mul<1, 1, 3, 1>(w, o, tv::view(w, s), tv::view(w, v));
        return o;
    }
};

// ---- the small PZs of the JRS in closed form (p1_reach.hip has the per-step twin and the reasons) ------------------------------------------
// Every lane evaluates the operators' arithmetic for ITS time step in registers, in their order -- lincomb<SZ,1>'s simplify() of the raw terms,
// the constant-left product's, transpose33's copies, the stack's -- and writes its rows; which rows exist is the wave's vote, as in the walks
// (a key stays while any lane keeps it; a lane that pruned it stores 0 and has |.| in its radius).  Same tables bit for bit as through the
// operators (launch digests, tools/ab.py --reps), without their round trips through the arena: a dozen per joint.
template <int SZ>
__device__ inline void jrs_put_row(const TPZ& out, int pos, int lane, int rl, pzkey_t key, const double* v) {   // (rl: the lane's place in a row)
    if (lane == 0) out.keys[pos] = key;
#pragma unroll
    for (int e = 0; e < SZ; e++) out.coef[((size_t)pos * SZ + e) * tv::GR + rl] = v[e];
}
__device__ inline bool jrs_small9(pzw::Wave& w, const double* m) {   // the verdict on a 3x3 term, tracked for the prune margin (pz_tv.h mtrk)
    double q = 0.0;
#pragma unroll
    for (int e = 0; e < 9; e++) q += m[e] * m[e];
    tv::mtrk(w.mabs, w.thr_sq, q);
    return q <= w.thr_sq;
}
// rotation about the joint axis from the cos / sin polynomials (four raw terms {k: cos_k, e_c: cos_e, k: sin_k, e_s: sin_e}, simplify()),
// R_i = R_rpy * it, and (with_rt) the transpose
__device__ inline void jrs_rotation_direct_tv(TChain& c, int i, const JrsScalars& js, const double* rp, bool with_rt) {
    const P1Cfg& cf = *c.cf;
    const int lane = c.w.w.lane, rl = c.w.rl, n = c.n, ax = cf.rb.axes[i];
    const double thr_sq = c.w.w.thr_sq;
    const bool active = c.w.active;
    const pzkey_t kk = pzkey_bit(2 * i), kc = pzkey_bit(5 * n + 2 * i), ks = pzkey_bit(7 * n + 2 * i);   // kk < kc < ks: the sorted order
    double cen[9], m0[9], m1[9], m2[9], t2[9];
    make_rotation(cen, js.cos_c, js.sin_c, ax, false);
    make_rotation(m0, js.cos_k, 0.0, ax, true);
    make_rotation(m1, js.cos_e, 0.0, ax, true);
    make_rotation(t2, 0.0, js.sin_k, ax, true);
    make_rotation(m2, 0.0, js.sin_e, ax, true);
#pragma unroll
    for (int e = 0; e < 9; e++) { cen[e] = 1.0 * cen[e]; m0[e] = 1.0 * m0[e] + 1.0 * t2[e]; m1[e] = 1.0 * m1[e]; m2[e] = 1.0 * m2[e]; }
    // simplify(), per lane; the pruned amounts join the lane's radius in key order
    const bool d0 = jrs_small9(c.w.w, m0), d1 = jrs_small9(c.w.w, m1), d2 = jrs_small9(c.w.w, m2);
    const bool k0 = !d0 && active, k1 = !d1 && active, k2 = !d2 && active;
    const bool e0 = __ballot(k0) != 0ull, e1 = __ballot(k1) != 0ull, e2 = __ballot(k2) != 0ull;   // rows of rot that exist (wave-uniform)
    double rind[9];
#pragma unroll
    for (int e = 0; e < 9; e++) {
        double ra = 0.0;
        ra += d0 ? fabs(m0[e]) : 0.0; ra += d1 ? fabs(m1[e]) : 0.0; ra += d2 ? fabs(m2[e]) : 0.0;
        rind[e] = 0.0 + ra;
        m0[e] = k0 ? m0[e] : 0.0; m1[e] = k1 ? m1[e] : 0.0; m2[e] = k2 ? m2[e] : 0.0;   // what the rows hold
    }
    // R = R_rpy * rot: the constant-left product over rot's rows
    typedef pzw::MulShape<3, 3, 3, 3> SH;
    double arp[9], base[9], Rcen[9], a0[9], a1[9], a2[9];
#pragma unroll
    for (int e = 0; e < 9; e++) arp[e] = fabs(rp[e]) + 0.0;
    SH::mul(arp, rind, base);           // (|c_a| + asum_a) * indep_b; the two terms with indep_a = 0 are exact zeros
    SH::mul(rp, cen, Rcen);
    SH::mul(rp, m0, a0);
    SH::mul(rp, m1, a1);
    SH::mul(rp, m2, a2);
    const bool s0 = jrs_small9(c.w.w, a0), s1 = jrs_small9(c.w.w, a1), s2 = jrs_small9(c.w.w, a2);   // (a row rot does not hold is zeros here: a zero norm is not tracked)
    const bool K0 = e0 && !s0 && active, K1 = e1 && !s1 && active, K2 = e2 && !s2 && active;
    const bool E0 = __ballot(K0) != 0ull, E1 = __ballot(K1) != 0ull, E2 = __ballot(K2) != 0ull;
    const TPZ R = c.R(i), Rt = c.Rt(i);
    double Rind[9], asum[9];
#pragma unroll
    for (int e = 0; e < 9; e++) {
        double rad = 0.0, as = 0.0;
        rad += (e0 && !K0) ? fabs(a0[e]) : 0.0; rad += (e1 && !K1) ? fabs(a1[e]) : 0.0; rad += (e2 && !K2) ? fabs(a2[e]) : 0.0;
        a0[e] = K0 ? a0[e] : 0.0; a1[e] = K1 ? a1[e] : 0.0; a2[e] = K2 ? a2[e] : 0.0;
        as += E0 ? fabs(a0[e]) : 0.0; as += E1 ? fabs(a1[e]) : 0.0; as += E2 ? fabs(a2[e]) : 0.0;
        Rind[e] = (0.0 + (base[e] + 0.0)) + rad;
        asum[e] = as;
    }
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int cc = 0; cc < 3; cc++) {
            const int e = r * 3 + cc, et = cc * 3 + r;
            tv::st_hdr(R, tv::H_CEN, e, rl, Rcen[e]); tv::st_hdr(R, tv::H_IND, e, rl, Rind[e]); tv::st_hdr(R, tv::H_IND2, e, rl, Rind[e]); tv::st_hdr(R, tv::H_ASUM, e, rl, asum[e]);
            if (with_rt) { tv::st_hdr(Rt, tv::H_CEN, et, rl, Rcen[e]); tv::st_hdr(Rt, tv::H_IND, et, rl, Rind[e]); tv::st_hdr(Rt, tv::H_IND2, et, rl, Rind[e]); tv::st_hdr(Rt, tv::H_ASUM, et, rl, asum[e]); }
        }
    int pos = 0;
    double tr[9];
    if (E0) {
        jrs_put_row<9>(R, pos, lane, rl, kk, a0);
        if (with_rt) { for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) tr[cc * 3 + r] = a0[r * 3 + cc]; jrs_put_row<9>(Rt, pos, lane, rl, kk, tr); }
        pos++;
    }
    if (E1) {
        jrs_put_row<9>(R, pos, lane, rl, kc, a1);
        if (with_rt) { for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) tr[cc * 3 + r] = a1[r * 3 + cc]; jrs_put_row<9>(Rt, pos, lane, rl, kc, tr); }
        pos++;
    }
    if (E2) {
        jrs_put_row<9>(R, pos, lane, rl, ks, a2);
        if (with_rt) { for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) tr[cc * 3 + r] = a2[r * 3 + cc]; jrs_put_row<9>(Rt, pos, lane, rl, ks, tr); }
        pos++;
    }
    if (lane == 0) { c.w.w.cnt[R.id] = pos; if (with_rt) c.w.w.cnt[Rt.id] = pos; }
}
// PZsparse(centre, {k: a, e: b}) of a velocity / acceleration polynomial, simplify()
__device__ inline void jrs_scalar_direct_tv(TChain& c, const TPZ& out, double cen, pzkey_t key0, double a, pzkey_t key1, double b) {
    const int lane = c.w.w.lane, rl = c.w.rl;
    const double thr = c.w.w.thr;
    const bool active = c.w.active;
    double va = 1.0 * a, vb = 1.0 * b;
    const bool da = fabs(va) <= thr, db = fabs(vb) <= thr;
    tv::mtrk(c.w.w.mabs, c.w.w.thr_sq, va * va); tv::mtrk(c.w.w.mabs, c.w.w.thr_sq, vb * vb);
    const bool ka = !da && active, kb = !db && active;
    const bool ea = __ballot(ka) != 0ull, eb = __ballot(kb) != 0ull;
    double ra = 0.0;
    ra += da ? fabs(va) : 0.0; ra += db ? fabs(vb) : 0.0;
    va = ka ? va : 0.0; vb = kb ? vb : 0.0;
    double as = 0.0;
    as += ea ? fabs(va) : 0.0; as += eb ? fabs(vb) : 0.0;
    tv::st_hdr(out, tv::H_CEN, 0, rl, 1.0 * cen); tv::st_hdr(out, tv::H_IND, 0, rl, 0.0 + ra); tv::st_hdr(out, tv::H_IND2, 0, rl, 0.0 + ra); tv::st_hdr(out, tv::H_ASUM, 0, rl, as);
    int pos = 0;
    if (ea) { jrs_put_row<1>(out, pos, lane, rl, key0, &va); pos++; }
    if (eb) { jrs_put_row<1>(out, pos, lane, rl, key1, &vb); pos++; }
    if (lane == 0) c.w.w.cnt[out.id] = pos;
}
// link box: three 1x1 PZs {centre_j, one generator on key field (j + 2) n}, each simplify()d, then stack()ed and simplify()d
__device__ inline void jrs_linkbox_direct_tv(TChain& c, int i) {
    const P1Cfg& cf = *c.cf;
    const int lane = c.w.w.lane, rl = c.w.rl, n = c.n;
    const double thr = c.w.w.thr, thr_sq = c.w.w.thr_sq;
    const bool active = c.w.active;
    const TPZ out = c.linkbox(i);
    double x[3], ind1[3];
    bool ex1[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {   // the 1x1 PZs
        const double g = 1.0 * cf.rb.link_zonotope_generators[3 * i + j];
        const bool d = fabs(g) <= thr, k = !d && active;
        tv::mtrk(c.w.w.mabs, thr_sq, g * g);
        ex1[j] = __ballot(k) != 0ull;
        ind1[j] = 0.0 + (0.0 + (d ? fabs(g) : 0.0));
        x[j] = k ? g : 0.0;
    }
    // stack(): term j embedded in entry j; the runs come in the order of j
    double ra[3] = {0.0, 0.0, 0.0}, as[3] = {0.0, 0.0, 0.0};
    bool k2[3], ex2[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const double v = 1.0 * x[j];
        const bool d = ex1[j] && (0.0 + v * v + 0.0 * 0.0 <= thr_sq);   // (the norm of (0, v, 0): zeros add nothing)
        tv::mtrk(c.w.w.mabs, thr_sq, 0.0 + v * v + 0.0 * 0.0);
        ra[j] += d ? fabs(v) : 0.0;
        k2[j] = ex1[j] && !d && active;
        ex2[j] = __ballot(k2[j]) != 0ull;
        x[j] = k2[j] ? v : 0.0;
        as[j] += ex2[j] ? fabs(x[j]) : 0.0;
    }
#pragma unroll
    for (int e = 0; e < 3; e++) {
        const double i0 = e == 0 ? ind1[0] * 1.0 : 0.0, i1 = e == 1 ? ind1[1] * 1.0 : 0.0, i2 = e == 2 ? ind1[2] * 1.0 : 0.0;
        const double r = ((i0 + i1) + i2) + ra[e];
        tv::st_hdr(out, tv::H_CEN, e, rl, 0.0 + 1.0 * (1.0 * cf.rb.link_zonotope_center[3 * i + e]));
        tv::st_hdr(out, tv::H_IND, e, rl, r); tv::st_hdr(out, tv::H_IND2, e, rl, r); tv::st_hdr(out, tv::H_ASUM, e, rl, as[e]);
    }
    int pos = 0;
#pragma unroll
    for (int j = 0; j < 3; j++)
        if (ex2[j]) {
            const double v[3] = {j == 0 ? x[0] : 0.0, j == 1 ? x[1] : 0.0, j == 2 ? x[2] : 0.0};
            jrs_put_row<3>(out, pos, lane, rl, pzkey_bit((j + 2) * n), v);
            pos++;
        }
    if (lane == 0) c.w.w.cnt[out.id] = pos;
}

// JRS of this lane's time interval + the constant PZs (see build_jrs above for the per-step form and the citations).
// Joint i is built by wave i % (number of waves) with that wave's scratch slots; the caller follows with a block barrier.
__device__ TV_NOINLINE void build_jrs_tv(TChain& c, int b, int t_lane, bool kin_only) {
    PZ_KEEP_RETURN_ADDRESS();
    const P1Cfg& cf = *c.cf;
    const int n = c.n, J = c.J;
    const double* bz = cf.bez + (size_t)b * 3 * n;
    for (int i = 0; i < J; i++) {
        const int role = i % c.L.nroles;
        if (!c.is(role)) continue;
        double rp[9];
        rpy_matrix(cf.rb.rots[3 * i], cf.rb.rots[3 * i + 1], cf.rb.rots[3 * i + 2], rp);
        if (i < n && cf.rb.axes[i] != 0) {
            const JrsScalars js = cf.mode == ARMOUR_MODE_ARMTD ? armtd_jrs_scalars(cf, bz[i], b, i, t_lane)   // (CMP/Trajectory.cu:29-61: offline tables)
                                                               : jrs_scalars(cf, bz[i], bz[n + i], bz[2 * n + i], i, t_lane);
            const pzkey_t kk = pzkey_bit(2 * i);
            jrs_rotation_direct_tv(c, i, js, rp, !kin_only);
            if (!kin_only) {
                jrs_scalar_direct_tv(c, c.qd(i), js.qd_c, kk, js.qd_k, pzkey_bit(2 * n + i), js.qd_e);
                jrs_scalar_direct_tv(c, c.qda(i), js.qd_c, kk, js.qd_k, pzkey_bit(3 * n + i), js.qda_e);
                jrs_scalar_direct_tv(c, c.qdda(i), js.qdd_c, kk, js.qdd_k, pzkey_bit(4 * n + i), js.qdd_e);
            }
        } else {
            tv::set_const(c.w, c.R(i), rp, nullptr);
            if (!kin_only) tv::transpose33(c.w, c.Rt(i), c.R(i));
        }
        tv::set_const(c.w, c.Ptr(i), &cf.rb.trans[3 * i], nullptr);
        if (!kin_only) {   // mass and inertia: constants with a second radius (no local arrays behind set_const's pointers: they would live in scratch memory)
            const int lane = c.w.w.lane, rl = c.w.rl;
            const TPZ pm = c.mass(i), pi = c.inertia(i);
            tv::st_hdr(pm, tv::H_CEN, 0, rl, cf.rb.mass[i]); tv::st_hdr(pm, tv::H_IND, 0, rl, 0.0);
            tv::st_hdr(pm, tv::H_IND2, 0, rl, armour_mass_uncertainty(&cf.rb, i) * fabs(cf.rb.mass[i])); tv::st_hdr(pm, tv::H_ASUM, 0, rl, 0.0);
            const double unc = armour_inertia_uncertainty(&cf.rb, i);
#pragma unroll
            for (int e = 0; e < 9; e++) {
                const double v = cf.rb.inertia[9 * i + e];
                tv::st_hdr(pi, tv::H_CEN, e, rl, v); tv::st_hdr(pi, tv::H_IND, e, rl, 0.0); tv::st_hdr(pi, tv::H_IND2, e, rl, unc * fabs(v)); tv::st_hdr(pi, tv::H_ASUM, e, rl, 0.0);
            }
            if (lane == 0) { c.w.w.cnt[pm.id] = 0; c.w.w.cnt[pi.id] = 0; }
        }
        jrs_linkbox_direct_tv(c, i);
    }
    WSYNC();   // this wave's rows and counts are written (the caller's block barrier does the same for the block)
    if (c.is(0)) {
        double id[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        tv::set_const(c.w, c.R(J), id, nullptr);
        tv::set_const(c.w, c.Ptr(J), &cf.rb.trans[3 * J], nullptr);
    }
}

// reduce_link_PZ (RT/PZsparse.cu:370-402) + the final link table entry, per lane.  Which class a monomial belongs to depends
// on its key alone (wave-uniform); whether a lane HAS the monomial is whether its coefficient vector is non-zero.
// (Found by fk_step through its chain argument; `t_lane` is this lane's time step.)
__device__ TV_NOINLINE void emit_link(TChain& c, const TPZ& p, int b, int l, int t_lane) {
    PZ_KEEP_RETURN_ADDRESS();
    TVP_FN(c.w, 7)
    const P1Cfg& cf = *c.cf;
    TW& t = c.w;
    const int lane = t.w.lane;
    const int n = c.n, cnt = tv::uni(t.w.cnt[p.id]);
    const pzkey_t kmax = pzkey_bit(2 * n), lmax = pzkey_bit(5 * n), kmask = kmax - 1;
    const size_t idx = ((size_t)b * c.J + l) * cf.T + t_lane;
    double* gens = cf.link_gens + (((size_t)b * cf.T + t_lane) * c.J + l) * 18;
    double g[18];
    for (int i = 0; i < 18; i++) g[i] = 0.0;
    double ra[3] = {0, 0, 0};
    int nk = 0, ng = 0;
    const GLB_AS double* pc = tv::uni_ptr(p.coef) + t.rl;
    for (int m0 = 0; m0 < cnt; m0 += 64) {
        const pzkey_t key_v = m0 + lane < cnt ? p.keys[m0 + lane] : 0ull;
        const int nn = min(64, cnt - m0);
        constexpr int kU = 8;   // monomials whose three rows are in flight together
        for (int q0 = 0; q0 < nn; q0 += kU) {
            double xs[kU][3];
#pragma unroll
            for (int u = 0; u < kU; u++) {
                const GLB_AS double* pr = pc + (size_t)(m0 + min(q0 + u, nn - 1)) * 3 * tv::GR;
                xs[u][0] = pr[0]; xs[u][1] = pr[tv::GR]; xs[u][2] = pr[2 * tv::GR];
            }
#pragma unroll
            for (int u = 0; u < kU; u++) {
                if (q0 + u >= nn) break;
                const pzkey_t key = pzkey_readlane(key_v, q0 + u);
                const double x = xs[u][0], y = xs[u][1], z = xs[u][2];
                const bool has = t.live && (x != 0.0 || y != 0.0 || z != 0.0);
                const bool isk = key < kmax;                               // (the class of a monomial is wave-uniform)
                const bool isg = !isk && key < lmax && (key & kmask) == 0;
                if (isk) {
                    if (has) {
                        if (nk < cf.capL) {
                            cf.link_keys[idx * cf.capL + nk] = (uint32_t)key;
                            cf.link_coeff[(idx * cf.capL + nk) * 3 + 0] = x; cf.link_coeff[(idx * cf.capL + nk) * 3 + 1] = y; cf.link_coeff[(idx * cf.capL + nk) * 3 + 2] = z;
                        }
                        nk++;
                    }
                } else if (isg) {
                    if (has) {
                        // (register arrays are not indexed by a per-lane value: three explicit generator columns)
                        if (ng == 0) { g[0] = x; g[6] = y; g[12] = z; }
                        else if (ng == 1) { g[1] = x; g[7] = y; g[13] = z; }
                        else if (ng == 2) { g[2] = x; g[8] = y; g[14] = z; }
                        ng++;
                    }
                } else {
                    ra[0] += has ? fabs(x) : 0.0; ra[1] += has ? fabs(y) : 0.0; ra[2] += has ? fabs(z) : 0.0;
                }
            }
        }
    }
    if (__ballot(t.live && ng > 3) != 0ull) pzw::flag(t.w, pzw::ERR_LINK_GENS);
    if (__ballot(t.live && nk > cf.capL) != 0ull) pzw::flag(t.w, pzw::ERR_TABLE_OVERFLOW);
    if (t.live) {
        if (nk > cf.capL) nk = cf.capL;
        cf.link_count[idx] = nk;
        for (int e = 0; e < 3; e++) {
            const double r = p.hdr[((size_t)tv::H_IND * 3 + e) * tv::GR + lane] + ra[e];   // (inside `if (t.live)`: a lane with a step of its own, its place in a row is its lane)
            cf.link_center[idx * 3 + e] = p.hdr[((size_t)tv::H_CEN * 3 + e) * tv::GR + lane];
            cf.link_indep[idx * 3 + e] = r;
            g[e * 6 + 3 + e] = r;
        }
        for (int i = 0; i < 18; i++) gens[i] = g[i];
    }
    WSYNC();
}

// disturbance, reduce(u_nom), robust-input radius (RT/armour_main.cu:133-141,172-205), per lane; see finish_torque above
// The tables of joint j are written by wave (j mod nshare) of the `nshare` waves that call this (all with the same u_nom).
__device__ TV_NOINLINE void finish_torque_tv(TChain& c, TPZ* u_nom, int b, int t_lane, int share_id, int nshare) {
    PZ_KEEP_RETURN_ADDRESS();
    const P1Cfg& cf = *c.cf;
    TW& t = c.w;
    const int lane = t.w.lane;
    const int n = c.n, T = cf.T;
    const pzkey_t kmax = pzkey_bit(2 * n);
    Itv rho = {0.0, 0.0};
    double tr[ARMOUR_MAX_FACTORS], un_ind[ARMOUR_MAX_FACTORS];
    for (int j = 0; j < n; j++) {
        const TPZ& p = u_nom[j];
        const int rl = t.rl;
        const double pcen = p.hdr[((size_t)tv::H_CEN) * tv::GR + rl], pind = p.hdr[((size_t)tv::H_IND) * tv::GR + rl], pind2 = p.hdr[((size_t)tv::H_IND2) * tv::GR + rl];
        const double dcen = pcen - pcen;
        const double rad = pind2 + pind;
        const double lo = dcen - rad, hi = dcen + rad;
        rho = iadd(rho, imul(iv(lo, hi), iv(lo, hi)));
        tr[j] = cf.rb.alpha * (cf.rb.M_max - cf.rb.M_min) * cf.ub.eps + 0.5 * fmax(fabs(lo), fabs(hi));
        if (j % nshare != share_id) continue;
        const int cnt = tv::uni(t.w.cnt[p.id]);
        const size_t idx = ((size_t)b * n + j) * T + t_lane;
        double ra = 0.0;
        int nk = 0;
        const GLB_AS double* pc = tv::uni_ptr(p.coef) + rl;
        for (int m0 = 0; m0 < cnt; m0 += 64) {
            const pzkey_t key_v = m0 + lane < cnt ? p.keys[m0 + lane] : 0ull;
            const int nn = min(64, cnt - m0);
            constexpr int kU = 16;   // coefficient rows in flight
            for (int q0 = 0; q0 < nn; q0 += kU) {
                double xs[kU];
#pragma unroll
                for (int u = 0; u < kU; u++) xs[u] = pc[(size_t)(m0 + min(q0 + u, nn - 1)) * tv::GR];
#pragma unroll
                for (int u = 0; u < kU; u++) {
                    if (q0 + u >= nn) break;
                    const pzkey_t key = pzkey_readlane(key_v, q0 + u);
                    const double x = xs[u];
                    if (key < kmax) {   // (wave-uniform)
                        if (t.live && x != 0.0) {
                            if (nk < cf.capT) { cf.tq_keys[idx * cf.capT + nk] = (uint32_t)key; cf.tq_coeff[idx * cf.capT + nk] = x; }
                            nk++;
                        }
                    } else {
                        ra += t.live ? fabs(x) : 0.0;   // (x = 0 adds nothing)
                    }
                }
            }
        }
        if (__ballot(t.live && nk > cf.capT) != 0ull) pzw::flag(t.w, pzw::ERR_TABLE_OVERFLOW);
        if (nk > cf.capT) nk = cf.capT;
        const double ind = pind + ra;
        if (t.live) { cf.tq_count[idx] = nk; cf.tq_center[idx] = pcen; cf.tq_indep[idx] = ind; }
        un_ind[j] = ind;
    }
    const double rho_hi = up(sqrt(rho.hi));
    for (int j = share_id; j < n; j += nshare) {
        double v = tr[j];
        v += 0.5 * rho_hi;
        v += un_ind[j];
        v += cf.rb.friction[j];
        if (t.live) cf.torque_radius[((size_t)b * n + j) * T + t_lane] = v;
    }
    WSYNC();
}

// LDS: NW x { sort buffers skey[cap] | sidx[cap] | status | staging rows for a product's short operand } | count table | mailbox
__host__ __device__ inline size_t tv_lds_fixed(int cap_key, int cap_raw) { return ((size_t)cap_key * sizeof(pzkey_t) + (size_t)cap_raw * 2 + pzw::ST_WORDS * sizeof(int) + 15) & ~(size_t)15; }
__host__ __device__ inline size_t tv_lds_fixed(int cap) { return tv_lds_fixed(key_cap(cap), cap); }   // (`cap` raw terms: p1_reach.hip key_cap)
__host__ __device__ inline size_t tv_lds_wave(int cap, int stage_rows) { return tv_lds_fixed(cap) + (size_t)stage_rows * 64 * sizeof(double); }
__host__ __device__ inline size_t tv_lds_shared() { return ((size_t)(kMaxSlots + kTvMbWords) * sizeof(int) + 15) & ~(size_t)15; }
// sort buffers of the forward-kinematics wave of a four-wave block: its own products have <= 0.9 k raw terms; the omega recursion it also
// carries has rotation x vector products, which are ranked (<= 4 runs over the vector's keys) and need room for the permutation only
constexpr int kTvFkCap = key_cap(1024), kTvFkCapRaw = 2048;   // (128-bit keys: the same bytes, half the key entries -- so that four-wave blocks still fit the LDS; p1_reach.hip key_cap)
__host__ __device__ inline size_t tv_lds_fixed_fk() { return tv_lds_fixed(kTvFkCap, kTvFkCapRaw); }
__host__ __device__ inline size_t tv_lds_bytes(int cap, int stage_rows, int stage_rows_other, int nw) {
    const int nmain = nw == 4 ? 3 : nw;
    return (size_t)nmain * tv_lds_fixed(cap) + (nw == 4 ? tv_lds_fixed_fk() : 0) + (size_t)(stage_rows + (nw - 1) * stage_rows_other) * 64 * sizeof(double) + tv_lds_shared();
}

// One block per (problem, time group) item, striding over the items; group g of a problem holds the time steps
// [g * lanes_per_group, min(T, (g + 1) * lanes_per_group)).  NW = 1: one wave plays every role in turn.  NW = 3: the roles of
// run_rnea run concurrently on three waves, each with its own sort buffers and staging rows (one block per CU: the latency of
// a chain is what a batch of 128 problems pays).
// items [0, n_items): the RNEA of a group; with fk_items > 0, items [n_items, n_items + fk_items) are the forward kinematics
// of group (it - n_items) -- it shares nothing with the RNEA but the JRS rotations, which it rebuilds -- and the first n_items
// then leave it out (the same split as the per-step kernel's).
// NW = 8: four role waves and a dedicated helper wave behind each (pz_tv.h "Dedicated helpers"): two waves per SIMD.
template <int NW>
__global__ __launch_bounds__(64 * NW) P1_TV_OCC void armour_p1_tv_kernel(P1Cfg cf) {
    P1_TV_PIN();
    constexpr int NP = NW == 8 ? 4 : NW;   // the waves that play roles
    const int groups_per_problem = cf.tv_groups, lanes_per_group = cf.tv_lanes, capTv = cf.tv_cap;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TChain c;
    c.cf = &cf;
    c.n = cf.n; c.J = cf.J;
    c.capW = capTv;
    c.L = make_tlayout(cf.J, cf.n, capTv, NP, NW == 8 ? 4 : 0);
    c.arena = (GLB_AS unsigned char*)cf.arena + (size_t)blockIdx.x * cf.arena_bytes;
    c.nw = NP;
    c.wid = NW == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool helper_wave = NW == 8 && c.wid >= NP;
    LDS_AS unsigned char* lds = (LDS_AS unsigned char*)smem;
    // per wave: sort buffers | status | staging rows.  In a three-wave block wave 1 -- the angular recursion, whose products all
    // have a joint rotation (36 rows) as the short operand and which is the busiest role -- gets the larger staging area.
    const int my_stage = (NP == 1 || c.wid == 1) ? cf.tv_stage_rows : cf.tv_stage_rows_other;
    const bool fk_bufs = NP == 4 && c.wid == 3;
    const int my_cap_key = fk_bufs ? kTvFkCap : cf.capKey, my_cap_raw = fk_bufs ? kTvFkCapRaw : cf.capRaw;
    // waves 0..2: [sort buffers (cap) | status | staging]; wave 3 of a four-wave block: the same with the smaller buffers
    const int lw = min(c.wid, 3);   // (a helper wave has no buffers of its own: the address below is not used)
    const int stage_before = NP == 1 ? 0 : (lw > 0 ? cf.tv_stage_rows_other : 0) + (lw > 1 ? cf.tv_stage_rows : 0) + (lw > 2 ? cf.tv_stage_rows_other : 0);
    LDS_AS unsigned char* mine = lds + (size_t)lw * tv_lds_fixed(cf.capRaw) + (size_t)stage_before * 64 * sizeof(double);
    c.w.w.skey = (LDS_AS pzkey_t*)mine;
    c.w.w.sidx = (LDS_AS uint16_t*)(mine + (size_t)my_cap_key * sizeof(pzkey_t));
    c.w.w.lstat = (LDS_AS int*)(mine + (size_t)my_cap_key * sizeof(pzkey_t) + (size_t)my_cap_raw * 2);
    c.w.w.mg = nullptr;   // (the time-vectorised walks keep the prune margin in registers: pz_tv.h mtrk)
    c.w.stage = (LDS_AS double*)(mine + tv_lds_fixed(my_cap_key, my_cap_raw));
    c.w.stage_rows = my_stage;
    LDS_AS unsigned char* shared = lds + tv_lds_bytes(cf.capRaw, cf.tv_stage_rows, cf.tv_stage_rows_other, NP) - tv_lds_shared();
    c.w.w.cnt = (LDS_AS int*)shared;
    c.mb = c.w.w.cnt + kMaxSlots;
    c.w.w.cap_raw = my_cap_raw;
    c.w.w.cap_key = my_cap_key;
    c.w.w.thr = cf.pr.simplify_threshold;
    c.w.w.thr_sq = pzw::sq_threshold(c.w.w.thr);
    c.w.w.lane = threadIdx.x & 63;
    pzw::solo(c.w.w);
    const int lane = c.w.w.lane;
    if constexpr (NW == 8) {
        // channel p: role wave p -> helper wave NP + (p + shift) % NP (which SIMD a wave lands on is the dispatcher's choice; the shift is a tuning knob)
        const int shift = cf.tv_help_shift & 3;
        const int p = helper_wave ? ((c.wid - NP - shift) & 3) : c.wid;
        if (helper_wave) c.w.w.lstat = c.mb + kHelpStat + (c.wid - NP) * pzw::ST_WORDS;
        if (threadIdx.x < kHelpChannels * tv::HJ_WORDS) c.mb[kHelpBase + threadIdx.x] = 0;
        c.w.hch = c.mb + kHelpBase + p * tv::HJ_WORDS;
        c.w.hded = true; c.w.hseq = 0;
        c.w.hnum = cf.tv_walk_helpers > 1 ? cf.tv_walk_helpers : 16; c.w.hmin = cf.tv_help_min;
    }
    if (lane < pzw::ST_WORDS) c.w.w.lstat[lane] = 0;
    if constexpr (NW == 8) __syncthreads();   // the channels are empty before anybody posts or polls
    for (int it0 = blockIdx.x; it0 < cf.n_items + cf.fk_items; it0 += gridDim.x) {
        const bool fk_only = it0 >= cf.n_items;
        const int it = fk_only ? it0 - cf.n_items : it0;
        const int b = it / groups_per_problem, g = it - b * groups_per_problem;
        const int t0 = g * lanes_per_group;
        const int nl = min(lanes_per_group, cf.T - t0);
        // a lane without a step of its own shadows step (lane mod nl) of the group in everything -- inputs, arithmetic -- and writes no final table; its
        // place in a row is its own while the rows have one for it, the shadowed lane's otherwise (the same value to the same address, from two
        // lanes; shadowing the LAST step put fifteen lanes on one address: B = 24 6.65 -> 6.84 ms) (pz_tv.h)
        const int sl = lane % nl;
#ifdef TV_FORCE_ROW_PLACES   // development: lanes beyond this share the shadowed lane's place although the rows are wider
        constexpr int kRowPlaces = TV_FORCE_ROW_PLACES;
#else
        constexpr int kRowPlaces = tv::GR;
#endif
        c.w.active = true; c.w.live = lane < nl; c.w.rl = lane < kRowPlaces ? lane : sl;
        const int t_lane = t0 + sl;
        c.w.w.mabs = __builtin_inf();   // this item's prune margin (pz_tv.h mtrk; reduced over the lanes -- the group's time steps -- below)
        if constexpr (NW == 8) {
            if (helper_wave) {   // until the primary's HK_EXIT at the end of the item
                tv::serve_loop(c.w, c.H(c.wid - NP));
                pzw::margin_reduce_store(c.w.w.mabs, lane, cf.margin ? cf.margin + (size_t)b : nullptr);
                continue;
            }
        }
        c.freeV = (1ull << c.L.nV) - 1ull;
        c.freeS = (1u << kTvNS) - 1u;
        c.role = 0;
        for (int i = threadIdx.x; i < kMaxSlots; i += 64 * NP) c.w.w.cnt[i] = 0;
        c.bar();
        c.phase_log = cf.phase_log ? cf.phase_log + (size_t)blockIdx.x * 8 : nullptr;
        if (cf.phase_log != nullptr && threadIdx.x == 0) { cf.phase_log[(size_t)blockIdx.x * 8 + 0] = clock64(); cf.phase_log[(size_t)blockIdx.x * 8 + 6] = it; }   // (ARMOUR_P1_TRACE: when this block's item began and ended)
#ifdef TV_PROFILE
        const long long tvp_start = clock64();
        c.w.c_wait = c.w.c_sort = c.w.c_walk = c.w.c_cc = c.w.n_raw = c.w.n_calls = c.w.n_emit = 0;
        c.w.c_hwait = c.w.n_shared = c.w.n_shared_terms = 0;
        for (int q = 0; q < 8; q++) { c.w.c_fn[q] = 0; c.w.n_fn[q] = 0; }
        for (int q = 0; q < 3; q++) { c.w.c_type[q] = 0; c.w.n_type[q] = 0; }
#endif
        build_jrs_tv(c, b, t_lane, fk_only);
        c.bar();
#ifdef TV_PROFILE
        if (threadIdx.x == 0 && blockIdx.x == 0) printf("[tv item %d] jrs %lld cycles\n", it, (long long)clock64() - tvp_start);
#endif
        TPZ u_nom[ARMOUR_MAX_FACTORS];
        if (fk_only) {
            if (c.is(2)) {
                FkStateT<TPZ> fk;
                c.role = 2;
                fk_begin(c, fk);
                for (int i = 0; i < c.J; i++) fk_step(c, fk, i, b, t_lane);
                c.freeVs(fk.T);
            }
        } else {
            if (NP >= kRoles && cf.tv_free_running) run_rnea_free(c, u_nom, b, t_lane);
            else run_rnea(c, u_nom, b, t_lane);
#ifdef TV_PROFILE
            const long long tvp_rnea = clock64();
#endif
            if (NP >= kRoles && cf.tv_free_running) {   // every wave takes its share of the joints' tables (the 1x1 slots of u_nom: the mailbox, past the last barrier of the RNEA)
                for (int j = 0; j < c.n; j++) u_nom[j] = c.S(t3_ld(&c.mb[T3_U + j]));
                finish_torque_tv(c, u_nom, b, t_lane, c.wid, NP);
            } else if (c.is(0)) finish_torque_tv(c, u_nom, b, t_lane, 0, 1);
#ifdef TV_PROFILE
            if (lane == 0 && blockIdx.x == 0) printf("[tv item %d wave %d] rnea done at %lld, torque tables %lld cycles\n", it, c.wid, tvp_rnea - tvp_start, (long long)clock64() - tvp_rnea);
#endif
        }
        pzw::margin_reduce_store(c.w.w.mabs, lane, cf.margin ? cf.margin + (size_t)b : nullptr);
        c.bar();
        if (cf.phase_log != nullptr && threadIdx.x == 0) cf.phase_log[(size_t)blockIdx.x * 8 + 5] = clock64();
        if constexpr (NW == 8) tv::hj_post_ctl(c.w, tv::HK_EXIT);
#ifdef TV_PROFILE
        if (lane == 0 && blockIdx.x == 0) printf("[tv item %d wave %d] walks by type: mul %lld cycles / %lld raw, cross %lld / %lld, sums %lld / %lld\n", it, c.wid, c.w.c_type[0], c.w.n_type[0], c.w.c_type[1], c.w.n_type[1], c.w.c_type[2], c.w.n_type[2]);
#ifdef TV_PROFILE_FULL
        if (threadIdx.x == 0 && blockIdx.x == 0) printf("[tv item %d] walk: load phase %lld, wait for the loads %lld, process phase %lld, chunk prologue %lld cycles, %lld batches, %lld terms\n", it, tv::g_tvprof[0], tv::g_tvprof[4], tv::g_tvprof[1], tv::g_tvprof[2], tv::g_tvprof[3], tv::g_tvprof[5]);
#endif
        if (lane == 0 && blockIdx.x == 0) printf("[tv item %d wave %d] whole calls (cycles / calls): sorted product %lld / %lld, constant-left product %lld / %lld, cross %lld / %lld, sums %lld / %lld, constant cross %lld / %lld, helper service %lld / %lld, set+transpose %lld / %lld, link tables %lld / %lld\n", it, c.wid,
                                                 c.w.c_fn[0], c.w.n_fn[0], c.w.c_fn[1], c.w.n_fn[1], c.w.c_fn[2], c.w.n_fn[2], c.w.c_fn[3], c.w.n_fn[3], c.w.c_fn[4], c.w.n_fn[4], c.w.c_fn[5], c.w.n_fn[5], c.w.c_fn[6], c.w.n_fn[6], c.w.c_fn[7], c.w.n_fn[7]);
        if (lane == 0 && blockIdx.x == 0) printf("[tv item %d wave %d] shared walks: %lld jobs, %lld raw terms; waited %lld cycles on the helper channel\n", it, c.wid, c.w.n_shared, c.w.n_shared_terms, c.w.c_hwait);
        if (lane == 0 && blockIdx.x == 0) printf("[tv item %d wave %d] total %lld cycles (waited %lld, %lld of it in the forward pass; forward done at %lld): sort %lld walk %lld cross_const %lld | %lld sorted operator calls, %lld raw terms, %lld emitted\n", it, c.wid, (long long)clock64() - tvp_start, c.w.c_wait, c.w.c_wait_fwd, c.w.c_fwd - tvp_start, c.w.c_sort, c.w.c_walk, c.w.c_cc, c.w.n_calls, c.w.n_raw, c.w.n_emit);
#endif
    }
#ifdef TV_PROFILE
    if (lane == 0 && blockIdx.x == 0) printf("[tv maxraw wave %d] %d out %d\n", c.wid, c.w.w.lstat[pzw::ST_MAX_RAW], c.w.w.lstat[pzw::ST_MAX_OUT]);
#endif
    if (lane == 0) {  // every wave reports its own flags and maxima
        if (c.w.w.lstat[pzw::ST_ERR]) atomicOr(&cf.status[pzw::ST_ERR], (unsigned)c.w.w.lstat[pzw::ST_ERR]);
        atomicMax(&cf.status[pzw::ST_MAX_RAW], (unsigned)c.w.w.lstat[pzw::ST_MAX_RAW]);
        atomicMax(&cf.status[pzw::ST_MAX_OUT], (unsigned)c.w.w.lstat[pzw::ST_MAX_OUT]);
    }
}

}  // namespace tvchain
