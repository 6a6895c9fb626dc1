// Included by p1_reach.hip (inside its anonymous namespace, after run_rnea): the free-running RNEA of a multi-wave block, shared by the
// per-step chain (Chain) and the time-vectorised one (tvchain::TChain).
// ---------------------------------------------------------------------------------------------------------------------
// The RNEA of a three-wave block WITHOUT lock-step barriers.  run_rnea (above) meets at a
// block barrier after every joint, so every joint costs the slowest role's time -- and which role that is changes: the
// angular recursion in the forward pass, the sums of the n-recursion in the backward pass (own work of the roles 14 / 19 /
// 10 M cycles, but 28 M from start to end).  Here each wave runs its recursion as far as its inputs allow:
//   forward   wave 1: state_{s+1} = f(state_s)                      never waits
//             wave 0: lacc_{s+1} = f(lacc_s, state_s)               waits for state_s
//             wave 2: N_{s-1}, F_{s-1} = f(state_s, lacc_s), FK     waits for state_s and lacc_s
//   backward  wave 1: a2_i = R f, c2_i = p x a2_i, f = a2_i + F_i   never waits
//             wave 0: n = N_i + R n + com x F_i + c2_i, u_i         waits for c2_i
// Progress counters and slot handles live in LDS; a producer finishes its stores (s_waitcnt) and then raises its counter, a
// consumer polls the counter (the waves of a block share the CU's L1: what __syncthreads() relied on as well).  A slot is
// freed by its owner once the counters show every reader past it (state_k: K joints back).  Same operators on the same
// operands as run_rnea: bit-identical tables.
enum { T3_ST = 0, T3_LA = 3 * (ARMOUR_MAX_JOINTS + 1), T3_N = T3_LA + ARMOUR_MAX_JOINTS + 1, T3_F = T3_N + ARMOUR_MAX_JOINTS, T3_C2 = T3_F + ARMOUR_MAX_JOINTS,
       T3_A2 = T3_C2 + ARMOUR_MAX_JOINTS, T3_X1 = T3_A2 + ARMOUR_MAX_JOINTS, T3_X2 = T3_X1 + ARMOUR_MAX_JOINTS, T3_NA = T3_X2 + ARMOUR_MAX_JOINTS,
       T3_X3 = T3_NA + ARMOUR_MAX_JOINTS,   // w x (w_aux x com) of a late link, built by a wave that is through with its recursion (tail_cross)
       T3_CNT = T3_X3 + ARMOUR_MAX_JOINTS + 1, T3_C1 = T3_CNT, T3_C0, T3_CC2, T3_B1, T3_B2, T3_CA, T3_C3, T3_C3P,
       T3_JR2,         // two CUs, level 3: the joints past the fourth have their JRS (jrs_late_joints)
       T3_NB, T3_UB,   // run_backward_remote: joints the n-recursion is through with | joints whose torque sum is built (both counted from the last joint)
       T3_NCNT = 12, T3_U = T3_CNT + T3_NCNT, T3_WORDS = T3_U + ARMOUR_MAX_FACTORS };   // T3_U: the 1x1 slots of u_nom, for the waves that share the torque tables
// LDS mailbox of run_rnea / run_rnea_free, followed by the two walk-helper channels of the time-vectorised four-wave blocks
// (pz_tv.h "One walk on two waves": channel 0 = f-recursion wave -> wave 2, channel 1 = n-recursion wave -> wave 3)
constexpr int kHelpBase = MB_WORDS > T3_WORDS ? MB_WORDS : T3_WORDS;
constexpr int kMbWords = kHelpBase + 2 * tv::HJ_WORDS;
// the time-vectorised blocks: four channels (eight-wave blocks: one per role wave, to its dedicated helper) and the helpers' status words
constexpr int kHelpChannels = 4;
constexpr int kHelpStat = kHelpBase + kHelpChannels * tv::HJ_WORDS;
constexpr int kTvMbWords = kHelpStat + kHelpChannels * pzw::ST_WORDS;
// an operator of `c`'s wave that shares its walk with the helper wave on channel k (chains without walk helpers: plain call)
template <class CH, class Fn>
__device__ inline auto with_walk_helper(CH& c, int k, bool on, Fn fn) {
    if constexpr (CH::kWalkHelpers) {
        if (c.w.hded) return fn();   // (a dedicated helper is behind every operator anyway)
        c.w.hch = on ? c.mb + kHelpBase + k * tv::HJ_WORDS : nullptr;
        auto r = fn();
        c.w.hch = nullptr;
        return r;
    } else return fn();
}
// (`tmp`: a scratch slot the helper keeps for the whole pass -- the primary reads the partial sums in its header rows after the
//  hand-back, so it must not be given to another operator of the helper wave in between)
template <class CH>
__device__ inline void serve_walk_helper(CH& c, int k, const typename CH::PZT& tmp) {
    if constexpr (CH::kWalkHelpers) {
        c.w.hch = c.mb + kHelpBase + k * tv::HJ_WORDS;
        tv::serve_walk(c.w, tmp);
        c.w.hch = nullptr;
    }
}
__device__ inline int t3_ld(LDS_AS int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <class CH>
__device__ inline bool w_lane0(const CH& c) { return c.wave().lane == 0; }
template <class CH>
__device__ inline void t3_signal(CH& c, int word, int value) {
    WSYNC();   // this wave's stores (result rows, keys, LDS count table, mailbox) are done
    if (c.wave().lane == 0) __hip_atomic_store(&c.mb[word], value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    c.prof_signal(word, value);
}
template <class CH>
__device__ inline void t3_wait(CH& c, int word, int value) {
    const long long tw0 = c.prof_clock();
    int spins = 0;
    while (t3_ld(&c.mb[word]) < value) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1 << 24)) { flag(c.wave(), ERR_SLOT_OVERFLOW); break; }   // never seen; ends the wait instead of the box
    }
    WSYNC();
    c.prof_waited(tw0);
}
template <class CH>
__device__ inline void t3_post(CH& c, int word, const typename CH::PZT& p) { if (c.wave().lane == 0) c.mb[word] = p.id - c.L.idV; }
template <class CH>
__device__ inline typename CH::PZT t3_take(const CH& c, int word) { return c.V(t3_ld(&c.mb[word])); }

// ---------------------------------------------------------------------------------------------------------------------
// The backward pass of a four-wave block of the PER-STEP kernel with two waves on every operator of the two recursions (round 4).
// A time step's backward pass is two chains of operators, each waiting for the one before -- f = R f + F on wave 1, n = ((N + R n) +
// com x F) + p x (R f) on wave 0 -- and in the per-step kernel an operator's lanes are its TERMS: 200 .. 1900 raw terms on 64 lanes,
// while waves 2 and 3 have nothing of their own left (release-build stamps of the slowest step of a Kinova build: 3.6 of the step's
// 5.9 M cycles are this pass, and waves 2 / 3 wait for 3.8 / 4.0 M of them).  So for this pass wave 2 becomes the second half of wave 1
// and wave 3 the second half of wave 0: both halves of a pair run the SAME role code -- allocations included, so that they agree on
// every slot -- and the operators stride their passes over the raw terms by 128 lanes (pz_wave.h: psync(), pair_split(), pair_finish()).
// What is not worth sharing runs on one half while the other does something else or waits: the constant cross products com x F (wave 0)
// and p x (R f) (wave 3, as before) side by side, the 1x1 torque sums on wave 0.
// Every bit of every result is that of the one-wave operators (pz_wave.h: a reduce pass adds its pruned amounts as the same two partial
// sums whichever waves work on it); ARMOUR_OPT_P1_STEP_PAIRS = 0 keeps the idle waves idle.
template <class CH>
__device__ inline void run_backward_pairs(CH& c, typename CH::PZT* u, int n_tail) {
    typedef typename CH::PZT TPZ;
    const P1Cfg& cf = *c.cf;
    auto& w = c.w;
    const int J = c.J;
    LDS_AS int* px = c.mb + kHelpBase;             // the words of the two walk-helper channels, which this chain does not use
    LDS_AS int* fvx = px + 2 * pzw::PW_WORDS;      // allocator state of waves 0 / 1 for their second halves
    static_assert(2 * pzw::PW_WORDS + 4 <= 2 * tv::HJ_WORDS, "pair words");
    if (threadIdx.x < 2 * pzw::PW_WORDS) px[threadIdx.x] = 0;
    if (c.wid <= 1 && w_lane0(c)) { fvx[2 * c.wid] = (int)(unsigned)(c.freeV & 0xffffffffull); fvx[2 * c.wid + 1] = (int)(unsigned)(c.freeV >> 32); }
    c.bar();   // (A2)
    const int first = (c.wid == 1 || c.wid == 2) ? 1 : 0;   // pairs: waves 1 + 2 (f), waves 0 + 3 (n)
    const unsigned long long saved = c.freeV, pf = c.part_mask(first);
    if (c.wid >= 2) {   // the second half allocates what the first allocates: it starts from the first's state of the first's part of the pool
        const unsigned long long fv = (unsigned long long)(unsigned)t3_ld(&fvx[2 * first]) | ((unsigned long long)(unsigned)t3_ld(&fvx[2 * first + 1]) << 32);
        c.freeV = (c.freeV & ~pf) | (fv & pf);
    }
    c.pair_begin(first, px + (first == 1 ? 0 : pzw::PW_WORDS));
    const bool h0 = w.half == 0;
    if (first == 1) {
        c.role = 1;
        TPZ f = c.allocV();
        if (h0) set_const(w, f, nullptr, nullptr);
        psync(w);
        for (int i = J - 1; i >= 0; i--) {
            const TPZ Rn = c.R(i + 1), Fi = t3_take(c, T3_F + i);
            TPZ a2 = c.mulMV(Rn, f);
            if (h0) { t3_post(c, T3_A2 + i, a2); t3_signal(c, T3_B1, J - i); }
            TPZ f2 = c.add(a2, Fi); c.freeVs(f);
            f = f2;
        }
        c.freeVs(f);
    } else {
        c.role = 0;
        TPZ nn = c.allocV();
        if (h0) set_const(w, nn, nullptr, nullptr);
        psync(w);
        for (int i = J - 1; i >= 0; i--) {
            const TPZ Rn = c.R(i + 1), Fi = t3_take(c, T3_F + i), Ni = t3_take(c, T3_N + i);
            TPZ a1 = c.mulMV(Rn, nn);
            TPZ c1;
            if (h0) c1 = c.crossMatPz(&cf.rb.com[3 * i], Fi);   // (an operator of this wave alone: no sort buffers)
            else {
                c1 = c.allocV();   // (the slot the first half has just taken)
                c.role = 3;
                t3_wait(c, T3_B1, J - i);
                TPZ c2 = c.crossMatPz(&cf.rb.trans[3 * (i + 1)], t3_take(c, T3_A2 + i));
                t3_post(c, T3_C2 + i, c2);
                c.role = 0;
            }
            psync(w);
            const TPZ c2 = t3_take(c, T3_C2 + i);
            TPZ n2 = c.sum4(Ni, a1, c1, c2); c.freeVs(a1); c.freeVs(c1); c.freeVs(nn);  // ((N + a1) + c1) + c2
            nn = n2;
            if (cf.rb.axes[i] != 0) {
                if (h0) {
                    const auto st = c.solo_begin();
                    const int ax = abs(cf.rb.axes[i]) - 1;
                    u[i] = c.comb3(elem(w, n2, ax), 1.0, view(w, c.qdda(i)), cf.rb.armature[i], view(w, c.qd(i)), cf.rb.damping[i]);
                    if (w_lane0(c)) c.mb[T3_U + i] = u[i].id - c.L.idS;
                    c.solo_end(st);
                }
                psync(w);   // the pair's sort buffers are free again
            }
        }
        c.freeVs(nn);
    }
    c.pair_end();
    if (c.wid == 2) c.freeV = saved;
    if (c.wid == 3) { const unsigned long long own = c.part_mask(3); c.freeV = (c.freeV & own) | (saved & ~own); }
    c.role = c.wid;
    c.bar();   // (B) both recursions are through
    if constexpr (CH::kTwoCu) c.phase(4);
    if (c.wid == 1) {
        for (int i = 0; i < J; i++) c.freeVs(t3_take(c, T3_A2 + i));
        for (int i = J - n_tail; i < J; i++) c.freeVs(t3_take(c, T3_N + i));
    } else if (c.wid == 3) {
        for (int i = 0; i < J; i++) c.freeVs(t3_take(c, T3_C2 + i));
    } else if (c.wid == 2) {
        for (int i = 0; i < J; i++) { if (i < J - n_tail) c.freeVs(t3_take(c, T3_N + i)); c.freeVs(t3_take(c, T3_F + i)); }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// A time step on TWO compute units (round 6; per-step kernel, four-wave blocks, a lone problem: 2 T blocks on 256 CUs).
// Of what a time step's forward pass computes, three families of products read nothing but the angular VELOCITY recursions -- w_s and w_aux_s,
// which are cheap (a rotation times a short vector, plus one entry) -- and none of the angular or linear ACCELERATION:
//     c3L_s = w_s x (w_aux_s x p_s)             the linear-acceleration step's cross product     (RT/Dynamics.cu:107-109)
//     c3C_s = w_s x (w_aux_s x com_{s-1})       the same for the link's centre of mass, in F     (:125-127)
//     crN_s = w_aux_s x (I_{s-1} w_s)           the gyroscopic half of the moment N              (:130-132)
// -- 21 PZ x PZ cross products of up to 1 700 raw terms each, 0.4 M of the 1.2 M cycles every wave of the block spends in the forward pass.
// A HELPER block on a second CU builds the JRS of the same time step, runs the two velocity recursions AGAIN (same operators on the same
// operands: the same bits as the main block's own), builds the three families on three waves and the step's forward kinematics behind the
// recursions on the fourth (the items of their own it used to be), and PUBLISHES each product; the main block takes them where it used to
// compute them.  Data flows one way, helper -> main, and a published slot is never written again within the item, so there is no
// acknowledgement, no allocator protocol, and the helper never waits for the main block.
// Hand-off (MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility"): the product's rows are plain stores into
// the helper's arena, drained by the producing wave (s_waitcnt vmcnt(0)); its header (count, slot, centre, radii: 80 bytes of the helper's LDS)
// goes into a 128-byte record of the item's exchange area with sc1 stores, drained again, then the flag word (sc1, tagged with the launch's
// epoch so that the area is never cleared).  The taking wave polls the flag with sc1 loads, invalidates its CU's L1 (agent-scope acquire:
// a line of the helper's arena that straddles two slots may have been fetched before its second half was written), reads the record with
// sc1 loads and makes a PZ whose rows point into the helper's arena and whose header is a slot of its own pool.  Plain stores are visible
// to another CU through the L2 only on the SAME XCD: both blocks publish their XCC_ID at the start of the item, the main block decides --
// same XCD: two CUs; different, or no sign of the helper: everything itself, as before -- and publishes the decision for the helper.  Block
// b's helper is block helper0 + b with helper0 a multiple of 8 (blocks are dealt round-robin to the XCDs: observed, not promised -- hence the check).
// A take that never sees its flag (cut off after ~0.1 s) raises ERR_HELPER; the host then builds again on one CU per step and keeps the handle there.
enum { XK_C3L = 0, XK_C3C = 1, XK_CRN = 2, XK_C4 = 3,
       XK_F = 4,      // main -> helper: F_i, for the f-recursion of the backward pass (run_backward_remote)
       XK_C2 = 5,     // helper -> main: p x (R f) of joint i
       XK_MISC = 6,   // flags without a record: [0] main -> helper, the forward pass is over here (every product has been taken); [1] helper -> main, the f-recursion is through (the F_i may go)
       XK_KINDS = 7 };
constexpr int kXchSlots = ARMOUR_MAX_JOINTS + 1;
constexpr size_t kXchFlags = 128, kXchRecs = 512, kXchRecBytes = 128, kXchBytes = 12288;   // control line | flag words | records
static_assert(XK_KINDS * kXchSlots * sizeof(unsigned) <= kXchRecs - kXchFlags && kXchRecs + XK_KINDS * kXchSlots * kXchRecBytes <= kXchBytes, "exchange area");
__device__ inline unsigned xch_xcc() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return (v & 0xfu) + 1u; }
__device__ inline void xch_st(GLB_AS unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline unsigned xch_ld(const GLB_AS unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class CH>
__device__ inline void xch_hello(CH& c, bool helper) {   // before the JRS: who runs where
    if (c.wid == 0 && w_lane0(c)) xch_st((GLB_AS unsigned*)c.xch + (helper ? 1 : 0), ((unsigned)c.xch_epoch << 8) | xch_xcc());
}
// every wave of the block (two block barriers inside): true = this item runs on two CUs
template <class CH>
__device__ inline bool xch_decide(CH& c, bool helper) {
    LDS_AS int* word = c.mb + kMbWords - 2;
    if (c.wid == 0) {
        const GLB_AS unsigned* src = (const GLB_AS unsigned*)c.xch + (helper ? 2 : 1);
        const unsigned ep = (unsigned)c.xch_epoch;
        unsigned v = 0;
        const int limit = helper ? (1 << 13) : (1 << 11);   // (~0.7 us per poll: the main block gives the helper ~1.5 ms to show up, the helper waits longer than that for the decision)
        for (int spins = 0; spins < limit; spins++) { v = xch_ld(src); if ((v >> 8) == ep) break; __builtin_amdgcn_s_sleep(16); }
        int mode = 2;
        if ((v >> 8) == ep) mode = helper ? (int)(v & 0xffu) : ((v & 0xffu) == xch_xcc() ? 1 : 2);
        else if (!helper) flag(c.wave(), ERR_HELPER_LATE);   // (1.5 ms spent waiting: the host keeps this handle off two CUs from now on)
        if (!helper && w_lane0(c)) xch_st((GLB_AS unsigned*)c.xch + 2, (ep << 8) | (unsigned)mode);
        if (w_lane0(c)) *word = mode;
    }
    c.bar();
    const bool two = t3_ld(word) == 1;
    c.bar();
    return two;
}
template <class CH>
__device__ inline void xch_raise(CH& c, int kind, int idx) {   // a flag alone (this wave's stores drained first)
    WSYNC();
    if (w_lane0(c)) xch_st((GLB_AS unsigned*)(c.xch + kXchFlags) + kind * kXchSlots + idx, (unsigned)c.xch_epoch);
}
template <class CH>
__device__ inline bool xch_await(CH& c, int kind, int idx) {
    LDS_AS int* broken = c.mb + kMbWords - 3;
    const GLB_AS unsigned* fl = (const GLB_AS unsigned*)(c.xch + kXchFlags) + kind * kXchSlots + idx;
    const long long tw0 = c.prof_clock();
    bool ok = t3_ld(broken) == 0;
    if (ok) {
        int spins = 0;
        while (xch_ld(fl) != (unsigned)c.xch_epoch) {
            __builtin_amdgcn_s_sleep(16);
            if (++spins > (1 << 17)) { ok = false; break; }
        }
    }
    c.prof_waited(tw0);
    c.prof_signal(T3_CNT + 300 + 16 * kind + idx, (int)((c.prof_clock() - tw0) >> 10));   // (-DP1_STAMPS: how long this take waited, in 1024 cycles)
    if (!ok) { flag(c.wave(), ERR_HELPER); if (w_lane0(c)) *broken = 1; }
    return ok;
}
template <class CH>
__device__ inline void xch_publish(CH& c, int kind, int idx, const typename CH::PZT& p) {
    auto& w = c.wave();
    WSYNC();   // the product's rows have left this CU
    GLB_AS unsigned long long* rec = (GLB_AS unsigned long long*)(c.xch + kXchRecs + (size_t)(kind * kXchSlots + idx) * kXchRecBytes);
    if (w.lane < 10) {
        unsigned long long v;
        if (w.lane == 0) v = (unsigned long long)(unsigned)w.cnt[p.id] | ((unsigned long long)(unsigned)(p.id - c.L.idV) << 32);
        else v = (unsigned long long)__double_as_longlong(p.cen[w.lane - 1]);   // centre | radius | second radius: nine doubles in a row (mk_slot)
        __hip_atomic_store(rec + w.lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (w.lane == 0) xch_st((GLB_AS unsigned*)(c.xch + kXchFlags) + kind * kXchSlots + idx, (unsigned)c.xch_epoch);
    c.prof_signal(T3_CNT + 100 + 16 * kind + idx, 0);
}
template <class CH>
__device__ inline typename CH::PZT xch_take(CH& c, int kind, int idx) {
    typedef typename CH::PZT TPZ;
    auto& w = c.wave();
    const bool ok = xch_await(c, kind, idx);   // (one lost take: the others of this item do not wait again)
    TPZ r = c.allocV();
    if (!ok) { set_const(w, r, nullptr, nullptr); return r; }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // this CU's L1 holds nothing older than the flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const GLB_AS unsigned long long* rec = (const GLB_AS unsigned long long*)(c.xch + kXchRecs + (size_t)(kind * kXchSlots + idx) * kXchRecBytes);
    unsigned long long v = 0ull;
    if (w.lane < 10) v = __hip_atomic_load(rec + w.lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned lo0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xffffffffull)), hi0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    int cnt = (int)lo0, slot = (int)hi0;
    if (cnt < 0 || cnt > r.cap || slot < 0 || slot >= c.L.nV) { flag(w, ERR_HELPER); cnt = 0; slot = 0; }
    const TPZ src = mk_slot(c.peer_arena, c.L.offV, slot, r.cap, 3, c.L.idV, c.ci);
    r.keys = src.keys; r.coef = src.coef;
    if (w.lane >= 1 && w.lane < 10) r.cen[w.lane - 1] = __longlong_as_double((long long)v);
    if (w.lane == 0) w.cnt[r.id] = cnt;
    WSYNC();
    return r;
}
// ---------------------------------------------------------------------------------------------------------------------
// The backward pass with the f-recursion on the HELPER block (round 6; a time step on two CUs, level 3).
// With two waves on every operator (run_backward_pairs) the pass is as long as the n-recursion -- R n (pair), com x F_i on one half beside
// p x (R f) on the other, the four-term sum (pair), the 1x1 torque sum on one half: 7 x 107 k cycles -- and the f-recursion's pair (R f,
// R f + F: 7 x 73 k) cannot take the side products without becoming the slower chain itself (built and measured: the pass 2 % longer,
// docs/experiments/r06_lean_backward_local.inc.txt).  The helper block's product waves are through before the forward pass ends, so:
//   * the main block publishes every F_i as the forward pass builds it, and a flag when the pass is over (every product of the helper has
//     been taken by then: the helper's product waves give their slots back);
//   * the helper runs the f-recursion on a pair of waves (R f, R f + F_i) and p x (R f) on a third, which publishes it;
//   * com x F_i is built in the main block's FORWARD pass by its fourth wave (idle there at level 3), all but the last link's;
//   * the main block's n-pair does R n and the four-term sum, nothing else; its second wave builds the torque sums u_j as the n_j appear
//     (n_j is posted in the mailbox and given back one step late, after its sum); its third wave waits for the helper's last word, after
//     which the F_i may be given back.
// Same operators on the same operands: same bits.
template <class CH>
__device__ inline void run_backward_remote(CH& c, typename CH::PZT* u, int n_tail) {
    typedef typename CH::PZT TPZ;
    const P1Cfg& cf = *c.cf;
    auto& w = c.w;
    const int J = c.J;
    LDS_AS int* px = c.mb + kHelpBase;
    LDS_AS int* fvx = px + 2 * pzw::PW_WORDS;
    if (threadIdx.x < 2 * pzw::PW_WORDS) px[threadIdx.x] = 0;
    if (c.wid == 0 && w_lane0(c)) { fvx[0] = (int)(unsigned)(c.freeV & 0xffffffffull); fvx[1] = (int)(unsigned)(c.freeV >> 32); }
    if (c.wid == 0) xch_raise(c, XK_MISC, 0);   // (behind barrier (A): every wave of this block is through with the helper's products)
    c.bar();   // (A2)
    if (c.wid == 0 || c.wid == 3) {
        const unsigned long long saved = c.freeV, pf = c.part_mask(0);
        if (c.wid == 3) {
            const unsigned long long fv = (unsigned long long)(unsigned)t3_ld(&fvx[0]) | ((unsigned long long)(unsigned)t3_ld(&fvx[1]) << 32);
            c.freeV = (c.freeV & ~pf) | (fv & pf);
        }
        c.pair_begin(0, px + pzw::PW_WORDS);
        const bool h0 = w.half == 0;
        c.role = 0;
        TPZ nn = c.allocV();
        if (h0) set_const(w, nn, nullptr, nullptr);
        psync(w);
        TPZ pend = nn; int pend_joint = -1;   // n of joint pend_joint: given back once its torque sum is built
        for (int i = J - 1; i >= 0; i--) {
            const TPZ Rn = c.R(i + 1), Fi = t3_take(c, T3_F + i), Ni = t3_take(c, T3_N + i);
            TPZ a1 = c.mulMV(Rn, nn);
            TPZ c1;
            const bool own_c1 = i == J - 1;   // (F of the last link ends the forward pass: its product is built here)
            TPZ c2;
            if (own_c1) {
                // ... and p x (R_J f_J) with f_J = 0 beside it, on the second half: the helper's first step builds the same constant, but its
                // recursion only starts when this pass does -- waiting for it cost the n-recursion its first 60 k cycles
                if (h0) c1 = c.crossMatPz(&cf.rb.com[3 * i], Fi);
                else {
                    c1 = c.allocV();
                    c.role = 3;
                    const auto st = c.solo_begin();
                    TPZ z = c.allocV();
                    set_const(w, z, nullptr, nullptr);
                    TPZ a2 = c.mulMV(Rn, z); c.freeVs(z);
                    TPZ cl = c.crossMatPz(&cf.rb.trans[3 * (i + 1)], a2); c.freeVs(a2);
                    t3_post(c, T3_C2 + i, cl);
                    c.solo_end(st);
                    c.role = 0;
                }
                psync(w);
                c2 = t3_take(c, T3_C2 + i);
            } else {
                c1 = t3_take(c, T3_X2 + i);
                c2 = xch_take(c, XK_C2, i);   // (both halves: the same record into the same slot)
                psync(w);
            }
            TPZ n2 = c.sum4(Ni, a1, c1, c2); c.freeVs(a1); if (own_c1) c.freeVs(c1); else c.freeVs(c2);  // ((N + a1) + c1) + c2
            if (pend_joint >= 0) {
                if (cf.rb.axes[pend_joint] != 0) t3_wait(c, T3_UB, J - pend_joint);
                c.freeVs(pend);
            }
            if (i == J - 1) { c.freeVs(nn); pend_joint = -1; } else { pend = nn; pend_joint = i + 1; }
            nn = n2;
            if (h0) { t3_post(c, T3_ST + i, n2); t3_signal(c, T3_NB, J - i); }
        }
        if (pend_joint >= 0) {
            if (cf.rb.axes[pend_joint] != 0) t3_wait(c, T3_UB, J - pend_joint);
            c.freeVs(pend);
        }
        t3_wait(c, T3_UB, J);   // every torque sum is built (n_0 is read last)
        c.freeVs(nn);
        c.pair_end();
        if (c.wid == 3) { const unsigned long long own = c.part_mask(3); c.freeV = (c.freeV & own) | (saved & ~own); }
        c.role = c.wid;
    } else if (c.wid == 1) {
        c.role = 1;
        for (int j = J - 1; j >= 0; j--) {
            if (cf.rb.axes[j] != 0) {
                t3_wait(c, T3_NB, J - j);
                const TPZ nj = t3_take(c, T3_ST + j);
                const int ax = abs(cf.rb.axes[j]) - 1;
                u[j] = c.comb3(elem(w, nj, ax), 1.0, view(w, c.qdda(j)), cf.rb.armature[j], view(w, c.qd(j)), cf.rb.damping[j]);
                if (w_lane0(c)) c.mb[T3_U + j] = u[j].id - c.L.idS;
            }
            t3_signal(c, T3_UB, J - j);
        }
    } else {
        (void)xch_await(c, XK_MISC, 1);   // the helper has read the last F_i
    }
    c.bar();   // (B) the n-recursion is through, nobody reads N_i, F_i, com x F_i any more
    c.phase(4);
    if (c.wid == 1) {
        for (int i = J - n_tail; i < J; i++) c.freeVs(t3_take(c, T3_N + i));
    } else if (c.wid == 3) {
        for (int i = 0; i < J - 1; i++) c.freeVs(t3_take(c, T3_X2 + i));
        c.freeVs(t3_take(c, T3_C2 + J - 1));
    } else if (c.wid == 2) {
        for (int i = 0; i < J; i++) { if (i < J - n_tail) c.freeVs(t3_take(c, T3_N + i)); c.freeVs(t3_take(c, T3_F + i)); }
    }
}
// ... and the helper block's side of it: waves 1 + 2 the f-recursion as a pair, wave 0 the cross products p x (R f).  (Wave 3 is in its
// forward kinematics: no block barrier in here -- the three waves meet through the mailbox.)
template <class CH>
__device__ inline void helper_backward(CH& c) {
    typedef typename CH::PZT TPZ;
    const P1Cfg& cf = *c.cf;
    auto& w = c.w;
    const int J = c.J;
    LDS_AS int* px = c.mb + kHelpBase;
    LDS_AS int* fvx = px + 2 * pzw::PW_WORDS;
    if (c.wid == 3) return;
    (void)xch_await(c, XK_MISC, 0);          // the main block has taken every product: this wave's part of the pool is free again
    c.freeV |= c.part_mask(c.wid);
    if (c.wid == 0) {
        c.role = 0;
        for (int i = J - 1; i >= 0; i--) {
            t3_wait(c, T3_B1, J - i);
            TPZ c2 = c.crossMatPz(&cf.rb.trans[3 * (i + 1)], t3_take(c, T3_A2 + i));
            if (i < J - 1) xch_publish(c, XK_C2, i, c2);   // (the last joint's is p x (R 0): the main block builds that constant itself instead of waiting for this one)
        }
        return;
    }
    // the pair: wave 1 leads (its part of the pool), wave 2 follows
    if (c.wid == 1) {
        if (w.lane < 2 * pzw::PW_WORDS) px[w.lane] = 0;
        if (w.lane == 0) { fvx[2] = (int)(unsigned)(c.freeV & 0xffffffffull); fvx[3] = (int)(unsigned)(c.freeV >> 32); }
        t3_signal(c, T3_B2, 1);
    } else {
        t3_wait(c, T3_B2, 1);
        const unsigned long long pf = c.part_mask(1);
        const unsigned long long fv = (unsigned long long)(unsigned)t3_ld(&fvx[2]) | ((unsigned long long)(unsigned)t3_ld(&fvx[3]) << 32);
        c.freeV = (c.freeV & ~pf) | (fv & pf);
    }
    c.pair_begin(1, px);
    const bool h0 = w.half == 0;
    c.role = 1;
    TPZ f = c.allocV();
    if (h0) set_const(w, f, nullptr, nullptr);
    psync(w);
    TPZ Fs[ARMOUR_MAX_JOINTS];
    for (int i = J - 1; i >= 0; i--) Fs[i] = xch_take(c, XK_F, i);   // (both halves; all of them are there by now: one wait, one invalidate each, off the recursion's chain)
    psync(w);
    for (int i = J - 1; i >= 0; i--) {
        const TPZ Rn = c.R(i + 1);
        const TPZ Fi = Fs[i];
        TPZ a2 = c.mulMV(Rn, f);
        if (h0) { t3_post(c, T3_A2 + i, a2); t3_signal(c, T3_B1, J - i); }
        TPZ f2 = c.add(a2, Fi); c.freeVs(f); c.freeVs(Fi);
        f = f2;
    }
    c.freeVs(f);
    if (h0) xch_raise(c, XK_MISC, 1);
    c.pair_end();
    c.role = c.wid;
}
// The helper block's item: the two velocity recursions and the forward kinematics on wave 3, one family of products on each of the others.
template <class CH>
__device__ PZW_NOINLINE void run_helper_free(CH& c, int b, int t_lane, bool publish) {
    PZ_KEEP_RETURN_ADDRESS();
    typedef typename CH::PZT TPZ;
    const P1Cfg& cf = *c.cf;
    auto& w = c.w;
    const int J = c.J;
    if (threadIdx.x < T3_NCNT) c.mb[T3_CNT + threadIdx.x] = 0;
    if (c.wid == 3) {
        c.role = 3;
        TPZ wv = c.allocV(); set_const(w, wv, nullptr, nullptr); t3_post(c, T3_ST + 0, wv);
        TPZ waux = c.allocV(); set_const(w, waux, nullptr, nullptr); t3_post(c, T3_ST + 2, waux);
    }
    c.bar();
    const bool c4h = c.two_level >= 3;   // R_t w_aux_s x (qd_s e_axis), the cross product of the angular-acceleration step (RT/Dynamics.cu:119-121), as a fourth family
    if (c.wid == 3) {
        c.role = 3;
        if (publish) {
            int freed_na = 0;   // (c4h) R_t w_aux_k, k < freed_na, have been given back
            for (int s = 0; s < J; s++) {   // w_{s+1} = R_t w_s + qd_s, w_aux_{s+1} = R_t w_aux_s + qda_s: as the main block's fourth wave runs them (run_rnea_free)
                if (c.late_jrs && s == 4) t3_wait(c, T3_JR2, 1);
                const int ax = abs(cf.rb.axes[s]) - 1;
                while (c4h && freed_na + 3 < s) {   // at most four of them alive: read by the second product wave at its step k
                    t3_wait(c, T3_C1, freed_na + 1);
                    if (cf.rb.axes[freed_na] != 0) c.freeVs(t3_take(c, T3_NA + freed_na));
                    freed_na++;
                }
                TPZ na = c.mulMV(c.Rt(s), t3_take(c, T3_ST + 3 * s + 2));   // (first: the angular step of the main block waits for what is made of it)
                if (c4h) t3_post(c, T3_NA + s, na);
                if (cf.rb.axes[s] != 0) { TPZ t2 = c.addOneDim(na, c.qda(s), ax); if (!c4h) c.freeVs(na); na = t2; }
                t3_post(c, T3_ST + 3 * (s + 1) + 2, na);
                if (c4h) t3_signal(c, T3_C3, s + 1);   // R_t w_aux_s is posted
                TPZ nw = c.mulMV(c.Rt(s), t3_take(c, T3_ST + 3 * s));
                if (cf.rb.axes[s] != 0) { TPZ t2 = c.addOneDim(nw, c.qd(s), ax); c.freeVs(nw); nw = t2; }
                t3_post(c, T3_ST + 3 * (s + 1), nw);
                t3_signal(c, T3_CA, s + 1);
            }
            if (c4h) {
                for (; freed_na < J; freed_na++) { t3_wait(c, T3_C1, freed_na + 1); if (cf.rb.axes[freed_na] != 0) c.freeVs(t3_take(c, T3_NA + freed_na)); }
            }
        }
        if (c.late_jrs) t3_wait(c, T3_JR2, 1);   // (a chain of four joints or fewer never waited above)
        FkStateT<TPZ> fk;
        fk_begin(c, fk);
        for (int s = 0; s < J; s++) fk_step(c, fk, s, b, t_lane);
        c.freeVs(fk.T);
    } else if (publish) {
        c.role = c.wid;
        if (c.late_jrs && c.wid == 2) { jrs_late_joints(c, c.jl, 4, true); t3_signal(c, T3_JR2, 1); }   // (this wave's products read no JRS slot, and the main block takes them last)
        const int s0 = c.wid == 0 ? 0 : 1, s1 = c.wid == 0 ? J - 1 : J;
        for (int s = s0; s <= s1; s++) {
            if (c.late_jrs && c.wid == 1 && s == 4) t3_wait(c, T3_JR2, 1);   // (qd_4 for the angular product, I_4 for the moment's)
            if (c4h && c.wid == 1) {   // the angular step s - 1 of the main block waits for this one: before the moment's product of link s - 1
                const int q = s - 1;
                t3_wait(c, T3_C3, q + 1);
                if (cf.rb.axes[q] != 0) {
                    TPZ temp = c.embedOneDim(c.qd(q), abs(cf.rb.axes[q]) - 1);
                    TPZ c4 = c.crossPzPz(t3_take(c, T3_NA + q), temp); c.freeVs(temp);
                    xch_publish(c, XK_C4, q, c4);
                }
                t3_signal(c, T3_C1, q + 1);   // (the fourth wave may give R_t w_aux_q back)
            }
            t3_wait(c, T3_CA, s);
            const TPZ wv = t3_take(c, T3_ST + 3 * s), waux = t3_take(c, T3_ST + 3 * s + 2);
            if (c.wid == 0) {
                TPZ c2 = c.crossPzMat(waux, &cf.rb.trans[3 * s]);
                TPZ c3 = c.crossPzPz(wv, c2); c.freeVs(c2);
                xch_publish(c, XK_C3L, s, c3);
            } else if (c.wid == 2) {
                TPZ c2 = c.crossPzMat(waux, &cf.rb.com[3 * (s - 1)]);
                TPZ c3 = c.crossPzPz(wv, c2); c.freeVs(c2);
                xch_publish(c, XK_C3C, s - 1, c3);
            } else {
                TPZ t2 = c.mulMV(c.inertia(s - 1), wv);
                TPZ cr = c.crossPzPz(waux, t2); c.freeVs(t2);
                xch_publish(c, XK_CRN, s - 1, cr);
            }
        }
    }
    c.prof_forward_done();
    c.phase(3);
    if (publish && c.two_level >= 3 && cf.step_pairs != 0 && cf.lean_back != 0) helper_backward(c);
    c.phase(4);
    c.bar();
}

template <class CH>
__device__ PZW_NOINLINE void run_rnea_free(CH& c, typename CH::PZT* u, int b, int t_lane) {
    PZ_KEEP_RETURN_ADDRESS();
    typedef typename CH::PZT TPZ;
    const P1Cfg& cf = *c.cf;
    auto& w = c.w;
    const int J = c.J;
    const bool fk_wave = c.nw == 4;                       // the forward kinematics has a wave of its own
    constexpr bool fused_cross = CH::kFusedCross;         // (time-vectorised chain) wdot x p and wdot x com are taken inside the sums that add them
    // The w_aux recursion w_aux_{s+1} = R_t w_aux_s + qda_s depends on nothing but itself (like omega): in four-wave blocks of the time-vectorised
    // chain it runs on the forward-kinematics wave next to omega, which hands the angular wave R_t w_aux_s for the one product of wdot's step
    // that needs it, and does its forward kinematics -- which nobody waits for -- behind the recursions.  The angular wave's step, the chain the
    // forward pass hangs on, is then three operators long instead of five.  (Measured: B = 128 9.77 -> 9.64 ms, 64: 8.86 -> 8.62, 16: 7.72 -> 7.52;
    // with the forward kinematics still interleaved the same move is SLOWER, 10.2 ms: the angular wave then waits for a wave that is busy.)
    const bool aux3 = fk_wave && cf.tv_aux_on_fk_wave != 0;
    const bool with_fk = cf.fk_items == 0 && !fk_wave;   // ... otherwise wave 2 runs it (unless other blocks do)
    constexpr int K = 3;  // joints a producer may run ahead of the slowest reader of its results
    const int wave_w = fk_wave ? 3 : 2;   // the wave that runs the omega recursion (see below)
    // Four-wave blocks (release-build stamps: forward pass done after 7.4 / 8.8 / 10.4 / 12.8 M cycles on the forward-kinematics, angular,
    // linear-acceleration and F / N waves): the angular wave builds the moments of the last two links once its recursion is through, the
    // forward-kinematics wave the two constant cross products of every linear-acceleration step, and the F / N wave does everything that
    // needs the state alone before it waits for the linear acceleration.  (Each of the three alone changes nothing: the pass ends with
    // F of the last link, which needs the last linear acceleration.)
    // (at most K + 1 of them: the states the tail reads must still be alive when the recursion ends -- it gives state k back K joints later)
    const int n_tail_max = J >= 5 ? 4 : J >= 4 ? 3 : 0;
    const int n_tail = fk_wave && cf.tail_cross >= 100 ? min(cf.tail_cross / 100 - 1, n_tail_max) : n_tail_max;   // (development: 100 * (n + 1) in cf.tail_cross holds the tail to n links)   (three waves: the F / N wave, which also carries omega there, is the last to finish the forward pass as well)
    // Round 4: the F / N wave is still the last one through the forward pass (per-step kernel, slowest step of a lone problem: the four waves are
    // through after 0.89 / 1.26 / 1.30 / 1.48 M cycles; time-vectorised blocks of a 128-problem batch: 9.5 / 8.1 / 9.2 / 10.1 M).  Of what it does for a
    // link, w x (w_aux x com) needs neither the linear acceleration nor anything of the F / N wave's own: for the last `tc_links` links (whose
    // states are still alive when the recursions end) a wave that is through with its recursion builds it -- wave `tc_wave` = 3 (the fourth
    // wave, when it has no forward kinematics to do) or 1 (the angular wave, before its tail moments) -- and the F / N wave takes it from the
    // mailbox.  Same operators on the same operands: same bits.  (cf.tail_cross: 0 off | links | 10 + links: on wave 1)
    // a time step on two CUs: the three families of velocity-only cross products come from the item's helper block (xch_take)
    bool two = false;
    int two_lvl = 0;
    if constexpr (CH::kTwoCu) { two = c.two_cu; two_lvl = two ? c.two_level : 0; }
    const bool two2 = two_lvl >= 2;   // nobody in this block reads w: no w recursion here; the linear-acceleration wave takes wdot x p itself
    const bool two3 = two_lvl >= 3;   // ... nor w_aux: the angular step's cross product comes from the helper too, the fourth wave has no recursion left
    bool late_jrs = false;            // the JRS of joints 4 .. J - 1 is built by the fourth wave while the others run their first steps: they wait for it before step 4
    if constexpr (CH::kTwoCu) late_jrs = two3 && c.late_jrs;
    auto jrs_ready = [&](int s) { if (late_jrs && s == 4) t3_wait(c, T3_JR2, 1); };
    bool lean_back = false;           // run_backward_remote: level 3 with paired waves -- the f-recursion of the backward pass on the helper block (the fourth wave builds com x F_i in this pass)
    if constexpr (CH::kPairs) lean_back = two3 && cf.step_pairs != 0 && cf.lean_back != 0;
    auto take_x = [&](int kind, int idx) -> TPZ { if constexpr (CH::kTwoCu) return xch_take(c, kind, idx); else return TPZ(); };
    const int tc_opt = two ? 0 : cf.tail_cross % 100;
    const int tc_wave = fk_wave && aux3 && tc_opt > 0 ? (tc_opt >= 10 ? 1 : 3) : -1;
    const int tc_links = tc_wave < 0 ? 0 : min(tc_opt % 10, min(K + 1, n_tail_max));   // (never more links than the tail may hold: a chain of J < 4 joints has none -- s = J - tc_links + 1 must stay >= 1)
    auto tail_cross = [&]() {   // (the wave that owns them frees them behind barrier (A))
        for (int s = J - tc_links + 1; s <= J; s++) {
            if (tc_wave == 1) t3_wait(c, T3_CA, s);   // (w_s, w_aux_s are the fourth wave's)
            const TPZ wv = t3_take(c, T3_ST + 3 * s), waux = t3_take(c, T3_ST + 3 * s + 2);
            TPZ c2 = c.crossPzMat(waux, &cf.rb.com[3 * (s - 1)]);
            TPZ c3 = c.crossPzPz(wv, c2); c.freeVs(c2);
            t3_post(c, T3_X3 + s, c3);
            t3_signal(c, T3_C3P, s);
        }
    };
    auto tail_cross_free = [&]() { for (int s = J - tc_links + 1; s <= J; s++) c.freeVs(t3_take(c, T3_X3 + s)); };
    if (threadIdx.x < T3_NCNT) c.mb[T3_CNT + threadIdx.x] = 0;
    // walk helpers (four-wave blocks of the time-vectorised kernel, backward pass): both channels start empty, every wave's job count at 0
    bool walk_helpers = CH::kWalkHelpers && fk_wave && cf.tv_walk_helpers != 0;
    if constexpr (CH::kWalkHelpers) {
        if (c.w.hded) walk_helpers = false;   // eight-wave blocks: every role wave has a helper of its own, the idle waves of this pass stay idle
        else {
            if (threadIdx.x < 2 * tv::HJ_WORDS) c.mb[kHelpBase + threadIdx.x] = 0;
            c.w.hseq = 0; c.w.hch = nullptr; c.w.hnum = cf.tv_walk_helpers > 1 ? cf.tv_walk_helpers : 16; c.w.hmin = cf.tv_help_min;
        }
    }
    if (c.wid == 1) {
        c.role = 1;
        TPZ wdot = c.allocV();
        set_const(w, wdot, nullptr, nullptr);
        t3_post(c, T3_ST + 1, wdot);
        if (!aux3) { TPZ waux = c.allocV(); set_const(w, waux, nullptr, nullptr); t3_post(c, T3_ST + 2, waux); }
    }
    if (c.wid == wave_w) {
        c.role = wave_w;
        TPZ wv = c.allocV();
        set_const(w, wv, nullptr, nullptr);
        t3_post(c, T3_ST + 0, wv);
        if (aux3) { TPZ waux = c.allocV(); set_const(w, waux, nullptr, nullptr); t3_post(c, T3_ST + 2, waux); }   // (w_aux belongs to the wave that runs its recursion)
    }
    if (c.wid == 0) {
        c.role = 0;
        TPZ lacc = c.allocV();
        double g[3] = {0.0, 0.0, cf.rb.gravity};
        set_const(w, lacc, g, nullptr);
        t3_post(c, T3_LA + 0, lacc);
    }
    c.bar();
    // ---------------- forward: state_k / lacc_k = state and linear acceleration of joint k-1 (k = 0: the base)
    // The angular velocity w_{s+1} = R_t w_s + qd_s depends on nothing but itself, and neither w_aux nor wdot reads it: it runs
    // on a wave with time to spare (wave 2 of three, the forward-kinematics wave of four), which owns its slots.
    auto omega_step = [&](int s, int& freed_w) {
        while (freed_w + K < s + 1) {   // w_k: read by wave 0 at step k (< J), by wave 2 at step k (>= 1)
            const int k = freed_w;
            if (k < J) t3_wait(c, T3_C0, k + 1);
            if (k >= 1) t3_wait(c, T3_CC2, k);
            c.freeVs(t3_take(c, T3_ST + 3 * k));
            if (aux3) c.freeVs(t3_take(c, T3_ST + 3 * k + 2));   // (w_aux_k: the same readers)
            freed_w++;
        }
        const TPZ wv = t3_take(c, T3_ST + 3 * s);
        TPZ nw = c.mulMV(c.Rt(s), wv);
        if (cf.rb.axes[s] != 0) { TPZ t2 = c.addOneDim(nw, c.qd(s), abs(cf.rb.axes[s]) - 1); c.freeVs(nw); nw = t2; }
        t3_post(c, T3_ST + 3 * (s + 1), nw);
        if (!aux3) t3_signal(c, T3_CA, s + 1);   // (with w_aux on this wave: raised once w_aux_{s+1} is posted as well)
    };
    if (c.wid == 1) {
        c.role = 1;
        int freed = 0;   // states [0, freed) have been given back
        int freed_w1 = 0;
        for (int s = 0; s < J; s++) {
            while (freed + K < s + 1) {   // (wdot, w_aux)_k, k = freed: read by wave 0 at step k (< J), by wave 2 at step k (>= 1), by this wave at step k (done)
                const int k = freed;
                if (k < J) t3_wait(c, T3_C0, k + 1);
                if (k >= 1) t3_wait(c, T3_CC2, k);
                for (int e = 1; e < (aux3 ? 2 : 3); e++) c.freeVs(t3_take(c, T3_ST + 3 * k + e));
                freed++;
            }
            if (wave_w == 1) omega_step(s, freed_w1);
            jrs_ready(s);
            const TPZ wdot = t3_take(c, T3_ST + 3 * s + 1);
            const TPZ Rt = c.Rt(s);
            const int ax = abs(cf.rb.axes[s]) - 1;
            TPZ na = wdot;
            if (!aux3) na = c.mulMV(Rt, t3_take(c, T3_ST + 3 * s + 2));
            TPZ nd = c.mulMV(Rt, wdot);
            if (cf.rb.axes[s] != 0) {
                TPZ c4;
                if (two3) c4 = take_x(XK_C4, s);
                else {
                    TPZ temp = c.embedOneDim(c.qd(s), ax);   // addOneDimPZ(0, qd_s, axis) (RT/Dynamics.cu:119-120)
                    if (aux3) { t3_wait(c, T3_CA, s + 1); na = t3_take(c, T3_NA + s); }   // R_t w_aux_s from the forward-kinematics wave
                    c4 = c.crossPzPz(na, temp); c.freeVs(temp);
                }
                TPZ nd2 = c.sum3(nd, c4, c.qdda(s), ax); c.freeVs(c4); c.freeVs(nd); nd = nd2;
                if (!aux3) { TPZ na2 = c.addOneDim(na, c.qda(s), ax); c.freeVs(na); na = na2; }
            }
            t3_post(c, T3_ST + 3 * (s + 1) + 1, nd);
            if (!aux3) t3_post(c, T3_ST + 3 * (s + 1) + 2, na);
            t3_signal(c, T3_C1, s + 1);
        }
        if (tc_wave == 1) tail_cross();   // (first: the F / N wave is waiting for these)
        for (int s = J; s > J - n_tail; s--) {   // N = I * wdot + cross(w_aux, I * w) of link s - 1
            if (!two2) t3_wait(c, T3_CA, s);
            const TPZ wv = t3_take(c, T3_ST + 3 * s), wdot = t3_take(c, T3_ST + 3 * s + 1), waux = t3_take(c, T3_ST + 3 * s + 2);
            const TPZ I = c.inertia(s - 1);
            TPZ t1 = c.mulMV(I, wdot);
            TPZ cr;
            if (two) cr = take_x(XK_CRN, s - 1);
            else { TPZ t2 = c.mulMV(I, wv); cr = c.crossPzPz(waux, t2); c.freeVs(t2); }
            TPZ N = c.add(t1, cr); c.freeVs(t1); c.freeVs(cr);
            t3_post(c, T3_N + s - 1, N);
        }
        c.prof_forward_done(); c.bar();   // (A) the forward pass is over everywhere
        if (tc_wave == 1) tail_cross_free();
        for (int k = freed; k <= J; k++)
            for (int e = 1; e < (aux3 ? 2 : 3); e++) c.freeVs(t3_take(c, T3_ST + 3 * k + e));
        if (wave_w == 1) for (int k = freed_w1; k <= J; k++) c.freeVs(t3_take(c, T3_ST + 3 * k));
    } else if (c.wid == 0) {
        c.role = 0;
        int freed = 0;
        for (int s = 0; s < J; s++) {
            jrs_ready(s);
            t3_wait(c, T3_C1, s);   // state_s
            if (!two2) t3_wait(c, T3_CA, s);
            while (freed + K < s + 1) {   // lacc_k: read by wave 2 at step k (>= 1), by this wave at step k (done)
                const int k = freed;
                if (k >= 1) t3_wait(c, T3_CC2, k);
                c.freeVs(t3_take(c, T3_LA + k));
                freed++;
            }
            const TPZ wv = t3_take(c, T3_ST + 3 * s), wdot = t3_take(c, T3_ST + 3 * s + 1), waux = t3_take(c, T3_ST + 3 * s + 2), lacc = t3_take(c, T3_LA + s);
            const double* tr = &cf.rb.trans[3 * s];
            TPZ c1 = wdot, c2 = waux;
            const bool own_cross = !fk_wave || (two2 && !two3);   // (two CUs, level 2: wdot x p is this wave's own -- the fourth wave's copy made it wait for a wave that waits for the angular one; level 3: the fourth wave has nothing else to do and builds it the moment wdot_s is posted)
            if (!own_cross) {   // wdot x p, w_aux x p: from the forward-kinematics wave (wdot x p only where it is a PZ of its own)
                t3_wait(c, T3_C3, s + 1);
                if constexpr (!fused_cross) c1 = t3_take(c, T3_X1 + s);
                if (!two) c2 = t3_take(c, T3_X2 + s);
            } else {
                if constexpr (!fused_cross) c1 = c.crossPzMat(wdot, tr);
                if (!two) c2 = c.crossPzMat(waux, tr);
            }
            TPZ c3 = two ? take_x(XK_C3L, s) : c.crossPzPz(wv, c2); if (own_cross && !two) c.freeVs(c2);
            TPZ s2;
            if constexpr (fused_cross) s2 = c.sum3x(lacc, wdot, tr, c3);   // (lacc + wdot x p) + c3, the cross product inside the sum
            else { s2 = c.sum3(lacc, c1, c3); if (own_cross) c.freeVs(c1); }
            c.freeVs(c3);
            TPZ nl = c.mulMV(c.Rt(s), s2); c.freeVs(s2);
            t3_post(c, T3_LA + s + 1, nl);
            t3_signal(c, T3_C0, s + 1);
        }
        c.prof_forward_done(); c.bar();   // (A)
        c.phase(3);
        for (int k = freed; k <= J; k++) c.freeVs(t3_take(c, T3_LA + k));
    } else if (c.wid == 3 && two2) {
        // Two CUs, level 2: of this wave's forward pass only the w_aux recursion is left (R_t w_aux_s for the angular step); level 3: nothing.
        c.role = 3;
        if (!two3) {
            int freed_na = 0;
            for (int s = 0; s < J; s++) {
                while (freed_na + 1 < s) {   // R_t w_aux_k: read by the angular wave at step k
                    t3_wait(c, T3_C1, freed_na + 1);
                    if (cf.rb.axes[freed_na] != 0) c.freeVs(t3_take(c, T3_NA + freed_na));
                    freed_na++;
                }
                const TPZ waux = t3_take(c, T3_ST + 3 * s + 2);
                TPZ na = c.mulMV(c.Rt(s), waux); c.freeVs(waux);   // (w_aux_s has no other reader left)
                t3_post(c, T3_NA + s, na);
                if (cf.rb.axes[s] != 0) na = c.addOneDim(na, c.qda(s), abs(cf.rb.axes[s]) - 1);
                t3_post(c, T3_ST + 3 * (s + 1) + 2, na);
                t3_signal(c, T3_CA, s + 1);
            }
            c.prof_forward_done(); c.bar();   // (A)
            c.freeVs(t3_take(c, T3_ST + 3 * J + 2));
            for (int k = freed_na; k < J; k++) if (cf.rb.axes[k] != 0) c.freeVs(t3_take(c, T3_NA + k));
        } else {
            c.freeVs(t3_take(c, T3_ST + 2));
            if constexpr (CH::kTwoCu) {
                if (late_jrs) { jrs_late_joints(c, c.jl, 4, false); t3_signal(c, T3_JR2, 1); }
            }
            if constexpr (!fused_cross) {
                int pc = 0;   // (lean backward pass) com x F_i, i < pc, are built
                auto com_cross = [&](bool all) {
                    while (lean_back && pc < J - 1) {
                        if (t3_ld(&c.mb[T3_CC2]) < pc + 1) { if (!all) return; t3_wait(c, T3_CC2, pc + 1); }
                        TPZ c1 = c.crossMatPz(&cf.rb.com[3 * pc], t3_take(c, T3_F + pc));
                        t3_post(c, T3_X2 + pc, c1);
                        pc++;
                    }
                };
                for (int s = 0; s < J; s++) {   // wdot x p for the linear-acceleration step s
                    if (s >= 2) { t3_wait(c, T3_C0, s - 1); c.freeVs(t3_take(c, T3_X1 + s - 2)); }
                    while (t3_ld(&c.mb[T3_C1]) < s) { com_cross(false); __builtin_amdgcn_s_sleep(8); }
                    t3_wait(c, T3_C1, s);
                    TPZ x1 = c.crossPzMat(t3_take(c, T3_ST + 3 * s + 1), &cf.rb.trans[3 * s]);
                    t3_post(c, T3_X1 + s, x1);
                    t3_signal(c, T3_C3, s + 1);
                }
                com_cross(true);
            }
            c.prof_forward_done(); c.bar();   // (A)
            if constexpr (!fused_cross) for (int s = J >= 2 ? J - 2 : 0; s < J; s++) c.freeVs(t3_take(c, T3_X1 + s));
        }
        c.freeVs(t3_take(c, T3_ST + 0));   // (the constant w_0 this wave posted before the pass)
    } else if (c.wid == 3) {
        // the forward kinematics shares nothing with the recursion but the JRS rotations: wave 2 was the last to finish the
        // forward pass while it carried it (14.1 M cycles against 9.4 / 10.1 M of the other two)
        c.role = 3;
        const bool w3_fk = cf.fk_items == 0;   // (the per-step kernel issues the forward kinematics of a lone problem as items of their own)
        FkStateT<TPZ> fk;
        if (w3_fk) fk_begin(c, fk);
        int freed_w = 0;
        int freed_na = 0;   // (aux3) R_t w_aux_k, k < freed_na, have been given back
        for (int s = 0; s < J; s++) {
            omega_step(s, freed_w);   // first: wave 0 waits for it
            if (aux3) {   // the w_aux recursion: R_t w_aux_s for the angular wave's step s, then w_aux_{s+1} = R_t w_aux_s + qda_s
                while (freed_na + 1 < s) {   // R_t w_aux_k: read by the angular wave at step k
                    t3_wait(c, T3_C1, freed_na + 1);
                    if (cf.rb.axes[freed_na] != 0) c.freeVs(t3_take(c, T3_NA + freed_na));   // (a fixed joint: it IS w_aux_{k+1})
                    freed_na++;
                }
                const TPZ waux = t3_take(c, T3_ST + 3 * s + 2);
                TPZ na = c.mulMV(c.Rt(s), waux);
                t3_post(c, T3_NA + s, na);
                if (cf.rb.axes[s] != 0) na = c.addOneDim(na, c.qda(s), abs(cf.rb.axes[s]) - 1);
                t3_post(c, T3_ST + 3 * (s + 1) + 2, na);
                t3_signal(c, T3_CA, s + 1);   // omega_{s+1}, w_aux_{s+1} and R_t w_aux_s are posted
            }
            {   // the two cross products with the joint offset for wave 0's step s, at most two steps ahead of it
                if (s >= 2) {
                    t3_wait(c, T3_C0, s - 1);
                    if constexpr (!fused_cross) c.freeVs(t3_take(c, T3_X1 + s - 2));
                    if (!two) c.freeVs(t3_take(c, T3_X2 + s - 2));
                }
                if (!aux3) t3_wait(c, T3_C1, s);   // (w_aux_s is this wave's own otherwise)
                const double* tr = &cf.rb.trans[3 * s];
                if (!two) {   // (w_aux x p only feeds w x (w_aux x p), which the helper block builds from its own recursion)
                    TPZ x2 = c.crossPzMat(t3_take(c, T3_ST + 3 * s + 2), tr);
                    t3_post(c, T3_X2 + s, x2);
                }
                if constexpr (!fused_cross) {   // wdot x p where it is a PZ of its own
                    if (aux3) t3_wait(c, T3_C1, s);
                    TPZ x1 = c.crossPzMat(t3_take(c, T3_ST + 3 * s + 1), tr);
                    t3_post(c, T3_X1 + s, x1);
                }
                t3_signal(c, T3_C3, s + 1);
            }
            if (!aux3 && w3_fk) fk_step(c, fk, s, b, t_lane);
        }
        if (tc_wave == 3) tail_cross();
        if (aux3 && w3_fk) for (int s = 0; s < J; s++) fk_step(c, fk, s, b, t_lane);   // (behind the recursions the other waves wait for)
        if (w3_fk) c.freeVs(fk.T);
        c.prof_forward_done(); c.bar();   // (A)
        if (tc_wave == 3) tail_cross_free();
        for (int k = freed_w; k <= J; k++) { c.freeVs(t3_take(c, T3_ST + 3 * k)); if (aux3) c.freeVs(t3_take(c, T3_ST + 3 * k + 2)); }
        if (aux3) for (int k = freed_na; k < J; k++) if (cf.rb.axes[k] != 0) c.freeVs(t3_take(c, T3_NA + k));
        for (int s = J >= 2 ? J - 2 : 0; s < J; s++) { if constexpr (!fused_cross) c.freeVs(t3_take(c, T3_X1 + s)); if (!two) c.freeVs(t3_take(c, T3_X2 + s)); }
    } else {
        c.role = 2;
        FkStateT<TPZ> fk;
        if (with_fk) fk_begin(c, fk);
        int freed_w = 0;
        for (int s = 0; s <= J; s++) {
            if (s < J && wave_w == 2) omega_step(s, freed_w);   // (first: wave 0 waits for it)
            if (s < J && with_fk) fk_step(c, fk, s, b, t_lane);   // (it waits for nobody)
            jrs_ready(s);
            if (s >= 1) {
                t3_wait(c, T3_C1, s);
                if (!two2) t3_wait(c, T3_CA, s);
                const TPZ wv = t3_take(c, T3_ST + 3 * s), wdot = t3_take(c, T3_ST + 3 * s + 1), waux = t3_take(c, T3_ST + 3 * s + 2);
                if (s <= J - n_tail) {   // N = I * wdot + cross(w_aux, I * w)   (the state alone: no need to wait for the linear acceleration yet)
                    const TPZ I = c.inertia(s - 1);
                    TPZ t1 = c.mulMV(I, wdot);
                    TPZ cr;
                    if (two) cr = take_x(XK_CRN, s - 1);
                    else { TPZ t2 = c.mulMV(I, wv); cr = c.crossPzPz(waux, t2); c.freeVs(t2); }
                    TPZ N = c.add(t1, cr); c.freeVs(t1); c.freeVs(cr);
                    t3_post(c, T3_N + s - 1, N);
                }
                {   // F = m * (linear_acc + cross(wdot, com) + cross(w, cross(w_aux, com))): all but the last sum before the linear acceleration is needed
                    const double* cm = &cf.rb.com[3 * (s - 1)];
                    TPZ c1 = wdot;
                    if constexpr (!fused_cross) c1 = c.crossPzMat(wdot, cm);
                    const bool from_tail = s > J - tc_links;   // w x (w_aux x com) comes from a wave that is through with its recursion
                    TPZ c3;
                    if (from_tail) { t3_wait(c, T3_C3P, s); c3 = t3_take(c, T3_X3 + s); }
                    else if (two) c3 = take_x(XK_C3C, s - 1);
                    else { TPZ c2 = c.crossPzMat(waux, cm); c3 = c.crossPzPz(wv, c2); c.freeVs(c2); }
                    t3_wait(c, T3_C0, s);
                    const TPZ lacc = t3_take(c, T3_LA + s);
                    TPZ s2;
                    if constexpr (fused_cross) s2 = c.sum3x(lacc, wdot, cm, c3);   // (lacc + wdot x com) + c3
                    else { s2 = c.sum3(lacc, c1, c3); c.freeVs(c1); }
                    if (!from_tail) c.freeVs(c3);
                    TPZ F = c.mulSV(c.mass(s - 1), s2); c.freeVs(s2);
                    t3_post(c, T3_F + s - 1, F);
                    if constexpr (CH::kTwoCu) { if (lean_back) xch_publish(c, XK_F, s - 1, F); }   // (the helper's f-recursion reads it where it lies)
                }
                t3_signal(c, T3_CC2, s);
            }
        }
        if (with_fk) c.freeVs(fk.T);
        c.prof_forward_done(); c.bar();   // (A)
        if (wave_w == 2) for (int k = freed_w; k <= J; k++) c.freeVs(t3_take(c, T3_ST + 3 * k));
    }
    // ---------------- backward: n = N + R n + com x F + p x (R f),  f = R f + F
    // The f-recursion (wave 1) is the chain everything hangs on: R f, then f = R f + F.  The cross product p x (R f) that the
    // n-recursion needs is a side product of it and goes to a wave that has nothing to do in this pass (wave 3, or wave 2).
    const int helper = fk_wave ? 3 : 2;
    if constexpr (CH::kPairs) {
        if (fk_wave && cf.step_pairs != 0) {
            if constexpr (CH::kTwoCu) { if (lean_back) { run_backward_remote(c, u, n_tail); return; } }
            run_backward_pairs(c, u, n_tail);
            return;
        }
    }
    if (c.wid == 1) {
        c.role = 1;
        TPZ f = c.allocV();
        set_const(w, f, nullptr, nullptr);
        for (int i = J - 1; i >= 0; i--) {
            const TPZ Rn = c.R(i + 1), Fi = t3_take(c, T3_F + i);
            TPZ a2 = with_walk_helper(c, 0, walk_helpers, [&] { return c.mulMV(Rn, f); });
            t3_post(c, T3_A2 + i, a2);
            t3_signal(c, T3_B1, J - i);
            TPZ f2 = with_walk_helper(c, 0, walk_helpers, [&] { return c.add(a2, Fi); }); c.freeVs(f);
            f = f2;
        }
        c.freeVs(f);
        c.bar();   // (B) the helper has read every R f
        for (int i = 0; i < J; i++) c.freeVs(t3_take(c, T3_A2 + i));
        for (int i = J - n_tail; i < J; i++) c.freeVs(t3_take(c, T3_N + i));   // (the tail moments are this wave's)
    } else if (c.wid == 0) {
        c.role = 0;
        TPZ nn = c.allocV();
        set_const(w, nn, nullptr, nullptr);
        for (int i = J - 1; i >= 0; i--) {
            const TPZ Rn = c.R(i + 1), Fi = t3_take(c, T3_F + i), Ni = t3_take(c, T3_N + i);
            TPZ a1 = with_walk_helper(c, 1, walk_helpers && cf.tv_help_n, [&] { return c.mulMV(Rn, nn); });
            TPZ c1 = c.crossMatPz(&cf.rb.com[3 * i], Fi);
            t3_wait(c, T3_B2, J - i);
            const TPZ c2 = t3_take(c, T3_C2 + i);
            TPZ n2 = with_walk_helper(c, 1, walk_helpers && cf.tv_help_n, [&] { return c.sum4(Ni, a1, c1, c2); }); c.freeVs(a1); c.freeVs(c1); c.freeVs(nn);  // ((N + a1) + c1) + c2
            nn = n2;
            if (cf.rb.axes[i] != 0) {
                const int ax = abs(cf.rb.axes[i]) - 1;
                u[i] = c.comb3(elem(w, n2, ax), 1.0, view(w, c.qdda(i)), cf.rb.armature[i], view(w, c.qd(i)), cf.rb.damping[i]);
                if (w_lane0(c)) c.mb[T3_U + i] = u[i].id - c.L.idS;
            }
        }
        c.freeVs(nn);
        c.bar();   // (B)
        c.phase(4);
    }
    if (c.wid == helper) {
        c.role = helper;
        TPZ htmp = c.allocV();
        for (int i = J - 1; i >= 0; i--) {
            if (walk_helpers && cf.tv_help_n) serve_walk_helper(c, 1, htmp);   // R n of joint i
            t3_wait(c, T3_B1, J - i);
            TPZ c2 = c.crossMatPz(&cf.rb.trans[3 * (i + 1)], t3_take(c, T3_A2 + i));
            t3_post(c, T3_C2 + i, c2);
            t3_signal(c, T3_B2, J - i);
            if (walk_helpers && cf.tv_help_n) serve_walk_helper(c, 1, htmp);   // the four-term sum of joint i
        }
        c.freeVs(htmp);
        c.bar();   // (B) wave 0 has read every p x (R f)
        for (int i = 0; i < J; i++) c.freeVs(t3_take(c, T3_C2 + i));
        if (helper == 2) { for (int i = 0; i < J; i++) { if (i < J - n_tail) c.freeVs(t3_take(c, T3_N + i)); c.freeVs(t3_take(c, T3_F + i)); } }
    } else if (c.wid == 2) {   // (four waves: wave 3 is the helper)
        c.role = 2;
        if (walk_helpers) {   // this wave has nothing of its own in the backward pass: it walks half of the f-recursion's R f and R f + F
            TPZ htmp = c.allocV();
            for (int i = J - 1; i >= 0; i--) { serve_walk_helper(c, 0, htmp); serve_walk_helper(c, 0, htmp); }
            c.freeVs(htmp);
        }
        c.bar();   // (B) both recursions are through: nobody reads N_i, F_i any more
        for (int i = 0; i < J; i++) { if (i < J - n_tail) c.freeVs(t3_take(c, T3_N + i)); c.freeVs(t3_take(c, T3_F + i)); }
    }
}

