// MEX gateway of libarmour_hip.so for MATLAB.  Neither MATLAB nor mex.h exists in the build image, so this file is NOT
// part of `make`; build it on a MATLAB host with
//     mex armour_hip_mex.cpp -I<repo>/include -L<repo>/armour_amd/lib -larmour_hip
// In this repository it is compiled against tests/stubs/mex.h (a functional test double of the MEX C API) and driven
// command by command by tests/test_mex_gateway.py -- on the CPU for the build, the argument checks and 'traj', on the
// MI355X for everything that needs a handle.  Conventions follow the reference's only MEX precedent, kinova_robust_controllers_mex/kinova_controller.cpp:19-84:
// column vectors of doubles in, mxCreateNumericMatrix out, mexErrMsgTxt on error.  Unlike that gateway the handle is
// persistent (mexLock / mexAtExit): the reach sets built by 'set_problem' stay on the device for the 'eval' calls of
// the solver loop.
//
//   armour_hip_mex('create', T)                                    T = NUM_TIME_STEPS (RT/Parameters.h:17)
//   [margin] = armour_hip_mex('set_problem', q0, qd0, qdd0, q_des, Z)   Z = 12 x nObs, columns = obstacle zonotope Z(:); the optional output is the
//                    build's prune margin (armour_get_prune_margin: how close a simplify() verdict came to flipping; also margin = armour_hip_mex('prune_margin'))
//   armour_hip_mex('set_problem_armtd', q0, qd0, q_des, JRS, k_range, Z)   ARMTD comparison mode: JRS = T x 6 x 7
//                    (per joint the columns c_cos g_cos r_cos c_sin g_sin r_sin that KSI/uarmtd_planner.m:277-312 writes
//                    into armtd.in), k_range = 7 x 1; the commands below then follow CMP/NLPclass.cu
//   [g, jac]             = armour_hip_mex('eval', k)               g: m x 1, jac: n x m (gradient of row i in column i)
//   [h, heq, grad_h, grad_heq] = armour_hip_mex('constraints', k)  the fmincon `nonlcon` shape of KSI/uarmtd_planner.m:776-796:
//                    h <= 0 feasible, grad_h n x numel(h); rows g - g_u first, then g_l - g of the two-sided rows
//   [h, heq, grad_h, grad_heq] = armour_hip_mex('constraints', k, 'pruned')   the same with the rows that can be violated for some k only: the pruned
//                    list of KSI/uarmtd_planner.m:577-583,628-690, computed on the device (armour_get_row_relevance); rel = armour_hip_mex('relevance')
//   [x_l, x_u, g_l, g_u] = armour_hip_mex('bounds')
//   [f, grad_f]          = armour_hip_mex('cost', k)
//   [k_opt, feasible, info] = armour_hip_mex('solve')              info = [cost; iterations; evaluations; status; ms]
//   [c, G, Grest, expMat, id] = armour_hip_mex('pz', which, i, t)  reach-set PZ as CORA polyZonotope fields: which =
//                    'link' | 'torque', i and t 1-based; polyZonotope_ROAHM(c, G, Grest, expMat, id) rebuilds it with the
//                    ids 1..7 of jrs_info.k_id (PZM/create_jrs_online.m:225); same mapping as armour_amd/cora.py
//   [q, qd, qdd] = armour_hip_mex('traj', q0, qd0, qdd0, k, t)     desired trajectory of a plan at time t (the Bernstein form of
//                    KSI/uarmtd_planner.m:846-905 in the NLP's own closed form; k_range and duration of the handle's parameters)
//   v = armour_hip_mex('violations', k)                            reduced outputs (armour_eval_violations): 6 x 1 =
//                    [l1_violation; worst; worst_row (1-based, 0 = none); n_violated; n_outside_slack; feasible] -- g stays on the device
//   armour_hip_mex('destroy')
//
// Several GPUs from the one MATLAB thread (armour_batch_*: one handle, stream and host thread per device slot; problems dealt in
// contiguous blocks):
//   armour_hip_mex('batch_create', T, devices)                     devices: vector of HIP device ordinals (one may repeat)
//   armour_hip_mex('batch_set_problems', Q0, QD0, QDD0, Q_DES, Z)  7 x B each, Z = (12 nObs) x B: column b = the obstacles of problem b
//   [K_opt, feasible, info] = armour_hip_mex('batch_solve'[, max_wall_time_s])   7 x B, 1 x B, 5 x B (rows as 'solve')
//   V = armour_hip_mex('batch_violations', K)                      K 7 x B -> 6 x B (rows as 'violations')
//   [G, JAC] = armour_hip_mex('batch_eval', K)                     m x B and (n m) x B: column b = g / the row-major Jacobian of problem b
//   armour_hip_mex('batch_destroy')
#include <string.h>

#include "armour_hip.h"
#include "mex.h"

static ArmourPlanner* g_h = nullptr;
static ArmourBatch* g_bt = nullptr;
static ArmourParams g_pr;

static void cleanup(void) {
    if (g_h) { armour_destroy(g_h); g_h = nullptr; }
    if (g_bt) { armour_batch_destroy(g_bt); g_bt = nullptr; }
}
static void violation_column(const ArmourViolation& v, double* p) {
    p[0] = v.l1_violation; p[1] = v.worst; p[2] = v.worst_row + 1; p[3] = v.n_violated; p[4] = v.n_outside_slack; p[5] = v.feasible;
}
static void chk(int rc) {
    if (rc < 0) mexErrMsgTxt(armour_last_error());
}
static void need(bool ok, const char* msg) {
    if (!ok) mexErrMsgTxt(msg);
}
static mxArray* col(mwSize n) { return mxCreateNumericMatrix(n, 1, mxDOUBLE_CLASS, mxREAL); }

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    char cmd[32];
    need(nrhs >= 1 && !mxGetString(prhs[0], cmd, sizeof(cmd)), "usage: armour_hip_mex(command, ...)");
    if (!strcmp(cmd, "create")) {
        need(nrhs >= 2, "create needs the number of time steps");
        ArmourRobot rb;
        ArmourParams& pr = g_pr;
        armour_robot_kinova_gen3_no_gripper(&rb);
        armour_params_default(&pr, (int)mxGetScalar(prhs[1]));
        // armour_hip_mex('create', T, turn_off_input_constraints): the reference's compile-time switch TURN_OFF_INPUT_CONSTRAINTS (RT/Parameters.h:46-47)
        if (nrhs >= 3) pr.input_constraints_off = mxGetScalar(prhs[2]) != 0.0 ? 1 : 0;
        if (g_h) { armour_destroy(g_h); g_h = nullptr; }
        chk(armour_create(&rb, &pr, nullptr, 0, &g_h));
        if (!mexIsLocked()) { mexAtExit(cleanup); mexLock(); }
        return;
    }
    if (!strcmp(cmd, "destroy")) {
        if (g_h) { armour_destroy(g_h); g_h = nullptr; }
        if (!g_bt && mexIsLocked()) mexUnlock();
        return;
    }
    // ---- the multi-device batch (its own object next to the single handle)
    if (!strcmp(cmd, "batch_create")) {
        need(nrhs >= 3 && mxGetNumberOfElements(prhs[2]) >= 1 && mxGetNumberOfElements(prhs[2]) <= 64, "batch_create needs T and a vector of device ordinals");
        ArmourRobot rb;
        ArmourParams pr;
        armour_robot_kinova_gen3_no_gripper(&rb);
        armour_params_default(&pr, (int)mxGetScalar(prhs[1]));
        if (nrhs >= 4) pr.input_constraints_off = mxGetScalar(prhs[3]) != 0.0 ? 1 : 0;   // ('batch_create', T, devices, turn_off_input_constraints)
        int32_t dev[64];
        const int nd = (int)mxGetNumberOfElements(prhs[2]);
        for (int i = 0; i < nd; i++) dev[i] = (int32_t)mxGetPr(prhs[2])[i];
        if (g_bt) { armour_batch_destroy(g_bt); g_bt = nullptr; }
        chk(armour_batch_create(&rb, &pr, nullptr, dev, nd, &g_bt));
        if (!mexIsLocked()) { mexAtExit(cleanup); mexLock(); }
        return;
    }
    if (!strcmp(cmd, "batch_destroy")) {
        if (g_bt) { armour_batch_destroy(g_bt); g_bt = nullptr; }
        if (!g_h && mexIsLocked()) mexUnlock();
        return;
    }
    if (!strncmp(cmd, "batch_", 6)) {
        need(g_bt != nullptr, "call armour_hip_mex('batch_create', T, devices) first");
        if (!strcmp(cmd, "batch_set_problems")) {
            need(nrhs == 6, "batch_set_problems needs Q0, QD0, QDD0, Q_DES (7 x B) and Z ((12 nObs) x B)");
            const int B = (int)(mxGetNumberOfElements(prhs[1]) / 7);
            for (int i = 1; i <= 4; i++) need((int)mxGetNumberOfElements(prhs[i]) == 7 * B && B >= 1, "state matrices must be 7 x B");
            const size_t nz = mxGetNumberOfElements(prhs[5]);
            need(nz % ((size_t)12 * B) == 0, "Z must be (12 nObs) x B");
            chk(armour_batch_set_problems(g_bt, B, (int)(nz / 12 / B), mxGetPr(prhs[1]), mxGetPr(prhs[2]), mxGetPr(prhs[3]), mxGetPr(prhs[4]), mxGetPr(prhs[5])));
            return;
        }
        int B, n, m, ns;
        chk(armour_batch_get_sizes(g_bt, &B, &n, &m, &ns));
        if (!strcmp(cmd, "batch_solve")) {
            ArmourSolveOptions so;
            armour_solve_options_default(&so);
            if (nrhs > 1) so.max_wall_time_s = mxGetScalar(prhs[1]);
            ArmourSolveResult* r = (ArmourSolveResult*)mxMalloc(sizeof(ArmourSolveResult) * B);
            chk(armour_batch_solve(g_bt, &so, r));
            plhs[0] = mxCreateDoubleMatrix(n, B, mxREAL);
            mxArray* feas = mxCreateDoubleMatrix(1, B, mxREAL);
            mxArray* info = mxCreateDoubleMatrix(5, B, mxREAL);
            for (int b = 0; b < B; b++) {
                memcpy(mxGetPr(plhs[0]) + (size_t)b * n, r[b].k_opt, n * sizeof(double));
                mxGetPr(feas)[b] = r[b].feasible != 0;
                double* p = mxGetPr(info) + (size_t)b * 5;
                p[0] = r[b].cost; p[1] = r[b].iterations; p[2] = r[b].evaluations; p[3] = r[b].status; p[4] = r[b].time_ms;
            }
            mxFree(r);
            if (nlhs > 1) plhs[1] = feas; else mxDestroyArray(feas);
            if (nlhs > 2) plhs[2] = info; else mxDestroyArray(info);
        } else if (!strcmp(cmd, "batch_violations")) {
            need(nrhs == 2 && (int)mxGetNumberOfElements(prhs[1]) == n * B, "batch_violations needs K (n x B)");
            ArmourViolation* v = (ArmourViolation*)mxMalloc(sizeof(ArmourViolation) * B);
            chk(armour_batch_eval_violations(g_bt, mxGetPr(prhs[1]), v));
            plhs[0] = mxCreateDoubleMatrix(6, B, mxREAL);
            for (int b = 0; b < B; b++) violation_column(v[b], mxGetPr(plhs[0]) + (size_t)b * 6);
            mxFree(v);
        } else if (!strcmp(cmd, "batch_eval")) {
            need(nrhs == 2 && (int)mxGetNumberOfElements(prhs[1]) == n * B, "batch_eval needs K (n x B)");
            plhs[0] = mxCreateDoubleMatrix(m, B, mxREAL);
            mxArray* jac = mxCreateDoubleMatrix((mwSize)n * m, B, mxREAL);
            chk(armour_batch_eval_g_jac(g_bt, mxGetPr(prhs[1]), mxGetPr(plhs[0]), mxGetPr(jac)));
            if (nlhs > 1) plhs[1] = jac; else mxDestroyArray(jac);
        } else {
            mexErrMsgTxt("unknown command");
        }
        return;
    }
    need(g_h != nullptr, "call armour_hip_mex('create', T) first");
    if (!strcmp(cmd, "set_problem")) {
        need(nrhs == 6, "set_problem needs q0, qd0, qdd0, q_des, Z");
        for (int i = 1; i <= 4; i++) need(mxGetNumberOfElements(prhs[i]) == 7, "state vectors must have 7 entries");
        need(mxGetNumberOfElements(prhs[5]) % 12 == 0, "Z must be 12 x nObs");
        const int nObs = (int)(mxGetNumberOfElements(prhs[5]) / 12);
        chk(armour_set_problems(g_h, 1, nObs, mxGetPr(prhs[1]), mxGetPr(prhs[2]), mxGetPr(prhs[3]), mxGetPr(prhs[4]), mxGetPr(prhs[5])));
        if (nlhs > 0) { plhs[0] = col(1); chk(armour_get_prune_margin(g_h, mxGetPr(plhs[0]))); }
        return;
    }
    if (!strcmp(cmd, "set_problem_armtd")) {
        need(nrhs == 7, "set_problem_armtd needs q0, qd0, q_des, JRS, k_range, Z");
        for (int i = 1; i <= 3; i++) need(mxGetNumberOfElements(prhs[i]) == 7, "state vectors must have 7 entries");
        need(mxGetNumberOfElements(prhs[4]) % 42 == 0 && mxGetNumberOfElements(prhs[5]) == 7, "JRS must be T x 6 x 7 and k_range 7 x 1");
        need(mxGetNumberOfElements(prhs[6]) % 12 == 0, "Z must be 12 x nObs");
        const int nObs = (int)(mxGetNumberOfElements(prhs[6]) / 12);
        // column-major T x 6 x 7 is the [joint][row][t] order of the C ABI; its T must be the T of 'create' (the library checks sizes only through it)
        chk(armour_set_problems_armtd(g_h, 1, nObs, mxGetPr(prhs[1]), mxGetPr(prhs[2]), mxGetPr(prhs[3]), mxGetPr(prhs[4]), mxGetPr(prhs[5]), mxGetPr(prhs[6])));
        return;
    }
    if (!strcmp(cmd, "traj")) {  // stateless: needs no problem set
        need(nrhs == 6, "traj needs q0, qd0, qdd0, k, t");
        for (int i = 1; i <= 4; i++) need(mxGetNumberOfElements(prhs[i]) == 7, "q0, qd0, qdd0, k must have 7 entries");
        mxArray* o[3] = {col(7), col(7), col(7)};
        chk(armour_desired_trajectory(7, mxGetPr(prhs[1]), mxGetPr(prhs[2]), mxGetPr(prhs[3]), g_pr.k_range, g_pr.duration, mxGetPr(prhs[4]),
                                      mxGetScalar(prhs[5]), mxGetPr(o[0]), mxGetPr(o[1]), mxGetPr(o[2])));
        for (int i = 0; i < 3; i++) { if (i < nlhs || i == 0) plhs[i] = o[i]; else mxDestroyArray(o[i]); }
        return;
    }
    int B, n, m;
    chk(armour_get_sizes(g_h, &B, &n, &m));
    if (!strcmp(cmd, "eval")) {
        need(nrhs == 2 && (int)mxGetNumberOfElements(prhs[1]) == n, "eval needs k (n x 1)");
        plhs[0] = col(m);
        mxArray* jac = mxCreateNumericMatrix(n, m, mxDOUBLE_CLASS, mxREAL);  // column-major n x m == row-major values[m][n]
        chk(armour_eval_g_jac(g_h, mxGetPr(prhs[1]), mxGetPr(plhs[0]), mxGetPr(jac)));
        if (nlhs > 1) plhs[1] = jac; else mxDestroyArray(jac);
    } else if (!strcmp(cmd, "constraints")) {
        need((nrhs == 2 || nrhs == 3) && (int)mxGetNumberOfElements(prhs[1]) == n, "constraints needs k (n x 1) [, 'pruned']");
        // ('constraints', k, 'pruned'): only the rows that can be violated for some k -- the list KSI/uarmtd_planner.m:577-583,628-690 builds by hand
        // (armour_get_row_relevance; the same rows at every k of a problem, so fmincon sees a fixed constraint count)
        bool pruned = false;
        if (nrhs == 3) { char opt[16] = ""; need(!mxGetString(prhs[2], opt, sizeof(opt)) && !strcmp(opt, "pruned"), "the only option of constraints is 'pruned'"); pruned = true; }
        unsigned char* keep = nullptr;
        if (pruned) { keep = (unsigned char*)mxMalloc((size_t)m); chk(armour_get_row_relevance(g_h, keep, nullptr, nullptr)); }
        mxArray* g = col(m);
        mxArray* jac = mxCreateNumericMatrix(n, m, mxDOUBLE_CLASS, mxREAL);
        mxArray* b[4] = {col(n), col(n), col(m), col(m)};
        chk(armour_eval_g_jac(g_h, mxGetPr(prhs[1]), mxGetPr(g), mxGetPr(jac)));
        chk(armour_get_bounds(g_h, mxGetPr(b[0]), mxGetPr(b[1]), mxGetPr(b[2]), mxGetPr(b[3])));
        const double *gv = mxGetPr(g), *jv = mxGetPr(jac), *gl = mxGetPr(b[2]), *gu = mxGetPr(b[3]);
        int two = 0, kept = 0;
        for (int i = 0; i < m; i++) { if (keep && !keep[i]) continue; kept++; two += gl[i] > -1e18; }  // collision rows are one-sided (g_l = -1e19, RT/NLPclass.cu:131-140)
        mxArray* h = col(kept + two);
        mxArray* gh = mxCreateNumericMatrix(n, kept + two, mxDOUBLE_CLASS, mxREAL);
        double *hv = mxGetPr(h), *ghv = mxGetPr(gh);
        int r = kept, u = 0;
        for (int i0 = 0; i0 < m; i0++) {
            if (keep && !keep[i0]) continue;
            const int i = u++;   // position among the kept rows; i0: the row of g
            hv[i] = gv[i0] - gu[i0];
            for (int j = 0; j < n; j++) ghv[(size_t)i * n + j] = jv[(size_t)i0 * n + j];
            if (gl[i0] > -1e18) {
                hv[r] = gl[i0] - gv[i0];
                for (int j = 0; j < n; j++) ghv[(size_t)r * n + j] = -jv[(size_t)i0 * n + j];
                r++;
            }
        }
        if (keep) mxFree(keep);
        mxDestroyArray(g); mxDestroyArray(jac);
        for (int i = 0; i < 4; i++) mxDestroyArray(b[i]);
        plhs[0] = h;
        if (nlhs > 1) plhs[1] = mxCreateDoubleMatrix(0, 0, mxREAL);   // heq = []
        if (nlhs > 2) plhs[2] = gh; else mxDestroyArray(gh);
        if (nlhs > 3) plhs[3] = mxCreateDoubleMatrix(n, 0, mxREAL);   // grad_heq: n x 0
    } else if (!strcmp(cmd, "relevance")) {   // rel = armour_hip_mex('relevance'): m x 1, 1 = the row can be violated for some k (armour_get_row_relevance)
        unsigned char* keep = (unsigned char*)mxMalloc((size_t)m);
        chk(armour_get_row_relevance(g_h, keep, nullptr, nullptr));
        plhs[0] = col(m);
        for (int i = 0; i < m; i++) mxGetPr(plhs[0])[i] = keep[i] ? 1.0 : 0.0;
        mxFree(keep);
    } else if (!strcmp(cmd, "prune_margin")) {   // margin = armour_hip_mex('prune_margin'): min |norm - SIMPLIFY_THRESHOLD| / threshold over the last build's simplify() verdicts
        plhs[0] = col(1);
        chk(armour_get_prune_margin(g_h, mxGetPr(plhs[0])));
    } else if (!strcmp(cmd, "bounds")) {
        mxArray* o[4] = {col(n), col(n), col(m), col(m)};
        chk(armour_get_bounds(g_h, mxGetPr(o[0]), mxGetPr(o[1]), mxGetPr(o[2]), mxGetPr(o[3])));
        for (int i = 0; i < 4; i++) { if (i < nlhs || i == 0) plhs[i] = o[i]; else mxDestroyArray(o[i]); }
    } else if (!strcmp(cmd, "cost")) {
        need(nrhs == 2 && (int)mxGetNumberOfElements(prhs[1]) == n, "cost needs k (n x 1)");
        plhs[0] = mxCreateDoubleMatrix(1, 1, mxREAL);
        chk(armour_eval_f(g_h, mxGetPr(prhs[1]), mxGetPr(plhs[0])));
        if (nlhs > 1) { plhs[1] = col(n); chk(armour_eval_grad_f(g_h, mxGetPr(prhs[1]), mxGetPr(plhs[1]))); }
    } else if (!strcmp(cmd, "solve")) {
        ArmourSolveOptions so;
        armour_solve_options_default(&so);
        if (nrhs > 1) so.max_wall_time_s = mxGetScalar(prhs[1]);
        ArmourSolveResult r;
        chk(armour_solve(g_h, &so, &r));
        plhs[0] = col(n);
        memcpy(mxGetPr(plhs[0]), r.k_opt, n * sizeof(double));
        if (nlhs > 1) plhs[1] = mxCreateLogicalScalar(r.feasible != 0);
        if (nlhs > 2) {
            plhs[2] = col(5);
            double* p = mxGetPr(plhs[2]);
            p[0] = r.cost; p[1] = r.iterations; p[2] = r.evaluations; p[3] = r.status; p[4] = r.time_ms;
        }
    } else if (!strcmp(cmd, "violations")) {
        need(nrhs == 2 && (int)mxGetNumberOfElements(prhs[1]) == n, "violations needs k (n x 1)");
        ArmourViolation v;
        chk(armour_eval_violations(g_h, mxGetPr(prhs[1]), &v));
        plhs[0] = col(6);
        violation_column(v, mxGetPr(plhs[0]));
    } else if (!strcmp(cmd, "pz")) {
        char which[16];
        need(nrhs == 4 && !mxGetString(prhs[1], which, sizeof(which)), "pz needs which ('link'|'torque'), i, t");
        const int w = !strcmp(which, "torque") ? 1 : 0, sz = w ? 1 : 3;
        const int i = (int)mxGetScalar(prhs[2]) - 1, t = (int)mxGetScalar(prhs[3]) - 1;
        double cen[6];
        const int cnt = armour_get_pz(g_h, 0, w, i, t, cen, nullptr, nullptr, 0);
        chk(cnt);
        uint64_t* keys = (uint64_t*)mxMalloc(sizeof(uint64_t) * (cnt > 0 ? cnt : 1));
        mxArray* G = mxCreateDoubleMatrix(sz, cnt, mxREAL);  // column-major sz x cnt == the ABI's coeffs[cnt][sz]
        chk(armour_get_pz(g_h, 0, w, i, t, cen, keys, mxGetPr(G), cnt));
        plhs[0] = col(sz);
        memcpy(mxGetPr(plhs[0]), cen, sz * sizeof(double));
        mxArray* o[4] = {G, mxCreateDoubleMatrix(sz, sz, mxREAL), mxCreateDoubleMatrix(n, cnt, mxREAL), col(n)};
        for (int e = 0; e < sz; e++) mxGetPr(o[1])[e * sz + e] = cen[sz + e];                         // Grest = diag(independent radius)
        for (int mo = 0; mo < cnt; mo++) for (int f = 0; f < n; f++) mxGetPr(o[2])[mo * n + f] = (double)((keys[mo] >> (2 * f)) & 3);  // RT/PZsparse.h:8-21
        for (int f = 0; f < n; f++) mxGetPr(o[3])[f] = f + 1;
        mxFree(keys);
        for (int k2 = 0; k2 < 4; k2++) { if (k2 + 1 < nlhs) plhs[k2 + 1] = o[k2]; else mxDestroyArray(o[k2]); }
    } else {
        mexErrMsgTxt("unknown command");
    }
}
