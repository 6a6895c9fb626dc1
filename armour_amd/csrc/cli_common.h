// The GPU side of the file-protocol executables (armour_worker.cpp): one planning iteration over the reference's file
// protocol, and the resident-planner mode.  The per-iteration programs armour_main / armtd_main are planner_client.cpp.
//
// The reference's MATLAB side spawns the planner executable once per planning iteration (KSI/uarmtd_planner.m:189,331:
// `system('./armour_main')`).  A fresh process pays the GPU runtime's start-up -- context, code objects, first
// allocations: ~0.2 s here, against ~2 ms for the reach sets and ~0.2 ms for the solve -- on every iteration.  So the
// executables have two shapes behind the same command line:
//   * `armour_main --serve [buffer_dir] [T]` stays resident: it owns the planner handles and listens on the UNIX socket
//     <buffer_dir>/armour.sock;
//   * `armour_main [buffer_dir] [T]` / `armtd_main [buffer_dir] [T]` first try that socket and, if a resident planner
//     answers, only forward the request (the process never touches the GPU); otherwise they start armour_worker for this
//     one iteration.
// Files, formats, exit codes and the `-1` conventions are the same either way (RT/armour_main.cu:36-76,312-372;
// CMP/armtd_main.cu:36-110,218-267).
#pragma once

#include <sys/stat.h>

#include <cerrno>
#include <charconv>
#include <chrono>
#include <csignal>
#include <sys/prctl.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <string>
#include <thread>
#include <vector>

#include "../../include/armour_hip.h"
#include "cli_socket.h"

namespace cli {

struct Session {  // planner handles kept across iterations (one per mode and T)
    ArmourPlanner* armour = nullptr; int armour_T = 0;
    ArmourPlanner* armtd = nullptr; int armtd_T = 0;
    ArmourPlanner* armour_nic = nullptr; int armour_nic_T = 0;   // ARMOUR without input constraints (TURN_OFF_INPUT_CONSTRAINTS, RT/Parameters.h:46-47)
    ~Session() { if (armour) armour_destroy(armour); if (armtd) armour_destroy(armtd); if (armour_nic) armour_destroy(armour_nic); }
};

inline int fail(const std::string& out1, const char* what) {
    std::ofstream o(out1);
    o << -1 << '\n';
    fprintf(stderr, "        HIP & C++: %s: %s\n", what, armour_last_error());
    return 1;
}
inline int bad_input(const std::string& out1, const char* what) {
    std::ofstream o(out1);
    o << -1;
    fprintf(stderr, "        HIP & C++: %s\n", what);
    return 1;
}

inline int get_handle(ArmourPlanner** h, int* have_T, int T, ArmourRobot* rb, ArmourParams* pr, bool input_constraints_off = false) {
    armour_robot_kinova_gen3_no_gripper(rb);
    armour_params_default(pr, T);
    pr->input_constraints_off = input_constraints_off ? 1 : 0;
    if (*h && *have_T == T) return ARMOUR_OK;
    if (*h) { armour_destroy(*h); *h = nullptr; }
    const int rc = armour_create(rb, pr, nullptr, 0, h);
    if (rc == ARMOUR_OK) *have_T = T;
    return rc;
}

// The output files (RT/armour_main.cu:312-372: `ofstream << setprecision(10)`, i.e. printf's %.10g; the constraints with precision 6).  Round 6: a
// file is formatted into one buffer with std::to_chars (chars_format::general with a precision IS %g: the same bytes) and written with one
// fwrite -- the iostream form spent 4 of a resident iteration's 6-7 ms of process wall on these ~36 000 numbers.
struct OutFile {
    std::vector<char> buf;
    size_t len = 0;
    explicit OutFile(size_t numbers) { buf.resize(numbers * 26 + 64); }   // (%.10g: at most sign + 1 + '.' + 9 + "e-308" = 17 characters, and a separator)
    void num(double v, int precision) {
        if (len + 40 > buf.size()) buf.resize(buf.size() * 2);
        const auto r = std::to_chars(buf.data() + len, buf.data() + buf.size(), v, std::chars_format::general, precision);
        len = (size_t)(r.ptr - buf.data());
    }
    void ch(char c) { if (len + 1 > buf.size()) buf.resize(buf.size() * 2); buf[len++] = c; }
    void integer(int v) { const auto r = std::to_chars(buf.data() + len, buf.data() + buf.size(), v); len = (size_t)(r.ptr - buf.data()); }
    bool write(const std::string& path) const {
        FILE* f = fopen(path.c_str(), "wb");
        if (!f) return false;
        const bool ok = fwrite(buf.data(), 1, len, f) == len;
        return fclose(f) == 0 && ok;
    }
};

// (`tr`: the torque radii [n][T] for armour_control_input_radius.out, or nullptr: RT/armour_main.cu:355 / the comparison planner has none)
// The three large files -- 12 600 + 14 700 + 2 100 numbers at T = 100, O = 10 -- are formatted side by side on threads of their own; the small ones here.
inline void write_common_outputs(const std::string& dir, const char* prefix, int T, int J, int n, int m, const ArmourSolveResult& res, double total_ms,
                                 const std::vector<double>& g, const std::vector<double>& cen, const std::vector<double>& gens, const double* tr = nullptr) {
    std::thread radius([&] {
        OutFile o((size_t)T * J * 18 + (size_t)T * J * 3);
        for (int i = 0; i < T * J; i++) for (int k = 0; k < 3; k++) { for (int l = 0; l < 6; l++) { o.num(gens[(size_t)i * 18 + k * 6 + l], 10); o.ch(' '); } o.ch('\n'); }
        o.write(dir + prefix + "_joint_position_radius.out");
    });
    std::thread constraints([&] {
        OutFile o((size_t)m);
        for (int i = 0; i < m; i++) { o.num(g[i], 6); o.ch('\n'); }
        o.write(dir + prefix + "_constraints.out");
    });
    {
        OutFile o((size_t)T * J * 3 + (size_t)T * J);
        for (int i = 0; i < T * J; i++) { for (int l = 0; l < 3; l++) { o.num(cen[(size_t)i * 3 + l], 10); o.ch(' '); } o.ch('\n'); }
        o.write(dir + prefix + "_joint_position_center.out");
    }
    if (tr) {
        OutFile o((size_t)T * n + (size_t)T);
        for (int i = 0; i < T; i++) { for (int j = 0; j < n; j++) { o.num(tr[(size_t)j * T + i], 10); o.ch(' '); } o.ch('\n'); }
        o.write(dir + prefix + "_control_input_radius.out");
    }
    radius.join();
    constraints.join();
    {   // last: the file the caller reads first (KSI/uarmtd_planner.m:209-219) exists once everything else does
        OutFile o((size_t)n + 2);
        if (res.feasible) for (int i = 0; i < n; i++) { o.num(res.k_opt[i], 10); o.ch('\n'); }
        else { o.integer(-1); o.ch('\n'); }
        o.num(total_ms, 10);
        o.write(dir + prefix + ".out");
    }
}
// ARMOUR_CLI_TIMING=1: where an iteration's wall time went, one line on stdout (the resident planner's log)
inline bool cli_timing() { static const bool on = [] { const char* e = getenv("ARMOUR_CLI_TIMING"); return e && e[0] && strcmp(e, "0") != 0; }(); return on; }

// armour.in -> armour.out + 4 files (RT/armour_main.cu).  The clock starts before the input is parsed; creating the
// planner handle (GPU context, allocations) is outside it when the handle already exists, as the reference keeps its
// allocations outside its own clock (armour_main.cu:86-88).
// input_off: the reference built with TURN_OFF_INPUT_CONSTRAINTS (RT/Parameters.h:46-47): no torque rows, no control-input-radius file (RT/armour_main.cu:355)
inline int iteration_armour(Session& s, const std::string& dir, int T, bool input_off = false) {
    const std::string out1 = dir + "armour.out";
    { std::ofstream touch(out1); }  // "declare this first and make sure we always have a new output" (armour_main.cu:36)
    ArmourRobot rb;
    ArmourParams pr;
    armour_robot_kinova_gen3_no_gripper(&rb);
    auto t0 = std::chrono::steady_clock::now();
    const int n = rb.num_factors, J = rb.num_joints;
    std::ifstream in(dir + "armour.in");
    if (!in.is_open()) return bad_input(out1, "Error reading input files !");
    std::vector<double> q0(n), qd0(n), qdd0(n), q_des(n);
    for (auto* v : {&q0, &qd0, &qdd0, &q_des}) for (int i = 0; i < n; i++) in >> (*v)[i];
    int nobs = 0;
    in >> nobs;
    if (!in || nobs < 0) return bad_input(out1, "bad obstacle count");
    std::vector<double> obs((size_t)nobs * 12);
    for (auto& v : obs) in >> v;
    if (!in && nobs > 0) return bad_input(out1, "input file too short");
    {
        const auto c0 = std::chrono::steady_clock::now();
        if (get_handle(input_off ? &s.armour_nic : &s.armour, input_off ? &s.armour_nic_T : &s.armour_T, T, &rb, &pr, input_off) != ARMOUR_OK) return fail(out1, "armour_create");
        t0 += std::chrono::steady_clock::now() - c0;  // creating the handle is outside the clock
    }
    ArmourPlanner* h = input_off ? s.armour_nic : s.armour;

    if (armour_set_problems(h, 1, nobs, q0.data(), qd0.data(), qdd0.data(), q_des.data(), obs.data()) != ARMOUR_OK) return fail(out1, "reach-set build");
    const double t_reach = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("        HIP & C++: Time taken by generating reachable sets: %.3f milliseconds\n", t_reach * 1e3);

    ArmourSolveOptions so;
    armour_solve_options_default(&so);
    so.max_wall_time_s = pr.duration * 0.5 - t_reach - 0.05;  // DURATION/2 - t(P1) - IPOPT_TIME_BUFFER (armour_main.cu:227-229)
    if (so.max_wall_time_s < 1e-3) so.max_wall_time_s = 1e-3;
    ArmourSolveResult res;
    if (armour_solve(h, &so, &res) != ARMOUR_OK) return fail(out1, "armour_solve");
    const double total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("        HIP & C++: %s (status %d, %d iterations, %d evaluations, cost %.6g)\n",
           res.feasible ? "Found a feasible solution!" : "Did not find a feasible solution!", res.status, res.iterations, res.evaluations,
           res.cost / pr.cost_scale);

    int B, nn, m;
    armour_get_sizes(h, &B, &nn, &m);
    std::vector<double> g(m), cen((size_t)T * J * 3), gens((size_t)T * J * 18), tr((size_t)n * T);
    if (armour_eval_g_jac(h, res.k_opt, g.data(), nullptr) != ARMOUR_OK) return fail(out1, "eval_g");
    armour_get_link_centers(h, res.k_opt, cen.data());
    armour_get_link_generators(h, gens.data());
    armour_get_torque_radius(h, tr.data());
    const auto t_diag = std::chrono::steady_clock::now();
    write_common_outputs(dir, "armour", T, J, n, m, res, total_ms, g, cen, gens, input_off ? nullptr : tr.data());   // (no control-input-radius file without input constraints: RT/armour_main.cu:355)
    if (cli_timing()) {
        const auto t_end = std::chrono::steady_clock::now();
        auto ms = [](std::chrono::steady_clock::duration d) { return std::chrono::duration<double, std::milli>(d).count(); };
        printf("        HIP & C++: [timing] input + reach sets %.3f ms, solve %.3f ms, diagnostics (g at k_opt, link centres, generators, radii) %.3f ms, five output files %.3f ms\n",
               t_reach * 1e3, total_ms - t_reach * 1e3, ms(t_diag - t0) - total_ms, ms(t_end - t_diag));
    }
    fflush(stdout);
    return 0;
}

// armtd.in -> armtd.out + 3 files (CMP/armtd_main.cu)
inline int iteration_armtd(Session& s, const std::string& dir, int T) {
    const std::string out1 = dir + "armtd.out";
    { std::ofstream touch(out1); }  // always a new output file (armtd_main.cu:36)
    ArmourRobot rb;
    ArmourParams pr;
    armour_robot_kinova_gen3_no_gripper(&rb);
    auto t0 = std::chrono::steady_clock::now();
    const int n = rb.num_factors, J = rb.num_joints;
    std::ifstream in(dir + "armtd.in");
    if (!in.is_open()) return bad_input(out1, "Error reading input files !");
    std::vector<double> q0(n), qd0(n), q_des(n), jrs((size_t)n * 6 * T), k_range(n);
    for (auto* v : {&q0, &qd0, &q_des}) for (int i = 0; i < n; i++) in >> (*v)[i];
    for (int i = 0; i < n; i++) {
        for (int e = 0; e < 6 * T; e++) in >> jrs[(size_t)i * 6 * T + e];
        in >> k_range[i];
    }
    int nobs = 0;
    in >> nobs;
    if (!in || nobs < 0) return bad_input(out1, "input file too short or bad obstacle count");
    std::vector<double> obs((size_t)nobs * 12);
    for (auto& v : obs) in >> v;
    if (!in && nobs > 0) return bad_input(out1, "input file too short");
    {
        const auto c0 = std::chrono::steady_clock::now();
        if (get_handle(&s.armtd, &s.armtd_T, T, &rb, &pr) != ARMOUR_OK) return fail(out1, "armour_create");
        t0 += std::chrono::steady_clock::now() - c0;  // creating the handle is outside the clock
    }
    ArmourPlanner* h = s.armtd;

    if (armour_set_problems_armtd(h, 1, nobs, q0.data(), qd0.data(), q_des.data(), jrs.data(), k_range.data(), obs.data()) != ARMOUR_OK)
        return fail(out1, "reach-set build");
    const double t_reach = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("        HIP & C++: Time taken by generating trajectory & forward kinematics: %.3f milliseconds\n", t_reach * 1e3);

    ArmourSolveOptions so;
    armour_solve_options_default(&so);
    so.tolerance = 1e-7;       // IPOPT_OPTIMIZATION_TOLERANCE, CMP/Parameters.h:39
    so.max_wall_time_s = 0.4;  // IPOPT_MAX_WALL_TIME, CMP/Parameters.h:41
    ArmourSolveResult res;
    if (armour_solve(h, &so, &res) != ARMOUR_OK) return fail(out1, "armour_solve");
    const double total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("        HIP & C++: %s (status %d, %d iterations, %d evaluations, cost %.6g)\n",
           res.feasible ? "Found a feasible solution!" : "Did not find a feasible solution!", res.status, res.iterations, res.evaluations,
           res.cost / pr.cost_scale);

    int B, nn, m;
    armour_get_sizes(h, &B, &nn, &m);
    std::vector<double> g(m), cen((size_t)T * J * 3), gens((size_t)T * J * 18);
    if (armour_eval_g_jac(h, res.k_opt, g.data(), nullptr) != ARMOUR_OK) return fail(out1, "eval_g");
    armour_get_link_centers(h, res.k_opt, cen.data());
    armour_get_link_generators(h, gens.data());
    write_common_outputs(dir, "armtd", T, J, n, m, res, total_ms, g, cen, gens);
    fflush(stdout);
    return 0;
}

// ---------------------------------------------------------------------------------------------- resident planner
// `--serve`: create the handles up front (for both protocols when their T is given), then answer requests until "quit",
// SIGTERM / SIGINT, or `idle_s` seconds without one (0 = stay).  One request at a time: a planning iteration owns the GPU.
inline volatile sig_atomic_t g_stop = 0;
inline void on_signal(int) { g_stop = 1; }

inline int serve(const std::string& dir, int T_armour, int T_armtd, double idle_s) {
    Session s;
    ArmourRobot rb;
    ArmourParams pr;
    if (T_armour > 0 && get_handle(&s.armour, &s.armour_T, T_armour, &rb, &pr) != ARMOUR_OK) { fprintf(stderr, "armour_create: %s\n", armour_last_error()); return 1; }
    if (T_armtd > 0 && get_handle(&s.armtd, &s.armtd_T, T_armtd, &rb, &pr) != ARMOUR_OK) { fprintf(stderr, "armour_create: %s\n", armour_last_error()); return 1; }
    if (s.armour) {  // load the code objects and size the work buffers before the first real request
        // (MAX_OBSTACLE_NUM = 40 boxes far away, RT/Parameters.h:22, so that no later request has to grow a device buffer)
        const int O = 40;
        std::vector<double> z(rb.num_factors, 0.0), boxes((size_t)O * 12, 0.0);
        for (int o = 0; o < O; o++) { double* b = &boxes[(size_t)o * 12]; b[0] = b[1] = b[2] = 5.0 + o; b[3] = b[7] = b[11] = 0.05; }
        (void)armour_set_problems(s.armour, 1, O, z.data(), z.data(), z.data(), z.data(), boxes.data());
        ArmourSolveOptions so;
        armour_solve_options_default(&so);
        so.max_iterations = 1;
        ArmourSolveResult r;
        (void)armour_solve(s.armour, &so, &r);
    }
    const std::string path = socket_path(dir);
    sockaddr_un a;
    if (!fill_addr(path, &a)) { fprintf(stderr, "socket path too long: %s\n", path.c_str()); return 1; }
    unlink(path.c_str());
    const int ls = socket(AF_UNIX, SOCK_STREAM, 0);
    if (ls < 0 || bind(ls, (sockaddr*)&a, sizeof(a)) != 0 || listen(ls, 4) != 0) { fprintf(stderr, "cannot listen on %s: %s\n", path.c_str(), strerror(errno)); return 1; }
    chmod(path.c_str(), 0600);
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = on_signal;  // no SA_RESTART: accept() returns EINTR and the loop sees g_stop
    sigaction(SIGTERM, &sa, nullptr);
    sigaction(SIGINT, &sa, nullptr);
    signal(SIGPIPE, SIG_IGN);
    printf("        HIP & C++: resident planner listening on %s\n", path.c_str());
    fflush(stdout);
    bool quit = false;
    while (!quit && !g_stop) {
        if (idle_s > 0) {
            fd_set rd;
            FD_ZERO(&rd);
            FD_SET(ls, &rd);
            timeval tv{(time_t)idle_s, (suseconds_t)((idle_s - (time_t)idle_s) * 1e6)};
            const int r = select(ls + 1, &rd, nullptr, nullptr, &tv);
            if (r == 0) break;         // idle for too long
            if (r < 0) continue;       // interrupted: the loop condition decides
        }
        const int fd = accept(ls, nullptr, nullptr);
        if (fd < 0) continue;
        char req[64];
        int got = 0;
        while (got < (int)sizeof(req) - 1) {
            const ssize_t r = read(fd, req + got, sizeof(req) - 1 - got);
            if (r <= 0) break;
            got += (int)r;
            if (req[got - 1] == '\n') break;
        }
        req[got] = 0;
        char kind[16] = "";
        int T = 0, code = 1;
        if (sscanf(req, "%15s %d", kind, &T) >= 1) {
            if (!strcmp(kind, "quit")) { quit = true; code = 0; }
            else if (!strcmp(kind, "armour") && T >= 2) code = iteration_armour(s, dir, T);
            else if (!strcmp(kind, "armour_nic") && T >= 2) code = iteration_armour(s, dir, T, true);
            else if (!strcmp(kind, "armtd") && T >= 2) code = iteration_armtd(s, dir, T);
        }
        char reply[32];
        const int len = snprintf(reply, sizeof(reply), "done %d\n", code);
        (void)!write(fd, reply, len);
        close(fd);
    }
    close(ls);
    unlink(path.c_str());
    return 0;
}

// main() of the worker: `armour_worker armour|armtd [--serve] [--idle-seconds S] [buffer_dir] [T] [T_armtd]`
inline int worker_main(int argc, char** argv) {
    if (argc < 2 || (strcmp(argv[1], "armour") && strcmp(argv[1], "armtd") && strcmp(argv[1], "armour_nic"))) { fprintf(stderr, "usage: armour_worker armour|armour_nic|armtd [--serve] [buffer_dir] [T]\n"); return 2; }
    // started by armour_main / armtd_main (planner_client.cpp): if that front end dies -- e.g. killed by its caller --
    // this process, which holds the GPU and the socket, gets SIGTERM instead of living on as an orphan
    // Only when a front end did start it (ARMOUR_WORKER_PARENT = its pid): a worker started directly -- `rocprofv3 -- armour_worker ...
    // --serve` from a script, nohup -- must not be terminated when the shell that launched it exits.  And the front end may have died
    // before this line runs: then the worker has already been re-parented, the death signal would never come, and it leaves now.
    if (const char* pp = getenv("ARMOUR_WORKER_PARENT")) {
        prctl(PR_SET_PDEATHSIG, SIGTERM);
        if ((long long)getppid() != atoll(pp)) { fprintf(stderr, "        HIP & C++: the front end that started this worker is gone\n"); return 1; }
    }
    const char* kind = argv[1];
    const bool is_nic = !strcmp(kind, "armour_nic");   // ARMOUR without input constraints
    const bool is_armour = !strcmp(kind, "armour") || is_nic;
    bool want_serve = false;
    double idle_s = 0;
    std::vector<std::string> pos;
    for (int i = 2; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--serve") want_serve = true;
        else if (a == "--idle-seconds" && i + 1 < argc) idle_s = atof(argv[++i]);
        else if (a == "--no-input-constraints") {}   // (the front end has turned it into the kind already)
        else pos.push_back(a);
    }
    const std::string dir = buffer_dir(pos.size() > 0 ? pos[0].c_str() : nullptr);
    const int T = pos.size() > 1 ? atoi(pos[1].c_str()) : (is_armour ? 128 /* RT/Parameters.h:17 */ : 100 /* CMP/Parameters.h:17 */);
    if (want_serve) {
        // serving as `armour`: both protocols from one process (T, then the comparison planner's T); as `armtd`: its own only
        const int T2 = pos.size() > 2 ? atoi(pos[2].c_str()) : 100;
        return serve(dir, is_armour ? T : 0, is_armour ? T2 : T, idle_s);
    }
    Session s;
    return is_armour ? iteration_armour(s, dir, T, is_nic) : iteration_armtd(s, dir, T);
}

}  // namespace cli
