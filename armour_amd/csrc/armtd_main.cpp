// armtd_main -- drop-in for the executable of the reference's ARMTD comparison planner (CMP/armtd_main.cu): the file
// protocol of KSI/uarmtd_planner.m:257-345 on top of libarmour_hip.so (armour_set_problems_armtd).
//
//   armtd_main [buffer_dir] [num_time_steps]
//
// reads  <buffer_dir>/armtd.in   (q0, qd0, q_des: 3 x 7 numbers; per joint 6 rows of T numbers -- centre, k-generator and
//        radius of cos, then of sin -- and its k_range; nObs; nObs x 12 numbers; armtd_main.cu:57-110)
// writes <buffer_dir>/armtd.out  (k_opt or -1, then the time in ms), armtd_joint_position_center.out,
//        armtd_joint_position_radius.out, armtd_constraints.out (armtd_main.cu:218-267)
// The reference compiles the buffer path in (BufferPath.h); here it is argv[1] (default "buffer/").  T defaults to the
// reference's NUM_TIME_STEPS = 100 (CMP/Parameters.h:17): the offline tables hold 100 intervals.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <string>
#include <vector>

#include "../../include/armour_hip.h"

static int fail(const std::string& out1, const char* what) {
    std::ofstream o(out1);
    o << -1 << '\n';
    fprintf(stderr, "        HIP & C++: %s: %s\n", what, armour_last_error());
    return 1;
}

int main(int argc, char** argv) {
    std::string dir = argc > 1 ? argv[1] : "buffer/";
    if (!dir.empty() && dir.back() != '/') dir += '/';
    const int T = argc > 2 ? atoi(argv[2]) : 100;
    const std::string out1 = dir + "armtd.out";
    { std::ofstream touch(out1); }  // always a new output file (armtd_main.cu:36)

    ArmourRobot rb;
    ArmourParams pr;
    armour_robot_kinova_gen3_no_gripper(&rb);
    armour_params_default(&pr, T);
    const int n = rb.num_factors, J = rb.num_joints;
    std::ifstream in(dir + "armtd.in");
    if (!in.is_open()) { std::ofstream o(out1); o << -1; fprintf(stderr, "        HIP & C++: Error reading input files !\n"); return 1; }
    std::vector<double> q0(n), qd0(n), q_des(n), jrs((size_t)n * 6 * T), k_range(n);
    for (auto* v : {&q0, &qd0, &q_des}) for (int i = 0; i < n; i++) in >> (*v)[i];
    for (int i = 0; i < n; i++) {
        for (int e = 0; e < 6 * T; e++) in >> jrs[(size_t)i * 6 * T + e];
        in >> k_range[i];
    }
    int nobs = 0;
    in >> nobs;
    if (!in || nobs < 0) { std::ofstream o(out1); o << -1; fprintf(stderr, "        HIP & C++: input file too short or bad obstacle count\n"); return 1; }
    std::vector<double> obs((size_t)nobs * 12);
    for (auto& v : obs) in >> v;
    if (!in && nobs > 0) { std::ofstream o(out1); o << -1; fprintf(stderr, "        HIP & C++: input file too short\n"); return 1; }

    const auto t0 = std::chrono::steady_clock::now();
    ArmourPlanner* h = nullptr;
    if (armour_create(&rb, &pr, nullptr, 0, &h) != ARMOUR_OK) return fail(out1, "armour_create");
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
    {
        std::ofstream o(out1);
        o << std::setprecision(10);
        if (res.feasible) for (int i = 0; i < n; i++) o << res.k_opt[i] << '\n';
        else o << -1 << '\n';
        o << total_ms;
    }
    {
        std::ofstream o(dir + "armtd_joint_position_center.out");
        o << std::setprecision(10);
        for (int i = 0; i < T * J; i++) { for (int l = 0; l < 3; l++) o << cen[(size_t)i * 3 + l] << ' '; o << '\n'; }
    }
    {
        std::ofstream o(dir + "armtd_joint_position_radius.out");
        o << std::setprecision(10);
        for (int i = 0; i < T * J; i++) for (int k = 0; k < 3; k++) { for (int l = 0; l < 6; l++) o << gens[(size_t)i * 18 + k * 6 + l] << ' '; o << '\n'; }
    }
    {
        std::ofstream o(dir + "armtd_constraints.out");
        o << std::setprecision(6);
        for (int i = 0; i < m; i++) o << g[i] << '\n';
    }
    armour_destroy(h);
    return 0;
}
