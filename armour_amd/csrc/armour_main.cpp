// armour_main -- drop-in for the reference's planner executable (RT/armour_main.cu): the file protocol of
// KSI/uarmtd_planner.m:158-219 on top of libarmour_hip.so.
//
//   armour_main [buffer_dir] [num_time_steps]
//
// reads  <buffer_dir>/armour.in   (4 x 7 numbers q0 qd0 qdd0 q_des, nObs, nObs x 12 numbers; armour_main.cu:53-76)
// writes <buffer_dir>/armour.out  (k_opt or -1, then the time in ms), armour_joint_position_center.out,
//        armour_joint_position_radius.out, armour_control_input_radius.out, armour_constraints.out (armour_main.cu:312-372)
// The reference compiles the buffer path in (BufferPath.h, kinova_src/initialize.m:32-36); here it is argv[1]
// (default "buffer/").  Exit code 0 = ran (even if no feasible plan), non-zero = error, with "-1" in armour.out,
// which is what uarmtd_planner.m:196-208 tests.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
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
    const int T = argc > 2 ? atoi(argv[2]) : 128;  // NUM_TIME_STEPS, RT/Parameters.h:17
    const std::string out1 = dir + "armour.out";
    { std::ofstream touch(out1); }  // "declare this first and make sure we always have a new output" (armour_main.cu:36)

    ArmourRobot rb;
    ArmourParams pr;
    armour_robot_kinova_gen3_no_gripper(&rb);
    armour_params_default(&pr, T);
    const int n = rb.num_factors;
    std::ifstream in(dir + "armour.in");
    if (!in.is_open()) { std::ofstream o(out1); o << -1; fprintf(stderr, "        HIP & C++: Error reading input files !\n"); return 1; }
    std::vector<double> q0(n), qd0(n), qdd0(n), q_des(n);
    for (auto* v : {&q0, &qd0, &qdd0, &q_des}) for (int i = 0; i < n; i++) in >> (*v)[i];
    int nobs = 0;
    in >> nobs;
    if (!in || nobs < 0) { std::ofstream o(out1); o << -1; fprintf(stderr, "        HIP & C++: bad obstacle count\n"); return 1; }
    std::vector<double> obs((size_t)nobs * 12);
    for (auto& v : obs) in >> v;
    if (!in && nobs > 0) { std::ofstream o(out1); o << -1; fprintf(stderr, "        HIP & C++: input file too short\n"); return 1; }

    const auto t0 = std::chrono::steady_clock::now();
    ArmourPlanner* h = nullptr;
    if (armour_create(&rb, &pr, nullptr, 0, &h) != ARMOUR_OK) return fail(out1, "armour_create");
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
    const int J = rb.num_joints;
    std::vector<double> g(m), cen((size_t)T * J * 3), gens((size_t)T * J * 18), tr((size_t)n * T);
    if (armour_eval_g_jac(h, res.k_opt, g.data(), nullptr) != ARMOUR_OK) return fail(out1, "eval_g");
    armour_get_link_centers(h, res.k_opt, cen.data());
    armour_get_link_generators(h, gens.data());
    armour_get_torque_radius(h, tr.data());
    {
        std::ofstream o(out1);
        o << std::setprecision(10);
        if (res.feasible) for (int i = 0; i < n; i++) o << res.k_opt[i] << '\n';
        else o << -1 << '\n';
        o << total_ms;
    }
    {
        std::ofstream o(dir + "armour_joint_position_center.out");
        o << std::setprecision(10);
        for (int i = 0; i < T * J; i++) { for (int l = 0; l < 3; l++) o << cen[(size_t)i * 3 + l] << ' '; o << '\n'; }
    }
    {
        std::ofstream o(dir + "armour_joint_position_radius.out");
        o << std::setprecision(10);
        for (int i = 0; i < T * J; i++) for (int k = 0; k < 3; k++) { for (int l = 0; l < 6; l++) o << gens[(size_t)i * 18 + k * 6 + l] << ' '; o << '\n'; }
    }
    {
        std::ofstream o(dir + "armour_control_input_radius.out");
        o << std::setprecision(10);
        for (int i = 0; i < T; i++) { for (int j = 0; j < n; j++) o << tr[(size_t)j * T + i] << ' '; o << '\n'; }
    }
    {
        std::ofstream o(dir + "armour_constraints.out");
        o << std::setprecision(6);
        for (int i = 0; i < m; i++) o << g[i] << '\n';
    }
    armour_destroy(h);
    return 0;
}
