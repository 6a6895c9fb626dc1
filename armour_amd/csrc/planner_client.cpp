// armour_main / armtd_main -- drop-ins for the two planner executables of the reference (RT/armour_main.cu,
// CMP/armtd_main.cu) over its file protocol (KSI/uarmtd_planner.m:158-219 and :257-345).  This source builds both; the
// name it is started under selects the protocol.
//
//   armour_main [buffer_dir] [num_time_steps=128]      one planning iteration: armour.in -> armour.out + 4 files
//   armtd_main  [buffer_dir] [num_time_steps=100]      one planning iteration: armtd.in  -> armtd.out  + 3 files
//   armour_main --serve [buffer_dir] [T_armour] [T_armtd] [--idle-seconds S]     stay resident for both of the above
//   armour_main --quit  [buffer_dir]                                             stop the resident planner
//
// The reference compiles the buffer path in (BufferPath.h, kinova_src/initialize.m:32-36); here it is the first
// argument (default "buffer/").  Exit code 0 = ran (even if no feasible plan), non-zero = error with "-1" in the .out
// file -- what uarmtd_planner.m:196-208 tests.
//
// This process never touches the GPU: it forwards the iteration to the resident planner listening on
// <buffer_dir>/armour.sock if there is one, and otherwise starts armour_worker (next to this executable) as a CHILD process,
// relays SIGTERM / SIGINT to it and returns its exit code.  It never replaces itself with the worker (exec): under a
// profiler (`rocprofv3 -- armour_main ...`) a preloaded library has initialised the GPU before main() runs, and an exec
// from a GPU-initialised process takes the machine down on this pool -- profile `armour_worker armour|armtd ...` directly
// instead.  The worker asks the kernel for SIGTERM should this process die (cli_common.h), so killing
// the `--serve` front end never leaves an orphan holding the GPU and the socket.
// File formats and conventions live in cli_common.h, which only the worker includes.
#include <libgen.h>
#include <signal.h>
#include <spawn.h>
#include <sys/wait.h>

#include <unistd.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "cli_socket.h"

extern char** environ;

static volatile pid_t g_child = 0;
static void relay_signal(int sig) { if (g_child > 0) kill(g_child, sig); }

static int run_worker(const std::string& self, const char* kind, int argc, char** argv) {
    std::string path = self;
    const size_t slash = path.find_last_of('/');
    path = (slash == std::string::npos ? std::string() : path.substr(0, slash + 1)) + "armour_worker";
    std::vector<char*> av;
    av.push_back(const_cast<char*>(path.c_str()));
    av.push_back(const_cast<char*>(kind));
    for (int i = 1; i < argc; i++) av.push_back(argv[i]);
    av.push_back(nullptr);
    // The relay is in place BEFORE the worker exists: the handlers are installed first and SIGTERM / SIGINT stay blocked until g_child
    // is set, so a signal that arrives while the worker is being started is delivered to the handler right after and relayed --
    // there is no window in which it kills this front end and leaves the worker behind.  (The worker gets the default, unblocked
    // dispositions through the spawn attributes.)
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = relay_signal;
    sigaction(SIGTERM, &sa, nullptr);
    sigaction(SIGINT, &sa, nullptr);
    sigset_t relay_set, old_set, none;
    sigemptyset(&relay_set); sigaddset(&relay_set, SIGTERM); sigaddset(&relay_set, SIGINT);
    sigemptyset(&none);
    sigprocmask(SIG_BLOCK, &relay_set, &old_set);
    posix_spawnattr_t attr;
    posix_spawnattr_init(&attr);
    posix_spawnattr_setsigmask(&attr, &none);
    posix_spawnattr_setsigdefault(&attr, &relay_set);
    posix_spawnattr_setflags(&attr, POSIX_SPAWN_SETSIGMASK | POSIX_SPAWN_SETSIGDEF);
    // tells the worker that a front end started it: only then does it ask for SIGTERM on this process's death (cli_common.h)
    setenv("ARMOUR_WORKER_PARENT", std::to_string((long long)getpid()).c_str(), 1);
    pid_t pid;
    const int spawn_rc = posix_spawn(&pid, path.c_str(), nullptr, &attr, av.data(), environ);
    posix_spawnattr_destroy(&attr);
    if (spawn_rc != 0) { sigprocmask(SIG_SETMASK, &old_set, nullptr); fprintf(stderr, "        HIP & C++: cannot start %s\n", path.c_str()); return 1; }
    g_child = pid;
    sigprocmask(SIG_SETMASK, &old_set, nullptr);
    int status = 0;
    while (waitpid(pid, &status, 0) < 0) {}  // (EINTR after a relayed signal: keep waiting for the worker's own exit)
    return WIFEXITED(status) ? WEXITSTATUS(status) : 1;
}

int main(int argc, char** argv) {
    std::string self = argv[0];
    std::string base = self.substr(self.find_last_of('/') == std::string::npos ? 0 : self.find_last_of('/') + 1);
    const char* kind = base.rfind("armtd", 0) == 0 ? "armtd" : "armour";
    bool serve = false, quit = false;
    // TURN_OFF_INPUT_CONSTRAINTS (RT/Parameters.h:46-47) is a compile-time switch of the reference; here: `--no-input-constraints`, or -- for a MATLAB
    // caller that cannot change the command line of uarmtd_planner.m:189 -- ARMOUR_TURN_OFF_INPUT_CONSTRAINTS=1 in the environment (setenv)
    const char* nic_env = getenv("ARMOUR_TURN_OFF_INPUT_CONSTRAINTS");
    bool nic = nic_env && nic_env[0] && strcmp(nic_env, "0") != 0;
    std::vector<const char*> pos;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--serve") serve = true;
        else if (a == "--quit") quit = true;
        else if (a == "--idle-seconds") i++;
        else if (a == "--no-input-constraints") nic = true;
        else pos.push_back(argv[i]);
    }
    if (nic && kind[3] == 'o') kind = "armour_nic";
    const std::string dir = cli::buffer_dir(pos.size() > 0 ? pos[0] : nullptr);
    int rc = 1;
    if (quit) return cli::try_resident(dir, "quit", 0, &rc) ? rc : 1;
    if (!serve) {
        const int T = pos.size() > 1 ? atoi(pos[1]) : (kind[3] == 'o' ? 128 : 100);  // RT/Parameters.h:17, CMP/Parameters.h:17
        if (cli::try_resident(dir, kind, T, &rc)) return rc;
    }
    return run_worker(self, kind, argc, argv);
}
