// Socket side of the resident-planner mode (see cli_common.h): nothing here touches the GPU or libarmour_hip, so the
// per-iteration executables that include only this header start in about a millisecond.
#pragma once

#include <sys/socket.h>
#include <sys/un.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <string>

namespace cli {

inline std::string buffer_dir(const char* arg) {
    std::string dir = arg ? arg : "buffer/";
    if (!dir.empty() && dir.back() != '/') dir += '/';
    return dir;
}

inline std::string socket_path(const std::string& dir) { return dir + "armour.sock"; }

inline bool fill_addr(const std::string& path, sockaddr_un* a) {
    memset(a, 0, sizeof(*a));
    a->sun_family = AF_UNIX;
    if (path.size() >= sizeof(a->sun_path)) return false;
    memcpy(a->sun_path, path.c_str(), path.size() + 1);
    return true;
}

// Forward one request ("armour <T>", "armtd <T>" or "quit") to the resident planner of `dir`.  Returns false if none
// answers (no socket, stale socket); otherwise true with the planner's exit code for the iteration in *rc.
inline bool try_resident(const std::string& dir, const char* kind, int T, int* rc) {
    sockaddr_un a;
    if (!fill_addr(socket_path(dir), &a)) return false;
    const int fd = socket(AF_UNIX, SOCK_STREAM, 0);
    if (fd < 0) return false;
    if (connect(fd, (sockaddr*)&a, sizeof(a)) != 0) { close(fd); return false; }
    char msg[64];
    const int len = snprintf(msg, sizeof(msg), "%s %d\n", kind, T);
    bool ok = write(fd, msg, len) == len;
    char reply[64];
    int got = 0;
    while (ok && got < (int)sizeof(reply) - 1) {
        const ssize_t r = read(fd, reply + got, sizeof(reply) - 1 - got);
        if (r <= 0) break;
        got += (int)r;
        if (reply[got - 1] == '\n') break;
    }
    close(fd);
    reply[got] = 0;
    int code = 1;
    if (!ok || sscanf(reply, "done %d", &code) != 1) return false;  // the planner went away mid-request: run it here
    *rc = code;
    return true;
}

}  // namespace cli
