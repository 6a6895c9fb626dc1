// In-process multi-device batch (include/armour_hip.h, "in-process multi-device batch").
//
// One caller thread drives several GPUs: an ArmourBatch owns one ArmourPlanner per device slot, B independent planning
// problems are dealt to the slots in contiguous blocks (the partition of armour_amd/sharding.py: sizes differ by at most
// one), and every call runs ONE HOST THREAD PER SLOT, each driving its own handle on its own device and stream.  The
// caller's arrays are problem-major, so slot d works on the sub-arrays that start at problem first[d]: nothing is copied
// or re-packed on the host, results land in place.  Problems share nothing: no collective, no device-to-device traffic.
// The reference has no counterpart (one problem per process, RT/armour_main.cu:11-375); this is the in-process form of
// bench.py's one-rank-per-GPU sharding, for a MATLAB / MEX caller that cannot start ranks.
//
// Host code only (the kernels are the single-handle entry points'); errors of the worker threads are collected and the
// first one becomes the caller thread's armour_last_error().
#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "common.h"

namespace {
// One worker thread per device slot, started by armour_batch_create and fed through a condition variable: the throughput entries
// (armour_batch_eval_*) are called once per NLP iterate, and a thread created and joined per slot and call cost tens of microseconds
// of each.  A job is a function of the slot index; the caller thread waits for all of them.
struct SlotWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    const std::function<int(int)>* job = nullptr;
    bool pending = false, done = false, stop = false;
    int index = 0, rc = ARMOUR_OK;
    std::string msg;
    void run() {
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return pending || stop; });
            if (stop) return;
            const std::function<int(int)>* j = job;
            lk.unlock();
            int r = ARMOUR_OK;
            std::string m;
            try {
                r = (*j)(index);
                if (r != ARMOUR_OK) m = armour_last_error();   // (the error string is thread-local: fetch it on the worker)
            } catch (...) { r = ARMOUR_EDEVICE; m = "exception in a device slot's worker"; }   // nothing may cross the extern "C" boundary
            lk.lock();
            rc = r; msg = std::move(m); pending = false; done = true;
            lk.unlock();
            cv.notify_all();
        }
    }
};
}  // namespace

struct ArmourBatch {
    std::vector<ArmourPlanner*> slot;   // one handle per device slot
    std::vector<int> device;
    std::vector<int> first;             // [n_slots + 1] problem ranges of the current problem set
    std::vector<std::unique_ptr<SlotWorker>> worker;   // (empty for a single slot: its calls run on the caller's thread)
    int user_build = 0;                 // ARMOUR_OPT_P1_BUILD as the caller set it on the batch (0: automatic, decided per problem set for ALL slots)
    int n = 0, T = 0;
    int B = 0, O = 0, m = 0;
    bool ready = false;
};

extern "C" int armour_batch_partition(int32_t B, int32_t n_slots, int32_t* first) {
    if (B < 0 || n_slots < 1 || !first) { armour_set_error("armour_batch_partition: bad argument"); return ARMOUR_EINVAL; }
    const int base = B / n_slots, extra = B % n_slots;
    for (int d = 0; d <= n_slots; d++) first[d] = d * base + (d < extra ? d : extra);
    return ARMOUR_OK;
}

namespace {

// fn(slot index) for every slot that owns at least one problem, each on its slot's worker thread (a single active slot: on the caller's);
// returns the first failing slot's code and makes its message the caller's last error
int on_every_slot(ArmourBatch* bt, const std::function<int(int)>& fn) {
    const int G = (int)bt->slot.size();
    std::vector<int> rc(G, ARMOUR_OK);
    std::vector<std::string> msg(G);
    int active = 0, last = -1;
    for (int d = 0; d < G; d++)
        if (bt->first[d + 1] > bt->first[d]) { active++; last = d; }
    if (active == 1 || bt->worker.empty()) {
        for (int d = 0; d < G; d++)
            if (bt->first[d + 1] > bt->first[d]) {
                rc[d] = fn(d);
                if (rc[d] != ARMOUR_OK) msg[d] = armour_last_error();
            }
        (void)last;
    } else {
        for (int d = 0; d < G; d++)
            if (bt->first[d + 1] > bt->first[d]) {
                SlotWorker& w = *bt->worker[d];
                { std::lock_guard<std::mutex> lk(w.mu); w.job = &fn; w.done = false; w.pending = true; }
                w.cv.notify_all();
            }
        for (int d = 0; d < G; d++)
            if (bt->first[d + 1] > bt->first[d]) {
                SlotWorker& w = *bt->worker[d];
                std::unique_lock<std::mutex> lk(w.mu);
                w.cv.wait(lk, [&] { return w.done; });
                rc[d] = w.rc; msg[d] = w.msg;
            }
    }
    for (int d = 0; d < G; d++)
        if (rc[d] != ARMOUR_OK) { armour_set_error("device slot %d (device %d): %s", d, bt->device[d], msg[d].c_str()); return rc[d]; }
    return ARMOUR_OK;
}

}  // namespace

extern "C" int armour_batch_create(const ArmourRobot* robot, const ArmourParams* params, const ArmourLimits* limits,
                                   const int32_t* devices, int32_t n_devices, ArmourBatch** out) {
    if (!robot || !params || !devices || n_devices < 1 || !out) { armour_set_error("armour_batch_create: bad argument"); return ARMOUR_EINVAL; }
    ArmourBatch* bt = new (std::nothrow) ArmourBatch();
    if (!bt) { armour_set_error("out of host memory"); return ARMOUR_EINVAL; }
    for (int d = 0; d < n_devices; d++) {
        ArmourPlanner* h = nullptr;
        const int rc = armour_create(robot, params, limits, devices[d], &h);
        if (rc != ARMOUR_OK) { armour_batch_destroy(bt); return rc; }
        bt->slot.push_back(h);
        bt->device.push_back(devices[d]);
    }
    bt->first.assign(n_devices + 1, 0);
    bt->n = robot->num_factors; bt->T = params->num_time_steps;
    if (n_devices > 1) {
        try {
            for (int d = 0; d < n_devices; d++) {
                bt->worker.emplace_back(new SlotWorker());
                SlotWorker* w = bt->worker.back().get();
                w->index = d;
                w->th = std::thread([w] { w->run(); });
            }
        } catch (const std::exception& e) {   // (std::system_error from the thread start, std::bad_alloc)
            armour_set_error("armour_batch_create: could not start the worker thread of device slot %d: %s", (int)bt->worker.size() - 1, e.what());
            if (!bt->worker.empty() && !bt->worker.back()->th.joinable()) bt->worker.pop_back();
            armour_batch_destroy(bt);
            return ARMOUR_EDEVICE;
        }
    }
    *out = bt;
    return ARMOUR_OK;
}

extern "C" void armour_batch_destroy(ArmourBatch* bt) {
    if (!bt) return;
    for (auto& w : bt->worker) {
        { std::lock_guard<std::mutex> lk(w->mu); w->stop = true; }
        w->cv.notify_all();
        if (w->th.joinable()) w->th.join();
    }
    for (ArmourPlanner* h : bt->slot) armour_destroy(h);
    delete bt;
}

extern "C" int armour_batch_set_option(ArmourBatch* bt, int32_t option, double value) {
    if (!bt) { armour_set_error("null batch"); return ARMOUR_EINVAL; }
    for (ArmourPlanner* h : bt->slot) {
        const int rc = armour_set_option(h, option, value);
        if (rc != ARMOUR_OK) return rc;
    }
    if (option == ARMOUR_OPT_P1_BUILD) bt->user_build = (int)value;
    return ARMOUR_OK;
}

extern "C" int armour_batch_set_problems(ArmourBatch* bt, int32_t B, int32_t O, const double* q0, const double* qd0, const double* qdd0,
                                         const double* q_des, const double* obstacles) {
    if (!bt || !q0 || !qd0 || !qdd0 || !q_des) { armour_set_error("null argument"); return ARMOUR_EINVAL; }
    if (B < 1 || O < 0 || (O > 0 && !obstacles)) { armour_set_error("bad batch/obstacle count (B=%d, O=%d)", B, O); return ARMOUR_EINVAL; }
    bt->ready = false;
    const int G = (int)bt->slot.size();
    armour_batch_partition(B, G, bt->first.data());
    const int n = bt->n;
    // ONE reach-set kernel for all slots of a problem set (every slot used to choose from its own shard, so the 18 + 17
    // problems of a batch of 35 over two slots were built by two different kernels, whose radii differ to 1e-12).  With the option left
    // automatic the batch takes the kernel the SMALLEST shard would take (armour_p1_build's rule: time-vectorised from B_shard * T >= 50 *
    // ARMOUR_OPT_P1_TV_MIN_GROUPS on) -- deciding from the whole batch instead would put shards of a few problems on the time-vectorised
    // kernel, three times slower there.  Tables still depend on the NUMBER of slots within the stated 1e-12; a caller that needs them
    // bit-identical for every device count pins ARMOUR_OPT_P1_BUILD to 1 or 2 (include/armour_hip.h).
    if (bt->user_build == 0) {
        int smallest = B;
        for (int d = 0; d < G; d++)
            if (bt->first[d + 1] > bt->first[d]) smallest = std::min(smallest, bt->first[d + 1] - bt->first[d]);
        double min_groups = 31.0;   // (= kTuning's default; the option always exists, this is only the value if the query failed)
        (void)armour_get_option(bt->slot[0], ARMOUR_OPT_P1_TV_MIN_GROUPS, &min_groups);
        const int build = (long long)smallest * bt->T >= 50ll * (long long)min_groups ? 2 : 1;
        for (ArmourPlanner* h : bt->slot) (void)armour_set_option(h, ARMOUR_OPT_P1_BUILD, build);
    }
    const int rc = on_every_slot(bt, [&](int d) {
        const size_t f = (size_t)bt->first[d];
        return armour_set_problems(bt->slot[d], bt->first[d + 1] - bt->first[d], O, q0 + f * n, qd0 + f * n, qdd0 + f * n, q_des + f * n,
                                   obstacles ? obstacles + f * O * ARMOUR_OBS_DOUBLES : nullptr);
    });
    if (rc != ARMOUR_OK) return rc;
    bt->B = B; bt->O = O;
    for (int d = 0; d < G; d++)
        if (bt->first[d + 1] > bt->first[d]) { int32_t m = 0; armour_get_sizes(bt->slot[d], nullptr, nullptr, &m); bt->m = m; }
    bt->ready = true;
    return ARMOUR_OK;
}

#define BATCH_READY(bt)                                                                                  \
    if (!(bt)) { armour_set_error("null batch"); return ARMOUR_EINVAL; }                                 \
    if (!(bt)->ready) { armour_set_error("no problem set: call armour_batch_set_problems first"); return ARMOUR_ESTATE; }

extern "C" int armour_batch_get_sizes(const ArmourBatch* bt, int32_t* B, int32_t* n, int32_t* m, int32_t* n_slots) {
    BATCH_READY(bt);
    if (B) *B = bt->B;
    if (n) *n = bt->n;
    if (m) *m = bt->m;
    if (n_slots) *n_slots = (int32_t)bt->slot.size();
    return ARMOUR_OK;
}

extern "C" int armour_batch_get_bounds(ArmourBatch* bt, double* x_l, double* x_u, double* g_l, double* g_u) {
    BATCH_READY(bt);
    const size_t m = (size_t)bt->m;
    return on_every_slot(bt, [&](int d) {
        const size_t f = (size_t)bt->first[d];
        return armour_get_bounds(bt->slot[d], d == 0 ? x_l : nullptr, d == 0 ? x_u : nullptr,
                                 g_l ? g_l + f * m : nullptr, g_u ? g_u + f * m : nullptr);
    });
}

extern "C" int armour_batch_eval_g_jac(ArmourBatch* bt, const double* k, double* g, double* jac) {
    BATCH_READY(bt);
    if (!k) { armour_set_error("k is null"); return ARMOUR_EINVAL; }
    const size_t n = (size_t)bt->n, m = (size_t)bt->m;
    return on_every_slot(bt, [&](int d) {
        const size_t f = (size_t)bt->first[d];
        return armour_eval_g_jac(bt->slot[d], k + f * n, g ? g + f * m : nullptr, jac ? jac + f * m * n : nullptr);
    });
}

extern "C" int armour_batch_eval_violations(ArmourBatch* bt, const double* k, ArmourViolation* out) {
    BATCH_READY(bt);
    if (!k || !out) { armour_set_error("null argument"); return ARMOUR_EINVAL; }
    const size_t n = (size_t)bt->n;
    return on_every_slot(bt, [&](int d) {
        const size_t f = (size_t)bt->first[d];
        return armour_eval_violations(bt->slot[d], k + f * n, out + f);
    });
}

extern "C" int armour_batch_solve(ArmourBatch* bt, const ArmourSolveOptions* opt, ArmourSolveResult* results) {
    BATCH_READY(bt);
    if (!results) { armour_set_error("null argument"); return ARMOUR_EINVAL; }
    return on_every_slot(bt, [&](int d) { return armour_solve(bt->slot[d], opt, results + bt->first[d]); });
}

extern "C" int armour_batch_get_build_info(ArmourBatch* bt, int32_t* info) {
    BATCH_READY(bt);
    if (!info) { armour_set_error("null argument"); return ARMOUR_EINVAL; }
    for (size_t d = 0; d < bt->slot.size(); d++) {
        int32_t* o = info + 4 * d;
        o[0] = o[1] = o[2] = o[3] = 0;
        if (bt->first[d + 1] > bt->first[d]) { const int rc = armour_get_build_info(bt->slot[d], o); if (rc != ARMOUR_OK) return rc; }
    }
    return ARMOUR_OK;
}

extern "C" int armour_batch_get_prune_margin(ArmourBatch* bt, double* margin) {
    BATCH_READY(bt);
    if (!margin) { armour_set_error("null argument"); return ARMOUR_EINVAL; }
    for (size_t d = 0; d < bt->slot.size(); d++)
        if (bt->first[d + 1] > bt->first[d]) { const int rc = armour_get_prune_margin(bt->slot[d], margin + bt->first[d]); if (rc != ARMOUR_OK) return rc; }
    return ARMOUR_OK;
}

extern "C" int armour_batch_get_build_ms(ArmourBatch* bt, double* max_ms, double* per_slot) {
    BATCH_READY(bt);
    double mx = 0.0;
    for (size_t d = 0; d < bt->slot.size(); d++) {
        double ms = 0.0;
        if (bt->first[d + 1] > bt->first[d]) { const int rc = armour_get_build_ms(bt->slot[d], &ms); if (rc != ARMOUR_OK) return rc; }
        if (per_slot) per_slot[d] = ms;
        if (ms > mx) mx = ms;
    }
    if (max_ms) *max_ms = mx;
    return ARMOUR_OK;
}
