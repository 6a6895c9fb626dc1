#!/usr/bin/env python3
"""Benchmark of ARMOUR's constraint callback on MI355X: planning iters/s (fused eval_g + eval_jac_g).

    python bench.py --gpus N --steps K --warmup W [--repeats R] [--batch B] [--obstacles O] [--time-steps T]

One "step" = one pass of the hot path (RT/NLPclass.cu:272-396: eval_g + eval_jac_g at a fresh k) over this
rank's batch of planning problems; one kernel launch per step, inputs (reach-set tables from
armour_set_problems, the k points) resident in HBM before the timed region.  The default workload is
BASELINE.json configs[1]: Kinova Gen3 7-DOF, 20 obstacles, 100 time steps, a single planning problem per
GPU.  Ranks (one per GPU) hold independent random worlds: the path shards with no collective, so scaling is
"weak" and `value` = problems*steps over all ranks / max-rank time.

Launching: `python bench.py --gpus N` starts its own N rank processes (one per device, rendezvous on
127.0.0.1) when it is not already running under torch.distributed.run; the parent process touches neither
torch nor HIP.  Under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the ranks are
the launcher's.

Timing: the K timed steps (one instantiated graph of K kernel nodes) are bracketed by barrier + device
synchronize on both sides and the interval is MAX-reduced over the ranks; this is repeated R times
(default 9) and the MEDIAN interval is what `value` and `ms_per_step` are computed from -- a single 20-step
interval is 0.1 ms and one scheduling hiccup moved the round-1 headline by 30 %.

The JSON line also carries
  roofline     HBM roofline of the P2 kernel: algorithmic bytes per launch / average launch
               duration measured with HIP events on the launch stream over the timed region;
  cpu_baseline the CPU oracle (oracle/, a port of the reference's host path) timed on this box's cores on a
               bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import glob
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "planning iters/s (eval_g+jac) Kinova 7-DOF, 100 timesteps × 20 obstacles"
OPTIONS = []   # --set-option ID=VALUE (development: A/B of a per-handle option, include/armour_hip.h), applied to every handle this run creates


def _opts(nlp):
    for opt, val in OPTIONS:
        nlp.set_option(opt, val)
    return nlp
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


# ----------------------------------------------------------------------------------------------- launcher
def launch_ranks(n_ranks, argv):
    """Start `n_ranks` fresh copies of this script, one per device, and return the largest exit code.  Runs in the
    parent BEFORE anything imports torch or touches HIP (a process that has initialised the GPU must not spawn-and-
    replace itself on this pool; plain child processes are fine).  Rank 0 inherits stdout, so its one JSON line is
    the parent's output."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                rc = max(rc, abs(code))
                if code != 0:           # one rank failed: the others would wait at the next barrier forever
                    for q in pending:
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


# ----------------------------------------------------------------------------------------------- measurement pieces
OMP_ENV = {"OMP_WAIT_POLICY": "passive", "OMP_PROC_BIND": "close", "OMP_PLACES": "cores"}


def cpu_baseline(T, O, seed, budget_s=12.0):
    """Parent side: the oracle is timed in a process of its own, started with the OpenMP settings that serve it best (VERDICT r5 item 7:
    waiting threads sleep instead of spinning, threads bound to neighbouring cores) -- libgomp reads them when it is loaded, and this
    process loaded it with torch.  The child touches no GPU.  Falls back to timing in this process when the child cannot be started."""
    env = dict(os.environ, **OMP_ENV)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "OMP_NUM_THREADS"):
        env.pop(k, None)
    try:
        env["BENCH_CPUS_USABLE"] = str(len(os.sched_getaffinity(0)))   # (this process's own mask: nothing here has bound its threads)
    except AttributeError:
        pass
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--time-steps", str(T), "--obstacles", str(O)]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        for line in r.stdout.splitlines():
            if line.startswith("CPU_BASELINE_JSON "):
                d = json.loads(line[len("CPU_BASELINE_JSON "):])
                d["openmp_environment"] = OMP_ENV
                return d
        err = (r.stderr or r.stdout)[-300:]
    except Exception as e:
        err = repr(e)
    d = cpu_baseline_body(T, O, seed, budget_s)
    d["openmp_environment"] = {"note": "timed in the bench process itself (default OpenMP settings): the child failed", "error": err}
    return d


def cpu_baseline_body(T, O, seed, budget_s=12.0):
    """Oracle (kind 'port') on the host cores: P2 evals/s on the same world, bounded to ~budget_s seconds."""
    # the CPUs this process may run on, BEFORE libgomp is loaded: with OMP_PROC_BIND set, loading it binds this thread to its first place -- one
    # core -- and the affinity mask read afterwards says 2 logical CPUs (round 6: the first run with the child's OpenMP settings swept {1, 2} threads)
    host = host_description()
    if os.environ.get("BENCH_CPUS_USABLE"):
        host["cpus_usable_by_this_process"] = int(os.environ["BENCH_CPUS_USABLE"])
    from oracle.cpu_oracle import Oracle, max_threads
    from armour_amd.worlds import random_k, random_problem
    p = random_problem(seed, O)
    o = Oracle(T=T).set_problem(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    ks = random_k(seed, 64)
    # thread count: the reference's default of 32 (NUM_THREADS, RT/Parameters.h:35), all cores and a few in between
    # (T = 100 time steps bound the useful parallelism, and a shared host punishes oversubscription); keep the fastest
    # (each setting of the sweep is timed for about a second: a burst of 16 evaluations -- 2 ms -- rated 64 threads at 8.4 k/s on a box
    #  where the sustained figure at that setting was 1.7 k/s; the host is shared, and what a wide team loses to preemption at its
    #  barriers does not show in a burst)
    # (the thread counts come from the CPUs this process may run on -- taken BEFORE the sweep: oracle_max_threads() is omp_get_max_threads(),
    #  which returns whatever the last omp_set_num_threads() of the sweep left behind -- round 4 timed P1 at {1, 16} instead of {1, 32, all})
    avail = int(host["cpus_usable_by_this_process"] or max_threads())
    best_t, best_rate, sweep = 0, 0.0, {}
    phys = int(host["physical_cores"] or avail)
    for th in sorted({min(c, avail) for c in (1, 8, 16, 32, 64, 128, phys, avail)}):
        o.time_eval(ks, 3, threads=th)  # warm-up
        burst = 16 / o.time_eval(ks, 16, threads=th)
        n = int(max(16, min(8000, burst * 1.0)))
        rate = n / o.time_eval(ks, n, threads=th)
        sweep[str(th)] = rate
        if rate > best_rate:
            best_t, best_rate = th, rate
    # the sample itself: budget_s seconds at the best setting, in four parts -- `value` is the whole sample's rate, the parts show how steady it was
    reps = int(max(20, min(20000, budget_s * best_rate)))
    part = max(5, reps // 4)
    part_secs = [o.time_eval(ks, part, threads=best_t) for _ in range(4)]
    reps, secs = 4 * part, sum(part_secs)
    # the reach-set build (P1, RT/armour_main.cu:96-216: OpenMP over the time steps) as its own timed figure: the same world at
    # 1 thread, at the reference's 32 (NUM_THREADS, RT/Parameters.h:35) and at all cores; best of 2 builds each (1 at one thread)
    p1 = {}
    for th in sorted({1, min(32, avail), min(phys, avail), avail}):
        ms = []
        for _ in range(1 if th == 1 else 2):
            ob = Oracle(T=T).set_problem(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"], threads=th)
            ms.append(ob.build_ms)
        p1[str(th)] = min(ms)
    return {
        "value": reps / secs, "unit": "iters/s", "cores": best_t, "kind": "port",
        "cores_note": "`cores` = the OpenMP threads of the fastest setting of the sweep (what `value` was measured with), not the box's core count: see host",
        "host": host, "cpu_model": host["cpu_model"], "physical_cores": host["physical_cores"], "logical_cpus": host["logical_cpus"],
        "thread_sweep_iters_per_s": sweep, "sample_parts_iters_per_s": [part / t for t in part_secs],
        "iters_per_s_at_all_physical_cores": sweep.get(str(min(phys, avail))), "iters_per_s_at_reference_32_threads": sweep.get(str(min(32, avail))),
        "sample": f"{reps} fused eval_g+eval_jac_g of 1 world (seed {seed}, O={O}, T={T}) at 64 cycling k points, "
                  f"OpenMP over time steps as RT/NLPclass.cu:304,376",
        "p1_build_ms_by_threads": p1, "p1_build_ms": min(p1.values()), "p1_pair_products": int(o.stats()["mul_pairs"]),
        "p1_sample": f"reach-set build (JRS, FK, RNEA x2, torque radius, half-space tables) of the same world, ms, by OpenMP threads",
    }


def host_description():
    """CPU model, sockets, physical cores and logical CPUs of this box (from /proc/cpuinfo; the CPU baseline is timed on them) and the
    CPUs this process may run on."""
    model, phys, logical, sockets = None, set(), 0, set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                logical += 1
            elif line.startswith("model name") and model is None:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip(); sockets.add(pid)
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip(); phys.add((pid, cid))
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count()
    return {"cpu_model": model, "sockets": len(sockets) or None, "physical_cores": len(phys) or None, "logical_cpus": logical or os.cpu_count(),
            "cpus_usable_by_this_process": usable}


def measured_traffic(B, O, T, name=None):
    """HBM bytes per P2 launch from the newest committed rocprofv3 PMC passes of this same configuration
    (profiles/r*_bench_{headline,configs2}_pmc.json: 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of
    MI355X_MICROARCH.md), with the file, the commit and the date the profile was taken at -- the number is read from a
    committed file, not measured in this run, so the line says which build it belongs to.  None when no profile of this
    configuration exists."""
    name = name or {(1, 20, 100): "headline", (128, 50, 100): "configs2"}.get((B, O, T))
    if name is None:
        return None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_bench_{name}_pmc.json")), reverse=True):
        try:
            d = json.load(open(path))
            return {"bytes": float(d["p2_hbm_traffic_bytes_per_launch"]), "profile": os.path.relpath(path, ROOT),
                    "profile_commit": d.get("commit"), "profile_date": d.get("date"), "l2_hit_rate": d.get("l2_hit_rate")}
        except Exception:
            continue
    return None


def p1_traffic(tag):
    """L2-miss traffic of every kernel of one reach-set build (bytes: 2 x FETCH_SIZE + WRITE_SIZE summed over the build's kernels) from the newest
    committed profiles/r*_p1_cache.json -- read from a committed file, not measured in this run -- or None."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_p1_cache.json")), reverse=True):
        try:
            d = json.load(open(path))
            ks = d["kernels"][tag]
            return {"bytes": float(sum(k["hbm_bytes_per_dispatch"] for k in ks.values())),
                    "reach_set_kernel_bytes": float(max(k["hbm_bytes_per_dispatch"] for k in ks.values())),
                    "l2_hit_rate_reach_set_kernel": max(ks.values(), key=lambda k: k["hbm_bytes_per_dispatch"]).get("l2_hit_rate"),
                    "profile": os.path.relpath(path, ROOT), "profile_commit": d.get("commit"), "profile_date": d.get("date")}
        except Exception:
            continue
    return None


def p1_accounting(device, T, O, p1_ms_b1, count_pairs):
    """set_problems ms per problem and monomial-pair products per second -- the pairs (sum over all operator* calls of
    (M_a + 1)(M_b + 1), RT/PZsparse.cu:864-994) counted by the CPU oracle on the same worlds (the oracle is the counter here, not the thing
    timed) -- for one problem and for a batch of 128 (configs[3]'s shard), with the build's counter traffic and its fraction of the HBM peak."""
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch, random_problem
    out = {}
    B = 128
    bp = random_batch(0, B, O)
    nlp = _opts(ArmourNLP(T=T, device=device))
    ms = []
    for _ in range(4):
        nlp.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
        ms.append(nlp.build_ms)
    info = nlp.build_info()
    margins = nlp.prune_margin()   # (armour_get_prune_margin: how close any simplify() verdict of each world's build came to flipping)
    nlp.close()
    pairs = {1: None, B: None}
    if count_pairs:
        from oracle.cpu_oracle import Oracle
        tot = 0
        for b in range(B):
            p = random_problem(b, O)
            tot += Oracle(T=T).set_problem(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"]).stats()["mul_pairs"]
            if b == 0:
                pairs[1] = tot
        pairs[B] = tot
    for tag, nb, build_ms in (("B=1", 1, p1_ms_b1), ("B=128", B, min(ms[1:]))):
        tr = p1_traffic(tag)
        e = {"problems": nb, "set_problems_ms": build_ms, "ms_per_problem": build_ms / nb, "pair_products": pairs[nb],
             "pair_products_per_s": (pairs[nb] / (build_ms * 1e-3)) if pairs[nb] else None,
             "traffic": tr, "frac_by_traffic": (tr["bytes"] / (build_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if tr else None}
        if nb == B:
            e["kernel"] = info
            e["prune_margin"] = {"min": float(margins.min()), "world": int(margins.argmin()), "below_1e-9": int((margins < 1e-9).sum()),
                                 "note": "smallest |norm - SIMPLIFY_THRESHOLD| / threshold over every simplify() verdict of the 128 builds, from the device (DESIGN.md 4.11): the parity tolerances hold while it stays above ~1e-9"}
        out[tag] = e
    # round 6: a lone problem's build runs every time step on two compute units -- the same world with the option as shipped and held to one CU
    from armour_amd import _lib
    p0 = random_problem(0, O)
    ab = {}
    for tag, opt in (("as_shipped", None), ("one_cu_per_time_step", 0)):
        nl = _opts(ArmourNLP(T=T, device=device))
        if opt is not None:
            nl.set_option(_lib.OPT_P1_STEP_TWO_CU, opt)
        t_ms = []
        for _ in range(5):
            nl.set_parameters(p0["q0"], p0["qd0"], p0["qdd0"], p0["q_des"], p0["obstacles"])
            t_ms.append(nl.build_ms)
        ab[tag] = {"set_problems_ms": min(t_ms[1:]), "ARMOUR_OPT_P1_STEP_TWO_CU": nl.get_option(_lib.OPT_P1_STEP_TWO_CU)}
        nl.close()
    out["B=1"]["two_cus_per_time_step"] = dict(ab, world="random_problem(0, O = %d)" % O,
                                               note="the option reads 0 after a build if the handle fell back to one CU per step (a helper block that did not deliver or started late)")
    out["note"] = ("set_problems_ms: HIP events around every kernel of the build (reach-set kernel + half-space kernels), best of the warm builds; "
                   "pair_products: counted by the CPU oracle on the same worlds (seeds 0..B-1, O = %d); traffic: L2-miss bytes of the same kernels from the "
                   "committed counter profile; no roofline is claimed for P1 beyond that fraction (irregular, latency-bound: DESIGN.md 4.2)" % O)
    return out


class Timed:
    """K fused evaluations of one handle's problems as one graph, timed R times.  Every interval is bracketed by
    barrier + synchronize on both sides and MAX-reduced over the ranks (armour_amd.sharding.reduce_max_elapsed)."""

    def __init__(self, nlp, dev, seed, K, W, use_dist, red_dev="same"):
        import torch
        from armour_amd.worlds import random_k
        self.torch, self.nlp, self.dev, self.K, self.W, self.use_dist = torch, nlp, dev, K, W, use_dist
        self.red_dev = (dev if use_dist else None) if red_dev == "same" else red_dev
        B, n, m = nlp.B, nlp.n, nlp.m
        self.ks = torch.tensor(random_k(seed, (K + W) * B, n=n).reshape(K + W, B, n), device=dev)  # a fresh k per step and problem
        self.d_g = torch.empty((B, m), device=dev, dtype=torch.float64)
        self.d_jac = torch.empty((B, m, n), device=dev, dtype=torch.float64)
        self.stream = torch.cuda.Stream(device=dev)  # a real (non-null) stream: HIP events on the null stream do not bracket the launches
        # the K timed launches go out as one instantiated graph (a chain of K kernel nodes), built before the timed region
        nlp.prepare_steps(self.ks[W:].data_ptr(), K, self.d_g.data_ptr(), self.d_jac.data_ptr())
        if W > 0:
            nlp.prepare_steps(self.ks.data_ptr(), W, self.d_g.data_ptr(), self.d_jac.data_ptr())   # (the warm-up steps in front of every timed interval)

    def barrier(self):
        if self.use_dist:
            import torch.distributed as dist
            dist.barrier()
        self.torch.cuda.synchronize()

    def run(self, repeats):
        from armour_amd.sharding import reduce_max_elapsed
        torch, nlp, K, W = self.torch, self.nlp, self.K, self.W
        sh = self.stream.cuda_stream
        wall, ev = [], []
        # Every timed interval is preceded by the W untimed warm-up steps (a 20-step interval is 0.1 ms: after the idle time between two
        # intervals its first launches found cold caches), and the interval is measured twice over -- once by the wall clock with nothing but
        # the K launches between the two barrier + synchronize brackets (`value`, `ms_per_step`), once more bracketed by HIP events on the
        # launch stream (the roofline's launch duration): the two event records cost a 20-step interval 5 % when they sat inside the wall-clock
        # bracket.  Same K steps, same k points, same graph both times.
        for _ in range(repeats):
            for timed_by_events in (False, True):
                if W > 0:
                    nlp.eval_g_jac_device_steps(self.ks.data_ptr(), W, self.d_g.data_ptr(), self.d_jac.data_ptr(), sh)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                self.barrier()
                t0 = time.perf_counter()
                if timed_by_events:
                    e0.record(self.stream)
                nlp.eval_g_jac_device_steps(self.ks[W:].data_ptr(), K, self.d_g.data_ptr(), self.d_jac.data_ptr(), sh)
                if timed_by_events:
                    e1.record(self.stream)
                while not self.stream.query():   # poll the stream: a sleeping synchronize adds its wake-up latency to a 0.1 ms interval
                    pass
                torch.cuda.synchronize()
                elapsed = time.perf_counter() - t0
                self.barrier()
                if timed_by_events:
                    ev.append(reduce_max_elapsed(e0.elapsed_time(e1) * 1e-3, device=self.red_dev))
                else:
                    wall.append(reduce_max_elapsed(elapsed, device=self.red_dev))
        return wall, ev

    def check(self, oracle_problems=None, tol_g=1e-9, tol_j=1e-8):
        """The last timed step's device outputs are finite and bit-equal to the synchronous host entry at the same k;
        with `oracle_problems` = [(b, world dict)], those problems' g / jac also agree with the CPU oracle."""
        import numpy as np
        k_last = self.ks[-1].cpu().numpy()
        g_host, jac_host = self.nlp.eval_g_jac(k_last)
        g_dev, jac_dev = self.d_g.cpu().numpy(), self.d_jac.cpu().numpy()
        assert np.isfinite(g_host).all() and np.isfinite(jac_host).all(), "non-finite constraint values"
        assert np.array_equal(g_host, g_dev) and np.array_equal(jac_host, jac_dev), "device entry != host entry"
        worst = None
        if oracle_problems:
            from oracle.cpu_oracle import Oracle
            dg = dj = 0.0
            for b, p in oracle_problems:
                o = Oracle(T=self.nlp.T).set_problem(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
                g_ref, jac_ref = o.eval_g_jac(k_last[b])
                dg = max(dg, float(np.abs(g_dev[b] - g_ref).max()))
                dj = max(dj, float(np.abs(jac_dev[b] - jac_ref).max()))
            assert dg <= tol_g and dj <= tol_j, (dg, dj)
            worst = {"problems": [b for b, _ in oracle_problems], "max_abs_dg": dg, "max_abs_djac": dj}
        return worst


def summarise(nlp, wall, ev, K, world, extra=None, traffic_name=None):
    """Three byte counts per launch, each divided by the HIP-event launch time:
      algorithmic  the reference formulation's bytes (1440 B per collision row + PZ tables + outputs) -- the yardstick
                   `roofline.achieved` is defined on; the kernel may legitimately move fewer (planes never needed are not stored)
      effective    what the layout the kernel actually reads holds (ArmourNLP.effective_bytes): a fraction from it cannot exceed 1
      traffic      HBM bytes by the rocprofv3 counters of the newest committed profile of this configuration (or None)"""
    b_alg, b_eff = nlp.algorithmic_bytes(), nlp.effective_bytes()
    med_wall, med_ev = statistics.median(wall), statistics.median(ev)
    launch_us = med_ev * 1e6 / K
    achieved = b_alg / (launch_us * 1e-6) / 1e9
    eff = b_eff / (launch_us * 1e-6) / 1e9
    tr = measured_traffic(nlp.B, nlp.O, nlp.T, traffic_name)
    out = {"problem_evals_per_s": world * nlp.B * K / med_wall, "ms_per_step": med_wall * 1e3 / K, "launch_us": launch_us,
           "algorithmic_bytes_per_launch": b_alg, "achieved_GBps": achieved, "algorithmic_equivalent_frac": achieved / HBM_PEAK_GBS,
           "effective_bytes_per_launch": b_eff, "effective_GBps": eff, "frac_effective": eff / HBM_PEAK_GBS,
           "traffic": tr, "frac_by_traffic": (tr["bytes"] / (launch_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if tr else None,
           "repeats": len(wall), "wall_ms_min_med_max": [min(wall) * 1e3, med_wall * 1e3, max(wall) * 1e3],
           "p1_set_problems_ms_per_problem": nlp.build_ms / nlp.B}
    if extra:
        out.update(extra)
    return out


def extra_config(device, dev, rank, world, use_dist, B, O, T, K, R, check_oracle, red_dev="same"):
    """One more BASELINE config measured exactly like the headline (never `value`): B worlds per GPU at O obstacles."""
    from armour_amd.planner import ArmourNLP
    from armour_amd.sharding import shard_seeds
    from armour_amd.worlds import random_batch, random_problem
    seeds = shard_seeds(1000, world * B, rank, world)
    probs = random_batch(seeds[0], len(seeds), O)
    nlp = _opts(ArmourNLP(T=T, device=device)).set_parameters(probs["q0"], probs["qd0"], probs["qdd0"], probs["q_des"], probs["obstacles"])
    tm = Timed(nlp, dev, 77 + rank, K, 4, use_dist, red_dev)
    wall, ev = tm.run(R)
    # output check: finite, device entry == host entry, two problems against the CPU oracle (rank 0; the oracle is the checker)
    spot = None
    if check_oracle is not None:
        picks = [(b, random_problem(seeds[b], O)) for b in ((0, B - 1) if check_oracle and rank == 0 else ())]
        spot = tm.check(picks)
    out = summarise(nlp, wall, ev, K, world, {"checked": "finite + device entry == host entry" + (" + oracle on problems 0 and B-1" if spot else ""),
                                               "oracle_spot_check": spot} if check_oracle is not None else {"checked": None})
    if check_oracle is not None:
        # the whole NLP solve of the same worlds (an extra, never `value`): the first solve after armour_set_problems -- which
        # builds the solver's row lists inside it -- and a repeated one; wall clock around the synchronous call
        nlp.solve()   # (untimed: the solver kernel's code object, the solver's buffers)
        nlp.set_parameters(probs["q0"], probs["qd0"], probs["qdd0"], probs["q_des"], probs["obstacles"])
        t1 = time.perf_counter(); sols = nlp.solve(); first_ms = (time.perf_counter() - t1) * 1e3
        rep = []
        for _ in range(3):
            t1 = time.perf_counter(); nlp.solve(); rep.append((time.perf_counter() - t1) * 1e3)
        _, n_col, n_tq, lists_ms = nlp.solver_rows()
        out["armour_solve"] = {"first_solve_ms": first_ms, "repeated_solve_ms": min(rep), "feasible": int(sum(int(s2["feasible"]) for s2 in sols)),
                               "iterations_max": int(max(s2["iterations"] for s2 in sols)),
                               "rows_walked": {"collision_mean": float(n_col.mean()), "collision_all": nlp.J * nlp.T * O, "torque_mean": float(n_tq.mean()), "torque_all": nlp.n * nlp.T,
                                               "lists_ms": lists_ms},
                               "note": "device-resident SQP on the rows that can pass its candidate filter for some k (ARMOUR_OPT_SOLVE_CULL automatic); same iterates as the full form"}
    nlp.close()
    return out


def fetch_config(device, dev, T, K, R, check):
    """BASELINE configs[4] (never `value`): the Fetch preset (CMP/FetchInfo.h: 9 links, 7 factors, mixed joint axes) with +-50 % mass /
    inertia uncertainty on the last link (the payload), 100 obstacles, one planning problem: reach-set build ms, fused evaluation
    us, and g / jac of the last timed step against the CPU oracle.  (The "8-DOF" of BASELINE.json would need an 8th factor, which the
    reference's 64-bit monomial key cannot hold: RT/PZsparse.h:8-21.)"""
    import numpy as np
    from armour_amd.planner import ArmourNLP, default_params, fetch_robot
    from armour_amd.worlds import random_fetch_problem
    O = 100
    p = random_fetch_problem(11, O)
    nlp = _opts(ArmourNLP(robot=fetch_robot(0.5), params=default_params(T), device=device))
    nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])   # (the second build: code objects loaded)
    tm = Timed(nlp, dev, 311, K, 4, False)
    wall, ev = tm.run(R)
    spot = None
    if check:
        from oracle.cpu_oracle import Oracle
        from oracle.cpu_oracle import default_params as oracle_params
        from oracle.cpu_oracle import fetch_robot as oracle_fetch
        o = Oracle(robot=oracle_fetch(0.5), params=oracle_params(T)).set_problem(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
        k_last = tm.ks[-1].cpu().numpy()
        g_ref, jac_ref = o.eval_g_jac(k_last[0])
        dg = float(np.abs(tm.d_g.cpu().numpy()[0] - g_ref).max())
        dj = float(np.abs(tm.d_jac.cpu().numpy()[0] - jac_ref).max())
        assert dg <= 1e-9 and dj <= 1e-8, (dg, dj)
        spot = {"max_abs_dg": dg, "max_abs_djac": dj, "oracle_build_ms": o.build_ms}
    out = summarise(nlp, wall, ev, K, 1, {"robot": "fetch (CMP/FetchInfo.h), last link +-50 % mass / inertia", "links": nlp.J, "obstacles": O,
                                          "constraints_m": nlp.m, "p1_set_problems_ms": nlp.build_ms, "oracle_spot_check": spot,
                                          "table_sizes": nlp.table_sizes()}, traffic_name="configs4")
    nlp.close()
    return out


def reference_scenes_config(device, dev, T, check):
    """The reference's own worlds as a workload (never `value`): first planning iteration of its 100 saved random scenes and 7 hard scenarios
    (armour_amd/scenes.py; kinova_src/saved_worlds/random/*.csv, KSI/kinova_scenarios/get_kinova_scenario_info.m) -- reach-set build ms, fused
    evaluation us, armour_solve ms, feasible count and SQP iterations of the 107 worlds as ONE batch (obstacle lists padded to 14 with a far box),
    every world ALONE with its own obstacle count (one problem per call is what the reference runs), and the same figures on 107 synthetic
    worlds of armour_amd/worlds.py at the same sizes."""
    import numpy as np
    import torch
    from armour_amd.planner import ArmourNLP
    from armour_amd.scenes import as_batch, reference_worlds
    from armour_amd.worlds import random_batch, random_k

    def batch_figures(bp, spot_world=None):
        B = bp["q0"].shape[0]
        nlp = _opts(ArmourNLP(T=T, device=device))
        for _ in range(2):
            nlp.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
        n, m = nlp.n, nlp.m
        st = torch.cuda.Stream()
        d_k = torch.from_numpy(random_k(91, B)).to(dev)
        d_g = torch.empty((B, m), device=dev, dtype=torch.float64)
        d_j = torch.empty((B, m, n), device=dev, dtype=torch.float64)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(2):
            if it:
                e0.record(st)
            for _ in range(20):
                nlp.eval_g_jac_device(d_k.data_ptr(), d_g.data_ptr(), d_j.data_ptr(), st.cuda_stream)
            if it:
                e1.record(st)
            st.synchronize()
        eval_us = e0.elapsed_time(e1) * 1e3 / 20
        spot = None
        if spot_world is not None:
            from oracle.cpu_oracle import Oracle
            b = spot_world
            o = Oracle(T=T).set_problem(bp["q0"][b], bp["qd0"][b], bp["qdd0"][b], bp["q_des"][b], bp["obstacles"][b])
            gr, jr = o.eval_g_jac(d_k[b].cpu().numpy())
            dg, dj = float(np.abs(d_g[b].cpu().numpy() - gr).max()), float(np.abs(d_j[b].cpu().numpy() - jr).max())
            assert dg <= 1e-9 and dj <= 1e-8, (dg, dj)
            spot = {"world": b, "max_abs_dg": dg, "max_abs_djac": dj}
        nlp.solve()
        nlp.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
        build_ms = nlp.build_ms
        t1 = time.perf_counter(); sols = nlp.solve(); first_ms = (time.perf_counter() - t1) * 1e3
        rep = []
        for _ in range(3):
            t1 = time.perf_counter(); nlp.solve(); rep.append((time.perf_counter() - t1) * 1e3)
        out = {"problems": B, "obstacles": int(bp["obstacles"].shape[1]), "constraints_m": m, "reach_sets_ms": build_ms, "eval_us": eval_us,
               "first_solve_ms": first_ms, "repeated_solve_ms": min(rep), "feasible": int(sum(int(s2["feasible"]) for s2 in sols)),
               "sqp_iterations_sum": int(sum(s2["iterations"] for s2 in sols)), "sqp_iterations_max": int(max(s2["iterations"] for s2 in sols)),
               "evaluations_sum": int(sum(s2["evaluations"] for s2 in sols)), "oracle_spot_check": spot}
        nlp.close()
        return out

    def lone_figures(problems):
        nlp = _opts(ArmourNLP(T=T, device=device))
        build, host, devf, auto, feas, its = [], [], [], [], 0, []
        for p in problems:
            for _ in range(2):
                nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
            build.append(nlp.build_ms)
            for kind, acc in (("host", host), ("device", devf), ("auto", auto)):
                best = 1e30
                for _ in range(3):
                    t1 = time.perf_counter()
                    sols = nlp.solve(host_qp=(kind == "host"), device_qp=(kind == "device"))
                    best = min(best, (time.perf_counter() - t1) * 1e3)
                acc.append(best)
            feas += int(sols[0]["feasible"]); its.append(int(sols[0]["iterations"]))
        nlp.close()
        med = statistics.median
        return {"problems": len(problems), "reach_sets_ms_median": med(build), "reach_sets_ms_max": max(build),
                "solve_ms_median": {"host_form": med(host), "device_form": med(devf), "automatic": med(auto)},
                "solve_ms_max": {"host_form": max(host), "device_form": max(devf), "automatic": max(auto)},
                "planning_iteration_ms_median": med([a + b for a, b in zip(build, auto)]), "feasible": feas, "sqp_iterations_median": med(its), "sqp_iterations_max": max(its)}

    ws = reference_worlds()
    bp = as_batch(ws)
    B, O = bp["obstacles"].shape[:2]
    syn = random_batch(7000, B, O)
    return {"reference_worlds_one_batch": batch_figures(bp, spot_world=103 if check else None),
            "reference_worlds_one_at_a_time": lone_figures([p for _, p in ws]),
            "synthetic_worlds_one_batch": batch_figures(syn),
            "synthetic_worlds_one_at_a_time": lone_figures([{k: syn[k][b] for k in ("q0", "qd0", "qdd0", "q_des", "obstacles")} for b in range(B)]),
            "note": "107 worlds = 100 saved random scenes (5..14 boxes) + 7 hard scenarios (1..12 boxes), first planning iteration (arm at rest at `start`, "
                    "q_des = the straight-line waypoint); the synthetic worlds are random_batch(7000, 107, 14) of armour_amd/worlds.py (random state and "
                    "velocity, boxes anywhere in reach); solve times are wall clock around the synchronous call, best of three"}


def fetch8_child(args):
    """BASELINE configs[4] as it reads -- "Fetch 8-DOF arm with payload-mass uncertainty, 100 obstacles" (never `value`): the Fetch arm behind a torso
    yaw joint (include/armour_robot_fetch.h: 9 links, 8 factors), +-50 % mass / inertia on the gripper link, one planning problem, O = 100, T = 100.
    Eight factors need the 128-bit-key ABI (the reference's key holds seven: RT/PZsparse.h:8-21), which is chosen per process: this function is the
    body of a CHILD process started with ARMOUR_KEY128=1 (libarmour_hip_k128.so / liboracle_k128.so); it prints one JSON object."""
    import numpy as np
    import torch
    from armour_amd import _lib
    from armour_amd.planner import ArmourNLP, default_params, fetch8_robot
    from armour_amd.worlds import random_fetch8_problem
    assert _lib.MAXF == 8, "fetch8_child runs with ARMOUR_KEY128=1"
    T, O, K, R = args.time_steps, 100, max(20, min(args.steps, 40)), max(1, args.repeats)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    p = random_fetch8_problem(11, O)

    def params(mod):
        pr = mod(T)
        pr.k_range[7] = pr.k_range[6]
        return pr
    nlp = _opts(ArmourNLP(robot=fetch8_robot(0.5), params=params(default_params), device=0))
    ms = []
    for _ in range(4):
        nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
        ms.append(nlp.build_ms)
    tm = Timed(nlp, dev, 311, K, 4, False)
    wall, ev = tm.run(R)
    spot = None
    if not args.no_check:
        from oracle import cpu_oracle as orc
        o = orc.Oracle(robot=orc.fetch8_robot(0.5), params=params(orc.default_params)).set_problem(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
        k_last = tm.ks[-1].cpu().numpy()
        g_ref, jac_ref = o.eval_g_jac(k_last[0])
        dg = float(np.abs(tm.d_g.cpu().numpy()[0] - g_ref).max())
        dj = float(np.abs(tm.d_jac.cpu().numpy()[0] - jac_ref).max())
        assert dg <= 1e-9 and dj <= 1e-8, (dg, dj)
        spot = {"max_abs_dg": dg, "max_abs_djac": dj, "oracle_build_ms": o.build_ms, "oracle_pair_products": int(o.stats()["mul_pairs"])}
    out = summarise(nlp, wall, ev, K, 1, {"robot": "fetch 8-DOF: torso yaw + the 7 arm joints of CMP/FetchInfo.h + gripper link with +-50 % mass / inertia (include/armour_robot_fetch.h)",
                                          "key_bits": 128, "links": nlp.J, "factors": nlp.n, "obstacles": O, "constraints_m": nlp.m,
                                          "p1_set_problems_ms": min(ms[1:]), "p1_kernel": nlp.build_info(), "oracle_spot_check": spot,
                                          "table_sizes": nlp.table_sizes()}, traffic_name="configs4_8factor")
    nlp.close()
    print("FETCH8_JSON " + json.dumps(out), flush=True)


def fetch8_config(args):
    """Parent side: run fetch8_child in a process of its own (the ABI is per process) and return its JSON, or the reason there is none."""
    env = dict(os.environ, ARMOUR_KEY128="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "ARMOUR_HIP_LIB"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.abspath(__file__), "--fetch8-child", "--steps", str(args.steps), "--repeats", str(args.repeats), "--time-steps", str(args.time_steps)]
    cmd += ["--no-check"] if (args.no_check or args.no_sync_probe) else []
    for opt, val in OPTIONS:
        cmd += ["--set-option", f"{opt}={val}"]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        for line in r.stdout.splitlines():
            if line.startswith("FETCH8_JSON "):
                return json.loads(line[len("FETCH8_JSON "):])
        return {"error": (r.stderr or r.stdout)[-400:]}
    except Exception as e:   # (the headline must not depend on this extra)
        return {"error": repr(e)}


# ----------------------------------------------------------------------------------------------- dry run (CPU, gloo)
def dry_run(args, rank, world):
    """Development / test mode: the launcher, the rendezvous (gloo) and the timing reduction of the real run with the GPU
    work replaced by a sleep.  Used by tests/test_host_logic.py on a CPU-only host; prints the same JSON shape."""
    import torch.distributed as dist
    from armour_amd.sharding import gather_counts, reduce_max_elapsed, shard_seeds
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    seeds = shard_seeds(0, world * args.batch, rank, world)
    wall = []
    for _ in range(args.repeats):
        if use_dist:
            dist.barrier()
        t0 = time.perf_counter()
        time.sleep(1e-4 * args.steps * (1 + rank))   # the slowest rank sets the interval
        el = time.perf_counter() - t0
        if use_dist:
            dist.barrier()
        wall.append(reduce_max_elapsed(el))
    counts = gather_counts(len(seeds))
    if rank == 0:
        med = statistics.median(wall)
        print(json.dumps({"metric": METRIC, "value": sum(counts) * args.steps / med, "unit": "iters/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": med * 1e3 / args.steps,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "dry-run (no GPU work)",
                          "config": {"workload": "dry run", "batch_per_gpu": args.batch, "problems_per_rank": counts},
                          "repeats": args.repeats}), flush=True)
    if use_dist:
        dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--repeats", type=int, default=9, help="timed intervals of --steps steps each; the median is reported")
    ap.add_argument("--batch", type=int, default=1, help="planning problems per GPU (1 = BASELINE configs[1])")
    ap.add_argument("--obstacles", type=int, default=20)
    ap.add_argument("--time-steps", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="development (ablated kernel builds): skip the output sanity check")
    ap.add_argument("--headline-only", action="store_true", help="skip the extra measurements (other configs, multi-point, planning iteration)")
    ap.add_argument("--no-sync-probe", action="store_true",
                    help="for runs under rocprofv3: no synchronous host-pointer calls at all (they add PCIe-bound launches of the same "
                         "kernel to the trace), i.e. skip the host-entry check and the sync-latency probe")
    ap.add_argument("--set-option", action="append", default=[], metavar="ID=VALUE", help="development: armour_set_option on every handle (A/B runs)")
    ap.add_argument("--dry-run", action="store_true", help="development / tests: launcher + rendezvous + reduction on CPU (gloo), no GPU work")
    ap.add_argument("--cpu-baseline-child", action="store_true", help="internal: the body of the cpu_baseline measurement (a process started with OMP_WAIT_POLICY=passive etc.; no GPU)")
    ap.add_argument("--fetch8-child", action="store_true", help="internal: the body of the configs[4] 8-factor measurement (a process with ARMOUR_KEY128=1)")
    ap.add_argument("--rank-devices", default="", metavar="D0,D1,...", help="development / tests: device ordinal of every rank (default: rank r on device r); "
                    "several ranks on one device need --dist-backend gloo (RCCL refuses two ranks on one GPU)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend of the barrier and the MAX reduction (nccl = RCCL)")
    args = ap.parse_args()
    for kv in args.set_option:
        OPTIONS.append((int(kv.split("=")[0]), float(kv.split("=")[1])))

    if args.cpu_baseline_child:
        print("CPU_BASELINE_JSON " + json.dumps(cpu_baseline_body(args.time_steps, args.obstacles, seed=0)), flush=True)
        return
    if args.fetch8_child:
        return fetch8_child(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))   # the parent: no torch, no HIP

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if args.dry_run:
        return dry_run(args, rank, world)

    import numpy as np
    import torch
    import torch.distributed as dist
    from armour_amd.planner import ArmourNLP
    from armour_amd.sharding import shard_seeds
    from armour_amd.worlds import random_batch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU path)")
    if args.rank_devices:
        local_rank = int(args.rank_devices.split(",")[rank])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ   # under a launcher even a single rank goes through RCCL
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    red_dev = dev if (use_dist and args.dist_backend == "nccl") else None   # where the reduction's tensor lives (gloo: host memory)
    B, O, T, K, W, R = args.batch, args.obstacles, args.time_steps, args.steps, args.warmup, max(1, args.repeats)
    # independent worlds per rank (block partition of world*B seeds, armour_amd/sharding.py); no data-path collective
    seeds = shard_seeds(0, world * B, rank, world)
    probs = random_batch(seeds[0], len(seeds), O)
    nlp = _opts(ArmourNLP(T=T, device=local_rank))
    nlp.set_parameters(probs["q0"], probs["qd0"], probs["qdd0"], probs["q_des"], probs["obstacles"])  # loads the code objects
    t0 = time.time()
    nlp.set_parameters(probs["q0"], probs["qd0"], probs["qdd0"], probs["q_des"], probs["obstacles"])
    p1_wall_ms = (time.time() - t0) * 1e3
    p1_dev_ms = nlp.build_ms

    n, m = nlp.n, nlp.m

    def sync_probe():
        # synchronous single-call latency as an IPOPT host loop would see it (H2D k, launch, D2H g+jac); VERDICT r5 item 3: every leg
        # with one untimed call first, the interpreter's garbage collector held off (a generation-2 pass of a process with torch loaded
        # is ~40 ms: DESIGN.md section 5), per-call times, and median / mean / max / index of the slowest call in the line
        import gc
        k1 = np.full((B, n), 0.25)

        def leg(call, reps):
            call()
            tt = []
            gc.collect(); gc.disable()
            try:
                for _ in range(reps):
                    t1 = time.perf_counter()
                    call()
                    tt.append((time.perf_counter() - t1) * 1e6)
            finally:
                gc.enable()
            return tt

        g_own, jac_own = np.zeros((B, m)), np.zeros((B, m, n))       # the caller's own arrays, reused: an IPOPT TNLP's `g` and `values`
        tp = leg(lambda: nlp.eval_g_jac(k1, out=(g_own, jac_own)), 50)
        tf = leg(lambda: nlp.eval_g_jac(k1), 20)                     # fresh numpy arrays per call (never-touched pages every time)
        tt = leg(lambda: nlp.eval_g_jac(k1, pinned=True), 50)
        tv = leg(lambda: nlp.eval_violations(k1), 50)                # the reduced-output entry: k in, one 32-byte record per problem out
        return {"pageable": statistics.median(tp), "pageable_mean": sum(tp) / len(tp), "pageable_max": max(tp), "pageable_argmax": tp.index(max(tp)),
                "pageable_fresh_arrays": statistics.median(tf), "pageable_fresh_arrays_mean": sum(tf) / len(tf), "pageable_fresh_arrays_max": max(tf),
                "pinned": statistics.median(tt), "violations": statistics.median(tv), "pinned_mean": sum(tt) / len(tt), "pinned_max": max(tt), "pinned_argmax": tt.index(max(tt)),
                "note": "medians of 50 calls after one untimed call, garbage collector held off; pageable = armour_eval_g_jac(h, x, g, values) on the caller's own "
                        "reused arrays; pageable_fresh_arrays = new numpy arrays per call (20 calls); pinned = buffers from armour_alloc_pinned (k in, g | jac out as "
                        "asynchronous DMA transfers on the handle's stream)"}

    probe_early = {}
    if os.environ.get("BENCH_PROBE_POINTS") and rank == 0:   # development: where in this process does the page-locked call get slow?
        probe_early["after_set_parameters"] = sync_probe()
    tm = Timed(nlp, dev, rank, K, W, use_dist, red_dev)
    if os.environ.get("BENCH_PROBE_POINTS") and rank == 0:
        probe_early["after_graph_built"] = sync_probe()
    wall, ev = tm.run(R)
    if os.environ.get("BENCH_PROBE_POINTS") and rank == 0:
        probe_early["after_timed_run"] = sync_probe()
    if not args.no_check and not args.no_sync_probe:
        tm.check()   # finite, and the synchronous host entry reproduces the device entry bit for bit

    from armour_amd.sharding import gather_counts
    counts = gather_counts(len(seeds), device=red_dev)   # what every rank built and evaluated (a collective: all ranks call it)
    out = None
    if rank == 0:
        s = summarise(nlp, wall, ev, K, world)
        cfg_id = 1 if (B, O, T) == (1, 20, 100) else 2 if (B, O, T) == (128, 50, 100) else 3 if (B, O, T) == (128, 20, 100) else "custom"
        out = {
            "metric": METRIC, "value": sum(counts) * K / statistics.median(wall), "unit": "iters/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": s["ms_per_step"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"Kinova Gen3 7-DOF, {O} obstacles, {T} time steps, {B} planning problem(s) per GPU "
                                   f"(BASELINE configs[{cfg_id}]), one fused eval_g+eval_jac_g launch per step at a fresh k",
                       "robot": "kinova_gen3_7dof_no_gripper", "batch_per_gpu": B, "obstacles": O, "time_steps": T,
                       "constraints_m": m, "parallelism": f"independent worlds x{world}, no collective", "problems_per_rank": counts,
                       "dist_backend": (args.dist_backend if use_dist else None), "rank_devices": args.rank_devices or None},
            "timing": {"repeats": R, "statistic": "median of the max-over-ranks wall time of K steps (barrier + synchronize on both sides)",
                       "wall_ms_min_med_max": s["wall_ms_min_med_max"]},
            "roofline": {"bound": "hbm", "achieved": s["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": s["algorithmic_equivalent_frac"], "traffic": s["traffic"]["bytes"] if s["traffic"] else None,
                         "achieved_definition": "ALGORITHMIC bytes of the reference formulation (SURVEY.md 8d: 1440 B per collision row + PZ tables + outputs) / mean launch time "
                                                "by HIP events -- an EQUIVALENT rate: the kernel reads a lean table (480 B per row), so this figure is not the bandwidth "
                                                "moved and may exceed the peak on a large batch; bytes actually moved: achieved_by_traffic_GBps / frac_by_traffic (rocprofv3 "
                                                "counters of the committed profile) and effective_GBps / frac_effective (the live layout's bytes)",
                         "algorithmic_equivalent_GBps": s["achieved_GBps"],
                         "achieved_by_traffic_GBps": (s["traffic"]["bytes"] / (s["launch_us"] * 1e-6) / 1e9) if s["traffic"] else None,
                         "effective_GBps": s["effective_GBps"],
                         "traffic_source": s["traffic"], "frac_by_traffic": s["frac_by_traffic"],
                         "effective_bytes_per_launch": s["effective_bytes_per_launch"], "frac_effective": s["frac_effective"],
                         "kernel": nlp.L.armour_p2_kernel_name().decode(), "algorithmic_bytes_per_launch": s["algorithmic_bytes_per_launch"],
                         "launch_us": s["launch_us"]},
            "p1_set_problems_ms": {"device": p1_dev_ms, "wall": p1_wall_ms, "per_problem_device": p1_dev_ms / B},
            "table_sizes": nlp.table_sizes(),
        }
        if not args.no_sync_probe:
            out["sync_host_call_us"] = sync_probe()
            if probe_early:
                out["sync_host_call_us_at"] = probe_early
        if not args.headline_only:
            # extra (never `value`): P points of the same problems per launch, tables held in registers across points
            P = 16
            L = max(1, min(K // P, 64))
            d_gm = torch.empty((P, B, m), device=dev, dtype=torch.float64)
            d_jm = torch.empty((P, B, m, n), device=dev, dtype=torch.float64)
            kp = tm.ks[:P].contiguous()
            sh = tm.stream.cuda_stream
            nlp.eval_g_jac_device_multi(kp.data_ptr(), P, d_gm.data_ptr(), d_jm.data_ptr(), sh)
            torch.cuda.synchronize()
            m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            m0.record(tm.stream)
            for _ in range(L):
                nlp.eval_g_jac_device_multi(kp.data_ptr(), P, d_gm.data_ptr(), d_jm.data_ptr(), sh)
            m1.record(tm.stream)
            torch.cuda.synchronize()
            mus = m0.elapsed_time(m1) * 1e3 / L
            out["multi_point_launch"] = {"points_per_launch": P, "launch_us": mus, "point_evals_per_s": P * B / (mus * 1e-6),
                                         "us_per_point": mus / P,
                                         "note": "armour_eval_g_jac_device_multi: one launch evaluates P trial points of the same "
                                                 "problems, tables read once; bit-identical to P single launches"}
            del d_gm, d_jm
        if world == 1 and not args.headline_only:
            # extra (never `value`): one whole planning iteration -- reach-set build + NLP solve -- of the reference's own
            # sample problem (RT/armour_main.cu:18-33), and the same for 128 copies with perturbed goals in lock step
            from armour_amd.worlds import reference_sample_problem
            sp = reference_sample_problem()
            plan = {}
            for pb in (1, 128):
                st = {k: np.stack([np.asarray(sp[k], dtype=float) + (0.002 * i if k == "q_des" else 0.0) for i in range(pb)]) for k in sp}
                pn = ArmourNLP(T=T, device=local_rank)
                for _ in range(2):
                    pn.set_parameters(st["q0"], st["qd0"], st["qdd0"], st["q_des"], st["obstacles"])
                    t1 = time.perf_counter()
                    sols = pn.solve()
                    solve_ms = (time.perf_counter() - t1) * 1e3
                plan[f"{pb} problem(s)"] = {"reach_sets_ms": pn.build_ms, "solve_ms": solve_ms,
                                             "evaluations": int(sols[0]["evaluations"]), "feasible": int(sum(int(s2["feasible"]) for s2 in sols))}
                pn.close()
            out["planning_iteration_sample_problem"] = plan
    nlp_done = (B, O, T) == (1, 20, 100) and not args.headline_only
    if nlp_done:
        # the other BASELINE configs, measured the same way by every rank (extras, never `value`):
        #   N = 1: configs[2] = 128 random worlds at O = 50 on one GPU, outputs checked (incl. two problems against the oracle)
        #   N > 1: configs[3] = 128 worlds per GPU at O = 20 (1024 over 8 GPUs), outputs checked
        KX = max(4, min(K, 40))
        chk = None if args.no_check or args.no_sync_probe else True
        if world == 1:
            oc = {"configs[2]: O=50, batch 128, T=100": extra_config(local_rank, dev, rank, world, use_dist, 128, 50, T, KX, R, chk, red_dev),
                  "configs[4]: Fetch, payload +-50 %, O=100, batch 1, T=100": fetch_config(local_rank, dev, T, max(KX, 20), R, chk is not None),
                  "configs[4] 8-factor: Fetch 8-DOF (torso yaw + arm), payload +-50 %, O=100, batch 1, T=100, 128-bit keys": fetch8_config(args),
                  "reference scenes": reference_scenes_config(local_rank, dev, T, chk is not None)}
        else:
            oc = {f"configs[3]: O=20, batch 128 per GPU ({128 * world} worlds over {world} GPUs), T=100":
                  extra_config(local_rank, dev, rank, world, use_dist, 128, 20, T, KX, R, None if chk is None else False, red_dev)}
        if rank == 0:
            out["other_configs"] = oc
    if rank == 0:
        if world == 1 and not args.headline_only and (B, O, T) == (1, 20, 100):
            out["p1"] = p1_accounting(local_rank, T, O, p1_dev_ms, count_pairs=not args.no_cpu_baseline)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(T, O, seed=0)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
