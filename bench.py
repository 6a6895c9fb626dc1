#!/usr/bin/env python3
"""Benchmark of ARMOUR's constraint callback on MI355X: planning iters/s (fused eval_g + eval_jac_g).

    python bench.py --gpus N --steps K --warmup W [--batch B] [--obstacles O] [--time-steps T]

One "step" = one pass of the hot path (RT/NLPclass.cu:272-396: eval_g + eval_jac_g at a fresh k) over this
rank's batch of planning problems; one kernel launch per step, inputs (reach-set tables from
armour_set_problems, the k points) resident in HBM before the timed region.  The default workload is
BASELINE.json configs[1]: Kinova Gen3 7-DOF, 20 obstacles, 100 time steps, a single planning problem per
GPU.  Ranks (one per GPU, launched by torch.distributed.run) hold independent random worlds: the path
shards with no collective, so scaling is "weak" and `value` = problems*steps over all ranks / max-rank time.

The JSON line also carries
  roofline     HBM roofline of the P2 kernel: algorithmic bytes per launch / average launch
               duration measured with HIP events on the launch stream over the timed region;
  cpu_baseline the CPU oracle (oracle/, a port of the reference's host path) timed on this box's cores on a
               bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "planning iters/s (eval_g+jac) Kinova 7-DOF, 100 timesteps × 20 obstacles"
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def cpu_baseline(T, O, seed, budget_s=12.0):
    """Oracle (kind 'port') on the host cores: P2 evals/s on the same world, bounded to ~budget_s seconds."""
    from oracle.cpu_oracle import Oracle, max_threads
    from armour_amd.worlds import random_k, random_problem
    p = random_problem(seed, O)
    o = Oracle(T=T).set_problem(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
    ks = random_k(seed, 64)
    # thread count: the reference's default of 32 (NUM_THREADS, RT/Parameters.h:35), all cores and a few in between
    # (T = 100 time steps bound the useful parallelism, and a shared host punishes oversubscription); keep the fastest
    best_t, best_rate = 0, 0.0
    for th in sorted({min(c, max_threads()) for c in (16, 32, 64, 128, max_threads())}):
        o.time_eval(ks, 3, threads=th)  # warm-up
        rate = 16 / o.time_eval(ks, 16, threads=th)
        if rate > best_rate:
            best_t, best_rate = th, rate
    reps = int(max(20, min(20000, budget_s * best_rate)))
    secs = o.time_eval(ks, reps, threads=best_t)
    return {
        "value": reps / secs, "unit": "iters/s", "cores": best_t, "kind": "port",
        "sample": f"{reps} fused eval_g+eval_jac_g of 1 world (seed {seed}, O={O}, T={T}) at 64 cycling k points, "
                  f"OpenMP over time steps as RT/NLPclass.cu:304,376; reach-set build {o.build_ms:.0f} ms on the same cores",
        "p1_build_ms": o.build_ms,
    }


def measured_traffic(B, O, T):
    """HBM bytes per P2 launch from the committed rocprofv3 PMC passes of this same command (profiles/*_pmc.json:
    2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md).  None when no profile of this config exists."""
    path = os.path.join(ROOT, "profiles", "r01_bench_headline_pmc.json")
    if (B, O, T) != (1, 20, 100) or not os.path.exists(path):
        return None
    try:
        return float(json.load(open(path))["p2_hbm_traffic_bytes_per_launch"])
    except Exception:
        return None


def other_configs(device):
    """BASELINE configs[2] (O=50, 128 random worlds on one GPU: the HBM-bound regime) measured the same way, reported
    next to the headline line (not as `value`)."""
    import torch
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch, random_k
    out = {}
    for name, B, O, T, K in (("configs[2]: O=50, batch 128, T=100", 128, 50, 100, 40),):
        dev = torch.device("cuda", device)
        probs = random_batch(1000, B, O)
        nlp = ArmourNLP(T=T, device=device).set_parameters(probs["q0"], probs["qd0"], probs["qdd0"], probs["q_des"], probs["obstacles"])
        ks = torch.tensor(random_k(77, (K + 4) * B).reshape(K + 4, B, nlp.n), device=dev)
        d_g = torch.empty((B, nlp.m), device=dev, dtype=torch.float64)
        d_jac = torch.empty((B, nlp.m, nlp.n), device=dev, dtype=torch.float64)
        st = torch.cuda.Stream(device=dev)
        nlp.prepare_steps(ks[4:].data_ptr(), K, d_g.data_ptr(), d_jac.data_ptr())
        nlp.eval_g_jac_device_steps(ks.data_ptr(), 4, d_g.data_ptr(), d_jac.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        nlp.eval_g_jac_device_steps(ks[4:].data_ptr(), K, d_g.data_ptr(), d_jac.data_ptr(), st.cuda_stream)
        e1.record(st)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / K
        b_alg = nlp.algorithmic_bytes()
        out[name] = {"problem_evals_per_s": B / (us * 1e-6), "launch_us": us, "algorithmic_bytes_per_launch": b_alg,
                     "achieved_GBps": b_alg / (us * 1e-6) / 1e9, "frac_of_hbm_peak": b_alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                     "p1_set_problems_ms_per_problem": nlp.build_ms / B}
        nlp.close()
        del d_g, d_jac, ks
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=1, help="planning problems per GPU (1 = BASELINE configs[1])")
    ap.add_argument("--obstacles", type=int, default=20)
    ap.add_argument("--time-steps", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="development (ablated kernel builds): skip the output sanity check")
    ap.add_argument("--headline-only", action="store_true", help="skip the extra configs[2] measurement (used under rocprofv3)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")

    import torch
    import torch.distributed as dist
    from armour_amd.planner import ArmourNLP
    from armour_amd.sharding import shard_seeds
    from armour_amd.worlds import random_batch, random_k

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ   # under torch.distributed.run even a single rank goes through RCCL
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    B, O, T, K, W = args.batch, args.obstacles, args.time_steps, args.steps, args.warmup
    # independent worlds per rank (block partition of world*B seeds, armour_amd/sharding.py); no data-path collective
    seeds = shard_seeds(0, world * B, rank, world)
    probs = random_batch(seeds[0], len(seeds), O)
    nlp = ArmourNLP(T=T, device=local_rank)
    nlp.set_parameters(probs["q0"], probs["qd0"], probs["qdd0"], probs["q_des"], probs["obstacles"])  # loads the code objects
    t0 = time.time()
    nlp.set_parameters(probs["q0"], probs["qd0"], probs["qdd0"], probs["q_des"], probs["obstacles"])
    p1_wall_ms = (time.time() - t0) * 1e3
    p1_dev_ms = nlp.build_ms

    n, m = nlp.n, nlp.m
    ks = torch.tensor(random_k(rank, (K + W) * B).reshape(K + W, B, n), device=dev)  # a fresh k per step and problem
    d_g = torch.empty((B, m), device=dev, dtype=torch.float64)
    d_jac = torch.empty((B, m, n), device=dev, dtype=torch.float64)
    stream = torch.cuda.Stream(device=dev)  # a real (non-null) stream: HIP events on the null stream do not bracket the launches
    sh = stream.cuda_stream
    torch.cuda.synchronize()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # the K timed launches go out as one instantiated graph (a chain of K kernel nodes): build it before the timed region
    nlp.prepare_steps(ks[W:].data_ptr(), K, d_g.data_ptr(), d_jac.data_ptr())
    # warm-up (untimed)
    nlp.eval_g_jac_device_steps(ks.data_ptr(), W, d_g.data_ptr(), d_jac.data_ptr(), sh)
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_start = time.perf_counter()
    e0.record(stream)
    nlp.eval_g_jac_device_steps(ks[W:].data_ptr(), K, d_g.data_ptr(), d_jac.data_ptr(), sh)
    e1.record(stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    barrier()
    ev_ms = e0.elapsed_time(e1)

    el = torch.tensor([elapsed, ev_ms], device=dev, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed_max, ev_ms_max = float(el[0]), float(el[1])

    # sanity: the last step's outputs are finite and the synchronous host entry agrees with the device entry
    g_host, jac_host = nlp.eval_g_jac(ks[-1].cpu().numpy())
    if not args.no_check:
        assert np.isfinite(g_host).all() and np.isfinite(jac_host).all()
        assert np.array_equal(g_host, d_g.cpu().numpy()) and np.array_equal(jac_host, d_jac.cpu().numpy())

    if rank == 0:
        b_alg = nlp.algorithmic_bytes()          # bytes one launch must move (all B problems)
        launch_us = ev_ms_max * 1e3 / K          # average launch duration over the timed region (HIP events)
        achieved = b_alg / (launch_us * 1e-6) / 1e9
        # synchronous single-call latency as an IPOPT host loop would see it (H2D k, launch, D2H g+jac)
        k1 = ks[0].cpu().numpy()
        t1 = time.perf_counter()
        for _ in range(20):
            nlp.eval_g_jac(k1)
        sync_us = (time.perf_counter() - t1) / 20 * 1e6
        nlp.eval_g_jac(k1, pinned=True)        # allocates the page-locked buffers
        t1 = time.perf_counter()
        for _ in range(50):
            nlp.eval_g_jac(k1, pinned=True)
        sync_pinned_us = (time.perf_counter() - t1) / 50 * 1e6
        out = {
            "metric": METRIC, "value": world * B * K / elapsed_max, "unit": "iters/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": elapsed_max * 1e3 / K, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"Kinova Gen3 7-DOF, {O} obstacles, {T} time steps, {B} planning problem(s) per GPU "
                                   f"(BASELINE configs[{1 if (B, O, T) == (1, 20, 100) else 2 if (B, O) == (128, 50) else 'custom'}]), "
                                   "one fused eval_g+eval_jac_g launch per step at a fresh k",
                       "robot": "kinova_gen3_7dof_no_gripper", "batch_per_gpu": B, "obstacles": O, "time_steps": T,
                       "constraints_m": m, "parallelism": f"independent worlds x{world}, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(B, O, T),
                         "kernel": nlp.L.armour_p2_kernel_name().decode(), "algorithmic_bytes_per_launch": b_alg,
                         "launch_us": launch_us},
            "p1_set_problems_ms": {"device": p1_dev_ms, "wall": p1_wall_ms, "per_problem_device": p1_dev_ms / B},
            "sync_host_call_us": {"pageable": sync_us, "pinned": sync_pinned_us},
            "table_sizes": nlp.table_sizes(),
        }
        if not args.headline_only:
            # extra (never `value`): P points of the same problems per launch, tables held in registers across points
            P = 16
            L = max(1, min(K // P, 64))
            d_gm = torch.empty((P, B, m), device=dev, dtype=torch.float64)
            d_jm = torch.empty((P, B, m, n), device=dev, dtype=torch.float64)
            kp = ks[:P].contiguous()
            nlp.eval_g_jac_device_multi(kp.data_ptr(), P, d_gm.data_ptr(), d_jm.data_ptr(), sh)
            torch.cuda.synchronize()
            m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            m0.record(stream)
            for _ in range(L):
                nlp.eval_g_jac_device_multi(kp.data_ptr(), P, d_gm.data_ptr(), d_jm.data_ptr(), sh)
            m1.record(stream)
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
                                             "evaluations": int(sols[0]["evaluations"]), "feasible": int(sum(int(s["feasible"]) for s in sols))}
                pn.close()
            out["planning_iteration_sample_problem"] = plan
        if world == 1 and (B, O, T) == (1, 20, 100) and not args.headline_only:
            out["other_configs"] = other_configs(local_rank)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(T, O, seed=0)
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
