"""One process that runs ONE named workload a few times -- what a rocprofv3 pass profiles (every dispatch of the process belongs to it).

    python tools/workload.py p1 B [O] [id=value ...]   reach-set build of B random worlds (T = 100, O = 20), three builds
    python tools/workload.py c4 [K]                    BASELINE configs[4] (Fetch preset, payload +-50 %, O = 100): K fused evaluations x 3
    python tools/workload.py c4_8f [K]                 the 8-factor configs[4] (Fetch 8-DOF; needs ARMOUR_KEY128=1 in the environment)
    python tools/workload.py solve [N]                 the reference's sample problem solved N times by the device-resident armour_solve
    python tools/workload.py solveb B O [id=value ...] armour_solve of B random worlds (seed 5000), best wall time of 5 per option set; prints the solver rows
    python tools/workload.py cull B O                  armour_eval_violations_device on every row, then on the relevant rows (ARMOUR_OPT_CULL_ROWS)
`id=value` arguments are per-handle options (include/armour_hip.h).  The handles are closed before interpreter exit (under rocprofv3 a handle
freed from the exit handlers crashed inside the tool library)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _opts(nlp, args):
    for kv in args:
        if "=" in kv:
            nlp.set_option(int(kv.split("=")[0]), float(kv.split("=")[1]))
    return nlp


def _steps(nlp, K, seed):
    import torch
    from armour_amd.worlds import random_k
    dev = torch.device("cuda", 0)
    ks = torch.tensor(random_k(seed, K * nlp.B, n=nlp.n).reshape(K, nlp.B, nlp.n), device=dev)
    d_g = torch.empty((nlp.B, nlp.m), device=dev, dtype=torch.float64)
    d_j = torch.empty((nlp.B, nlp.m, nlp.n), device=dev, dtype=torch.float64)
    st = torch.cuda.Stream(device=dev)
    for _ in range(3):
        nlp.eval_g_jac_device_steps(ks.data_ptr(), K, d_g.data_ptr(), d_j.data_ptr(), st.cuda_stream)
    st.synchronize()
    return bool(torch.isfinite(d_g).all())


def main(argv):
    what, args = argv[1], argv[2:]
    from armour_amd.planner import ArmourNLP, default_params
    if what == "p1":
        from armour_amd.worlds import random_batch
        B = int(args[0]); O = int(args[1]) if len(args) > 1 and "=" not in args[1] else 20
        pb = random_batch(0, B, O)
        nlp = _opts(ArmourNLP(T=100), args)
        for _ in range(3):
            nlp.set_parameters(pb["q0"], pb["qd0"], pb["qdd0"], pb["q_des"], pb["obstacles"])
        print("p1: B", B, "O", O, "build ms", nlp.build_ms, nlp.build_info())
    elif what == "c4":
        from armour_amd.planner import fetch_robot
        from armour_amd.worlds import random_fetch_problem
        K = int(args[0]) if args else 40
        p = random_fetch_problem(11, 100)
        nlp = ArmourNLP(robot=fetch_robot(0.5), params=default_params(100)).set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
        print("configs[4]:", K, "steps x 3, m =", nlp.m, "finite:", _steps(nlp, K, 311))
    elif what == "c4_8f":
        from armour_amd.planner import fetch8_robot
        from armour_amd.worlds import random_fetch8_problem
        K = int(args[0]) if args else 40
        p = random_fetch8_problem(11, 100)
        pr = default_params(100); pr.k_range[7] = pr.k_range[6]
        nlp = ArmourNLP(robot=fetch8_robot(0.5), params=pr)
        for _ in range(3):
            nlp.set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
        print("configs[4] 8-factor:", K, "steps x 3, m =", nlp.m, "build ms", nlp.build_ms, "finite:", _steps(nlp, K, 311))
    elif what == "solve":
        from helpers import SAMPLE_PROBLEM as p
        N = int(args[0]) if args else 20
        nlp = ArmourNLP(T=100).set_parameters(p["q0"], p["qd0"], p["qdd0"], p["q_des"], p["obstacles"])
        for _ in range(N):
            s = nlp.solve(device_qp=True)[0]   # (one problem would take the host-QP form by itself: the persistent kernel is what is profiled)
        print("solved", N, "times:", {k: s[k] for k in ("feasible", "iterations", "evaluations", "status")})
    elif what == "solveb":
        import time, gc
        from armour_amd import _lib
        from armour_amd.worlds import random_batch
        B, O = int(args[0]), int(args[1])
        bp = random_batch(5000, B, O)
        nlp = _opts(ArmourNLP(T=100), args).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
        ref = None
        for cull in (0, 1, -1):
            nlp.set_option(_lib.OPT_SOLVE_CULL, cull)
            nlp.solve()
            gc.collect(); gc.disable()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); res = nlp.solve(); ts.append(time.perf_counter() - t0)
            gc.enable()
            key = [(tuple(r["k_opt"]), r["feasible"], r["iterations"], r["evaluations"], r["status"]) for r in res]
            ref = ref or key
            print(f"B={B} O={O} ARMOUR_OPT_SOLVE_CULL={cull}: armour_solve best {min(ts) * 1e3:.3f} ms, median {sorted(ts)[2] * 1e3:.3f} ms; feasible {sum(r['feasible'] for r in res)}, "
                  f"iterations max {max(r['iterations'] for r in res)}; same results as the full form: {key == ref}", flush=True)
        # a FRESH problem set: what the first solve after armour_set_problems costs (the lists are built inside it)
        for cull in (0, 1):
            nlp.set_option(_lib.OPT_SOLVE_CULL, cull)
            ts = []
            for _ in range(3):
                nlp.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
                t0 = time.perf_counter(); nlp.solve(); ts.append(time.perf_counter() - t0)
            print(f"   first solve after set_problems, cull={cull}: best {min(ts) * 1e3:.3f} ms", flush=True)
        mask, cnt, tq, ms = nlp.solver_rows()
        print(f"   solver rows: collision mean {cnt.mean():.0f} of {nlp.J * nlp.T * O} ({cnt.mean() / max(1, nlp.J * nlp.T * O):.3f}), torque rows mean {tq.mean():.1f} of {nlp.n * nlp.T}; masks + lists {ms:.3f} ms")
    elif what == "cull":
        import torch
        from armour_amd import _lib
        from armour_amd.worlds import random_batch, random_k
        B, O = int(args[0]), int(args[1])
        bp = random_batch(1000, B, O)
        dev = torch.device("cuda", 0)
        res = {}
        for tag, cull in (("full", 0), ("culled", 1)):
            nlp = ArmourNLP(T=100).set_option(_lib.OPT_CULL_ROWS, cull).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
            ks = torch.tensor(random_k(3, 40 * B).reshape(40, B, nlp.n), device=dev)
            rec = torch.zeros((B, 32), dtype=torch.uint8, device=dev)
            st = torch.cuda.Stream(device=dev)
            for i in range(5):
                nlp.eval_violations_device(ks[i].data_ptr(), rec.data_ptr(), st.cuda_stream)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for i in range(40):
                nlp.eval_violations_device(ks[i].data_ptr(), rec.data_ptr(), st.cuda_stream)
            e1.record(st)
            torch.cuda.synchronize()
            rel, cnt, ms = nlp.row_relevance()
            res[tag] = (e0.elapsed_time(e1) * 1e3 / 40, nlp.eval_violations(ks[-1].cpu().numpy()))
            print(f"{tag}: {res[tag][0]:.1f} us per call (B = {B}, O = {O}); relevant collision rows mean {cnt.mean():.0f} of {nlp.J * nlp.T * O}, relevance test {ms:.3f} ms", flush=True)
            nlp.close()
        same = all(a["l1_violation"] == c["l1_violation"] and a["n_violated"] == c["n_violated"] and a["feasible"] == c["feasible"] for a, c in zip(res["full"][1], res["culled"][1]))
        print("records identical:", same, "speed-up", round(res["full"][0] / res["culled"][0], 2))
        return
    else:
        raise SystemExit(__doc__)
    nlp.close()


if __name__ == "__main__":
    main(sys.argv)
