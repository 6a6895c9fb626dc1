"""dev: armour_solve at B = 128, O = 20 / 50 over world seeds: sub-batched device form (default), one launch (ARMOUR_SOLVE_SUB_TILES=100000), host form."""
import os, subprocess, sys
sys.path.insert(0, '/root/repo')
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import time
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch
    for seed in (0, 3, 7, 11):
        for O in (20, 50):
            bp = random_batch(seed, 128, O)
            nlp = ArmourNLP(T=100).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
            host = sys.argv[2] == "host"
            r = nlp.solve(host_qp=host)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); r = nlp.solve(host_qp=host); ts.append((time.perf_counter() - t0) * 1e3)
            print(f"  seed {seed} O={O}: {min(ts):.2f} ms, feasible {sum(x['feasible'] for x in r)}/128, max iterations {max(x['iterations'] for x in r)}", flush=True)
            nlp.close()
    sys.exit(0)
for name, env, mode in (("sub-batched (48)", {}, "dev"), ("one launch", {"ARMOUR_SOLVE_SUB_TILES": "100000"}, "dev"), ("host form", {}, "host")):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, __file__, "child", mode], env=e, capture_output=True, text=True, timeout=900)
    print(f"{name}\n{r.stdout.rstrip()}\n{r.stderr.strip()[-300:] if r.returncode else ''}", flush=True)
