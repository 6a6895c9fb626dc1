"""dev: armour_solve (device-resident form) at B = 128 over obstacle counts and the sub-batching threshold ARMOUR_SOLVE_SUB_TILES (tiles a block may walk
before the batch is cut into co-resident sub-batches; a large value = one launch)."""
import os, subprocess, sys
sys.path.insert(0, '/root/repo')
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import time
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch
    for B, O in ((128, 20), (128, 30), (128, 40), (128, 50), (64, 50), (256, 20)):
        bp = random_batch(11, B, O)
        nlp = ArmourNLP(T=100).set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"])
        nlp.solve()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); nlp.solve(); ts.append((time.perf_counter() - t0) * 1e3)
        print(f"  B={B} O={O}: {min(ts):.2f} ms", flush=True)
        nlp.close()
    sys.exit(0)
for st in ("12", "24", "48", "96", "100000"):
    e = dict(os.environ); e["ARMOUR_SOLVE_SUB_TILES"] = st
    r = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True, timeout=600)
    print(f"ARMOUR_SOLVE_SUB_TILES={st}\n{r.stdout.rstrip()}\n{r.stderr.strip()[-300:] if r.returncode else ''}", flush=True)
