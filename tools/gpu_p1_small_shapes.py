"""dev: per-step reach-set build of small batches: one-wave against three-wave blocks (ARMOUR_P1_WAVES, one child process per setting), build ms best of 3."""
import os, subprocess, sys
sys.path.insert(0, '/root/repo')
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch
    for B in (1, 2, 3, 4, 5, 6, 8, 10):
        bp = random_batch(5, B, 20)
        nlp = ArmourNLP(T=100)
        ms = []
        for _ in range(3):
            nlp.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"]); ms.append(nlp.build_ms)
        print(f"  B={B}: {min(ms):.2f} ms", flush=True); nlp.close()
    sys.exit(0)
for env in ({}, {"ARMOUR_P1_WAVES": "1"}, {"ARMOUR_P1_WAVES": "3"}):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True, timeout=600)
    print(f"{env or 'default'}\n{r.stdout.rstrip()}\n{r.stderr.strip()[-300:] if r.returncode else ''}", flush=True)
